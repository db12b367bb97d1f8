// reinsert_gpu.h — the whole-iteration BVH2 reinsertion pass on a HIP device (reinsert_gpu.cpp), called by builder.cpp.
#pragma once
#include <cstdint>
#include <string>

namespace trx {

// Device-side copy of the tree being optimised and the scratch of the searches; lives for one reinsertion pass.
struct ReinsertDevice;

// Opens the context on `device` for a tree of n_nodes Node2 records (builder.cpp layout, 40 bytes).  false + err on failure.
bool reinsert_dev_open(int device, size_t n_nodes, ReinsertDevice **out, std::string &err);
void reinsert_dev_close(ReinsertDevice *ctx);

// One whole iteration with the tree RESIDENT on the device (reinsert_dev_upload first): the `take` nodes with the largest
// area (largest first, ties by index; the root and its children excluded - Reinserter::select_candidates' order) are chosen
// there (keys + radix sort); each gets the node next to which re-inserting it shrinks the summed area of the inner nodes
// the most - the same search, in the same order, with the same binary32 operations as Reinserter::find in builder.cpp -
// and the moves are applied there too - the set Reinserter::apply_batch applies, found as a fixed point instead of in order
// (reinsert_gpu.cpp says how and why it is the same set) - then the boxes recomputed.  *moved = moves applied.  *to_host =
// true: nothing was applied (a search outgrew its stack, or a move's target came to lie below the node it moves, which only
// the sequential pass resolves): ids[take] / found[take] hold the candidates and their places, the caller downloads the
// tree, applies them in order and uploads the result.  force_host: stop after the searches and hand the iteration over (tests);
// *rounds_out: rounds the fixed point took.
bool reinsert_dev_upload(ReinsertDevice *ctx, const void *nodes, const uint32_t *parent, std::string &err);
bool reinsert_dev_download(ReinsertDevice *ctx, void *nodes, uint32_t *parent, std::string &err);
bool reinsert_dev_iteration_resident(ReinsertDevice *ctx, uint32_t take, uint32_t *ids, uint32_t *found, uint32_t *moved, bool *to_host,
                                     double *seconds, std::string &err, bool force_host = false, uint32_t *rounds_out = nullptr);

constexpr uint32_t kReinsertNone = 0xffffffffu, kReinsertOverflow = 0xfffffffeu;

} // namespace trx
