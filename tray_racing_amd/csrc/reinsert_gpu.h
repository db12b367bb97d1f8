// reinsert_gpu.h — the searches of the BVH2 reinsertion pass on a HIP device (reinsert_gpu.cpp), called by builder.cpp.
#pragma once
#include <cstdint>
#include <string>

namespace trx {

// Device-side copy of the tree being optimised and the scratch of the searches; lives for one reinsertion pass.
struct ReinsertDevice;

// Opens the context on `device` for a tree of n_nodes Node2 records (builder.cpp layout, 40 bytes).  false + err on failure.
bool reinsert_dev_open(int device, size_t n_nodes, ReinsertDevice **out, std::string &err);
void reinsert_dev_close(ReinsertDevice *ctx);

// One batch: the tree as it stands (nodes, parent links) goes to the device and every candidate cand[k] (a node id) gets
// found[k] = the node next to which re-inserting it shrinks the summed area of the inner nodes the most, or 0xffffffff
// when no place is better than where it is — the same search, in the same order, with the same binary32 operations as
// Reinserter::find in builder.cpp.  found[k] = 0xfffffffe: the search ran out of its (fixed) device stack and the caller
// must run it on the host.  `seconds` (may be null) accumulates the kernel time.
bool reinsert_dev_search(ReinsertDevice *ctx, const void *nodes, const uint32_t *parent, const uint32_t *cand, uint32_t n_cand,
                         uint32_t *found, double *seconds, std::string &err);

// One whole iteration: the tree goes to the device, the `take` nodes with the largest area (largest first, ties by index;
// the root and its children excluded - Reinserter::select_candidates' order) are chosen there (keys + radix sort), searched,
// and both the chosen nodes (ids[take]) and their places (found[take]) come back.
bool reinsert_dev_iteration(ReinsertDevice *ctx, const void *nodes, const uint32_t *parent, uint32_t take, uint32_t *ids,
                            uint32_t *found, double *seconds, std::string &err);

constexpr uint32_t kReinsertNone = 0xffffffffu, kReinsertOverflow = 0xfffffffeu;

} // namespace trx
