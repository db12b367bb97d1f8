// collapse_gpu.h — BVH2 -> CWBVH (collapse decisions, emission order, node encoding) on a HIP device (collapse_gpu.cpp),
// called by builder.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "cwbvh_format.h"

namespace trx {

// `nodes`: n_nodes Node2 records (builder.cpp layout, 40 bytes), root at index 0; the layout is otherwise free (children are
// followed through left/right) and `count` need only tell leaves (1) from inner nodes (> 1): it is recomputed here.  On
// return out_nodes / out_prims hold what Collapser::compute_costs + emit_all of builder.cpp write for the same tree in
// pre-order layout - byte for byte (tests/test_gpu_builder.py) - and *root_cost the collapse cost of the root (Decision 0
// of node 0).  `seconds` (may be null) accumulates the kernel time.  false + err on failure.
bool collapse_encode_device(int device, const void *nodes, size_t n_nodes, uint32_t max_prims_per_leaf, float traversal_cost,
                            float prim_cost, std::vector<CwbvhNode> &out_nodes, std::vector<uint32_t> &out_prims,
                            float *root_cost, double *seconds, std::string &err);

} // namespace trx
