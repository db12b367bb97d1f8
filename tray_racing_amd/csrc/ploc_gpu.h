// ploc_gpu.h — the PLOC BVH2 stage on a HIP device (ploc_gpu.cpp), called by builder.cpp.
#pragma once
#include <cstdint>
#include <string>

#include "cwbvh_format.h"

namespace trx {

// Builds the BVH2 over `n` >= 2 boxes on `device`: Morton sort of the centroids at sort_bits (64 | 128), then PLOC
// rounds with the given search radius (1 for the first depth_threshold rounds).  nodes_out receives 2n-1 Node2 records
// (builder.cpp layout: box, left, right, prim, count) in creation order — leaves 0..n-1 in curve order, inner nodes
// after them, every child before its parent, `count` of inner nodes left 0 — and *root_out the root's index.
// Returns false with `err` set on any failure (no device, out of memory): the caller reports it, nothing falls back.
bool ploc_bvh2_device(int device, const Aabb *boxes, const float *centroids, uint32_t n, uint32_t radius,
                      uint32_t depth_threshold, uint32_t sort_bits, void *nodes_out, uint32_t *root_out, double *seconds,
                      std::string &err);

} // namespace trx
