// builder.h — host-side CWBVH construction (CPU).
//
// Stands in for obvhs::cwbvh::builder::{build_cwbvh_from_tris, build_cwbvh}
// (called from src/cwbvh.rs:97,132).  obvhs is an un-vendored dependency, so
// this is an independent builder: binned-SAH BVH2 -> optimal SAH collapse to
// 8-wide nodes (Ylitie et al. 2017, the algorithm obvhs/tinybvh also use) ->
// octant child ordering (embree/src/bvh_embree.rs:284-349) -> 80-byte node
// encoding (embree/src/bvh_embree_to_cwbvh.rs:85-186).
#pragma once
#include <cstdint>
#include <vector>

#include "cwbvh_format.h"

namespace trx {

struct CwBvh {
    std::vector<CwbvhNode> nodes;
    std::vector<uint32_t> primitive_indices;
    std::vector<Aabb> primitive_boxes; // pre-split builds only: the (clipped) box each entry of primitive_indices was built with
    Aabb total_aabb;
    double build_seconds = 0.0;
    float sah_cost = 0.f; // collapse cost of the root / root area
};

struct BuildParams {
    uint32_t max_prims_per_leaf = 3; // CWBVH limit (src/main.rs:176-178)
    float traversal_cost = 1.0f;     // collapse_traversal_cost analogue (src/main.rs:158-163)
    float prim_cost = 0.3f;
    // BVH2 reinsertion pass: fraction of the nodes (largest area first) re-placed
    // per iteration (obvhs `reinsertion_batch_ratio`, src/main.rs:113-118); 0 = off
    float reinsertion_batch_ratio = 0.02f;
    int reinsertion_iterations = 4;
    // false: candidates one at a time, each on the tree the previous move left (this library's own pipeline);
    // true: batches of 128 candidates search the same tree on every core and their moves are applied in order,
    // stale ones skipped (the parallel formulation of the paper, as obvhs runs it) — the ploc_cwbvh pipeline
    bool reinsertion_batched = false;
    // true: ONE batch per iteration - every candidate of an iteration searches the tree the previous iteration left (the
    // paper's own formulation); the searches run on `ploc_device` when that is >= 0 (reinsert_gpu.cpp), else on the host
    // cores, with the same result.  Takes precedence over reinsertion_batched.
    bool reinsertion_whole_iterations = false;
    // pre-splitting (obvhs pre_split / --split): up to this fraction of extra triangle references, spent on
    // the triangles whose boxes are emptiest; 0 = off (the reference's default)
    float pre_split_ratio = 0.0f;
    // BVH2 stage.  0: top-down binned SAH (this builder's own).  > 0: PLOC (Meister & Bittner 2018, the BVH2 stage of
    // obvhs' ploc builder): bottom-up merging of mutual nearest neighbours found within this many places either
    // side in Morton order — obvhs `ploc_search_distance`, the reference's --search-distance (src/main.rs:85-92).
    uint32_t ploc_search_distance = 0;
    // PLOC: the first this-many merge rounds search one place either side only (obvhs `search_depth_threshold`,
    // "Below this depth a search distance of 1 will be used for ploc", src/main.rs:93-98)
    uint32_t ploc_search_depth_threshold = 2;
    int ploc_device = -1;         // >= 0: PLOC (sort + merge rounds) of builds with >= kDevicePlocMinPrims primitives runs on this HIP device
    uint32_t ploc_sort_bits = 64; // Morton code width: 64 (21 bits per axis) or 128 (42 bits per axis), --sort-precision
    int sah_bins = 32;       // BVH2: SAH bins per axis (2..32)
    uint32_t sweep_max = 48; // BVH2: ranges of at most this many primitives get the exact SAH sweep (<= 64)
    int threads = 0; // <= 0: hardware_concurrency
};

constexpr uint64_t kDevicePlocMinPrims = 32768;

// Host cores this process may really use: hardware threads, capped by the CPU affinity mask and the cgroup CPU quota
// (a container that sees 256 hardware threads but owns 16 must not start 256 workers).
int usable_threads();

// Build over arbitrary primitive boxes (TLAS path, src/cwbvh.rs:114,132).
void build_cwbvh_from_aabbs(const Aabb *boxes, uint64_t n, const BuildParams &params, CwBvh &out);
// verts: n * 9 floats (v0, v1, v2).
void build_cwbvh_from_tris(const float *verts, uint64_t n, const BuildParams &params, CwBvh &out);

} // namespace trx
