// cwbvh_format.h — byte layouts shared by the host builder and the HIP kernels.
//
// CwbvhNode: the 80-byte node of the reference
//   HLSL view   src/rt_gpu/rt_gpu_software_query.hlsl:40-43 (uint4 data[5])
//   field order embree/src/bvh_embree_to_cwbvh.rs:172-185, src/tinybvh.rs:155-168
//   size        embree/src/bvh_embree_to_cwbvh.rs:91, src/rt_gpu/mod.rs:105
// Memory order of the quantised planes is min_x,max_x,min_y,max_y,min_z,max_z
// (q_lo_x = data[2].xy, q_hi_x = data[2].zw: query.hlsl:257-264).
#pragma once
#include <cstdint>

namespace trx {

struct CwbvhNode {
    float p[3];              // data[0].xyz  quantisation origin (node box min)
    uint8_t e[3];            // data[0].w bytes 0..2  biased IEEE exponent per axis
    uint8_t imask;           // data[0].w byte 3      bit s set = slot s is an inner child
    uint32_t child_base_idx; // data[1].x
    uint32_t primitive_base_idx; // data[1].y
    uint8_t child_meta[8];   // data[1].zw
    uint8_t child_min_x[8];  // data[2].xy
    uint8_t child_max_x[8];  // data[2].zw
    uint8_t child_min_y[8];  // data[3].xy
    uint8_t child_max_y[8];  // data[3].zw
    uint8_t child_min_z[8];  // data[4].xy
    uint8_t child_max_z[8];  // data[4].zw
};
static_assert(sizeof(CwbvhNode) == 80, "CWBVH node must be 80 bytes");

// Device triangle record: 48 B = 3 x float4 so one triangle is three
// global_load_dwordx4.  e1 = v0 - v1, e2 = v2 - v0 (the sign convention used
// inside intersect_ray_tri, query.hlsl:91-93).  The w lanes carry ng = cross(e1, e2)
// (query.hlsl:93), which does not depend on the ray: computed once at upload with the
// kernel's own operation order (separate multiplies and subtract, no contraction), so
// the value is bit-identical to recomputing it per test as the HLSL does.
struct TriDev {
    float v0[3];
    float ngx;
    float e1[3];
    float ngy;
    float e2[3];
    float ngz;
};
static_assert(sizeof(TriDev) == 48, "device triangle must be 48 bytes");

struct Aabb {
    float mn[3];
    float mx[3];
};

} // namespace trx
