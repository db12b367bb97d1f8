// kernels.hip — gfx950 (MI355X, wave64) CWBVH closest-hit kernels.
//
// One persistent wave = 64 ray slots.  Waves pull 64-item chunks (an 8x8 pixel
// tile, or 64 explicit rays) from one work queue per XCD, heaviest tiles first
// (order filed by the first frame of a view from each tile's measured work), traverse with a
// per-lane stack striped through LDS (entry i of lane l at [i*64 + l]:
// conflict-free ds_write_b64/ds_read_b64) that spills to HBM past kLdsStack
// entries, and run the triangle tests of a node step cooperatively: the wave's
// (ray, triangle) pairs are spread over all 64 lanes and committed by their
// owners in order.  DESIGN.md section 4 has the measurements behind each choice.
//
// What each function restates (paths relative to the tray_racing checkout):
//   node_intersect   src/rt_gpu/rt_gpu_software_query.hlsl:213-303
//   tri test         src/rt_gpu/rt_gpu_software_query.hlsl:89-129
//   traversal        src/rt_gpu/rt_gpu_software_query.hlsl:328-438,
//                    src/rt_gpu/rt_gpu_software_query_tlas.hlsl:333-500
//   primary ray      src/rt_gpu/rt_gpu_software.hlsl:69-80, src/rt_cpu/rt_cpu.rs:38-55
//   AO ray           src/rt_gpu/rt_gpu_software.hlsl:105-121, src/rt_cpu/rt_cpu.rs:61-76,
//                    src/rt_gpu/sampling.hlsl:5-51
//
// Arithmetic contract (DESIGN.md "Numerics"): binary32, no contraction
// (-ffp-contract=off; the only fused op is the explicit fmaf of
// TRX_SEM_NODE_FMA), IEEE divide and sqrt, dot = (ax*bx + ay*by) + az*bz.
#include "kernels.h"

#ifndef TRX_NODE_UNROLL
#define TRX_NODE_UNROLL 2
#endif

#pragma clang fp contract(off)

// Lanes-per-ray steps of the thin walk: 1 = eight lanes from 8 rays on (shipped), 2 = also four from 16, 3 = also two from
// 32.  Measured (profiles/r04_thin_waves.log): 8 / 16 / 32 rays = hairball-class AO pass 0.917 / 0.909 / 0.945 ms against
// 1.044 without, bistro-class 0.984 / 0.999 / 1.035 against 1.008 - the earlier steps cost what they gain (a re-pack and a
// drain trip per step, and the waves no longer merge), so the product compiles the last step only.  Measured again over
// fresh processes once the thin walk's requests had become plain loads (profiles/r04_ab_procs_11_thin_levels.log):
// 0.810 / 0.817 / 0.857 ms hairball-class, 0.896 / 0.899 / 0.924 bistro-class - the same conclusion.
#ifndef TRX_THIN_LEVELS
#define TRX_THIN_LEVELS 1
#endif

// TRX_STAMPS (diagnostic builds only): per-wave cycle accounting of the loop's phases.  Every stamp
// drains the memory counters first, so a phase owns the latency of what it issued.
#ifdef TRX_STAMPS
#define TRX_STAMP(acc)                                                  \
    do {                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        acc += now_ - stamp_prev;                                       \
        stamp_prev = now_;                                              \
    } while (0)
#else
#define TRX_STAMP(acc) do { } while (0)
#endif

namespace trx {
namespace {

#define TRX_F32_MAX 3.402823466e+38f
#define TRX_F32_EPSILON 1.1920929e-7f
#define TRX_INVALID 0xFFFFFFFFu

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return (ax * bx + ay * by) + az * bz;
}

struct Ray {
    float ox, oy, oz;
    float dx, dy, dz;    // after the zero-direction fix (query.hlsl:334)
    float ix, iy, iz;    // 1/d, IEEE divide (TRX_SEM_NODE_RCP)
    float tmin;
    uint32_t oct_inv4;
};

// NODE bit0: multiply by precomputed reciprocal, bit1: fused plane evaluation
template <int NODE>
__device__ __forceinline__ float plane(float q, float a, float b) {
    if (NODE & 2) return __builtin_fmaf(q, a, b);
    return q * a + b;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Four registers with no defined contents (a frozen undef: costs no instruction), for values that only the lanes
// which go on to load them will read.
__device__ __forceinline__ float4 unspecified4() {
    f32x4 v;
    v = __builtin_nondeterministic_value(v);
    return make_float4(v.x, v.y, v.z, v.w);
}

// two planes at once: q * a + b on both halves (separate roundings unless NODE & 2)
template <int NODE>
__device__ __forceinline__ f32x2 plane2(f32x2 q, float a, float b) {
    const f32x2 a2 = {a, a}, b2 = {b, b};
    if (NODE & 2) return __builtin_elementwise_fma(q, a2, b2);
    return q * a2 + b2;
}

__device__ __forceinline__ float ubyte(uint32_t x, int j) { return (float)((x >> (8 * j)) & 0xffu); }

template <int NODE>
__device__ __forceinline__ uint32_t node_intersect(const Ray &r, float max_distance, const uint4 n0,
                                                   const uint4 n1, const uint4 n2, const uint4 n3,
                                                   const uint4 n4, const bool pow2) {
    const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
    const uint32_t e_imask = n0.w;
    const float ex = __uint_as_float((e_imask & 0xffu) << 23);
    const float ey = __uint_as_float(((e_imask >> 8) & 0xffu) << 23);
    const float ez = __uint_as_float(((e_imask >> 16) & 0xffu) << 23);
    float ax, ay, az, bx, by, bz;
    if (NODE & 1) {
        ax = ex * r.ix;
        ay = ey * r.iy;
        az = ez * r.iz;
        bx = (px - r.ox) * r.ix;
        by = (py - r.oy) * r.iy;
        bz = (pz - r.oz) * r.iz;
    } else if (pow2) {
        // The shader divides per node (query.hlsl:237-243).  e is a power of two, and a correctly rounded quotient
        // scales exactly with powers of two: RN(2^k / d) = 2^k * RN(1 / d) bit for bit, as long as neither side leaves
        // the normal range - which pow2_exact() below guarantees from the ray (every |d| normal and <= 2^20) and the
        // scene (every exponent byte 0 or >= 21); overflow to infinity happens to both sides at the same threshold.
        // Three of the six IEEE divisions of a node step become multiplications by the ray's 1/d.
        ax = ex * r.ix;
        ay = ey * r.iy;
        az = ez * r.iz;
        bx = (px - r.ox) / r.dx;
        by = (py - r.oy) / r.dy;
        bz = (pz - r.oz) / r.dz;
    } else {
        ax = ex / r.dx;
        ay = ey / r.dy;
        az = ez / r.dz;
        bx = (px - r.ox) / r.dx;
        by = (py - r.oy) / r.dy;
        bz = (pz - r.oz) / r.dz;
    }
    const bool nx = r.dx < 0.0f, ny = r.dy < 0.0f, nz = r.dz < 0.0f;
    uint32_t hit_mask = 0;
#pragma unroll TRX_NODE_UNROLL
    for (int i = 0; i < 2; i++) {
        const uint32_t meta4 = i == 0 ? n1.z : n1.w;
        const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
        const uint32_t bit_index4 = (meta4 ^ (r.oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
        const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
        const uint32_t q_lo_x = i == 0 ? n2.x : n2.y, q_hi_x = i == 0 ? n2.z : n2.w;
        const uint32_t q_lo_y = i == 0 ? n3.x : n3.y, q_hi_y = i == 0 ? n3.z : n3.w;
        const uint32_t q_lo_z = i == 0 ? n4.x : n4.y, q_hi_z = i == 0 ? n4.z : n4.w;
        const uint32_t x_min = nx ? q_hi_x : q_lo_x, x_max = nx ? q_lo_x : q_hi_x;
        const uint32_t y_min = ny ? q_hi_y : q_lo_y, y_max = ny ? q_lo_y : q_hi_y;
        const uint32_t z_min = nz ? q_hi_z : q_lo_z, z_max = nz ? q_lo_z : q_hi_z;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // the near and far plane of one axis go through the same multiply and add: as a float2
            // they are one v_pk_mul_f32 + one v_pk_add_f32 (two IEEE operations per lane in the issue
            // slot of one), or one v_pk_fma_f32; the VALU issue rate is what bounds this kernel
            const f32x2 tx = plane2<NODE>(f32x2{ubyte(x_min, j), ubyte(x_max, j)}, ax, bx);
            const f32x2 ty = plane2<NODE>(f32x2{ubyte(y_min, j), ubyte(y_max, j)}, ay, by);
            const f32x2 tz = plane2<NODE>(f32x2{ubyte(z_min, j), ubyte(z_max, j)}, az, bz);
            const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0001f);
            const float tmax = fminf(fminf(fminf(tx.y, ty.y), tz.y), max_distance);
            if (tmin <= tmax) {
                const uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
                const uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
                hit_mask |= child_bits << bit_index;
            }
        }
    }
    return hit_mask;
}

// The same test over child planes that were converted to floats ONCE for the wave (a wave-uniform node step: every
// lane visits the same node, so the 48 byte-to-float conversions and the 12 near / far selects of the per-lane test are
// the same work 64 times over).  Two tables in LDS, [3 axes][8 children][2]: dec_pos holds {min, max} of a child's
// axis, dec_neg {max, min}; a lane picks the table of an axis by ADDRESS (the sign of its direction: the near plane is
// the max plane when the direction is negative) and reads two children at a time - every lane of a sign class reads
// the same 16 bytes, which LDS broadcasts - so that each {near, far} pair arrives in the two consecutive registers the
// packed arithmetic of plane2 wants.  (Until round 4 the table was plane-major and the pairs were put together with
// two register moves each: 37 moves in a 178-instruction test.)  Same values, same operations, same mask.
template <int NODE>
__device__ __forceinline__ uint32_t node_intersect_dec(const Ray &r, float max_distance, const uint4 n0, const uint4 n1,
                                                       const float *dec_pos, const float *dec_neg, const bool pow2) {
    const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
    const uint32_t e_imask = n0.w;
    const float ex = __uint_as_float((e_imask & 0xffu) << 23);
    const float ey = __uint_as_float(((e_imask >> 8) & 0xffu) << 23);
    const float ez = __uint_as_float(((e_imask >> 16) & 0xffu) << 23);
    float ax, ay, az, bx, by, bz;
    if (NODE & 1) {
        ax = ex * r.ix;
        ay = ey * r.iy;
        az = ez * r.iz;
        bx = (px - r.ox) * r.ix;
        by = (py - r.oy) * r.iy;
        bz = (pz - r.oz) * r.iz;
    } else if (pow2) {
        // The shader divides per node (query.hlsl:237-243).  e is a power of two, and a correctly rounded quotient
        // scales exactly with powers of two: RN(2^k / d) = 2^k * RN(1 / d) bit for bit, as long as neither side leaves
        // the normal range - which pow2_exact() below guarantees from the ray (every |d| normal and <= 2^20) and the
        // scene (every exponent byte 0 or >= 21); overflow to infinity happens to both sides at the same threshold.
        // Three of the six IEEE divisions of a node step become multiplications by the ray's 1/d.
        ax = ex * r.ix;
        ay = ey * r.iy;
        az = ez * r.iz;
        bx = (px - r.ox) / r.dx;
        by = (py - r.oy) / r.dy;
        bz = (pz - r.oz) / r.dz;
    } else {
        ax = ex / r.dx;
        ay = ey / r.dy;
        az = ez / r.dz;
        bx = (px - r.ox) / r.dx;
        by = (py - r.oy) / r.dy;
        bz = (pz - r.oz) / r.dz;
    }
    const float4 *const qx = reinterpret_cast<const float4 *>(r.dx < 0.0f ? dec_neg : dec_pos);
    const float4 *const qy = reinterpret_cast<const float4 *>((r.dy < 0.0f ? dec_neg : dec_pos) + 16);
    const float4 *const qz = reinterpret_cast<const float4 *>((r.dz < 0.0f ? dec_neg : dec_pos) + 32);
    uint32_t hit_mask = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t meta4 = i == 0 ? n1.z : n1.w;
        const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
        const uint32_t bit_index4 = (meta4 ^ (r.oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
        const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
        for (int h = 0; h < 2; h++) { // children 4 i + 2 h, 4 i + 2 h + 1
            const float4 x2 = qx[2 * i + h], y2 = qy[2 * i + h], z2 = qz[2 * i + h]; // {near, far, near, far}
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int j = 2 * h + k;
                const f32x2 tx = plane2<NODE>(k == 0 ? f32x2{x2.x, x2.y} : f32x2{x2.z, x2.w}, ax, bx);
                const f32x2 ty = plane2<NODE>(k == 0 ? f32x2{y2.x, y2.y} : f32x2{y2.z, y2.w}, ay, by);
                const f32x2 tz = plane2<NODE>(k == 0 ? f32x2{z2.x, z2.y} : f32x2{z2.z, z2.w}, az, bz);
                const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0001f);
                const float tmax = fminf(fminf(fminf(tx.y, ty.y), tz.y), max_distance);
                if (tmin <= tmax) {
                    const uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
                    const uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
                    hit_mask |= child_bits << bit_index;
                }
            }
        }
    }
    return hit_mask;
}

// C consecutive CHILDREN of a node (thin waves: L = 8 / C lanes share a ray).  The same IEEE operations on the same
// operands as the per-child body of node_intersect: this lane's contribution to the hit mask, child_bits << bit_index
// for every child of its share whose box the ray enters.  q = the six plane words of the share, C bytes each, in the
// order {x near, x far, y near, y far, z near, z far} (near = the max plane where the direction is negative); meta = its
// C child_meta bytes.
template <int NODE, int C>
__device__ __forceinline__ uint32_t node_children_intersect(const Ray &r, float max_distance, const uint4 n0, uint32_t meta,
                                                            const uint32_t q[6], const bool pow2) {
    const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
    const uint32_t e_imask = n0.w;
    const float ex = __uint_as_float((e_imask & 0xffu) << 23);
    const float ey = __uint_as_float(((e_imask >> 8) & 0xffu) << 23);
    const float ez = __uint_as_float(((e_imask >> 16) & 0xffu) << 23);
    float ax, ay, az, bx, by, bz;
    if (NODE & 1) {
        ax = ex * r.ix;
        ay = ey * r.iy;
        az = ez * r.iz;
        bx = (px - r.ox) * r.ix;
        by = (py - r.oy) * r.iy;
        bz = (pz - r.oz) * r.iz;
    } else if (pow2) { // (see node_intersect)
        ax = ex * r.ix;
        ay = ey * r.iy;
        az = ez * r.iz;
        bx = (px - r.ox) / r.dx;
        by = (py - r.oy) / r.dy;
        bz = (pz - r.oz) / r.dz;
    } else {
        ax = ex / r.dx;
        ay = ey / r.dy;
        az = ez / r.dz;
        bx = (px - r.ox) / r.dx;
        by = (py - r.oy) / r.dy;
        bz = (pz - r.oz) / r.dz;
    }
    const uint32_t is_inner4 = (meta & (meta << 1)) & 0x10101010u;
    const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
    const uint32_t bit_index4 = (meta ^ (r.oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
    const uint32_t child_bits4 = (meta >> 5) & 0x07070707u;
    uint32_t hit_mask = 0u;
#pragma unroll
    for (int j = 0; j < C; j++) {
        const f32x2 tx = plane2<NODE>(f32x2{ubyte(q[0], j), ubyte(q[1], j)}, ax, bx);
        const f32x2 ty = plane2<NODE>(f32x2{ubyte(q[2], j), ubyte(q[3], j)}, ay, by);
        const f32x2 tz = plane2<NODE>(f32x2{ubyte(q[4], j), ubyte(q[5], j)}, az, bz);
        const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0001f);
        const float tmax = fminf(fminf(fminf(tx.y, ty.y), tz.y), max_distance);
        if (tmin <= tmax) hit_mask |= ((child_bits4 >> (8 * j)) & 0xffu) << ((bit_index4 >> (8 * j)) & 0xffu);
    }
    return hit_mask;
}

// C bytes (1, 2 or 4) of a node at a byte address aligned to C: one load instruction.
template <int C>
__device__ __forceinline__ uint32_t load_bytes(const uint8_t *p) {
    if (C == 1) return *p;
    if (C == 2) return *reinterpret_cast<const uint16_t *>(p);
    return *reinterpret_cast<const uint32_t *>(p);
}

// Groups of L = 2, 4 or 8 lanes (thin waves): the value of a group's first lane in all of them, and the OR of all of
// them in the first - both on the DPP network (VALU only).  Every lane of the wave must be active.
template <int L>
__device__ __forceinline__ uint32_t group_first(uint32_t v) {
    if (L == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xa0, 0xf, 0xf, false); // quad_perm [0,0,2,2]
    const uint32_t q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, false);  // quad_perm [0,0,0,0]
    if (L == 4) return q;
    const uint32_t h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)q, 0x114, 0xf, 0xf, false); // row_shr:4
    return (__lane_id() & 4u) ? h : q;
}
template <int L>
__device__ __forceinline__ uint32_t group_or_to_first(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); // row_shl:1 (0 past the row's end)
    if (L >= 4) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true); // row_shl:2
    if (L >= 8) v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true); // row_shl:4
    return v; // (complete in the first lane of every group; the others hold partial ORs)
}

// TriDev = {v0, e1 = v0 - v1, e2 = v2 - v0} as three float4; ng = cross(e1, e2) (query.hlsl:93) rides in the
// w lanes, evaluated at upload exactly as written there.
template <bool EARLY = false>
__device__ __forceinline__ bool intersect_tri(const Ray &r, const float4 a, const float4 b, const float4 c4,
                                              float &t, bool tie_first) {
    const float e1x = b.x, e1y = b.y, e1z = b.z;
    const float e2x = c4.x, e2y = c4.y, e2z = c4.z;
    const float ngx = a.w, ngy = b.w, ngz = c4.w;
    const float cx = a.x - r.ox, cy = a.y - r.oy, cz = a.z - r.oz;
    const float rx = r.dy * cz - r.dz * cy;
    const float ry = r.dz * cx - r.dx * cz;
    const float rz = r.dx * cy - r.dy * cx;
    const float det = dot3(ngx, ngy, ngz, r.dx, r.dy, r.dz);
    const float un = dot3(rx, ry, rz, e2x, e2y, e2z), vn = dot3(rx, ry, rz, e1x, e1y, e1z);
    if (EARLY) {
        // u = un * (1 / det) and v = vn * (1 / det) carry the sign bits sign(un) ^ sign(det), sign(vn) ^ sign(det)
        // whatever the magnitudes (a product's sign is the XOR of its operands', 1 / x keeps the sign of x, also
        // for zeros and infinities); where a NaN appears the test below rejects anyway.  A set sign bit rejects the
        // triangle (query.hlsl:111-116), so when NO lane of the wave can still accept, the IEEE divide, w, tt and the
        // range test are skipped for the whole wave.  Lanes that go on compute exactly what they always did.
        const uint32_t neg = ((__float_as_uint(un) ^ __float_as_uint(det)) | (__float_as_uint(vn) ^ __float_as_uint(det))) & 0x80000000u;
        if (__ballot(neg == 0u) == 0ull) return false;
    }
    const float inv_det = 1.0f / det;
    const float u = un * inv_det;
    const float v = vn * inv_det;
    const float w = 1.0f - u - v;
    const uint32_t hit = __float_as_uint(u) | __float_as_uint(v) | __float_as_uint(w);
    if (inv_det != 0.0f && (hit & 0x80000000u) == 0) {
        const float tt = dot3(ngx, ngy, ngz, cx, cy, cz) * inv_det;
        const bool closer = tie_first ? (tt < t) : (tt <= t);
        if (tt >= r.tmin && closer) {
            t = tt;
            return true;
        }
    }
    return false;
}


__device__ __forceinline__ void finish_ray_dir(Ray &r, float dx, float dy, float dz) {
    r.dx = dx == 0.0f ? TRX_F32_EPSILON : dx;
    r.dy = dy == 0.0f ? TRX_F32_EPSILON : dy;
    r.dz = dz == 0.0f ? TRX_F32_EPSILON : dz;
    r.ix = 1.0f / r.dx;
    r.iy = 1.0f / r.dy;
    r.iz = 1.0f / r.dz;
    r.oct_inv4 = (r.dx < 0.0f ? 0u : 0x04040404u) | (r.dy < 0.0f ? 0u : 0x02020202u) |
                 (r.dz < 0.0f ? 0u : 0x01010101u);
    // bit 31 (masked out wherever oct_inv4 is used): this ray's direction does NOT allow e / d = e * (1/d) exactly
    const float lo = 1.17549435e-38f, hi = 1048576.0f; // 2^-126 (smallest normal), 2^20; a NaN fails both
    const bool exact = fabsf(r.dx) >= lo && fabsf(r.dx) <= hi && fabsf(r.dy) >= lo && fabsf(r.dy) <= hi &&
                       fabsf(r.dz) >= lo && fabsf(r.dz) <= hi;
    if (!exact) r.oct_inv4 |= 0x80000000u;
}

// Wave-uniform: may this node step of the literal-division variants multiply by 1/d (node_intersect, pow2)?
__device__ __forceinline__ bool pow2_exact(const TraceParams &P, const Ray &r, bool act) {
    return P.exp_exact != 0u && __ballot(act && (r.oct_inv4 >> 31) != 0u) == 0ull;
}

__device__ __forceinline__ void mat4_mul(const float *m, float v0, float v1, float v2, float v3, float &r0,
                                         float &r1, float &r2, float &r3) {
    r0 = ((m[0] * v0 + m[4] * v1) + m[8] * v2) + m[12] * v3;
    r1 = ((m[1] * v0 + m[5] * v1) + m[9] * v2) + m[13] * v3;
    r2 = ((m[2] * v0 + m[6] * v1) + m[10] * v2) + m[14] * v3;
    r3 = ((m[3] * v0 + m[7] * v1) + m[11] * v2) + m[15] * v3;
}

__device__ __forceinline__ void primary_dir(const ViewDev &view, uint32_t w, uint32_t h, uint32_t px,
                                            uint32_t py, float &dx, float &dy, float &dz) {
    const float u = (float)px / (float)w;
    float v = (float)py / (float)h;
    v = 1.0f - v;
    const float cx = u * 2.0f - 1.0f, cy = v * 2.0f - 1.0f;
    float vx, vy, vz, vw;
    mat4_mul(view.proj_inv, cx, cy, 1.0f, 1.0f, vx, vy, vz, vw);
    const float s = vw;
    vx = vx / s;
    vy = vy / s;
    vz = vz / s;
    vw = vw / s;
    float wx, wy, wz, ww;
    mat4_mul(view.view_inv, vx, vy, vz, vw, wx, wy, wz, ww);
    dx = wx - view.eye[0];
    dy = wy - view.eye[1];
    dz = wz - view.eye[2];
    const float inv = 1.0f / sqrtf(dot3(dx, dy, dz, dx, dy, dz));
    dx *= inv;
    dy *= inv;
    dz *= inv;
}

__device__ __forceinline__ uint32_t uhash(uint32_t a, uint32_t b) {
    uint32_t x = (a * 1597334673u) ^ (b * 3812015801u);
    x = x ^ (x >> 16);
    x *= 0x7feb352du;
    x = x ^ (x >> 15);
    x *= 0x846ca68bu;
    x = x ^ (x >> 16);
    return x;
}
__device__ __forceinline__ float hash_noise(uint32_t x, uint32_t y, uint32_t frame) {
    return (float)uhash(x, (y << 11) + frame) * (1.0f / 4294967296.0f);
}

// Explicit sin / cos of theta in [0, 2 pi], operation for operation what the test suite's CPU restatement evaluates:
// the binary64 algorithm the GNU C library publishes for sinf / cosf - quadrant through a scaled integer
// conversion, r = x - n * pi/2, a degree-7 odd and a degree-8 even polynomial - rounded once to binary32.  On a Linux
// host that IS the reference CPU path's sin / cos (Rust's f32::sin / cos are libm's; checked bit for bit against glibc
// for every binary32 in [0, 2 pi]).  Once per AO ray; binary64 vector arithmetic runs at full rate on this part.
__device__ __forceinline__ void sincos_det(float theta, float &s, float &c) {
    const uint32_t top = (__float_as_uint(theta) >> 20) & 0x7ffu;
    double x = (double)theta;
    int n = 0;
    if (top >= 0x3f4u) { // |theta| >= 0.75
        const double q = x * 0x1.45F306DC9C883p+23; // 2/pi * 2^24
        n = ((int)q + 0x800000) >> 24;
        x = x - (double)n * 0x1.921FB54442D18p0;
    }
    const double x2 = x * x;
    const double x3 = x * x2;
    const double s1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
    const double x7 = x3 * x2;
    const double sa = x + x3 * -0x1.555545995a603p-3;
    float sp = (float)(sa + x7 * s1);
    const double x4 = x2 * x2;
    const double c2 = -0x1.6c087e89a359dp-10 + x2 * 0x1.99343027bf8c3p-16;
    const double c1 = 0x1p0 + x2 * -0x1.ffffffd0c621cp-2;
    const double x6 = x4 * x2;
    const double ca = c1 + x4 * 0x1.55553e1068f19p-5;
    float cp = (float)(ca + x6 * c2);
    if (top < 0x398u) { // |theta| < 2^-12
        sp = theta;
        cp = 1.0f;
    }
    switch (n & 3) {
    case 0: s = sp; c = cp; break;
    case 1: s = cp; c = -sp; break;
    case 2: s = -sp; c = -cp; break;
    default: s = -cp; c = sp; break;
    }
}

__device__ __forceinline__ uint32_t read_xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x;
}

// Wave64 inclusive scans on the DPP network (row shifts, then row broadcasts): VALU only, no LDS.
// Every lane of the wave must be active.
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}

// Order-preserving image of a float's bits (a < b as floats <=> image(a) < image(b) as unsigned, for non-NaN values
// with -0.0 below +0.0: callers canonicalise zeros first) and its inverse.
__device__ __forceinline__ uint32_t ordered_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ uint32_t unordered_bits(uint32_t k) {
    return k ^ ((k & 0x80000000u) ? 0x80000000u : 0xffffffffu);
}

// Position of the (j+1)-th highest set bit of m (bits 0..23; m has more than j bits set): rank from the bottom, then
// a branch-free descent over 12 / 6 / 3 / 1 / 1 bits.
__device__ __forceinline__ uint32_t select_from_top(uint32_t m, uint32_t j) {
    uint32_t r = (uint32_t)__popc(m) - j, pos = 0u, c;
    c = (uint32_t)__popc(m & 0xfffu);
    if (r > c) { r -= c; pos = 12u; }
    c = (uint32_t)__popc((m >> pos) & 0x3fu);
    if (r > c) { r -= c; pos += 6u; }
    c = (uint32_t)__popc((m >> pos) & 0x7u);
    if (r > c) { r -= c; pos += 3u; }
    c = (m >> pos) & 1u;
    if (r > c) { r -= c; pos += 1u; }
    c = (m >> pos) & 1u;
    if (r > c) pos += 1u;
    return pos;
}

__device__ __forceinline__ uint32_t lane_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// x / d for a launch-uniform d whose m = floor(2^32 / d) comes with the launch parameters (m = 2^32 - 1 for d <= 1): the
// estimate mulhi(x, m) is x / d or one less for every 32-bit x, one correction step makes it exact.  (A division left to
// the compiler computes the same kind of reciprocal on the vector unit and keeps it in a vector register across the
// whole walk - four of them in the refill of kernels that sit at the register budget.)
__device__ __forceinline__ uint32_t div_uniform(uint32_t x, uint32_t d, uint32_t m) {
    const uint32_t q = __umulhi(x, m);
    return x - q * d >= d ? q + 1u : q;
}

// The kernel arguments, through a pointer whose origin the compiler cannot see (k_trace's refill): loads through it
// are scalar loads from the kernel-argument segment that stay where they are written.
__device__ __forceinline__ const TraceParams *refill_params() {
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    zero = __builtin_amdgcn_readfirstlane(zero); // (tells the compiler what the constraint cannot: wave-uniform)
    return (const TraceParams *)((const char *)__builtin_amdgcn_kernarg_segment_ptr() + zero);
}

// The LDS part of the per-lane stacks is addressed through an explicit LDS (address space 3) pointer.  A pop reads either
// LDS or, past kLdsStack entries, HBM: as two loads through generic pointers of the same shape the optimizer merges them
// into ONE load of a selected address - a flat load on the walk's hot path, which also waits for every node record
// requested ahead (vmcnt).  Round 4 shipped that for a while: AO passes +8-10 %, profiles/r04_flat_stack_pop.log.
// tests/test_kernel_resources.py holds the line (no flat_load_dwordx2 in any trace kernel).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
__device__ __forceinline__ uint2 lds_ld(const lds_u32x2 *p) {
    const u32x2 v = *p;
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void lds_st(lds_u32x2 *p, uint2 e) { *p = u32x2{e.x, e.y}; }

// Appends the parked {tile, list} entries to the next frame's tile lists.  A wave files all its tiles of a class under one
// shard, so its entries fall into a few groups (one per class it met); the first lane of a group claims the group's
// slots with ONE returning atomic, the groups' atomics go out together, and every lane stores its tile at its rank
// within the group.  (One atomic per ENTRY - round 1 and 2 - was 32 400 a frame on 128 counters, most of them as the
// waves leave; in a natural-order frame neighbouring tiles share a class and a few counters took them all: 0.08 ms of a
// bistro-class first frame, 0.11 ms of a kitchen-class one, profiles/r03_filing_cost.log.)
__device__ __forceinline__ void flush_pending(const TraceParams &P, uint32_t *wr_set, const uint2 *lds_pend, uint32_t n, uint32_t lane) {
    __builtin_amdgcn_wave_barrier();
    uint2 e = make_uint2(0u, 0xffffffffu);
    // (the address is formed HERE: left alone the compiler forms it at kernel start and carries - or spills - it to the epilogue)
    asm volatile("" : "+v"(lane));
    if (lane < n) e = lds_pend[lane];
    const bool valid = e.y < 16u * kLptShards; // (a list index can only be out of range if LDS was corrupted: never turn that into a stray global atomic)
    // the lanes that share this lane's list: one ballot per class (the shard is the wave's, so the class names the list)
    unsigned long long mine = 0ull;
    const uint32_t cls = valid ? e.y / kLptShards : 0xffffffffu;
#pragma nounroll
    for (uint32_t c = 0; c < 16u; c++) { // (a loop, not sixteen copies: this runs once or twice per wave and must not cost the walk a register)
        const unsigned long long m = __ballot(cls == c);
        if (cls == c) mine = m;
    }
    const uint32_t leader = (uint32_t)__ffsll((long long)mine) - 1u; // (mine != 0 for a valid lane: it contains the lane itself)
    const uint32_t rank = (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
    uint32_t base = 0u;
    if (valid && lane == leader) base = atomicAdd(&wr_set[e.y], (uint32_t)__popcll(mine));
    base = (uint32_t)__shfl((int)base, (int)(valid ? leader : lane));
    if (valid) {
        const uint32_t pos = base + rank;
        if (pos < P.lpt_cap) wr_set[16u * kLptShards + (size_t)e.y * P.lpt_cap + pos] = e.x;
    }
    __builtin_amdgcn_wave_barrier();
}

// The same without the grouping, one atomic per entry: for the flush in the middle of a frame (a wave that has parked
// kLptPend tiles - 4K frames), where the walk's registers are live and the grouped version would cost it a spill.
__device__ __forceinline__ void flush_pending_plain(const TraceParams &P, uint32_t *wr_set, const uint2 *lds_pend, uint32_t n, uint32_t lane) {
    __builtin_amdgcn_wave_barrier();
    if (lane < n) {
        const uint2 e = lds_pend[lane];
        if (e.y < 16u * kLptShards) {
            const uint32_t pos = atomicAdd(&wr_set[e.y], 1u);
            if (pos < P.lpt_cap) wr_set[16u * kLptShards + (size_t)e.y * P.lpt_cap + pos] = e.x;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// PIPE (BLAS-only walks): the fetch of a ray's NEXT node is issued right after the node test that names it, before
// the triangle phase of the node just tested, so the two memory round trips of a step overlap (the next node does
// not depend on the triangles' results, only its test does: it reads the shrunken t).  A ray then occupies its lane
// for one trip more (its last trip has triangle work only), which is why coherent, issue-bound frames do not want it.
template <int MODE, bool TLAS, int NODE, bool PIPE, bool COUNT>
__global__ void __launch_bounds__(kMaxBlock, TRX_MIN_WAVES) k_trace(const TraceParams P) {
    // (P is the kernel's ONLY parameter: refill_params() reads it back from offset 0 of the kernel-argument segment)
    static_assert(!(PIPE && TLAS), "the pipelined walk is BLAS-only");
    // triangles of a lane requested together in a per-lane triangle round (the two-level walk has fewer registers to spare)
    constexpr int kBatch = TLAS ? kTriBatchTlas : (PIPE ? kTriBatchPipe : kTriBatch);
    // one LDS region per wave of the workgroup; waves do not synchronise with each other (incoherent passes:
    // one barrier at the very start and a wait-free hand-over of rays between the two waves of a workgroup, see "The drain")
    extern __shared__ float4 lds_dyn[];
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t wave_in_block = threadIdx.x / kWave;
    // (a macro-like lambda, recomputed at each of its few uses - start-up, a tile filed, the epilogue - rather than a value
    // carried through the walk: the two-level AO kernel sits at the register budget)
    auto wave_id = [&]() -> uint32_t {
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave));
    };
#define wave_global (wave_id())
    char *const lds_wave = reinterpret_cast<char *>(lds_dyn) + wave_in_block * kLdsBytesPerWave;
    lds_u32x2 *const lds_stack =                                                           // [kLdsStack][64], see lds_u32x2
        (lds_u32x2 *)((__attribute__((address_space(3))) char *)lds_dyn + wave_in_block * kLdsBytesPerWave);
    float4 *const lds_ray = reinterpret_cast<float4 *>(lds_wave + kLdsStack * kWave * 8);  // [64][2]: o,tmin | d
    uint2 *const lds_grp = reinterpret_cast<uint2 *>(lds_ray + 2 * kWave);                 // [64] triangle group of the lane
    uint2 *const lds_res = lds_grp + kWave;                                                // [64] {tt bits, triangle}
    uint32_t *const lds_pref = reinterpret_cast<uint32_t *>(lds_res + kWave);              // [64] first pair of the lane
    uint32_t *const lds_head = lds_pref + kWave;                                           // [64] run starts of a window
    uint2 *const lds_pend = reinterpret_cast<uint2 *>(lds_head + kWave);                   // [kLptPend] {tile, list} to append
    float *const lds_dec = reinterpret_cast<float *>(lds_pend + kLptPend);                 // [3][8][2] decoded child planes {min, max} of a wave-uniform node step
    float *const lds_dec_neg = reinterpret_cast<float *>(lds_head);                        // [3][8][2] the same as {max, min} (node_intersect_dec); shares lds_head
    // (wave-uniform base, so that it lives in scalar registers: the HBM part of a stack is touched on rare paths only)
    uint2 *const spill = P.spill + (size_t)wave_global * kWaveScratch;
    float *const wray = reinterpret_cast<float *>(spill + kSpillStack * kWave); // [6][64]: the lanes' world-space rays (kernels.h)
    const bool tie_first = P.tie_first != 0;
    if (P.wave_times && lane == 0) P.wave_times[kWaveTimeStride * wave_global] = wall_clock64();
    // (counting kernels, never timed: what refill_params() reads back IS this launch's parameter block - a second kernel
    // parameter or kernel-argument preloading would move it; a mismatch is reported as a failed launch)
    if (COUNT && lane == 0 && refill_params()->n_items != P.n_items) atomicAdd(&P.ctr->overflow, 1u);
#ifdef TRX_TAIL_DIAG
    // (diagnostic builds only, tools/gpu_tail.py: what was every wave's LAST tile, and when did it start?)
    unsigned long long diag_t0 = 0ull, diag_chunk = 0ull, diag_tiles = 0ull;
    uint32_t diag_pl = 0u, diag_cw = 0u, diag_dry_alive = 0u, diag_dry_age = 0u;
    unsigned long long diag_dry_t = 0ull;
#endif
#ifdef TRX_STAMPS
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    unsigned long long k_refill = 0, k_fetch = 0, k_test = 0, k_tri = 0, k_pop = 0, k_iters = 0;
#endif

    // per-lane ray slot
    bool has_ray = false;
    Ray r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = r.tmin = 0.0f;
    r.oct_inv4 = 0;
    float t = 0.0f;
    uint32_t prim = TRX_INVALID, out_index = 0, sp = 0, steps = 0;
    uint32_t trip = 0; // traversal-loop trips of this wave (uniform)
    uint32_t uni_cool = 0; // (uniform) trips for which the decode-once look is skipped
    uint32_t tlas_sp = TRX_INVALID, bvh_off = 0;
    // instance transforms (TLAS): the instance being walked / the one the current hit was found in, and the
    // world-space ray (origin, direction as given) to come back to when the BLAS is left
    uint32_t cur_inst = TRX_INVALID, hit_inst = TRX_INVALID;
    uint2 cur = make_uint2(0u, 0u);
    // pipelined walk: node in flight / fetched for this lane, and the triangle group its last node test left
    bool fetched = false;
    uint4 pn0 = make_uint4(0u, 0u, 0u, 0u), pn1 = pn0, pn2 = pn0, pn3 = pn0, pn4 = pn0;
    uint2 ptri = make_uint2(0u, 0u);
    uint32_t overflow = 0; // sticky, set on the rare paths only (stack past its HBM part, step cap)
    // kModeFused (the reference's one-dispatch frame, rt_gpu_software.hlsl:47-144): a lane whose primary ray has ended in
    // a hit keeps t / prim / out_index and waits (`pend`) until the wave next refills; it is then turned into that pixel's
    // AO ray in place (`is_ao`), together with every other waiting lane and with the lanes that take new pixels - one
    // pass through the ray set-up code for all of them, not one per finished ray
    constexpr bool kFused = MODE == kModeFused;
    bool pend = false, is_ao = false;
    // The drain (incoherent single-level passes - AO, explicit rays - run two waves to a workgroup).  Every wave of such a pass finds the queues dry
    // holding about 45 rays and then spends a full ray lifetime finishing them at falling occupancy (13 % of a
    // bistro-class AO pass, 64 % of a hairball-class one: profiles/r03_ao_order.log).  Fewer waves draining is the remedy:
    // once the two waves of a workgroup are both dry and their rays fit one wave, the second wave parks its rays' state
    // in its own (now idle) LDS region, offers them with one LDS compare-and-swap and leaves; the first wave picks them
    // up into its idle lanes - stack entries included, they are in the same workgroup's LDS - and goes on.  A ray is
    // the same ray whichever wave steps it, so the hits are those of the unmerged pass.  No wave ever waits for the other:
    // control word 0 goes 0 -> n (rays offered) or 0 -> kMergeClosed (the first wave left first), whichever swap lands.
    // (Single-level walks only.  The two-level kernels were given the hand-over too - ten words more per ray, 32 rays at most: with it on
    // they run 2.5-4 % faster than with it off, but the kernel that contains the code is 3 % slower on the 4K two-level AO pass than the
    // kernel that does not, profiles/r03_drain_merge.log - the code for it stays below, compiled out.)
    // Thin waves (incoherent single-level passes, queues dry, at most P.thin_max rays left): two, four, eight lanes to a ray, see thin_walk.
#ifdef TRX_NO_THIN_CODE // (A/B builds: the kernels without the thin walk's code at all)
    constexpr bool kThin = false;
#else
    constexpr bool kThin = !TLAS && MODE != kModePrimary && !COUNT;
#endif
    bool go_thin = false; // wave-uniform
    // (two-level walks: explicit rays only - measured on the 4K two-level scene, profiles/r05_ab_5_tlas.log: rays -4 %, AO pass +2.5 %)
    constexpr bool kMerge = (!TLAS || MODE == kModeRays) && MODE != kModePrimary && MODE != kModeFused && !COUNT;
    constexpr uint32_t kMergeWords = TLAS ? 31u : 21u, kMergeMax = TLAS ? 32u : 48u, kMergeClosed = 0xffffffffu;
    const bool merging = kMerge && P.merge != 0u && blockDim.x == 2u * kWave;
    bool merge_open = merging; // wave-uniform: this wave has not offered / taken / refused rays yet
    uint32_t need_take = 0u;   // wave-uniform: rays the first wave found offered as it was about to leave
    char *const lds_wave0 = reinterpret_cast<char *>(lds_dyn), *const lds_wave1 = lds_wave0 + kLdsBytesPerWave;
    // control words (in wave 0's decode table, which only coherent primary passes use): [0] offer, [1] free lanes of wave 0
    uint32_t *const merge_ctl = reinterpret_cast<uint32_t *>(lds_wave0 + (reinterpret_cast<char *>(lds_dec) - lds_wave));
    // parking area: wave 1's region from its ray copies up to and including its decode table (4 032 B = 48 x 21 words)
    uint32_t *const merge_box = reinterpret_cast<uint32_t *>(lds_wave1 + kLdsStack * kWave * 8);
    const uint2 *const merge_stack1 = reinterpret_cast<const uint2 *>(lds_wave1);
    if (merging) {
        if (threadIdx.x == 0u) {
            merge_ctl[0] = 0u;
            merge_ctl[1] = 0u;
        }
        __syncthreads(); // (the only barrier of the kernel: both waves are at their very start)
    }
    // COUNT only
    uint32_t c_node = 0, c_tri = 0, c_rays = 0, c_hits = 0, c_maxsp = 0, c_over = 0;
    uint32_t c_wnode = 0, c_wtri = 0; // wave-level executions (leader lane only): SIMD-efficiency denominators
    uint32_t s_wnode = 0, s_wtri = 0; // their values when the current tile started

    // ---- work queues ---------------------------------------------------------------------
    // Work is cut into chunks of 64 items (one 8x8 tile, or 64 explicit rays).  Chunk p belongs
    // to queue p % 8 (ticket k of queue q is chunk 8k + q); a wave pulls from the queue of the XCD
    // it runs on (one atomic head per XCD: a single head saturates near 90 dequeues/us) and
    // steals from the other queues when its own runs dry.  The next ticket is taken when it is needed
    // (heavy tiles, passes without an order) or a tile ahead (light tiles of an ordered frame), see below.
    //
    // Tile order: a wave that finishes a tile of a LEARNING frame (the first frame of a view; every frame of
    // a moving camera) files it in one of 16 cost buckets (sqrt(2)-wide classes of the tile's traversal-loop
    // trips) of the other list set; a frame reads the set on file heaviest bucket first, so chunk p is the p-th
    // heaviest tile (longest-processing-time-first) and every queue starts with heavy tiles - and while the
    // view stays the same it reads that set unchanged and files nothing (`frozen`, below).
    const uint32_t n_chunks = (P.n_items + 63u) >> 6;
    uint32_t my_q = P.single_queue ? 0u : (read_xcc_id() & 7u);
    uint32_t q_probes = 0;
    uint32_t pending = 0; // prefetched ticket of queue my_q (lane 0)
    bool have_pending = true; // uniform: a ticket of queue my_q is in flight / held in `pending`
    if (lane == 0) pending = atomicAdd(&P.ctr->heads[my_q].taken, 1u);
    uint32_t chunk_next = 0, chunk_left = 0; // items of the current chunk not yet handed to a lane
    uint32_t n_pend = 0;                     // tile-list entries parked in lds_pend (uniform)
    uint32_t tile_slot = TRX_INVALID;        // tile being timed (cost feedback)
    unsigned long long tile_t0 = 0;
    uint32_t tile_trip0 = 0;
    uint32_t cur_tile = 0, my_tile = 0;      // tile of the current chunk (uniform) / of this lane's item
    uint32_t cur_vf = 0, my_vf = 0;          // AO batches: frame (noise seed) of the current chunk (uniform) / of this lane's item
    // Lists are kept per (cost bucket, shard): kLptShards appenders per bucket (a wave always appends to its own shard), because one atomic
    // word saturates near 90 appends/us.  Entry e = (15 - bucket) * kLptShards + shard is the e-th
    // list of the heaviest-first concatenation; lane l holds the (exclusive) ends of entries l, l+64.
    uint32_t end_a = 0, end_b = 0;
    bool ordered = false;
    // The feedback machinery tunes itself (exit protocol below): a slot whose frames measured faster WITHOUT it runs without
    // it - natural order, no tile timing, no list appends - until the next re-evaluation.  Every wave reads the same word.
    // (the host files an order for primary and one-seed AO passes only: explicit-ray batches and one-launch frames carry none of this)
    constexpr bool kOrder = MODE != kModeRays;
    const uint32_t fb_mode = (kOrder && P.fb != nullptr && !P.new_view) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)P.fb->mode) : 0u;
    const bool fb_off = fb_mode != 0u;
    // (mode 2: a frame whose tiles are ragged - a few rays of a tile run ten times longer than the rest - keeps its lanes
    // busier by replacing finished rays mid-tile; one frame per launch only, the frame of a batch is taken from whole tiles)
    constexpr uint32_t kFbRefill = 16u;
    const uint32_t refill_idle = (fb_mode == 2u && P.n_frames == 1u) ? kFbRefill : P.refill_idle;
    if (kOrder && P.fb && wave_global == 0u && lane == 0u) P.fb->t0 = wall_clock64(); // (about when the frame's first waves start)
    // The order lives in one of two list sets; the word *lpt_sel (flipped by the exit wave of a frame that filed a new
    // order) says which one is read.  The other set is empty and takes what this frame files - unless the frame finds a
    // COMPLETE order and its views are the previous frame's (same_view): then the order is frozen - replayed as it is, no
    // tile is timed, nothing is filed, the set stays.  Measured (profiles/r04_order_dynamics.log): an order that keeps
    // being rewritten from each frame's completion order wanders - the shards deal a class out to eight lists and
    // concatenate them again, a permutation with short cycles: frame times settle over 10-30 frames or alternate between
    // two values 5 % apart - while the order filed by the very FIRST frame of a view, frozen, runs 1 % faster than the
    // settled one from its second frame on (bistro-class frame 0.420 against 0.425 ms; the filing machinery is off too).
    const bool have_lists = kOrder && P.lpt_sets != nullptr;
    const uint32_t lpt_rd = (have_lists && __builtin_amdgcn_readfirstlane((int)*P.lpt_sel) != 0) ? P.lpt_set_words : 0u;
    uint32_t *const rd_set = P.lpt_sets + lpt_rd, *const wr_set = P.lpt_sets + (P.lpt_set_words - lpt_rd);
    const uint32_t *const rd_lists = rd_set + 16u * kLptShards;
    if (have_lists && !fb_off && !P.no_order) {
        end_a = rd_set[(15u - (lane >> 3)) * kLptShards + (lane & 7u)];
        end_b = rd_set[(15u - ((lane + 64u) >> 3)) * kLptShards + (lane & 7u)];
        // a list that overflowed its capacity dropped entries: fall back to the natural order
        const bool intact = __ballot(end_a > P.lpt_cap || end_b > P.lpt_cap) == 0ull;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t ya = __shfl_up(end_a, off), yb = __shfl_up(end_b, off);
            if ((int)lane >= off) {
                end_a += ya;
                end_b += yb;
            }
        }
        end_b += (uint32_t)__builtin_amdgcn_readlane((int)end_a, 63);
        ordered = intact && (uint32_t)__builtin_amdgcn_readlane((int)end_b, 63) == n_chunks; // complete lists
    }
    bool frozen = ordered && P.same_view != 0u;
#ifdef TRX_DEV_TUNE
    if (P.tune & 0x400u) frozen = false; // (A/B: the order is rewritten by every frame, as until round 3)
#endif
    // (through readfirstlane: a wave-uniform flag that lives to the epilogue belongs in a scalar register)
    const bool lpt_write = __builtin_amdgcn_readfirstlane((have_lists && !fb_off && !frozen) ? 1 : 0) != 0;

    // Chunks below this index of the heaviest-first order take their successor's ticket LATE (when the wave is idle), the
    // rest a tile ahead (hides the atomic's 1-2 us round trip, which only matters next to a tile of a few us).  Round 2
    // cut the order in half by position (never / always / heaviest 3 % / 12 % / 50 %, profiles/r02_late_binding.log);
    // round 3 looked at what the last waves out of a frame had been doing (tools/gpu_tail.py): starting a 20-30 us tile
    // they had reserved a tile earlier, while hundreds of idle waves found the queues dry and left - the position cut
    // sits in the middle of the 20-40 us tiles of the bistro-class frame.  The cut is a class now: tiles of 16 trips or more (class 7 and up, about 35 us mid-frame) bind late (profiles/r03_tile_classes.log:
    // classes 6..9 within 1 % of each other, all ahead of the position cut).
    constexpr uint32_t kLateClass = 7u;
    uint32_t late_entry = (15u - kLateClass) * kLptShards + (kLptShards - 1u); // last list of that class in the concatenation
    uint32_t late_cut = 0u;
#ifdef TRX_DEV_TUNE
    const uint32_t late_sel = (P.tune >> 4) & 7u; // 1 always late, 2 / 3 / 4: heaviest 3 / 12 / 50 % by position, 5 never, 6: class = tune bits 20..23
    if (late_sel == 6u) late_entry = (15u - ((P.tune >> 20) & 15u)) * kLptShards + (kLptShards - 1u);
#endif
    if (ordered)
        late_cut = late_entry < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)end_a, (int)late_entry)
                                    : (uint32_t)__builtin_amdgcn_readlane((int)end_b, (int)(late_entry - 64u));
#ifdef TRX_DEV_TUNE
    if (late_sel >= 1u && late_sel <= 5u)
        late_cut = late_sel == 1u ? 0xffffffffu : late_sel == 5u ? 0u : late_sel == 4u ? (n_chunks >> 1) : P.prio_cut[late_sel - 2u];
#endif
    // ... and so do the last tiles of the order, half as many as there are waves: a ticket reserved a tile ahead there is a
    // tile that an idle wave could have started (profiles/r03_tile_classes.log section 9: 0 / 1024 / 2048 / 4096 / 8192
    // tiles with 4096 waves; kitchen-class -2 %, bistro-class -1 %, the others unchanged up to 2048, slower beyond)
    uint32_t tail_tiles = (gridDim.x * (blockDim.x / kWave)) >> 1;
#ifdef TRX_DEV_TUNE
    if (P.tune >> 28) tail_tiles = ((P.tune >> 28) == 15u ? 0u : (P.tune >> 28) * 1024u);
#endif
    const uint32_t tail_cut = n_chunks > tail_tiles ? n_chunks - tail_tiles : 0u;
    bool exhausted = false; // wave-uniform
    // (defined out here, ahead of the loop, because the thin walk also runs after it: see the end of the loop)
    // Stack push / pop.  Fast path: every lane's top is inside the LDS part, so the write needs no
    // predication at all (an entry above a lane's top is free to clobber) and the whole push is one
    // ds_write_b64 plus a conditional increment; lanes past the LDS part (rare: depth > kLdsStack) take the
    // general path behind a wave-uniform branch.
    auto stack_push = [&](uint2 e, bool cond) {
        if (__builtin_expect(__ballot(sp >= (uint32_t)kLdsStack) != 0ull, 0)) {
            if (cond) {
                if (sp < (uint32_t)kLdsStack) lds_st(&lds_stack[sp * kWave + lane], e);
                else if (sp < (uint32_t)(kLdsStack + kSpillStack)) spill[(sp - kLdsStack) * kWave + lane] = e;
                else overflow = 1u;
            }
        } else {
            lds_st(&lds_stack[sp * kWave + lane], e);
        }
        sp += cond ? 1u : 0u;
        if (COUNT) c_maxsp = max(c_maxsp, sp);
    };
    auto stack_pop = [&]() -> uint2 { // callers guarantee sp != 0
        sp--;
        if (__builtin_expect(__ballot(sp >= (uint32_t)kLdsStack) != 0ull, 0)) {
            if (sp < (uint32_t)kLdsStack) return lds_ld(&lds_stack[sp * kWave + lane]);
            if (sp < (uint32_t)(kLdsStack + kSpillStack)) return spill[(sp - kLdsStack) * kWave + lane];
            return make_uint2(0u, 0u);
        }
        return lds_ld(&lds_stack[sp * kWave + lane]);
    };

    // Finished ray: the hit record (or the any-hit flag) leaves the lane.
    auto finish_lane = [&]() {
        if (MODE == kModeRays && P.any_hit != 0u) {
            reinterpret_cast<uint8_t *>(P.out)[out_index] = prim != TRX_INVALID ? 1u : 0u;
        } else {
            trx_hit h;
            h.t = prim != TRX_INVALID ? t : __builtin_inff();
            h.prim = prim;
            const uint32_t hi = prim != TRX_INVALID ? hit_inst : TRX_INVALID;
            if (kFused && is_ao) {
                P.out_ao[out_index] = h;
                if (TLAS && P.out_ao_inst) P.out_ao_inst[out_index] = hi;
            } else {
                P.out[out_index] = h;
                if (TLAS && P.out_inst) P.out_inst[out_index] = hi;
                if (kFused) {
                    // the reference's pixel program goes on with the AO ray when the primary ray hit
                    // (rt_gpu_software.hlsl:105); a miss ends the pixel: its AO record is a miss as well
                    if (prim != TRX_INVALID) {
                        pend = true;
                    } else {
                        P.out_ao[out_index] = h;
                        if (TLAS && P.out_ao_inst) P.out_ao_inst[out_index] = TRX_INVALID;
                    }
                }
            }
        }
        if (COUNT) {
            c_rays++;
            c_hits += prim != TRX_INVALID;
        }
        c_over += overflow;
        has_ray = false;
    };

    // Thin waves.  An incoherent pass ends when its longest rays do (two thirds of a hairball-class AO pass is its
    // drain), and at the end those rays sit one or two to a wave: the wave then issues alone on its SIMD, one
    // instruction every four to five cycles whatever the instruction is, so a ray's trip costs its INSTRUCTION COUNT -
    // 213 vector instructions for the node test of one lane while 63 lanes idle.  So once a dry wave is down to 8 rays
    // they are moved to lanes 0, 8, 16 ... and every ray gets L = 8 lanes (the code is generic in L = 2, 4, 8; see
    // TRX_THIN_LEVELS for why only 8 ships): lane j of a group tests children j 8/L ... of the node (their plane bytes,
    // loaded by address; the same IEEE operations as the per-lane test: node_children_intersect),
    // the contributions are ORed on the DPP network, and a leaf's triangles are tested L at a time and folded
    // into the ray's 64-bit {t, sequence} key with the same LDS atomic min as the cooperative rounds - node order,
    // triangle order, tie rule and every t are those of the one-lane walk (tested bit for bit).  A trip is about a
    // third of the instructions.  The walk is the plain one, rotated like the pipelined one: triangles of the node
    // tested last, then the next node.
    // (wave-uniform) may this wave go thin now?  Queues dry, a handful of rays, every stack inside its LDS part, no
    // hand-over with the other wave of the workgroup pending, no lane of a fused frame waiting to become an AO ray.
    auto thin_now = [&](uint32_t alive) -> bool {
        // (never more rays than the walks compiled in can hold, whatever the launch parameter says: 8 << (TRX_THIN_LEVELS - 1))
        constexpr uint32_t kHold = kFused ? 8u : 8u << (TRX_THIN_LEVELS - 1);
        return exhausted && alive != 0u && alive <= (P.thin_max < kHold ? P.thin_max : kHold) && !(kMerge && merge_open) &&
               __ballot(has_ray && sp > (uint32_t)kLdsStack) == 0ull && !(kFused && __ballot(pend) != 0ull);
    };
    auto thin_walk = [&](auto lanes_per_ray) {
        constexpr uint32_t L = (uint32_t)decltype(lanes_per_ray)::value; // lanes to a ray: 2, 4 or 8
        constexpr uint32_t C = 8u / L;                                   // children of a node to a lane
        constexpr uint32_t kCap = (uint32_t)kWave / L;                   // rays the wave holds this way
        const uint32_t sub = lane & (L - 1u), first = lane & ~(L - 1u);
        {   // ---- move ray k (in lane order) to lane L k, stack column and all; give its L - 1 helpers the ray
            const unsigned long long act = __ballot(has_ray);
#ifdef TRX_TAIL_DIAG
            // (diagnostic builds, tools/gpu_timeline_ao.py: when the wave first went thin, with how many rays, after how many trips)
            if (diag_t0 == 0ull) {
                diag_t0 = wall_clock64();
                diag_chunk = (unsigned long long)__popcll(act);
                diag_tiles = trip;
            }
#endif
            // the k-th ray's lane, through a table in LDS (the cooperative rounds' run-head table, idle here)
            if (has_ray) lds_head[lane_rank(act)] = lane;
            __builtin_amdgcn_wave_barrier();
            const bool filled = lane / L < (uint32_t)__popcll(act);
            const int src = filled ? (int)lds_head[lane / L] : (int)lane;
            __builtin_amdgcn_wave_barrier();
            const bool owner = filled && sub == 0u;
            r.ox = __shfl(r.ox, src); r.oy = __shfl(r.oy, src); r.oz = __shfl(r.oz, src);
            r.dx = __shfl(r.dx, src); r.dy = __shfl(r.dy, src); r.dz = __shfl(r.dz, src);
            r.ix = __shfl(r.ix, src); r.iy = __shfl(r.iy, src); r.iz = __shfl(r.iz, src);
            r.tmin = __shfl(r.tmin, src);
            r.oct_inv4 = (uint32_t)__shfl((int)r.oct_inv4, src);
            t = __shfl(t, src);
            prim = (uint32_t)__shfl((int)prim, src);
            out_index = (uint32_t)__shfl((int)out_index, src);
            const uint32_t age = (uint32_t)__shfl((int)(trip - steps), src);
            steps = trip - age;
            cur.x = (uint32_t)__shfl((int)cur.x, src); cur.y = (uint32_t)__shfl((int)cur.y, src);
            ptri.x = (uint32_t)__shfl((int)ptri.x, src); ptri.y = (uint32_t)__shfl((int)ptri.y, src);
            overflow = (uint32_t)__shfl((int)overflow, src);
            if (kFused) is_ao = __shfl(is_ao ? 1 : 0, src) != 0;
            const uint32_t sp_src = (uint32_t)__shfl((int)sp, src);
            uint32_t sp_max = owner ? sp_src : 0u;
            for (int off = 32; off > 0; off >>= 1) sp_max = max(sp_max, (uint32_t)__shfl_xor((int)sp_max, off));
            // (the destination address is formed HERE: left alone the compiler forms it at kernel start and the one-launch
            // frame, at the register budget, spills it)
            uint32_t dst = lane;
            asm volatile("" : "+v"(dst));
            for (uint32_t k = 0; k < sp_max; k++) { // (entries beyond a ray's own top are copied too: harmless)
                const uint2 e = lds_ld(&lds_stack[k * kWave + (uint32_t)src]);
                __builtin_amdgcn_wave_barrier();
                if (owner) lds_st(&lds_stack[k * kWave + dst], e);
                __builtin_amdgcn_wave_barrier();
            }
            sp = owner ? sp_src : 0u;
            has_ray = owner;
            if (!owner) {
                ptri = make_uint2(0u, 0u);
                cur = make_uint2(0u, 0u);
                if (!filled) r.oct_inv4 &= 0x7fffffffu; // (a group without a ray must not veto the exact-reciprocal shortcut)
            }
            if (owner) {
                lds_ray[2u * lane] = make_float4(r.ox, r.oy, r.oz, r.tmin);
                lds_ray[2u * lane + 1u] = make_float4(r.dx, r.dy, r.dz, 0.0f);
            }
            fetched = false;
            __builtin_amdgcn_s_setprio(L == 8u ? 3 : 2);
        }
        unsigned long long *const lds_key = reinterpret_cast<unsigned long long *>(lds_res);
        // The trip is rotated like the pipelined walk's, and one step further: (1) the ray's next node is chosen and its
        // bytes are requested, (2) the triangles of the node tested in the PREVIOUS trip are tested - the first L
        // records were requested at the end of that trip - (3) a ray with no node left is finished, (4) the requested node
        // is tested with the t those triangles left, (5) the first L triangle records of the new leaf hits are
        // requested.  A thin wave's time is its rays' dependent round trips to memory.  (The requests are plain loads: an
        // `asm volatile` barrier naming the loaded registers right behind them - there until late in round 4 "to keep
        // the loads early" - makes the compiler wait for them on the spot; without it the AO pass of a hairball-class
        // scene runs 6 % faster, random rays 6-7 %.  Requesting (5) right behind (1) instead, both in flight together
        // and nothing across the back-edge, was built as well and measures the same to slightly slower.)
        uint4 n0 = make_uint4(0u, 0u, 0u, 0u), n1 = n0;
        uint32_t q0 = 0u, q1 = 0u, q2 = 0u, q3 = 0u, q4 = 0u, q5 = 0u;
        float4 ta = unspecified4(), tb = unspecified4(), tc = unspecified4();
        uint32_t gx = 0u, gy = 0u, cnt = 0u, pre_local = 0u;
        // (the one-launch frame comes back from here into its loop, whose registers stay live meanwhile: it does without
        // the early triangle request, twelve registers carried from trip to trip)
        constexpr bool kPre = !kFused;
        auto request_triangles = [&]() { // (5): the group's pending triangle group, and this lane's record of its first L
            cnt = 0u;
            if (__ballot(ptri.y != 0u) == 0ull) return; // (most trips of a handful of rays find no leaf)
            gx = group_first<(int)L>(ptri.x);
            gy = group_first<(int)L>(ptri.y);
            cnt = (uint32_t)__popc(gy);
            if (kPre && sub < cnt) {
                pre_local = select_from_top(gy, sub); // (kept for (2): the search is thirty-five instructions of a lone wave's trip)
                const float4 *tp = P.tris + (size_t)(gx + pre_local) * 3;
                ta = tp[0];
                tb = tp[1];
                tc = tp[2];
            }
        };
        request_triangles(); // (a wave that comes from the pipelined walk brings pending triangle groups)
        // wave-uniform: what is in flight is finished, then out - a fused frame's primary ray has hit, or the rays have
        // become few enough for twice the lanes each
        bool leaving = false;
        for (;;) {
            trip++;
            // ---- (1) the next node of every ray that holds a node group; its bytes are requested
            const bool had_group = has_ray && (cur.y & 0xff000000u) != 0u;
            const bool stepping = had_group && !leaving;
            uint32_t node_index = 0u;
            if (stepping) {
                const uint32_t hits_imask = cur.y;
                const uint32_t child_bit = 31u - (uint32_t)__builtin_clz(hits_imask);
                cur.y &= ~(1u << child_bit);
                const uint32_t slot = (child_bit - 24u) ^ (r.oct_inv4 & 0xffu);
                node_index = cur.x + (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
                stack_push(cur, (cur.y & 0xff000000u) != 0u);
            }
            const bool gstep = ((__ballot(stepping) >> first) & 1ull) != 0ull;
            if (gstep) {
                const uint4 *np = P.nodes + (size_t)group_first<(int)L>(node_index) * 5;
                const uint8_t *nb = reinterpret_cast<const uint8_t *>(np) + sub * C;
                n0 = np[0];
                n1 = np[1];
                // the plane bytes of this lane's C children by address: min planes at +32 / +48 / +64, max planes eight bytes
                // on; near = max where d < 0
                const uint32_t xn = r.dx < 0.0f ? 8u : 0u, yn = r.dy < 0.0f ? 8u : 0u, zn = r.dz < 0.0f ? 8u : 0u;
                q0 = load_bytes<(int)C>(nb + 32u + xn); q1 = load_bytes<(int)C>(nb + 32u + (xn ^ 8u));
                q2 = load_bytes<(int)C>(nb + 48u + yn); q3 = load_bytes<(int)C>(nb + 48u + (yn ^ 8u));
                q4 = load_bytes<(int)C>(nb + 64u + zn); q5 = load_bytes<(int)C>(nb + 64u + (zn ^ 8u));
            }
            // ---- (2) triangles of the node tested in the previous trip, L at a time; the first L are here already
            if (__ballot(cnt != 0u) != 0ull) {
                const uint32_t init_lo = tie_first ? 0u : 0xffu;
                if (sub == 0u) lds_res[lane] = make_uint2(init_lo, ordered_bits(t + 0.0f));
                __builtin_amdgcn_wave_barrier();
                auto test_one = [&](uint32_t local, const float4 &a, const float4 &b, const float4 &c) {
                    float tt = TRX_F32_MAX; // the tie test against the ray's t is the atomic min (see the cooperative rounds)
                    if (intersect_tri(r, a, b, c, tt, false)) {
                        const uint32_t neg_zero = __float_as_uint(tt) == 0x80000000u ? 1u : 0u;
                        const uint32_t lo = ((tie_first ? 31u - local : local) << 1) | neg_zero;
                        atomicMin(&lds_key[first], ((unsigned long long)ordered_bits(tt + 0.0f) << 32) | lo);
                    }
                };
                // (the first L stand apart from the loop: a loop that loads waits at its head for everything in flight -
                // the node bytes requested in (1) included - where these only need their own records)
                uint32_t j = sub;
                if (kPre) {
                    if (j < cnt) test_one(pre_local, ta, tb, tc);
                    j += L;
                }
                for (; __ballot(j < cnt) != 0ull; j += L) {
                    if (j < cnt) {
                        const uint32_t local = select_from_top(gy, j);
                        const float4 *tp = P.tris + (size_t)(gx + local) * 3;
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        test_one(local, a, b, c);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (sub == 0u && cnt != 0u) {
                    const uint2 won = lds_res[lane];
                    if (won.x != init_lo) {
                        const uint32_t local = tie_first ? 31u - (won.x >> 1) : (won.x >> 1);
                        t = (won.x & 1u) ? -0.0f : __uint_as_float(unordered_bits(won.y));
                        prim = ptri.x + local;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            ptri = make_uint2(0u, 0u);
            // ---- (3) the end of a ray: it had no node left to visit and its last triangles are in
            if (has_ray) {
                bool done = !had_group;
                if (MODE == kModeRays && P.any_hit != 0u && prim != TRX_INVALID) done = true; // (a node in flight is simply not looked at)
                if (__builtin_expect((trip & 1023u) == 0u, 0)) { // step cap: every wave reaches an exit
                    if (trip - steps > kMaxSteps) {
                        overflow = 1u;
                        done = true;
                    }
                }
                if (done) finish_lane();
            }
            if (__ballot(has_ray) == 0ull) break;
            // (fused frames: a primary ray has hit and waits to become an AO ray at the refill point.  What is in flight is
            // finished first - this trip's node tests, then, in one more trip that chooses no node, their triangles - so
            // that every ray left holds a node group and nothing pending: the state the other walks expect.)
            if (leaving) break;
            const unsigned long long alive_mask = __ballot(has_ray);
            if ((kFused && __ballot(pend) != 0ull) || (L < 8u && (uint32_t)__popcll(alive_mask) <= kCap / 2u)) leaving = true;
            // ---- (4) node test, C children per lane, for the rays still there
            const bool galive = ((alive_mask >> first) & 1ull) != 0ull;
            const float gt = __uint_as_float(group_first<(int)L>(__float_as_uint(t)));
            const bool pow2 = (NODE & 1) ? false : pow2_exact(P, r, gstep && galive);
            uint32_t contrib = 0u;
            if (gstep && galive) {
                const uint32_t meta = (sub * C < 4u ? n1.z : n1.w) >> (8u * ((sub * C) & 3u));
                const uint32_t q[6] = {q0, q1, q2, q3, q4, q5};
                contrib = node_children_intersect<NODE, (int)C>(r, gt, n0, meta, q, pow2);
            }
            const uint32_t hitmask = group_or_to_first<(int)L>(contrib);
            if (stepping && has_ray) {
                cur.x = n1.x;
                ptri.x = n1.y;
                cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                ptri.y = hitmask & 0x00ffffffu;
                if ((cur.y & 0xff000000u) == 0u && sp != 0u) {
                    cur = stack_pop();
                    if (__builtin_expect(overflow != 0u, 0)) cur = make_uint2(0u, 0u); // past the last entry: finish
                }
            }
            // ---- (5)
            request_triangles();
        }
        __builtin_amdgcn_s_setprio(0);
        if (PIPE) { // (no node was in flight all the while: tell the register allocator so, it cannot see through `fetched`)
            const float4 u0 = unspecified4(), u1 = unspecified4(), u2 = unspecified4(), u3 = unspecified4(), u4 = unspecified4();
            pn0 = make_uint4(__float_as_uint(u0.x), __float_as_uint(u0.y), __float_as_uint(u0.z), __float_as_uint(u0.w));
            pn1 = make_uint4(__float_as_uint(u1.x), __float_as_uint(u1.y), __float_as_uint(u1.z), __float_as_uint(u1.w));
            pn2 = make_uint4(__float_as_uint(u2.x), __float_as_uint(u2.y), __float_as_uint(u2.z), __float_as_uint(u2.w));
            pn3 = make_uint4(__float_as_uint(u3.x), __float_as_uint(u3.y), __float_as_uint(u3.z), __float_as_uint(u3.w));
            pn4 = make_uint4(__float_as_uint(u4.x), __float_as_uint(u4.y), __float_as_uint(u4.z), __float_as_uint(u4.w));
        }
    };
    // The thin walk of a dry wave, with as many lanes to a ray as its rays allow; returns with no ray left or, fused frames,
    // with a lane waiting to become an AO ray (thin_walk<2 or 4> returns once twice the lanes fit).
    auto thin_all = [&]() {
        // (straight-line, not a loop over the three walks: the lanes to a ray only ever double)
        auto rays_left = [&]() -> uint32_t {
            return (kFused && __ballot(pend) != 0ull) ? 0u : (uint32_t)__popcll(__ballot(has_ray));
        };
        // (the one-launch frame, whose loop the walk returns into, has the registers for the last step only)
        if constexpr (!kFused && TRX_THIN_LEVELS >= 3) {
            if (rays_left() > 16u) thin_walk(std::integral_constant<int, 2>{});
        }
        if constexpr (!kFused && TRX_THIN_LEVELS >= 2) {
            if (rays_left() > 8u) thin_walk(std::integral_constant<int, 4>{});
        }
        if (rays_left() != 0u) thin_walk(std::integral_constant<int, 8>{});
    };

    for (;;) {
        TRX_STAMP(k_pop);
        // ---- refill idle lanes from the queues --------------------------------------------
        const unsigned long long idle = __ballot(!has_ray);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (tile_slot != TRX_INVALID && n_idle == (uint32_t)kWave) {
            // the whole tile is done: record what it cost (feeds the next frame's tile order)
            if (lane == 0) {
                const uint32_t c = (uint32_t)(wall_clock64() - tile_t0);
#ifdef TRX_TAIL_DIAG
                if (P.cost && (P.tune & 0x2000000u)) {
                    P.cost[tile_slot] = ((uint32_t)diag_chunk << 16) | min(c, 65535u); // position in the order | cost
                    if (P.tile_iters && !COUNT) P.tile_iters[tile_slot] = (min(trip - tile_trip0, 4095u) << 20) | (min(diag_pl, 1023u) << 10) | min(diag_cw, 1023u);
                } else
#endif
                if (P.cost) P.cost[tile_slot] = c;
                if (lpt_write) {
                    // The tile's class for the next frame's order = its WORK, in half-octaves of traversal-loop trips (1, 2,
                    // 3, 4, 6, 8 ... 256+), not its duration.  How long a tile takes depends on when it ran - at the start
                    // of the frame with issue priority, mid-frame against four waves a SIMD, in the tail against none - so
                    // a frame ordered by last frame's durations reshuffles itself every frame (static camera: 43 % of the
                    // tiles moved by two duration classes or more from one frame to the next, tools/gpu_tail.py); trips
                    // are a property of the tile and the view, the same every frame.  A least-squares fit of mid-frame tile
                    // time on trips, per-lane triangle rounds and cooperative rounds explains no more than trips alone
                    // (residual 28.8 against 29.4 us rms), and classes of weighted work measured slower, so: trips.
                    // profiles/r03_tile_classes.log: hairball-class frame -6 %, dense -1.8 %, bistro-class -1.5 %.
                    uint32_t b;
#ifdef TRX_DEV_TUNE
                    if (P.tune & 0x4000000u) { // round-2 classes: 2*log2(duration), 2.56 us .. 0.49 ms
                        const uint32_t msb = 31u - (uint32_t)__clz((int)(c | 1u));
                        const uint32_t kk = 2u * msb + (msb ? (c >> (msb - 1u)) & 1u : 0u);
                        b = kk < 16u ? 0u : min(kk - 16u, 15u);
                    } else
#endif
                    {
                        const uint32_t wk = max(trip - tile_trip0, 1u);
                        const uint32_t msb = 31u - (uint32_t)__clz((int)wk);
                        const uint32_t kk = 2u * msb + (msb ? (wk >> (msb - 1u)) & 1u : 0u);
                        b = kk == 0u ? 0u : min(kk - 1u, 15u);
                    }
                    uint32_t list = b * kLptShards + (wave_global & (kLptShards - 1u)); // (one shard per wave: see flush_pending)
#ifdef TRX_DEV_TUNE
                    if (P.tune & 0x200u) list = b * kLptShards; // (experiment: one list per class)
#endif
                    // park the entry in LDS: the appends (returning atomics) are issued together,
                    // one lane each, when the buffer fills or the wave exits, off every tile's path
                    lds_pend[n_pend] = make_uint2(tile_slot, list);
                }
            }
            if (lpt_write) {
                n_pend++;
                if (n_pend == (uint32_t)kLptPend) {
                    flush_pending_plain(P, wr_set, lds_pend, n_pend, lane);
                    n_pend = 0;
                }
            }
            if (COUNT && P.tile_iters) {
                // wave-level node steps and triangle rounds of this tile (each is counted by one lane)
                uint32_t dn = c_wnode - s_wnode, dt = c_wtri - s_wtri;
                for (int off = 32; off > 0; off >>= 1) {
                    dn += __shfl_xor(dn, off);
                    dt += __shfl_xor(dt, off);
                }
                if (lane == 0) P.tile_iters[tile_slot] = (min(dn, 65535u) << 16) | min(dt, 65535u);
            }
            tile_slot = TRX_INVALID;
        }
        // (fused frames: lanes waiting to become AO rays are idle but take no new pixel)
        const unsigned long long freem = kFused ? (idle & ~__ballot(pend)) : idle;
        const uint32_t n_free = kFused ? (uint32_t)__popcll(freem) : n_idle;
        const bool take = !exhausted && n_idle >= refill_idle;
        if (take || (kFused && freem != idle)) {
            const uint32_t rank = lane_rank(freem);
            uint32_t given = 0, item = TRX_INVALID;
            while (take && given < n_free) {
                if (chunk_left == 0u) {
                    // take the prefetched ticket; walk to the next queue when this one is dry
                    if (!have_pending && lane == 0) pending = atomicAdd(&P.ctr->heads[my_q].taken, 1u);
                    uint32_t ticket = __builtin_amdgcn_readfirstlane(pending);
                    uint32_t q_count = P.single_queue ? n_chunks : ((n_chunks + 7u - my_q) >> 3);
                    while (ticket >= q_count) {
                        if (++q_probes >= (P.single_queue ? 1u : 8u)) break;
                        my_q = (my_q + 1u) & 7u;
                        if (lane == 0) pending = atomicAdd(&P.ctr->heads[my_q].taken, 1u);
                        ticket = __builtin_amdgcn_readfirstlane(pending);
                        q_count = (n_chunks + 7u - my_q) >> 3;
                    }
                    if (ticket >= q_count) {
                        exhausted = true;
#ifdef TRX_TAIL_DIAG
                        diag_dry_t = wall_clock64();                                 // when the wave found every queue dry ...
                        diag_dry_alive = (uint32_t)__popcll(__ballot(has_ray));      // ... how many rays it still held ...
                        {   // ... and the age of the oldest, in trips
                            uint32_t age = has_ray ? trip - steps : 0u;
                            for (int off = 32; off > 0; off >>= 1) age = max(age, (uint32_t)__shfl_xor((int)age, off));
                            diag_dry_age = age;
                        }
#endif
                        break;
                    }
                    // a ticket taken a tile ahead hides the atomic's round trip (1-2 us) but binds the wave's NEXT tile
                    // while it still works on this one: a wave stuck on a 0.3 ms tile then sits on a second one that an idle
                    // wave could have started.  Late binding wins wherever tiles are long or of unknown cost - the heavier
                    // half of the heaviest-first order, and every pass that has no learnt order (first frame, AO, explicit
                    // rays); only the light half of an ordered frame keeps the prefetch
                    have_pending = ordered && ticket * (P.single_queue ? 1u : 8u) >= late_cut && ticket * (P.single_queue ? 1u : 8u) < tail_cut;
                    if (have_pending && lane == 0) pending = atomicAdd(&P.ctr->heads[my_q].taken, 1u);
                    const uint32_t chunk = P.single_queue ? ticket : ticket * 8u + my_q;
                    chunk_next = chunk << 6;
                    chunk_left = min(64u, P.n_items - chunk_next);
                    cur_tile = chunk;
                    if (MODE == kModeAo && P.n_frames > 1u) {
                        // AO batch (one view, one primary hit buffer, n_frames noise seeds - configs[3]'s "4 spp"): the
                        // seeds of a tile are consecutive tickets of ONE queue, so the same XCD walks the same origins
                        // through the same upper nodes n_frames times over; a tile id past the image is the padding of
                        // the last group of eight tiles (every queue has the same number of tickets)
                        const uint32_t tq = div_uniform(ticket, P.n_frames, P.rcp_n_frames);
                        cur_vf = ticket - tq * P.n_frames;
                        cur_tile = P.single_queue ? tq : tq * 8u + my_q;
                        if (cur_tile >= P.tiles_per_frame) {
                            chunk_left = 0u;
                            continue;
                        }
                    }
                    if (ordered) {
                        // chunk -> bucket (heaviest first) -> tile
                        uint32_t j = (uint32_t)__popcll(__ballot(chunk >= end_a));
                        if (j == 64u) j += (uint32_t)__popcll(__ballot(chunk >= end_b));
                        const uint32_t start = j == 0u ? 0u
                                               : j <= 64u ? (uint32_t)__builtin_amdgcn_readlane((int)end_a, (int)(j - 1u))
                                                          : (uint32_t)__builtin_amdgcn_readlane((int)end_b, (int)(j - 65u));
                        const uint32_t list = (15u - (j >> 3)) * kLptShards + (j & 7u);
                        cur_tile = rd_lists[(size_t)list * P.lpt_cap + (chunk - start)];
                        cur_tile = __builtin_amdgcn_readfirstlane(cur_tile);
                        if (cur_tile >= n_chunks) cur_tile = chunk; // (only a corrupted list can name such a tile: never turn it into an out-of-range pixel)
                        // the tiles that set the frame's critical path get issue priority over the
                        // waves they share a SIMD with
                        if (chunk < P.prio_cut[0]) __builtin_amdgcn_s_setprio(3);
                        else if (chunk < P.prio_cut[1]) __builtin_amdgcn_s_setprio(2);
                        else if (chunk < P.prio_cut[2]) __builtin_amdgcn_s_setprio(1);
                        else __builtin_amdgcn_s_setprio(0);
                    }
                    if ((P.cost || lpt_write) && n_idle == (uint32_t)kWave && given == 0u) {
                        tile_slot = cur_tile; // cost is filed under the tile, not the chunk
                        tile_t0 = wall_clock64();
                        tile_trip0 = trip;
#ifdef TRX_TAIL_DIAG
                        diag_pl = diag_cw = 0u;
#endif
#ifdef TRX_TAIL_DIAG
                        diag_t0 = tile_t0;
                        diag_chunk = chunk;
                        diag_tiles++;
#endif
                        s_wnode = c_wnode;
                        s_wtri = c_wtri;
                    }
                }
                const uint32_t n_take = min(n_free - given, chunk_left);
                if (rank >= given && rank < given + n_take) {
                    item = chunk_next + (rank - given);
                    my_tile = cur_tile;
                    if (MODE == kModeAo) my_vf = cur_vf;
                }
                chunk_next += n_take;
                chunk_left -= n_take;
                given += n_take;
            }
            const bool conv = kFused && pend; // this lane's primary ray hit something: it becomes the pixel's AO ray
            if (kFused && pend) item = TRX_INVALID; // (a waiting lane has no rank among the free ones)
            if (!has_ray && (item != TRX_INVALID || conv)) {
                bool ok = false;
                float dx = 0.0f, dy = 0.0f, dz = 0.0f;
                if (MODE == kModeRays) {
                    const float4 *rp = reinterpret_cast<const float4 *>(P.rays + item);
                    const float4 a = rp[0], b = rp[1];
                    r.ox = a.x; r.oy = a.y; r.oz = a.z; r.tmin = a.w;
                    dx = b.x; dy = b.y; dz = b.z;
                    t = fminf(b.w, TRX_F32_MAX);
                    out_index = item;
                    ok = true;
                } else {
                    // The launch parameters only a refill needs - the view, the image geometry, the primary hits - are read from
                    // the kernel-argument segment HERE, through a pointer the compiler cannot trace back to it: left alone it
                    // loads them once, ahead of the walk, and the walk then carries (or spills to vector-register lanes) some
                    // forty scalar registers it never reads.
                    // (Where refills are whole tiles - primary rays, a wave refills a few times per frame - or the kernel sits at
                    // the register budget; the single-level AO pass refills whenever sixteen lanes idle and would wait on
                    // these loads thirty times as often: +5 % measured, profiles/r04_ab_procs.log.)
                    constexpr bool kReload = MODE == kModePrimary || MODE == kModeFused || TLAS;
                    const TraceParams &R = kReload ? *refill_params() : P;
                    // whole-tile refills only when frames are batched, so the frame is wave-uniform
                    uint32_t local_tile = my_tile, frame_base = 0u, vf = 0u, seed = R.frame;
                    if (MODE == kModeAo) {
                        // an AO batch shares its view and its primary hits; the lanes of a wave may hold different seeds
                        // (mid-tile refills stay on), so nothing here is wave-uniform
                        if (R.n_frames > 1u) {
                            frame_base = my_vf * R.frame_stride;
                            seed += my_vf;
                        }
                    } else if (R.n_frames > 1u) {
                        vf = __builtin_amdgcn_readfirstlane(div_uniform(my_tile, R.tiles_per_frame, R.rcp_tiles_per_frame));
                        local_tile = my_tile - vf * R.tiles_per_frame;
                        frame_base = vf * R.frame_stride;
                    }
                    const ViewDev &view = R.views[vf];
                    uint32_t px, py;
                    if (conv) {
                        // the pixel this lane's primary ray belonged to, back from its record index
                        if (R.compact) {
                            const uint32_t tile = (out_index >> 6) * R.shard_count + R.shard_index, k = out_index & 63u;
                            const uint32_t ty = div_uniform(tile, R.tiles_x, R.rcp_tiles_x);
                            px = (tile - ty * R.tiles_x) * 8u + (k & 7u);
                            py = ty * 8u + (k >> 3);
                        } else {
                            py = div_uniform(out_index, R.width, R.rcp_width);
                            px = out_index - py * R.width;
                        }
                    } else {
                        const uint32_t tile = local_tile * R.shard_count + R.shard_index;
                        const uint32_t k = item & 63u;
                        const uint32_t ty = div_uniform(tile, R.tiles_x, R.rcp_tiles_x);
                        px = (tile - ty * R.tiles_x) * 8u + (k & 7u);
                        py = ty * 8u + (k >> 3);
                        if (px < R.width && py < R.height) out_index = frame_base + (R.compact ? local_tile * 64u + k : py * R.width + px);
                    }
                    if (px < R.width && py < R.height) {
                        primary_dir(view, R.width, R.height, px, py, dx, dy, dz);
                        if (MODE == kModePrimary || (kFused && !conv)) {
                            r.ox = view.eye[0]; r.oy = view.eye[1]; r.oz = view.eye[2];
                            if (kFused) is_ao = false;
                            ok = true;
                        } else {
                            trx_hit ph;
                            if (kFused) { // the primary hit is still in this lane's registers
                                ph.t = t;
                                ph.prim = prim;
                                is_ao = true;
                            } else {
                                ph = R.primary[out_index - frame_base];
                            }
                            if (ph.t < TRX_F32_MAX && ph.prim != TRX_INVALID) {
                                // normal of the hit triangle, flipped toward the viewer
                                const float4 *tp = P.tris + (size_t)ph.prim * 3;
                                float nx = tp[0].w, ny = tp[1].w, nz = tp[2].w; // cross(e1, e2)
                                if (TLAS && R.inst_xform) {
                                    // object-space normal -> world: transpose of world-to-object
                                    const uint32_t pi = kFused ? hit_inst : R.primary_inst[out_index - frame_base];
                                    if (pi != TRX_INVALID) {
                                        const float4 *m = R.inst_xform + (size_t)pi * 3;
                                        const float4 r0 = m[0], r1 = m[1], r2 = m[2];
                                        const float ax = (r0.x * nx + r1.x * ny) + r2.x * nz;
                                        const float ay = (r0.y * nx + r1.y * ny) + r2.y * nz;
                                        const float az = (r0.z * nx + r1.z * ny) + r2.z * nz;
                                        nx = ax; ny = ay; nz = az;
                                    }
                                }
                                const float ninv = 1.0f / sqrtf(dot3(nx, ny, nz, nx, ny, nz));
                                nx *= ninv; ny *= ninv; nz *= ninv;
                                const float nd = (nx * -dx + ny * -dy) + nz * -dz;
                                const float sg = copysignf(1.0f, nd);
                                nx *= sg; ny *= sg; nz *= sg;
                                r.ox = (view.eye[0] + dx * ph.t) - dx * R.ao_eps;
                                r.oy = (view.eye[1] + dy * ph.t) - dy * R.ao_eps;
                                r.oz = (view.eye[2] + dz * ph.t) - dz * R.ao_eps;
                                const float u1 = hash_noise(px, py, seed);
                                const float u2 = hash_noise(px, py, seed + 1024u);
                                const float rr = sqrtf(u1);
                                const float theta = u2 * 6.28318530717958647692f;
                                float sn, cs;
                                sincos_det(theta, sn, cs);
                                const float lx = rr * cs, ly = rr * sn, lz = sqrtf(fmaxf(0.0f, 1.0f - u1));
                                const float sign = nz >= 0.0f ? 1.0f : -1.0f;
                                const float a = -1.0f / (sign + nz);
                                const float bb = nx * ny * a;
                                const float b1x = 1.0f + sign * nx * nx * a, b1y = sign * bb, b1z = -sign * nx;
                                const float b2x = bb, b2y = sign + ny * ny * a, b2z = -ny;
                                dx = (b1x * lx + b2x * ly) + nx * lz;
                                dy = (b1y * lx + b2y * ly) + ny * lz;
                                dz = (b1z * lx + b2z * ly) + nz * lz;
                                const float dinv = 1.0f / sqrtf(dot3(dx, dy, dz, dx, dy, dz));
                                dx *= dinv; dy *= dinv; dz *= dinv;
                                ok = true;
                            } else if (!kFused) { // (a fused frame writes its AO misses where the primary ray ends)
                                trx_hit miss;
                                miss.t = __builtin_inff();
                                miss.prim = TRX_INVALID;
                                P.out[out_index] = miss;
                                if (TLAS && P.out_inst) P.out_inst[out_index] = TRX_INVALID;
                            }
                        }
                        if (MODE != kModeRays) {
                            r.tmin = 0.0f;
                            t = TRX_F32_MAX;
                        }
                    }
                }
                if (kFused) pend = false;
                if (ok) {
                    finish_ray_dir(r, dx, dy, dz);
                    lds_ray[2u * lane] = make_float4(r.ox, r.oy, r.oz, r.tmin);
                    lds_ray[2u * lane + 1u] = make_float4(r.dx, r.dy, r.dz, 0.0f);
                    prim = TRX_INVALID;
                    sp = 0;
                    steps = trip; // the wave trip this ray starts at
                    overflow = 0u;
                    cur = make_uint2(0u, 0x80000000u);
                    if (TLAS) {
                        tlas_sp = TRX_INVALID;
                        bvh_off = P.tlas_start;
                        cur_inst = hit_inst = TRX_INVALID;
                        if (P.inst_xform) { // (what the walk comes back to when it leaves a BLAS: kept in the wave's HBM area)
                            wray[0 * kWave + lane] = r.ox; wray[1 * kWave + lane] = r.oy; wray[2 * kWave + lane] = r.oz;
                            wray[3 * kWave + lane] = dx; wray[4 * kWave + lane] = dy; wray[5 * kWave + lane] = dz;
                        }
                    }
                    has_ray = true;
                }
            }
        }
        TRX_STAMP(k_refill);
        if (__ballot(has_ray) == 0ull) {
            if (exhausted) {
                if (kMerge && merge_open && wave_in_block == 0u) {
                    // leaving: close the door, or find that the second wave has just offered its rays
                    uint32_t old = 0u;
                    if (lane == 0u) old = atomicCAS(&merge_ctl[0], 0u, kMergeClosed);
                    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
                    merge_open = false;
                    if (old != 0u) {
                        merge_open = true;
                        need_take = old;
                    }
                }
                if (!(kMerge && need_take != 0u)) break;
            } else {
                continue;
            }
        }

        // ---- traverse ------------------------------------------------------------
        // The loop is wave-uniform (every lane iterates, work is predicated on `act`), so that the
        // triangle phase can use all 64 lanes whichever lanes own the triangles.
        const uint32_t keep = kWave - refill_idle; // leave when this few lanes remain
        // ---- triangle phase ---------------------------------------------------------
            // Each lane owns cnt triangle tests (the hit leaves of its node, highest bit first).
            // Few per lane: every owner tests its own, one round per triangle.  Otherwise the
            // wave's (ray, triangle) pairs are laid out densely over all 64 lanes: pair g goes to
            // lane g % 64, tests read the owner's ray from LDS, and owners then commit their
            // results in order, so a ray's triangle sequence (and its tie rule) is unchanged.
        auto triangle_phase = [&](uint2 tri) {
            const uint32_t cnt = (uint32_t)__popc(tri.y);
            if (__ballot(cnt != 0u) != 0ull) {
                if (COUNT) {
                    uint32_t mx = cnt, sum = cnt;
                    for (int off = 32; off > 0; off >>= 1) {
                        mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
                        sum += (uint32_t)__shfl_xor((int)sum, off);
                    }
                    if (lane == 0) {
                        atomicAdd(&P.ctr->hist_max[min(mx, 15u)], 1u);
#ifdef TRX_DEV_TUNE
                        if (!(P.tune & 0x100u))
#endif
                            atomicAdd(&P.ctr->hist_total[min((sum + 7u) >> 3, 15u)], 1u);
                    }
                }
                // Cooperative rounds cost about three per-lane rounds of VALU work (scans, owner look-up, the LDS
                // hand-offs), so they only pay when the wave's triangles sit in few lanes: 64 coherent rays testing
                // the same two triangles are 128 pairs = 2 cooperative rounds, but also just 2 per-lane rounds.
                bool coop = false;
                uint32_t incl = 0u, total = 0u, mx = 1u;
                if (__ballot(cnt >= P.tri_compact_min) != 0ull) {
                    incl = wave_scan_add(cnt);
                    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    mx = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_max(cnt), 63);
                    // per-lane rounds carry kBatch triangles of a lane each, cooperative rounds 64 pairs of the wave
                    coop = (mx + (uint32_t)kBatch - 1u) / (uint32_t)kBatch > P.tri_coop_ratio * ((total + 63u) >> 6);
                }
#ifdef TRX_TAIL_DIAG
                if (coop) diag_cw += (total + 63u) >> 6; else diag_pl += mx; // cooperative windows / per-lane rounds of the tile
#endif
                if (!coop) {
                    // Per-lane rounds, up to kBatch triangles of a lane per round: their records are requested
                    // together (one memory round trip per round instead of one per triangle; on incoherent rays the
                    // triangle phase is 55-60 % of a trip and most of that is these round trips) and tested in the
                    // lane's own order, highest bit first, each against the t its predecessor left.
                    while (tri.y != 0u) {
                        uint32_t gidx[kBatch];
                        bool have[kBatch];
                        float4 ta[kBatch], tb[kBatch], tc[kBatch];
#pragma unroll
                        for (int k = 0; k < kBatch; k++) {
                            have[k] = tri.y != 0u;
                            const uint32_t local = 31u - (uint32_t)__builtin_clz(tri.y | 1u);
                            tri.y &= ~(1u << local);
                            gidx[k] = tri.x + local;
                            ta[k] = tb[k] = tc[k] = unspecified4(); // no lane without a triangle reads them
                            if (have[k]) {
                                const float4 *tp = P.tris + (size_t)gidx[k] * 3;
                                ta[k] = tp[0];
                                tb[k] = tp[1];
                                tc[k] = tp[2];
                            }
                        }
                        // keep the loads of a round together and ahead of the arithmetic: left alone the compiler
                        // sinks them behind the first determinant, serialising the memory latencies again
#pragma unroll
                        for (int k = 0; k < kBatch; k++)
                            asm volatile("" : "+v"(ta[k].x), "+v"(ta[k].y), "+v"(ta[k].z), "+v"(ta[k].w), "+v"(tb[k].x), "+v"(tb[k].y), "+v"(tb[k].z), "+v"(tb[k].w), "+v"(tc[k].x), "+v"(tc[k].y), "+v"(tc[k].z), "+v"(tc[k].w));
#pragma unroll
                        for (int k = 0; k < kBatch; k++) {
                            if (have[k]) {
                                if (COUNT) {
                                    c_tri++;
                                    if (lane_rank(__ballot(1)) == 0) c_wtri++;
                                    if (P.touch_tris) P.touch_tris[gidx[k]] = 1;
                                }
                                if (intersect_tri<true>(r, ta[k], tb[k], tc[k], t, tie_first)) {
                                    prim = gidx[k];
                                    if (TLAS) hit_inst = cur_inst;
                                }
                            }
                        }
                    }
                } else {
                    // Cooperative rounds: the wave's (ray, triangle) pairs laid out densely, pair g on lane g % 64.  The
                    // tester finds its owner (a max-scan over the run heads), picks the owner's j-th triangle, tests it
                    // against the owner's ray and folds the result into the owner's 64-bit key with ONE LDS atomic min:
                    // key = {order-preserving image of t, tie word}.  The owner's key starts as its current t with a tie
                    // word that wins (TIE_FIRST: tt < t commits) or loses (tt <= t commits) every tie, and a pair's
                    // tie word is its place in the owner's own sequence (highest bit first), so the minimum IS the
                    // result of the sequential loop "for each triangle in order: if closer, commit": no per-owner
                    // commit loop, no result table, one barrier-free pass per window.
                    const uint32_t excl = incl - cnt;
                    lds_grp[lane] = tri;
                    lds_pref[lane] = excl;
                    const uint32_t init_lo = tie_first ? 0u : 0xffu;
                    unsigned long long *const lds_key = reinterpret_cast<unsigned long long *>(lds_res);
                    lds_res[lane] = make_uint2(init_lo, ordered_bits(t + 0.0f));
                    if (COUNT) c_tri += cnt;
                    for (uint32_t base = 0; base < total; base += kWave) {
                        // owner of every pair of this window: heads mark where each owner's run starts
                        // (the table is cleared for every window; tagging the heads with a window number instead, so
                        // that stale ones lose the max-scan and the clearing store goes, measured -1 % on one pass
                        // and was not kept)
                        lds_head[lane] = 0u;
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t run_begin = max(excl, base), run_end = min(excl + cnt, base + kWave);
                        if (run_begin < run_end) lds_head[run_begin - base] = lane + 1u;
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t owner1 = wave_scan_max(lds_head[lane]);
                        const uint32_t g = base + lane;
                        if (g < total) {
                            const uint32_t ol = (owner1 - 1u) & 63u; // (owner1 is 1..64; the mask keeps a corrupted table inside this wave's LDS)
                            const uint2 grp = lds_grp[ol];
                            const uint32_t local = select_from_top(grp.y, g - lds_pref[ol]);
                            const uint32_t gidx = grp.x + local;
                            if (COUNT && P.touch_tris) P.touch_tris[gidx] = 1;
                            const float4 *tp = P.tris + (size_t)gidx * 3;
                            float4 a = tp[0], b = tp[1], c4 = tp[2];
                            const float4 ro = lds_ray[2u * ol], rd = lds_ray[2u * ol + 1u];
                            asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w), "+v"(c4.x), "+v"(c4.y), "+v"(c4.z), "+v"(c4.w));
                            Ray orr;
                            orr.ox = ro.x; orr.oy = ro.y; orr.oz = ro.z; orr.tmin = ro.w;
                            orr.dx = rd.x; orr.dy = rd.y; orr.dz = rd.z;
                            float tt = TRX_F32_MAX; // the tie test against the owner's t is the atomic min
                            if (intersect_tri(orr, a, b, c4, tt, false)) {
                                // -0.0 and +0.0 are one value to the sequential compare: they share a key, and the
                                // tie word's lowest bit remembers which one the pair really produced
                                const uint32_t neg_zero = __float_as_uint(tt) == 0x80000000u ? 1u : 0u;
                                const uint32_t lo = ((tie_first ? 31u - local : local) << 1) | neg_zero;
                                const unsigned long long key = ((unsigned long long)ordered_bits(tt + 0.0f) << 32) | lo;
                                atomicMin(&lds_key[ol], key);
                            }
                        }
                        if (COUNT && lane == 0) c_wtri++;
                    }
                    __builtin_amdgcn_wave_barrier();
                    const uint2 won = lds_res[lane];
                    if (cnt != 0u && won.x != init_lo) { // some pair of this lane beat (or tied its way past) the current t
                        const uint32_t local = tie_first ? 31u - (won.x >> 1) : (won.x >> 1);
                        t = (won.x & 1u) ? -0.0f : __uint_as_float(unordered_bits(won.y));
                        prim = tri.x + local;
                        if (TLAS) hit_inst = cur_inst;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }

        };




        // The second wave of a draining workgroup offers its rays (see "The drain" above); true = taken over, this wave is empty.
        auto merge_offer = [&](uint32_t alive) -> bool {
            const uint32_t avail = *reinterpret_cast<volatile uint32_t *>(&merge_ctl[1]);
            if (avail < alive || __ballot(has_ray && sp > (uint32_t)kLdsStack) != 0ull) return false; // (stacks past the LDS part stay)
            const uint32_t j = lane_rank(__ballot(has_ray));
            if (has_ray) {
                uint32_t *m = merge_box + j * kMergeWords;
                m[0] = __float_as_uint(r.ox); m[1] = __float_as_uint(r.oy); m[2] = __float_as_uint(r.oz);
                m[3] = __float_as_uint(r.dx); m[4] = __float_as_uint(r.dy); m[5] = __float_as_uint(r.dz);
                m[6] = __float_as_uint(r.ix); m[7] = __float_as_uint(r.iy); m[8] = __float_as_uint(r.iz);
                m[9] = __float_as_uint(r.tmin); m[10] = r.oct_inv4;
                m[11] = __float_as_uint(t); m[12] = prim; m[13] = out_index; m[14] = trip - steps;
                m[15] = cur.x; m[16] = cur.y; m[17] = sp; m[18] = ptri.x; m[19] = ptri.y; m[20] = lane;
                if (TLAS) {
                    m[21] = bvh_off; m[22] = tlas_sp; m[23] = cur_inst; m[24] = hit_inst;
                    for (int c = 0; c < 6; c++) m[25 + c] = __float_as_uint(wray[c * kWave + lane]); // (a ray's world-space copy travels with it)
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uint32_t old = 0u;
            if (lane == 0u) old = atomicCAS(&merge_ctl[0], 0u, alive);
            old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
            merge_open = false; // one offer per wave
            if (old != 0u) { // the first wave has left: finish them here (the parking area covered this wave's ray copies)
                if (has_ray) {
                    lds_ray[2u * lane] = make_float4(r.ox, r.oy, r.oz, r.tmin);
                    lds_ray[2u * lane + 1u] = make_float4(r.dx, r.dy, r.dz, 0.0f);
                }
                return false;
            }
            has_ray = false;
            return true;
        };
        // The first wave takes what the second offered, if anything (n = control word 0).
        auto merge_take = [&](uint32_t n) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t j = lane_rank(__ballot(!has_ray));
            if (!has_ray && j < n) {
                const uint32_t *m = merge_box + j * kMergeWords;
                r.ox = __uint_as_float(m[0]); r.oy = __uint_as_float(m[1]); r.oz = __uint_as_float(m[2]);
                r.dx = __uint_as_float(m[3]); r.dy = __uint_as_float(m[4]); r.dz = __uint_as_float(m[5]);
                r.ix = __uint_as_float(m[6]); r.iy = __uint_as_float(m[7]); r.iz = __uint_as_float(m[8]);
                r.tmin = __uint_as_float(m[9]); r.oct_inv4 = m[10];
                t = __uint_as_float(m[11]); prim = m[12]; out_index = m[13]; steps = trip - m[14];
                cur = make_uint2(m[15], m[16]); sp = m[17]; ptri = make_uint2(m[18], m[19]);
                if (TLAS) {
                    bvh_off = m[21]; tlas_sp = m[22]; cur_inst = m[23]; hit_inst = m[24];
                    for (int c = 0; c < 6; c++) wray[c * kWave + lane] = __uint_as_float(m[25 + c]);
                }
                const uint32_t from = m[20];
                for (uint32_t k = 0; k < sp; k++) lds_st(&lds_stack[k * kWave + lane], merge_stack1[k * kWave + from]);
                lds_ray[2u * lane] = make_float4(r.ox, r.oy, r.oz, r.tmin);
                lds_ray[2u * lane + 1u] = make_float4(r.dx, r.dy, r.dz, 0.0f);
                fetched = false;
                overflow = 0u;
                has_ray = true;
            }
            merge_open = false;
        };
        // End of a trip of a dry wave: the second wave offers once its rays fit the first one's idle lanes; the first
        // publishes its idle lanes and looks for an offer.
        auto merge_step = [&](uint32_t alive) {
            if (wave_in_block == 1u) {
                // (a wave that is down to a handful of rays finishes them itself, eight lanes to a ray - thin_walk - rather
                // than hand them to a wave that steps them one lane each)
                if (kThin && alive <= P.thin_max) merge_open = false;
                else if (alive != 0u && alive <= kMergeMax) (void)merge_offer(alive);
            } else {
                if (lane == 0u) *reinterpret_cast<volatile uint32_t *>(&merge_ctl[1]) = (uint32_t)kWave - alive;
                const uint32_t n = *reinterpret_cast<volatile uint32_t *>(&merge_ctl[0]);
                if (n != 0u) {
                    merge_take(n);
                } else if (kThin && alive <= P.thin_max) {
                    // ... and the first wave, down to a handful, closes the door - unless an offer has just landed
                    uint32_t old = 0u;
                    if (lane == 0u) old = atomicCAS(&merge_ctl[0], 0u, kMergeClosed);
                    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
                    if (old == 0u) merge_open = false;
                    else merge_take(old);
                }
            }
        };

        // Incoherent passes end when their longest rays do (a third of the hairball-class AO pass's wave time is waves
        // waiting for them, and dealing the tiles longest ray first does not move that tail: profiles/r03_ao_order.log, tools/gpu_timeline_ao.py), so
        // a wave that holds an old ray wins the issue arbitration of its SIMD: priority 2 once its oldest ray has run
        // kOldRay trips, 3 from twice that, looked at every eighth trip.  Worth 1-2 % of an AO pass (four alternating
        // repetitions on four scenes, all four means lower; the run-to-run spread of an AO pass is +-2 %).
        constexpr uint32_t kOldRay = 32u;
        auto old_ray_priority = [&]() {
            const uint32_t age = has_ray ? trip - steps : 0u;
            if (__ballot(age >= 2u * kOldRay) != 0ull) __builtin_amdgcn_s_setprio(3);
            else if (__ballot(age >= kOldRay) != 0ull) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
        };

        if (kMerge && need_take != 0u) {
            merge_take(need_take);
            need_take = 0u;
        }
        if constexpr (!PIPE) {
            for (;;) {
                const bool act = has_ray;
                const bool pow2 = (NODE & 1) ? false : pow2_exact(P, r, act); // (literal-division variants only)
                uint2 tri = make_uint2(0u, 0u);
                trip++;
#ifdef TRX_DEV_TUNE
                if (!(P.tune & 0x10000000u)) // (A/B: no priority by ray age)
#endif
                if (!TLAS && MODE != kModePrimary && (trip & 7u) == 0u) old_ray_priority(); // (measured without effect in the two-level kernels)
                // Coherent primary rays (BLAS only): when every lane that steps visits the SAME node - 47 % of the wave-level
                // node steps on the bistro-class frame, 90 % on the kitchen-class one - its 48 quantised plane bytes are
                // converted once, one byte per lane, parked in LDS as floats and read back by address (node_intersect_dec)
                // (two-level walks too since round 5 - absolute node indices, lanes with a parked triangle group excluded: in round 4
                // the two-level primary kernel spilled with it (4K frame +4 %, profiles/r04_ab_procs_16); with the world-space ray
                // copies out of the registers (kernels.h, kWaveScratch) it has 121 and the 4K frame runs 3.3 % faster,
                // profiles/r05_ab_5_tlas.log)
                constexpr bool kUni = MODE == kModePrimary && !COUNT;
                bool uni_done = false;
                if constexpr (kUni) {
#ifndef TRX_UNI_COOL
#define TRX_UNI_COOL 1
#endif
                    // (the look itself is some thirty instructions: after a trip whose lanes wanted different nodes the next
                    // TRX_UNI_COOL trips do not look - one trip: bistro-class frame -0.5 %, hairball-class -0.8 %, kitchen-class
                    // -0.2 %; two: -0.1 / -1.2 / +0.5 %, profiles/r05_ab_7_unicool.log)
                    if (TRX_UNI_COOL && uni_cool != 0u) {
                        uni_cool--;
                    } else
                    if (P.uni_decode) {
                        uint32_t node_index = 0u, child_bit = 0u;
                        // (two-level walks: a lane may hold a parked triangle group instead of a node group, and node
                        // indices are absolute - rays inside different instances of one BLAS do share its nodes, each
                        // with its own object-space ray)
                        const bool group = act && (!TLAS || (cur.y & 0xff000000u) != 0u);
                        if (group) {
                            const uint32_t hits_imask = cur.y;
                            child_bit = 31u - (uint32_t)__builtin_clz(hits_imask);
                            const uint32_t slot = (child_bit - 24u) ^ (r.oct_inv4 & 0xffu);
                            node_index = cur.x + (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
                            if (TLAS) node_index += bvh_off;
                        }
                        const unsigned long long stepping = __ballot(act);
                        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)node_index, (int)(__ffsll((long long)stepping) - 1));
                        if (__ballot(act && (!group || node_index != first)) == 0ull) { // wave-uniform: every lane takes this branch or none does
                            const uint4 *np = P.nodes + (size_t)first * 5;
                            // 48 lanes convert one byte each (whether or not they hold a ray), LDS hands the floats to all
                            // (byte 32 + 8 p + c = plane p of child c, planes in the order min_x max_x min_y max_y min_z max_z; the
                            // second table lives in the triangle phase's window area, idle during a node step)
                            if (lane < 48u) {
                                const float v = (float)reinterpret_cast<const uint8_t *>(np)[32u + lane];
                                const uint32_t slot = (lane >> 4) * 16u + (lane & 7u) * 2u, is_max = (lane >> 3) & 1u;
                                lds_dec[slot + is_max] = v;
                                lds_dec_neg[slot + (is_max ^ 1u)] = v;
                            }
                            __builtin_amdgcn_wave_barrier();
                            if (act) {
                                const uint4 n0 = np[0], n1 = np[1];
                                cur.y &= ~(1u << child_bit);
                                stack_push(cur, (cur.y & 0xff000000u) != 0u);
                                const uint32_t hitmask = node_intersect_dec<NODE>(r, t, n0, n1, lds_dec, lds_dec_neg, pow2);
                                cur.x = n1.x;
                                tri.x = n1.y;
                                cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                                tri.y = hitmask & 0x00ffffffu;
                            }
                            __builtin_amdgcn_wave_barrier();
                            uni_done = true;
                        } else if (TRX_UNI_COOL) {
                            uni_cool = TRX_UNI_COOL;
                        }
                    }
                }
                if (act && !uni_done) {
                    // BLAS-only: a lane at the top of the loop always holds a node group (triangle groups are drained
                    // in the trip that found them; only the TLAS walk parks them on the stack)
                    if (!TLAS || (cur.y & 0xff000000u)) {
                        const uint32_t hits_imask = cur.y;
                        const uint32_t child_bit = 31u - (uint32_t)__builtin_clz(hits_imask); // hits_imask != 0
                        const uint32_t child_base = cur.x;
                        cur.y &= ~(1u << child_bit);
                        const uint32_t slot = (child_bit - 24u) ^ (r.oct_inv4 & 0xffu);
                        const uint32_t rel = (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
                        uint32_t node_index = child_base + rel;
                        if (TLAS) node_index += bvh_off;
                        // (tried: when every lane wants the same node - 47 % of the wave-level steps on the bistro-class frame,
                        // 90 % on the kitchen-class one - one copy through the scalar cache instead of 64 through the vector
                        // path: no change in frame time)
                        const uint4 *np = P.nodes + (size_t)node_index * 5;
                        const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
                        stack_push(cur, (cur.y & 0xff000000u) != 0u); // after the loads are on their way
                        TRX_STAMP(k_fetch);
                        if (COUNT) {
                            c_node++;
                            if (lane_rank(__ballot(1)) == 0) c_wnode++;
                            if (P.touch_nodes) P.touch_nodes[node_index] = 1;
    #ifdef TRX_DEV_TUNE
                            if (P.tune & 0x100u) {
                                // diagnostics: distinct nodes among the lanes of this wave-level node step
                                unsigned long long todo = __ballot(1);
                                const uint32_t first = (uint32_t)__ffsll((long long)todo) - 1u;
                                uint32_t distinct = 0;
                                while (todo) {
                                    const uint32_t l = (uint32_t)__ffsll((long long)todo) - 1u;
                                    const uint32_t v = (uint32_t)__shfl((int)node_index, (int)l);
                                    todo &= ~__ballot(node_index == v);
                                    distinct++;
                                }
                                if (lane == first) atomicAdd(&P.ctr->hist_total[min(distinct, 15u)], 1u);
                            }
    #endif
                        }
                        const uint32_t hitmask = node_intersect<NODE>(r, t, n0, n1, n2, n3, n4, pow2);
                        cur.x = n1.x;
                        tri.x = n1.y;
                        cur.y = (hitmask & 0xff000000u) | (n0.w >> 24);
                        tri.y = hitmask & 0x00ffffffu;
    #ifdef TRX_DEV_TUNE
                        if (P.tune & 2u) tri.y = 0u; // ablation (timing only, results wrong): no triangle phase at all
    #endif
                    } else {
                        tri = cur;
                        cur = make_uint2(0u, 0u);
                    }
                }
                if (TLAS && act && tlas_sp == TRX_INVALID && tri.y != 0u) { // (after either kind of node step)
                    // a TLAS primitive is an instance (query_tlas.hlsl:410-446): enter its BLAS
                    const uint32_t local = 31u - (uint32_t)__clz((int)tri.y);
                    tri.y &= ~(1u << local);
                    const uint32_t gidx = tri.x + local;
                    stack_push(tri, tri.y != 0u);
                    stack_push(cur, (cur.y & 0xff000000u) != 0u);
                    tlas_sp = sp;
                    bvh_off = P.inst[gidx];
                    cur_inst = gidx;
                    if (P.inst_xform) {
                        // the ray in the instance's object space; the direction is not renormalised, so t keeps
                        // its world-space meaning (the TODO at query_tlas.hlsl:433)
                        const float4 *m = P.inst_xform + (size_t)gidx * 3;
                        const float4 r0 = m[0], r1 = m[1], r2 = m[2];
                        const float wox = wray[0 * kWave + lane], woy = wray[1 * kWave + lane], woz = wray[2 * kWave + lane];
                        const float wdx = wray[3 * kWave + lane], wdy = wray[4 * kWave + lane], wdz = wray[5 * kWave + lane];
                        r.ox = ((r0.x * wox + r0.y * woy) + r0.z * woz) + r0.w;
                        r.oy = ((r1.x * wox + r1.y * woy) + r1.z * woz) + r1.w;
                        r.oz = ((r2.x * wox + r2.y * woy) + r2.z * woz) + r2.w;
                        const float odx = (r0.x * wdx + r0.y * wdy) + r0.z * wdz;
                        const float ody = (r1.x * wdx + r1.y * wdy) + r1.z * wdz;
                        const float odz = (r2.x * wdx + r2.y * wdy) + r2.z * wdz;
                        finish_ray_dir(r, odx, ody, odz);
                        lds_ray[2u * lane] = make_float4(r.ox, r.oy, r.oz, r.tmin);
                        lds_ray[2u * lane + 1u] = make_float4(r.dx, r.dy, r.dz, 0.0f);
                    }
                    // the walk of the BLAS starts at its node 0 (query_tlas.hlsl:443) - or, for a TLAS primitive that
                    // stands for a SUBTREE of its BLAS (re-braided scenes, trx_scene_set_instance_entry_nodes), at
                    // that subtree's node: the group {child_base = entry, one hit} makes the next node step fetch it
                    cur = make_uint2(P.inst_entry ? P.inst_entry[gidx] : 0u, 0x80000000u);
                    tri.y = 0u;
                }

                TRX_STAMP(k_test);
                triangle_phase(tri);
                TRX_STAMP(k_tri);
    #ifdef TRX_STAMPS
                k_iters++;
    #endif
                if (act) {
                    // a lane whose node group is spent pops the next one, or is finished when its stack is empty
                    const bool spent = (cur.y & 0xff000000u) == 0u;
                    bool done = spent && sp == 0u;
                    if (spent && sp != 0u) {
                        if (TLAS && sp == tlas_sp) { // back to the TLAS (query_tlas.hlsl:480-486)
                            tlas_sp = TRX_INVALID;
                            bvh_off = P.tlas_start;
                            cur_inst = TRX_INVALID;
                            if (P.inst_xform) { // "Reset Ray to untransformed version" (query_tlas.hlsl:484)
                                r.ox = wray[0 * kWave + lane]; r.oy = wray[1 * kWave + lane]; r.oz = wray[2 * kWave + lane];
                                finish_ray_dir(r, wray[3 * kWave + lane], wray[4 * kWave + lane], wray[5 * kWave + lane]);
                                lds_ray[2u * lane] = make_float4(r.ox, r.oy, r.oz, r.tmin);
                                lds_ray[2u * lane + 1u] = make_float4(r.dx, r.dy, r.dz, 0.0f);
                            }
                        }
                        cur = stack_pop();
                        if (__builtin_expect(overflow != 0u, 0)) done = true; // past the last stack entry: the sentinel is not a group
                    }
                    // step cap (every wave reaches an exit whatever the tree): a ray's steps are bounded by the wave's
                    // trips since it started; looked at once per 1024 trips, so the common trip pays nothing for it
                    if (__builtin_expect((trip & 1023u) == 0u, 0)) {
                        if (trip - steps > kMaxSteps) {
                            overflow = 1u;
                            done = true;
                        }
                    }
                    // any-hit query (intersects_bl_bvh, query.hlsl:440-445): the first accepted triangle settles it.
                    // Until a hit is accepted the walk is the closest-hit walk, so hit / no hit is the same answer.
                    if (MODE == kModeRays && P.any_hit != 0u && prim != TRX_INVALID) done = true;
                    if (done) finish_lane();
                }
                const uint32_t alive = (uint32_t)__popcll(__ballot(has_ray));
                TRX_STAMP(k_pop);
                if (kMerge && merge_open && exhausted) {
                    merge_step(alive);
                    if (__ballot(has_ray) == 0ull) break;
                    continue;
                }
                if (alive == 0u || (!exhausted && alive <= keep)) break;
                // fused frame, queues dry: waiting lanes are turned into AO rays once enough of them have gathered
                if (kFused && exhausted && (uint32_t)__popcll(__ballot(pend)) >= P.pend_min) break;
                if (kThin && thin_now(alive)) {
                    go_thin = true;
                    break;
                }
            }

        } else {
            // Pipelined walk (BLAS only).  Per trip: (1) lanes whose group names a next node issue its fetch and push
            // the group's remainder; (2) the triangle phase of the node tested in the PREVIOUS trip runs under those
            // loads; (3) rays with nothing left to fetch are finished (their last triangles are in); (4) the node test
            // of the fetched node, with the t the triangles left, and the pop of a spent group.  Stack traffic, node
            // order, triangle order and the t every test sees are those of the plain walk.
            for (;;) {
                const bool act = has_ray;
                const bool pow2 = (NODE & 1) ? false : pow2_exact(P, r, act); // (literal-division variants only)
                trip++;
#ifdef TRX_DEV_TUNE
                if (!(P.tune & 0x10000000u)) // (A/B: no priority by ray age)
#endif
                if (!TLAS && MODE != kModePrimary && (trip & 7u) == 0u) old_ray_priority(); // (measured without effect in the two-level kernels)
                // The fetched node record lives from (1) to (4) of ONE trip and is deliberately left uninitialised here:
                // carried across trips (a variable of the kernel, as it was) the five loads merge with the previous
                // trip's values, the register allocator is free to land them in scratch registers and copy them home at
                // once - and that copy waits for the loads thirteen instructions after they were issued instead of a
                // triangle phase later.  Round 4 shipped that for a while (AO passes +5-8 %): tests/test_kernel_resources.py
                // now measures the distance from the fetch to the first wait in the compiled walk.
                uint4 fn0, fn1, fn2, fn3, fn4;
                // (1)
                if (act && !fetched && (cur.y & 0xff000000u)) {
                    const uint32_t hits_imask = cur.y;
                    const uint32_t child_bit = 31u - (uint32_t)__builtin_clz(hits_imask);
                    const uint32_t child_base = cur.x;
                    cur.y &= ~(1u << child_bit);
                    const uint32_t slot = (child_bit - 24u) ^ (r.oct_inv4 & 0xffu);
                    const uint32_t rel = (uint32_t)__popc(hits_imask & ~(0xffffffffu << slot));
                    const uint32_t node_index = child_base + rel;
                    const uint4 *np = P.nodes + (size_t)node_index * 5;
                    fn0 = np[0]; fn1 = np[1]; fn2 = np[2]; fn3 = np[3];
                    {   // (the last 16 bytes as two 8-byte loads: pairs of registers are easier to keep than a fifth quadruple)
                        const uint2 *h = reinterpret_cast<const uint2 *>(np + 4);
                        const uint2 a = h[0], b = h[1];
                        fn4 = make_uint4(a.x, a.y, b.x, b.y);
                    }
                    stack_push(cur, (cur.y & 0xff000000u) != 0u); // after the loads are on their way
                    fetched = true;
                    if (COUNT) {
                        c_node++;
                        if (lane_rank(__ballot(1)) == 0) c_wnode++;
                        if (P.touch_nodes) P.touch_nodes[node_index] = 1;
                    }
                }
                TRX_STAMP(k_fetch);
                // (2)
                triangle_phase(ptri);
                ptri = make_uint2(0u, 0u);
                TRX_STAMP(k_tri);
#ifdef TRX_STAMPS
                k_iters++;
#endif
                // (3)
                if (act) {
                    bool done = !fetched; // no node left to visit, and the last node's triangles are in
                    if (MODE == kModeRays && P.any_hit != 0u && prim != TRX_INVALID) done = true;
                    if (__builtin_expect((trip & 1023u) == 0u, 0)) { // step cap: every wave reaches an exit
                        if (trip - steps > kMaxSteps) {
                            overflow = 1u;
                            done = true;
                        }
                    }
                    if (done) {
                        fetched = false; // a fetch still in flight (any-hit, step cap) is simply not looked at
                        finish_lane();
                    }
                }
                const uint32_t alive = (uint32_t)__popcll(__ballot(has_ray));
                const bool leave = alive == 0u || (!exhausted && alive <= keep);
                TRX_STAMP(k_pop);
                // (4)
                if (fetched) {
                    const uint32_t hitmask = node_intersect<NODE>(r, t, fn0, fn1, fn2, fn3, fn4, pow2);
                    cur.x = fn1.x;
                    ptri.x = fn1.y;
                    cur.y = (hitmask & 0xff000000u) | (fn0.w >> 24);
                    ptri.y = hitmask & 0x00ffffffu;
                    fetched = false;
                    if ((cur.y & 0xff000000u) == 0u && sp != 0u) {
                        cur = stack_pop();
                        if (__builtin_expect(overflow != 0u, 0)) cur = make_uint2(0u, 0u); // past the last entry: finish
                    }
                }
                TRX_STAMP(k_test);
                if (kMerge && merge_open && exhausted) {
                    merge_step(alive);
                    if (__ballot(has_ray) == 0ull) break;
                    continue;
                }
                if (leave) break;
                if (kFused && exhausted && (uint32_t)__popcll(__ballot(pend)) >= P.pend_min) break;
                if (kThin && thin_now(alive)) {
                    go_thin = true;
                    break;
                }
            }
        }
        if (kThin && go_thin) {
            // (nothing comes back from the thin walk of a dry wave - except a fused frame's lane that waits to become an AO
            // ray - so it runs AFTER the loop, where the registers the loop carries are dead)
            if (!kFused) break;
            go_thin = false;
            thin_all(); // returns with no ray left, or with a lane waiting to become an AO ray
        }
    }
    if (kThin && !kFused && go_thin) thin_all(); // returns with no ray left

    // ---- epilogue: flags, counters, queue reset ---------------------------------------
    if (lpt_write && n_pend) flush_pending(P, wr_set, lds_pend, n_pend, lane);
    if (c_over) {
        atomicAdd(&P.ctr->overflow, c_over);
        if (P.over_host) *P.over_host = 1u;
    }
    if (COUNT) {
        atomicAdd(&P.ctr->n_rays, (unsigned long long)c_rays);
        atomicAdd(&P.ctr->n_node, (unsigned long long)c_node);
        atomicAdd(&P.ctr->n_tri, (unsigned long long)c_tri);
        atomicAdd(&P.ctr->n_wave_node, (unsigned long long)c_wnode);
        atomicAdd(&P.ctr->n_wave_tri, (unsigned long long)c_wtri);
        atomicAdd(&P.ctr->n_hits, (unsigned long long)c_hits);
        atomicMax(&P.ctr->max_stack, c_maxsp);
    }
    if (lane == 0) {
        // the last wave out re-arms the queue for the next launch on this slot
        if (P.wave_times) {
            P.wave_times[kWaveTimeStride * wave_global + 1] = wall_clock64();
#ifdef TRX_TAIL_DIAG
            P.wave_times[kWaveTimeStride * wave_global + 2] = diag_t0;     // start of the wave's last tile
            P.wave_times[kWaveTimeStride * wave_global + 3] = diag_chunk;  // its position in the frame's order
            P.wave_times[kWaveTimeStride * wave_global + 4] = diag_tiles;  // tiles the wave traced
            P.wave_times[kWaveTimeStride * wave_global + 5] = diag_dry_t;
            P.wave_times[kWaveTimeStride * wave_global + 6] = diag_dry_alive;
            P.wave_times[kWaveTimeStride * wave_global + 7] = ((unsigned long long)trip << 32) | diag_dry_age; // trips of the wave | oldest ray at that moment
#endif
#ifdef TRX_STAMPS
            unsigned long long *wt = P.wave_times + kWaveTimeStride * wave_global;
            wt[2] = k_refill; wt[3] = k_fetch; wt[4] = k_test; wt[5] = k_tri; wt[6] = k_pop; wt[7] = k_iters;
#endif
        }
        const unsigned int ticket = atomicAdd(&P.ctr->waves_done, 1u);
        if (ticket == gridDim.x * (blockDim.x / kWave) - 1u) {
            for (int q = 0; q < 8; q++) atomicExch(&P.ctr->heads[q].taken, 0u);
            // the lists this frame consumed become the next frame's (empty) write lists
            // a frame that filed a new order: the set it read is emptied (it takes the next new order) and the selector flips
            if (lpt_write) {
                for (uint32_t b = 0; b < 16u * kLptShards; b++) atomicExch(&rd_set[b], 0u);
                atomicExch(P.lpt_sel, lpt_rd ? 0u : 1u);
            }
            if (kOrder && P.fb) {
                // Which schedule suits this slot's frames?  The tile-order feedback costs about 2 us a tile (timing, the list
                // look-up behind the queue atomic, the appends) and repays that many times over where a few tiles set the
                // frame's critical path - not on a room seen from inside (the kitchen-class frame runs 13 % faster without
                // it), and a frame of ragged tiles (the hairball-class one: a few rays of a tile run ten times longer than
                // the rest) does better still replacing finished rays mid-tile.  So the slot measures: kFbOn frames ordered
                // (the first three relearn the order and do not count), kFbProbe frames in natural order, kFbProbe with
                // mid-tile refills; the ordered mode stays unless another is 3 % faster; the winner holds for kFbHold frames,
                // then everything is measured again.  Frame time = last wave out minus first wave in, best of a phase.
                // (kFbOn was 24 until round 4: a bench protocol of 5 warm-up + 20 timed frames then had its last timed
                // frames run as natural-order probes, 20-40 % slower each; a slot now runs ordered for its first 128 frames)
                constexpr unsigned int kFbOn = 128u, kFbProbe = 4u, kFbHold = 1024u;
                FbState &c = *P.fb;
                if (P.new_view) c.mode = c.frames = c.phase = c.t[0] = c.t[1] = c.t[2] = 0u; // a new view measures afresh
                const unsigned int dur = (unsigned int)min(wall_clock64() - c.t0, 0xffffffffull);
                const unsigned int f = c.frames + 1u;
                const unsigned int phase = c.phase;
                if (phase < 3u) {
                    // measuring mode `phase`: the ordered mode needs three frames to relearn its order, the others one
                    const unsigned int skip = phase == 0u ? 3u : 1u, len = phase == 0u ? kFbOn : kFbProbe;
                    if (f > skip) c.t[phase] = c.t[phase] ? min(c.t[phase], dur) : dur;
                    if (f < len) {
                        c.frames = f;
                    } else if (phase < 2u && !(phase == 1u && P.n_frames != 1u)) {
                        c.phase = phase + 1u; // next candidate
                        c.mode = phase + 1u;
                        c.frames = 0u;
                        c.t[phase + 1u] = 0u;
                    } else {
                        // decision: the ordered mode unless another is clearly (3 %) faster; of those, the faster
                        unsigned int best = 0u;
                        unsigned long long t_best = (unsigned long long)c.t[0] * 97ull;
                        for (unsigned int m = 1u; m <= phase; m++)
                            if (c.t[m] != 0u && c.t[0] != 0u && (unsigned long long)c.t[m] * 100ull < t_best) {
                                best = m;
                                t_best = (unsigned long long)c.t[m] * 100ull;
                            }
                        c.mode = best;
                        c.phase = 3u;
                        c.frames = 0u;
                    }
                } else if (f >= kFbHold) { // held long enough: measure again, from the ordered mode
                    c.mode = 0u;
                    c.phase = 0u;
                    c.frames = 0u;
                    c.t[0] = 0u;
                } else {
                    c.frames = f;
                }
            }
            atomicExch(&P.ctr->waves_done, 0u);
        }
    }
}


#undef wave_global

template <int MODE, bool TLAS, int NODE, bool PIPE, bool COUNT>
hipError_t launch_one(const TraceParams &p, int grid, hipStream_t stream) {
    // grid = total waves; p.waves_per_block waves share a workgroup (and nothing else)
    const int wpb = (int)p.waves_per_block;
    const size_t lds = (size_t)wpb * kLdsBytesPerWave;
    hipLaunchKernelGGL((k_trace<MODE, TLAS, NODE, PIPE, COUNT>), dim3(grid / wpb), dim3(kWave * wpb), lds, stream, p);
    return hipGetLastError();
}

template <int MODE, bool TLAS, bool PIPE, bool COUNT>
hipError_t launch_node(const TraceParams &p, int node, int grid, hipStream_t stream) {
    switch (node) {
    case 0: return launch_one<MODE, TLAS, 0, PIPE, COUNT>(p, grid, stream);
    case 1: return launch_one<MODE, TLAS, 1, PIPE, COUNT>(p, grid, stream);
    case 2: return launch_one<MODE, TLAS, 2, PIPE, COUNT>(p, grid, stream);
    default: return launch_one<MODE, TLAS, 3, PIPE, COUNT>(p, grid, stream);
    }
}

template <int MODE>
hipError_t launch_mode(const TraceParams &p, bool tlas, int node, bool count, bool pipe, int grid, hipStream_t stream) {
    // (the one-launch frame exists for single-level scenes: the two-level walk has no registers to spare for the in-place
    // hand-over - it spills - so trx_trace_frame_dev runs a two-level frame as two launches, api.cpp)
    if constexpr (MODE == kModeFused) {
        if (tlas) return hipErrorInvalidValue;
    } else
    if (tlas) { // the two-level walk is not pipelined
        if (count) return launch_node<MODE, true, false, true>(p, node, grid, stream);
        return launch_node<MODE, true, false, false>(p, node, grid, stream);
    }
    // (coherent primary rays do not gain from the pipelined walk)
    if constexpr (MODE != kModePrimary) {
        if (pipe) {
            if (count) return launch_node<MODE, false, true, true>(p, node, grid, stream);
            return launch_node<MODE, false, true, false>(p, node, grid, stream);
        }
    }
    if (count) return launch_node<MODE, false, false, true>(p, node, grid, stream);
    return launch_node<MODE, false, false, false>(p, node, grid, stream);
}

template <int MODE, bool TLAS, int NODE, bool PIPE, bool COUNT>
int occupancy_one() {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_trace<MODE, TLAS, NODE, PIPE, COUNT>, kWave,
                                                     kLdsBytesPerWave) != hipSuccess)
        return 0;
    return blocks;
}

int node_variant(uint32_t sem) {
    return ((sem & TRX_SEM_NODE_RCP) ? 1 : 0) | ((sem & TRX_SEM_NODE_FMA) ? 2 : 0);
}

} // namespace

int trace_grid_size(int device, int mode, bool tlas, uint32_t sem, bool count) {
    (void)mode;
    (void)sem;
    (void)count;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    // Every product variant is held to <= 128 VGPRs without scratch (tests/test_kernel_resources.py) and owns the same
    // kLdsBytesPerWave of LDS, so all of them fit four waves to a SIMD (16 to a CU: 16 x 10 176 B of the 160 KB): the
    // grid sized from one variant per TLAS flavour is resident for every variant, whatever the workgroup shape
    int per_cu = tlas ? occupancy_one<kModePrimary, true, 1, false, false>() : occupancy_one<kModePrimary, false, 1, false, false>();
    if (per_cu <= 0) per_cu = 8;
    if (per_cu > 32) per_cu = 32;
    return per_cu * prop.multiProcessorCount;
}

hipError_t launch_trace(const TraceParams &p, int mode, bool tlas, uint32_t sem, bool count, bool pipe, int grid,
                        hipStream_t stream) {
    const int node = node_variant(sem);
    switch (mode) {
    case kModePrimary: return launch_mode<kModePrimary>(p, tlas, node, count, pipe, grid, stream);
    case kModeAo: return launch_mode<kModeAo>(p, tlas, node, count, pipe, grid, stream);
    case kModeRays: return launch_mode<kModeRays>(p, tlas, node, count, pipe, grid, stream);
    case kModeFused: return launch_mode<kModeFused>(p, tlas, node, count, pipe, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace trx
