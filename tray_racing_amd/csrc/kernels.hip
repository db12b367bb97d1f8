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

// Diagnostic builds (never the product): TRX_DEV_TUNE = run-time experiment switches (TraceParams::tune), TRX_TAIL_DIAG =
// what every wave did last and when, TRX_STAMPS = per-wave cycle accounting of the loop's phases (every stamp drains the
// memory counters first, so a phase owns the latency of what it issued).  They are compile-time flags tested with
// `if constexpr` / plain `if` on a constant - the product build folds every such branch away (same instructions as
// without them: build/kernels.s is compared) - instead of preprocessor blocks woven through the walk.
#ifdef TRX_DEV_TUNE
#define TRX_K_TUNE true
#else
#define TRX_K_TUNE false
#endif
#ifdef TRX_TAIL_DIAG
#define TRX_K_TAIL true
#else
#define TRX_K_TAIL false
#endif
#ifdef TRX_STAMPS
#define TRX_K_STAMPS true
#else
#define TRX_K_STAMPS false
#endif
#define TRX_STAMP(acc)                                                  \
    do {                                                                \
        if constexpr (kStamps) {                                        \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            acc += now_ - stamp_prev;                                   \
            stamp_prev = now_;                                          \
        }                                                               \
    } while (0)

namespace trx {
namespace {

// __ballot() takes an int: a bool that lives as a lane mask is first turned into 0 / 1 per lane and compared again (two
// vector instructions where the mask was already there); the builtin takes the bool
__device__ __forceinline__ unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

constexpr bool kTune = TRX_K_TUNE, kTailDiag = TRX_K_TAIL, kStamps = TRX_K_STAMPS;

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

// The quantisation frame of a node as the ray sees it (query.hlsl:237-243): a = e / d, b = (p - o) / d per axis, in the
// arithmetic the NODE variant asks for - shared by the three node tests below (per lane, decode-once, children-per-lane).
// A macro, not a function: it declares ax .. bz in the test that uses it (as a function with reference parameters the
// compiler schedules the tests differently; as text the product's instructions are those of the three written-out copies).
//   pow2: the shader divides per node (query.hlsl:237-243).  e is a power of two, and a correctly rounded quotient
// scales exactly with powers of two: RN(2^k / d) = 2^k * RN(1 / d) bit for bit, as long as neither side leaves
// the normal range - which pow2_exact() below guarantees from the ray (every |d| normal and <= 2^20) and the
// scene (every exponent byte 0 or >= 21); overflow to infinity happens to both sides at the same threshold.
// Three of the six IEEE divisions of a node step become multiplications by the ray's 1/d.
//   The other three, b = (p - o) / d, have a numerator that is no power of two.  With y = RN(1 / d) - the ray's r.ix, an IEEE
// division at ray set-up - the correctly rounded quotient takes ONE correction step (Markstein): q0 = RN(a y),
// rem = a - d q0 (exact in one fma), q = RN(q0 + rem y) = RN(a / d), as long as no intermediate leaves the normal range
// (div_by_rcp: three instructions where the compiler's `/` is eleven).  Whether the identity holds depends on the two
// significands only - scaling by powers of two commutes with every step - and tools/ubench/div_exhaustive.hip checks
// all 2^46 pairs of them on the GPU against `/` (profiles/r05_div_exhaustive.log: 0 mismatches in 7.04e13), plus 2^36
// random pairs over the admitted exponents and zero numerators.  Admitted: 2^-30 <= |d| <= 2^20 and a = +0 or
// 2^-60 <= |a| <= 2^60 - every product, remainder and quotient then stays between 2^-127+24 and 2^127; a = +0 comes out
// as the zero of the right sign (worked through in DESIGN.md section 3), a = -0 would not.  Nothing of that is looked
// at per node step: the RAY carries a flag (finish_ray_dir: every origin component 0 or 2^-36 <= |o| <= 2^59) and the
// SCENE one (trx_scene_create in api.cpp, exp_exact = 2: every component of every node's p is +0 or 2^-36 <= |p| <= 2^59), and the
// difference of two such floats is +0 or a multiple of 2^-59 no larger than 2^60.  shortcut = 2: both shortcuts for
// this step (every lane that takes it), 1: the power-of-two one only, 0: the shader's six divisions.
// packet culling of wave-uniform node steps (trace_walk_plain.inc): on / off
#ifndef TRX_PACKET_CULL
#define TRX_PACKET_CULL 1
#endif
#ifndef TRX_PACKET_CULL_LITERAL
#define TRX_PACKET_CULL_LITERAL 1
#endif
#ifndef TRX_PACKET_CULL_TLAS
#define TRX_PACKET_CULL_TLAS 1
#endif
#ifndef TRX_DIV_BY_RCP
#define TRX_DIV_BY_RCP 1
#endif
// RN(a / d) from y = RN(1 / d): see above for when
__device__ __forceinline__ float div_by_rcp(float a, float d, float y) {
    const float q0 = a * y;
    const float rem = __builtin_fmaf(-d, q0, a);
    return __builtin_fmaf(rem, y, q0);
}
#define TRX_NODE_FRAME(r, n0, pow2)                                                                         \
    const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);         \
    const uint32_t e_imask = n0.w;                                                                          \
    const float ex = __uint_as_float((e_imask & 0xffu) << 23);                                              \
    const float ey = __uint_as_float(((e_imask >> 8) & 0xffu) << 23);                                       \
    const float ez = __uint_as_float(((e_imask >> 16) & 0xffu) << 23);                                      \
    float ax, ay, az, bx, by, bz;                                                                           \
    if (NODE & 1) {                                                                                         \
        ax = ex * r.ix;                                                                                     \
        ay = ey * r.iy;                                                                                     \
        az = ez * r.iz;                                                                                     \
        bx = (px - r.ox) * r.ix;                                                                            \
        by = (py - r.oy) * r.iy;                                                                            \
        bz = (pz - r.oz) * r.iz;                                                                            \
    } else if (pow2) {                                                                                      \
        ax = ex * r.ix;                                                                                     \
        ay = ey * r.iy;                                                                                     \
        az = ez * r.iz;                                                                                     \
        if (TRX_DIV_BY_RCP && pow2 > 1) {                                                                   \
            bx = div_by_rcp(px - r.ox, r.dx, r.ix);                                                         \
            by = div_by_rcp(py - r.oy, r.dy, r.iy);                                                         \
            bz = div_by_rcp(pz - r.oz, r.dz, r.iz);                                                         \
        } else {                                                                                            \
            bx = (px - r.ox) / r.dx;                                                                        \
            by = (py - r.oy) / r.dy;                                                                        \
            bz = (pz - r.oz) / r.dz;                                                                        \
        }                                                                                                   \
    } else {                                                                                                \
        ax = ex / r.dx;                                                                                     \
        ay = ey / r.dy;                                                                                     \
        az = ez / r.dz;                                                                                     \
        bx = (px - r.ox) / r.dx;                                                                            \
        by = (py - r.oy) / r.dy;                                                                            \
        bz = (pz - r.oz) / r.dz;                                                                            \
    }

template <int NODE>
__device__ __forceinline__ uint32_t node_intersect(const Ray &r, float max_distance, const uint4 n0,
                                                   const uint4 n1, const uint4 n2, const uint4 n3,
                                                   const uint4 n4, const int pow2) {
    TRX_NODE_FRAME(r, n0, pow2)
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
                                                       const float *dec_pos, const float *dec_neg, const int pow2) {
    TRX_NODE_FRAME(r, n0, pow2)
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

// The decode-once test over the children a PACKET test left (round 5).  In a wave-uniform node step of coherent primary
// rays most of the eight children are missed by every ray of the wave (bistro-class frame: 1.65 children left on average,
// none in a fifth of the steps; profiles/r05_cullhist.log).  The packet test (trace_walk_plain.inc): with the rays' common
// origin and bounds [lo, hi] of their 1/d per axis, each {child, plane} lane of the decode stage also evaluates the
// plane's parameter at the end of the interval that bounds it from below (near planes) or above (far planes) - the
// SAME multiplications and additions as the per-ray test, at operands that bound every ray's: round-to-nearest
// multiplication and addition are monotone in each operand, so the values bound what any ray of the wave computes - and
// a child whose largest lower bound exceeds its smallest upper bound is entered by no ray: it is left out for the whole
// wave.  This function is the per-ray test of the children that remain (`keep`, wave-uniform, bit c = child c): the
// operations of node_intersect_dec on the same operands, child by child in a scalar loop; a child left out would have
// contributed nothing to the mask.  keep = 0xff: all eight (a wave whose rays do not qualify for the packet test).
//   The literal-division variants (NODE bit 0 clear) take part where a step's rays all take both division shortcuts
// (shortcut level 2: a = e RN(1/d) as here, b = RN(c / d) by div_by_rcp).  RN(c / d) is within 3 x 2^-24 (relative) of
// RN(c RN(1/d)), the value the bounds are computed from, so the packet test widens the b interval by 2^-21 of its ends'
// magnitudes (nothing is near the subnormals there: the level-2 flags keep |c| in 2^-60 .. 2^60 or zero, and a zero c gives
// a zero b on both sides); a is the same product in both variants.
struct NodeFrame {
    float ax, ay, az, bx, by, bz;
};
template <int NODE>
__device__ __forceinline__ NodeFrame node_frame(const Ray &r, const uint4 n0, const int pow2) {
    TRX_NODE_FRAME(r, n0, pow2)
    return NodeFrame{ax, ay, az, bx, by, bz};
}
template <int NODE>
__device__ __forceinline__ uint32_t node_intersect_kept(const Ray &r, float max_distance, const NodeFrame f, const uint4 n1,
                                                        const float *dec_pos, const float *dec_neg, uint32_t keep) {
    const float ax = f.ax, ay = f.ay, az = f.az, bx = f.bx, by = f.by, bz = f.bz;
    const float2 *const qx = reinterpret_cast<const float2 *>(r.dx < 0.0f ? dec_neg : dec_pos);
    const float2 *const qy = reinterpret_cast<const float2 *>((r.dy < 0.0f ? dec_neg : dec_pos) + 16);
    const float2 *const qz = reinterpret_cast<const float2 *>((r.dz < 0.0f ? dec_neg : dec_pos) + 32);
    // (the node is the same for every lane: its child_meta bytes are scalars, and so is everything derived from them but
    // the slot of an inner child, which depends on the ray's octant)
    const uint32_t meta_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)n1.z), meta_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)n1.w);
    const uint32_t oct = r.oct_inv4 & 0xffu;
    uint32_t hit_mask = 0;
    // (requesting the planes of the next child before this one's are used - the loop software-pipelined - measured slower,
    // profiles/r05_ab_19_cull_pipelined.log)
    for (uint32_t m = keep; m != 0u; m &= m - 1u) {
        const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)(__builtin_ctz(m))); // (wave-uniform: a scalar)
        const float2 x = qx[j], y = qy[j], z = qz[j]; // {near, far}
        const f32x2 tx = plane2<NODE>(f32x2{x.x, x.y}, ax, bx);
        const f32x2 ty = plane2<NODE>(f32x2{y.x, y.y}, ay, by);
        const f32x2 tz = plane2<NODE>(f32x2{z.x, z.y}, az, bz);
        const float tmin = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), 0.0001f);
        const float tmax = fminf(fminf(fminf(tx.y, ty.y), tz.y), max_distance);
        const uint32_t mj = ((j < 4u ? meta_lo : meta_hi) >> (8u * (j & 3u))) & 0xffu;
        const uint32_t child_bits = mj >> 5;
        // child_bits << bit_index with bit_index = (meta ^ (oct & inner_mask)) & 0x1f, as in node_intersect
        uint32_t word;
        if ((mj & 0x18u) == 0x18u) word = child_bits << ((mj ^ oct) & 0x1fu);
        else word = child_bits << (mj & 0x1fu);
        if (tmin <= tmax) hit_mask |= word;
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
                                                            const uint32_t q[6], const int pow2) {
    TRX_NODE_FRAME(r, n0, pow2)
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
        // (tie_first ? tt < t : tt <= t, written so that the wave-uniform flag is a scalar mask operation: as a select
        // between the two comparisons it compiles to five vector instructions per test)
        const bool lt = tt < t, eq = tt == t;
        const bool closer = lt | (eq & !tie_first); // (bitwise: no short-circuit branches)
        if ((tt >= r.tmin) & closer) {
            t = tt;
            return true;
        }
    }
    return false;
}


// (LITERAL: a node test that divides - the flags are only looked at there)
template <bool LITERAL>
__device__ __forceinline__ void finish_ray_dir(Ray &r, float dx, float dy, float dz) {
    r.dx = dx == 0.0f ? TRX_F32_EPSILON : dx;
    r.dy = dy == 0.0f ? TRX_F32_EPSILON : dy;
    r.dz = dz == 0.0f ? TRX_F32_EPSILON : dz;
    r.ix = 1.0f / r.dx;
    r.iy = 1.0f / r.dy;
    r.iz = 1.0f / r.dz;
    r.oct_inv4 = (r.dx < 0.0f ? 0u : 0x04040404u) | (r.dy < 0.0f ? 0u : 0x02020202u) |
                 (r.dz < 0.0f ? 0u : 0x01010101u);
    if (!LITERAL) return;
    // bits 31 and 30 (masked out wherever oct_inv4 is used) switch the literal-division shortcuts of TRX_NODE_FRAME off:
    // bit 31 - a direction component outside 2^-30 .. 2^20 (or NaN): this ray divides six times per node like the shader's
    // text (e / d = e * (1/d) needs |d| normal and <= 2^20; the lower bound is the tighter one of div_by_rcp);
    // bit 30 - an origin component that is neither 0 nor within 2^-36 .. 2^59: (p - o) / d is divided
    const float lo = 0x1p-30f, hi = 1048576.0f; // a NaN fails both
    const bool exact = fabsf(r.dx) >= lo && fabsf(r.dx) <= hi && fabsf(r.dy) >= lo && fabsf(r.dy) <= hi &&
                       fabsf(r.dz) >= lo && fabsf(r.dz) <= hi;
    if (!exact) r.oct_inv4 |= 0x80000000u;
    const float olo = 0x1p-36f, ohi = 0x1p59f;
    const bool org = (r.ox == 0.0f || (fabsf(r.ox) >= olo && fabsf(r.ox) <= ohi)) && (r.oy == 0.0f || (fabsf(r.oy) >= olo && fabsf(r.oy) <= ohi)) &&
                     (r.oz == 0.0f || (fabsf(r.oz) >= olo && fabsf(r.oz) <= ohi));
    if (!org) r.oct_inv4 |= 0x40000000u;
}

// Wave-uniform: which shortcuts may this node step of the literal-division variants take (TRX_NODE_FRAME: 0, 1 or 2)?
__device__ __forceinline__ int pow2_exact(const TraceParams &P, const Ray &r, bool act) {
    if (P.exp_exact == 0u) return 0;
    if (__ballot(act && (r.oct_inv4 >> 30) != 0u) == 0ull) return (int)P.exp_exact;
    return __ballot(act && (r.oct_inv4 >> 31) != 0u) == 0ull ? 1 : 0;
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
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
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

// 16-byte and 8-byte accesses that go all the way to memory (`sc0 sc1`: system scope): the ray service's porter reads request
// granules out of pinned host memory and writes answers there while the host looks on (trace_service.inc)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 sys_load128(const void *p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
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
#include "trace_state.inc"
#include "trace_queues.inc"
#include "trace_stack.inc"
#include "trace_thin.inc"
    if constexpr (MODE == kModeService) {
#include "trace_service.inc"
    } else {
    for (;;) {
        TRX_STAMP(k_pop);
#include "trace_refill.inc"
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

#include "trace_triangles.inc"
#include "trace_drain.inc"
        if (kMerge && need_take != 0u) {
            merge_take(need_take);
            need_take = 0u;
        }
        if constexpr (!PIPE) {
#include "trace_walk_plain.inc"
        } else {
#include "trace_walk_pipe.inc"
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

    }

#include "trace_epilogue.inc"
}


#undef wave_global
#undef spill
#undef wray

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
    if constexpr (MODE == kModeService) {
        // (the resident kernel is the plain thin walk, single- or two-level: no pipelining, no counting)
        if (count) return hipErrorInvalidValue;
        return tlas ? launch_node<MODE, true, false, false>(p, node, grid, stream) : launch_node<MODE, false, false, false>(p, node, grid, stream);
    } else {
        // (the one-launch frame exists for single-level scenes: the two-level walk has no registers to spare for the in-place
        // hand-over - it spills - so trx_trace_frame_dev runs a two-level frame as two launches, api_trace.cpp)
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
    // kLdsBytesPerWave of LDS, so all of them fit four waves to a SIMD (16 to a CU: 16 x 10 240 B = the 160 KB exactly): the
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
    case kModeService: return launch_mode<kModeService>(p, tlas, node, count, pipe, grid, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace trx
