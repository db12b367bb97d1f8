/*
 * trx_oracle.c — CPU restatement of tray_racing's CWBVH closest-hit path.
 *
 * TEST INFRASTRUCTURE ONLY (see trx_oracle.h).  PARITY UNPINNED: no reference
 * golden vectors exist and the reference cannot be executed here; the
 * restatement is validated against the brute-force query at the bottom of
 * this file.
 *
 * Normative text, in priority order (paths relative to the tray_racing checkout):
 *   src/rt_gpu/rt_gpu_software_query.hlsl       node test :213-303, traversal :328-438,
 *                                               triangle test :89-129, unpack :75-85, octant :314-326
 *   src/rt_gpu/rt_gpu_software_query_tlas.hlsl  two-level traversal :333-500
 *   src/rt_cpu/rt_cpu.rs                        ray-gen / AO / shading order :38-92
 *   src/rt_gpu/sampling.hlsl                    hash + sampling :5-51
 *   src/main.rs                                 camera matrices :602-616
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md "Numerics"):
 * IEEE-754 binary32 everywhere, round-to-nearest-even, no contraction (built
 * with -ffp-contract=off; the one fused operation is the explicit fmaf under
 * ORC_SEM_NODE_FMA), correctly rounded / and sqrt, dot(a,b) = (ax*bx + ay*by) + az*bz,
 * cross(a,b) = (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx).
 */
#include "trx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#define ORC_HAVE_AVX2 1
#endif

#define F32_MAX 3.402823466e+38f
#define F32_EPSILON 1.1920929e-7f
#define INVALID 0xFFFFFFFFu

static inline uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float u2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline void cross3(const float a[3], const float b[3], float r[3]) {
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void normalize3(float v[3]) {
    float len = sqrtf(dot3(v, v));
    float inv = 1.0f / len;
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}
static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}
static int pick_threads(int threads) {
#ifdef _OPENMP
    return threads > 0 ? threads : omp_get_num_procs();
#else
    (void)threads;
    return 1;
#endif
}

/* ---- triangle formats ------------------------------------------------------- */

/* obvhs RtTriangle::from(&Triangle) as used at src/rt_cpu/mod.rs:38-43:
 * e1 = v0 - v1, e2 = v2 - v0 (the sign the HLSL applies at query.hlsl:91-92). */
void orc_tris_from_verts(const float *verts, uint64_t n, float *out9) {
    for (uint64_t i = 0; i < n; i++) {
        const float *v = verts + 9 * i;
        float *o = out9 + 9 * i;
        for (int k = 0; k < 3; k++) {
            o[k] = v[k];
            o[3 + k] = v[k] - v[3 + k];
            o[6 + k] = v[6 + k] - v[k];
        }
    }
}

static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u2f(sign);
        float f = (float)man * 5.9604644775390625e-8f; /* man * 2^-24, exact */
        return sign ? -f : f;
    }
    if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
    return u2f(sign | ((exp + 112) << 23) | (man << 13));
}

/* unpack_triangle, src/rt_gpu/rt_gpu_software_query.hlsl:75-85: low half of
 * e[k] is e2[k] = (v2-v0)[k], high half is e1[k] = (v1-v0)[k]; intersect negates e1. */
void orc_tris_from_f16(const void *tri24, uint64_t n, float *out9) {
    const uint8_t *b = (const uint8_t *)tri24;
    for (uint64_t i = 0; i < n; i++) {
        float v[3];
        uint32_t e[3];
        memcpy(v, b + 24 * i, 12);
        memcpy(e, b + 24 * i + 12, 12);
        float *o = out9 + 9 * i;
        for (int k = 0; k < 3; k++) {
            o[k] = v[k];
            o[3 + k] = -half_to_float((uint16_t)(e[k] >> 16));
            o[6 + k] = half_to_float((uint16_t)(e[k] & 0xffff));
        }
    }
}

/* ---- camera (src/main.rs:602-616, glam column-major) --------------------------- */

static void mat4_inverse(const float m[16], float out[16]) {
    /* cofactor expansion in double, rounded once */
    double a[16], inv[16];
    for (int i = 0; i < 16; i++) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    double r = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = (float)(inv[i] * r);
}

void orc_view_from_camera(const float eye[3], const float look_at[3], float fov_deg, float width,
                          float height, orc_view *out) {
    memset(out, 0, sizeof(*out));
    float aspect = width / height;
    float fov = fov_deg * (3.14159265358979323846f / 180.0f);
    /* perspective_infinite_reverse_rh(fov, aspect, 0.01) */
    float f = 1.0f / tanf(0.5f * fov);
    float proj[16] = {0};
    proj[0] = f / aspect;
    proj[5] = f;
    proj[11] = -1.0f; /* col 2 = (0,0,0,-1) */
    proj[14] = 0.01f; /* col 3 = (0,0,z_near,0) */
    mat4_inverse(proj, out->proj_inv);
    /* look_at_rh(eye, look_at, +Y) */
    float fw[3] = {look_at[0] - eye[0], look_at[1] - eye[1], look_at[2] - eye[2]};
    normalize3(fw);
    float up[3] = {0, 1, 0}, s[3], u[3];
    cross3(fw, up, s);
    normalize3(s);
    cross3(s, fw, u);
    float view[16] = {s[0], u[0], -fw[0], 0, s[1], u[1], -fw[1], 0, s[2], u[2], -fw[2], 0,
                      -dot3(s, eye), -dot3(u, eye), dot3(fw, eye), 1};
    mat4_inverse(view, out->view_inv);
    out->eye[0] = eye[0];
    out->eye[1] = eye[1];
    out->eye[2] = eye[2];
}

/* ---- ray generation: src/rt_gpu/rt_gpu_software.hlsl:69-80, src/rt_cpu/rt_cpu.rs:38-55 ---- */

static inline void mat4_mul_vec4(const float m[16], const float v[4], float r[4]) {
    for (int i = 0; i < 4; i++) r[i] = ((m[i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2]) + m[12 + i] * v[3];
}

void orc_primary_ray(const orc_view *view, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                     float o[3], float d[3]) {
    float u = (float)px / (float)w;
    float v = (float)py / (float)h;
    v = 1.0f - v;
    float clip[4] = {u * 2.0f - 1.0f, v * 2.0f - 1.0f, 1.0f, 1.0f};
    float vs[4], wp[4];
    mat4_mul_vec4(view->proj_inv, clip, vs);
    float vw = vs[3];
    vs[0] = vs[0] / vw;
    vs[1] = vs[1] / vw;
    vs[2] = vs[2] / vw;
    vs[3] = vs[3] / vw;
    mat4_mul_vec4(view->view_inv, vs, wp);
    d[0] = wp[0] - view->eye[0];
    d[1] = wp[1] - view->eye[1];
    d[2] = wp[2] - view->eye[2];
    normalize3(d);
    o[0] = view->eye[0];
    o[1] = view->eye[1];
    o[2] = view->eye[2];
}

/* ---- sampling: src/rt_gpu/sampling.hlsl:5-51 ------------------------------------- */

uint32_t orc_uhash(uint32_t a, uint32_t b) {
    uint32_t x = (a * 1597334673u) ^ (b * 3812015801u);
    x = x ^ (x >> 16);
    x *= 0x7feb352du;
    x = x ^ (x >> 15);
    x *= 0x846ca68bu;
    x = x ^ (x >> 16);
    return x;
}

float orc_hash_noise(uint32_t x, uint32_t y, uint32_t frame) {
    uint32_t urnd = orc_uhash(x, (y << 11) + frame);
    return (float)urnd * (1.0f / 4294967296.0f); /* 1/float(0xffffffff): float(0xffffffff) == 2^32 */
}

/* sin/cos of theta in [0, 2*pi].  The reference calls its platform's sin / cos (sampling.hlsl:33-34; obvhs
 * test_util::sampling on the CPU path, src/rt_cpu/rt_cpu.rs:5-8,69-73, where Rust's f32::sin / cos are the C library's
 * sinf / cosf).  A platform call is not reproducible on a GPU, so this oracle and the kernels both evaluate the SAME
 * explicit binary64 algorithm and round once to binary32 - and the algorithm is the one the GNU C library (2.28 and
 * later; also bionic and the Arm Optimized Routines it comes from) publishes for sinf / cosf: quadrant n =
 * round(x * 2/pi) through a scaled integer conversion, r = x - n * pi/2 in binary64, then a degree-7 odd / degree-8 even
 * polynomial in binary64.  Measured against this container's glibc 2.35: bit-identical sinf and cosf for EVERY binary32
 * in [0, 2 pi] (orc_sincos_libm_mismatches, tests/test_oracle.py, profiles/r04_sincos_vs_libm.log), so on a Linux host
 * the AO directions are the reference CPU path's own.  (Until round 3 both sides used a 2-3 ulp binary32 polynomial;
 * against libm 5-7 % of the AO rays then differed in their last bits of t.  A correctly rounded sin / cos - what a
 * CORE-MATH based C library returns - differs from this algorithm by one ulp for 1.3 % of the arguments: the exposure
 * tool reports that column too.) */
static int g_ao_libm = 0;
/* Exposure measurement only (tests/analysis/semantics_exposure.py): AO directions from this platform's libm sinf / cosf (1), the
 * way the reference calls its platform's, or from a correctly rounded sin / cos (2: binary64 libm rounded once to
 * binary32), instead of the explicit evaluation below.  The product has no such switch. */
void orc_set_ao_libm(int on) { g_ao_libm = on; }

static const double SC_HPI_INV = 0x1.45F306DC9C883p+23; /* 2/pi * 2^24 */
static const double SC_HPI = 0x1.921FB54442D18p0;       /* pi/2 */
static const double SC_C0 = 0x1p0, SC_C1 = -0x1.ffffffd0c621cp-2, SC_C2 = 0x1.55553e1068f19p-5,
                    SC_C3 = -0x1.6c087e89a359dp-10, SC_C4 = 0x1.99343027bf8c3p-16;
static const double SC_S1 = -0x1.555545995a603p-3, SC_S2 = 0x1.1107605230bc4p-7, SC_S3 = -0x1.994eb3774cf24p-13;

void orc_sincos(float theta, float *s, float *c) {
    if (g_ao_libm == 1) {
        *s = sinf(theta);
        *c = cosf(theta);
        return;
    }
    if (g_ao_libm == 2) {
        *s = (float)sin((double)theta);
        *c = (float)cos((double)theta);
        return;
    }
    const uint32_t top = (f2u(theta) >> 20) & 0x7ffu; /* sign cleared, exponent and three mantissa bits */
    double x = (double)theta;
    int n = 0;
    if (top >= 0x3f4u) { /* |theta| >= 0.75: reduce by multiples of pi/2 (theta <= 2 pi here, far below the 120 the method allows) */
        const double q = x * SC_HPI_INV;
        n = ((int32_t)q + 0x800000) >> 24;
        x = x - (double)n * SC_HPI;
    }
    const double x2 = x * x;
    /* sine polynomial */
    const double x3 = x * x2;
    const double s1 = SC_S2 + x2 * SC_S3;
    const double x7 = x3 * x2;
    const double sa = x + x3 * SC_S1;
    float sp = (float)(sa + x7 * s1);
    /* cosine polynomial */
    const double x4 = x2 * x2;
    const double c2 = SC_C3 + x2 * SC_C4;
    const double c1 = SC_C0 + x2 * SC_C1;
    const double x6 = x4 * x2;
    const double ca = c1 + x4 * SC_C2;
    float cp = (float)(ca + x6 * c2);
    if (top < 0x398u) { /* |theta| < 2^-12 */
        sp = theta;
        cp = 1.0f;
    }
    switch (n & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}

/* Every binary32 whose bits lie in [lo_bits, hi_bits], `stride` apart: how many have orc_sincos's sine or cosine differ
 * from this platform's sinf / cosf (bit compare).  0 over [0, 2 pi] on glibc >= 2.28. */
uint64_t orc_sincos_libm_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride) {
    uint64_t bad = 0;
    if (stride == 0) stride = 1;
    const int64_t n = ((int64_t)hi_bits - (int64_t)lo_bits) / stride + 1;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (int64_t i = 0; i < n; i++) {
        const float th = u2f(lo_bits + (uint32_t)i * stride);
        float s, c;
        orc_sincos(th, &s, &c);
        bad += (f2u(s) != f2u(sinf(th))) || (f2u(c) != f2u(cosf(th)));
    }
    return bad;
}

/* AO ray of src/rt_cpu/rt_cpu.rs:61-76 / src/rt_gpu/rt_gpu_software.hlsl:105-121.
 * Returns 0 when the primary ray missed. */
void orc_xform_point(const float m[12], const float p[3], float out[3]) {
    for (int r = 0; r < 3; r++) out[r] = ((m[4 * r] * p[0] + m[4 * r + 1] * p[1]) + m[4 * r + 2] * p[2]) + m[4 * r + 3];
}
void orc_xform_dir(const float m[12], const float v[3], float out[3]) {
    for (int r = 0; r < 3; r++) out[r] = (m[4 * r] * v[0] + m[4 * r + 1] * v[1]) + m[4 * r + 2] * v[2];
}

int orc_ao_ray(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t px,
               uint32_t py, orc_hit primary, uint32_t frame, float ao_eps, float o[3], float d[3]) {
    return orc_ao_ray_inst(s, view, w, h, px, py, primary, INVALID, frame, ao_eps, o, d);
}

int orc_ao_ray_inst(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                    orc_hit primary, uint32_t primary_inst, uint32_t frame, float ao_eps, float o[3], float d[3]) {
    if (!(primary.t < F32_MAX) || primary.prim == INVALID) return 0;
    float ro[3], rd[3];
    orc_primary_ray(view, w, h, px, py, ro, rd);
    const float *tri = s->tris + 9 * (uint64_t)primary.prim;
    float n[3];
    cross3(tri + 3, tri + 6, n); /* ng = e1 x e2, compute_normal() = ng.normalize() */
    if (s->instance_w2o && primary_inst != INVALID) {
        /* object-space normal -> world: transpose of world-to-object (= inverse transpose of object-to-world) */
        const float *m = s->instance_w2o + 12 * (uint64_t)primary_inst;
        float nw[3];
        for (int c = 0; c < 3; c++) nw[c] = (m[c] * n[0] + m[4 + c] * n[1]) + m[8 + c] * n[2];
        n[0] = nw[0];
        n[1] = nw[1];
        n[2] = nw[2];
    }
    normalize3(n);
    float nd = (n[0] * -rd[0] + n[1] * -rd[1]) + n[2] * -rd[2];
    float sg = copysignf(1.0f, nd); /* f32::signum: rt_cpu.rs:65 */
    n[0] *= sg;
    n[1] *= sg;
    n[2] *= sg;
    for (int k = 0; k < 3; k++) o[k] = (view->eye[k] + rd[k] * primary.t) - rd[k] * ao_eps;
    float u1 = orc_hash_noise(px, py, frame);
    float u2 = orc_hash_noise(px, py, frame + 1024u);
    /* cosine_sample_hemisphere, sampling.hlsl:28-37 */
    float r = sqrtf(u1);
    float theta = u2 * 6.28318530717958647692f;
    float sn, cs;
    orc_sincos(theta, &sn, &cs);
    float lx = r * cs, ly = r * sn, lz = sqrtf(fmaxf(0.0f, 1.0f - u1));
    /* build_orthonormal_basis, sampling.hlsl:40-51 */
    float sign = n[2] >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign + n[2]);
    float b = n[0] * n[1] * a;
    float b1[3] = {1.0f + sign * n[0] * n[0] * a, sign * b, -sign * n[0]};
    float b2[3] = {b, sign + n[1] * n[1] * a, -n[1]};
    for (int k = 0; k < 3; k++) d[k] = (b1[k] * lx + b2[k] * ly) + n[k] * lz;
    normalize3(d);
    return 1;
}

/* ---- node test: src/rt_gpu/rt_gpu_software_query.hlsl:213-303 ---------------------- */

uint32_t orc_octant_inv4(const float d[3]) { /* :314-326 */
    return (d[0] < 0.0f ? 0u : 0x04040404u) | (d[1] < 0.0f ? 0u : 0x02020202u) | (d[2] < 0.0f ? 0u : 0x01010101u);
}

static inline uint32_t extract_byte(uint32_t x, uint32_t b) { return (x >> (b * 8)) & 0xffu; }

static inline float plane(float q, float adj_inv, float adj_org, uint32_t sem) {
    if (sem & ORC_SEM_NODE_FMA) return fmaf(q, adj_inv, adj_org);
    return q * adj_inv + adj_org;
}

/* Second implementation of the node test: the eight children at once in AVX2 registers (obvhs' CPU node test
 * is SIMD too; this is the form the cpu_baseline leg of bench.py times).  Same IEEE operations per child in the
 * same order - cvt, mul, add (or one fma under ORC_SEM_NODE_FMA), max, max, max, min, min, min, compare - so the
 * mask is the scalar one bit for bit; the one place vector max/min differ from fmaxf/fminf is a NaN operand
 * (0 x inf planes of rays with a denormal direction component), and a node that produces one is handed to the
 * scalar code.  Selected at run time by orc_set_simd(1) when the CPU has AVX2 (and FMA for the fused variant). */
static int g_simd = 0;
static uint32_t node_intersect_scalar(const float o[3], const float d[3], const float inv_d[3], uint32_t oct_inv4,
                                      float max_distance, const uint32_t node[20], uint32_t sem);
#ifdef ORC_HAVE_AVX2
__attribute__((target("avx2,fma"))) static inline uint32_t node_intersect_avx2(const float o[3], const float d[3],
                                                                        const float inv_d[3], uint32_t oct_inv4,
                                                                        float max_distance, const uint32_t node[20],
                                                                        uint32_t sem) {
    const float p[3] = {u2f(node[0]), u2f(node[1]), u2f(node[2])};
    const uint32_t e_imask = node[3];
    const float e3[3] = {u2f(extract_byte(e_imask, 0) << 23), u2f(extract_byte(e_imask, 1) << 23),
                         u2f(extract_byte(e_imask, 2) << 23)};
    float adj_inv[3], adj_org[3];
    for (int k = 0; k < 3; k++) {
        if (sem & ORC_SEM_NODE_RCP) {
            adj_inv[k] = e3[k] * inv_d[k];
            adj_org[k] = (p[k] - o[k]) * inv_d[k];
        } else {
            adj_inv[k] = e3[k] / d[k];
            adj_org[k] = (p[k] - o[k]) / d[k];
        }
    }
    __m256 lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        /* data[2 + k] = {min[0..4], min[4..8], max[0..4], max[4..8]}: 8 bytes each, child j in byte j */
        const __m128i qmin = _mm_loadl_epi64((const __m128i *)(node + 8 + 4 * k));
        const __m128i qmax = _mm_loadl_epi64((const __m128i *)(node + 10 + 4 * k));
        const __m256 fmin = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(qmin));
        const __m256 fmax = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(qmax));
        const __m256 a = _mm256_set1_ps(adj_inv[k]), b = _mm256_set1_ps(adj_org[k]);
        __m256 tlo, thi;
        if (sem & ORC_SEM_NODE_FMA) {
            tlo = _mm256_fmadd_ps(fmin, a, b);
            thi = _mm256_fmadd_ps(fmax, a, b);
        } else {
            tlo = _mm256_add_ps(_mm256_mul_ps(fmin, a), b);
            thi = _mm256_add_ps(_mm256_mul_ps(fmax, a), b);
        }
        /* near / far plane by the sign of the direction (query.hlsl:266-273) */
        if (d[k] < 0.0f) {
            lo[k] = thi;
            hi[k] = tlo;
        } else {
            lo[k] = tlo;
            hi[k] = thi;
        }
    }
    /* any NaN among the 48 planes: let the scalar code (fmaxf / fminf semantics) decide this node */
    const __m256 sum = _mm256_add_ps(_mm256_add_ps(_mm256_add_ps(lo[0], lo[1]), _mm256_add_ps(lo[2], hi[0])),
                                     _mm256_add_ps(hi[1], hi[2]));
    if (_mm256_movemask_ps(_mm256_cmp_ps(sum, sum, _CMP_UNORD_Q)))
        return node_intersect_scalar(o, d, inv_d, oct_inv4, max_distance, node, sem);
    const __m256 tmin = _mm256_max_ps(_mm256_max_ps(_mm256_max_ps(lo[0], lo[1]), lo[2]), _mm256_set1_ps(0.0001f));
    const __m256 tmax = _mm256_min_ps(_mm256_min_ps(_mm256_min_ps(hi[0], hi[1]), hi[2]), _mm256_set1_ps(max_distance));
    const __m256i hit = _mm256_castps_si256(_mm256_cmp_ps(tmin, tmax, _CMP_LE_OQ));
    /* child_meta -> bit positions (query.hlsl:249-254,294-297), one child per 32-bit lane */
    const __m256i meta = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(node + 6)));
    const __m256i is_inner = _mm256_cmpeq_epi32(_mm256_and_si256(meta, _mm256_set1_epi32(0x18)), _mm256_set1_epi32(0x18));
    const __m256i oct = _mm256_and_si256(is_inner, _mm256_set1_epi32((int)(oct_inv4 & 0xffu)));
    const __m256i bit_index = _mm256_and_si256(_mm256_xor_si256(meta, oct), _mm256_set1_epi32(0x1f));
    const __m256i child_bits = _mm256_and_si256(_mm256_srli_epi32(meta, 5), _mm256_set1_epi32(7));
    __m256i bits = _mm256_and_si256(_mm256_sllv_epi32(child_bits, bit_index), hit);
    bits = _mm256_or_si256(bits, _mm256_permute2x128_si256(bits, bits, 1));
    bits = _mm256_or_si256(bits, _mm256_shuffle_epi32(bits, 0x4e));
    bits = _mm256_or_si256(bits, _mm256_shuffle_epi32(bits, 0xb1));
    return (uint32_t)_mm256_cvtsi256_si32(bits);
}
#endif

void orc_set_simd(int on) {
#ifdef ORC_HAVE_AVX2
    g_simd = on && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
#else
    (void)on;
    g_simd = 0;
#endif
}
int orc_get_simd(void) { return g_simd; }

uint32_t orc_node_intersect(const float o[3], const float d[3], const float inv_d[3], uint32_t oct_inv4,
                            float max_distance, const uint32_t node[20], uint32_t sem) {
#ifdef ORC_HAVE_AVX2
    if (g_simd) return node_intersect_avx2(o, d, inv_d, oct_inv4, max_distance, node, sem);
#endif
    return node_intersect_scalar(o, d, inv_d, oct_inv4, max_distance, node, sem);
}

static uint32_t node_intersect_scalar(const float o[3], const float d[3], const float inv_d[3], uint32_t oct_inv4,
                                      float max_distance, const uint32_t node[20], uint32_t sem) {
    const float p[3] = {u2f(node[0]), u2f(node[1]), u2f(node[2])};
    const uint32_t e_imask = node[3];
    float ex = u2f(extract_byte(e_imask, 0) << 23);
    float ey = u2f(extract_byte(e_imask, 1) << 23);
    float ez = u2f(extract_byte(e_imask, 2) << 23);
    float adj_inv[3], adj_org[3];
    if (sem & ORC_SEM_NODE_RCP) {
        adj_inv[0] = ex * inv_d[0];
        adj_inv[1] = ey * inv_d[1];
        adj_inv[2] = ez * inv_d[2];
        for (int k = 0; k < 3; k++) adj_org[k] = (p[k] - o[k]) * inv_d[k];
    } else {
        adj_inv[0] = ex / d[0];
        adj_inv[1] = ey / d[1];
        adj_inv[2] = ez / d[2];
        for (int k = 0; k < 3; k++) adj_org[k] = (p[k] - o[k]) / d[k];
    }
    uint32_t hit_mask = 0;
    for (int i = 0; i < 2; i++) {
        const uint32_t meta4 = node[6 + i]; /* data[1].z / .w */
        const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
        const uint32_t bit_index4 = (meta4 ^ (oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
        const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
        const uint32_t q_lo_x = node[8 + i], q_hi_x = node[10 + i];   /* data[2].xy / .zw */
        const uint32_t q_lo_y = node[12 + i], q_hi_y = node[14 + i];  /* data[3] */
        const uint32_t q_lo_z = node[16 + i], q_hi_z = node[18 + i];  /* data[4] */
        const uint32_t x_min = d[0] < 0.0f ? q_hi_x : q_lo_x, x_max = d[0] < 0.0f ? q_lo_x : q_hi_x;
        const uint32_t y_min = d[1] < 0.0f ? q_hi_y : q_lo_y, y_max = d[1] < 0.0f ? q_lo_y : q_hi_y;
        const uint32_t z_min = d[2] < 0.0f ? q_hi_z : q_lo_z, z_max = d[2] < 0.0f ? q_lo_z : q_hi_z;
        const float EPSILON = 0.0001f;
        for (uint32_t j = 0; j < 4; j++) {
            float tminx = plane((float)extract_byte(x_min, j), adj_inv[0], adj_org[0], sem);
            float tminy = plane((float)extract_byte(y_min, j), adj_inv[1], adj_org[1], sem);
            float tminz = plane((float)extract_byte(z_min, j), adj_inv[2], adj_org[2], sem);
            float tmaxx = plane((float)extract_byte(x_max, j), adj_inv[0], adj_org[0], sem);
            float tmaxy = plane((float)extract_byte(y_max, j), adj_inv[1], adj_org[1], sem);
            float tmaxz = plane((float)extract_byte(z_max, j), adj_inv[2], adj_org[2], sem);
            float tmin = fmaxf(fmaxf(fmaxf(tminx, tminy), tminz), EPSILON);
            float tmax = fminf(fminf(fminf(tmaxx, tmaxy), tmaxz), max_distance);
            if (tmin <= tmax) {
                uint32_t child_bits = extract_byte(child_bits4, j);
                uint32_t bit_index = extract_byte(bit_index4, j);
                hit_mask |= child_bits << bit_index;
            }
        }
    }
    return hit_mask;
}

/* ---- triangle test: src/rt_gpu/rt_gpu_software_query.hlsl:89-129 ---------------------
 * tri9 = {v0, e1 = v0 - v1, e2 = v2 - v0}.  tmin generalises `tt >= 0.0` to
 * Ray::new(.., tmin, tmax) (src/rt_cpu/rt_cpu.rs:50-55 passes 0.0). */
int orc_intersect_tri(const float o[3], const float d[3], const float tri9[9], float tmin, float *t,
                      uint32_t sem) {
    const float *v0 = tri9, *e1 = tri9 + 3, *e2 = tri9 + 6;
    float ng[3], c[3], r[3];
    cross3(e1, e2, ng);
    c[0] = v0[0] - o[0];
    c[1] = v0[1] - o[1];
    c[2] = v0[2] - o[2];
    cross3(d, c, r);
    float inv_det = 1.0f / dot3(ng, d);
    float u = dot3(r, e2) * inv_det;
    float v = dot3(r, e1) * inv_det;
    float w = 1.0f - u - v;
    uint32_t hit = f2u(u) | f2u(v) | f2u(w);
    if (inv_det != 0.0f && (hit & 0x80000000u) == 0) {
        float tt = dot3(ng, c) * inv_det;
        int closer = (sem & ORC_SEM_TIE_FIRST) ? (tt < *t) : (tt <= *t);
        if (tt >= tmin && closer) {
            *t = tt;
            return 1;
        }
    }
    return 0;
}

/* ---- traversal: query.hlsl:328-438 (BLAS only), query_tlas.hlsl:333-500 (TLAS+BLAS) ---- */

typedef struct {
    uint32_t x, y;
} u2;

static inline uint32_t firstbithigh(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }

orc_hit orc_traverse(const orc_scene *s, const float o[3], const float d_in[3], float tmin, float tmax,
                     uint32_t sem, orc_stats *st) {
    return orc_traverse_inst(s, o, d_in, tmin, tmax, sem, st, NULL);
}

/* One body, compiled twice: with the scalar node test, and - inside a function built for AVX2 - with the 8-wide one
 * inlined into the loop (orc_set_simd).  Everything but the node test is the same scalar code in both. */
static inline __attribute__((always_inline)) orc_hit traverse_body(const orc_scene *s, const float o_in[3], const float d_in[3],
                                                                   float tmin, float tmax, uint32_t sem, orc_stats *st,
                                                                   uint32_t *inst_out, const int simd) {
    float o[3] = {o_in[0], o_in[1], o_in[2]}, d[3], inv_d[3];
    for (int k = 0; k < 3; k++) { /* :334 zero-direction fix, seen by node AND triangle tests */
        d[k] = d_in[k] == 0.0f ? F32_EPSILON : d_in[k];
        inv_d[k] = 1.0f / d[k];
    }
    uint32_t cur_inst = INVALID, hit_inst = INVALID;
    const int tlas = s->n_instances > 0;
    u2 stack[ORC_STACK_SIZE];
    uint32_t sp = 0, max_sp = 0;
    int overflow = 0;
    uint32_t tlas_stack_size = INVALID;
    uint32_t bvh_offset = tlas ? s->tlas_start : 0;
    uint32_t oct_inv4 = orc_octant_inv4(d);
    u2 cur = {0, 0x80000000u};
    float t = fminf(tmax, F32_MAX);
    uint32_t prim = INVALID;
    uint64_t n_node = 0, n_tri = 0, n_tlas_node = 0, n_inst_enter = 0;

#define PUSH(g)                                   \
    do {                                          \
        if (sp < ORC_STACK_SIZE) stack[sp] = (g); \
        else overflow = 1;                        \
        sp++;                                     \
        if (sp > max_sp) max_sp = sp;             \
    } while (0)

    for (;;) {
        u2 tri;
        if (cur.y & 0xff000000u) {
            uint32_t hits_imask = cur.y;
            uint32_t child_index_offset = firstbithigh(hits_imask);
            uint32_t child_index_base = cur.x;
            cur.y &= ~(1u << child_index_offset);
            if (cur.y & 0xff000000u) PUSH(cur);
            uint32_t slot_index = (child_index_offset - 24) ^ (oct_inv4 & 0xff);
            uint32_t relative_index = (uint32_t)__builtin_popcount(hits_imask & ~(0xffffffffu << slot_index));
            uint32_t child_node_index = child_index_base + relative_index;
            const uint32_t *node = s->nodes + 20 * (uint64_t)(bvh_offset + child_node_index);
            n_node++;
            n_tlas_node += tlas && tlas_stack_size == INVALID;
#ifdef ORC_HAVE_AVX2
            uint32_t hitmask = simd ? node_intersect_avx2(o, d, inv_d, oct_inv4, t, node, sem)
                                    : node_intersect_scalar(o, d, inv_d, oct_inv4, t, node, sem);
#else
            uint32_t hitmask = node_intersect_scalar(o, d, inv_d, oct_inv4, t, node, sem);
            (void)simd;
#endif
            uint32_t imask = extract_byte(node[3], 3);
            cur.x = node[4];
            tri.x = node[5];
            cur.y = (hitmask & 0xff000000u) | imask;
            tri.y = hitmask & 0x00ffffffu;
        } else {
            tri = cur;
            cur.x = 0;
            cur.y = 0;
        }
        while (tri.y != 0) {
            uint32_t local = firstbithigh(tri.y);
            tri.y &= ~(1u << local);
            uint32_t global = tri.x + local;
            if (tlas && tlas_stack_size == INVALID) {
                /* a TLAS primitive is an instance: query_tlas.hlsl:410-446 */
                if (tri.y != 0) PUSH(tri);
                if (cur.y & 0xff000000u) PUSH(cur);
                tlas_stack_size = sp;
                n_inst_enter++;
                bvh_offset = s->instance_offsets[global];
                cur_inst = global;
                if (s->instance_w2o) {
                    /* the ray in the instance's object space; the direction is not renormalised, so t keeps its
                     * world-space meaning (jan-van-bergen BVH8.h:222, the TODO at query_tlas.hlsl:433) */
                    const float *m = s->instance_w2o + 12 * (uint64_t)global;
                    float od[3];
                    orc_xform_point(m, o_in, o);
                    orc_xform_dir(m, d_in, od);
                    for (int k = 0; k < 3; k++) {
                        d[k] = od[k] == 0.0f ? F32_EPSILON : od[k];
                        inv_d[k] = 1.0f / d[k];
                    }
                    oct_inv4 = orc_octant_inv4(d);
                }
                cur.x = s->instance_entry ? s->instance_entry[global] : 0; /* a subtree of the BLAS, or its root */
                cur.y = 0x80000000u;
                break;
            }
            n_tri++;
            if (orc_intersect_tri(o, d, s->tris + 9 * (uint64_t)global, tmin, &t, sem)) {
                prim = global;
                hit_inst = cur_inst;
            }
        }
        if ((cur.y & 0xff000000u) == 0) {
            if (sp == 0) break;
            if (tlas && sp == tlas_stack_size) { /* query_tlas.hlsl:480-486 */
                tlas_stack_size = INVALID;
                bvh_offset = s->tlas_start;
                cur_inst = INVALID;
                if (s->instance_w2o) { /* "Reset Ray to untransformed version", query_tlas.hlsl:484 */
                    for (int k = 0; k < 3; k++) {
                        o[k] = o_in[k];
                        d[k] = d_in[k] == 0.0f ? F32_EPSILON : d_in[k];
                        inv_d[k] = 1.0f / d[k];
                    }
                    oct_inv4 = orc_octant_inv4(d);
                }
            }
            sp--;
            if (sp < ORC_STACK_SIZE) cur = stack[sp];
            else { cur.x = 0; cur.y = 0; }
        }
    }
#undef PUSH
    orc_hit h;
    if (prim != INVALID) {
        h.t = t;
        h.prim = prim;
    } else {
        h.t = INFINITY;
        h.prim = INVALID;
    }
    if (inst_out) *inst_out = prim != INVALID ? hit_inst : INVALID;
    if (st) {
        st->n_rays++;
        st->n_node += n_node;
        st->n_tri += n_tri;
        st->n_tlas_node += n_tlas_node;
        st->n_inst_enter += n_inst_enter;
        st->n_hits += prim != INVALID;
        if (max_sp > st->max_stack) st->max_stack = max_sp;
        st->overflow += (uint32_t)overflow;
    }
    return h;
}

static orc_hit traverse_scalar(const orc_scene *s, const float o[3], const float d[3], float tmin, float tmax, uint32_t sem,
                               orc_stats *st, uint32_t *inst_out) {
    return traverse_body(s, o, d, tmin, tmax, sem, st, inst_out, 0);
}
#ifdef ORC_HAVE_AVX2
__attribute__((target("avx2,fma"))) static orc_hit traverse_avx2(const orc_scene *s, const float o[3], const float d[3],
                                                                 float tmin, float tmax, uint32_t sem, orc_stats *st,
                                                                 uint32_t *inst_out) {
    return traverse_body(s, o, d, tmin, tmax, sem, st, inst_out, 1);
}
#endif

orc_hit orc_traverse_inst(const orc_scene *s, const float o[3], const float d[3], float tmin, float tmax, uint32_t sem,
                          orc_stats *st, uint32_t *inst_out) {
#ifdef ORC_HAVE_AVX2
    if (g_simd) return traverse_avx2(s, o, d, tmin, tmax, sem, st, inst_out);
#endif
    return traverse_scalar(s, o, d, tmin, tmax, sem, st, inst_out);
}

/* ---- frames ----------------------------------------------------------------------- */

static void stats_merge(orc_stats *dst, const orc_stats *src) {
    dst->n_rays += src->n_rays;
    dst->n_node += src->n_node;
    dst->n_tri += src->n_tri;
    dst->n_tlas_node += src->n_tlas_node;
    dst->n_inst_enter += src->n_inst_enter;
    dst->n_hits += src->n_hits;
    if (src->max_stack > dst->max_stack) dst->max_stack = src->max_stack;
    dst->overflow += src->overflow;
}

/* tile t (8x8 pixels, row-major tile numbering) belongs to the shard iff t % count == index */
void orc_trace_primary(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h,
                       uint32_t shard_index, uint32_t shard_count, uint32_t sem, int threads,
                       orc_hit *hits, orc_stats *st) {
    orc_trace_primary_inst(s, view, w, h, shard_index, shard_count, sem, threads, hits, NULL, st);
}

void orc_trace_primary_inst(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h,
                            uint32_t shard_index, uint32_t shard_count, uint32_t sem, int threads,
                            orc_hit *hits, uint32_t *inst, orc_stats *st) {
    if (shard_count == 0) shard_count = 1;
    const uint32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
    const int64_t n_tiles = (int64_t)tx * ty;
    threads = pick_threads(threads);
    orc_stats total;
    memset(&total, 0, sizeof(total));
    double t0 = now_s();
#pragma omp parallel num_threads(threads)
    {
        orc_stats loc;
        memset(&loc, 0, sizeof(loc));
#pragma omp for schedule(dynamic, 16)
        for (int64_t tile = 0; tile < n_tiles; tile++) {
            if ((uint32_t)(tile % shard_count) != shard_index) continue;
            uint32_t x0 = (uint32_t)(tile % tx) * 8, y0 = (uint32_t)(tile / tx) * 8;
            for (uint32_t k = 0; k < 64; k++) {
                uint32_t px = x0 + (k & 7), py = y0 + (k >> 3);
                if (px >= w || py >= h) continue;
                float o[3], d[3];
                orc_primary_ray(view, w, h, px, py, o, d);
                uint32_t hi = INVALID;
                hits[(uint64_t)py * w + px] = orc_traverse_inst(s, o, d, 0.0f, F32_MAX, sem, &loc, &hi);
                if (inst) inst[(uint64_t)py * w + px] = hi;
            }
        }
#pragma omp critical
        stats_merge(&total, &loc);
    }
    total.seconds = now_s() - t0;
    total.threads = threads;
    if (st) *st = total;
}

void orc_trace_ao(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t shard_index,
                  uint32_t shard_count, uint32_t sem, uint32_t frame, float ao_eps, int threads,
                  const orc_hit *primary, orc_hit *ao, orc_stats *st) {
    orc_trace_ao_inst(s, view, w, h, shard_index, shard_count, sem, frame, ao_eps, threads, primary, NULL, ao, NULL, st);
}

void orc_trace_ao_inst(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t shard_index,
                       uint32_t shard_count, uint32_t sem, uint32_t frame, float ao_eps, int threads,
                       const orc_hit *primary, const uint32_t *primary_inst, orc_hit *ao, uint32_t *ao_inst,
                       orc_stats *st) {
    if (shard_count == 0) shard_count = 1;
    const uint32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
    const int64_t n_tiles = (int64_t)tx * ty;
    threads = pick_threads(threads);
    orc_stats total;
    memset(&total, 0, sizeof(total));
    double t0 = now_s();
#pragma omp parallel num_threads(threads)
    {
        orc_stats loc;
        memset(&loc, 0, sizeof(loc));
#pragma omp for schedule(dynamic, 16)
        for (int64_t tile = 0; tile < n_tiles; tile++) {
            if ((uint32_t)(tile % shard_count) != shard_index) continue;
            uint32_t x0 = (uint32_t)(tile % tx) * 8, y0 = (uint32_t)(tile / tx) * 8;
            for (uint32_t k = 0; k < 64; k++) {
                uint32_t px = x0 + (k & 7), py = y0 + (k >> 3);
                if (px >= w || py >= h) continue;
                uint64_t i = (uint64_t)py * w + px;
                float o[3], d[3];
                uint32_t hi = INVALID;
                if (orc_ao_ray_inst(s, view, w, h, px, py, primary[i], primary_inst ? primary_inst[i] : INVALID, frame,
                                    ao_eps, o, d)) {
                    ao[i] = orc_traverse_inst(s, o, d, 0.0f, F32_MAX, sem, &loc, &hi);
                } else {
                    ao[i].t = INFINITY;
                    ao[i].prim = INVALID;
                }
                if (ao_inst) ao_inst[i] = hi;
            }
        }
#pragma omp critical
        stats_merge(&total, &loc);
    }
    total.seconds = now_s() - t0;
    total.threads = threads;
    if (st) *st = total;
}

void orc_trace_rays(const orc_scene *s, const orc_ray *rays, uint64_t n, uint32_t sem, int threads,
                    orc_hit *hits, orc_stats *st) {
    orc_trace_rays_inst(s, rays, n, sem, threads, hits, NULL, st);
}

void orc_trace_rays_inst(const orc_scene *s, const orc_ray *rays, uint64_t n, uint32_t sem, int threads,
                         orc_hit *hits, uint32_t *inst, orc_stats *st) {
    threads = pick_threads(threads);
    orc_stats total;
    memset(&total, 0, sizeof(total));
    double t0 = now_s();
#pragma omp parallel num_threads(threads)
    {
        orc_stats loc;
        memset(&loc, 0, sizeof(loc));
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            uint32_t hi = INVALID;
            hits[i] = orc_traverse_inst(s, rays[i].origin, rays[i].direction, rays[i].tmin, rays[i].tmax, sem, &loc, &hi);
            if (inst) inst[i] = hi;
        }
#pragma omp critical
        stats_merge(&total, &loc);
    }
    total.seconds = now_s() - t0;
    total.threads = threads;
    if (st) *st = total;
}

/* The reference's CPU frame, src/rt_cpu/rt_cpu.rs:35-92: per pixel primary ray,
 * and if it hit one AO ray, shade.  rgb may be NULL.  Returns wall seconds. */
double orc_render_frame(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t sem,
                        uint32_t frame, float ao_eps, int threads, float *rgb) {
    threads = pick_threads(threads);
    double t0 = now_s();
    const int64_t n = (int64_t)w * h;
    double sink = 0.0;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads) reduction(+ : sink)
    for (int64_t i = 0; i < n; i++) {
        uint32_t px = (uint32_t)(i % w), py = (uint32_t)(i / w);
        float o[3], d[3];
        orc_primary_ray(view, w, h, px, py, o, d);
        orc_hit hit = orc_traverse(s, o, d, 0.0f, F32_MAX, sem, NULL);
        float col = 1.0f / hit.t;
        if (hit.t < F32_MAX) {
            float ao_o[3], ao_d[3];
            orc_ao_ray(s, view, w, h, px, py, hit, frame, ao_eps, ao_o, ao_d);
            orc_hit ah = orc_traverse(s, ao_o, ao_d, 0.0f, F32_MAX, sem, NULL);
            col = ah.t < F32_MAX ? ah.t / (1.0f + ah.t) : 1.0f;
        }
        if (rgb) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = col;
        sink += col;
    }
    if (sink == 12345.678) fprintf(stderr, " ");
    return now_s() - t0;
}

/* ---- brute force: BVH-independent ground truth ------------------------------------ */

static orc_hit brute_one(const float *tris9, uint64_t n_tris, const float o[3], const float d_in[3],
                         float tmin, float tmax, uint32_t sem) {
    float d[3];
    for (int k = 0; k < 3; k++) d[k] = d_in[k] == 0.0f ? F32_EPSILON : d_in[k];
    float t = fminf(tmax, F32_MAX);
    uint32_t prim = INVALID;
    for (uint64_t i = 0; i < n_tris; i++)
        if (orc_intersect_tri(o, d, tris9 + 9 * i, tmin, &t, sem)) prim = (uint32_t)i;
    orc_hit h;
    h.t = prim != INVALID ? t : INFINITY;
    h.prim = prim;
    return h;
}

void orc_brute_rays(const float *tris9, uint64_t n_tris, const orc_ray *rays, uint64_t n, uint32_t sem,
                    int threads, orc_hit *hits) {
    threads = pick_threads(threads);
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++)
        hits[i] = brute_one(tris9, n_tris, rays[i].origin, rays[i].direction, rays[i].tmin, rays[i].tmax, sem);
}

void orc_brute_primary(const float *tris9, uint64_t n_tris, const orc_view *view, uint32_t w,
                       uint32_t h, uint32_t sem, int threads, orc_hit *hits) {
    threads = pick_threads(threads);
    const int64_t n = (int64_t)w * h;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
    for (int64_t i = 0; i < n; i++) {
        float o[3], d[3];
        orc_primary_ray(view, w, h, (uint32_t)(i % w), (uint32_t)(i / w), o, d);
        hits[i] = brute_one(tris9, n_tris, o, d, 0.0f, F32_MAX, sem);
    }
}

/* ---- structural validation --------------------------------------------------------- */

typedef struct {
    double mn[3], mx[3];
} boxd;

typedef struct {
    const orc_scene *s;
    const float *verts;
    const float *boxes; /* optional: n_tris * 6, the box each primitive entry was built with */
    uint8_t *prim_seen;
    uint8_t *node_seen;
    char *err;
    int err_len;
    int failed;
} vctx;

static void vfail(vctx *c, const char *fmt, uint64_t a, uint64_t b) {
    if (!c->failed && c->err) snprintf(c->err, (size_t)c->err_len, fmt, (unsigned long long)a, (unsigned long long)b);
    c->failed = 1;
}

static boxd box_empty(void) {
    boxd b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    return b;
}
static void box_grow(boxd *a, const boxd *b) {
    for (int k = 0; k < 3; k++) {
        if (b->mn[k] < a->mn[k]) a->mn[k] = b->mn[k];
        if (b->mx[k] > a->mx[k]) a->mx[k] = b->mx[k];
    }
}

static boxd validate_node(vctx *c, uint32_t bvh_offset, uint32_t seg_end, uint32_t idx, int is_tlas, int depth);

static boxd validate_blas_root(vctx *c, uint32_t offset, uint32_t entry) {
    /* segment end: next larger instance offset or tlas_start */
    uint32_t end = c->s->tlas_start;
    for (uint32_t i = 0; i < c->s->n_instances; i++) {
        uint32_t o = c->s->instance_offsets[i];
        if (o > offset && o < end) end = o;
    }
    return validate_node(c, offset, end, entry, 0, 0);
}

static boxd validate_node(vctx *c, uint32_t bvh_offset, uint32_t seg_end, uint32_t idx, int is_tlas, int depth) {
    boxd total = box_empty();
    if (c->failed) return total;
    uint64_t gi = (uint64_t)bvh_offset + idx;
    if (gi >= seg_end || gi >= c->s->n_nodes) {
        vfail(c, "node index %llu outside its BVH segment (end %llu)", gi, seg_end);
        return total;
    }
    if (depth > 512) {
        vfail(c, "node %llu deeper than 512 levels (cycle?)%llu", gi, 0);
        return total;
    }
    if (!is_tlas) {
        if (c->node_seen[gi]) {
            vfail(c, "node %llu referenced twice%llu", gi, 0);
            return total;
        }
        c->node_seen[gi] = 1;
    }
    const uint32_t *n = c->s->nodes + 20 * gi;
    const uint8_t *nb = (const uint8_t *)n;
    double p[3] = {u2f(n[0]), u2f(n[1]), u2f(n[2])};
    double e[3];
    for (int k = 0; k < 3; k++) e[k] = ldexp(1.0, (int)nb[12 + k] - 127);
    uint8_t imask = nb[15];
    const uint8_t *meta = nb + 24;
    const uint8_t *q[6] = {nb + 32, nb + 40, nb + 48, nb + 56, nb + 64, nb + 72}; /* minx maxx miny maxy minz maxz */
    uint32_t inner_rank = 0, tri_total = 0;
    for (int sl = 0; sl < 8; sl++) {
        uint8_t m = meta[sl];
        if (m == 0) {
            if (imask & (1u << sl)) vfail(c, "node %llu slot %llu empty but imask set", gi, (uint64_t)sl);
            continue;
        }
        boxd qb;
        for (int k = 0; k < 3; k++) {
            qb.mn[k] = p[k] + (double)q[2 * k][sl] * e[k];
            qb.mx[k] = p[k] + (double)q[2 * k + 1][sl] * e[k];
        }
        boxd child = box_empty();
        int is_inner = (m & 0x18) == 0x18;
        if (is_inner) {
            if ((m >> 5) != 1 || (uint32_t)(m & 0x1f) != 24u + (uint32_t)sl || !(imask & (1u << sl)))
                vfail(c, "node %llu slot %llu: inconsistent inner meta/imask", gi, (uint64_t)sl);
            child = validate_node(c, bvh_offset, seg_end, n[4] + inner_rank, is_tlas, depth + 1);
            inner_rank++;
        } else {
            if (imask & (1u << sl)) vfail(c, "node %llu slot %llu leaf but imask set", gi, (uint64_t)sl);
            uint32_t bits = m >> 5, off = m & 0x1f;
            uint32_t cnt = bits == 1 ? 1 : bits == 3 ? 2 : bits == 7 ? 3 : 0;
            if (!cnt || off != tri_total) vfail(c, "node %llu slot %llu: bad leaf meta", gi, (uint64_t)sl);
            tri_total += cnt;
            if (tri_total > 24) vfail(c, "node %llu holds %llu > 24 primitives", gi, tri_total);
            for (uint32_t t = 0; t < cnt && !c->failed; t++) {
                uint64_t prim = (uint64_t)n[5] + off + t;
                if (is_tlas) {
                    if (prim >= c->s->n_instances) {
                        vfail(c, "instance %llu out of range (%llu)", prim, c->s->n_instances);
                        break;
                    }
                    boxd b = validate_blas_root(c, c->s->instance_offsets[prim], c->s->instance_entry ? c->s->instance_entry[prim] : 0);
                    box_grow(&child, &b);
                } else {
                    if (prim >= c->s->n_tris) {
                        vfail(c, "primitive %llu out of range (%llu)", prim, c->s->n_tris);
                        break;
                    }
                    if (c->prim_seen[prim]) vfail(c, "primitive %llu referenced twice%llu", prim, 0);
                    c->prim_seen[prim] = 1;
                    if (c->boxes) { /* pre-split builds: this entry stands for the part of its triangle inside its box */
                        const float *b = c->boxes + 6 * prim;
                        for (int k = 0; k < 3; k++) {
                            if (b[k] < child.mn[k]) child.mn[k] = b[k];
                            if (b[3 + k] > child.mx[k]) child.mx[k] = b[3 + k];
                        }
                    } else {
                        const float *v = c->verts + 9 * prim;
                        for (int a = 0; a < 3; a++)
                            for (int k = 0; k < 3; k++) {
                                double x = v[3 * a + k];
                                if (x < child.mn[k]) child.mn[k] = x;
                                if (x > child.mx[k]) child.mx[k] = x;
                            }
                    }
                }
            }
        }
        for (int k = 0; k < 3 && !c->failed; k++)
            if (child.mn[k] < qb.mn[k] || child.mx[k] > qb.mx[k])
                vfail(c, "node %llu slot %llu: quantised box does not contain its subtree", gi, (uint64_t)sl);
        box_grow(&total, &child);
    }
    return total;
}

int orc_validate(const orc_scene *s, const float *verts, const float *boxes, char *err, int err_len) {
    vctx c;
    memset(&c, 0, sizeof(c));
    c.s = s;
    c.verts = verts;
    c.boxes = boxes;
    c.err = err;
    c.err_len = err_len;
    c.prim_seen = (uint8_t *)calloc(s->n_tris + 1, 1);
    c.node_seen = (uint8_t *)calloc(s->n_nodes + 1, 1);
    if (err && err_len) err[0] = 0;
    if (s->n_instances)
        validate_node(&c, s->tlas_start, (uint32_t)s->n_nodes, 0, 1, 0);
    else
        validate_node(&c, 0, (uint32_t)s->n_nodes, 0, 0, 0);
    if (!c.failed)
        for (uint64_t i = 0; i < s->n_tris; i++)
            if (!c.prim_seen[i]) {
                vfail(&c, "primitive %llu never referenced%llu", i, 0);
                break;
            }
    free(c.prim_seen);
    free(c.node_seen);
    return c.failed ? -1 : 0;
}

/* per-ray PROFILE_RT counters of a primary frame (the heat-map of
 * src/rt_gpu/rt_gpu_software.hlsl:93-102 as numbers) */
void orc_count_primary_per_ray(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t sem,
                               int threads, uint16_t *n_node, uint16_t *n_tri) {
    threads = pick_threads(threads);
    const int64_t n = (int64_t)w * h;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
    for (int64_t i = 0; i < n; i++) {
        float o[3], d[3];
        orc_stats st;
        memset(&st, 0, sizeof(st));
        orc_primary_ray(view, w, h, (uint32_t)(i % w), (uint32_t)(i / w), o, d);
        orc_traverse(s, o, d, 0.0f, F32_MAX, sem, &st);
        n_node[i] = (uint16_t)(st.n_node > 65535 ? 65535 : st.n_node);
        n_tri[i] = (uint16_t)(st.n_tri > 65535 ? 65535 : st.n_tri);
    }
}
