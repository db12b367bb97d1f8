// node_rates.hip — development microbenchmark (not part of the product): SIMD cycles per CWBVH node test
// (register-resident node, no memory traffic) for the arithmetic variants of kernels.hip and for
// prototype formulations, at 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -I ../../tray_racing_amd/csrc node_rates.hip -o node_rates
#include "../../tray_racing_amd/csrc/kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace trx {
namespace {

// scalar formulation of the RCP variant (no packed f32)
template <int NODE>
__device__ __forceinline__ uint32_t node_intersect_scalar(const Ray &r, float max_distance, const uint4 n0, const uint4 n1,
                                                          const uint4 n2, const uint4 n3, const uint4 n4) {
    const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
    const uint32_t e_imask = n0.w;
    const float ex = __uint_as_float((e_imask & 0xffu) << 23);
    const float ey = __uint_as_float(((e_imask >> 8) & 0xffu) << 23);
    const float ez = __uint_as_float(((e_imask >> 16) & 0xffu) << 23);
    const float ax = ex * r.ix, ay = ey * r.iy, az = ez * r.iz;
    const float bx = (px - r.ox) * r.ix, by = (py - r.oy) * r.iy, bz = (pz - r.oz) * r.iz;
    const bool nx = r.dx < 0.0f, ny = r.dy < 0.0f, nz = r.dz < 0.0f;
    uint32_t hit_mask = 0;
#pragma unroll
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
            const float t0x = plane<NODE>(ubyte(x_min, j), ax, bx), t1x = plane<NODE>(ubyte(x_max, j), ax, bx);
            const float t0y = plane<NODE>(ubyte(y_min, j), ay, by), t1y = plane<NODE>(ubyte(y_max, j), ay, by);
            const float t0z = plane<NODE>(ubyte(z_min, j), az, bz), t1z = plane<NODE>(ubyte(z_max, j), az, bz);
            const float tmin = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), 0.0001f);
            const float tmax = fminf(fminf(fminf(t1x, t1y), t1z), max_distance);
            if (tmin <= tmax) {
                const uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
                const uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
                hit_mask |= child_bits << bit_index;
            }
        }
    }
    return hit_mask;
}

// prototype: child planes stored as f16 numbers (0..255 are exact), one uint4 = one plane set of the 8
// children; q * a evaluated by v_fma_mix_f32 (f16 x f32 + (-0) -> one rounding, the same as cvt + mul)
template <int HALF>
__device__ __forceinline__ float mixmul(uint32_t h2, float a, float negzero) {
    float d;
    if (HALF == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(a), "v"(negzero));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(a), "v"(negzero));
    return d;
}
template <int HALF>
__device__ __forceinline__ float mixfma(uint32_t h2, float a, float b) {
    float d;
    if (HALF == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(a), "v"(b));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t word(const uint4 v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

// xn/xf...: near / far plane sets already chosen by the ray's sign (per-lane load offsets in the real kernel)
template <int FMA>
__device__ __forceinline__ uint32_t node_intersect_h16(const Ray &r, float max_distance, const uint4 n0, const uint4 n1,
                                                       const uint4 xn, const uint4 xf, const uint4 yn, const uint4 yf,
                                                       const uint4 zn, const uint4 zf, float negzero) {
    const float px = __uint_as_float(n0.x), py = __uint_as_float(n0.y), pz = __uint_as_float(n0.z);
    const uint32_t e_imask = n0.w;
    const float ex = __uint_as_float((e_imask & 0xffu) << 23);
    const float ey = __uint_as_float(((e_imask >> 8) & 0xffu) << 23);
    const float ez = __uint_as_float(((e_imask >> 16) & 0xffu) << 23);
    const float ax = ex * r.ix, ay = ey * r.iy, az = ez * r.iz;
    const float bx = (px - r.ox) * r.ix, by = (py - r.oy) * r.iy, bz = (pz - r.oz) * r.iz;
    uint32_t hit_mask = 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t meta4 = i == 0 ? n1.z : n1.w;
        const uint32_t is_inner4 = (meta4 & (meta4 << 1)) & 0x10101010u;
        const uint32_t inner_mask4 = (is_inner4 >> 4) * 0xffu;
        const uint32_t bit_index4 = (meta4 ^ (r.oct_inv4 & inner_mask4)) & 0x1f1f1f1fu;
        const uint32_t child_bits4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = 4 * i + j;
            float t0x, t1x, t0y, t1y, t0z, t1z;
            if (FMA) {
                if (c & 1) {
                    t0x = mixfma<1>(word(xn, c >> 1), ax, bx); t1x = mixfma<1>(word(xf, c >> 1), ax, bx);
                    t0y = mixfma<1>(word(yn, c >> 1), ay, by); t1y = mixfma<1>(word(yf, c >> 1), ay, by);
                    t0z = mixfma<1>(word(zn, c >> 1), az, bz); t1z = mixfma<1>(word(zf, c >> 1), az, bz);
                } else {
                    t0x = mixfma<0>(word(xn, c >> 1), ax, bx); t1x = mixfma<0>(word(xf, c >> 1), ax, bx);
                    t0y = mixfma<0>(word(yn, c >> 1), ay, by); t1y = mixfma<0>(word(yf, c >> 1), ay, by);
                    t0z = mixfma<0>(word(zn, c >> 1), az, bz); t1z = mixfma<0>(word(zf, c >> 1), az, bz);
                }
            } else if (c & 1) {
                t0x = mixmul<1>(word(xn, c >> 1), ax, negzero) + bx; t1x = mixmul<1>(word(xf, c >> 1), ax, negzero) + bx;
                t0y = mixmul<1>(word(yn, c >> 1), ay, negzero) + by; t1y = mixmul<1>(word(yf, c >> 1), ay, negzero) + by;
                t0z = mixmul<1>(word(zn, c >> 1), az, negzero) + bz; t1z = mixmul<1>(word(zf, c >> 1), az, negzero) + bz;
            } else {
                t0x = mixmul<0>(word(xn, c >> 1), ax, negzero) + bx; t1x = mixmul<0>(word(xf, c >> 1), ax, negzero) + bx;
                t0y = mixmul<0>(word(yn, c >> 1), ay, negzero) + by; t1y = mixmul<0>(word(yf, c >> 1), ay, negzero) + by;
                t0z = mixmul<0>(word(zn, c >> 1), az, negzero) + bz; t1z = mixmul<0>(word(zf, c >> 1), az, negzero) + bz;
            }
            const float tmin = fmaxf(fmaxf(fmaxf(t0x, t0y), t0z), 0.0001f);
            const float tmax = fminf(fminf(fminf(t1x, t1y), t1z), max_distance);
            if (tmin <= tmax) {
                const uint32_t child_bits = (child_bits4 >> (8 * j)) & 0xffu;
                const uint32_t bit_index = (bit_index4 >> (8 * j)) & 0xffu;
                hit_mask |= child_bits << bit_index;
            }
        }
    }
    return hit_mask;
}

// VAR 0..3: node_intersect<VAR>; 4: scalar RCP; 5: scalar RCP+FMA; 6: f16 planes, mul+add; 7: f16 planes, fused
template <int VAR>
__global__ void __launch_bounds__(1024) k_node(const uint4 *nodes, uint32_t n_nodes, uint32_t *out, unsigned long long *cyc,
                                               int iters, float negzero) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 *np = nodes + (size_t)(tid % n_nodes) * 8;
    uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4], n5 = np[5], n6 = np[6], n7 = np[7];
    Ray r;
    r.ox = 0.1f * (tid & 7); r.oy = 0.2f * ((tid >> 3) & 7); r.oz = -3.0f; r.tmin = 0.f;
    float dx = 0.01f * (tid & 63) - 0.3f, dy = 0.02f * ((tid >> 2) & 15) - 0.1f, dz = 1.0f;
    finish_ray_dir(r, dx, dy, dz);
    float t = 100.0f;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint32_t hit;
        if (VAR < 4) hit = node_intersect<(VAR & 3)>(r, t, n0, n1, n2, n3, n4);
        else if (VAR == 4) hit = node_intersect_scalar<1>(r, t, n0, n1, n2, n3, n4);
        else if (VAR == 5) hit = node_intersect_scalar<3>(r, t, n0, n1, n2, n3, n4);
        else if (VAR == 6) hit = node_intersect_h16<0>(r, t, n0, n1, n2, n3, n4, n5, n6, n7, negzero);
        else hit = node_intersect_h16<1>(r, t, n0, n1, n2, n3, n4, n5, n6, n7, negzero);
        acc += hit;
        // keep the node loop-variant so the conversions stay inside the loop (12 + 12 cheap ops)
        const uint32_t d = hit & 1u;
        n2.x ^= d; n2.y ^= d; n2.z ^= d; n2.w ^= d; n3.x ^= d; n3.y ^= d; n3.z ^= d; n3.w ^= d;
        n4.x ^= d; n4.y ^= d; n4.z ^= d; n4.w ^= d;
        if (VAR >= 6) { n5.x ^= d; n5.y ^= d; n5.z ^= d; n5.w ^= d; n6.x ^= d; n6.y ^= d; n6.z ^= d; n6.w ^= d;
                        n7.x ^= d; n7.y ^= d; n7.z ^= d; n7.w ^= d; }
        t = t * 0.9999f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[tid] = acc;
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

} // namespace
} // namespace trx

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static uint16_t f16_of_byte(uint32_t q) { // exact: integers 0..255 fit the 11-bit significand
    if (q == 0) return 0;
    int e = 31 - __builtin_clz(q);
    return (uint16_t)(((e + 15) << 10) | ((q << (10 - e)) & 0x3ff));
}

template <int VAR>
static void run(const char *name, const uint4 *d_nodes, uint32_t n_nodes, uint32_t *d_out, unsigned long long *d_cyc, int cus) {
    std::vector<unsigned long long> h(cus * 32);
    const int iters = 4000;
    printf("%-28s", name);
    for (int w : {1, 2, 4}) {
        const int block = 256 * w, grid = cus;
        hipLaunchKernelGGL((trx::k_node<VAR>), dim3(grid), dim3(block), 0, 0, d_nodes, n_nodes, d_out, d_cyc, 10, -0.0f);
        hipLaunchKernelGGL((trx::k_node<VAR>), dim3(grid), dim3(block), 0, 0, d_nodes, n_nodes, d_out, d_cyc, iters, -0.0f);
        CK(hipDeviceSynchronize());
        const int waves = grid * block / 64;
        CK(hipMemcpy(h.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for (int i = 0; i < waves; i++) sum += (double)h[i];
        printf(" %9.1f", sum / waves / iters / w);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const uint32_t n_nodes = 4096;
    std::vector<uint32_t> host(n_nodes * 32);
    uint32_t s = 12345u;
    for (size_t i = 0; i < host.size(); i++) {
        s = s * 1664525u + 1013904223u;
        host[i] = s;
    }
    for (uint32_t n = 0; n < n_nodes; n++) { // plausible header: p in [-1,1], e around 2^-7, f16 planes 0..255
        float p[3] = {-1.f + 0.001f * n, -0.5f, 0.f};
        std::memcpy(&host[n * 32], p, 12);
        host[n * 32 + 3] = 0x00787878u | (0x0fu << 24);
        for (int k = 8; k < 32; k++) {
            const uint32_t a = host[n * 32 + k] & 0xff, b = (host[n * 32 + k] >> 8) & 0xff;
            const uint16_t ua = f16_of_byte(a), ub = f16_of_byte(b);
            if (k >= 20) host[n * 32 + k] = ua | ((uint32_t)ub << 16); // words 20.. are only read as f16 by VAR 6/7 (n5..n7)
        }
    }
    uint4 *d_nodes;
    uint32_t *d_out;
    unsigned long long *d_cyc;
    CK(hipMalloc(&d_nodes, host.size() * 4));
    CK(hipMemcpy(d_nodes, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)cus * 1024 * 4));
    CK(hipMalloc(&d_cyc, (size_t)cus * 32 * 8));
    printf("%-28s %9s %9s %9s   SIMD cycles per node test (wave cycles / waves per SIMD)\n", "variant", "1w/SIMD", "2w", "4w");
    run<0>("hlsl: divides, pk mul+add", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<1>("rcp, pk mul+add (bench)", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<3>("rcp, pk fma", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<4>("rcp, scalar mul+add", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<5>("rcp, scalar fma", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<6>("f16 planes, mix-mul + add", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<7>("f16 planes, mix-fma", d_nodes, n_nodes, d_out, d_cyc, cus);
    return 0;
}
