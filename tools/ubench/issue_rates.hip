// issue_rates.hip — development microbenchmark (not part of the product): what the non-VALU instructions and the
// code footprint around a CWBVH node test cost on gfx950.  Every kernel runs the product's node test (RCP variant,
// register-resident node) ITERS times per wave; variants add scalar ALU work, taken branches, LDS traffic or unroll
// the body so that the loop no longer fits a few instruction-cache lines.  Prints SIMD cycles per node test at
// 1 / 2 / 4 waves per SIMD (wave cycles / waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -I ../../tray_racing_amd/csrc issue_rates.hip -o issue_rates
#include "../../tray_racing_amd/csrc/kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace trx {
namespace {

#define SALU8 "s_add_u32 s20, s20, 1\n s_xor_b32 s21, s21, s20\n s_add_u32 s22, s22, 3\n s_xor_b32 s23, s23, s22\n" \
              "s_add_u32 s20, s20, 5\n s_xor_b32 s21, s21, s20\n s_add_u32 s22, s22, 7\n s_xor_b32 s23, s23, s22\n"
#define BR1(n) "s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n v_mov_b32 %0, 0\n1:\n"

// VAR: 0 plain, 1 = +96 SALU, 2 = +24 taken branches (each skips 12 instructions = 96 bytes), 3 = +8 LDS write/read pairs,
//      4 = body unrolled 8x (about 12 KB of code), 5 = +96 SALU +24 branches
template <int VAR>
__global__ void __launch_bounds__(1024) k_issue(const uint4 *nodes, uint32_t n_nodes, uint32_t *out, unsigned long long *cyc,
                                                int iters, float negzero) {
    __shared__ uint2 lds[1024 * 2];
    if (VAR >= 8) { lds[threadIdx.x] = make_uint2(threadIdx.x * 2654435761u, threadIdx.x); lds[threadIdx.x + 1024] = make_uint2(threadIdx.x * 40503u, 77u); __syncthreads(); }
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 *np = nodes + (size_t)(tid % n_nodes) * 8;
    uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
    Ray r;
    r.ox = 0.1f * (tid & 7); r.oy = 0.2f * ((tid >> 3) & 7); r.oz = -3.0f; r.tmin = 0.f;
    float dx = 0.01f * (tid & 63) - 0.3f, dy = 0.02f * ((tid >> 2) & 15) - 0.1f, dz = 1.0f;
    finish_ray_dir(r, dx, dy, dz);
    float t = 100.0f;
    uint32_t acc = 0, dummy = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; it += (VAR == 4 ? 8 : 1)) {
#pragma unroll
        for (int u = 0; u < (VAR == 4 ? 8 : 1); u++) {
            uint32_t hit = node_intersect<1>(r, t, n0, n1, n2, n3, n4);
            acc += hit;
            const uint32_t d = hit & 1u;
            n2.x ^= d; n2.y ^= d; n2.z ^= d; n2.w ^= d; n3.x ^= d; n3.y ^= d; n3.z ^= d; n3.w ^= d;
            n4.x ^= d; n4.y ^= d; n4.z ^= d; n4.w ^= d;
            t = t * 0.9999f;
            if (VAR == 1 || VAR == 5) {
                asm volatile(SALU8 SALU8 SALU8 SALU8 SALU8 SALU8 SALU8 SALU8 SALU8 SALU8 SALU8 SALU8 ::: "s20", "s21", "s22", "s23", "scc");
            }
            if (VAR == 2 || VAR == 5) {
                asm volatile(BR1(0) BR1(1) BR1(2) BR1(3) BR1(4) BR1(5) BR1(6) BR1(7) : "+v"(dummy) :: "s20", "scc");
                asm volatile(BR1(0) BR1(1) BR1(2) BR1(3) BR1(4) BR1(5) BR1(6) BR1(7) : "+v"(dummy) :: "s20", "scc");
                asm volatile(BR1(0) BR1(1) BR1(2) BR1(3) BR1(4) BR1(5) BR1(6) BR1(7) : "+v"(dummy) :: "s20", "scc");
            }
            if (VAR == 6 || VAR == 7) {
                // the next node's 80 bytes (L1 / L2 hits): VAR 6 every lane the same node, VAR 7 a node per lane
                const uint32_t idx = VAR == 6 ? ((acc >> 3) + blockIdx.x) % n_nodes : ((acc >> 3) + tid * 7u) % n_nodes;
                const uint4 *q = nodes + (size_t)idx * 8;
                n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; n4 = q[4];
            }
            if (VAR == 8 || VAR == 9) {
                const uint32_t idx = VAR == 8 ? ((acc >> 3) & 63u) : ((acc >> 3) + threadIdx.x) & 63u;
                const uint4 *q = reinterpret_cast<const uint4 *>(lds) + idx * 5;
                n2 = q[2]; n3 = q[3]; n4 = q[4];
                const uint4 a = q[0], b = q[1];
                n0.w = a.w; n1.z = b.z; n1.w = b.w;
            }
            if (VAR == 3) {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    lds[threadIdx.x + 1024 * (k & 1)] = make_uint2(acc, hit + k);
                    __builtin_amdgcn_wave_barrier();
                    const uint2 v = lds[threadIdx.x + 1024 * (k & 1)];
                    acc ^= v.y;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[tid] = acc + dummy;
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

} // namespace
} // namespace trx

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VAR>
static void run(const char *name, const uint4 *d_nodes, uint32_t n_nodes, uint32_t *d_out, unsigned long long *d_cyc, int cus) {
    std::vector<unsigned long long> h(cus * 32);
    const int iters = 4000;
    printf("%-44s", name);
    for (int w : {1, 2, 4}) {
        const int block = 256 * w, grid = cus;
        hipLaunchKernelGGL((trx::k_issue<VAR>), dim3(grid), dim3(block), 0, 0, d_nodes, n_nodes, d_out, d_cyc, 16, -0.0f);
        hipLaunchKernelGGL((trx::k_issue<VAR>), dim3(grid), dim3(block), 0, 0, d_nodes, n_nodes, d_out, d_cyc, iters, -0.0f);
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
    std::vector<uint32_t> host((size_t)n_nodes * 32);
    uint32_t s = 12345;
    for (auto &x : host) { s = s * 1664525u + 1013904223u; x = s; }
    for (uint32_t n = 0; n < n_nodes; n++) {
        float p[3] = {-1.f, -1.f, 1.f};
        memcpy(&host[n * 32], p, 12);
        host[n * 32 + 3] = 0x78787878u & 0x00ffffffu;
    }
    uint4 *d_nodes; uint32_t *d_out; unsigned long long *d_cyc;
    CK(hipMalloc(&d_nodes, host.size() * 4));
    CK(hipMemcpy(d_nodes, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)cus * 1024 * 4));
    CK(hipMalloc(&d_cyc, (size_t)cus * 32 * 8));
    printf("%-44s %9s %9s %9s   SIMD cycles per node test (wave cycles / waves per SIMD)\n", "variant", "1w/SIMD", "2w", "4w");
    run<0>("node test", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<1>("+ 96 SALU", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<2>("+ 24 taken branches (96 B skipped each)", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<3>("+ 8 LDS write/read pairs", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<4>("body unrolled 8x (~12 KB loop)", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<5>("+ 96 SALU + 24 taken branches", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<6>("+ node fetch, 5 x dwordx4, wave-uniform", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<7>("+ node fetch, 5 x dwordx4, a node per lane", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<8>("+ node from LDS, 5 x b128, wave-uniform", d_nodes, n_nodes, d_out, d_cyc, cus);
    run<9>("+ node from LDS, 5 x b128, a node per lane", d_nodes, n_nodes, d_out, d_cyc, cus);
    return 0;
}
