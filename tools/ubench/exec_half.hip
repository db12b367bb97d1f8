// exec_half.hip — development microbenchmark: does a wave64 VALU instruction cost less when one half of the wave
// (lanes 32..63, or 0..31) is switched off in EXEC?  (SIMD-32 issues a wave64 in two passes.)  Also: a quarter.
//   hipcc -O3 --offload-arch=gfx950 exec_half.hip -o exec_half && ./exec_half
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BLOCK8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I)
#define I_MIX(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n v_max3_f32 %" #k ", %8, %9, %" #k "\n v_pk_mul_f32 v[20:21], v[22:23], v[24:25]\n v_fma_f32 %" #k ", %8, %9, %" #k "\n"

__global__ void k_mix(unsigned long long *out, float *sink, int iters, float seed, unsigned long long mask) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7,
          b = seed * 0.5f, c = seed * 0.25f;
    unsigned long long t0 = 0, t1 = 0;
    const unsigned lane = threadIdx.x & 63;
    if ((mask >> lane) & 1ull) {   // only these lanes execute the block: EXEC = mask inside
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
            asm volatile(BLOCK8(I_MIX)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(b), "v"(c)
                         : "vcc", "v20", "v21", "v22", "v23", "v24", "v25");
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    const unsigned first = __ffsll((long long)mask) - 1;
    if (lane == first) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) sink[0] = a0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d_out;
    float *d_sink;
    CK(hipMalloc(&d_out, (size_t)cus * 32 * 8));
    CK(hipMalloc(&d_sink, 64));
    std::vector<unsigned long long> h(cus * 32);
    const int iters = 1000, per_trip = 256;
    struct { const char *name; unsigned long long mask; } masks[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0..31", 0xffffffffull}, {"lanes 32..63", 0xffffffff00000000ull},
        {"lanes 0..15", 0xffffull}, {"every other lane", 0x5555555555555555ull}, {"lanes 0..31 + lane 63", 0x80000000ffffffffull}, {"one lane", 1ull}};
    printf("%-24s %8s %8s %8s   SIMD cycles per instruction (mixed cvt / max3 / pk_mul / fma stream)\n", "EXEC", "1w/SIMD", "2w", "4w");
    for (auto &m : masks) {
        printf("%-24s", m.name);
        for (int w : {1, 2, 4}) {
            const int block = 256 * w, grid = cus;
            hipLaunchKernelGGL(k_mix, dim3(grid), dim3(block), 0, 0, d_out, d_sink, 10, 1.0f, m.mask);
            hipLaunchKernelGGL(k_mix, dim3(grid), dim3(block), 0, 0, d_out, d_sink, iters, 1.0f, m.mask);
            CK(hipDeviceSynchronize());
            const int waves = grid * block / 64;
            CK(hipMemcpy(h.data(), d_out, (size_t)waves * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < waves; i++) sum += (double)h[i];
            printf(" %8.2f", sum / waves / ((double)iters * per_trip) / w);
        }
        printf("\n");
    }
    return 0;
}
