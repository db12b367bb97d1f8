// valu_rates.hip — development microbenchmark (not part of the product): issue cost of the VALU
// instructions the CWBVH node test is made of, on gfx950, at 1 / 2 / 4 / 8 waves per SIMD.
// Prints cycles per wave-instruction per SIMD (s_memtime ticks = shader cycles).
//   hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// one kernel per instruction: 8 independent chains x 8 repeats per loop trip = 64 instructions
#define DEF_KERNEL_F(NAME, ASM)                                                              \
    __global__ void k_##NAME(unsigned long long *out, float *sink, int iters, float seed) { \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, \
              a7 = seed + 7;                                                                 \
        float b = seed * 0.5f, c = seed * 0.25f;                                             \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
        for (int i = 0; i < iters; i++) {                                                    \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                  \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
            }                                                                                \
        }                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) sink[0] = a0;              \
    }

#define DEF_KERNEL_P(NAME, ASM)                                                              \
    __global__ void k_##NAME(unsigned long long *out, float *sink, int iters, float seed) { \
        f32x2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, \
              a7 = a0 + 7.f;                                                                 \
        f32x2 b = a0 * 0.5f, c = a0 * 0.25f;                                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                               \
        for (int i = 0; i < iters; i++) {                                                    \
            _Pragma("unroll") for (int r = 0; r < 8; r++) {                                  \
                asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
                asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21");                              \
            }                                                                                \
        }                                                                                    \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                               \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
        f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                     \
        if (s.x + s.y == 12345.678f) sink[0] = s.x;                                          \
    }

DEF_KERNEL_F(fma, "v_fma_f32 %0, %1, %2, %0")
DEF_KERNEL_F(mul, "v_mul_f32 %0, %1, %0")
DEF_KERNEL_F(add, "v_add_f32 %0, %1, %0")
DEF_KERNEL_F(max, "v_max_f32 %0, %1, %0")
DEF_KERNEL_F(max3, "v_max3_f32 %0, %1, %2, %0")
DEF_KERNEL_F(min3, "v_min3_f32 %0, %1, %2, %0")
DEF_KERNEL_F(cvt_ub0, "v_cvt_f32_ubyte0 %0, %0")
DEF_KERNEL_F(cvt_ub1, "v_cvt_f32_ubyte1 %0, %0")
DEF_KERNEL_F(cvt_ub3, "v_cvt_f32_ubyte3 %0, %0")
DEF_KERNEL_F(cvt_u32, "v_cvt_f32_u32 %0, %0")
DEF_KERNEL_F(cvt_f16, "v_cvt_f32_f16 %0, %0")
DEF_KERNEL_F(cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
DEF_KERNEL_F(cmp_vcc, "v_cmp_le_f32 vcc, %1, %0")
DEF_KERNEL_F(cmp_sgpr, "v_cmp_le_f32 s[20:21], %1, %0")
DEF_KERNEL_F(lshl_or, "v_lshl_or_b32 %0, %1, 3, %0")
DEF_KERNEL_F(and_or, "v_and_or_b32 %0, %1, %2, %0")
DEF_KERNEL_F(bfe, "v_bfe_u32 %0, %0, 8, 8")
DEF_KERNEL_F(lshlrev, "v_lshlrev_b32 %0, 1, %0")
DEF_KERNEL_F(perm, "v_perm_b32 %0, %1, %0, %2")
DEF_KERNEL_F(fma_mix_lo, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]")
DEF_KERNEL_F(fma_mix_hi, "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
DEF_KERNEL_F(mul_sdwa, "v_mul_f32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
DEF_KERNEL_F(cvt_sdwa, "v_cvt_f32_ubyte0_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2")
DEF_KERNEL_F(mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL_F(add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL_F(rcp, "v_rcp_f32 %0, %0")
DEF_KERNEL_F(readlane, "v_readlane_b32 s20, %0, 3")
DEF_KERNEL_F(ffbh, "v_ffbh_u32 %0, %0")
DEF_KERNEL_F(bcnt, "v_bcnt_u32_b32 %0, %1, %0")
DEF_KERNEL_F(mbcnt, "v_mbcnt_lo_u32_b32 %0, -1, %0")
DEF_KERNEL_P(pk_mul, "v_pk_mul_f32 %0, %1, %0")
DEF_KERNEL_P(pk_add, "v_pk_add_f32 %0, %1, %0")
DEF_KERNEL_P(pk_fma, "v_pk_fma_f32 %0, %1, %2, %0")

struct Entry {
    const char *name;
    void (*fn)(unsigned long long *, float *, int, float);
};
#define E(NAME) {#NAME, k_##NAME}
static const Entry entries[] = {
    E(fma), E(mul), E(add), E(max), E(max3), E(min3), E(cvt_ub0), E(cvt_ub1), E(cvt_ub3), E(cvt_u32), E(cvt_f16),
    E(cndmask), E(cmp_vcc), E(cmp_sgpr), E(lshl_or), E(and_or), E(bfe), E(lshlrev), E(perm), E(fma_mix_lo), E(fma_mix_hi),
    E(mul_sdwa), E(cvt_sdwa), E(mov_dpp), E(add_dpp), E(rcp), E(readlane), E(ffbh), E(bcnt), E(mbcnt), E(pk_mul), E(pk_add), E(pk_fma),
};

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                    \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long *d_out;
    float *d_sink;
    const int max_waves = cus * 32;
    CK(hipMalloc(&d_out, max_waves * 8));
    CK(hipMalloc(&d_sink, 64));
    std::vector<unsigned long long> h(max_waves);
    const int iters = 2000, per_iter = 64;
    printf("%-12s %8s %8s %8s %8s   (cycles per wave-instruction per SIMD; %d CUs)\n", "instr", "1w/SIMD", "2w", "4w", "8w", cus);
    for (const Entry &e : entries) {
        printf("%-12s", e.name);
        for (int w : {1, 2, 4, 8}) {
            // one workgroup per CU of w*4 waves: w waves land on each SIMD
            const int block = 256 * w > 1024 ? 1024 : 256 * w;
            const int blocks_per_cu = (256 * w) / block;
            const int grid = cus * blocks_per_cu;
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(block), 0, 0, d_out, d_sink, 10, 1.0f); // warm-up
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(block), 0, 0, d_out, d_sink, iters, 1.0f);
            CK(hipDeviceSynchronize());
            const int waves = grid * block / 64;
            CK(hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < waves; i++) sum += (double)h[i];
            const double cyc_per_wave = sum / waves;
            // w waves share a SIMD: SIMD time per instruction = wave time / (instructions of one wave * w)
            printf(" %8.2f", cyc_per_wave / ((double)iters * per_iter) / w);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
