// valu_rates2.hip — development microbenchmark (not part of the product): issue cost of single instructions on
// gfx950 at 1 / 2 / 4 / 8 waves per SIMD, in one inline-asm block of 64 instructions per loop trip (8 independent
// chains x 8), so the compiler inserts nothing between them (valu_rates.hip got an s_nop after every instruction).
// Prints SIMD cycles per instruction (wave cycles per instruction / waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 valu_rates2.hip -o valu_rates2 && ./valu_rates2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define BLOCK8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I)

#define DEF(NAME, I)                                                                                          \
    __global__ void k_##NAME(unsigned long long *out, float *sink, int iters, float seed) {                  \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, \
              a7 = seed + 7, b = seed * 0.5f, c = seed * 0.25f;                                               \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int i = 0; i < iters; i++) {                                                                     \
            asm volatile(BLOCK8(I)                                                                           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c)                                                                     \
                         : "vcc", "s20", "s21", "s22", "s23", "scc");                                        \
        }                                                                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;           \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) sink[0] = a0;                               \
    }

#define I_FMA(k) "v_fma_f32 %" #k ", %8, %9, %" #k "\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %8, %" #k "\n"
#define I_MAX(k) "v_max_f32 %" #k ", %8, %" #k "\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %8, %9, %" #k "\n"
#define I_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
#define I_PKMUL(k) "v_pk_mul_f32 v[20:21], v[22:23], v[24:25]\n"
#define I_PKADD(k) "v_pk_add_f32 v[20:21], v[22:23], v[24:25]\n"
#define I_PKFMA(k) "v_pk_fma_f32 v[20:21], v[22:23], v[24:25], v[26:27]\n"
#define I_AND(k) "v_and_b32 %" #k ", %8, %" #k "\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %8, %" #k "\n"
#define I_CNDVCC(k) "v_cndmask_b32 %" #k ", %8, %" #k ", vcc\n"
#define I_CNDSG(k) "v_cndmask_b32_e64 %" #k ", %8, %" #k ", s[20:21]\n"
#define I_CMP(k) "v_cmp_le_f32 vcc, %8, %" #k "\n"
#define I_CMPCND(k) "v_cmp_le_f32 vcc, %8, %" #k "\n v_cndmask_b32 %" #k ", %8, %" #k ", vcc\n"
#define I_OR3(k) "v_or3_b32 %" #k ", %8, %9, %" #k "\n"
#define I_LSHLSDWA(k) "v_lshlrev_b32_sdwa %" #k ", %8, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_MADU64(k) "v_mad_u64_u32 v[20:21], s[22:23], %8, %9, v[24:25]\n"
#define I_FFBH(k) "v_ffbh_u32 %" #k ", %" #k "\n"
#define I_BCNT(k) "v_bcnt_u32_b32 %" #k ", %8, %" #k "\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %8, %" #k "\n"
#define I_MULU24(k) "v_mul_u32_u24 %" #k ", %8, %" #k "\n"
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %8, 2, %" #k "\n"
#define I_SUBU(k) "v_sub_u32 %" #k ", %8, %" #k "\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_MOVDPP(k) "v_mov_b32_dpp %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_SAND(k) "s_and_b64 s[20:21], s[20:21], s[22:23]\n"
#define I_SADD(k) "s_add_u32 s20, s20, s22\n"
#define I_SNOP(k) "s_nop 0\n"
#define I_FMA_SALU(k) "v_fma_f32 %" #k ", %8, %9, %" #k "\n s_add_u32 s20, s20, s22\n"
#define I_MIX(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n v_max3_f32 %" #k ", %8, %9, %" #k "\n v_fma_f32 %" #k ", %8, %9, %" #k "\n v_and_b32 %" #k ", %8, %" #k "\n"

DEF(fma, I_FMA) DEF(mul, I_MUL) DEF(max, I_MAX) DEF(max3, I_MAX3) DEF(cvt_ub, I_CVTUB) DEF(pk_mul, I_PKMUL) DEF(pk_add, I_PKADD)
DEF(pk_fma, I_PKFMA) DEF(and_b32, I_AND) DEF(lshl, I_LSHL) DEF(bfe, I_BFE) DEF(add_u32, I_ADDU) DEF(cnd_vcc, I_CNDVCC)
DEF(cnd_sgpr, I_CNDSG) DEF(cmp, I_CMP) DEF(cmp_cnd, I_CMPCND) DEF(or3, I_OR3) DEF(lshl_sdwa, I_LSHLSDWA) DEF(mad_u64, I_MADU64)
DEF(ffbh, I_FFBH) DEF(bcnt, I_BCNT) DEF(rcp, I_RCP) DEF(mov_dpp, I_MOVDPP) DEF(s_and, I_SAND) DEF(s_add, I_SADD) DEF(s_nop, I_SNOP)
DEF(fma_salu, I_FMA_SALU) DEF(mix4, I_MIX) DEF(mul_lo, I_MULLO) DEF(mul_u24, I_MULU24) DEF(lshl_or, I_LSHLOR) DEF(sub_u32, I_SUBU)

struct Entry {
    const char *name;
    void (*fn)(unsigned long long *, float *, int, float);
    int per_trip;
};
#define E(n, c) {#n, k_##n, c}
static const Entry entries[] = {
    E(fma, 64), E(mul, 64), E(max, 64), E(max3, 64), E(cvt_ub, 64), E(pk_mul, 64), E(pk_add, 64), E(pk_fma, 64), E(and_b32, 64), E(lshl, 64),
    E(bfe, 64), E(add_u32, 64), E(cnd_vcc, 64), E(cnd_sgpr, 64), E(cmp, 64), E(cmp_cnd, 128), E(or3, 64), E(lshl_sdwa, 64),
    E(mad_u64, 64), E(ffbh, 64), E(bcnt, 64), E(rcp, 64), E(mov_dpp, 64), E(s_and, 64), E(s_add, 64), E(s_nop, 64), E(fma_salu, 128), E(mix4, 256), E(mul_lo, 64), E(mul_u24, 64), E(lshl_or, 64), E(sub_u32, 64),
};

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
    const int iters = 2000;
    printf("%-10s %8s %8s %8s %8s   SIMD cycles per instruction (wave cycles per instruction / waves per SIMD); %d CUs\n", "instr", "1w/SIMD", "2w", "4w", "8w", cus);
    for (const Entry &e : entries) {
        printf("%-10s", e.name);
        for (int w : {1, 2, 4, 8}) {
            const int block = 256 * w > 1024 ? 1024 : 256 * w;
            const int grid = cus * ((256 * w) / block);
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(block), 0, 0, d_out, d_sink, 10, 1.0f);
            hipLaunchKernelGGL(e.fn, dim3(grid), dim3(block), 0, 0, d_out, d_sink, iters, 1.0f);
            CK(hipDeviceSynchronize());
            const int waves = grid * block / 64;
            CK(hipMemcpy(h.data(), d_out, (size_t)waves * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < waves; i++) sum += (double)h[i];
            printf(" %8.2f", sum / waves / ((double)iters * e.per_trip) / w);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
