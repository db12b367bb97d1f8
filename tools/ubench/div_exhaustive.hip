// div_exhaustive.hip - is  q = fma(fma(-d, a*y, a), y, a*y)  with  y = RN(1/d)  the IEEE quotient RN(a/d)?
//
// The literal-HLSL node test divides three times per node by the ray's direction (query.hlsl:237-243, b = (p - o) / d).
// kernels.hip computes those quotients from the ray's correctly rounded reciprocal (Markstein's correction step:
// q0 = RN(a y), r = a - d q0 exactly in one fma, q = RN(q0 + r y)) where the ray's and the scene's flags allow it.
// Binary32 multiplication, fma and division commute with scaling by powers of two as long as nothing leaves the normal
// range, so whether the identity holds depends on the two SIGNIFICANDS only (and not on the signs: round-to-nearest is
// symmetric).  This program checks
//   pass 1: every pair of significands, 2^23 x 2^23 = 7.04e13 quotients (a, d in [1, 2));
//   pass 2: 2^36 random pairs with random signs and exponents over the range the kernel admits
//           (a = +0 or 2^-60 <= |a| <= 2^60, 2^-30 <= |d| <= 2^20) - the scaling argument, measured;
// against the compiler's own `/` (built with -fhip-fp32-correctly-rounded-divide-sqrt, like the kernels).
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math \
//         tools/ubench/div_exhaustive.hip -o /tmp/div_exhaustive && /tmp/div_exhaustive [slices]
// (slices: how many of the 32 slices of pass 1 to run; default all.  About a minute on one MI355X.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                \
        }                                                                           \
    } while (0)

struct Report {
    unsigned long long mismatches;
    unsigned long long tested;
    uint32_t a, d, want, got; // the first mismatch seen (any thread)
};

__device__ __forceinline__ float div_by_rcp(float a, float d, float y) {
    const float q0 = a * y;
    const float rem = __builtin_fmaf(-d, q0, a);
    return __builtin_fmaf(rem, y, q0);
}

__device__ __forceinline__ void note(Report *rep, float a, float d, float want, float got) {
    if (atomicAdd(&rep->mismatches, 1ull) == 0ull) {
        rep->a = __float_as_uint(a);
        rep->d = __float_as_uint(d);
        rep->want = __float_as_uint(want);
        rep->got = __float_as_uint(got);
    }
}

// one thread per divisor significand of the slice; every dividend significand in turn
__global__ void __launch_bounds__(256) k_all_significands(uint32_t d_first, Report *rep) {
    const uint32_t md = d_first + blockIdx.x * blockDim.x + threadIdx.x;
    const float d = __uint_as_float(0x3f800000u | md);
    const float y = 1.0f / d;
    uint32_t bad = 0u;
#pragma unroll 8
    for (uint32_t ma = 0u; ma < (1u << 23); ma++) {
        const float a = __uint_as_float(0x3f800000u | ma);
        const float want = a / d, got = div_by_rcp(a, d, y);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            bad++;
            note(rep, a, d, want, got);
        }
    }
    (void)bad;
    if (threadIdx.x == 0) atomicAdd(&rep->tested, (unsigned long long)blockDim.x << 23);
}

__device__ __forceinline__ uint64_t mix(uint64_t s) {
    s ^= s >> 30; s *= 0xBF58476D1CE4E5B9ull;
    s ^= s >> 27; s *= 0x94D049BB133111EBull;
    s ^= s >> 31;
    return s;
}

// random signs, exponents over the admitted range (edges included), random significands
__global__ void __launch_bounds__(256) k_random_exponents(uint64_t seed, uint32_t per_thread, Report *rep) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = mix(seed + tid * 0x9E3779B97F4A7C15ull);
    for (uint32_t i = 0; i < per_thread; i++) {
        s = mix(s + 0x9E3779B97F4A7C15ull);
        const uint64_t u = s;
        s = mix(s + 0x9E3779B97F4A7C15ull);
        const uint64_t v = s;
        const uint32_t ea = 127u - 60u + (uint32_t)((u >> 48) % 121u); // 2^-60 .. 2^60
        const uint32_t ed = 127u - 30u + (uint32_t)((v >> 48) % 51u);  // 2^-30 .. 2^20
        uint32_t ma = (uint32_t)u & 0x7fffffu, md = (uint32_t)v & 0x7fffffu;
        // (the admitted range is closed: at the top exponent only the power of two itself)
        if (ea == 127u + 60u) ma = 0u;
        if (ed == 127u + 20u) md = 0u;
        // one pair in eight from the corners of the significand range
        if (((u >> 40) & 7u) == 0u) ma = ((u >> 43) & 1u) ? 0x7fffffu - (ma & 3u) : (ma & 3u);
        if (((v >> 40) & 7u) == 0u) md = ((v >> 43) & 1u) ? 0x7fffffu - (md & 3u) : (md & 3u);
        float a = __uint_as_float(((uint32_t)(u >> 32) & 0x80000000u) | (ea << 23) | ma);
        const float d = __uint_as_float(((uint32_t)(v >> 32) & 0x80000000u) | (ed << 23) | md);
        if (((u >> 44) & 63u) == 0u) a = 0.0f; // one numerator in 64 is +0 (p - o with p == o): the quotient is the zero of d's sign
        const float y = 1.0f / d;
        const float want = a / d, got = div_by_rcp(a, d, y);
        if (__float_as_uint(want) != __float_as_uint(got)) note(rep, a, d, want, got);
    }
    if (threadIdx.x == 0) atomicAdd(&rep->tested, (unsigned long long)blockDim.x * per_thread);
}

int main(int argc, char **argv) {
    const int slices = argc > 1 ? atoi(argv[1]) : 32;
    Report *rep = nullptr;
    CHECK(hipMalloc(&rep, sizeof(Report)));
    CHECK(hipMemset(rep, 0, sizeof(Report)));
    Report host;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    const uint32_t per_slice = (1u << 23) / 32u;
    for (int s = 0; s < slices && s < 32; s++) {
        k_all_significands<<<per_slice / 256u, 256>>>((uint32_t)s * per_slice, rep);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(&host, rep, sizeof(Report), hipMemcpyDeviceToHost));
        if ((s & 7) == 7 || s + 1 == slices) {
            printf("pass 1, slice %d/32: %llu quotients, %llu mismatches\n", s + 1, host.tested, host.mismatches);
            fflush(stdout);
        }
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(&host, rep, sizeof(Report), hipMemcpyDeviceToHost));
    printf("pass 1 (every pair of significands): %llu quotients, %llu mismatches, %.1f s\n", host.tested, host.mismatches, ms * 1e-3);
    if (host.mismatches) printf("  first: a=%08x d=%08x want=%08x got=%08x\n", host.a, host.d, host.want, host.got);
    const unsigned long long m1 = host.mismatches;

    CHECK(hipMemset(rep, 0, sizeof(Report)));
    CHECK(hipEventRecord(e0));
    for (int s = 0; s < 16; s++) { // 16 x 2^20 threads x 4096 = 2^36
        k_random_exponents<<<(1u << 20) / 256u, 256>>>(0x5eedull + (uint64_t)s * 0x100000001ull, 4096u, rep);
        CHECK(hipGetLastError());
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(&host, rep, sizeof(Report), hipMemcpyDeviceToHost));
    printf("pass 2 (random signs and exponents, a = +0 or 2^-60 <= |a| <= 2^60, 2^-30 <= |d| <= 2^20): %llu quotients, %llu mismatches, %.1f s\n",
           host.tested, host.mismatches, ms * 1e-3);
    if (host.mismatches) printf("  first: a=%08x d=%08x want=%08x got=%08x\n", host.a, host.d, host.want, host.got);
    CHECK(hipFree(rep));
    return (m1 || host.mismatches) ? 1 : 0;
}
