// host_poll.hip - what a look into pinned host memory costs one wave, by how it looks (round 6, the ray service's porter):
// scalar loads (s_load_dword / x4 / x16, 1 .. 8 in flight) against vector loads (1 lane .. 24 lanes x 16 bytes).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/host_poll.hip -o /tmp/host_poll && /tmp/host_poll
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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(64) k(const uint32_t *host, int mode, int reps, unsigned long long *out) {
    const uint32_t lane = threadIdx.x;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < reps; i++) {
        switch (mode) {
        case 0: { uint32_t v; asm volatile("s_load_dword %0, %1, 0x2c glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(host) : "memory"); acc += v; break; }
        case 1: { u32x4 v; asm volatile("s_load_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(host) : "memory"); acc += v.x; break; }
        case 2: { u32x16 v; asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(host) : "memory"); acc += v.x; break; }
        case 3: { u32x4 a, b; asm volatile("s_load_dwordx4 %0, %2, 0x0 glc\n\ts_load_dwordx4 %1, %2, 0x80 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(host) : "memory"); acc += a.x + b.x; break; }
        case 4: { u32x4 a, b, c; asm volatile("s_load_dwordx4 %0, %3, 0x0 glc\n\ts_load_dwordx4 %1, %3, 0x10 glc\n\ts_load_dwordx4 %2, %3, 0x20 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(host) : "memory"); acc += a.x + b.x + c.x; break; }
        case 5: { u32x4 a, b, c, d; asm volatile("s_load_dwordx4 %0, %4, 0x0 glc\n\ts_load_dwordx4 %1, %4, 0x80 glc\n\ts_load_dwordx4 %2, %4, 0x100 glc\n\ts_load_dwordx4 %3, %4, 0x180 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d) : "s"(host) : "memory"); acc += a.x + b.x + c.x + d.x; break; }
        case 6: { uint32_t a, b, c, d, e, f, g, h; asm volatile("s_load_dword %0, %8, 0x2c glc\n\ts_load_dword %1, %8, 0xac glc\n\ts_load_dword %2, %8, 0x12c glc\n\ts_load_dword %3, %8, 0x1ac glc\n\ts_load_dword %4, %8, 0x22c glc\n\ts_load_dword %5, %8, 0x2ac glc\n\ts_load_dword %6, %8, 0x32c glc\n\ts_load_dword %7, %8, 0x3ac glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b), "=&s"(c), "=&s"(d), "=&s"(e), "=&s"(f), "=&s"(g), "=&s"(h) : "s"(host) : "memory"); acc += a + b + c + d + e + f + g + h; break; }
        case 7: { u32x16 a, b; asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(host) : "memory"); acc += a.x + b.x; break; }
        case 8: { u32x4 v = {0u, 0u, 0u, 0u}; if (lane == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(host) : "memory"); asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); acc += v.x; break; }
        case 9: { u32x4 v = {0u, 0u, 0u, 0u}; if (lane < 24 && (lane & 7) < 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(host + (lane >> 3) * 32 + (lane & 7) * 4) : "memory"); asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); acc += v.x; break; }
        case 10: { u32x4 v = {0u, 0u, 0u, 0u}; if ((lane & 7) < 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(host + (lane >> 3) * 32 + (lane & 7) * 4) : "memory"); asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); acc += v.x; break; }
        case 11: { uint32_t v = 0u; if (lane < 8) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(host + lane * 32 + 11) : "memory"); asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory"); acc += v; break; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) {
        out[0] = t1 - t0;
        out[1] = acc;
    }
}

int main() {
    uint32_t *host;
    unsigned long long *out;
    CHECK(hipHostMalloc((void **)&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipHostMalloc((void **)&out, 64, hipHostMallocCoherent | hipHostMallocMapped));
    for (int i = 0; i < 1024; i++) host[i] = i;
    const char *names[] = {"scalar: one dword", "scalar: one x4 (16 B)", "scalar: one x16 (64 B)", "scalar: two x4, two lines", "scalar: three x4, one line",
                           "scalar: four x4, four lines", "scalar: eight dwords, eight lines", "scalar: two x16, two lines", "vector: one lane x 16 B",
                           "vector: 9 lanes x 16 B (3 slots)", "vector: 24 lanes x 16 B (8 slots)", "vector: 8 lanes x 4 B, eight lines"};
    const int reps = 2000;
    for (int mode = 0; mode < 12; mode++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, host, mode, reps, out);
        CHECK(hipDeviceSynchronize());
        printf("%-36s: %7.0f cycles per look\n", names[mode], (double)out[0] / reps);
    }
    return 0;
}
