// tcp_order.hip - does a wave's cache-hit load wait for ANOTHER wave's slow load on the same CU?
//
// Found in the ray service (round 6): its walker wave's node requests - L2 hits - took 1 800 cycles while the porter wave of
// the same workgroup kept reads of pinned host memory in flight, 600 once the porter polled through the scalar cache.
// This program measures the general form.  One workgroup, two waves (same CU): wave 0 chases a pointer through a 4 KB ring
// (every step a dependent L1 / L2 hit) and reports cycles per step; wave 1 meanwhile does one of
//   0  nothing (sleeps)
//   1  dependent loads striding through 1 GB (one HBM miss in flight)
//   2  the same, eight independent misses in flight
//   3  returning atomic adds on one global word
//   4  dependent loads from pinned host memory (one PCIe read in flight)
//   5  the same through the SCALAR cache (s_load_dword glc)
//   6  stores to pinned host memory
// Second half: the same with wave 1 in ANOTHER workgroup (another CU) - the control.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/tcp_order.hip -o /tmp/tcp_order && /tmp/tcp_order
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

__global__ void __launch_bounds__(128) k(const uint32_t *ring, const uint32_t *big, uint32_t big_words, uint32_t *word, uint32_t *host,
                                         int mode, int split, int steps, unsigned long long *out, volatile uint32_t *stop) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool chaser = split ? (blockIdx.x == 0 && wave == 0) : wave == 0;
    const bool noisy = split ? (blockIdx.x == 1 && wave == 0) : wave == 1;
    if (chaser) {
        uint32_t p = lane * 16u; // (every lane its own chain: 64 lanes x 16 words)
        // warm
        for (int i = 0; i < 64; i++) p = ring[p];
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < steps; i++) p = ring[p];
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0) {
            out[0] = t1 - t0;
            out[1] = p;
            __threadfence();
            *stop = 1u;
        }
    } else if (noisy) {
        uint32_t a = (uint32_t)lane * 4099u, acc = 0u;
        unsigned long long n = 0;
        while (*stop == 0u) {
            switch (mode) {
            case 0: __builtin_amdgcn_s_sleep(64); break;
            case 1:
                a = (a * 1664525u + 1013904223u + acc) % big_words;
                acc += __builtin_nontemporal_load(big + a) & 1u;
                break;
            case 2:
                for (int j = 0; j < 8; j++) {
                    a = (a * 1664525u + 1013904223u) % big_words;
                    acc += __builtin_nontemporal_load(big + a);
                }
                break;
            case 3:
                if (lane == 0) acc += atomicAdd(word, 1u);
                break;
            case 4:
                if (lane == 0) {
                    uint32_t v;
                    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(host + (n & 15u) * 32u) : "memory");
                    acc += v;
                }
                break;
            case 5: {
                uint32_t v;
                asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(host) : "memory");
                acc += v;
                break;
            }
            case 6:
                if (lane == 0) asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(host + 64u + (n & 15u) * 32u), "v"(acc) : "memory");
                break;
            }
            n++;
        }
        if (lane == 0) {
            out[2] = n;
            out[3] = acc;
        }
    }
}

int main() {
    uint32_t *ring, *big, *word, *host, *stop;
    unsigned long long *out;
    const uint32_t big_words = 1u << 28; // 1 GB
    CHECK(hipMalloc(&ring, 4096));
    CHECK(hipMalloc(&big, (size_t)big_words * 4));
    CHECK(hipMemset(big, 0, (size_t)big_words * 4));
    CHECK(hipMalloc(&word, 4));
    CHECK(hipMemset(word, 0, 4));
    CHECK(hipHostMalloc((void **)&host, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    CHECK(hipMalloc(&stop, 64));   // (device memory: the other wave's look at it must not be a read of host memory itself)
    CHECK(hipMalloc(&out, 64));
    for (int i = 0; i < 1024; i++) host[i] = 0;
    uint32_t h[1024];
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 16; j++) h[l * 16 + j] = l * 16 + (j * 5 + 3) % 16; // a 16-cycle per lane
    CHECK(hipMemcpy(ring, h, 4096, hipMemcpyHostToDevice));
    const char *names[] = {"sleeps", "one HBM miss in flight", "eight HBM misses in flight", "returning atomics on one word", "vector loads of pinned host memory",
                           "scalar loads of pinned host memory", "vector stores to pinned host memory"};
    const int steps = 20000;
    for (int split = 0; split < 2; split++) {
        printf("%s\n", split ? "the other wave in ANOTHER workgroup (another CU):" : "the other wave in the SAME workgroup (same CU):");
        for (int mode = 0; mode < 7; mode++) {
            CHECK(hipMemset(stop, 0, 64));
            CHECK(hipMemset(out, 0, 64));
            hipLaunchKernelGGL(k, dim3(split ? 2 : 1), dim3(128), 0, 0, ring, big, big_words, word, host, mode, split, steps, out, stop);
            CHECK(hipDeviceSynchronize());
            unsigned long long o[4];
            CHECK(hipMemcpy(o, out, 32, hipMemcpyDeviceToHost));
            printf("  other wave %-38s: %7.1f cycles per dependent hit of the chaser (%llu operations of the other wave meanwhile, %.0f cycles each)\n", names[mode],
                   (double)o[0] / steps, o[2], o[2] ? (double)o[0] / (double)o[2] : 0.0);
        }
    }
    return 0;
}
