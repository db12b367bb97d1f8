// node_fetch.hip - what does it cost a wave to fetch 64 DIFFERENT 80-byte nodes, and does it matter who loads what?
//
// The incoherent passes (AO, explicit rays) fetch one node per lane and trip: five global_load_dwordx4 per lane, every lane
// in its own cache line(s) - 320 line look-ups per wave and node step.  TCP_TOTAL_CACHE_ACCESSES of the hairball-class AO
// pass (180 M per launch) is within 6 % of TA_TA_BUSY (190 M cycles): the texture path looks up about one line per
// cycle, whatever the lanes do with it.  If FIVE LANES fetch the five 16-byte pieces of ONE node (consecutive addresses,
// one or two lines), an instruction touches ~13 nodes = ~19 lines instead of 64, and the pieces change lanes through LDS
// (ds_write_b128 / ds_read_b128).  This program times both at the kernels' occupancy (16 waves per CU, one dependent
// fetch in flight per wave), with and without a block of arithmetic per step the size of the node test.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/node_fetch.hip -o /tmp/node_fetch && /tmp/node_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

constexpr int kWavesPerBlock = 2, kWave = 64;

__device__ __forceinline__ uint32_t next_index(uint32_t acc, uint32_t n_nodes) {
    uint32_t h = acc * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return (uint32_t)(((unsigned long long)h * n_nodes) >> 32);
}

// a stand-in for the node test: ALU dependent on the loaded words, `alu` rounds of 16 instructions
__device__ __forceinline__ uint32_t chew(uint4 n0, uint4 n1, uint4 n2, uint4 n3, uint4 n4, int alu) {
    float a = __uint_as_float((n0.x & 0x007fffffu) | 0x3f800000u), b = __uint_as_float((n1.y & 0x007fffffu) | 0x3f800000u);
    float c = __uint_as_float((n2.z & 0x007fffffu) | 0x3f800000u), d = __uint_as_float((n3.w & 0x007fffffu) | 0x3f800000u);
    for (int i = 0; i < alu; i++) {
        a = fmaxf(a * 1.0001f + b, c);
        b = fminf(b * 0.9999f + c, d);
        c = fmaxf(c * 1.0002f + d, a);
        d = fminf(d * 0.9998f + a, b);
    }
    return __float_as_uint(a) ^ __float_as_uint(b) ^ __float_as_uint(c) ^ __float_as_uint(d) ^ n0.y ^ n0.z ^ n0.w ^ n1.x ^ n1.z ^ n1.w ^
           n2.x ^ n2.y ^ n2.w ^ n3.x ^ n3.y ^ n3.z ^ n4.x ^ n4.y ^ n4.z ^ n4.w ^ n0.x ^ n1.y ^ n2.z ^ n3.w;
}

// MODE 0: every lane loads its own node (five loads).  MODE 1: five lanes load one node, pieces change lanes through LDS.
// MODE 2: as 1 in two halves of 32 nodes (half the LDS).
template <int MODE>
__global__ void __launch_bounds__(kWavesPerBlock * kWave) k_fetch(const uint4 *nodes, uint32_t n_nodes, int steps, int alu, uint32_t *out) {
    extern __shared__ uint4 lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4 *const box = lds + wave * (MODE == 2 ? 32 * 5 : 64 * 5);
    uint32_t acc = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t idx = next_index(acc, n_nodes);
    const uint32_t j = lane / 5u, c = lane - j * 5u; // this lane's node of an instruction's twelve, and its piece
    for (int s = 0; s < steps; s++) {
        uint4 n0, n1, n2, n3, n4;
        if (MODE == 0) {
            const uint4 *np = nodes + (size_t)idx * 5;
            n0 = np[0]; n1 = np[1]; n2 = np[2]; n3 = np[3]; n4 = np[4];
        } else if (MODE == 1) {
            uint4 p[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const uint32_t src = 12u * k + j; // the lane whose node this lane helps to fetch
                const uint32_t other = (uint32_t)__shfl((int)idx, (int)(src < 64u ? src : 63u));
                if (j < 12u && src < 64u) p[k] = nodes[(size_t)other * 5 + c];
            }
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const uint32_t src = 12u * k + j;
                if (j < 12u && src < 64u) box[src * 5u + c] = p[k];
            }
            __builtin_amdgcn_wave_barrier();
            n0 = box[lane * 5u]; n1 = box[lane * 5u + 1u]; n2 = box[lane * 5u + 2u]; n3 = box[lane * 5u + 3u]; n4 = box[lane * 5u + 4u];
            __builtin_amdgcn_wave_barrier();
        } else {
            uint4 p[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const uint32_t src = 12u * k + j;
                const uint32_t other = (uint32_t)__shfl((int)idx, (int)(src < 64u ? src : 63u));
                if (j < 12u && src < 64u) p[k] = nodes[(size_t)other * 5 + c];
            }
            // nodes 0..31 are in p[0], p[1] and the first eight groups of p[2]
#pragma unroll
            for (int half = 0; half < 2; half++) {
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const uint32_t src = 12u * k + j;
                    if (j < 12u && src < 64u && (src >> 5) == (uint32_t)half) box[(src & 31u) * 5u + c] = p[k];
                }
                __builtin_amdgcn_wave_barrier();
                if ((lane >> 5) == (uint32_t)half) {
                    const uint32_t l = lane & 31u;
                    n0 = box[l * 5u]; n1 = box[l * 5u + 1u]; n2 = box[l * 5u + 2u]; n3 = box[l * 5u + 3u]; n4 = box[l * 5u + 4u];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        acc ^= chew(n0, n1, n2, n3, n4, alu);
        idx = next_index(acc + (uint32_t)s, n_nodes);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
    const uint32_t n_nodes = argc > 1 ? (uint32_t)atoi(argv[1]) : 409943u; // bistro-class tree
    const int steps = 2000;
    std::vector<uint32_t> host((size_t)n_nodes * 20);
    uint32_t s = 1u;
    for (auto &v : host) { s = s * 1664525u + 1013904223u; v = s; }
    uint4 *nodes = nullptr;
    CHECK(hipMalloc(&nodes, host.size() * 4));
    CHECK(hipMemcpy(nodes, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 16 / kWavesPerBlock;
    uint32_t *out = nullptr;
    CHECK(hipMalloc(&out, (size_t)blocks * kWavesPerBlock * kWave * 4));
    std::vector<uint32_t> sum[3];
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%u nodes (%.1f MB), %d waves, %d dependent steps per wave; us per step of a wave, Gnodes/s chip-wide\n", n_nodes, n_nodes * 80e-6,
           blocks * kWavesPerBlock, steps);
    for (int alu : {0, 4, 8, 16}) {
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                const size_t lds1 = (size_t)kWavesPerBlock * 64 * 80, lds2 = lds1 / 2;
                // (every variant is given the kernels' LDS footprint at least, so that occupancy is 16 waves per CU for all three)
                const size_t pad = (size_t)kWavesPerBlock * 10176;
                if (mode == 0) k_fetch<0><<<blocks, kWavesPerBlock * kWave, pad>>>(nodes, n_nodes, steps, alu, out);
                if (mode == 1) k_fetch<1><<<blocks, kWavesPerBlock * kWave, pad > lds1 ? pad : lds1>>>(nodes, n_nodes, steps, alu, out);
                if (mode == 2) k_fetch<2><<<blocks, kWavesPerBlock * kWave, pad > lds2 ? pad : lds2>>>(nodes, n_nodes, steps, alu, out);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.0f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<uint32_t> got((size_t)blocks * kWavesPerBlock * kWave);
            CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
            unsigned long long chk = 0ull;
            for (uint32_t v : got) chk += v;
            printf("alu rounds %2d  mode %d (%s): %8.3f ms  %6.3f us/step  %7.2f Gnodes/s  checksum %llx\n", alu, mode,
                   mode == 0 ? "a node per lane      " : mode == 1 ? "five lanes per node  " : "five lanes, two halves", best, best * 1e3 / steps,
                   (double)blocks * kWavesPerBlock * kWave * steps / (best * 1e6), chk);
            fflush(stdout);
        }
    }
    return 0;
}
