// probe.cpp - trx_debug_fetch_rate (include/trx_dev.h): the rate at which this GPU serves scattered fetches of a scene's own
// nodes and triangle records - the measured yardstick bench.py holds the node-fetch loop of the incoherent passes against.
//
// north_star asks for the node-fetch loop against a MEASURED roofline.  For incoherent rays that is not the HBM copy
// rate: a lane fetches an 80-byte node (five 16-byte loads, one or two 128-byte lines) wherever its ray went, and what
// bounds that is the rate at which the memory system serves scattered lines from the L2s and the Infinity Cache.  This
// kernel measures exactly that, with nothing else in the way: the tracer's persistent grid at the tracer's occupancy
// (two waves to a workgroup, the tracer's LDS footprint: 16 waves per CU), every lane fetching one UNIFORMLY RANDOM node
// of this scene per step - and, tris_x256 / 256 times per step on average, one random 48-byte triangle record - the
// next index a hash of what arrived, so each wave has one step in flight like a traversing wave; the triangle request
// of a step does not wait for its node (the pipelined walk's overlap, the most favourable form).  Measured (MI355X,
// bistro-class 220 MB scene): 39.7 G nodes/s alone, 28 G nodes/s + 15 G triangles/s together = 3.0 TB/s.  It is a
// no-locality rate, not an upper bound: a walk's upper levels stay in L1 / L2 and a leaf's triangles share lines, so
// the incoherent passes ask for 1.25 - 1.95 x as much per second (bench.py, `fetch_vs_random`) - they run on their caches,
// beyond what the memory system delivers to scattered requests.
#include "api_internal.h"

namespace {

__device__ __forceinline__ uint32_t probe_index(uint32_t acc, uint32_t n) {
    uint32_t h = acc * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    h ^= h >> 13;
    return (uint32_t)(((unsigned long long)h * n) >> 32);
}

__global__ void __launch_bounds__(2 * kWave) k_fetch_probe(const uint4 *nodes, uint32_t n_nodes, const float4 *tris, uint32_t n_tris,
                                                            uint32_t steps, uint32_t tris_x256, uint32_t *sink) {
    uint32_t acc = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t idx = probe_index(acc, n_nodes);
    for (uint32_t s = 0; s < steps; s++) {
        const uint4 *np = nodes + (size_t)idx * 5;
        const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        const uint32_t t_count = (((s + 1u) * tris_x256) >> 8) - ((s * tris_x256) >> 8); // wave-uniform
        uint32_t tacc = 0u;
        for (uint32_t k = 0; k < t_count; k++) {
            const float4 *tp = tris + (size_t)probe_index(acc + 0x51ed27u * (k + 1u), n_tris) * 3;
            const float4 a = tp[0], b = tp[1], c = tp[2];
            tacc ^= __float_as_uint(a.x) ^ __float_as_uint(a.w) ^ __float_as_uint(b.y) ^ __float_as_uint(b.w) ^ __float_as_uint(c.z) ^ __float_as_uint(c.w);
        }
        acc ^= tacc ^ n0.x ^ n0.w ^ n1.y ^ n1.z ^ n2.x ^ n2.w ^ n3.y ^ n3.z ^ n4.x ^ n4.w;
        idx = probe_index(acc + s, n_nodes);
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// The streaming ceiling: float4 in, float4 out, grid-stride over `n` float4 - the copy kernel the platform guide measures
// (MI355X_MICROARCH.md: 6.29 TB/s read + written); hipMemcpyDtoD, which bench.py used until round 5, reaches about 5.0 TB/s
// on the same box.  U independent float4 per lane and trip (bytes in flight per lane), NT = non-temporal loads and stores;
// trx_debug_copy_rate reports the fastest shape, as a ceiling should.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy_probe(const float4 *__restrict__ src4, float4 *__restrict__ dst4, size_t n) {
    const f32x4_t *__restrict__ src = reinterpret_cast<const f32x4_t *>(src4);
    f32x4_t *__restrict__ dst = reinterpret_cast<f32x4_t *>(dst4);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (size_t)(U - 1) * stride < n; i += (size_t)U * stride) {
        f32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + (size_t)u * stride) : src[i + (size_t)u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + (size_t)u * stride);
            else dst[i + (size_t)u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

} // namespace

extern "C" int trx_debug_copy_rate(int device, uint64_t bytes, uint32_t reps, double *out_bytes_per_s) {
    if (!out_bytes_per_s || reps == 0 || reps > 64) return fail(TRX_ERR_INVALID, "reps in 1 .. 64");
    if (bytes < (1ull << 20) || bytes > (8ull << 30) || (bytes & 15u)) return fail(TRX_ERR_INVALID, "bytes: a multiple of 16 in 1 MiB .. 8 GiB");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return fail(TRX_ERR_NO_DEVICE, "no HIP device %d", device);
    HIP_TRY(hipSetDevice(device));
    float4 *src = nullptr, *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&src, bytes);
    if (e == hipSuccess) e = hipMalloc(&dst, bytes);
    if (e == hipSuccess) e = hipMemsetAsync(src, 1, bytes, nullptr);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const size_t n = bytes / sizeof(float4);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    float best = 0.0f;
    if (e == hipSuccess) {
        typedef void (*copy_kernel)(const float4 *, float4 *, size_t);
        const copy_kernel shapes[6] = {k_copy_probe<1, false>, k_copy_probe<4, false>, k_copy_probe<8, false>,
                                       k_copy_probe<1, true>,  k_copy_probe<4, true>,  k_copy_probe<8, true>};
        for (int shape = 0; shape < 6 && e == hipSuccess; shape++) {
            for (int per_cu = 8; per_cu <= 32 && e == hipSuccess; per_cu *= 2) { // (2 048 .. 8 192 lanes per CU in flight)
                const int blocks = prop.multiProcessorCount * per_cu;
                for (uint32_t rep = 0; rep < reps + 1u && e == hipSuccess; rep++) { // (one warm-up pass per shape)
                    e = hipEventRecord(e0, nullptr);
                    if (e == hipSuccess) {
                        hipLaunchKernelGGL(shapes[shape], dim3(blocks), dim3(256), 0, nullptr, src, dst, n);
                        e = hipGetLastError();
                    }
                    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
                    if (e == hipSuccess) e = hipEventSynchronize(e1);
                    float ms = 0.0f;
                    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
                    if (e == hipSuccess && rep >= 1u && (best == 0.0f || ms < best)) best = ms;
                }
            }
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (e != hipSuccess || best <= 0.0f) return fail(e == hipErrorOutOfMemory ? TRX_ERR_OOM : TRX_ERR_NO_DEVICE, "copy probe failed: %s", hipGetErrorString(e));
    *out_bytes_per_s = 2.0 * (double)bytes / (best * 1e-3); // read + written
    return TRX_OK;
}

extern "C" int trx_debug_fetch_rate(trx_scene *s, uint32_t steps, uint32_t tris_per_node_x256, double *out_nodes_per_s,
                                    double *out_tris_per_s) {
    if (!s || !out_nodes_per_s || !out_tris_per_s) return fail(TRX_ERR_INVALID, "null argument");
    if (steps == 0 || steps > (1u << 20) || tris_per_node_x256 > 16u * 256u) return fail(TRX_ERR_INVALID, "steps in 1 .. 2^20, at most 16 triangles per node");
    if (s->n_nodes == 0 || s->n_tris == 0) return fail(TRX_ERR_INVALID, "empty scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the scene's event pair
    HIP_TRY(hipSetDevice(s->device));
    const int waves = s->grid > 0 ? s->grid : 4096, blocks = (waves + 1) / 2;
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc(&sink, (size_t)blocks * 2 * kWave * sizeof(uint32_t)));
    float best = 0.0f;
    hipError_t e = hipSuccess;
    for (int rep = 0; rep < 3 && e == hipSuccess; rep++) { // (the first repetition warms clocks and caches)
        e = hipEventRecord(s->ev0, nullptr);
        if (e == hipSuccess) {
            k_fetch_probe<<<blocks, 2 * kWave, 2 * kLdsBytesPerWave, nullptr>>>(s->d_nodes, (uint32_t)s->n_nodes, s->d_tris, (uint32_t)s->n_tris,
                                                                                 steps, tris_per_node_x256, sink);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(s->ev1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(s->ev1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, s->ev0, s->ev1);
        if (e == hipSuccess && rep > 0 && (best == 0.0f || ms < best)) best = ms;
    }
    (void)hipFree(sink);
    if (e != hipSuccess || best <= 0.0f) return fail(TRX_ERR_NO_DEVICE, "fetch probe failed: %s", hipGetErrorString(e));
    const double lanes = (double)blocks * 2 * kWave, secs = best * 1e-3;
    *out_nodes_per_s = lanes * steps / secs;
    *out_tris_per_s = lanes * (double)(((unsigned long long)steps * tris_per_node_x256) >> 8) / secs;
    return TRX_OK;
}
