// ploc_gpu.cpp — the BVH2 stage of the ploc_cwbvh build on the GPU (gfx950): Morton codes, radix sort and the PLOC
// merge rounds (Meister & Bittner 2018) as kernels.  Same algorithm, same tie-breaking and the same binary32 / binary64
// operations in the same order as PlocBuilder::run in builder.cpp (no contraction), so the tree it returns is THE tree
// the CPU stage returns, node for node (tests/test_gpu_builder.py compares them); the reinsertion pass, the 8-wide
// collapse and the encoder then run on the host as they do after the CPU stage.
//
// What it stands in for: the BVH2 build inside obvhs build_cwbvh_from_tris (src/cwbvh.rs:97), parameters
// src/main.rs:571-585 (ploc_search_distance, search_depth_threshold, sort_precision).
#include "ploc_gpu.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <limits>
#include <string>
#include <vector>

namespace trx {
namespace {

struct DevNode { // = Node2 of builder.cpp (40 bytes)
    Aabb box;
    uint32_t left, right, prim, count;
};
static_assert(sizeof(DevNode) == 40, "Node2 layout");

__device__ __forceinline__ float half_area_dev(const Aabb &b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}
__device__ __forceinline__ void grow_dev(Aabb &a, const Aabb &b) {
    for (int k = 0; k < 3; k++) {
        a.mn[k] = fminf(a.mn[k], b.mn[k]);
        a.mx[k] = fmaxf(a.mx[k], b.mx[k]);
    }
}

__device__ __forceinline__ uint64_t spread21_dev(uint64_t x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

struct MortonParams {
    double lo[3], scale[3];
};

// bits == 21: one 63-bit key in key_lo.  bits == 42: the 126-bit key as (key_hi << 64) | key_lo, where the spread of
// the low 21 bits of each axis fills bits 0..62 and the spread of the high 21 bits starts at bit 63 (builder.cpp spread42)
__global__ void k_morton(const float *cen, uint32_t n, MortonParams mp, int bits, uint64_t *key_lo, uint64_t *key_hi,
                         uint32_t *index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t q[3];
    for (int k = 0; k < 3; k++) {
        const double v = ((double)cen[3 * (size_t)i + k] - mp.lo[k]) * mp.scale[k];
        q[k] = v <= 0.0 ? 0ull : (uint64_t)v;
    }
    index[i] = i;
    if (bits == 21) {
        key_lo[i] = spread21_dev(q[0]) | (spread21_dev(q[1]) << 1) | (spread21_dev(q[2]) << 2);
    } else {
        // 126-bit value: low part L (63 bits used, bit 63 and up belong to the high spread) and high part H
        const uint64_t l = spread21_dev(q[0] & 0x1fffffull) | (spread21_dev(q[1] & 0x1fffffull) << 1) | (spread21_dev(q[2] & 0x1fffffull) << 2);
        const uint64_t h = spread21_dev(q[0] >> 21) | (spread21_dev(q[1] >> 21) << 1) | (spread21_dev(q[2] >> 21) << 2); // << 63 overall
        key_lo[i] = l | (h << 63);
        key_hi[i] = h >> 1;
    }
}

__global__ void k_gather_keys(const uint64_t *src, const uint32_t *index, uint32_t n, uint64_t *dst) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[index[i]];
}

__global__ void k_leaves(const Aabb *boxes, const uint32_t *order, uint32_t n, DevNode *nodes, uint32_t *cluster, Aabb *cbox) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DevNode leaf;
    leaf.box = boxes[order[i]];
    leaf.left = leaf.right = 0;
    leaf.prim = order[i];
    leaf.count = 1;
    nodes[i] = leaf;
    cluster[i] = i;
    cbox[i] = leaf.box;
}

__global__ void k_nearest(const Aabb *cbox, uint32_t m, uint32_t r, uint32_t *nn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const Aabb bi = cbox[i];
    const uint32_t j0 = i > r ? i - r : 0u, j1 = min(m - 1u, i + r);
    float best = std::numeric_limits<float>::infinity();
    uint32_t best_j = i == 0 ? 1u : i - 1u;
    for (uint32_t j = j0; j <= j1; j++) {
        if (j == i) continue;
        Aabb u = bi;
        grow_dev(u, cbox[j]);
        const float a = half_area_dev(u);
        if (a < best) { // first of equals: the lowest index
            best = a;
            best_j = j;
        }
    }
    nn[i] = best_j;
}

// per cluster: low word = it survives into the next round (kept, or the first of a merging pair), high word = it
// starts a merge (a new node is created); an exclusive scan of the packed words gives both positions at once
__global__ void k_flags(const uint32_t *nn, uint32_t m, uint64_t *flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t j = nn[i];
    const bool mutual = nn[j] == i;
    const uint64_t keep = (!mutual || i < j) ? 1ull : 0ull;
    const uint64_t merge = (mutual && i < j) ? 1ull : 0ull;
    flags[i] = keep | (merge << 32);
}

__global__ void k_apply(const uint32_t *nn, const uint64_t *flags, const uint64_t *scan, uint32_t m, uint32_t next_node,
                        const uint32_t *cluster, const Aabb *cbox, DevNode *nodes, uint32_t *cluster_out, Aabb *cbox_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t f = flags[i];
    if (!(f & 1ull)) return; // the second of a merging pair disappears
    const uint32_t pos = (uint32_t)(scan[i] & 0xffffffffull);
    if (f >> 32) {
        const uint32_t j = nn[i];
        DevNode p;
        p.left = cluster[i];
        p.right = cluster[j];
        p.box = cbox[i];
        grow_dev(p.box, cbox[j]);
        p.prim = 0;
        p.count = 0; // filled on the host (children precede their parent in creation order)
        const uint32_t id = next_node + (uint32_t)(scan[i] >> 32);
        nodes[id] = p;
        cluster_out[pos] = id;
        cbox_out[pos] = p.box;
    } else {
        cluster_out[pos] = cluster[i];
        cbox_out[pos] = cbox[i];
    }
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T>
    T *as() { return static_cast<T *>(p); }
};

#define PG_TRY(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
            return false;                                                                        \
        }                                                                                        \
    } while (0)

} // namespace

bool ploc_bvh2_device(int device, const Aabb *boxes, const float *centroids, uint32_t n, uint32_t radius,
                      uint32_t depth_threshold, uint32_t sort_bits, void *nodes_out, uint32_t *root_out, double *seconds,
                      std::string &err) {
    if (n < 2) {
        err = "ploc_bvh2_device needs at least two primitives";
        return false;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        err = "no HIP device " + std::to_string(device) + " for the GPU build stage";
        return false;
    }
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    struct DeviceGuard { // the caller's current device is put back whatever happens below
        int prev;
        ~DeviceGuard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } device_guard{prev_device};
    PG_TRY(hipSetDevice(device));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    PG_TRY(hipEventCreate(&ev0));
    PG_TRY(hipEventCreate(&ev1));
    struct EvGuard {
        hipEvent_t a, b;
        ~EvGuard() {
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
        }
    } evg{ev0, ev1};

    // centroid bounds on the host, exactly as PlocBuilder::run computes them
    MortonParams mp;
    float lo[3] = {std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity()};
    float hi[3] = {-lo[0], -lo[1], -lo[2]};
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], centroids[3 * (size_t)i + k]);
            hi[k] = std::max(hi[k], centroids[3 * (size_t)i + k]);
        }
    const int bits = sort_bits == 128 ? 42 : 21;
    for (int k = 0; k < 3; k++) {
        const double ext = (double)hi[k] - (double)lo[k];
        mp.lo[k] = (double)lo[k];
        mp.scale[k] = ext > 0.0 ? ((double)((1ull << bits) - 1ull)) / ext : 0.0;
    }

    const size_t total = 2 * (size_t)n - 1;
    DevBuf d_boxes, d_cen, d_nodes, d_key_a, d_key_b, d_key_hi, d_idx_a, d_idx_b, d_cluster_a, d_cluster_b, d_cbox_a, d_cbox_b, d_nn,
        d_flags, d_scan, d_tmp;
    PG_TRY(d_boxes.alloc((size_t)n * sizeof(Aabb)));
    PG_TRY(d_cen.alloc((size_t)n * 12));
    PG_TRY(d_nodes.alloc(total * sizeof(DevNode)));
    PG_TRY(d_key_a.alloc((size_t)n * 8));
    PG_TRY(d_key_b.alloc((size_t)n * 8));
    PG_TRY(d_key_hi.alloc((size_t)n * 8));
    PG_TRY(d_idx_a.alloc((size_t)n * 4));
    PG_TRY(d_idx_b.alloc((size_t)n * 4));
    PG_TRY(d_cluster_a.alloc((size_t)n * 4));
    PG_TRY(d_cluster_b.alloc((size_t)n * 4));
    PG_TRY(d_cbox_a.alloc((size_t)n * sizeof(Aabb)));
    PG_TRY(d_cbox_b.alloc((size_t)n * sizeof(Aabb)));
    PG_TRY(d_nn.alloc((size_t)n * 4));
    PG_TRY(d_flags.alloc((size_t)n * 8));
    PG_TRY(d_scan.alloc((size_t)n * 8));
    PG_TRY(hipMemcpy(d_boxes.p, boxes, (size_t)n * sizeof(Aabb), hipMemcpyHostToDevice));
    PG_TRY(hipMemcpy(d_cen.p, centroids, (size_t)n * 12, hipMemcpyHostToDevice));

    size_t tmp_sort = 0, tmp_scan = 0;
    PG_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, d_key_a.as<uint64_t>(), d_key_b.as<uint64_t>(), d_idx_a.as<uint32_t>(),
                                              d_idx_b.as<uint32_t>(), (int)n, 0, 64, nullptr));
    PG_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, d_flags.as<uint64_t>(), d_scan.as<uint64_t>(), (int)n, nullptr));
    size_t tmp_bytes = std::max(tmp_sort, tmp_scan);
    PG_TRY(d_tmp.alloc(tmp_bytes));

    const uint32_t block = 256;
    auto grid = [&](uint32_t items) { return dim3((items + block - 1) / block); };
    PG_TRY(hipEventRecord(ev0, nullptr));
    // Morton order (stable LSD radix sort; 128-bit codes as two stable passes, low word first)
    hipLaunchKernelGGL(k_morton, grid(n), dim3(block), 0, nullptr, d_cen.as<float>(), n, mp, bits, d_key_a.as<uint64_t>(),
                       d_key_hi.as<uint64_t>(), d_idx_a.as<uint32_t>());
    size_t tb = tmp_bytes;
    PG_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tb, d_key_a.as<uint64_t>(), d_key_b.as<uint64_t>(), d_idx_a.as<uint32_t>(),
                                              d_idx_b.as<uint32_t>(), (int)n, 0, 64, nullptr));
    uint32_t *order = d_idx_b.as<uint32_t>();
    if (bits == 42) {
        hipLaunchKernelGGL(k_gather_keys, grid(n), dim3(block), 0, nullptr, d_key_hi.as<uint64_t>(), d_idx_b.as<uint32_t>(), n,
                           d_key_a.as<uint64_t>());
        tb = tmp_bytes;
        PG_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tb, d_key_a.as<uint64_t>(), d_key_b.as<uint64_t>(), d_idx_b.as<uint32_t>(),
                                                  d_idx_a.as<uint32_t>(), (int)n, 0, 64, nullptr));
        order = d_idx_a.as<uint32_t>();
    }
    hipLaunchKernelGGL(k_leaves, grid(n), dim3(block), 0, nullptr, d_boxes.as<Aabb>(), order, n, d_nodes.as<DevNode>(),
                       d_cluster_a.as<uint32_t>(), d_cbox_a.as<Aabb>());
    PG_TRY(hipGetLastError());

    uint32_t *cluster = d_cluster_a.as<uint32_t>(), *cluster_next = d_cluster_b.as<uint32_t>();
    Aabb *cbox = d_cbox_a.as<Aabb>(), *cbox_next = d_cbox_b.as<Aabb>();
    uint32_t m = n, next_node = n;
    for (uint32_t round = 0; m > 1; round++) {
        const uint32_t r = round < depth_threshold ? 1u : std::max(1u, radius);
        hipLaunchKernelGGL(k_nearest, grid(m), dim3(block), 0, nullptr, cbox, m, r, d_nn.as<uint32_t>());
        hipLaunchKernelGGL(k_flags, grid(m), dim3(block), 0, nullptr, d_nn.as<uint32_t>(), m, d_flags.as<uint64_t>());
        tb = tmp_bytes;
        PG_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, d_flags.as<uint64_t>(), d_scan.as<uint64_t>(), (int)m, nullptr));
        hipLaunchKernelGGL(k_apply, grid(m), dim3(block), 0, nullptr, d_nn.as<uint32_t>(), d_flags.as<uint64_t>(), d_scan.as<uint64_t>(), m,
                           next_node, cluster, cbox, d_nodes.as<DevNode>(), cluster_next, cbox_next);
        PG_TRY(hipGetLastError());
        uint64_t last_scan = 0, last_flag = 0;
        PG_TRY(hipMemcpy(&last_scan, d_scan.as<uint64_t>() + (m - 1), 8, hipMemcpyDeviceToHost));
        PG_TRY(hipMemcpy(&last_flag, d_flags.as<uint64_t>() + (m - 1), 8, hipMemcpyDeviceToHost));
        const uint64_t sum = last_scan + last_flag;
        const uint32_t m_new = (uint32_t)(sum & 0xffffffffull), merges = (uint32_t)(sum >> 32);
        if (merges == 0 || m_new >= m || (size_t)next_node + merges > total) {
            err = "PLOC round made no progress (device)";
            return false;
        }
        next_node += merges;
        m = m_new;
        std::swap(cluster, cluster_next);
        std::swap(cbox, cbox_next);
    }
    PG_TRY(hipEventRecord(ev1, nullptr));
    uint32_t root = 0;
    PG_TRY(hipMemcpy(&root, cluster, 4, hipMemcpyDeviceToHost));
    PG_TRY(hipMemcpy(nodes_out, d_nodes.p, total * sizeof(DevNode), hipMemcpyDeviceToHost));
    PG_TRY(hipEventSynchronize(ev1));
    float ms = 0.f;
    PG_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    if (seconds) *seconds = ms * 1e-3;
    if (next_node != total || root != total - 1) {
        err = "PLOC (device) ended with " + std::to_string(next_node) + " nodes, root " + std::to_string(root);
        return false;
    }
    *root_out = root;
    return true;
}

} // namespace trx
