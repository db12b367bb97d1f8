// reinsert_gpu.cpp — the searches of the BVH2 reinsertion pass (Meister & Bittner 2018, "Parallel Reinsertion for
// Bounding Volume Hierarchy Optimization"; the pass obvhs runs after PLOC, knob reinsertion_batch_ratio behind the
// reference's -r, src/main.rs:113-118) on the GPU: one thread per candidate, every candidate of a batch against the same
// tree.  A search is a branch-and-bound walk of a few hundred dependent node reads - latency on a CPU core (2-3 s of a
// reference-default build of a 3.9 M triangle scene on 16 cores), throughput here.  Same search order, same tie rule, same
// binary32 operations without contraction as Reinserter::find in builder.cpp, so found[] is the host's found[], candidate
// for candidate (tests/test_gpu_builder.py); the moves are then applied on the host, in candidate order, as they are
// after host searches.
#include "reinsert_gpu.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "cwbvh_format.h"

namespace trx {
namespace {

struct DevNode { // = Node2 of builder.cpp (40 bytes)
    Aabb box;
    uint32_t left, right, prim, count;
};
static_assert(sizeof(DevNode) == 40, "Node2 layout");

constexpr uint32_t kStackCap = 128; // entries per search (a DFS that pushes two and pops one per level: depth + 1)
constexpr int kBlock = 256;

__device__ __forceinline__ float half_area_dev(const Aabb &b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}
// std::min / std::max as builder.cpp's grow() writes them (a if !(b < a), a if !(a < b)): same result for every input
__device__ __forceinline__ void grow_dev(Aabb &a, const Aabb &b) {
    for (int k = 0; k < 3; k++) {
        a.mn[k] = b.mn[k] < a.mn[k] ? b.mn[k] : a.mn[k];
        a.mx[k] = a.mx[k] < b.mx[k] ? b.mx[k] : a.mx[k];
    }
}

__device__ __forceinline__ DevNode load_node(const DevNode *nodes, uint32_t i) {
    // 40 bytes, 8-byte aligned: five 8-byte loads
    const uint2 *p = reinterpret_cast<const uint2 *>(nodes + i);
    const uint2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    DevNode n;
    n.box.mn[0] = __uint_as_float(a.x); n.box.mn[1] = __uint_as_float(a.y); n.box.mn[2] = __uint_as_float(b.x);
    n.box.mx[0] = __uint_as_float(b.y); n.box.mx[1] = __uint_as_float(c.x); n.box.mx[2] = __uint_as_float(c.y);
    n.left = d.x; n.right = d.y; n.prim = e.x; n.count = e.y;
    return n;
}

// Reinserter::search: best place below `top` for a box of area `area`; `gain` is what the tree has saved so far by
// taking the node out.  The stack holds {gain, node}; an entry that cannot win is not even pushed (the host prunes it
// when it pops it: best_gain only grows in between, so the same entries are dropped, a little earlier).
__device__ __forceinline__ bool search_dev(const DevNode *nodes, uint2 *stk, uint32_t stride, uint32_t top, float gain,
                                           const Aabb &box, float area, uint32_t &best_to, float &best_gain) {
    uint32_t sp = 0;
    float g = gain;
    uint32_t id = top;
    bool have = true;
    for (;;) {
        if (!have) {
            if (sp == 0) return true;
            sp--;
            const uint2 e = stk[(size_t)sp * stride];
            g = __uint_as_float(e.x);
            id = e.y;
        }
        have = false;
        if (g - area <= best_gain) continue; // even a zero-growth insertion cannot win
        const DevNode dst = load_node(nodes, id);
        Aabb merged = dst.box;
        grow_dev(merged, box);
        const float here = g - half_area_dev(merged); // new inner node holding {dst, node}
        if (here > best_gain) {
            best_gain = here;
            best_to = id;
        }
        if (dst.count > 1) {
            // going below dst instead grows dst to `merged`
            const float below = here + half_area_dev(dst.box);
            if (below - area <= best_gain) continue;
            // the host pushes left, then right, and pops right first: right is walked now, left waits
            if (sp + 1 > kStackCap) return false;
            stk[(size_t)sp * stride] = make_uint2(__float_as_uint(below), dst.left);
            sp++;
            g = below;
            id = dst.right;
            have = true;
        }
    }
}

__global__ void __launch_bounds__(kBlock) k_find(const DevNode *nodes, const uint32_t *parent, const uint32_t *cand, uint32_t n_cand,
                                                 uint32_t *found, uint2 *stacks) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    uint2 *const stk = stacks + tid;
    for (uint32_t k = tid; k < n_cand; k += stride) {
        const uint32_t from = cand[k];
        const uint32_t p = parent[from];
        if (p == 0u || p == kReinsertNone) { // the root's children stay (as on the host)
            found[k] = kReinsertNone;
            continue;
        }
        const DevNode self = load_node(nodes, from);
        const Aabb box = self.box;
        const float area = half_area_dev(box);
        DevNode pn = load_node(nodes, p);
        float gain = half_area_dev(pn.box); // p disappears
        float best_gain = 0.f;
        uint32_t best_to = kReinsertNone;
        uint32_t sib = pn.left == from ? pn.right : pn.left;
        bool ok = search_dev(nodes, stk, stride, sib, gain, box, area, best_to, best_gain);
        Aabb shrunk = load_node(nodes, sib).box; // what the path node looks like without `from`
        uint32_t cur = p;
        while (ok) {
            const uint32_t up = parent[cur];
            if (up == kReinsertNone) break;
            const DevNode un = load_node(nodes, up);
            sib = un.left == cur ? un.right : un.left;
            ok = search_dev(nodes, stk, stride, sib, gain, box, area, best_to, best_gain);
            grow_dev(shrunk, load_node(nodes, sib).box);
            gain += half_area_dev(un.box) - half_area_dev(shrunk);
            cur = up;
        }
        found[k] = ok ? best_to : kReinsertOverflow;
    }
}

// Candidate keys: {area bits, ~index} - descending key order is area descending, index ascending (areas are >= 0, so their
// bit patterns order like the floats); the root, its children and anything without a parent get key 0 and sort last.
__global__ void __launch_bounds__(kBlock) k_keys(const DevNode *nodes, const uint32_t *parent, uint32_t n, unsigned long long *keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long key = 0ull;
    if (i != 0u && parent[i] != 0u && parent[i] != kReinsertNone) {
        const DevNode nd = load_node(nodes, i);
        key = ((unsigned long long)__float_as_uint(half_area_dev(nd.box)) << 32) | (unsigned long long)(~i);
    }
    keys[i] = key;
}
__global__ void __launch_bounds__(kBlock) k_ids(const unsigned long long *keys, uint32_t take, uint32_t *ids) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < take) ids[k] = ~(uint32_t)keys[k];
}

} // namespace

struct ReinsertDevice {
    unsigned long long *d_keys_a = nullptr, *d_keys_b = nullptr;
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int device = -1;
    size_t n_nodes = 0;
    DevNode *d_nodes = nullptr;
    uint32_t *d_parent = nullptr, *d_cand = nullptr, *d_found = nullptr;
    uint2 *d_stacks = nullptr;
    int grid = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define RG_TRY(expr)                                                     \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);     \
            return false;                                                \
        }                                                                \
    } while (0)

void reinsert_dev_close(ReinsertDevice *c) {
    if (!c) return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(c->device) == hipSuccess) {
        if (c->d_nodes) (void)hipFree(c->d_nodes);
        if (c->d_parent) (void)hipFree(c->d_parent);
        if (c->d_cand) (void)hipFree(c->d_cand);
        if (c->d_found) (void)hipFree(c->d_found);
        if (c->d_stacks) (void)hipFree(c->d_stacks);
        if (c->d_keys_a) (void)hipFree(c->d_keys_a);
        if (c->d_keys_b) (void)hipFree(c->d_keys_b);
        if (c->d_sort_tmp) (void)hipFree(c->d_sort_tmp);
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    delete c;
}

bool reinsert_dev_open(int device, size_t n_nodes, ReinsertDevice **out, std::string &err) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        err = "no HIP device " + std::to_string(device) + " for the GPU build stage";
        return false;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Guard {
        int prev;
        ~Guard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } guard{prev};
    ReinsertDevice *c = new ReinsertDevice;
    c->device = device;
    c->n_nodes = n_nodes;
    struct Closer { // (whatever fails below, nothing leaks)
        ReinsertDevice *c;
        ~Closer() {
            if (c) reinsert_dev_close(c);
        }
    } closer{c};
    RG_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    RG_TRY(hipGetDeviceProperties(&prop, device));
    c->grid = std::max(1, prop.multiProcessorCount) * 4; // four 256-thread blocks per CU: ~260 k searches in flight
    RG_TRY(hipMalloc(&c->d_nodes, std::max<size_t>(n_nodes, 1) * sizeof(DevNode)));
    RG_TRY(hipMalloc(&c->d_parent, std::max<size_t>(n_nodes, 1) * 4));
    RG_TRY(hipMalloc(&c->d_cand, std::max<size_t>(n_nodes, 1) * 4));
    RG_TRY(hipMalloc(&c->d_found, std::max<size_t>(n_nodes, 1) * 4));
    RG_TRY(hipMalloc(&c->d_stacks, (size_t)c->grid * kBlock * kStackCap * sizeof(uint2)));
    RG_TRY(hipEventCreate(&c->ev0));
    RG_TRY(hipEventCreate(&c->ev1));
    closer.c = nullptr;
    *out = c;
    return true;
}

bool reinsert_dev_search(ReinsertDevice *c, const void *nodes, const uint32_t *parent, const uint32_t *cand, uint32_t n_cand,
                         uint32_t *found, double *seconds, std::string &err) {
    if (!c || n_cand > c->n_nodes) {
        err = "reinsert_dev_search: bad arguments";
        return false;
    }
    if (n_cand == 0) return true;
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Guard {
        int prev;
        ~Guard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } guard{prev};
    RG_TRY(hipSetDevice(c->device));
    RG_TRY(hipMemcpy(c->d_nodes, nodes, c->n_nodes * sizeof(DevNode), hipMemcpyHostToDevice));
    RG_TRY(hipMemcpy(c->d_parent, parent, c->n_nodes * 4, hipMemcpyHostToDevice));
    RG_TRY(hipMemcpy(c->d_cand, cand, (size_t)n_cand * 4, hipMemcpyHostToDevice));
    RG_TRY(hipEventRecord(c->ev0, nullptr));
    const int blocks = (int)std::min<size_t>((size_t)c->grid, ((size_t)n_cand + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_find, dim3(blocks), dim3(kBlock), 0, nullptr, c->d_nodes, c->d_parent, c->d_cand, n_cand, c->d_found,
                       c->d_stacks);
    RG_TRY(hipGetLastError());
    RG_TRY(hipEventRecord(c->ev1, nullptr));
    RG_TRY(hipMemcpy(found, c->d_found, (size_t)n_cand * 4, hipMemcpyDeviceToHost));
    if (seconds) {
        float ms = 0.f;
        RG_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *seconds += ms * 1e-3;
    }
    return true;
}

} // namespace trx

namespace trx {

bool reinsert_dev_iteration(ReinsertDevice *c, const void *nodes, const uint32_t *parent, uint32_t take, uint32_t *ids,
                            uint32_t *found, double *seconds, std::string &err) {
    if (!c || take > c->n_nodes || c->n_nodes > 0x7fffffffull) {
        err = "reinsert_dev_iteration: bad arguments";
        return false;
    }
    if (take == 0) return true;
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Guard {
        int prev;
        ~Guard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } guard{prev};
    RG_TRY(hipSetDevice(c->device));
    const uint32_t n = (uint32_t)c->n_nodes;
    if (!c->d_keys_a) {
        RG_TRY(hipMalloc(&c->d_keys_a, (size_t)n * 8));
        RG_TRY(hipMalloc(&c->d_keys_b, (size_t)n * 8));
        size_t bytes = 0;
        RG_TRY(hipcub::DeviceRadixSort::SortKeysDescending(nullptr, bytes, c->d_keys_a, c->d_keys_b, (int)n, 0, 64, (hipStream_t) nullptr));
        RG_TRY(hipMalloc(&c->d_sort_tmp, bytes ? bytes : 16));
        c->sort_tmp_bytes = bytes;
    }
    RG_TRY(hipMemcpy(c->d_nodes, nodes, c->n_nodes * sizeof(DevNode), hipMemcpyHostToDevice));
    RG_TRY(hipMemcpy(c->d_parent, parent, c->n_nodes * 4, hipMemcpyHostToDevice));
    RG_TRY(hipEventRecord(c->ev0, nullptr));
    hipLaunchKernelGGL(k_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, c->d_nodes, c->d_parent, n, c->d_keys_a);
    RG_TRY(hipGetLastError());
    size_t bytes = c->sort_tmp_bytes;
    RG_TRY(hipcub::DeviceRadixSort::SortKeysDescending(c->d_sort_tmp, bytes, c->d_keys_a, c->d_keys_b, (int)n, 0, 64, (hipStream_t) nullptr));
    hipLaunchKernelGGL(k_ids, dim3((take + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, c->d_keys_b, take, c->d_cand);
    RG_TRY(hipGetLastError());
    const int blocks = (int)std::min<size_t>((size_t)c->grid, ((size_t)take + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_find, dim3(blocks), dim3(kBlock), 0, nullptr, c->d_nodes, c->d_parent, c->d_cand, take, c->d_found, c->d_stacks);
    RG_TRY(hipGetLastError());
    RG_TRY(hipEventRecord(c->ev1, nullptr));
    RG_TRY(hipMemcpy(ids, c->d_cand, (size_t)take * 4, hipMemcpyDeviceToHost));
    RG_TRY(hipMemcpy(found, c->d_found, (size_t)take * 4, hipMemcpyDeviceToHost));
    if (seconds) {
        float ms = 0.f;
        RG_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *seconds += ms * 1e-3;
    }
    return true;
}

} // namespace trx
