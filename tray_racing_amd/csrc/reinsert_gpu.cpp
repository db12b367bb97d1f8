// reinsert_gpu.cpp — the BVH2 reinsertion pass with one batch per iteration (Meister & Bittner 2018, "Parallel Reinsertion for
// Bounding Volume Hierarchy Optimization"; the pass obvhs runs after PLOC, knob reinsertion_batch_ratio behind the
// reference's -r, src/main.rs:113-118) on the GPU: one thread per candidate, every candidate of a batch against the same
// tree.  A search is a branch-and-bound walk of a few hundred dependent node reads - latency on a CPU core (2-3 s of a
// reference-default build of a 3.9 M triangle scene on 16 cores), throughput here.  Same search order, same tie rule, same
// binary32 operations without contraction as Reinserter::find in builder.cpp, so found[] is the host's found[], candidate
// for candidate (tests/test_gpu_builder.py).  The moves are applied here too (further down): the tree stays on the device
// from the first iteration to the last.
#include "reinsert_gpu.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

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

// ---- the moves of a whole iteration, applied on the device -------------------------------------------------------------
// Reinserter::apply_batch takes the candidates in order: a move is skipped when an earlier APPLIED move re-linked one of
// its six nodes, or when its target has meanwhile come to lie below the node it moves.  Without the second rule that is a
// greedy independent set in candidate order, which has one fixed point whatever order it is computed in: rounds of "claim
// my nodes with my index (atomic min); mine on all six -> accepted; lost one to an accepted move -> rejected".  The second
// rule is then CHECKED for the accepted moves, each against the tree as the accepted moves before it leave it (parent
// links followed through those moves' re-linkings); if no move trips it, the sequential pass would not have tripped it
// either and the accepted set is exactly what it applies - so it is applied here, all moves at once (their six-node sets
// are disjoint), and the boxes are recomputed level by level.  If one does trip it the iteration goes to the host.
struct Six {
    uint32_t from, p, s, g, to, tp; // p == kReinsertNone: this candidate has no move to make
};
enum : uint8_t { kNoMove = 0, kUndecided = 1, kAccepted = 2, kRejected = 3 };

__global__ void __launch_bounds__(kBlock) k_six(const DevNode *nodes, const uint32_t *parent, const uint32_t *cand, const uint32_t *found,
                                                uint32_t n_cand, Six *six, uint8_t *status) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    Six m;
    m.from = cand[k];
    m.to = found[k];
    m.p = m.s = m.g = m.tp = kReinsertNone;
    bool ok = m.to != kReinsertNone && m.to != kReinsertOverflow;
    if (ok) {
        const uint32_t p = parent[m.from];
        ok = p != 0u && p != kReinsertNone && m.to != p && m.to != m.from && parent[m.to] != kReinsertNone;
        if (ok) {
            uint32_t left, right;
            {
                const uint2 d = reinterpret_cast<const uint2 *>(nodes + p)[3];
                left = d.x;
                right = d.y;
            }
            const uint32_t s = left == m.from ? right : left;
            ok = m.to != s; // already its sibling: nothing to gain
            if (ok) {
                m.p = p;
                m.s = s;
                m.g = parent[p];
                m.tp = parent[m.to];
            }
        }
    }
    six[k] = m;
    status[k] = ok ? kUndecided : kNoMove;
}
__global__ void __launch_bounds__(kBlock) k_overflowed(const uint32_t *found, uint32_t n_cand, uint32_t *flags) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_cand && found[k] == kReinsertOverflow) atomicOr(flags, 4u);
}
__global__ void __launch_bounds__(kBlock) k_claim(const Six *six, const uint8_t *status, uint32_t n_cand, uint32_t *claim, int reset) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand) return;
    const uint8_t st = status[k];
    if (st == kNoMove) return;
    const Six m = six[k];
    const uint32_t v[6] = {m.from, m.p, m.s, m.g, m.to, m.tp};
    if (reset) { // every node some candidate may claim: free again
        for (int i = 0; i < 6; i++) claim[v[i]] = kReinsertNone;
    } else if (st == kUndecided || st == kAccepted) {
        for (int i = 0; i < 6; i++) atomicMin(&claim[v[i]], k);
    }
}
__global__ void __launch_bounds__(kBlock) k_decide(const Six *six, uint8_t *status, uint32_t n_cand, const uint32_t *claim, uint32_t *undecided) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand || status[k] != kUndecided) return;
    const Six m = six[k];
    const uint32_t v[6] = {m.from, m.p, m.s, m.g, m.to, m.tp};
    bool mine = true, lost = false;
    for (int i = 0; i < 6; i++) {
        const uint32_t c = claim[v[i]];
        if (c != k) {
            mine = false;
            // (an accepted move's status is final; one accepted in this very launch may still read as undecided here:
            // the rejection then comes a round later)
            if (c < k && reinterpret_cast<const volatile uint8_t *>(status)[c] == kAccepted) lost = true;
        }
    }
    if (mine)
        status[k] = kAccepted;
    else if (lost)
        status[k] = kRejected;
    else
        atomicAdd(undecided, 1u);
}
// claim[] holds the accepted moves only.  Does `to` lie below `from` once the accepted moves before k are in?
__global__ void __launch_bounds__(kBlock) k_verify(const Six *six, const uint8_t *status, uint32_t n_cand, const uint32_t *parent,
                                                   const uint32_t *claim, uint32_t *flags) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand || status[k] != kAccepted) return;
    const Six m = six[k];
    uint32_t x = m.to;
    for (uint32_t steps = 0;; steps++) {
        if (steps > 65536u) { // (cannot happen on a tree: some earlier move must have tripped the rule itself)
            atomicOr(flags, 2u);
            return;
        }
        uint32_t up = parent[x];
        const uint32_t j = claim[x];
        if (j < k) { // x is re-linked by an accepted move that comes first
            const Six o = six[j];
            if (x == o.from || x == o.to) up = o.p;
            else if (x == o.p) up = o.tp;
            else if (x == o.s) up = o.g;
        }
        if (up == kReinsertNone) return;
        if (up == m.from) {
            atomicOr(flags, 1u);
            return;
        }
        x = up;
    }
}
// Reinserter::move's re-linking for every accepted move (the boxes follow in k_refit)
__global__ void __launch_bounds__(kBlock) k_apply(const Six *six, const uint8_t *status, uint32_t n_cand, DevNode *nodes, uint32_t *parent,
                                                  uint32_t *moved) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_cand || status[k] != kAccepted) return;
    const Six m = six[k];
    // replace_child(g, p, s)
    if (nodes[m.g].left == m.p) nodes[m.g].left = m.s; else nodes[m.g].right = m.s;
    parent[m.s] = m.g;
    // replace_child(tp, to, p): the target's parent as it is now (g and tp may be one node, s and tp too)
    const uint32_t tp = parent[m.to];
    if (nodes[tp].left == m.to) nodes[tp].left = m.p; else nodes[tp].right = m.p;
    parent[m.p] = tp;
    nodes[m.p].left = m.to;
    nodes[m.p].right = m.from;
    parent[m.to] = m.p;
    parent[m.from] = m.p;
    atomicAdd(moved, 1u);
}

// Boxes of the inner nodes, level by level from the deepest: box = left's grown by right's (min / max: the value does not
// depend on the order, so this is what refit_up leaves on the host).
__device__ __forceinline__ uint32_t wave_append2(uint32_t *counter, uint32_t want) {
    const uint32_t lane = __lane_id();
    uint32_t scan = want;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(scan, d, 64);
        if ((int)lane >= d) scan += up;
    }
    const uint32_t total = __shfl(scan, 63, 64);
    uint32_t base = 0;
    if (lane == 63 && total) base = atomicAdd(counter, total);
    base = __shfl(base, 63, 64);
    return base + scan - want;
}
// (capacity = entries `out` can take: links that no longer describe a tree - the very case the host reports - must not turn
// into stores past the list; an append that does not fit is dropped, the counter still says how many were wanted)
__global__ void __launch_bounds__(kBlock) k_level(const DevNode *nodes, const uint32_t *in, uint32_t n_in, uint32_t *out, uint32_t *counter,
                                                   uint32_t capacity) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    uint32_t left = 0, right = 0, count = 0;
    if (t < n_in) {
        const uint2 *p = reinterpret_cast<const uint2 *>(nodes + in[t]);
        const uint2 d = p[3], e = p[4];
        left = d.x;
        right = d.y;
        count = e.y;
    }
    const bool inner = count > 1;
    const uint32_t at = wave_append2(counter, inner ? 2u : 0u);
    if (inner && at + 2u <= capacity) {
        out[at] = left;
        out[at + 1] = right;
    }
}
__global__ void __launch_bounds__(kBlock) k_refit(DevNode *nodes, const uint32_t *list, uint32_t n) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const uint32_t i = list[t];
    const DevNode nd = load_node(nodes, i);
    if (nd.count == 1) return;
    Aabb b = load_node(nodes, nd.left).box;
    grow_dev(b, load_node(nodes, nd.right).box);
    nodes[i].box = b;
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
    void *d_six = nullptr;      // Six[n_nodes] (at most every node is a candidate)
    uint8_t *d_status = nullptr;
    uint32_t *d_claim = nullptr, *d_list = nullptr, *d_words = nullptr; // d_words: undecided, flags, moved, level counter
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
        if (c->d_six) (void)hipFree(c->d_six);
        if (c->d_status) (void)hipFree(c->d_status);
        if (c->d_claim) (void)hipFree(c->d_claim);
        if (c->d_list) (void)hipFree(c->d_list);
        if (c->d_words) (void)hipFree(c->d_words);
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

} // namespace trx



namespace trx {

namespace {
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
} // namespace

bool reinsert_dev_upload(ReinsertDevice *c, const void *nodes, const uint32_t *parent, std::string &err) {
    if (!c) {
        err = "reinsert_dev_upload: no context";
        return false;
    }
    DeviceGuard guard;
    RG_TRY(hipSetDevice(c->device));
    RG_TRY(hipMemcpy(c->d_nodes, nodes, c->n_nodes * sizeof(DevNode), hipMemcpyHostToDevice));
    RG_TRY(hipMemcpy(c->d_parent, parent, c->n_nodes * 4, hipMemcpyHostToDevice));
    return true;
}

bool reinsert_dev_download(ReinsertDevice *c, void *nodes, uint32_t *parent, std::string &err) {
    if (!c) {
        err = "reinsert_dev_download: no context";
        return false;
    }
    DeviceGuard guard;
    RG_TRY(hipSetDevice(c->device));
    RG_TRY(hipMemcpy(nodes, c->d_nodes, c->n_nodes * sizeof(DevNode), hipMemcpyDeviceToHost));
    RG_TRY(hipMemcpy(parent, c->d_parent, c->n_nodes * 4, hipMemcpyDeviceToHost));
    return true;
}

bool reinsert_dev_iteration_resident(ReinsertDevice *c, uint32_t take, uint32_t *ids, uint32_t *found, uint32_t *moved, bool *to_host,
                                     double *seconds, std::string &err, bool force_host, uint32_t *rounds_out) {
    if (!c || take > c->n_nodes || c->n_nodes > 0x7fffffffull || !moved || !to_host) {
        err = "reinsert_dev_iteration_resident: bad arguments";
        return false;
    }
    *moved = 0;
    *to_host = false;
    if (take == 0) return true;
    DeviceGuard guard;
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
    if (!c->d_six) {
        RG_TRY(hipMalloc(&c->d_six, (size_t)n * sizeof(Six)));
        RG_TRY(hipMalloc(&c->d_status, (size_t)n));
        RG_TRY(hipMalloc(&c->d_claim, (size_t)n * 4));
        RG_TRY(hipMalloc(&c->d_list, (size_t)n * 4));
        RG_TRY(hipMalloc(&c->d_words, 64));
        RG_TRY(hipMemset(c->d_claim, 0xff, (size_t)n * 4)); // nobody claims anything; kept that way between iterations
    }
    Six *const six = static_cast<Six *>(c->d_six);
    RG_TRY(hipEventRecord(c->ev0, nullptr));
    // candidates and their places, as reinsert_dev_iteration finds them - on the tree that is already here
    hipLaunchKernelGGL(k_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, c->d_nodes, c->d_parent, n, c->d_keys_a);
    RG_TRY(hipGetLastError());
    size_t bytes = c->sort_tmp_bytes;
    RG_TRY(hipcub::DeviceRadixSort::SortKeysDescending(c->d_sort_tmp, bytes, c->d_keys_a, c->d_keys_b, (int)n, 0, 64, (hipStream_t) nullptr));
    const dim3 over_take((take + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_ids, over_take, dim3(kBlock), 0, nullptr, c->d_keys_b, take, c->d_cand);
    const int blocks = (int)std::min<size_t>((size_t)c->grid, ((size_t)take + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_find, dim3(blocks), dim3(kBlock), 0, nullptr, c->d_nodes, c->d_parent, c->d_cand, take, c->d_found, c->d_stacks);
    RG_TRY(hipGetLastError());
    // a search that outgrew its stack is repeated on the host: the whole iteration goes there
    hipLaunchKernelGGL(k_six, over_take, dim3(kBlock), 0, nullptr, c->d_nodes, c->d_parent, c->d_cand, c->d_found, take, six, c->d_status);
    RG_TRY(hipGetLastError());
    uint32_t words[4] = {0, 0, 0, 0}; // undecided, flags, moved, level counter
    bool host = force_host;
    {
        // (an overflowed search: found == kReinsertOverflow somewhere; counted with a tiny reduction over found on the host
        // side would need the array - the flag kernel below does it in place)
        RG_TRY(hipMemset(c->d_words, 0, 16));
    }
    // independent set in candidate order
    int rounds = 0;
    for (; !host; rounds++) {
        if (rounds > 512) { // (a dependency chain this long is not a tree anyone built; the host decides)
            host = true;
            break;
        }
        RG_TRY(hipMemsetAsync(c->d_words, 0, 4, nullptr));
        hipLaunchKernelGGL(k_claim, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_claim, 1);
        hipLaunchKernelGGL(k_claim, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_claim, 0);
        hipLaunchKernelGGL(k_decide, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_claim, c->d_words);
        RG_TRY(hipGetLastError());
        RG_TRY(hipMemcpy(words, c->d_words, 4, hipMemcpyDeviceToHost));
        if (words[0] == 0) break;
    }
    if (!host) {
        // claims of the accepted moves only, then the rule the independent set does not know
        hipLaunchKernelGGL(k_claim, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_claim, 1);
        hipLaunchKernelGGL(k_claim, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_claim, 0);
        hipLaunchKernelGGL(k_verify, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_parent, c->d_claim, c->d_words + 1);
        hipLaunchKernelGGL(k_overflowed, over_take, dim3(kBlock), 0, nullptr, c->d_found, take, c->d_words + 1);
        RG_TRY(hipGetLastError());
        RG_TRY(hipMemcpy(words, c->d_words, 16, hipMemcpyDeviceToHost));
        host = words[1] != 0;
    }
    if (rounds_out) *rounds_out = (uint32_t)rounds;
    // (free the claims again whatever happens next)
    hipLaunchKernelGGL(k_claim, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_claim, 1);
    RG_TRY(hipGetLastError());
    if (host) {
        *to_host = true;
        RG_TRY(hipEventRecord(c->ev1, nullptr));
        RG_TRY(hipMemcpy(ids, c->d_cand, (size_t)take * 4, hipMemcpyDeviceToHost));
        RG_TRY(hipMemcpy(found, c->d_found, (size_t)take * 4, hipMemcpyDeviceToHost));
    } else {
        hipLaunchKernelGGL(k_apply, over_take, dim3(kBlock), 0, nullptr, six, c->d_status, take, c->d_nodes, c->d_parent, c->d_words + 2);
        RG_TRY(hipGetLastError());
        // boxes: levels top-down (kept as lists), then refit bottom-up
        std::vector<uint32_t> level{0u, 1u};
        const uint32_t zero = 0;
        RG_TRY(hipMemcpy(c->d_list, &zero, 4, hipMemcpyHostToDevice));
        for (;;) {
            const uint32_t begin = level[level.size() - 2], end = level.back();
            if (end == begin || end >= n) break;
            RG_TRY(hipMemsetAsync(c->d_words + 3, 0, 4, nullptr));
            hipLaunchKernelGGL(k_level, dim3((end - begin + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, c->d_nodes, c->d_list + begin,
                               end - begin, c->d_list + end, c->d_words + 3, (uint32_t)(n - end));
            RG_TRY(hipGetLastError());
            uint32_t made = 0;
            RG_TRY(hipMemcpy(&made, c->d_words + 3, 4, hipMemcpyDeviceToHost));
            if ((size_t)end + made > n) {
                err = "reinsert_dev_iteration_resident: the tree came apart (links do not describe n_nodes nodes)";
                return false;
            }
            level.push_back(end + made);
        }
        if (level.back() != n) {
            err = "reinsert_dev_iteration_resident: " + std::to_string(n - level.back()) + " nodes are no longer reachable from the root";
            return false;
        }
        for (size_t L = level.size() - 1; L-- > 0;)
            hipLaunchKernelGGL(k_refit, dim3((level[L + 1] - level[L] + kBlock - 1) / kBlock), dim3(kBlock), 0, nullptr, c->d_nodes,
                               c->d_list + level[L], level[L + 1] - level[L]);
        RG_TRY(hipGetLastError());
        RG_TRY(hipEventRecord(c->ev1, nullptr));
        RG_TRY(hipMemcpy(words, c->d_words, 16, hipMemcpyDeviceToHost));
        *moved = words[2];
    }
    if (seconds) {
        float ms = 0.f;
        RG_TRY(hipEventSynchronize(c->ev1));
        RG_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
        *seconds += ms * 1e-3;
    }
    return true;
}

} // namespace trx
