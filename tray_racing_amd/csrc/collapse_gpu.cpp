// collapse_gpu.cpp — the last stage of a GPU build: BVH2 -> 8-wide compressed nodes (Ylitie et al. 2017, section 4.2: the
// seven-entry cost table per BVH2 node, then the tree those decisions describe; what obvhs does between its BVH2 and the
// CwBvh the reference uploads, src/main.rs:170-186, and what embree/src/bvh_embree_to_cwbvh.rs:85-186 does for the embree
// builder) as level-synchronous kernels:
//   1. BVH2 levels       top-down frontier expansion; the frontiers are kept, so every level is a list of node ids
//   2. cost table        one thread per node, deepest level first (both children are done when a node runs)
//   3. CWBVH levels      top-down over the collapsed tree: children of a node (distribute decisions followed), octant slot
//                        assignment, one record per node; a node's inner children are consecutive records in slot order
//   4. subtree sizes     bottom-up over the records: nodes and primitives below each record
//   5. output offsets    top-down: the sequential emission (a node's children allocated when it is visited, its primitives
//                        appended then, children visited in slot order) fixes every index as a function of those sizes
//   6. encode            one thread per record: quantisation frame, child boxes, meta bytes, primitive indices
// Same decisions, same tie rules, same binary32 / binary64 operations without contraction as Collapser in builder.cpp,
// so the bytes are the host's bytes.
#include "collapse_gpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

namespace trx {
namespace {

struct DevNode { // = Node2 of builder.cpp (40 bytes)
    Aabb box;
    uint32_t left, right, prim, count;
};
static_assert(sizeof(DevNode) == 40, "Node2 layout");

enum : uint32_t { kLeaf = 0, kInternal = 1, kDistribute = 2 };
struct Dec { // = Decision of builder.cpp
    float cost;
    uint32_t type, dl, dr;
};
// 8 bytes in memory: cost | type, dl, dr, 0
__device__ __forceinline__ Dec load_dec(const uint2 *dec, size_t i) {
    const uint2 v = dec[i];
    return Dec{__uint_as_float(v.x), v.y & 0xffu, (v.y >> 8) & 0xffu, (v.y >> 16) & 0xffu};
}
__device__ __forceinline__ uint2 pack_dec(float cost, uint32_t type, uint32_t dl, uint32_t dr) {
    return make_uint2(__float_as_uint(cost), type | (dl << 8) | (dr << 16));
}

struct Rec { // one per CWBVH node, in discovery order (level by level)
    uint32_t n2;          // the BVH2 node it is made from
    uint32_t first_child; // record of its first inner child; the others follow in slot order
    uint32_t sub_nodes, sub_prims; // records / primitives in its subtree, itself included
    uint32_t out_idx, child_base, prim_base;
    uint32_t info;        // imask | n_inner << 8 | total_tris << 16
    uint32_t slot_n2[8];  // BVH2 node per slot, kEmpty = none
};
static_assert(sizeof(Rec) == 64, "record size");
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr int kBlock = 256;

__device__ __forceinline__ float half_area_dev(const Aabb &b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}

__device__ __forceinline__ DevNode load_node(const DevNode *nodes, uint32_t i) {
    const uint2 *p = reinterpret_cast<const uint2 *>(nodes + i);
    const uint2 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
    DevNode n;
    n.box.mn[0] = __uint_as_float(a.x); n.box.mn[1] = __uint_as_float(a.y); n.box.mn[2] = __uint_as_float(b.x);
    n.box.mx[0] = __uint_as_float(b.y); n.box.mx[1] = __uint_as_float(c.x); n.box.mx[2] = __uint_as_float(c.y);
    n.left = d.x; n.right = d.y; n.prim = e.x; n.count = e.y;
    return n;
}
__device__ __forceinline__ void load_links(const DevNode *nodes, uint32_t i, uint32_t &left, uint32_t &right, uint32_t &prim, uint32_t &count) {
    const uint2 *p = reinterpret_cast<const uint2 *>(nodes + i);
    const uint2 d = p[3], e = p[4];
    left = d.x; right = d.y; prim = e.x; count = e.y;
}

// `want` consecutive places at the end of a list whose length is *counter: one atomic per wave.
__device__ __forceinline__ uint32_t wave_append(uint32_t *counter, uint32_t want) {
    const uint32_t lane = __lane_id();
    uint32_t scan = want; // inclusive prefix sum over the wave
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

// ---- 1. BVH2 levels
// (capacity = entries `out` can take; an append past it is dropped and the host, which reads the counter, reports the links)
__global__ __launch_bounds__(kBlock) void k2_expand(const DevNode *nodes, const uint32_t *in, uint32_t n_in, uint32_t *out, uint32_t *counter,
                                                     uint32_t capacity) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    uint32_t left = 0, right = 0, prim, count = 0;
    if (t < n_in) load_links(nodes, in[t], left, right, prim, count);
    const bool inner = count > 1;
    const uint32_t at = wave_append(counter, inner ? 2u : 0u);
    if (inner && at + 2u <= capacity) {
        out[at] = left;
        out[at + 1] = right;
    }
}

// ---- 2. cost table (Collapser::cost_range for one node)
__global__ __launch_bounds__(kBlock) void k2_cost(DevNode *nodes, uint2 *dec, const uint32_t *list, uint32_t n, uint32_t max_prims,
                                                    float traversal_cost, float prim_cost) {
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const uint32_t ni = list[t];
    DevNode nd = load_node(nodes, ni);
    if (nd.count != 1) { // primitives below: the children (a level down) have theirs; the caller's may be stale
        nd.count = nodes[nd.left].count + nodes[nd.right].count;
        nodes[ni].count = nd.count;
    }
    uint2 *d = dec + (size_t)ni * 7;
    const float area = half_area_dev(nd.box);
    if (nd.count == 1) {
        const uint2 v = pack_dec(area * prim_cost, kLeaf, 0xff, 0xff);
        for (int i = 0; i < 7; i++) d[i] = v;
        return;
    }
    float cl[7], cr[7];
    for (int k = 0; k < 7; k++) {
        cl[k] = __uint_as_float(dec[(size_t)nd.left * 7 + k].x);
        cr[k] = __uint_as_float(dec[(size_t)nd.right * 7 + k].x);
    }
    const float inf = __builtin_inff();
    const float cost_leaf = nd.count <= max_prims ? area * (float)nd.count * prim_cost : inf;
    float cost_dist = inf;
    uint32_t bl = 0xff, br = 0xff;
    for (int k = 0; k < 7; k++) {
        const float c = cl[k] + cr[6 - k];
        if (c < cost_dist) {
            cost_dist = c;
            bl = (uint32_t)k;
            br = (uint32_t)(6 - k);
        }
    }
    const float cost_internal = cost_dist + area * traversal_cost;
    uint2 prev = cost_leaf < cost_internal ? pack_dec(cost_leaf, kLeaf, bl, br) : pack_dec(cost_internal, kInternal, bl, br);
    d[0] = prev;
    for (int i = 1; i < 7; i++) {
        float best = __uint_as_float(prev.x);
        uint32_t l = 0xff, r = 0xff;
        for (int k = 0; k < i; k++) {
            const float c = cl[k] + cr[i - k - 1];
            if (c < best) {
                best = c;
                l = (uint32_t)k;
                r = (uint32_t)(i - k - 1);
            }
        }
        if (l != 0xff) prev = pack_dec(best, kDistribute, l, r);
        d[i] = prev;
    }
}

// ---- 3. CWBVH levels
// Collapser::get_children(ni, 0): the BVH2 nodes that become the children of the CWBVH node made from ni, left to right.
__device__ __forceinline__ int children_of(const DevNode *nodes, const uint2 *dec, uint32_t ni, uint32_t *children) {
    // pending visits, the next one on top: {node, decision index | expand flag << 8}
    uint32_t st_n[16], st_i[16];
    int sp = 0, count = 0;
    st_n[0] = ni;
    st_i[0] = 0x100u;
    sp = 1;
    while (sp > 0) {
        sp--;
        const uint32_t n = st_n[sp], ii = st_i[sp];
        if (!(ii & 0x100u)) {
            if (count < 8) children[count] = n;
            count++;
            continue;
        }
        uint32_t left, right, prim, cnt;
        load_links(nodes, n, left, right, prim, cnt);
        if (cnt == 1) {
            if (count < 8) children[count] = n;
            count++;
            continue;
        }
        const Dec d = load_dec(dec, (size_t)n * 7 + (ii & 0xffu));
        const bool xr = load_dec(dec, (size_t)right * 7 + d.dr).type == kDistribute;
        const bool xl = load_dec(dec, (size_t)left * 7 + d.dl).type == kDistribute;
        if (sp + 2 > 16) return 9; // cannot happen: at most eight children, a pending visit per child
        st_n[sp] = right;
        st_i[sp] = xr ? (0x100u | d.dr) : 0u;
        sp++;
        st_n[sp] = left;
        st_i[sp] = xl ? (0x100u | d.dl) : 0u;
        sp++;
    }
    return count;
}

// Collapser::order_children: greedy octant-slot assignment (embree/src/bvh_embree.rs:284-349).  slot[s] = index into
// children or -1.
__device__ __forceinline__ void order_children_dev(const DevNode *nodes, const Aabb &box, const uint32_t *children, int count, int *slot_child) {
    const float pc[3] = {0.5f * (box.mn[0] + box.mx[0]), 0.5f * (box.mn[1] + box.mx[1]), 0.5f * (box.mn[2] + box.mx[2])};
    float cost[8][8];
    for (int c = 0; c < count; c++) {
        const DevNode ch = load_node(nodes, children[c]);
        const float d[3] = {0.5f * (ch.box.mn[0] + ch.box.mx[0]) - pc[0], 0.5f * (ch.box.mn[1] + ch.box.mx[1]) - pc[1],
                            0.5f * (ch.box.mn[2] + ch.box.mx[2]) - pc[2]};
        for (int s = 0; s < 8; s++) {
            const float sx = (s & 4) ? -1.f : 1.f, sy = (s & 2) ? -1.f : 1.f, sz = (s & 1) ? -1.f : 1.f;
            cost[c][s] = d[0] * sx + d[1] * sy + d[2] * sz;
        }
    }
    int assignment[8];
    bool filled[8];
    for (int c = 0; c < 8; c++) {
        assignment[c] = -1;
        filled[c] = false;
        slot_child[c] = -1;
    }
    for (;;) {
        float min_cost = 3.402823466e+38f;
        int min_slot = -1, min_index = -1;
        for (int c = 0; c < count; c++) {
            if (assignment[c] != -1) continue;
            for (int s = 0; s < 8; s++) {
                if (!filled[s] && cost[c][s] < min_cost) {
                    min_cost = cost[c][s];
                    min_slot = s;
                    min_index = c;
                }
            }
        }
        if (min_slot < 0) break;
        filled[min_slot] = true;
        assignment[min_index] = min_slot;
    }
    for (int c = 0; c < count; c++) {
        int s = assignment[c];
        if (s < 0) { // non-finite centre: first free slot
            for (s = 0; s < 8 && filled[s]; s++) {}
            if (s > 7) s = 7;
            filled[s] = true;
        }
        slot_child[s] = c;
    }
}

__global__ __launch_bounds__(kBlock) void k8_expand(const DevNode *nodes, const uint2 *dec, Rec *recs, uint32_t begin, uint32_t end,
                                                      uint32_t *n_recs, uint32_t *trouble) {
    const uint32_t r = begin + blockIdx.x * kBlock + threadIdx.x;
    uint32_t n_inner = 0, imask = 0, total_tris = 0;
    uint32_t slot_n2[8];
    for (int s = 0; s < 8; s++) slot_n2[s] = kEmpty;
    if (r < end) {
        const uint32_t ni = recs[r].n2;
        uint32_t children[8];
        const int count = children_of(nodes, dec, ni, children);
        if (count > 8) {
            atomicOr(trouble, 1u);
        } else {
            const DevNode nd = load_node(nodes, ni);
            int slot_child[8];
            order_children_dev(nodes, nd.box, children, count, slot_child);
            for (int s = 0; s < 8; s++) {
                if (slot_child[s] < 0) continue;
                const uint32_t c = children[slot_child[s]];
                slot_n2[s] = c;
                if (load_dec(dec, (size_t)c * 7).type == kInternal) {
                    imask |= 1u << s;
                    n_inner++;
                } else {
                    uint32_t left, right, prim, cnt;
                    load_links(nodes, c, left, right, prim, cnt);
                    total_tris += cnt;
                }
            }
            if (total_tris > 24) atomicOr(trouble, 2u);
        }
    }
    const uint32_t first = wave_append(n_recs, n_inner);
    if (r < end) {
        Rec &rec = recs[r];
        rec.first_child = first;
        rec.info = imask | (n_inner << 8) | (total_tris << 16);
        uint32_t k = 0;
        for (int s = 0; s < 8; s++) {
            rec.slot_n2[s] = slot_n2[s];
            if (imask & (1u << s)) recs[first + k++].n2 = slot_n2[s];
        }
    }
}

// ---- 4. subtree sizes (deepest level first)
__global__ __launch_bounds__(kBlock) void k8_sizes(Rec *recs, uint32_t begin, uint32_t end) {
    const uint32_t r = begin + blockIdx.x * kBlock + threadIdx.x;
    if (r >= end) return;
    const uint32_t info = recs[r].info, n_inner = (info >> 8) & 0xffu, first = recs[r].first_child;
    uint32_t nodes = 1, prims = info >> 16;
    for (uint32_t k = 0; k < n_inner; k++) {
        nodes += recs[first + k].sub_nodes;
        prims += recs[first + k].sub_prims;
    }
    recs[r].sub_nodes = nodes;
    recs[r].sub_prims = prims;
}

// ---- 5. output offsets (root first).  Collapser::emit visits a node, allocates its inner children at the end of the node
// array, appends its primitives, then visits the children in slot order: child k's own children start after everything
// children 0..k-1 put below themselves.
__global__ __launch_bounds__(kBlock) void k8_offsets(Rec *recs, uint32_t begin, uint32_t end) {
    const uint32_t r = begin + blockIdx.x * kBlock + threadIdx.x;
    if (r >= end) return;
    if (r == 0) {
        recs[0].out_idx = 0;
        recs[0].child_base = 1;
        recs[0].prim_base = 0;
    }
    const uint32_t info = recs[r].info, n_inner = (info >> 8) & 0xffu, first = recs[r].first_child;
    const uint32_t child_base = recs[r].child_base;
    uint32_t next_nodes = child_base + n_inner, next_prims = recs[r].prim_base + (info >> 16);
    for (uint32_t k = 0; k < n_inner; k++) {
        Rec &c = recs[first + k];
        c.out_idx = child_base + k;
        c.child_base = next_nodes;
        c.prim_base = next_prims;
        next_nodes += c.sub_nodes - 1;
        next_prims += c.sub_prims;
    }
}

// ---- 6. encode (Collapser::emit for one node; embree/src/bvh_embree_to_cwbvh.rs:85-186)
__global__ __launch_bounds__(kBlock) void k8_encode(const DevNode *nodes, const Rec *recs, uint32_t n_recs, uint4 *out_nodes, uint32_t *out_prims) {
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= n_recs) return;
    const Rec rec = recs[r];
    const DevNode nd = load_node(nodes, rec.n2);
    float e[3], p[3];
    uint32_t ebyte[3];
    for (int k = 0; k < 3; k++) {
        p[k] = nd.box.mn[k];
        // quant_scale: the smallest power of two >= max(extent, 1e-20) / 255
        const float extent = nd.box.mx[k] - nd.box.mn[k];
        const float x = (extent < 1e-20f ? 1e-20f : extent) * (1.0f / 255.0f);
        const uint32_t xb = __float_as_uint(x);
        e[k] = __uint_as_float((xb & 0x7fffffu) ? ((xb >> 23) + 1u) << 23 : xb);
        // make sure 255 steps reach the far plane after rounding
        while (ceil(((double)nd.box.mx[k] - (double)nd.box.mn[k]) / (double)e[k]) > 255.0) e[k] *= 2.0f;
        ebyte[k] = (__float_as_uint(e[k]) >> 23) & 0xffu;
    }
    const uint32_t imask = rec.info & 0xffu;
    uint32_t meta[2] = {0, 0}, q[6][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}}; // min_x,max_x,min_y,max_y,min_z,max_z
    uint32_t total_tris = 0;
    for (int s = 0; s < 8; s++) {
        const uint32_t c = rec.slot_n2[s];
        if (c == kEmpty) continue;
        const DevNode ch = load_node(nodes, c);
        for (int k = 0; k < 3; k++) {
            const float rcp = 1.0f / e[k];
            float lo = floorf((ch.box.mn[k] - p[k]) * rcp);
            float hi = ceilf((ch.box.mx[k] - p[k]) * rcp);
            lo = lo < 0.0f ? 0.0f : lo; // std::min(std::max(v, 0), 255)
            lo = 255.0f < lo ? 255.0f : lo;
            hi = hi < 0.0f ? 0.0f : hi;
            hi = 255.0f < hi ? 255.0f : hi;
            // keep the decoded planes conservative under f32 rounding of (c - p)
            while (lo > 0.0f && (double)p[k] + (double)lo * (double)e[k] > (double)ch.box.mn[k]) lo -= 1.0f;
            while (hi < 255.0f && (double)p[k] + (double)hi * (double)e[k] < (double)ch.box.mx[k]) hi += 1.0f;
            q[2 * k][s >> 2] |= ((uint32_t)lo & 0xffu) << (8 * (s & 3));
            q[2 * k + 1][s >> 2] |= ((uint32_t)hi & 0xffu) << (8 * (s & 3));
        }
        uint32_t m;
        if (imask & (1u << s)) {
            m = (24u + (uint32_t)s) | 0x20u;
        } else {
            // Collapser::collect_prims: the leaves below c in pre-order (left before right); at most three
            uint32_t st[4];
            int sp = 0;
            st[sp++] = c;
            uint32_t np = 0;
            while (sp > 0) {
                uint32_t left, right, prim, cnt;
                load_links(nodes, st[--sp], left, right, prim, cnt);
                if (cnt == 1) {
                    out_prims[rec.prim_base + total_tris + np] = prim;
                    np++;
                } else {
                    st[sp++] = right;
                    st[sp++] = left;
                }
            }
            const uint32_t unary = np == 1 ? 0x20u : np == 2 ? 0x60u : np == 3 ? 0xE0u : 0u;
            m = (total_tris | unary) & 0xffu;
            total_tris += np;
        }
        meta[s >> 2] |= m << (8 * (s & 3));
    }
    uint4 *o = out_nodes + (size_t)rec.out_idx * 5;
    o[0] = make_uint4(__float_as_uint(p[0]), __float_as_uint(p[1]), __float_as_uint(p[2]),
                      ebyte[0] | (ebyte[1] << 8) | (ebyte[2] << 16) | (imask << 24));
    o[1] = make_uint4(rec.child_base, rec.prim_base, meta[0], meta[1]);
    o[2] = make_uint4(q[0][0], q[0][1], q[1][0], q[1][1]);
    o[3] = make_uint4(q[2][0], q[2][1], q[3][0], q[3][1]);
    o[4] = make_uint4(q[4][0], q[4][1], q[5][0], q[5][1]);
}

#define CG_TRY(expr)                                                     \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);     \
            return false;                                                \
        }                                                                \
    } while (0)

struct Buffers {
    DevNode *nodes = nullptr;
    uint2 *dec = nullptr;
    uint32_t *list = nullptr, *counter = nullptr, *out_prims = nullptr;
    Rec *recs = nullptr;
    uint4 *out_nodes = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~Buffers() {
        if (nodes) (void)hipFree(nodes);
        if (dec) (void)hipFree(dec);
        if (list) (void)hipFree(list);
        if (counter) (void)hipFree(counter);
        if (out_prims) (void)hipFree(out_prims);
        if (recs) (void)hipFree(recs);
        if (out_nodes) (void)hipFree(out_nodes);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};

inline dim3 grid_for(size_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

} // namespace

bool collapse_encode_device(int device, const void *nodes, size_t n_nodes, uint32_t max_prims_per_leaf, float traversal_cost,
                            float prim_cost, std::vector<CwbvhNode> &out_nodes, std::vector<uint32_t> &out_prims,
                            float *root_cost, double *seconds, std::string &err) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        err = "no HIP device " + std::to_string(device) + " for the GPU build stage";
        return false;
    }
    if (n_nodes < 3 || n_nodes > 0x7fffffffull || (n_nodes & 1) == 0) {
        err = "collapse_encode_device: needs a tree of 2n-1 nodes, n >= 2";
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
    CG_TRY(hipSetDevice(device));
    const uint32_t n = (uint32_t)n_nodes, n_prims = (n + 1) / 2;
    Buffers b;
    CG_TRY(hipMalloc(&b.nodes, (size_t)n * sizeof(DevNode)));
    CG_TRY(hipMalloc(&b.dec, (size_t)n * 7 * sizeof(uint2)));
    CG_TRY(hipMalloc(&b.list, (size_t)n * 4));
    CG_TRY(hipMalloc(&b.counter, 16));
    // a CWBVH node has at least two children unless it is the root of a one-primitive scene, so there are fewer nodes than
    // primitives
    CG_TRY(hipMalloc(&b.recs, (size_t)n_prims * sizeof(Rec)));
    CG_TRY(hipMalloc(&b.out_prims, (size_t)n_prims * 4));
    CG_TRY(hipEventCreate(&b.ev0));
    CG_TRY(hipEventCreate(&b.ev1));
    CG_TRY(hipMemcpy(b.nodes, nodes, (size_t)n * sizeof(DevNode), hipMemcpyHostToDevice));
    CG_TRY(hipEventRecord(b.ev0, nullptr));

    // 1. BVH2 levels
    std::vector<uint32_t> level{0u, 1u}; // level L = list[level[L] .. level[L + 1])
    {
        const uint32_t zero = 0;
        CG_TRY(hipMemcpy(b.list, &zero, 4, hipMemcpyHostToDevice));
    }
    for (;;) {
        const uint32_t begin = level[level.size() - 2], end = level.back();
        if (end == begin) {
            level.pop_back();
            break;
        }
        if (end >= n) break; // every node is listed: the last level holds leaves only
        CG_TRY(hipMemsetAsync(b.counter, 0, 4, nullptr));
        hipLaunchKernelGGL(k2_expand, grid_for(end - begin), dim3(kBlock), 0, nullptr, b.nodes, b.list + begin, end - begin, b.list + end, b.counter,
                           (uint32_t)(n - end));
        CG_TRY(hipGetLastError());
        uint32_t made = 0;
        CG_TRY(hipMemcpy(&made, b.counter, 4, hipMemcpyDeviceToHost));
        if ((size_t)end + made > n) {
            err = "collapse_encode_device: the links do not describe a tree of n_nodes nodes";
            return false;
        }
        level.push_back(end + made);
    }
    if (level.back() != n) {
        err = "collapse_encode_device: " + std::to_string(n - level.back()) + " nodes are not reachable from node 0";
        return false;
    }
    // 2. cost table, deepest level first
    for (size_t L = level.size() - 1; L-- > 0;) {
        const uint32_t begin = level[L], end = level[L + 1];
        hipLaunchKernelGGL(k2_cost, grid_for(end - begin), dim3(kBlock), 0, nullptr, b.nodes, b.dec, b.list + begin, end - begin,
                           max_prims_per_leaf, traversal_cost, prim_cost);
    }
    CG_TRY(hipGetLastError());
    // 3. CWBVH levels
    std::vector<uint32_t> level8{0u, 1u};
    {
        const uint32_t first[2] = {1u, 0u}; // records so far; trouble flags
        CG_TRY(hipMemcpy(b.counter, first, 8, hipMemcpyHostToDevice));
        const uint32_t zero = 0;
        CG_TRY(hipMemcpy(&b.recs[0].n2, &zero, 4, hipMemcpyHostToDevice));
    }
    for (;;) {
        const uint32_t begin = level8[level8.size() - 2], end = level8.back();
        if (end == begin) {
            level8.pop_back();
            break;
        }
        hipLaunchKernelGGL(k8_expand, grid_for(end - begin), dim3(kBlock), 0, nullptr, b.nodes, b.dec, b.recs, begin, end, b.counter, b.counter + 1);
        CG_TRY(hipGetLastError());
        uint32_t state[2] = {0, 0};
        CG_TRY(hipMemcpy(state, b.counter, 8, hipMemcpyDeviceToHost));
        if (state[1] || state[0] > n_prims || state[0] < end) {
            err = "collapse_encode_device: the decisions give a node more than eight children or 24 primitives";
            return false;
        }
        level8.push_back(state[0]);
    }
    const uint32_t n_recs = level8.back();
    // 4. subtree sizes, 5. output offsets
    for (size_t L = level8.size() - 1; L-- > 0;)
        hipLaunchKernelGGL(k8_sizes, grid_for(level8[L + 1] - level8[L]), dim3(kBlock), 0, nullptr, b.recs, level8[L], level8[L + 1]);
    for (size_t L = 0; L + 1 < level8.size(); L++)
        hipLaunchKernelGGL(k8_offsets, grid_for(level8[L + 1] - level8[L]), dim3(kBlock), 0, nullptr, b.recs, level8[L], level8[L + 1]);
    CG_TRY(hipGetLastError());
    // 6. encode
    CG_TRY(hipMalloc(&b.out_nodes, (size_t)n_recs * sizeof(CwbvhNode)));
    hipLaunchKernelGGL(k8_encode, grid_for(n_recs), dim3(kBlock), 0, nullptr, b.nodes, b.recs, n_recs, b.out_nodes, b.out_prims);
    CG_TRY(hipGetLastError());
    CG_TRY(hipEventRecord(b.ev1, nullptr));
    Rec root;
    CG_TRY(hipMemcpy(&root, b.recs, sizeof(Rec), hipMemcpyDeviceToHost));
    if (root.sub_nodes != n_recs || root.sub_prims != n_prims) {
        err = "collapse_encode_device: subtree sizes do not add up";
        return false;
    }
    out_nodes.resize(n_recs);
    out_prims.resize(n_prims);
    CG_TRY(hipMemcpy(out_nodes.data(), b.out_nodes, (size_t)n_recs * sizeof(CwbvhNode), hipMemcpyDeviceToHost));
    CG_TRY(hipMemcpy(out_prims.data(), b.out_prims, (size_t)n_prims * 4, hipMemcpyDeviceToHost));
    if (root_cost) {
        uint2 d0;
        CG_TRY(hipMemcpy(&d0, b.dec, 8, hipMemcpyDeviceToHost));
        std::memcpy(root_cost, &d0.x, 4);
    }
    if (seconds) {
        float ms = 0.f;
        CG_TRY(hipEventElapsedTime(&ms, b.ev0, b.ev1));
        *seconds += ms * 1e-3;
    }
    return true;
}

} // namespace trx
