// api.cpp — implementation of the C-ABI declared in include/trx.h (and of the development entry points of
// include/trx_dev.h).
//
// Host side of the HIP backend: scene upload (replaces the buffer creation of
// src/rt_gpu/rt_gpu_software.rs:83-160), launch + hipEvent timing (replaces
// the dispatch and src/timestamp.rs), and the flat-buffer assembly of
// cwbvh_gpu_runner (src/rt_gpu/mod.rs:16-112).  There is deliberately no CPU
// traversal in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "../../include/trx.h"
#include "../../include/trx_dev.h"
#include "builder.h"
#include "cwbvh_format.h"
#include "kernels.h"
#include "scenes.h"

using namespace trx;

namespace {

thread_local std::string g_err;
std::atomic<uint32_t> g_variant{0}; // tuning aid, read once per launch
// Builder settings for subsequent builds (trx_set_build_*): process-wide, guarded by g_build_mu; a build takes
// a snapshot when it starts, so concurrent builds and setters do not race.
struct BuildSettings {
    float traversal_cost = 1.0f, prim_cost = 0.3f;
    float reinsert_ratio = 0.02f;
    int reinsert_iters = 4;
    int sah_bins = 32;
    uint32_t sweep_max = 48;
    float pre_split = 0.0f;
    uint32_t ploc_distance = 0; // 0: binned-SAH BVH2; > 0: PLOC with this search distance
    uint32_t ploc_depth_threshold = 2, ploc_sort_bits = 64;
    int ploc_device = -1;       // >= 0: the PLOC stage of large builds runs on this HIP device (trx_set_build_device)
    bool reinsert_batched = false;
    bool reinsert_whole = false; // one batch per iteration (trx_set_build_reinsertion_batches)
    float rebraid_area = 1.0f / 4096.0f; // TLAS: open BLAS subtrees whose box exceeds this share of the scene box's area (0 = never)
};
BuildSettings g_build;
std::mutex g_build_mu;
BuildSettings build_settings() {
    std::lock_guard<std::mutex> lock(g_build_mu);
    return g_build;
}
BuildParams to_build_params(const BuildSettings &b, uint32_t max_prims, int threads) {
    BuildParams bp;
    bp.max_prims_per_leaf = max_prims;
    bp.threads = threads;
    bp.traversal_cost = b.traversal_cost;
    bp.prim_cost = b.prim_cost;
    bp.reinsertion_batch_ratio = b.reinsert_ratio;
    bp.reinsertion_iterations = b.reinsert_iters;
    bp.sah_bins = b.sah_bins;
    bp.sweep_max = b.sweep_max;
    bp.pre_split_ratio = b.pre_split;
    bp.ploc_search_distance = b.ploc_distance;
    bp.ploc_search_depth_threshold = b.ploc_depth_threshold;
    bp.ploc_sort_bits = b.ploc_sort_bits;
    bp.ploc_device = b.ploc_device;
    bp.reinsertion_batched = b.reinsert_batched;
    bp.reinsertion_whole_iterations = b.reinsert_whole;
    return bp;
}

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? TRX_ERR_OOM : TRX_ERR_NO_DEVICE,       \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                    \
    } while (0)

constexpr int kSlots = 8;
constexpr uint32_t kDefaultWavesPerBlock = 1;

struct Slot {
    SlotCounters *ctr = nullptr;
    uint2 *spill = nullptr;
    uint32_t spill_waves = 0;  // waves the spill area is sized for
    hipEvent_t done = nullptr; // everything enqueued for this slot has finished
    bool used = false;
    hipStream_t last_stream = nullptr; // stream of the last launch on this slot
    uint64_t last_use = 0;             // launch counter at that time (oldest slot is recycled first)
    // tile-cost feedback: the previous frame of a kind (primary / AO) traced on this slot measured every tile; the
    // next one of that kind with the same image geometry starts its heaviest tiles first.  One state per kind: the
    // reference's frame loop runs both passes on one queue, and each has its own order.
    struct Order {
        uint32_t *lists = nullptr; // two sets of {16 counts, 16 lists}
        uint32_t capacity = 0;
        bool have_views = false; // view[] holds the views of a previous launch
        uint64_t key = 0;    // (width, height, shard, mode) the lists were measured for; 0 = none
        ViewDev view[kMaxBatchFrames]{}; // views of the last launch that read or wrote the lists (camera-cut detection)
    } order[2];
};

} // namespace

namespace trx {
// error reporting for the other translation units of the library (comm.cpp)
int fail_msg(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
} // namespace trx

// trx_traverse1 is Traversable::traverse(&self, Ray) -> RayHit (traversable/src/lib.rs:13-28), called per pixel from every
// worker of a thread pool at once (src/rt_cpu/rt_cpu.rs:35-57).  One launch per ray would be a host-to-device copy, a
// one-wave launch, a copy back and a stream synchronisation for 32 bytes of work (rounds 1-4: ~25 k rays a second per
// thread); so the callers that are inside trx_traverse1 at the same time share launches.  A caller drops its ray into the
// open batch (pinned host memory the kernel reads and writes in place: no copies) and takes a ticket; the first one in
// is the batch's leader: it waits until arrivals stop for a few microseconds (or kCap rays, or kMaxWait), closes the
// batch, launches it on the batch's own stream, waits for it and wakes the others, who read their records by ticket.
// Rays and results are those of the single-ray path; a caller still blocks for one GPU round trip (launch + completion,
// 15-30 us), so the rate is (callers inside at once) / (round trip): it scales with the thread count, not with the GPU.
struct RayCombiner {
    static constexpr uint32_t kCap = 4096, kBatches = 4;
    static constexpr int64_t kQuietNs = 3000, kMaxWaitNs = 50000;
    // A wave steps a handful of rays about twice as fast as a few dozen (eight lanes to a ray, kernels.hip "thin waves"),
    // and a small batch is all latency: its first kSpread rays are dealt eight to a 64-ray chunk - one wave each - the rest
    // of a chunk being rays that end at the root (tmax < 0).  Ticket i's record sits at slot(i).
    static constexpr uint32_t kSpread = 512, kSpreadSlots = kSpread / 8 * 64, kSlots = kSpreadSlots + (kCap - kSpread);
    static uint32_t slot(uint32_t i) { return i < kSpread ? (i >> 3) * 64u + (i & 7u) : kSpreadSlots + (i - kSpread); }
    static uint32_t slots_used(uint32_t n) { return n <= kSpread ? ((n + 7u) >> 3) * 64u : kSpreadSlots + (n - kSpread); }
    static trx_ray null_ray() {
        trx_ray r;
        std::memset(&r, 0, sizeof(r));
        r.direction[0] = 1.0f;
        r.tmax = -1.0f; // nothing lies in [0, -1]: the root's test fails and the ray is finished after one step
        return r;
    }
    struct Batch {
        trx_ray *rays = nullptr;  // pinned, device-visible
        trx_hit *hits = nullptr;
        uint32_t *inst = nullptr;
        uint32_t *over = nullptr; // pinned word the kernel sets when a ray of the batch overflowed its stack / hit the step cap
        hipStream_t stream = nullptr;
        uint32_t n = 0, sem = 0;
        std::atomic<uint32_t> read{0};       // callers that have taken their record (the last one frees the batch)
        int rc = 0;
        std::string err;
        enum State { kFree, kOpen, kFlying, kDone } state = kFree;
        // bumped when the batch's results are in: followers spin on it, then sleep on it (a futex: a woken follower reads
        // its record and leaves without taking any lock - woken through a condition variable they queued up on its mutex,
        // 5-10 us each, and arrived at the next batch one by one)
        std::atomic<uint32_t> done_epoch{0};
    } batch[kBatches];
    std::mutex mu;
    std::condition_variable cv;              // the open batch changed, or a batch became free
    std::atomic<int> inside{0};              // callers inside trx_traverse1 (spinning only pays while they fit the host's cores)
    int cores = 1;
    int open = -1;
    int device = 0;
    bool ok = false;
    std::string init_err;
    uint64_t launches = 0, rays = 0; // statistics (trx_debug_traverse1_stats)

    explicit RayCombiner(int dev) : device(dev) {
        for (Batch &b : batch) {
            hipError_t e = hipHostMalloc((void **)&b.rays, kSlots * sizeof(trx_ray), hipHostMallocDefault);
            if (e == hipSuccess) e = hipHostMalloc((void **)&b.hits, kSlots * sizeof(trx_hit), hipHostMallocDefault);
            if (e == hipSuccess) e = hipHostMalloc((void **)&b.inst, kSlots * sizeof(uint32_t), hipHostMallocDefault);
            if (e == hipSuccess)
                for (uint32_t i = 0; i < kSpreadSlots; i++) b.rays[i] = null_ray();
            if (e == hipSuccess) e = hipHostMalloc((void **)&b.over, 64, hipHostMallocDefault);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&b.stream, hipStreamNonBlocking);
            if (e != hipSuccess) {
                init_err = hipGetErrorString(e);
                return;
            }
        }
        cores = (int)std::max(1u, std::thread::hardware_concurrency());
        ok = true;
    }
    ~RayCombiner() {
        for (Batch &b : batch) {
            if (b.stream) (void)hipStreamDestroy(b.stream);
            if (b.rays) (void)hipHostFree(b.rays);
            if (b.hits) (void)hipHostFree(b.hits);
            if (b.inst) (void)hipHostFree(b.inst);
            if (b.over) (void)hipHostFree(b.over);
        }
    }
};

static void futex_wait(std::atomic<uint32_t> *a, uint32_t while_value) {
    static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "futex word");
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAIT_PRIVATE, while_value, nullptr, nullptr, 0);
}
static void futex_wake_all(std::atomic<uint32_t> *a) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(a), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}
static int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

struct trx_scene {
    int device = 0;
    uint4 *d_nodes = nullptr;
    float4 *d_tris = nullptr;
    uint32_t *d_inst = nullptr;
    uint32_t *d_inst_entry = nullptr;        // entry node per TLAS primitive (re-braided scenes), or null
    std::vector<uint32_t> h_inst;            // host copy of the instance offsets (entry-node validation)
    uint64_t n_nodes = 0, n_tris = 0;
    uint32_t n_inst = 0, tlas_start = 0;
    bool tlas = false;
    float scene_diag = 0.f; // diagonal of the root node's box (camera-cut detection scales with it)
    uint32_t exp_exact = 0u; // 1: every node exponent byte is 0 or >= 21; 2: and every node origin admits div_by_rcp (TraceParams::exp_exact)
    int grid = 0;      // default number of persistent waves
    int cu_count = 0;
    unsigned long long *d_wave_times = nullptr; // diagnostics only (trx_debug_wave_timeline)
    uint32_t *dbg_cost = nullptr, *dbg_iters = nullptr; // diagnostics only (trx_debug_tile_profile)
    Slot slots[kSlots];
    uint64_t launches = 0;
    std::mutex mu;      // launch slots (every enqueue)
    std::recursive_mutex host_mu; // scratch buffers and event pair of the synchronous entry points
    // scratch for the host-buffer convenience entry points
    trx_hit *d_scratch_a = nullptr, *d_scratch_b = nullptr;
    uint32_t *d_scratch_ia = nullptr, *d_scratch_ib = nullptr; // instance ids beside scratch_a / scratch_b
    trx_ray *d_scratch_rays = nullptr;
    uint64_t scratch_hits = 0, scratch_rays = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<uint32_t> blas_tri_start; // geometry_id lookup for trx_traverse1
    RayCombiner *comb = nullptr;   // trx_traverse1: concurrent single-ray callers share launches (created on first use)
    std::once_flag comb_once;
    // instance transforms (TLAS scenes): object-to-world as given (get_instance_transform), world-to-object rows as
    // the kernels use them, and their device copy; empty / null = identity
    std::vector<float> inst_o2w, inst_w2o;
    float4 *d_inst_xform = nullptr;
};

struct trx_bvh {
    CwBvh bvh;
};

namespace {

// Structural validation of untrusted node buffers: a kernel that walks a
// malformed tree would read out of bounds (and can take the GPU down), so
// every index the kernel can form is range-checked here once.
int validate_nodes(const CwbvhNode *nodes, uint64_t n_nodes, uint64_t n_tris, const uint32_t *inst,
                   uint32_t n_inst, uint32_t tlas_start) {
    std::vector<uint32_t> seg; // BLAS segment starts, sorted
    const bool tlas = n_inst > 0;
    if (tlas) {
        if (tlas_start >= n_nodes) return fail(TRX_ERR_FORMAT, "tlas_start %u >= n_nodes %llu", tlas_start, (unsigned long long)n_nodes);
        seg.assign(inst, inst + n_inst);
        for (uint32_t o : seg)
            if (o >= tlas_start) return fail(TRX_ERR_FORMAT, "instance offset %u not below tlas_start %u", o, tlas_start);
        std::sort(seg.begin(), seg.end());
        seg.erase(std::unique(seg.begin(), seg.end()), seg.end());
    }
    for (uint64_t i = 0; i < n_nodes; i++) {
        const CwbvhNode &n = nodes[i];
        uint64_t seg_begin = 0, seg_end = n_nodes;
        bool in_tlas = false;
        if (tlas) {
            if (i >= tlas_start) {
                seg_begin = tlas_start;
                in_tlas = true;
            } else {
                auto it = std::upper_bound(seg.begin(), seg.end(), (uint32_t)i);
                seg_begin = it == seg.begin() ? 0 : *(it - 1);
                seg_end = it == seg.end() ? tlas_start : *it;
            }
        }
        uint32_t inner = 0, prims = 0;
        for (int s = 0; s < 8; s++) {
            uint8_t m = n.child_meta[s];
            if (m == 0) continue;
            if ((m & 0x18) == 0x18) {
                inner++;
            } else {
                uint32_t bits = m >> 5, cnt = bits == 1 ? 1 : bits == 3 ? 2 : bits == 7 ? 3 : 0;
                if (!cnt) return fail(TRX_ERR_FORMAT, "node %llu slot %d: bad leaf meta 0x%02x", (unsigned long long)i, s, m);
                prims = std::max(prims, (uint32_t)(m & 0x1f) + cnt);
            }
        }
        if (inner != (uint32_t)__builtin_popcount(n.imask))
            return fail(TRX_ERR_FORMAT, "node %llu: imask 0x%02x disagrees with child_meta", (unsigned long long)i, n.imask);
        if (inner && (uint64_t)n.child_base_idx + inner > seg_end - seg_begin)
            return fail(TRX_ERR_FORMAT, "node %llu: children [%u,+%u) leave the BVH segment", (unsigned long long)i, n.child_base_idx, inner);
        if (prims > 24) return fail(TRX_ERR_FORMAT, "node %llu: more than 24 primitives", (unsigned long long)i);
        uint64_t limit = in_tlas ? n_inst : n_tris;
        if (prims && (uint64_t)n.primitive_base_idx + prims > limit)
            return fail(TRX_ERR_FORMAT, "node %llu: primitives [%u,+%u) out of range %llu", (unsigned long long)i,
                        n.primitive_base_idx, prims, (unsigned long long)limit);
    }
    return TRX_OK;
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            float f = (float)man * 5.9604644775390625e-8f;
            return sign ? -f : f;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// Any accepted input format -> 48-byte device record {v0, e1 = v0 - v1, e2 = v2 - v0, ng = e1 x e2 in the w lanes}.
void convert_tris(const void *src, uint64_t n, uint32_t fmt, TriDev *dst) {
    const uint8_t *b = (const uint8_t *)src;
    for (uint64_t i = 0; i < n; i++) {
        TriDev t;
        std::memset(&t, 0, sizeof(t));
        if (fmt == TRX_TRI_F16_24) {
            float v[3];
            uint32_t e[3];
            std::memcpy(v, b + 24 * i, 12);
            std::memcpy(e, b + 24 * i + 12, 12);
            for (int k = 0; k < 3; k++) {
                t.v0[k] = v[k];
                t.e1[k] = -half_to_float((uint16_t)(e[k] >> 16)); // stored e1 = v1 - v0
                t.e2[k] = half_to_float((uint16_t)(e[k] & 0xffff));
            }
        } else {
            float v[9];
            std::memcpy(v, b + 36 * i, 36);
            for (int k = 0; k < 3; k++) {
                t.v0[k] = v[k];
                if (fmt == TRX_TRI_VERTS_36) {
                    t.e1[k] = v[k] - v[3 + k];
                    t.e2[k] = v[6 + k] - v[k];
                } else {
                    t.e1[k] = v[3 + k];
                    t.e2[k] = v[6 + k];
                }
            }
        }
        t.ngx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
        t.ngy = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
        t.ngz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
        dst[i] = t;
    }
}

int ensure_scratch(trx_scene *s, uint64_t hits, uint64_t rays) {
    if (hits > s->scratch_hits) {
        if (s->d_scratch_a) (void)hipFree(s->d_scratch_a);
        if (s->d_scratch_b) (void)hipFree(s->d_scratch_b);
        if (s->d_scratch_ia) (void)hipFree(s->d_scratch_ia);
        if (s->d_scratch_ib) (void)hipFree(s->d_scratch_ib);
        s->d_scratch_a = s->d_scratch_b = nullptr;
        s->d_scratch_ia = s->d_scratch_ib = nullptr;
        s->scratch_hits = 0;
        HIP_TRY(hipMalloc(&s->d_scratch_a, hits * sizeof(trx_hit)));
        HIP_TRY(hipMalloc(&s->d_scratch_b, hits * sizeof(trx_hit)));
        if (s->tlas) {
            HIP_TRY(hipMalloc(&s->d_scratch_ia, hits * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&s->d_scratch_ib, hits * sizeof(uint32_t)));
        }
        s->scratch_hits = hits;
    }
    if (rays > s->scratch_rays) {
        if (s->d_scratch_rays) (void)hipFree(s->d_scratch_rays);
        s->d_scratch_rays = nullptr;
        s->scratch_rays = 0;
        HIP_TRY(hipMalloc(&s->d_scratch_rays, rays * sizeof(trx_ray)));
        s->scratch_rays = rays;
    }
    return TRX_OK;
}

void fill_view(const trx_view *v, ViewDev &out) {
    std::memcpy(out.view_inv, v->view_inv, 64);
    std::memcpy(out.proj_inv, v->proj_inv, 64);
    std::memcpy(out.eye, v->eye, 12);
    out.pad = 0.f;
}

// Enqueue one traversal kernel on a launch slot.  Slots make the scene
// re-entrant (Traversable requires Sync, src/rt_cpu/rt_cpu.rs:17-20): a slot is
// reused only after the stream has waited for its previous kernel.
int enqueue(trx_scene *s, TraceParams &p, int mode, uint32_t sem, bool count, hipStream_t stream,
            SlotCounters **ctr_out) {
    if (sem & ~7u) return fail(TRX_ERR_INVALID, "unknown semantics bits 0x%x", sem);
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->d_inst_xform && mode == kModeAo && !p.primary_inst)
        return fail(TRX_ERR_INVALID, "this scene has instance transforms: the AO pass needs the primary pass's instance ids "
                                     "(trx_trace_ao_inst_dev) to take the hit normal into world space");
    // A stream keeps its slot: its launches are ordered anyway, and the slot's tile-order feedback
    // stays with the caller's frame loop.  Otherwise take an unused slot, else the oldest one, and
    // make the stream wait for that slot's last kernel.
    int pick = -1;
    for (int i = 0; i < kSlots && pick < 0; i++)
        if (s->slots[i].used && s->slots[i].last_stream == stream) pick = i;
    for (int i = 0; i < kSlots && pick < 0; i++)
        if (!s->slots[i].used) pick = i;
    if (pick < 0) {
        pick = 0;
        for (int i = 1; i < kSlots; i++)
            if (s->slots[i].last_use < s->slots[pick].last_use) pick = i;
    }
    Slot &slot = s->slots[pick];
    const bool same_stream = slot.used && slot.last_stream == stream;
    const uint32_t variant = g_variant.load(std::memory_order_relaxed);
    // tuning overrides (trx_set_kernel_variant): bits 8..12 waves per CU, bits 16..19 waves per workgroup
    uint32_t wpb = (variant >> 16) & 0x7u; // (bit 19: the tile-order feedback does not tune itself off, see below)
    // incoherent single-level passes (AO, explicit rays) run two waves to a workgroup, so that the second can hand its last rays to the first
    // when both are draining (kernels.hip, "drain"); an explicit 1 or 4 here switches that off
    // (two-level scenes: explicit rays only, see kMerge in kernels.hip)
    const bool merge_default = (wpb != 1 && wpb != 2 && wpb != 4) && mode != kModePrimary && mode != kModeFused && (!s->tlas || mode == kModeRays) && !count;
    if (wpb != 1 && wpb != 2 && wpb != 4) wpb = merge_default ? 2u : kDefaultWavesPerBlock;
    const uint32_t per_cu = (variant >> 8) & 0x1fu;
    int grid = per_cu ? (int)(std::min(per_cu, 32u) * (uint32_t)s->cu_count) : s->grid;
    // no more waves than chunks of work: a batch of one ray (trx_traverse1) is a one-wave launch with a
    // one-wave spill area
    const uint64_t n_chunks = ((uint64_t)p.n_items + 63u) >> 6;
    if ((uint64_t)grid > n_chunks) grid = (int)std::max<uint64_t>(n_chunks, 1);
    grid = std::max((int)wpb, (grid + (int)wpb - 1) / (int)wpb * (int)wpb);
    if (!slot.ctr) {
        // all or nothing: a slot is either fully usable or untouched
        SlotCounters *ctr = nullptr;
        hipEvent_t done = nullptr;
        HIP_TRY(hipMalloc(&ctr, sizeof(SlotCounters)));
        // On the LAUNCH stream: hipMemset returns before a device-side fill has run and orders it on the null stream only,
        // which a non-blocking user stream does not wait for - the first kernel of a slot could start, take tickets and
        // count exiting waves, and THEN have its queue heads and exit ticket zeroed under it (chunks dealt twice, the exit
        // ticket never reaching the grid size, the heads never re-armed: the next launch on the slot finds every queue dry
        // and writes nothing).  Seen once four processes time-shared the GPU; found by the sentinel check of bench.py's
        // test mode (profiles/r03_slot_init_race.log).
        hipError_t e = hipMemsetAsync(ctr, 0, sizeof(SlotCounters), stream);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipFree(ctr);
            return fail(TRX_ERR_NO_DEVICE, "launch slot set-up failed: %s", hipGetErrorString(e));
        }
        slot.ctr = ctr;
        slot.done = done;
    }
    if (slot.used && !same_stream) HIP_TRY(hipStreamWaitEvent(stream, slot.done, 0));
    if (slot.spill_waves < (uint32_t)grid) {
        // stack spill area (entries kLdsStack.. of every lane), sized for the grid actually launched; growing it
        // waits for the slot's previous kernel, which may still be writing the old one
        if (slot.used) HIP_TRY(hipEventSynchronize(slot.done));
        if (slot.spill) (void)hipFree(slot.spill);
        slot.spill = nullptr;
        slot.spill_waves = 0;
        HIP_TRY(hipMalloc(&slot.spill, (size_t)grid * kWaveScratch * sizeof(uint2)));
        slot.spill_waves = (uint32_t)grid;
    }
    slot.last_stream = stream;
    slot.last_use = ++s->launches;
    p.nodes = s->d_nodes;
    p.tris = s->d_tris;
    p.inst = s->d_inst;
    p.inst_entry = s->d_inst_entry;
    p.inst_xform = s->d_inst_xform;
    p.tlas_start = s->tlas_start;
    p.exp_exact = s->exp_exact;
    p.ctr = slot.ctr;
    p.spill = slot.spill;
    p.tie_first = (sem & TRX_SEM_TIE_FIRST) ? 1u : 0u;
    uint32_t refill = variant & 0x7fu;
    // coherent primary rays: refill a wave only when its whole tile is done (mixing tiles costs more
    // coherence than idle lanes cost; a slot whose frames measure faster with mid-tile refills switches itself, see the
    // kernel's exit protocol); incoherent rays (AO, explicit batches): replace finished rays
    // once 16 lanes idle (whole-tile refills: bistro-class AO pass 1.59 ms in round 1; re-swept with the round-3 kernels,
    // profiles/r03_refill_sweep.log: 12 / 16 / 20 idle lanes = 0.880 / 0.886 / 0.884 ms bistro-class, 0.864 / 0.860 / 0.851
    // hairball-class, 1.382 / 1.369 / 1.383 dense, 0.402 / 0.383 / 0.389 kitchen-class)
    p.refill_idle = refill ? std::min(refill, 64u) : (mode == kModePrimary ? 64u : 16u);
    // fused frames, queues dry: lanes whose primary ray has hit wait for this many of their kind before the wave runs
    // the AO ray set-up for them (tuning: variant bits 14..15)
    {
        static const uint32_t pend[4] = {8u, 1u, 16u, 32u};
        p.pend_min = pend[(variant >> 14) & 3u];
    }
    if (p.n_frames > 1 && mode == kModePrimary) p.refill_idle = 64u; // the kernel takes the frame of a wave from its (whole) tile
    p.variant = variant;
#ifdef TRX_DEV_TUNE
    {   // development builds only (make KFLAGS=-DTRX_DEV_TUNE): experiment switches of kernels.hip, some of which
        // produce wrong results on purpose (ablation timing); the product has no such environment variable
        const char *tune = getenv("TRX_TUNE");
        p.tune = tune ? (uint32_t)strtoul(tune, nullptr, 0) : 0u;
    }
#endif
    p.n_tris = (uint32_t)s->n_tris;
    p.n_nodes = (uint32_t)s->n_nodes;
    {   // (kernels.hip, div_uniform)
        auto rcp32 = [](uint32_t d) -> uint32_t { return d <= 1u ? 0xffffffffu : (uint32_t)((1ull << 32) / d); };
        p.rcp_tiles_x = rcp32(p.tiles_x);
        p.rcp_width = rcp32(p.width);
        p.rcp_tiles_per_frame = rcp32(p.tiles_per_frame);
        p.rcp_n_frames = rcp32(p.n_frames);
    }
    {   // tuning: variant bits 25..27 = compaction threshold (0 = default, 7 = never); bit 28 = no thin waves (A/B runs)
        const uint32_t c = (variant >> 25) & 0x7u;
#ifndef TRX_THIN_MAX_DEFAULT
#define TRX_THIN_MAX_DEFAULT 8u // (tuning builds: 16 with -DTRX_THIN_LEVELS=2, 32 with 3)
#endif
        p.thin_max = ((variant >> 28) & 1u) ? 0u : TRX_THIN_MAX_DEFAULT;
#ifdef TRX_DEV_TUNE
        {   // (development builds: TRX_THIN_MAX = 0 / 8 / 16 / 32)
            const char *tm = getenv("TRX_THIN_MAX");
            if (tm) p.thin_max = (uint32_t)strtoul(tm, nullptr, 0);
        }
#endif
        // a lane's first kTriBatch triangles go in one per-lane round: the scans are only worth computing beyond that
        p.tri_compact_min = c == 0u ? (uint32_t)(s->tlas ? kTriBatchTlas : kTriBatch) + 1u : c == 7u ? 0xffffffffu : c;
        // tuning: variant bits 29..31 = per-lane rounds one cooperative round is worth (0 = default 2; 7 = always cooperative)
        const uint32_t r = (variant >> 29) & 0x7u;
        p.tri_coop_ratio = r == 0u ? 2u : r == 7u ? 0u : r;
        // cooperative rounds are chosen when rounds(largest per-lane count) > ratio x windows, and there is at least one
        // window: a wave whose largest count is at most ratio x kTriBatch can never choose them, so it need not run the two
        // wave scans that decide (twelve DPP instructions a trip; a largest count of exactly two is the common case on
        // coherent rays).  Same decisions, fewer scans.
        if (c == 0u && p.tri_coop_ratio != 0u)
            p.tri_compact_min = std::max(p.tri_compact_min, p.tri_coop_ratio * (uint32_t)(s->tlas ? kTriBatchTlas : kTriBatch) + 1u);
    }
    p.waves_per_block = wpb;
    p.merge = merge_default ? 1u : 0u;
    // Decode-once node test on wave-uniform node steps (kernels.hip, node_intersect_dec): every primary pass (two-level
    // scenes since round 5: san-miguel-class 4K frame -3.3 %, profiles/r05_ab_5_tlas.log).
    // With the plane-major table of round 3 it paid only where almost every step is uniform (kitchen-class frame -4 %, 90 %
    // of its steps) and was kept to scenes of up to 32 MiB; with the {near, far} pair tables of round 4 the bistro-class
    // frame (47 % uniform steps) gains 2 % and the dense and hairball-class frames, whose steps rarely are uniform, pay
    // 0.3 % for the test that finds that out (profiles/r04_ab_procs_15_decode_once.log).
    p.uni_decode = mode == kModePrimary ? 1u : 0u;
#ifdef TRX_DEV_TUNE
    if (p.tune & 0x40000u) p.uni_decode = 1u;
    if (p.tune & 0x80000u) p.uni_decode = 0u;
#endif
    p.wave_times = s->d_wave_times;
    p.single_queue = ((variant >> 21) & 1u) | (p.single_queue ? 1u : 0u); // (a caller may ask for it: trx_traverse1's small batches)
    // tile order feedback (image modes, whole-tile refills only)
    // (an AO batch deals its tiles seed by seed within a queue: it has no tile order to learn)
    const bool lpt = mode != kModeRays && mode != kModeFused && p.refill_idle == 64u && !((variant >> 20) & 1u) &&
                     !(mode == kModeAo && p.n_frames > 1);
    // the drain's parking area covers the second wave's parked tile-list entries (lds_pend): a pass that files tiles
    // (whole-tile refills with the order feedback on - reachable for AO through trx_set_kernel_variant) does not merge
    if (lpt) p.merge = 0u;
    const uint32_t n_tiles = (p.n_items + 63u) >> 6;
    uint64_t key = 0;
    if (lpt) {
        // per slot: two sets of {16 bucket counts, 16 lists of n_tiles tile ids}; a frame reads the
        // set the previous frame on this slot wrote and writes the other one
        const uint32_t n_lists = 16 * kLptShards;
        const uint32_t list_cap = n_tiles / 2 + 64; // a list holds ~1/8 of one bucket; overflow only drops the order
        const size_t set_words = n_lists + (size_t)n_lists * list_cap;
        if (2 * set_words > 0xffffffffull) return fail(TRX_ERR_INVALID, "image too large for the tile-order lists");
        Slot::Order &ord = slot.order[mode == kModeAo ? 1 : 0];
        bool fresh = false;
        if (ord.capacity != n_tiles) {
            // (the slot's previous kernel may still be appending to the old lists: wait for it before they go)
            if (ord.lists && slot.used) HIP_TRY(hipEventSynchronize(slot.done));
            if (ord.lists) (void)hipFree(ord.lists);
            ord.lists = nullptr;
            ord.capacity = 0;
            ord.key = 0;
            HIP_TRY(hipMalloc(&ord.lists, 2 * set_words * sizeof(uint32_t)));
            ord.capacity = n_tiles;
            fresh = true;
        }
        key = ((uint64_t)p.width << 40) ^ ((uint64_t)p.height << 20) ^ ((uint64_t)p.shard_count << 8) ^ p.shard_index ^
              ((uint64_t)(mode + 1) << 60) ^ ((uint64_t)p.n_frames << 56);
        uint32_t *set[2] = {ord.lists, ord.lists + set_words};
        // The order was learnt for an image geometry (the key); it is replayed whatever the camera did since.  Round 3
        // first emptied the lists at a camera cut - natural order while the new view is measured - and then measured
        // that choice once the classes were trips instead of durations (profiles/r03_camera_cut.log): a camera turning
        // 5 / 10 / 20 / 45 degrees PER FRAME runs 0.51 / 0.58 / 0.61 / 0.60 ms replaying the previous frame's order
        // against 0.58 / 0.63 / 0.62 / 0.61 ms in natural order, one moving 0.5 / 1 / 2 / 4 m per frame 0.46 / 0.48 /
        // 0.42 / 0.41 ms against 0.61 / 0.61 / 0.52 / 0.50 - a stale order is never worse than none, and far better
        // for any motion a renderer would call continuous.  What a cut (the eye jumped by more than 1 % of the scene's
        // diagonal, the view turned by more than 2 degrees, or the projection changed) still does is restart the
        // schedule tuner, whose choice (ordered / natural order / mid-tile refills) was measured for the old view.
        // (A probe pass that predicts the order of a first frame - one centre ray per tile - was built and measured
        // too: bound by the latency of its longest ray, it costs more than the order gains, profiles/r03_probe_cap.log.)
        // Variant bit 7: every frame runs as the first frame of its geometry (bench.py's first-frame leg).
        const bool no_order = ord.key != key || ((variant >> 7) & 1u);
        bool cut = no_order;
        // (a batched launch is a cut when any of its frames is; the key holds n_frames, so the stored views match in number.
        // The tuner's timings are per LAUNCH SHAPE: a key change - another n_frames included - resets them.)
        for (uint32_t f = 0; f < std::max(p.n_frames, 1u) && !cut; f++) {
            const ViewDev &a = ord.view[f], &b = p.views[f];
            const float ex = a.eye[0] - b.eye[0], ey = a.eye[1] - b.eye[1], ez = a.eye[2] - b.eye[2];
            const float moved2 = ex * ex + ey * ey + ez * ez, lim = 0.01f * s->scene_diag;
            const float turn = a.view_inv[8] * b.view_inv[8] + a.view_inv[9] * b.view_inv[9] + a.view_inv[10] * b.view_inv[10];
            cut = !(moved2 <= lim * lim) || !(turn >= 0.99939f) || std::memcmp(a.proj_inv, b.proj_inv, sizeof(a.proj_inv)) != 0;
        }
        ord.key = key;
        // (variant bit 19: feedback always on, for A/B runs)
        p.fb = (s->dbg_cost || ((variant >> 19) & 1u)) ? nullptr : &slot.ctr->fb[mode == kModeAo ? 1 : 0];
        p.no_order = no_order ? 1u : 0u;
        p.new_view = cut ? 1u : 0u;
#ifdef TRX_DEV_TUNE
        if (p.tune & 0x8000000u) p.no_order = cut ? 1u : 0u; // (A/B: the round-3 first version, natural order after a cut)
#endif
        // A frame whose views are bit for bit those of the previous launch of this kind on the slot replays a complete
        // order as it stands (the kernel decides: it alone knows whether the set it reads is complete) - the order filed
        // by the first frame of a view, frozen, is the fastest one measured and costs no filing (kernels.hip); any other
        // frame (a moving camera, the first frame of a geometry) files a new order while it runs, as before.
        bool same_view = !no_order && ord.have_views;
        for (uint32_t f = 0; f < std::max(p.n_frames, 1u) && same_view; f++)
            same_view = std::memcmp(&ord.view[f], &p.views[f], sizeof(ViewDev)) == 0;
        for (uint32_t f = 0; f < std::max(p.n_frames, 1u); f++) ord.view[f] = p.views[f];
        ord.have_views = true;
        unsigned int *sel = &slot.ctr->lpt_sel[mode == kModeAo ? 1 : 0];
        if (fresh) { // new lists start empty; from then on a frame that files an order empties the set it read
            HIP_TRY(hipMemsetAsync(set[0], 0, n_lists * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(set[1], 0, n_lists * sizeof(uint32_t), stream));
            HIP_TRY(hipMemsetAsync(sel, 0, sizeof(unsigned int), stream));
        }
        p.lpt_sets = ord.lists;
        p.lpt_sel = sel;
        p.lpt_set_words = (uint32_t)set_words;
        p.lpt_cap = list_cap;
        p.same_view = same_view ? 1u : 0u;
        // priority classes over the heaviest-first order (tuning: variant bits 22..24 pick the cuts)
        // measured on bistro-class 1080p: {32,8,2} 0.566 ms, {64,16,4} 0.572, {128,32,8} 0.585, none 0.630
        static const uint32_t cuts[8][3] = {{32, 8, 2}, {0, 0, 0}, {256, 64, 16}, {64, 16, 4}, {512, 128, 32},
                                            {128, 0, 0}, {128, 32, 8}, {1024, 256, 64}};
        const uint32_t *c = cuts[(variant >> 22) & 7u];
        for (int i = 0; i < 3; i++) p.prio_cut[i] = c[i] ? n_tiles / c[i] : 0u;
    }
    if (s->dbg_cost) { // diagnostics: cold tile order, costs / iteration counts into the caller's buffers
        p.no_order = 1u;
#ifdef TRX_DEV_TUNE
        if (p.tune & 0x2000000u) p.no_order = 0u; // (tools/gpu_tail.py: the costs of a frame in its LEARNT order)
#endif
        p.cost = s->dbg_cost;
        p.tile_iters = s->dbg_iters;
    }
    // The pipelined walk (next node's fetch issued under the triangle phase) pays where a node fetch leaves the L2s:
    // incoherent passes over scenes larger than the eight L2s together (measured: hairball-class AO -4..-6 %, dense
    // bistro-class -3 %, a 3 MB kitchen-class scene +4 %; coherent primary rays +-1 %: DESIGN.md section 4).
#ifndef TRX_PIPE_MIN_BYTES
#define TRX_PIPE_MIN_BYTES (32ull << 20) // (tuning builds: 0 = always)
#endif
    bool pipe = mode != kModePrimary && !s->tlas && s->n_nodes * TRX_NODE_BYTES + s->n_tris * sizeof(TriDev) >= (size_t)TRX_PIPE_MIN_BYTES + 1u;
#ifdef TRX_DEV_TUNE
    if (p.tune & 0x1000u) pipe = true;
    if (p.tune & 0x10000u) pipe = false;
#endif
    HIP_TRY(launch_trace(p, mode, s->tlas, sem, count, pipe, grid, stream));
    HIP_TRY(hipEventRecord(slot.done, stream));
    slot.used = true;
    if (ctr_out) *ctr_out = slot.ctr;
    return TRX_OK;
}

int image_params(TraceParams &p, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard) {
    if (!view) return fail(TRX_ERR_INVALID, "view is null");
    if (w == 0 || h == 0) return fail(TRX_ERR_INVALID, "empty image %ux%u", w, h);
    if ((uint64_t)w * h > 0x7fffffffull) return fail(TRX_ERR_INVALID, "image %ux%u too large", w, h);
    if (shard.count == 0) shard.count = 1;
    if (shard.index >= shard.count) return fail(TRX_ERR_INVALID, "shard %u of %u", shard.index, shard.count);
    if (shard.layout > TRX_LAYOUT_SHARD) return fail(TRX_ERR_INVALID, "unknown shard layout %u", shard.layout);
    const uint32_t tx = (w + 7) / 8, ty = (h + 7) / 8;
    const uint64_t tiles = (uint64_t)tx * ty;
    const uint64_t local = tiles > shard.index ? (tiles - shard.index + shard.count - 1) / shard.count : 0;
    p.width = w;
    p.height = h;
    p.tiles_x = tx;
    p.shard_index = shard.index;
    p.shard_count = shard.count;
    p.compact = shard.layout == TRX_LAYOUT_SHARD ? 1u : 0u;
    p.n_items = (uint32_t)(local * 64);
    p.n_frames = 1;
    p.tiles_per_frame = (uint32_t)local;
    p.frame_stride = 0;
    fill_view(view, p.views[0]);
    return TRX_OK;
}

int read_overflow(trx_scene *s, SlotCounters *ctr) {
    unsigned int over = 0;
    HIP_TRY(hipMemcpy(&over, &ctr->overflow, sizeof(over), hipMemcpyDeviceToHost));
    if (over) {
        const unsigned int zero = 0; // (a blocking copy, not hipMemset: see the launch-slot set-up in enqueue())
        HIP_TRY(hipMemcpy(&ctr->overflow, &zero, sizeof(zero), hipMemcpyHostToDevice));
        return fail(TRX_ERR_STACK_OVERFLOW, "%u rays overflowed the %d-entry traversal stack (or the step cap)", over,
                    kLdsStack + kSpillStack);
    }
    (void)s;
    return TRX_OK;
}

} // namespace

extern "C" {

const char *trx_last_error(void) { return g_err.c_str(); }
uint32_t trx_abi_version(void) { return TRX_ABI_VERSION; }

int trx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int trx_device_name(int device, char *buf, size_t buf_len) {
    if (!buf || !buf_len) return fail(TRX_ERR_INVALID, "null buffer");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buf_len, "%s", prop.gcnArchName);
    return TRX_OK;
}

uint32_t trx_tri_format_bytes(uint32_t f) {
    switch (f) {
    case TRX_TRI_F16_24: return 24;
    case TRX_TRI_VERTS_36: return 36;
    case TRX_TRI_EDGES_36: return 36;
    default: return 0;
    }
}

uint32_t trx_shard_tiles(uint32_t w, uint32_t h, trx_shard shard) {
    if (shard.count == 0) shard.count = 1;
    const uint64_t tiles = (uint64_t)((w + 7) / 8) * ((h + 7) / 8);
    if (shard.index >= shard.count || tiles <= shard.index) return 0;
    return (uint32_t)((tiles - shard.index + shard.count - 1) / shard.count);
}

uint32_t trx_set_kernel_variant(uint32_t variant) {
    return g_variant.exchange(variant);
}

int trx_scene_create(const void *bvh_bytes, uint64_t n_nodes, const void *tri_bytes, uint64_t n_tris,
                     uint32_t tri_format, const uint32_t *instance_offsets, uint32_t n_instances,
                     uint32_t tlas_start, int device, trx_scene **out) {
    if (!out) return fail(TRX_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!bvh_bytes || n_nodes == 0) return fail(TRX_ERR_INVALID, "empty node buffer");
    if (n_tris && !tri_bytes) return fail(TRX_ERR_INVALID, "tri_bytes is null");
    if (!trx_tri_format_bytes(tri_format)) return fail(TRX_ERR_INVALID, "unknown tri_format %u", tri_format);
    if (n_nodes >= 0xffffffffull || n_tris >= 0xffffffffull) return fail(TRX_ERR_INVALID, "buffer too large for u32 indices");
    if (n_instances && !instance_offsets) return fail(TRX_ERR_INVALID, "instance_offsets is null");
    if (!n_instances && tlas_start != 0) return fail(TRX_ERR_INVALID, "tlas_start without instances");
    int rc = validate_nodes((const CwbvhNode *)bvh_bytes, n_nodes, n_tris, instance_offsets, n_instances, tlas_start);
    if (rc) return rc;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(TRX_ERR_NO_DEVICE, "no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TRX_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));

    trx_scene *s = new (std::nothrow) trx_scene();
    if (!s) return fail(TRX_ERR_OOM, "host allocation failed");
    s->device = device;
    s->n_nodes = n_nodes;
    s->n_tris = n_tris;
    s->n_inst = n_instances;
    s->tlas_start = tlas_start;
    s->tlas = n_instances > 0;

    std::vector<TriDev> tris(std::max<uint64_t>(n_tris, 1));
    convert_tris(tri_bytes, n_tris, tri_format, tris.data());

    auto cleanup = [&](int code) {
        trx_scene_destroy(s);
        return code;
    };
#define HIP_TRY_S(expr)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? TRX_ERR_OOM : TRX_ERR_NO_DEVICE, "%s failed: %s", \
                                #expr, hipGetErrorString(e_)));                                              \
    } while (0)
    HIP_TRY_S(hipMalloc(&s->d_nodes, n_nodes * TRX_NODE_BYTES));
    HIP_TRY_S(hipMemcpy(s->d_nodes, bvh_bytes, n_nodes * TRX_NODE_BYTES, hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_tris, tris.size() * sizeof(TriDev)));
    HIP_TRY_S(hipMemcpy(s->d_tris, tris.data(), tris.size() * sizeof(TriDev), hipMemcpyHostToDevice));
    uint32_t zero = 0;
    if (n_instances) s->h_inst.assign(instance_offsets, instance_offsets + n_instances);
    HIP_TRY_S(hipMalloc(&s->d_inst, std::max<uint32_t>(n_instances, 4) * sizeof(uint32_t)));
    HIP_TRY_S(hipMemcpy(s->d_inst, n_instances ? instance_offsets : &zero, (n_instances ? n_instances : 1) * sizeof(uint32_t),
                        hipMemcpyHostToDevice));
    HIP_TRY_S(hipEventCreate(&s->ev0));
    HIP_TRY_S(hipEventCreate(&s->ev1));
#undef HIP_TRY_S
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return cleanup(fail(TRX_ERR_NO_DEVICE, "hipGetDeviceProperties failed"));
        s->cu_count = prop.multiProcessorCount;
    }
    s->grid = trace_grid_size(device, 0, s->tlas, 0, false);
    {   // extent of the root node's quantisation frame (255 steps of 2^(e - 127) per axis): the scene's scale
        const CwbvhNode &root = ((const CwbvhNode *)bvh_bytes)[s->tlas ? tlas_start : 0];
        double d2 = 0.0;
        for (int k = 0; k < 3; k++) {
            const double ext = 255.0 * std::ldexp(1.0, (int)root.e[k] - 127);
            d2 += ext * ext;
        }
        s->scene_diag = (float)std::sqrt(d2);
    }
    {   // may the literal-division node test multiply e = 2^(byte - 127) by the ray's 1/d instead of dividing by d?  Exact
        // while the product stays normal: exponent bytes 0 (e = 0) or >= 21 with |1/d| >= 2^-20 (kernels.hip, pow2)
        bool ok = true;
        const CwbvhNode *nodes = (const CwbvhNode *)bvh_bytes;
        for (uint64_t i = 0; i < n_nodes && ok; i++)
            for (int k = 0; k < 3; k++) ok = ok && (nodes[i].e[k] == 0 || nodes[i].e[k] >= 21);
        // ... and may it compute (p - o) / d from the ray's 1/d by one correction step (kernels.hip, div_by_rcp)?  Every
        // component of every node's p is +0 or 2^-36 <= |p| <= 2^59 (no -0, no NaN, nothing tiny or enormous): with the
        // ray's own flag (origin components 0 or in the same range) p - o is then +0 or 2^-59 <= |p - o| <= 2^60
        bool org = ok;
        for (uint64_t i = 0; i < n_nodes && org; i++)
            for (int k = 0; k < 3; k++) {
                uint32_t bits;
                std::memcpy(&bits, &nodes[i].p[k], 4);
                const float a = std::fabs(nodes[i].p[k]);
                org = org && (bits == 0u || (a >= 0x1p-36f && a <= 0x1p59f));
            }
        s->exp_exact = ok ? (org ? 2u : 1u) : 0u;
    }
    if (s->grid <= 0) return cleanup(fail(TRX_ERR_NO_DEVICE, "could not size the persistent grid"));
    *out = s;
    return TRX_OK;
}

void trx_scene_destroy(trx_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_tris) (void)hipFree(s->d_tris);
    if (s->d_inst) (void)hipFree(s->d_inst);
    if (s->d_inst_entry) (void)hipFree(s->d_inst_entry);
    if (s->d_scratch_a) (void)hipFree(s->d_scratch_a);
    if (s->d_scratch_b) (void)hipFree(s->d_scratch_b);
    if (s->d_scratch_ia) (void)hipFree(s->d_scratch_ia);
    if (s->d_scratch_ib) (void)hipFree(s->d_scratch_ib);
    if (s->d_scratch_rays) (void)hipFree(s->d_scratch_rays);
    if (s->d_wave_times) (void)hipFree(s->d_wave_times);
    if (s->d_inst_xform) (void)hipFree(s->d_inst_xform);
    for (Slot &sl : s->slots) {
        if (sl.ctr) (void)hipFree(sl.ctr);
        if (sl.spill) (void)hipFree(sl.spill);
        for (auto &o : sl.order)
            if (o.lists) (void)hipFree(o.lists);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s->comb;
    delete s;
}

uint64_t trx_scene_device_bytes(const trx_scene *s) {
    if (!s) return 0;
    uint64_t bytes = s->n_nodes * TRX_NODE_BYTES + s->n_tris * sizeof(TriDev) + (uint64_t)s->n_inst * 4;
    // launch slots claimed so far: stack spill areas and tile-order lists
    std::lock_guard<std::mutex> lock(const_cast<trx_scene *>(s)->mu);
    for (const Slot &sl : s->slots) {
        bytes += (uint64_t)sl.spill_waves * kWaveScratch * sizeof(uint2);
        for (const auto &o : sl.order)
            if (o.lists) bytes += 2ull * (16 * kLptShards + (uint64_t)16 * kLptShards * (o.capacity / 2 + 64)) * sizeof(uint32_t);
        if (sl.ctr) bytes += sizeof(SlotCounters);
    }
    return bytes;
}
int trx_scene_device(const trx_scene *s) { return s ? s->device : -1; }

int trx_scene_set_geometry_ranges(trx_scene *s, const uint32_t *blas_tri_start, uint32_t n_blas) {
    if (!s || (!blas_tri_start && n_blas)) return fail(TRX_ERR_INVALID, "null argument");
    s->blas_tri_start.assign(blas_tri_start, blas_tri_start + n_blas + (n_blas ? 1 : 0));
    return TRX_OK;
}

// The scene under a launch slot's frozen tile order changed (instances moved, entry nodes replaced): the next frame of
// every slot files a new order even if its camera has not moved - an order learnt for other geometry is only stale, never
// wrong, but a static camera would replay it for ever.  Call with s->mu held.
static void forget_tile_orders(trx_scene *s) {
    for (Slot &sl : s->slots)
        for (auto &o : sl.order) o.have_views = false;
}

// Entry nodes: TLAS primitive k starts its BLAS walk at node entry_nodes[k] of the BLAS at instance_offsets[k] instead
// of node 0, so one BLAS can be referenced as several subtrees (re-braiding: a BLAS whose box spans the scene no longer
// makes every ray enter it at the root).  Validated like the node buffer: an entry must lie inside its BLAS segment.
int trx_scene_set_instance_entry_nodes(trx_scene *s, const uint32_t *entry_nodes, uint32_t n) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu);
    HIP_TRY(hipSetDevice(s->device));
    uint32_t *fresh = nullptr;
    if (entry_nodes && n) {
        if (!s->tlas) return fail(TRX_ERR_INVALID, "entry nodes need a TLAS scene");
        if (n != s->n_inst) return fail(TRX_ERR_INVALID, "%u entry nodes for %u instances", n, s->n_inst);
        std::vector<uint32_t> seg(s->h_inst);
        std::sort(seg.begin(), seg.end());
        seg.erase(std::unique(seg.begin(), seg.end()), seg.end());
        for (uint32_t k = 0; k < n; k++) {
            auto it = std::upper_bound(seg.begin(), seg.end(), s->h_inst[k]);
            const uint32_t seg_end = it == seg.end() ? s->tlas_start : *it;
            if ((uint64_t)s->h_inst[k] + entry_nodes[k] >= seg_end)
                return fail(TRX_ERR_FORMAT, "instance %u: entry node %u leaves its BLAS [%u, %u)", k, entry_nodes[k], s->h_inst[k], seg_end);
        }
        HIP_TRY(hipMalloc(&fresh, (size_t)n * sizeof(uint32_t)));
        const hipError_t e = hipMemcpy(fresh, entry_nodes, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(TRX_ERR_NO_DEVICE, "upload of the entry nodes failed: %s", hipGetErrorString(e));
        }
    }
    uint32_t *old = nullptr;
    {   // swapped under the launch mutex, freed after every kernel enqueued before the swap has drained
        std::lock_guard<std::mutex> lock(s->mu);
        old = s->d_inst_entry;
        s->d_inst_entry = fresh;
        forget_tile_orders(s);
    }
    HIP_TRY(hipDeviceSynchronize());
    if (old) (void)hipFree(old);
    return TRX_OK;
}

// Instance transforms: the TODOs at query_tlas.hlsl:409,433,484 and Traversable::get_instance_transform
// (traversable/src/lib.rs:25-27).  object_to_world: n column-major 4x4 matrices (glam Mat4) in TLAS-primitive order.
int trx_scene_set_instance_transforms(trx_scene *s, const float *object_to_world, uint32_t n) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu);
    HIP_TRY(hipSetDevice(s->device));
    // The new table is built and uploaded first, the pointer is swapped under the launch mutex (enqueue() reads it
    // there, so no launch can pick up a table that is about to be freed), and the old table is freed only after
    // every kernel enqueued before the swap has drained.
    auto swap_table = [&](float4 *fresh) -> int {
        float4 *old = nullptr;
        {
            std::lock_guard<std::mutex> lock(s->mu);
            old = s->d_inst_xform;
            s->d_inst_xform = fresh;
            forget_tile_orders(s);
        }
        HIP_TRY(hipDeviceSynchronize());
        if (old) (void)hipFree(old);
        return TRX_OK;
    };
    if (!object_to_world || n == 0) { // back to identity
        const int rc = swap_table(nullptr);
        if (rc) return rc;
        s->inst_o2w.clear();
        s->inst_w2o.clear();
        return TRX_OK;
    }
    if (!s->tlas) return fail(TRX_ERR_INVALID, "instance transforms need a TLAS scene");
    if (n != s->n_inst) return fail(TRX_ERR_INVALID, "%u transforms for %u instances", n, s->n_inst);
    std::vector<float> w2o((size_t)n * 12);
    for (uint32_t k = 0; k < n; k++) {
        const float *m = object_to_world + (size_t)k * 16; // m[c*4 + r]
        if (m[3] != 0.f || m[7] != 0.f || m[11] != 0.f || m[15] != 1.f)
            return fail(TRX_ERR_INVALID, "instance %u: transform is not affine (last row must be 0 0 0 1)", k);
        // inverse of the affine map in double, rounded once to f32
        const double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
        const double tx = m[12], ty = m[13], tz = m[14];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return fail(TRX_ERR_INVALID, "instance %u: singular transform", k);
        const double r = 1.0 / det;
        const double inv[9] = {(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r,
                               (f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r,
                               (d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r};
        float *o = &w2o[(size_t)k * 12];
        for (int row = 0; row < 3; row++) {
            o[4 * row + 0] = (float)inv[3 * row + 0];
            o[4 * row + 1] = (float)inv[3 * row + 1];
            o[4 * row + 2] = (float)inv[3 * row + 2];
            o[4 * row + 3] = (float)-(inv[3 * row + 0] * tx + inv[3 * row + 1] * ty + inv[3 * row + 2] * tz);
        }
    }
    float4 *d = nullptr;
    HIP_TRY(hipMalloc(&d, w2o.size() * sizeof(float)));
    hipError_t e = hipMemcpy(d, w2o.data(), w2o.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(TRX_ERR_NO_DEVICE, "upload of the instance transforms failed: %s", hipGetErrorString(e));
    }
    const int rc_swap = swap_table(d);
    if (rc_swap) return rc_swap;
    s->inst_o2w.assign(object_to_world, object_to_world + (size_t)n * 16);
    s->inst_w2o.swap(w2o);
    return TRX_OK;
}

int trx_scene_get_instance_transform(const trx_scene *s, uint32_t instance_id, float out_object_to_world[16]) {
    if (!s || !out_object_to_world) return fail(TRX_ERR_INVALID, "null argument");
    if (instance_id >= std::max<uint32_t>(s->n_inst, 1)) return fail(TRX_ERR_INVALID, "instance %u of %u", instance_id, s->n_inst);
    if (s->inst_o2w.empty()) { // Mat4::default(): identity, like the reference (src/cwbvh.rs:163-165,189-191)
        for (int k = 0; k < 16; k++) out_object_to_world[k] = (k % 5 == 0) ? 1.f : 0.f;
        return TRX_OK;
    }
    std::memcpy(out_object_to_world, &s->inst_o2w[(size_t)instance_id * 16], 64);
    return TRX_OK;
}

int trx_scene_get_instance_world_to_object(const trx_scene *s, uint32_t instance_id, float out_rows[12]) {
    if (!s || !out_rows) return fail(TRX_ERR_INVALID, "null argument");
    if (instance_id >= std::max<uint32_t>(s->n_inst, 1)) return fail(TRX_ERR_INVALID, "instance %u of %u", instance_id, s->n_inst);
    if (s->inst_w2o.empty()) {
        static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        std::memcpy(out_rows, ident, 48);
        return TRX_OK;
    }
    std::memcpy(out_rows, &s->inst_w2o[(size_t)instance_id * 12], 48);
    return TRX_OK;
}

// ---- camera: ViewUniform::from_camera, src/main.rs:602-616 -----------------------------
static void mat4_inverse_f32(const float m[16], float out[16]) {
    // cofactor expansion (same scheme as glam's Mat4::inverse), evaluated in f32
    const float a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3], a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7];
    const float a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11], a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15];
    const float b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10;
    const float b03 = a01 * a12 - a02 * a11, b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
    const float b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30, b08 = a20 * a33 - a23 * a30;
    const float b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
    const float det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
    const float r = 1.0f / det;
    out[0] = (a11 * b11 - a12 * b10 + a13 * b09) * r;
    out[1] = (a02 * b10 - a01 * b11 - a03 * b09) * r;
    out[2] = (a31 * b05 - a32 * b04 + a33 * b03) * r;
    out[3] = (a22 * b04 - a21 * b05 - a23 * b03) * r;
    out[4] = (a12 * b08 - a10 * b11 - a13 * b07) * r;
    out[5] = (a00 * b11 - a02 * b08 + a03 * b07) * r;
    out[6] = (a32 * b02 - a30 * b05 - a33 * b01) * r;
    out[7] = (a20 * b05 - a22 * b02 + a23 * b01) * r;
    out[8] = (a10 * b10 - a11 * b08 + a13 * b06) * r;
    out[9] = (a01 * b08 - a00 * b10 - a03 * b06) * r;
    out[10] = (a30 * b04 - a31 * b02 + a33 * b00) * r;
    out[11] = (a21 * b02 - a20 * b04 - a23 * b00) * r;
    out[12] = (a11 * b07 - a10 * b09 - a12 * b06) * r;
    out[13] = (a00 * b09 - a01 * b07 + a02 * b06) * r;
    out[14] = (a31 * b01 - a30 * b03 - a32 * b00) * r;
    out[15] = (a20 * b03 - a21 * b01 + a22 * b00) * r;
}

int trx_view_from_camera(const float eye[3], const float look_at[3], float fov_deg, float width, float height,
                         trx_view *out) {
    if (!eye || !look_at || !out) return fail(TRX_ERR_INVALID, "null argument");
    if (!(width > 0.f) || !(height > 0.f)) return fail(TRX_ERR_INVALID, "bad image size");
    std::memset(out, 0, sizeof(*out));
    const float aspect = width / height;
    const float fov = fov_deg * (3.14159265358979323846f / 180.0f);
    const float f = 1.0f / std::tan(0.5f * fov);
    float proj[16] = {0};
    proj[0] = f / aspect;
    proj[5] = f;
    proj[11] = -1.0f;
    proj[14] = 0.01f;
    mat4_inverse_f32(proj, out->proj_inv);
    float fw[3] = {look_at[0] - eye[0], look_at[1] - eye[1], look_at[2] - eye[2]};
    float fl = std::sqrt(fw[0] * fw[0] + fw[1] * fw[1] + fw[2] * fw[2]);
    if (!(fl > 0.f)) return fail(TRX_ERR_INVALID, "eye == look_at");
    for (float &x : fw) x /= fl;
    // s = normalize(f x up), up = +Y
    float s[3] = {fw[1] * 0.f - fw[2] * 1.f, fw[2] * 0.f - fw[0] * 0.f, fw[0] * 1.f - fw[1] * 0.f};
    float sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    if (!(sl > 0.f)) return fail(TRX_ERR_INVALID, "view direction parallel to +Y");
    for (float &x : s) x /= sl;
    float u[3] = {s[1] * fw[2] - s[2] * fw[1], s[2] * fw[0] - s[0] * fw[2], s[0] * fw[1] - s[1] * fw[0]};
    auto dot = [](const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    float view[16] = {s[0], u[0], -fw[0], 0, s[1], u[1], -fw[1], 0, s[2], u[2], -fw[2], 0,
                      -dot(s, eye), -dot(u, eye), dot(fw, eye), 1};
    mat4_inverse_f32(view, out->view_inv);
    std::memcpy(out->eye, eye, 12);
    return TRX_OK;
}

// ---- tracing: device-resident -----------------------------------------------------------

int trx_trace_primary_inst_dev(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard,
                               uint32_t sem, trx_hit *d_hits, uint32_t *d_inst, void *stream) {
    if (!s || !d_hits) return fail(TRX_ERR_INVALID, "null argument");
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, shard);
    if (rc) return rc;
    p.out = d_hits;
    p.out_inst = d_inst;
    if (p.n_items == 0) return TRX_OK;
    return enqueue(s, p, kModePrimary, sem, false, (hipStream_t)stream, nullptr);
}

int trx_trace_primary_dev(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard,
                          uint32_t sem, trx_hit *d_hits, void *stream) {
    return trx_trace_primary_inst_dev(s, view, w, h, shard, sem, d_hits, nullptr, stream);
}

int trx_trace_primary_batch_dev(trx_scene *s, const trx_view *views, uint32_t n_frames, uint32_t w, uint32_t h,
                                trx_shard shard, uint32_t sem, trx_hit *d_hits, uint64_t frame_stride, void *stream) {
    if (!s || !d_hits || !views) return fail(TRX_ERR_INVALID, "null argument");
    if (n_frames == 0 || n_frames > (uint32_t)kMaxBatchFrames)
        return fail(TRX_ERR_INVALID, "n_frames %u outside 1..%d", n_frames, kMaxBatchFrames);
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, views, w, h, shard);
    if (rc) return rc;
    const uint64_t frame_records = p.compact ? (uint64_t)p.tiles_per_frame * 64 : (uint64_t)w * h;
    if (n_frames > 1 && frame_stride < frame_records)
        return fail(TRX_ERR_INVALID, "frame_stride %llu < %llu records of one frame", (unsigned long long)frame_stride,
                    (unsigned long long)frame_records);
    if ((uint64_t)p.n_items * n_frames > 0x7fffffffull || frame_stride * (n_frames - 1) + frame_records > 0xffffffffull)
        return fail(TRX_ERR_INVALID, "batch of %u frames too large", n_frames);
    for (uint32_t f = 1; f < n_frames; f++) fill_view(&views[f], p.views[f]);
    p.n_frames = n_frames;
    p.frame_stride = (uint32_t)frame_stride;
    p.n_items *= n_frames;
    p.out = d_hits;
    if (p.n_items == 0) return TRX_OK;
    return enqueue(s, p, kModePrimary, sem, false, (hipStream_t)stream, nullptr);
}

int trx_trace_ao_inst_dev(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard, uint32_t sem,
                          uint32_t frame, float ao_eps, const trx_hit *d_primary, const uint32_t *d_primary_inst,
                          trx_hit *d_ao, uint32_t *d_ao_inst, void *stream) {
    if (!s || !d_primary || !d_ao) return fail(TRX_ERR_INVALID, "null argument");
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, shard);
    if (rc) return rc;
    p.primary = d_primary;
    p.primary_inst = d_primary_inst;
    p.out = d_ao;
    p.out_inst = d_ao_inst;
    p.frame = frame;
    p.ao_eps = ao_eps;
    if (p.n_items == 0) return TRX_OK;
    return enqueue(s, p, kModeAo, sem, false, (hipStream_t)stream, nullptr);
}

int trx_trace_ao_dev(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard, uint32_t sem,
                     uint32_t frame, float ao_eps, const trx_hit *d_primary, trx_hit *d_ao, void *stream) {
    return trx_trace_ao_inst_dev(s, view, w, h, shard, sem, frame, ao_eps, d_primary, nullptr, d_ao, nullptr, stream);
}

int trx_trace_frame_dev(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard, uint32_t sem,
                        uint32_t frame, float ao_eps, trx_hit *d_primary, uint32_t *d_primary_inst, trx_hit *d_ao,
                        uint32_t *d_ao_inst, void *stream) {
    if (!s || !d_primary || !d_ao) return fail(TRX_ERR_INVALID, "null argument");
    if (s->tlas || ((g_variant.load(std::memory_order_relaxed) >> 13) & 1u)) {
        // two-level scenes: the two-level walk has no registers to spare for the in-place hand-over (the kernel that
        // contains it spills), so their frame stays two launches on the caller's stream - same records.  (Variant bit 13:
        // every frame this way, for A/B runs.)
        if (s->d_inst_xform && !d_primary_inst)
            return fail(TRX_ERR_INVALID, "this scene has instance transforms: the frame needs d_primary_inst (the AO pass takes "
                                         "the hit normal into world space with the primary pass's instance ids)");
        int rc2 = trx_trace_primary_inst_dev(s, view, w, h, shard, sem, d_primary, d_primary_inst, stream);
        if (rc2) return rc2;
        return trx_trace_ao_inst_dev(s, view, w, h, shard, sem, frame, ao_eps, d_primary, d_primary_inst, d_ao, d_ao_inst, stream);
    }
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, shard);
    if (rc) return rc;
    p.out = d_primary;
    p.out_inst = d_primary_inst;
    p.out_ao = d_ao;
    p.out_ao_inst = d_ao_inst;
    p.frame = frame;
    p.ao_eps = ao_eps;
    if (p.n_items == 0) return TRX_OK;
    return enqueue(s, p, kModeFused, sem, false, (hipStream_t)stream, nullptr);
}

int trx_trace_ao_batch_dev(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard, uint32_t sem,
                           uint32_t frame0, uint32_t n_frames, float ao_eps, const trx_hit *d_primary,
                           const uint32_t *d_primary_inst, trx_hit *d_ao, uint32_t *d_ao_inst, uint64_t frame_stride,
                           void *stream) {
    if (!s || !d_primary || !d_ao) return fail(TRX_ERR_INVALID, "null argument");
    if (n_frames == 0 || n_frames > (uint32_t)kMaxBatchFrames)
        return fail(TRX_ERR_INVALID, "n_frames %u outside 1..%d", n_frames, kMaxBatchFrames);
    if (n_frames == 1)
        return trx_trace_ao_inst_dev(s, view, w, h, shard, sem, frame0, ao_eps, d_primary, d_primary_inst, d_ao, d_ao_inst, stream);
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, shard);
    if (rc) return rc;
    const uint64_t frame_records = p.compact ? (uint64_t)p.tiles_per_frame * 64 : (uint64_t)w * h;
    if (frame_stride < frame_records)
        return fail(TRX_ERR_INVALID, "frame_stride %llu < %llu records of one frame", (unsigned long long)frame_stride,
                    (unsigned long long)frame_records);
    // every queue gets the same number of tickets: the tile count is padded to a multiple of eight (the kernel skips
    // the padding), and the seeds of a tile are consecutive tickets of one queue
    const uint64_t tiles8 = ((uint64_t)p.tiles_per_frame + 7u) & ~7ull;
    if (tiles8 * 64 * n_frames > 0x7fffffffull || frame_stride * (n_frames - 1) + frame_records > 0xffffffffull)
        return fail(TRX_ERR_INVALID, "batch of %u frames too large", n_frames);
    if (p.n_items == 0) return TRX_OK;
    p.n_frames = n_frames;
    p.frame_stride = (uint32_t)frame_stride;
    p.n_items = (uint32_t)(tiles8 * 64 * n_frames);
    p.primary = d_primary;
    p.primary_inst = d_primary_inst;
    p.out = d_ao;
    p.out_inst = d_ao_inst;
    p.frame = frame0;
    p.ao_eps = ao_eps;
    return enqueue(s, p, kModeAo, sem, false, (hipStream_t)stream, nullptr);
}

static int trace_rays_impl(trx_scene *s, const trx_ray *d_rays, uint64_t n, uint32_t sem, trx_hit *d_hits,
                           hipStream_t stream, bool count, SlotCounters **ctr, bool any_hit = false,
                           uint32_t *d_inst = nullptr, uint32_t *over_host = nullptr, bool one_queue = false) {
    // the work queue is 32-bit: split very large batches
    const uint64_t chunk = 1ull << 30;
    for (uint64_t off = 0; off < n; off += chunk) {
        TraceParams p;
        std::memset(&p, 0, sizeof(p));
        p.rays = d_rays + off;
        p.out = any_hit ? reinterpret_cast<trx_hit *>(reinterpret_cast<uint8_t *>(d_hits) + off) : d_hits + off;
        p.any_hit = any_hit ? 1u : 0u;
        p.out_inst = d_inst ? d_inst + off : nullptr;
        p.over_host = over_host;
        p.single_queue = one_queue ? 1u : 0u;
        p.n_items = (uint32_t)std::min(chunk, n - off);
        int rc = enqueue(s, p, kModeRays, sem, count, stream, ctr);
        if (rc) return rc;
    }
    return TRX_OK;
}

int trx_trace_rays_inst_dev(trx_scene *s, const trx_ray *d_rays, uint64_t n, uint32_t sem, trx_hit *d_hits,
                            uint32_t *d_inst, void *stream) {
    if (!s || (n && (!d_rays || !d_hits))) return fail(TRX_ERR_INVALID, "null argument");
    if (n == 0) return TRX_OK;
    return trace_rays_impl(s, d_rays, n, sem, d_hits, (hipStream_t)stream, false, nullptr, false, d_inst);
}

int trx_trace_rays_dev(trx_scene *s, const trx_ray *d_rays, uint64_t n, uint32_t sem, trx_hit *d_hits, void *stream) {
    return trx_trace_rays_inst_dev(s, d_rays, n, sem, d_hits, nullptr, stream);
}

int trx_trace_occluded_dev(trx_scene *s, const trx_ray *d_rays, uint64_t n, uint32_t sem, uint8_t *d_flags, void *stream) {
    if (!s || (n && (!d_rays || !d_flags))) return fail(TRX_ERR_INVALID, "null argument");
    if (n == 0) return TRX_OK;
    return trace_rays_impl(s, d_rays, n, sem, reinterpret_cast<trx_hit *>(d_flags), (hipStream_t)stream, false, nullptr,
                           true);
}

static int finish_count(trx_scene *s, SlotCounters *ctr, trx_stats *stats, uint32_t *hist = nullptr) {
    HIP_TRY(hipEventRecord(s->ev1, nullptr));
    HIP_TRY(hipEventSynchronize(s->ev1));
    SlotCounters c;
    HIP_TRY(hipMemcpy(&c, ctr, sizeof(c), hipMemcpyDeviceToHost));
    SlotCounters z = c;
    z.n_rays = z.n_node = z.n_tri = z.n_hits = 0;
    z.n_wave_node = z.n_wave_tri = 0;
    z.max_stack = 0;
    z.overflow = 0;
    std::memset(z.hist_max, 0, sizeof(z.hist_max));
    std::memset(z.hist_total, 0, sizeof(z.hist_total));
    if (hist) {
        std::memcpy(hist, c.hist_max, sizeof(c.hist_max));
        std::memcpy(hist + 16, c.hist_total, sizeof(c.hist_total));
    }
    HIP_TRY(hipMemcpy(ctr, &z, sizeof(z), hipMemcpyHostToDevice));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    if (stats) {
        stats->n_rays = c.n_rays;
        stats->n_node = c.n_node;
        stats->n_tri = c.n_tri;
        stats->n_hits = c.n_hits;
        stats->max_stack = c.max_stack;
        stats->overflow = c.overflow;
        stats->kernel_ms = ms;
        stats->_pad = 0.f;
        stats->n_wave_node = c.n_wave_node;
        stats->n_wave_tri = c.n_wave_tri;
    }
    if (c.overflow) return fail(TRX_ERR_STACK_OVERFLOW, "%u rays overflowed the traversal stack", c.overflow);
    return TRX_OK;
}

int trx_count_primary(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard, uint32_t sem,
                      trx_hit *d_hits, trx_stats *stats) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    HIP_TRY(hipSetDevice(s->device));
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, shard);
    if (rc) return rc;
    if (!d_hits) {
        // the shard layout addresses local_tile * 64 + k: whole tiles, also where the image ends mid-tile
        rc = ensure_scratch(s, std::max<uint64_t>((uint64_t)w * h, (uint64_t)p.n_items), 0);
        if (rc) return rc;
        d_hits = s->d_scratch_a;
    }
    p.out = d_hits;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    SlotCounters *ctr = nullptr;
    rc = enqueue(s, p, kModePrimary, sem, true, nullptr, &ctr);
    if (rc) return rc;
    return finish_count(s, ctr, stats);
}

// Compulsory footprint of one primary frame: distinct nodes fetched and distinct triangles tested.
int trx_debug_footprint(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, uint64_t *out_nodes,
                        uint64_t *out_tris) {
    if (!s || !out_nodes || !out_tris) return fail(TRX_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu);
    HIP_TRY(hipSetDevice(s->device));
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, trx_shard{0, 1, 0, 0});
    if (rc) return rc;
    rc = ensure_scratch(s, (uint64_t)w * h, 0);
    if (rc) return rc;
    p.out = s->d_scratch_a;
    const size_t nb = s->n_nodes, tb = std::max<uint64_t>(s->n_tris, 1);
    uint8_t *d_marks = nullptr;
    HIP_TRY(hipMalloc(&d_marks, nb + tb));
    hipError_t e = hipMemset(d_marks, 0, nb + tb);
    p.touch_nodes = d_marks;
    p.touch_tris = d_marks + nb;
    SlotCounters *ctr = nullptr;
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipEventRecord(s->ev0, nullptr);
    if (e == hipSuccess) rc = enqueue(s, p, kModePrimary, sem, true, nullptr, &ctr);
    if (e == hipSuccess && !rc) rc = finish_count(s, ctr, nullptr);
    std::vector<uint8_t> host(nb + tb);
    if (e == hipSuccess && !rc) e = hipMemcpy(host.data(), d_marks, nb + tb, hipMemcpyDeviceToHost);
    (void)hipFree(d_marks);
    if (rc) return rc;
    if (e != hipSuccess) return fail(TRX_ERR_NO_DEVICE, "footprint pass failed: %s", hipGetErrorString(e));
    uint64_t n = 0, t = 0;
    for (size_t i = 0; i < nb; i++) n += host[i] != 0;
    for (size_t i = 0; i < s->n_tris; i++) t += host[nb + i] != 0;
    *out_nodes = n;
    *out_tris = t;
    return TRX_OK;
}

int trx_debug_tri_histogram(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, uint32_t out_hist[32]) {
    if (!s || !out_hist) return fail(TRX_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu);
    HIP_TRY(hipSetDevice(s->device));
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, trx_shard{0, 1, 0, 0});
    if (rc) return rc;
    rc = ensure_scratch(s, (uint64_t)w * h, 0);
    if (rc) return rc;
    p.out = s->d_scratch_a;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    SlotCounters *ctr = nullptr;
    rc = enqueue(s, p, kModePrimary, sem, true, nullptr, &ctr);
    if (rc) return rc;
    return finish_count(s, ctr, nullptr, out_hist);
}

int trx_count_ao(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard, uint32_t sem,
                 uint32_t frame, float ao_eps, const trx_hit *d_primary, trx_hit *d_ao, trx_stats *stats) {
    if (!s || !d_primary || !d_ao) return fail(TRX_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    HIP_TRY(hipSetDevice(s->device));
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    int rc = image_params(p, view, w, h, shard);
    if (rc) return rc;
    p.primary = d_primary;
    p.out = d_ao;
    p.frame = frame;
    p.ao_eps = ao_eps;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    SlotCounters *ctr = nullptr;
    rc = enqueue(s, p, kModeAo, sem, true, nullptr, &ctr);
    if (rc) return rc;
    return finish_count(s, ctr, stats);
}

int trx_count_rays(trx_scene *s, const trx_ray *d_rays, uint64_t n, uint32_t sem, trx_hit *d_hits, trx_stats *stats) {
    if (!s || !d_rays || n == 0 || n > (1ull << 30)) return fail(TRX_ERR_INVALID, "bad ray batch");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    HIP_TRY(hipSetDevice(s->device));
    if (!d_hits) {
        int rc = ensure_scratch(s, n, 0);
        if (rc) return rc;
        d_hits = s->d_scratch_a;
    }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    SlotCounters *ctr = nullptr;
    int rc = trace_rays_impl(s, d_rays, n, sem, d_hits, nullptr, true, &ctr);
    if (rc) return rc;
    return finish_count(s, ctr, stats);
}

int trx_scene_check(trx_scene *s, void *stream) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    for (Slot &sl : s->slots) {
        if (!sl.ctr) continue;
        int rc = read_overflow(s, sl.ctr);
        if (rc) return rc;
    }
    return TRX_OK;
}

// ---- tracing: host buffers ------------------------------------------------------------------

int trx_trace_primary(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, trx_hit *out_hits,
                      float *out_ms) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_scratch(s, (uint64_t)w * h, 0);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    rc = trx_trace_primary_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev1, nullptr));
    HIP_TRY(hipEventSynchronize(s->ev1));
    if (out_ms) HIP_TRY(hipEventElapsedTime(out_ms, s->ev0, s->ev1));
    if (out_hits) HIP_TRY(hipMemcpy(out_hits, s->d_scratch_a, (uint64_t)w * h * sizeof(trx_hit), hipMemcpyDeviceToHost));
    return trx_scene_check(s, nullptr);
}

int trx_trace_primary_ao(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, uint32_t frame,
                         float ao_eps, trx_hit *out_primary, trx_hit *out_ao, float *out_ms) {
    return trx_trace_primary_ao_inst(s, view, w, h, sem, frame, ao_eps, out_primary, nullptr, out_ao, nullptr, out_ms);
}

int trx_trace_primary_ao_inst(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, uint32_t frame,
                              float ao_eps, trx_hit *out_primary, uint32_t *out_primary_inst, trx_hit *out_ao,
                              uint32_t *out_ao_inst, float *out_ms) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_scratch(s, (uint64_t)w * h, 0);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    // instance ids travel with the hits whenever the scene has a TLAS (the AO pass needs them once transforms are set)
    rc = trx_trace_primary_inst_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, s->d_scratch_ia, nullptr);
    if (rc) return rc;
    rc = trx_trace_ao_inst_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, frame, ao_eps, s->d_scratch_a, s->d_scratch_ia,
                               s->d_scratch_b, s->d_scratch_ib, nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev1, nullptr));
    HIP_TRY(hipEventSynchronize(s->ev1));
    if (out_ms) HIP_TRY(hipEventElapsedTime(out_ms, s->ev0, s->ev1));
    const uint64_t bytes = (uint64_t)w * h * sizeof(trx_hit);
    if (out_primary) HIP_TRY(hipMemcpy(out_primary, s->d_scratch_a, bytes, hipMemcpyDeviceToHost));
    if (out_ao) HIP_TRY(hipMemcpy(out_ao, s->d_scratch_b, bytes, hipMemcpyDeviceToHost));
    for (int k = 0; k < 2; k++) {
        uint32_t *dst = k ? out_ao_inst : out_primary_inst;
        const uint32_t *src = k ? s->d_scratch_ib : s->d_scratch_ia;
        if (!dst) continue;
        if (src) HIP_TRY(hipMemcpy(dst, src, (uint64_t)w * h * 4, hipMemcpyDeviceToHost));
        else std::memset(dst, 0xff, (uint64_t)w * h * 4); // no TLAS: no instances
    }
    return trx_scene_check(s, nullptr);
}

int trx_trace_rays(trx_scene *s, const trx_ray *rays, uint64_t n, uint32_t sem, trx_hit *out_hits, float *out_ms) {
    return trx_trace_rays_inst(s, rays, n, sem, out_hits, nullptr, out_ms);
}

int trx_trace_rays_inst(trx_scene *s, const trx_ray *rays, uint64_t n, uint32_t sem, trx_hit *out_hits, uint32_t *out_inst,
                        float *out_ms) {
    if (!s || (n && !rays)) return fail(TRX_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    if (n == 0) return TRX_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_scratch(s, n, n);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(s->d_scratch_rays, rays, n * sizeof(trx_ray), hipMemcpyHostToDevice));
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    rc = trx_trace_rays_inst_dev(s, s->d_scratch_rays, n, sem, s->d_scratch_a, out_inst ? s->d_scratch_ia : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev1, nullptr));
    HIP_TRY(hipEventSynchronize(s->ev1));
    if (out_ms) HIP_TRY(hipEventElapsedTime(out_ms, s->ev0, s->ev1));
    if (out_hits) HIP_TRY(hipMemcpy(out_hits, s->d_scratch_a, n * sizeof(trx_hit), hipMemcpyDeviceToHost));
    if (out_inst) {
        if (s->d_scratch_ia) HIP_TRY(hipMemcpy(out_inst, s->d_scratch_ia, n * 4, hipMemcpyDeviceToHost));
        else std::memset(out_inst, 0xff, n * 4);
    }
    return trx_scene_check(s, nullptr);
}

int trx_trace_occluded(trx_scene *s, const trx_ray *rays, uint64_t n, uint32_t sem, uint8_t *out_flags, float *out_ms) {
    if (!s || (n && !rays)) return fail(TRX_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu);
    if (n == 0) return TRX_OK;
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_scratch(s, n, n);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(s->d_scratch_rays, rays, n * sizeof(trx_ray), hipMemcpyHostToDevice));
    HIP_TRY(hipEventRecord(s->ev0, nullptr));
    rc = trx_trace_occluded_dev(s, s->d_scratch_rays, n, sem, reinterpret_cast<uint8_t *>(s->d_scratch_a), nullptr);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(s->ev1, nullptr));
    HIP_TRY(hipEventSynchronize(s->ev1));
    if (out_ms) HIP_TRY(hipEventElapsedTime(out_ms, s->ev0, s->ev1));
    if (out_flags) HIP_TRY(hipMemcpy(out_flags, s->d_scratch_a, n, hipMemcpyDeviceToHost));
    return trx_scene_check(s, nullptr);
}

// {t, global triangle} (+ the TLAS primitive it was found in) as obvhs' RayHit: (geometry_id, primitive_id local to
// that geometry) like src/cwbvh.rs:151-160 when the geometry ranges are known, RayHit::none() for a miss.
static void to_rayhit(const trx_scene *s, const trx_hit h, uint32_t inst, trx_rayhit *out) {
    out->t = h.t;
    out->instance_id = 0xFFFFFFFFu;
    if (h.prim == 0xFFFFFFFFu) { // RayHit::none()
        out->primitive_id = out->geometry_id = 0xFFFFFFFFu;
        return;
    }
    out->primitive_id = h.prim;
    out->geometry_id = 0;
    if (s->blas_tri_start.size() > 1) {
        auto it = std::upper_bound(s->blas_tri_start.begin(), s->blas_tri_start.end(), h.prim);
        uint32_t g = (uint32_t)(it - s->blas_tri_start.begin()) - 1;
        out->geometry_id = g;
        out->primitive_id = h.prim - s->blas_tri_start[g];
        out->instance_id = g;
    }
    if (s->tlas) out->instance_id = inst; // the TLAS primitive the hit was found in
}

int trx_traverse1(trx_scene *s, const trx_ray *ray, uint32_t sem, trx_rayhit *out) {
    if (!s || !ray || !out) return fail(TRX_ERR_INVALID, "null argument");
    if (sem & ~7u) return fail(TRX_ERR_INVALID, "unknown semantics bits 0x%x", sem);
    HIP_TRY(hipSetDevice(s->device));
    std::call_once(s->comb_once, [s]() { s->comb = new (std::nothrow) RayCombiner(s->device); });
    RayCombiner *c = s->comb;
    if (!c || !c->ok) return fail(TRX_ERR_OOM, "single-ray combiner: %s", c ? c->init_err.c_str() : "allocation failed");
    struct Inside { // (counted while inside: a caller spins for its batch only while the callers fit the host's cores)
        std::atomic<int> &n;
        explicit Inside(std::atomic<int> &a) : n(a) { n.fetch_add(1, std::memory_order_relaxed); }
        ~Inside() { n.fetch_sub(1, std::memory_order_relaxed); }
    } inside(c->inside);
    std::unique_lock<std::mutex> lock(c->mu);
    bool leader = false;
    int bi = -1;
    for (;;) {
        if (c->open >= 0) {
            if (c->batch[c->open].sem == sem) {
                bi = c->open;
                break;
            }
            c->cv.wait(lock); // an open batch of another semantics closes within kMaxWait: wait for it rather than mix
            continue;
        }
        for (int i = 0; i < (int)RayCombiner::kBatches && bi < 0; i++)
            if (c->batch[i].state == RayCombiner::Batch::kFree) bi = i;
        if (bi >= 0) {
            RayCombiner::Batch &nb = c->batch[bi];
            nb.state = RayCombiner::Batch::kOpen;
            nb.sem = sem;
            nb.n = 0;
            nb.read.store(0, std::memory_order_relaxed);
            nb.rc = 0;
            c->open = bi;
            leader = true;
            break;
        }
        c->cv.wait(lock); // every batch is in flight or being read: one frees up when its last reader leaves
    }
    RayCombiner::Batch &b = c->batch[bi];
    const uint32_t idx = b.n++;
    b.rays[RayCombiner::slot(idx)] = *ray;
    const uint32_t epoch = b.done_epoch.load(std::memory_order_relaxed);
    if (b.n == RayCombiner::kCap) c->open = -1; // full: closed to later arrivals (its leader notices)
    if (leader) {
        // Wait for company.  Everyone who can still join is inside this function and not attached to another batch: once
        // they are all here (and nobody new has turned up for kQuiet), go; kMaxWait bounds the wait either way.
        const int64_t t0 = now_ns();
        int64_t t_last = t0;
        uint32_t seen = b.n;
        while (c->open == bi) {
            lock.unlock();
            if (c->inside.load(std::memory_order_relaxed) > c->cores) std::this_thread::yield(); // (callers that have no core yet)
            else for (int k = 0; k < 16; k++) cpu_relax();
            lock.lock();
            const int64_t t = now_ns();
            if (b.n != seen) {
                seen = b.n;
                t_last = t;
            }
            uint32_t elsewhere = 0;
            for (int j = 0; j < (int)RayCombiner::kBatches; j++)
                if (j != bi && c->batch[j].state != RayCombiner::Batch::kFree)
                    elsewhere += c->batch[j].n - std::min(c->batch[j].n, c->batch[j].read.load(std::memory_order_relaxed));
            const int expected = c->inside.load(std::memory_order_relaxed) - (int)elsewhere;
            if (((int)b.n >= expected && t - t_last > RayCombiner::kQuietNs) || t - t0 > RayCombiner::kMaxWaitNs) break;
        }
        if (c->open == bi) c->open = -1;
        const uint32_t n = b.n;
        b.state = RayCombiner::Batch::kFlying;
        c->launches++;
        c->rays += n;
        lock.unlock();
        c->cv.notify_all(); // (callers waiting for an open batch of their own semantics)
        *b.over = 0u;
        int rc = trace_rays_impl(s, b.rays, RayCombiner::slots_used(n), sem, b.hits, b.stream, false, nullptr, false,
                                 s->tlas ? b.inst : nullptr, b.over, n <= RayCombiner::kSpread);
        if (!rc && hipStreamSynchronize(b.stream) != hipSuccess) rc = fail(TRX_ERR_NO_DEVICE, "sync failed");
        if (!rc && *reinterpret_cast<volatile uint32_t *>(b.over) != 0u)
            rc = fail(TRX_ERR_STACK_OVERFLOW, "a ray overflowed the %d-entry traversal stack (or the step cap)", kLdsStack + kSpillStack);
        b.rc = rc;
        if (rc) b.err = g_err;
        b.done_epoch.fetch_add(1, std::memory_order_release);
        if (n > 1) futex_wake_all(&b.done_epoch);
    } else {
        // a follower spins on the batch's epoch for about a round trip (when it has a core to spin on), then sleeps on it
        lock.unlock();
        if (c->inside.load(std::memory_order_relaxed) <= c->cores) {
            const int64_t t0 = now_ns();
            while (b.done_epoch.load(std::memory_order_acquire) == epoch && now_ns() - t0 < 300000)
                for (int k = 0; k < 16; k++) cpu_relax();
        }
        while (b.done_epoch.load(std::memory_order_acquire) == epoch) futex_wait(&b.done_epoch, epoch);
    }
    // (no lock: the batch's records stay put until its last reader has left)
    const int rc = b.rc;
    if (rc && !leader) g_err = b.err;
    const uint32_t at = RayCombiner::slot(idx);
    const trx_hit h = b.hits[at];
    const uint32_t inst = s->tlas ? b.inst[at] : 0xFFFFFFFFu;
    if (idx < RayCombiner::kSpread) b.rays[at] = RayCombiner::null_ray(); // (the slot goes back to being padding)
    const uint32_t n_final = b.n;
    if (b.read.fetch_add(1, std::memory_order_acq_rel) + 1 == n_final) { // last reader out: the batch can be opened again
        lock.lock();
        b.state = RayCombiner::Batch::kFree;
        lock.unlock();
        c->cv.notify_all();
    }
    if (rc) return rc;
    to_rayhit(s, h, inst, out);
    return TRX_OK;
}

// The reference's CPU pixel loop over the literal Traversable::traverse (src/rt_cpu/rt_cpu.rs:35-57) as a measuring aid:
// `threads` host threads, thread k calls trx_traverse1 for rays k, k + threads, ...; wall-clock seconds of the loop and the
// launches its calls shared come back.  (The calls are the public entry point's; only the thread pool lives here, so that a
// Python caller is not measuring its interpreter lock.)
int trx_debug_traverse1_threads(trx_scene *s, const trx_ray *rays, uint64_t n, uint32_t threads, uint32_t sem, trx_rayhit *out,
                                double *out_seconds, uint64_t *out_launches) {
    if (!s || (n && (!rays || !out)) || threads == 0 || threads > 4096) return fail(TRX_ERR_INVALID, "bad argument");
    uint64_t l0 = 0, l1 = 0;
    if (n) { // the first call creates the combiner: not part of the loop's time
        const int rc = trx_traverse1(s, &rays[0], sem, &out[0]);
        if (rc) return rc;
    }
    (void)trx_debug_traverse1_stats(s, &l0, nullptr);
    std::atomic<int> first_rc{0};
    std::string first_err;
    std::mutex err_mu;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    try {
        for (uint32_t k = 0; k < threads; k++)
            pool.emplace_back([&, k]() {
                for (uint64_t i = k; i < n && first_rc.load(std::memory_order_relaxed) == 0; i += threads) {
                    const int rc = trx_traverse1(s, &rays[i], sem, &out[i]);
                    if (rc) {
                        std::lock_guard<std::mutex> g(err_mu);
                        if (first_rc.load() == 0) {
                            first_err = g_err;
                            first_rc.store(rc);
                        }
                    }
                }
            });
    } catch (const std::exception &) {
        first_rc.store(TRX_ERR_OOM);
        first_err = "could not start the threads";
    }
    for (auto &th : pool) th.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)trx_debug_traverse1_stats(s, &l1, nullptr);
    if (first_rc.load()) {
        g_err = first_err;
        return first_rc.load();
    }
    if (out_seconds) *out_seconds = secs;
    if (out_launches) *out_launches = l1 - l0;
    return TRX_OK;
}

// Launches and rays the single-ray combiner has served so far (development / tests: rays / launches = callers per launch).
int trx_debug_traverse1_stats(trx_scene *s, uint64_t *out_launches, uint64_t *out_rays) {
    if (!s) return fail(TRX_ERR_INVALID, "null argument");
    uint64_t l = 0, r = 0;
    if (s->comb) {
        std::lock_guard<std::mutex> lock(s->comb->mu);
        l = s->comb->launches;
        r = s->comb->rays;
    }
    if (out_launches) *out_launches = l;
    if (out_rays) *out_rays = r;
    return TRX_OK;
}

// Traversable::traverse for a batch: one launch, then the same {t, prim} -> RayHit mapping as trx_traverse1.
int trx_traverse_batch(trx_scene *s, const trx_ray *rays, uint64_t n, uint32_t sem, trx_rayhit *out, float *out_ms) {
    if (!s || (n && (!rays || !out))) return fail(TRX_ERR_INVALID, "null argument");
    if (n == 0) return TRX_OK;
    std::vector<trx_hit> hits;
    std::vector<uint32_t> inst;
    try {
        hits.resize(n);
        if (s->tlas) inst.resize(n);
    } catch (const std::exception &) {
        return fail(TRX_ERR_OOM, "host allocation failed");
    }
    int rc = trx_trace_rays_inst(s, rays, n, sem, hits.data(), s->tlas ? inst.data() : nullptr, out_ms);
    if (rc) return rc;
    for (uint64_t i = 0; i < n; i++) to_rayhit(s, hits[i], s->tlas ? inst[i] : 0xFFFFFFFFu, &out[i]);
    return TRX_OK;
}

int trx_bench_primary(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, uint32_t warmup,
                      uint32_t frames, float *out_min_ms, float *out_mean_ms) {
    if (!s || frames == 0) return fail(TRX_ERR_INVALID, "bad argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_scratch(s, (uint64_t)w * h, 0);
    if (rc) return rc;
    for (uint32_t i = 0; i < warmup; i++) {
        rc = trx_trace_primary_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, nullptr);
        if (rc) return rc;
    }
    float mn = 1e30f;
    double sum = 0.0;
    for (uint32_t i = 0; i < frames; i++) {
        HIP_TRY(hipEventRecord(s->ev0, nullptr));
        rc = trx_trace_primary_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, nullptr);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s->ev1, nullptr));
        HIP_TRY(hipEventSynchronize(s->ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
        mn = std::min(mn, ms);
        sum += ms;
    }
    if (out_min_ms) *out_min_ms = mn;
    if (out_mean_ms) *out_mean_ms = (float)(sum / frames);
    return trx_scene_check(s, nullptr);
}

// Diagnostics: per-tile cost (100 MHz ticks, from a normal frame) and per-tile wave-level
// iteration counts ((node steps << 16) | triangle rounds, from a counting frame), cold tile order.
int trx_debug_tile_profile(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem,
                           uint32_t *out_cost, uint32_t *out_iters, uint32_t n_tiles) {
    if (!s || !out_cost || !out_iters) return fail(TRX_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    if (n_tiles != ((w + 7) / 8) * ((h + 7) / 8)) return fail(TRX_ERR_INVALID, "n_tiles does not match the image");
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_scratch(s, (uint64_t)w * h, 0);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMalloc(&s->dbg_cost, (size_t)n_tiles * 4));
    hipError_t e = hipMalloc(&s->dbg_iters, (size_t)n_tiles * 4);
    if (e == hipSuccess) e = hipMemset(s->dbg_iters, 0, (size_t)n_tiles * 4);
    trx_stats st;
    if (e == hipSuccess) rc = trx_count_primary(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, &st);
    if (e == hipSuccess && !rc) e = hipMemcpy(out_iters, s->dbg_iters, (size_t)n_tiles * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && !rc) {
        uint32_t *iters = s->dbg_iters;
        s->dbg_iters = nullptr; // second pass: the normal kernel
#ifdef TRX_DEV_TUNE
        { const char *tune = getenv("TRX_TUNE"); if (tune && (strtoul(tune, nullptr, 0) & 0x2000000u)) s->dbg_iters = iters; } // (diag builds: trips / rounds per tile)
#endif
        for (int i = 0; i < 3 && !rc; i++) rc = trx_trace_primary_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, nullptr);
        s->dbg_iters = iters;
        if (!rc) e = hipDeviceSynchronize();
        if (e == hipSuccess && !rc) e = hipMemcpy(out_cost, s->dbg_cost, (size_t)n_tiles * 4, hipMemcpyDeviceToHost);
#ifdef TRX_DEV_TUNE
        if (e == hipSuccess && !rc && s->dbg_iters) e = hipMemcpy(out_iters, s->dbg_iters, (size_t)n_tiles * 4, hipMemcpyDeviceToHost);
#endif
    }
    (void)hipDeviceSynchronize();
    (void)hipFree(s->dbg_cost);
    if (s->dbg_iters) (void)hipFree(s->dbg_iters);
    s->dbg_cost = s->dbg_iters = nullptr;
    if (rc) return rc;
    if (e != hipSuccess) return fail(TRX_ERR_NO_DEVICE, "tile profile failed: %s", hipGetErrorString(e));
    return TRX_OK;
}

// Diagnostics: per-wave records of one primary frame: [start, end] wall-clock stamps (100 MHz ticks) and, in
// TRX_STAMPS builds, the shader cycles spent in {refill, node fetch, node test, triangle phase, pop} + loop trips.
static int wave_records(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, uint64_t *out,
                        uint32_t fields, uint32_t max_waves, uint32_t *out_waves, bool ao = false) {
    if (!s || !out || !out_waves) return fail(TRX_ERR_INVALID, "null argument");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the shared scratch / events
    HIP_TRY(hipSetDevice(s->device));
    int rc = ensure_scratch(s, (uint64_t)w * h, 0);
    if (rc) return rc;
    const size_t n = (size_t)s->cu_count * 32;
    const size_t words = n * kWaveTimeStride;
    if (ao) { // the AO pass is the one recorded: its input first, without records
        rc = trx_trace_primary_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, nullptr);
        if (rc) return rc;
    }
    HIP_TRY(hipDeviceSynchronize());
    if (!s->d_wave_times) HIP_TRY(hipMalloc(&s->d_wave_times, words * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->d_wave_times, 0, words * sizeof(unsigned long long)));
    if (ao)
        rc = trx_trace_ao_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, 0u, 0.01f, s->d_scratch_a, s->d_scratch_b, nullptr);
    else
        rc = trx_trace_primary_dev(s, view, w, h, trx_shard{0, 1, 0, 0}, sem, s->d_scratch_a, nullptr);
    hipError_t e = hipDeviceSynchronize();
    std::vector<unsigned long long> host(words);
    if (e == hipSuccess) e = hipMemcpy(host.data(), s->d_wave_times, words * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(s->d_wave_times);
    s->d_wave_times = nullptr;
    if (rc) return rc;
    if (e != hipSuccess) return fail(TRX_ERR_NO_DEVICE, "timeline read-back failed: %s", hipGetErrorString(e));
    uint32_t k = 0;
    for (size_t i = 0; i < n && k < max_waves; i++)
        if (host[kWaveTimeStride * i]) {
            for (uint32_t f = 0; f < fields; f++) out[(size_t)fields * k + f] = host[kWaveTimeStride * i + f];
            k++;
        }
    *out_waves = k;
    return TRX_OK;
}

int trx_debug_wave_timeline(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem,
                            uint64_t *out_times, uint32_t max_waves, uint32_t *out_waves) {
    return wave_records(s, view, w, h, sem, out_times, 2, max_waves, out_waves);
}

int trx_debug_wave_timeline_ao(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem,
                               uint64_t *out_records, uint32_t max_waves, uint32_t *out_waves) {
    return wave_records(s, view, w, h, sem, out_records, (uint32_t)kWaveTimeStride, max_waves, out_waves, true);
}

int trx_debug_wave_phases(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem,
                          uint64_t *out_records, uint32_t max_waves, uint32_t *out_waves) {
    return wave_records(s, view, w, h, sem, out_records, (uint32_t)kWaveTimeStride, max_waves, out_waves);
}

// ---- host side: builder ----------------------------------------------------------------------

int trx_bvh_build_tris(const float *verts, uint64_t n, uint32_t max_prims, int threads, trx_bvh **out) {
    if (!out || (n && !verts)) return fail(TRX_ERR_INVALID, "null argument");
    if (max_prims < 1 || max_prims > 3)
        return fail(TRX_ERR_INVALID, "CWBVH only supports a maximum of 3 primitives per leaf."); // src/main.rs:176-178
    if (n >= 0x7fffffffull) return fail(TRX_ERR_INVALID, "too many primitives");
    trx_bvh *b = new (std::nothrow) trx_bvh();
    if (!b) return fail(TRX_ERR_OOM, "host allocation failed");
    const BuildParams bp = to_build_params(build_settings(), max_prims, threads);
    try {
        build_cwbvh_from_tris(verts, n, bp, b->bvh);
    } catch (const std::runtime_error &e) { // the GPU build stage reports its own failures
        delete b;
        return fail(TRX_ERR_NO_DEVICE, "%s", e.what());
    } catch (const std::exception &) {
        delete b;
        return fail(TRX_ERR_OOM, "out of memory building the BVH");
    }
    *out = b;
    return TRX_OK;
}

int trx_bvh_build_aabbs(const float *aabbs, uint64_t n, uint32_t max_prims, int threads, trx_bvh **out) {
    if (!out || (n && !aabbs)) return fail(TRX_ERR_INVALID, "null argument");
    if (max_prims < 1 || max_prims > 3) return fail(TRX_ERR_INVALID, "CWBVH only supports a maximum of 3 primitives per leaf.");
    if (n >= 0x7fffffffull) return fail(TRX_ERR_INVALID, "too many primitives");
    trx_bvh *b = new (std::nothrow) trx_bvh();
    if (!b) return fail(TRX_ERR_OOM, "host allocation failed");
    BuildParams bp = to_build_params(build_settings(), max_prims, threads);
    bp.reinsertion_batch_ratio = 0.f; // boxes of instances: see trx_flat_build
    try {
        build_cwbvh_from_aabbs((const Aabb *)aabbs, n, bp, b->bvh);
    } catch (const std::runtime_error &e) { // the GPU build stage reports its own failures
        delete b;
        return fail(TRX_ERR_NO_DEVICE, "%s", e.what());
    } catch (const std::exception &) {
        delete b;
        return fail(TRX_ERR_OOM, "out of memory building the BVH");
    }
    *out = b;
    return TRX_OK;
}

int trx_set_build_costs(float traversal_cost, float prim_cost) {
    if (!(traversal_cost > 0.f) || !(prim_cost > 0.f)) return fail(TRX_ERR_INVALID, "costs must be positive");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.traversal_cost = traversal_cost;
    g_build.prim_cost = prim_cost;
    return TRX_OK;
}

int trx_set_build_device(int device) {
    if (device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n) return fail(TRX_ERR_NO_DEVICE, "no HIP device %d for the build stage", device);
    }
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.ploc_device = device < 0 ? -1 : device;
    return TRX_OK;
}

int trx_set_build_split(float extra_ratio) {
    if (!(extra_ratio >= 0.f) || extra_ratio > 4.f) return fail(TRX_ERR_INVALID, "split: extra reference ratio in [0, 4]");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.pre_split = extra_ratio;
    return TRX_OK;
}

int trx_set_build_rebraid(float area_fraction) {
    if (!(area_fraction >= 0.f) || area_fraction > 1.f) return fail(TRX_ERR_INVALID, "rebraid: area fraction in [0, 1]");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.rebraid_area = area_fraction;
    return TRX_OK;
}

int trx_set_build_preset(const char *name) {
    // {bins, sweep, reinsertion ratio, iterations, pre-split}: build time against tree quality, like the obvhs
    // presets (which switch pre_split on from slow_build upwards); "" restores the defaults
    struct Preset { const char *name; int bins; uint32_t sweep; float ratio; int iters; float split; };
    static const Preset presets[] = {
        {"fastest_build", 8, 0, 0.0f, 0, 0.0f},   {"very_fast_build", 16, 8, 0.01f, 1, 0.0f}, {"fast_build", 16, 24, 0.02f, 2, 0.0f},
        {"medium_build", 32, 48, 0.02f, 4, 0.0f}, {"slow_build", 32, 64, 0.05f, 6, 0.3f},     {"very_slow_build", 32, 64, 0.15f, 8, 0.3f},
        {"", 32, 48, 0.02f, 4, 0.0f},
    };
    if (!name) return fail(TRX_ERR_INVALID, "preset is null");
    for (const Preset &p : presets) {
        if (std::strcmp(name, p.name) == 0) {
            std::lock_guard<std::mutex> lock(g_build_mu);
            g_build.sah_bins = p.bins;
            g_build.sweep_max = p.sweep;
            g_build.reinsert_ratio = p.ratio;
            g_build.reinsert_iters = p.iters;
            g_build.pre_split = p.split;
            return TRX_OK;
        }
    }
    return fail(TRX_ERR_INVALID, "unknown preset '%s'", name);
}

int trx_set_build_reinsertion(float batch_ratio, int iterations) {
    if (!(batch_ratio >= 0.f) || batch_ratio > 1.f || iterations < 0)
        return fail(TRX_ERR_INVALID, "reinsertion: ratio in [0,1], iterations >= 0");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.reinsert_ratio = batch_ratio;
    g_build.reinsert_iters = iterations;
    return TRX_OK;
}

int trx_set_build_reinsertion_batches(int whole_iterations) {
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.reinsert_whole = whole_iterations != 0;
    return TRX_OK;
}

void trx_bvh_destroy(trx_bvh *b) { delete b; }
uint64_t trx_bvh_node_count(const trx_bvh *b) { return b ? b->bvh.nodes.size() : 0; }
uint64_t trx_bvh_prim_count(const trx_bvh *b) { return b ? b->bvh.primitive_indices.size() : 0; }
const void *trx_bvh_nodes(const trx_bvh *b) { return b ? b->bvh.nodes.data() : nullptr; }
const uint32_t *trx_bvh_primitive_indices(const trx_bvh *b) { return b ? b->bvh.primitive_indices.data() : nullptr; }
void trx_bvh_total_aabb(const trx_bvh *b, float out6[6]) {
    if (!b || !out6) return;
    std::memcpy(out6, b->bvh.total_aabb.mn, 12);
    std::memcpy(out6 + 3, b->bvh.total_aabb.mx, 12);
}
double trx_bvh_build_seconds(const trx_bvh *b) { return b ? b->bvh.build_seconds : 0.0; }

static int flat_build_impl(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                           uint32_t max_prims, int threads, const BuildSettings &settings, trx_flat **out,
                           const uint32_t *instance_object = nullptr, const float *instance_o2w = nullptr,
                           uint32_t n_instances = 0);

// BvhBuildParams of the reference (src/main.rs:571-585) for one build: the BVH2 comes from PLOC with the caller's
// search distance, depth threshold and Morton width, is optimised by the reinsertion pass at the caller's batch ratio
// and collapsed at the caller's traversal cost.  post_collapse_reinsertion_batch_ratio_multiplier is "For BVH2 only"
// in the reference's own words (src/main.rs:119-123): a CWBVH build has no such pass.
int trx_flat_build_params(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                          const trx_build_params *bp, int threads, trx_flat **out) {
    if (!bp) return fail(TRX_ERR_INVALID, "build params are null");
    if (bp->sort_precision != 64 && bp->sort_precision != 128) return fail(TRX_ERR_INVALID, "Unsupported sort precision");
    if (!(bp->reinsertion_batch_ratio >= 0.f) || !(bp->collapse_traversal_cost > 0.f))
        return fail(TRX_ERR_INVALID, "reinsertion_batch_ratio >= 0 and collapse_traversal_cost > 0 required");
    BuildSettings b = build_settings(); // bins / sweep threshold stay those of the current preset
    b.traversal_cost = bp->collapse_traversal_cost;
    // obvhs: 0..1 is the candidate ratio of one pass, above 1 the whole set is evaluated several times
    b.reinsert_ratio = std::min(bp->reinsertion_batch_ratio, 1.0f);
    b.reinsert_iters = bp->reinsertion_batch_ratio > 1.f ? (int)std::ceil(bp->reinsertion_batch_ratio)
                       : bp->reinsertion_batch_ratio > 0.f ? std::max(1, b.reinsert_iters) : 0; // passes: this library's (4)
    b.pre_split = bp->pre_split ? 0.3f : 0.0f;
    if (bp->ploc_search_distance < 1 || bp->ploc_search_distance > 32)
        return fail(TRX_ERR_INVALID, "ploc_search_distance %u outside 1..32", bp->ploc_search_distance);
    b.ploc_distance = bp->ploc_search_distance;
    b.ploc_depth_threshold = bp->search_depth_threshold;
    b.ploc_sort_bits = bp->sort_precision;
    b.reinsert_batched = true; // the parallel reinsertion pass, as in the reference's builder
    return flat_build_impl(verts, object_tri_counts, n_objects, use_tlas, bp->max_prims_per_leaf, threads, b, out);
}

void trx_build_params_default(trx_build_params *bp) {
    if (!bp) return;
    // the defaults of the reference's command line (src/main.rs:85-124,158-163)
    bp->pre_split = 0;
    bp->ploc_search_distance = 14;
    bp->search_depth_threshold = 2;
    bp->reinsertion_batch_ratio = 0.15f;
    bp->sort_precision = 64;
    bp->max_prims_per_leaf = 3;
    bp->post_collapse_reinsertion_batch_ratio_multiplier = 0.0f;
    bp->collapse_traversal_cost = 1.0f;
}

// cwbvh_gpu_runner, src/rt_gpu/mod.rs:16-112
int trx_flat_build(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                   uint32_t max_prims, int threads, trx_flat **out) {
    return flat_build_impl(verts, object_tri_counts, n_objects, use_tlas, max_prims, threads, build_settings(), out);
}

// One BLAS per object and a TLAS over INSTANCES of them: instance k places object instance_object[k] with the
// affine object-to-world matrix instance_object_to_world + 16 k (column-major; NULL = identity for all).  The TLAS
// boxes bound the transformed BLAS boxes; trx_flat.instance_transforms / instance_source come back in
// TLAS-primitive order, ready for trx_scene_create + trx_scene_set_instance_transforms.
int trx_flat_build_instanced(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects,
                             const uint32_t *instance_object, const float *instance_object_to_world, uint32_t n_instances,
                             uint32_t max_prims, int threads, trx_flat **out) {
    if (!instance_object || n_instances == 0) return fail(TRX_ERR_INVALID, "no instances");
    for (uint32_t k = 0; k < n_instances; k++) {
        if (instance_object[k] >= n_objects) return fail(TRX_ERR_INVALID, "instance %u names object %u of %u", k, instance_object[k], n_objects);
        if (object_tri_counts && object_tri_counts[instance_object[k]] == 0) return fail(TRX_ERR_INVALID, "instance %u names an empty object", k);
    }
    return flat_build_impl(verts, object_tri_counts, n_objects, 1, max_prims, threads, build_settings(), out, instance_object,
                           instance_object_to_world, n_instances);
}

static int flat_build_impl(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                           uint32_t max_prims, int threads, const BuildSettings &settings, trx_flat **out,
                           const uint32_t *instance_object, const float *instance_o2w, uint32_t n_instances) {
    if (!out || !object_tri_counts || n_objects == 0) return fail(TRX_ERR_INVALID, "null argument");
    if (max_prims < 1 || max_prims > 3) return fail(TRX_ERR_INVALID, "CWBVH only supports a maximum of 3 primitives per leaf.");
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_objects; i++) total += object_tri_counts[i];
    if (total && !verts) return fail(TRX_ERR_INVALID, "verts is null");
    if (total >= 0x7fffffffull) return fail(TRX_ERR_INVALID, "too many triangles");
    const BuildParams bp = to_build_params(settings, max_prims, threads);
    try {
        // without --tlas everything is flattened into the first object (src/main.rs:300-308)
        std::vector<uint64_t> counts;
        std::vector<uint32_t> blas_of_object(n_objects, 0xFFFFFFFFu); // objects without triangles have no BLAS
        if (use_tlas) {
            for (uint32_t i = 0; i < n_objects; i++)
                if (object_tri_counts[i]) {
                    blas_of_object[i] = (uint32_t)counts.size();
                    counts.push_back(object_tri_counts[i]);
                }
            if (counts.empty()) counts.push_back(0);
        } else {
            counts.push_back(total);
        }
        std::vector<CwbvhNode> nodes;
        std::vector<uint32_t> blas_offset, blas_tri_start;
        std::vector<Aabb> blas_aabb;
        // The three arrays with one entry per triangle reference go straight into the buffers the caller receives (malloc:
        // no value-initialisation, no copy at the end - 250 MB each way on a 3.9 M triangle scene, a tenth of a second of
        // one core); their size is known once the BLASes are built.
        struct Grow { // entries filled so far / capacity, in triangle references
            float *tri = nullptr, *box = nullptr;
            uint32_t *src = nullptr;
            size_t n = 0, cap = 0;
            ~Grow() {
                std::free(tri);
                std::free(box);
                std::free(src);
            }
        } refs;
        double blas_s = 0.0, tlas_s = 0.0;
        // BLAS builds: large objects one after the other with every thread, the (many) small ones of a
        // TLAS scene concurrently with one thread each; assembly below stays in object order
        std::vector<CwBvh> built(counts.size());
        std::vector<uint64_t> firsts(counts.size());
        {
            uint64_t f0 = 0;
            for (size_t i = 0; i < counts.size(); i++) { firsts[i] = f0; f0 += counts[i]; }
            const auto t0 = std::chrono::steady_clock::now();
            int nthreads = threads > 0 ? threads : usable_threads();
            if (nthreads < 1) nthreads = 1;
            // objects the GPU stage would take (trx_set_build_device: >= kDevicePlocMinPrims primitives) are never
            // handed to the one-thread host pool
            const uint64_t kSmall = bp.ploc_device >= 0 ? std::min<uint64_t>(65536, kDevicePlocMinPrims - 1) : 65536;
            std::vector<size_t> small;
            for (size_t i = 0; i < counts.size(); i++) {
                if (counts[i] > kSmall || counts.size() == 1 || nthreads == 1)
                    build_cwbvh_from_tris(verts + firsts[i] * 9, counts[i], bp, built[i]);
                else
                    small.push_back(i);
            }
            if (!small.empty()) {
                BuildParams one = bp;
                one.threads = 1;
                one.ploc_device = -1; // the many small BLASes of a TLAS scene stay on the host cores
                std::atomic<size_t> next{0};
                std::mutex err_mu;
                std::exception_ptr first_error; // rethrown as it was (a builder failure is not an allocation failure)
                auto worker = [&]() {
                    try {
                        for (size_t k = next.fetch_add(1); k < small.size(); k = next.fetch_add(1)) {
                            const size_t i = small[k];
                            build_cwbvh_from_tris(verts + firsts[i] * 9, counts[i], one, built[i]);
                        }
                    } catch (...) {
                        std::lock_guard<std::mutex> g(err_mu);
                        if (!first_error) first_error = std::current_exception();
                        next.store(small.size()); // the other workers stop taking objects
                    }
                };
                std::vector<std::thread> pool;
                const int n = (int)std::min<size_t>((size_t)nthreads, small.size());
                for (int t = 0; t < n; t++) pool.emplace_back(worker);
                for (auto &th : pool) th.join();
                if (first_error) std::rethrow_exception(first_error);
            }
            blas_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        {
            size_t all = 0;
            for (const CwBvh &b : built) all += b.primitive_indices.size();
            refs.cap = std::max<size_t>(all, 1);
            refs.tri = (float *)std::malloc(refs.cap * 36);
            refs.box = (float *)std::malloc(refs.cap * 24);
            refs.src = (uint32_t *)std::malloc(refs.cap * 4);
            if (!refs.tri || !refs.box || !refs.src) throw std::bad_alloc();
        }
        uint64_t first = 0;
        for (size_t bi = 0; bi < counts.size(); bi++) {
            const uint64_t cnt = counts[bi];
            CwBvh &bvh = built[bi];
            const uint32_t tri_offset = (uint32_t)refs.n;
            blas_tri_start.push_back(tri_offset);
            // permute triangles into primitive_indices order (mod.rs:38-43); the entries are independent, so a large
            // BLAS is filled by every core
            {
                const size_t np = bvh.primitive_indices.size();
                const size_t r0 = refs.n;
                refs.n += np;
                auto fill = [&](size_t k0, size_t k1) {
                    for (size_t k = k0; k < k1; k++) {
                        const uint32_t pi = bvh.primitive_indices[k];
                        const float *v = verts + (first + pi) * 9;
                        std::memcpy(refs.tri + (r0 + k) * 9, v, 36);
                        refs.src[r0 + k] = (uint32_t)(first + pi);
                        // the box this entry was built with: the triangle's own, or its clipped part after pre-splitting
                        float *bx = refs.box + (r0 + k) * 6;
                        if (!bvh.primitive_boxes.empty()) {
                            for (int a = 0; a < 3; a++) { bx[a] = bvh.primitive_boxes[k].mn[a]; bx[3 + a] = bvh.primitive_boxes[k].mx[a]; }
                        } else {
                            for (int a = 0; a < 3; a++) {
                                bx[a] = std::min(v[a], std::min(v[3 + a], v[6 + a]));
                                bx[3 + a] = std::max(v[a], std::max(v[3 + a], v[6 + a]));
                            }
                        }
                    }
                };
                const int nt = (int)std::min<size_t>((size_t)std::max(1, threads > 0 ? threads : usable_threads()), np / 65536 + 1);
                if (nt <= 1) {
                    fill(0, np);
                } else {
                    std::vector<std::thread> pool;
                    for (int t = 0; t < nt; t++) pool.emplace_back(fill, np * t / nt, np * (t + 1) / nt);
                    for (auto &th : pool) th.join();
                }
            }
            // global triangle buffer: offset primitive_base_idx (mod.rs:44-48)
            for (CwbvhNode &n : bvh.nodes) n.primitive_base_idx += tri_offset;
            blas_offset.push_back((uint32_t)nodes.size());
            blas_aabb.push_back(bvh.total_aabb);
            nodes.insert(nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
            first += cnt;
        }
        blas_tri_start.push_back((uint32_t)refs.n);
        std::vector<uint32_t> inst, inst_source, inst_entry;
        std::vector<float> inst_xf;
        uint32_t tlas_start = 0;
        if (use_tlas) {
            // what the TLAS is built over: one box per BLAS (the reference, src/cwbvh.rs:114), or one box per
            // instance = the BLAS box carried to world space by the instance's transform, padded by a few ulps of
            // its magnitude (the ray is taken to object space by the rounded INVERSE, which does not commute
            // exactly with transforming the box forward)
            std::vector<Aabb> tlas_boxes = blas_aabb;
            // Re-braiding (own builder only; the reference builds its TLAS over whole BLAS boxes, src/cwbvh.rs:114): a
            // BLAS whose box is large against the scene is referenced through the subtrees under its root instead -
            // repeatedly, largest box first, as long as the node has inner children only (a leaf child's triangles
            // could not be reached through any subtree) - so a floor or a shell that spans the scene stops making
            // every ray walk it from the root.  Each such TLAS primitive carries the node its walk starts at.
            std::vector<uint32_t> prim_blas, prim_entry;
            if (!instance_object && settings.rebraid_area > 0.f && counts.size() > 1) {
                auto area = [](const Aabb &b) {
                    const double dx = std::max(0.0, (double)b.mx[0] - b.mn[0]), dy = std::max(0.0, (double)b.mx[1] - b.mn[1]),
                                 dz = std::max(0.0, (double)b.mx[2] - b.mn[2]);
                    return 2.0 * (dx * dy + dy * dz + dz * dx);
                };
                Aabb scene_box = blas_aabb[0];
                for (const Aabb &b : blas_aabb)
                    for (int a = 0; a < 3; a++) { scene_box.mn[a] = std::min(scene_box.mn[a], b.mn[a]); scene_box.mx[a] = std::max(scene_box.mx[a], b.mx[a]); }
                const double limit = (double)settings.rebraid_area * area(scene_box);
                struct Item { double a; uint32_t blas, entry; Aabb box; };
                auto less = [](const Item &x, const Item &y) { return x.a < y.a || (x.a == y.a && (x.blas > y.blas || (x.blas == y.blas && x.entry > y.entry))); };
                std::priority_queue<Item, std::vector<Item>, decltype(less)> heap(less);
                std::vector<Item> final_items;
                for (uint32_t b = 0; b < (uint32_t)blas_aabb.size(); b++) heap.push(Item{area(blas_aabb[b]), b, 0u, blas_aabb[b]});
                // (measured on the san-miguel-class scene, three views, profiles/r03_rebraid_tlas_variants.log: 16 K / 64 K /
                // 160 K / 225 K primitives = 3.44 / 2.99 / 3.15 / 4.51 ms against 4.73 ms unopened; beyond ~160 K the
                // TLAS - a plain binned-SAH tree without the BLAS builder's reinsertion pass - becomes the worse upper tree)
                const size_t max_prims_tlas = blas_aabb.size() + 262144;
                // // (instance ids stay far below 2^24 triangle-group indices)
                while (!heap.empty()) {
                    Item it = heap.top();
                    heap.pop();
                    // (the BLAS nodes were moved into `nodes`; BLAS b starts at blas_offset[b])
                    const CwbvhNode &n = nodes[(size_t)blas_offset[it.blas] + it.entry];
                    bool openable = it.a > limit && n.imask != 0 && heap.size() + final_items.size() + 8 <= max_prims_tlas;
                    for (int sl = 0; sl < 8 && openable; sl++)
                        if (n.child_meta[sl] != 0 && (n.child_meta[sl] & 0x18) != 0x18) openable = false; // a leaf child
                    if (!openable) {
                        final_items.push_back(it);
                        continue;
                    }
                    uint32_t rank = 0;
                    for (int sl = 0; sl < 8; sl++) {
                        if (!((n.imask >> sl) & 1u)) continue;
                        // the child's quantised box, decoded exactly (24-bit origin + 8-bit step count x a power of two
                        // fits a double) and rounded outwards to f32, clipped to the box it was opened from
                        Aabb cb;
                        const uint8_t *qlo[3] = {n.child_min_x, n.child_min_y, n.child_min_z}, *qhi[3] = {n.child_max_x, n.child_max_y, n.child_max_z};
                        for (int a = 0; a < 3; a++) {
                            const double ex = std::ldexp(1.0, (int)n.e[a] - 127);
                            const double lo = (double)n.p[a] + qlo[a][sl] * ex, hi = (double)n.p[a] + qhi[a][sl] * ex;
                            float flo = (float)lo, fhi = (float)hi;
                            if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
                            if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
                            cb.mn[a] = std::max(flo, it.box.mn[a]);
                            cb.mx[a] = std::min(fhi, it.box.mx[a]);
                        }
                        heap.push(Item{area(cb), it.blas, n.child_base_idx + rank, cb});
                        rank++;
                    }
                }
                if (final_items.size() > blas_aabb.size()) {
                    // deterministic order: by BLAS, then entry node
                    std::sort(final_items.begin(), final_items.end(), [](const Item &x, const Item &y) { return x.blas < y.blas || (x.blas == y.blas && x.entry < y.entry); });
                    tlas_boxes.clear();
                    for (const Item &it : final_items) {
                        tlas_boxes.push_back(it.box);
                        prim_blas.push_back(it.blas);
                        prim_entry.push_back(it.entry);
                    }
                }
            }
            if (instance_object) {
                tlas_boxes.assign(n_instances, Aabb{});
                for (uint32_t k = 0; k < n_instances; k++) {
                    const Aabb &bb = blas_aabb[blas_of_object[instance_object[k]]];
                    Aabb wb;
                    for (int a = 0; a < 3; a++) { wb.mn[a] = 3.402823466e+38f; wb.mx[a] = -3.402823466e+38f; }
                    for (int c = 0; c < 8; c++) {
                        const float p[3] = {c & 1 ? bb.mx[0] : bb.mn[0], c & 2 ? bb.mx[1] : bb.mn[1], c & 4 ? bb.mx[2] : bb.mn[2]};
                        float q[3] = {p[0], p[1], p[2]};
                        if (instance_o2w) {
                            const float *m = instance_o2w + (size_t)k * 16;
                            for (int r = 0; r < 3; r++) q[r] = m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r];
                        }
                        for (int a = 0; a < 3; a++) { wb.mn[a] = std::min(wb.mn[a], q[a]); wb.mx[a] = std::max(wb.mx[a], q[a]); }
                    }
                    for (int a = 0; a < 3; a++) {
                        const float pad = 1e-5f * (std::max(std::fabs(wb.mn[a]), std::fabs(wb.mx[a])) + (wb.mx[a] - wb.mn[a])) + 1e-30f;
                        wb.mn[a] -= pad;
                        wb.mx[a] += pad;
                    }
                    tlas_boxes[k] = wb;
                }
            }
            // TLAS over the BLAS boxes (src/cwbvh.rs:114,132); instance table in TLAS
            // primitive order (mod.rs:72-78); TLAS nodes appended last (mod.rs:88-99)
            CwBvh tlas;
            // no reinsertion pass over instance boxes: the SAH's constant leaf cost misprices an instance
            // (a whole BLAS traversal), and the pass measured worse there (san-miguel-class stand-in:
            // 62.9 -> 67.4 node visits per ray with it, 62.7 with the pass in the BLASes only)
            BuildParams bpt = bp;
            bpt.reinsertion_batch_ratio = 0.f;
            // ... and an instance is dearer than a node visit: cost 3 instead of 0.3 keeps one instance per leaf
            // slot, each with its own quantised box (same scene: 62.7 -> 60.6 node visits per ray)
            bpt.prim_cost = std::max(bpt.prim_cost, 3.0f);
            build_cwbvh_from_aabbs(tlas_boxes.data(), tlas_boxes.size(), bpt, tlas);
            tlas_s = tlas.build_seconds;
            for (uint32_t pi : tlas.primitive_indices) {
                if (!prim_blas.empty()) { // re-braided: TLAS primitive pi is the subtree at node prim_entry[pi] of BLAS prim_blas[pi]
                    inst.push_back(blas_offset[prim_blas[pi]]);
                    inst_source.push_back(prim_blas[pi]);
                    inst_entry.push_back(prim_entry[pi]);
                    continue;
                }
                inst.push_back(blas_offset[instance_object ? blas_of_object[instance_object[pi]] : pi]);
                inst_source.push_back(pi);
                if (instance_o2w) inst_xf.insert(inst_xf.end(), instance_o2w + (size_t)pi * 16, instance_o2w + (size_t)pi * 16 + 16);
            }
            tlas_start = (uint32_t)nodes.size();
            nodes.insert(nodes.end(), tlas.nodes.begin(), tlas.nodes.end());
        }
        trx_flat *f = (trx_flat *)std::calloc(1, sizeof(trx_flat));
        if (!f) return fail(TRX_ERR_OOM, "host allocation failed");
        auto dup = [](const void *src, size_t bytes) -> void * {
            void *p = std::malloc(bytes ? bytes : 1);
            if (p && bytes) std::memcpy(p, src, bytes);
            return p;
        };
        f->n_nodes = nodes.size();
        f->bvh_bytes = dup(nodes.data(), nodes.size() * sizeof(CwbvhNode));
        f->n_tris = refs.n;
        f->tri_verts = refs.tri; // (handed over: see `refs`)
        refs.tri = nullptr;
        f->n_instances = (uint32_t)inst.size();
        f->instance_offsets = (uint32_t *)dup(inst.data(), inst.size() * 4);
        f->tlas_start = tlas_start;
        f->tri_source = refs.src;
        f->tri_boxes = refs.box;
        refs.src = nullptr;
        refs.box = nullptr;
        f->n_blas = (uint32_t)counts.size();
        f->blas_tri_start = (uint32_t *)dup(blas_tri_start.data(), blas_tri_start.size() * 4);
        f->blas_build_s = blas_s;
        f->tlas_build_s = tlas_s;
        f->instance_source = (uint32_t *)dup(inst_source.data(), inst_source.size() * 4);
        f->instance_transforms = inst_xf.empty() ? nullptr : (float *)dup(inst_xf.data(), inst_xf.size() * 4);
        f->instance_entry_nodes = inst_entry.empty() ? nullptr : (uint32_t *)dup(inst_entry.data(), inst_entry.size() * 4);
        if (!f->instance_source || (!inst_xf.empty() && !f->instance_transforms) || (!inst_entry.empty() && !f->instance_entry_nodes)) {
            trx_flat_destroy(f);
            return fail(TRX_ERR_OOM, "host allocation failed");
        }
        if (!f->bvh_bytes || !f->tri_verts || !f->instance_offsets || !f->tri_source || !f->blas_tri_start || !f->tri_boxes) {
            trx_flat_destroy(f);
            return fail(TRX_ERR_OOM, "host allocation failed");
        }
        *out = f;
    } catch (const std::runtime_error &e) { // the GPU build stage reports its own failures
        return fail(TRX_ERR_NO_DEVICE, "%s", e.what());
    } catch (const std::exception &) {
        return fail(TRX_ERR_OOM, "out of memory building the scene");
    }
    return TRX_OK;
}

void trx_flat_destroy(trx_flat *f) {
    if (!f) return;
    std::free(f->bvh_bytes);
    std::free(f->tri_verts);
    std::free(f->instance_offsets);
    std::free(f->tri_source);
    std::free(f->blas_tri_start);
    std::free(f->tri_boxes);
    std::free(f->instance_source);
    std::free(f->instance_transforms);
    std::free(f->instance_entry_nodes);
    std::free(f);
}

// ---- host side: scenes ---------------------------------------------------------------------------

static int export_mesh(std::vector<float> &verts, std::vector<uint64_t> &objects, float **out_verts,
                       uint64_t *out_n, uint64_t **out_counts, uint32_t *out_nobj) {
    uint64_t n = verts.size() / 9;
    float *v = (float *)std::malloc(std::max<size_t>(verts.size() * 4, 4));
    uint64_t *c = (uint64_t *)std::malloc(std::max<size_t>(objects.size() * 8, 8));
    if (!v || !c) {
        std::free(v);
        std::free(c);
        return fail(TRX_ERR_OOM, "host allocation failed");
    }
    if (!verts.empty()) std::memcpy(v, verts.data(), verts.size() * 4);
    if (!objects.empty()) std::memcpy(c, objects.data(), objects.size() * 8);
    *out_verts = v;
    *out_n = n;
    if (out_counts) *out_counts = c;
    else std::free(c);
    if (out_nobj) *out_nobj = (uint32_t)objects.size();
    return TRX_OK;
}

int trx_gen_scene(const char *name, uint64_t n_tris, uint64_t seed, float **out_verts, uint64_t *out_n,
                  uint64_t **out_counts, uint32_t *out_nobj) {
    if (!name || !out_verts || !out_n) return fail(TRX_ERR_INVALID, "null argument");
    std::vector<float> verts;
    std::vector<uint64_t> objects;
    try {
        if (!gen_scene(name, n_tris, seed, verts, objects)) return fail(TRX_ERR_INVALID, "unknown scene '%s'", name);
    } catch (const std::exception &) {
        return fail(TRX_ERR_OOM, "out of memory generating '%s'", name);
    }
    return export_mesh(verts, objects, out_verts, out_n, out_counts, out_nobj);
}

int trx_scene_camera(const char *name, float eye[3], float look_at[3], float *fov) {
    if (!name || !eye || !look_at || !fov) return fail(TRX_ERR_INVALID, "null argument");
    if (!scene_camera(name, eye, look_at, fov)) return fail(TRX_ERR_INVALID, "unknown scene '%s'", name);
    return TRX_OK;
}

int trx_load_model(const char *path, float **out_verts, uint64_t *out_n, uint64_t **out_counts, uint32_t *out_nobj) {
    if (!path || !out_verts || !out_n) return fail(TRX_ERR_INVALID, "null argument");
    std::vector<float> verts;
    std::vector<uint64_t> objects;
    try {
        if (!load_model(path, verts, objects)) return fail(TRX_ERR_IO, "Error while loading model file \"%s\"", path);
    } catch (const std::exception &) {
        return fail(TRX_ERR_OOM, "out of memory loading '%s'", path);
    }
    return export_mesh(verts, objects, out_verts, out_n, out_counts, out_nobj);
}

void trx_free(void *p) { std::free(p); }

} // extern "C"
