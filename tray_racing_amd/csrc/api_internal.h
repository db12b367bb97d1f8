// api_internal.h - what the translation units behind include/trx.h share: the scene object, its launch slots, the
// single-ray combiner, error reporting and the launch path (api_launch.cpp).  Not installed; not part of the ABI.
//   api.cpp          errors, devices, scene upload and its setters, the camera
//   api_launch.cpp   enqueue(): launch slots, kernel parameters, tile-order state, the schedule tuner's host side
//   api_trace.cpp    the trace / count / bench / diagnostic entry points (device-resident and host-buffer forms)
//   api_traverse.cpp Traversable::traverse for one ray (concurrent callers share launches) and for batches
//   api_build.cpp    builders, flat-buffer assembly (cwbvh_gpu_runner's host half), scene generators and loaders
//   probe.cpp        trx_debug_fetch_rate: the measured ceiling of the node-fetch loop on a scene's buffers
#ifndef TRX_API_INTERNAL_H
#define TRX_API_INTERNAL_H
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

namespace trxapi {

// thread-local error string of trx_last_error(); returns `code`
int fail(int code, const char *fmt, ...);
std::string &err_string(); // this thread's error string itself (a launch made for one thread reports to another: api_traverse.cpp)
extern std::atomic<uint32_t> g_variant; // tuning aid (trx_set_kernel_variant), read once per launch

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
    bool pinned = false;               // a resident kernel (the ray service) runs on this slot: never recycled for another stream
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

} // namespace trxapi

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
    int waiting = 0;                         // ... of which wait on cv (another semantics' batch is open, or no batch is free): mu held
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

// trx_frame_loop's streams, events and device buffers (created on first use)
struct FrameLoop {
    static constexpr int kBuffers = 4; // primary-hit buffers: the primary passes may run this many frames ahead of the AO passes
    hipStream_t stream[2] = {nullptr, nullptr};
    hipEvent_t prim_done[kBuffers] = {}, ao_done[kBuffers] = {}, t0 = nullptr, t1 = nullptr;
    trx_hit *prim[kBuffers] = {}, *ao = nullptr;
    uint32_t *prim_inst[kBuffers] = {}, *ao_inst = nullptr;
    uint64_t records = 0;
};

// trx_traverse1 over single-level scenes: a RESIDENT kernel answers the callers' rays out of a ring in pinned host memory
// (kernels.h, kSvcRays; kernel side in trace_service.inc) - no launch, no stream synchronisation per ray.  A caller claims
// a slot, writes its ray as three 16-byte granules that carry a fresh sequence number, and spins on the slot's answer
// word.  The kernel is started by the first call and stopped by a watchdog thread after kIdleStopNs without a call (or by
// trx_scene_destroy), so that device-wide synchronisations elsewhere in the process wait for it no longer than that; the
// same thread bumps the heartbeat without which the kernel leaves by itself.
struct RayService {
    static constexpr uint32_t kGroups = 16, kSlots = kGroups * trx::kSvcRays;
    static constexpr int64_t kIdleStopNs = 50 * 1000 * 1000, kBeatNs = 2 * 1000 * 1000, kGiveUpNs = 5LL * 1000 * 1000 * 1000;
    trx_scene *scene = nullptr;
    uint32_t sem = 0;
    uint32_t *ring = nullptr, *ctl = nullptr; // pinned, device-visible
    hipStream_t stream = nullptr;
    std::mutex mu;                    // start / stop
    std::atomic<bool> running{false};
    // Per slot, on cache lines of its own: everything a call writes on the host side.  (Until late in round 6 every call
    // bumped seven process-wide words - a count of callers inside, the last use, four statistics - and the slots' busy
    // flags sat sixteen to a line: from ten threads on the calls queued for those lines, 2.2-2.5 us apiece system-wide,
    // whatever the walk took - 0.40-0.47 Mrays/s was that queue, not the GPU.)
    struct alignas(128) Caller {
        std::atomic<uint32_t> busy{0};     // a caller owns the slot
        uint32_t seq = 0;                  // last sequence number used on the slot (its current owner's to bump)
        std::atomic<int64_t> last_use_ns{0};
        std::atomic<uint64_t> rays{0}, walk_ticks{0}, walk_trips{0}, call_ns{0}; // statistics (trx_debug_service_stats)
    };
    Caller caller[kSlots];
    std::thread watchdog;
    std::atomic<bool> quit{false};
    bool ok = false;
    std::string init_err;
    std::atomic<uint64_t> starts{0};
    bool idle_since(int64_t t_ns) const { // no slot owned, none used since t_ns
        for (const Caller &c : caller)
            if (c.busy.load(std::memory_order_acquire) != 0u || c.last_use_ns.load(std::memory_order_relaxed) > t_ns) return false;
        return true;
    }
    uint64_t sum(std::atomic<uint64_t> Caller::*field) const {
        uint64_t n = 0;
        for (const Caller &c : caller) n += (c.*field).load(std::memory_order_relaxed);
        return n;
    }
    RayService(trx_scene *s, uint32_t semantics);
    ~RayService();
    int start_locked();  // mu held
    void stop_locked();  // mu held
};

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
    trxapi::Slot slots[trxapi::kSlots];
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
    FrameLoop loop;                // trx_frame_loop
    RayCombiner *comb = nullptr;   // trx_traverse1 over two-level scenes: concurrent single-ray callers share launches (created on first use)
    std::once_flag comb_once;
    RayService *svc[8] = {};       // trx_traverse1 over single-level scenes: one resident kernel per semantics word in use
    std::mutex svc_mu;             // creation of the services
    // instance transforms (TLAS scenes): object-to-world as given (get_instance_transform), world-to-object rows as
    // the kernels use them, and their device copy; empty / null = identity
    std::vector<float> inst_o2w, inst_w2o;
    float4 *d_inst_xform = nullptr;
};

struct trx_bvh {
    CwBvh bvh;
};

namespace trxapi {

int ensure_scratch(trx_scene *s, uint64_t hits, uint64_t rays);
void fill_view(const trx_view *v, trx::ViewDev &out);
// Enqueues one traversal kernel on a launch slot of the scene (api_launch.cpp)
int enqueue(trx_scene *s, trx::TraceParams &p, int mode, uint32_t sem, bool count, hipStream_t stream, trx::SlotCounters **ctr_out);
int image_params(trx::TraceParams &p, const trx_view *view, uint32_t w, uint32_t h, trx_shard shard);
int read_overflow(trx_scene *s, trx::SlotCounters *ctr);
// explicit rays (api_trace.cpp; trx_traverse1's batches launch through it)
int trace_rays_impl(trx_scene *s, const trx_ray *d_rays, uint64_t n, uint32_t sem, trx_hit *d_hits, hipStream_t stream, bool count,
                    trx::SlotCounters **ctr, bool any_hit = false, uint32_t *d_inst = nullptr, uint32_t *over_host = nullptr,
                    bool one_queue = false);

} // namespace trxapi

using namespace trxapi;

#endif // TRX_API_INTERNAL_H
