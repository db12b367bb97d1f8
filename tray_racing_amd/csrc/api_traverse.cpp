// api_traverse.cpp - Traversable::traverse (traversable/src/lib.rs:13-28) for one ray and for batches.
#include "api_internal.h"

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

// ---- the ray service (single-level scenes): host side ----------------------------------------------------------------
#include <emmintrin.h>

// Every live service, so that a process that exits without destroying its scenes still stops its resident kernels first
// (the HIP runtime's own teardown would otherwise meet a kernel that waits for a heartbeat nobody sends any more).
static std::mutex g_services_mu;
static std::vector<RayService *> g_services;
static void stop_services_at_exit() {
    std::lock_guard<std::mutex> lock(g_services_mu);
    for (RayService *v : g_services) {
        v->quit.store(true, std::memory_order_release);
        if (v->watchdog.joinable()) v->watchdog.join();
        std::lock_guard<std::mutex> l2(v->mu);
        v->stop_locked();
    }
    g_services.clear();
}

RayService::RayService(trx_scene *s, uint32_t semantics) : scene(s), sem(semantics) {
    hipError_t e = hipHostMalloc((void **)&ring, (size_t)kSlots * kSvcSlotWords * 4, hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctl, 64, hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        init_err = hipGetErrorString(e);
        return;
    }
    std::memset(ring, 0, (size_t)kSlots * kSvcSlotWords * 4);
    std::memset(ctl, 0, 64);
    caller[0].last_use_ns.store(now_ns(), std::memory_order_relaxed);
    try {
        watchdog = std::thread([this]() {
            (void)hipSetDevice(scene->device);
            int64_t beat = 0;
            while (!quit.load(std::memory_order_acquire)) {
                std::this_thread::sleep_for(std::chrono::nanoseconds(kBeatNs));
                reinterpret_cast<volatile uint32_t *>(ctl)[1] = (uint32_t)++beat;
                // (a caller that claims a slot at this very moment finds the service stopped and starts it again - its request
                // is still in the ring: traverse1_service looks every 1024 spins)
                if (running.load(std::memory_order_acquire) && idle_since(now_ns() - kIdleStopNs)) {
                    std::lock_guard<std::mutex> lock(mu);
                    if (idle_since(now_ns() - kIdleStopNs)) stop_locked();
                }
            }
        });
    } catch (const std::exception &) {
        init_err = "could not start the service's watchdog thread";
        return;
    }
    {
        std::lock_guard<std::mutex> lock(g_services_mu);
        static bool registered = false;
        if (!registered) {
            registered = true;
            std::atexit(stop_services_at_exit);
        }
        g_services.push_back(this);
    }
    ok = true;
}

RayService::~RayService() {
    {
        std::lock_guard<std::mutex> lock(g_services_mu);
        g_services.erase(std::remove(g_services.begin(), g_services.end(), this), g_services.end());
    }
    quit.store(true, std::memory_order_release);
    if (watchdog.joinable()) watchdog.join();
    {
        std::lock_guard<std::mutex> lock(mu);
        stop_locked();
    }
    if (stream) (void)hipStreamDestroy(stream);
    if (ring) (void)hipHostFree(ring);
    if (ctl) (void)hipHostFree(ctl);
}

int RayService::start_locked() {
    if (running.load(std::memory_order_acquire)) return TRX_OK;
    reinterpret_cast<volatile uint32_t *>(ctl)[0] = 0u;
    TraceParams p;
    std::memset(&p, 0, sizeof(p));
    p.n_items = 2u * kGroups * 64u; // (a chunk per wave: enqueue() sizes the grid from it - kGroups workgroups of two waves)
    p.n_frames = 1;
    p.svc_ring = ring;
    p.svc_ctl = ctl;
    const int rc = enqueue(scene, p, kModeService, sem, false, stream, nullptr);
    if (rc) return rc;
    running.store(true, std::memory_order_release);
    starts.fetch_add(1, std::memory_order_relaxed);
    return TRX_OK;
}

void RayService::stop_locked() {
    if (!running.load(std::memory_order_acquire)) return;
    (void)hipSetDevice(scene->device);
    reinterpret_cast<volatile uint32_t *>(ctl)[0] = 1u;
    (void)hipStreamSynchronize(stream);
#ifdef TRX_SVC_PHASES   // (tuning builds, trace_thin.inc: cycles per step of the walkers' trips)
    for (uint32_t g = 0; g < kGroups; g++) {
        const volatile uint32_t *w = ring + (size_t)g * trx::kSvcRays * trx::kSvcSlotWords + 24;
        if (w[7] != 0u)
            fprintf(stderr, "SVC_PHASES group %u: %u trips; cycles per trip: (0) %.0f (1) %.0f (2) %.0f (3) %.0f (4) %.0f (5) %.0f back-edge %.0f\n", g, w[7],
                    (double)w[0] / w[7], (double)w[1] / w[7], (double)w[2] / w[7], (double)w[3] / w[7], (double)w[4] / w[7], (double)w[5] / w[7], (double)w[6] / w[7]);
    }
#endif
    {
        std::lock_guard<std::mutex> lock(scene->mu);
        for (Slot &sl : scene->slots)
            if (sl.used && sl.last_stream == stream) sl.pinned = false;
    }
    running.store(false, std::memory_order_release);
}

static int traverse1_service(trx_scene *s, const trx_ray *ray, uint32_t sem, trx_rayhit *out);

extern "C" {

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
    // the resident ray service (no launch per ray).  (Two-level scenes went through the launch combiner below until the
    // thin walk learned its two levels, late in round 6; TRX_TRAVERSE1_COMBINER=1 in the environment still sends them there.)
    static const bool combiner_for_tlas = [] { const char *e = std::getenv("TRX_TRAVERSE1_COMBINER"); return e && e[0] == '1'; }();
    if (!s->tlas || !combiner_for_tlas) return traverse1_service(s, ray, sem, out);
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
            // an open batch of another semantics closes within kMaxWait: wait for it rather than mix.  (Counted: such a caller
            // is inside trx_traverse1 but will not join the open batch, so its leader must not expect it.)
            c->waiting++;
            c->cv.wait(lock);
            c->waiting--;
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
        c->waiting++;
        c->cv.wait(lock); // every batch is in flight or being read: one frees up when its last reader leaves
        c->waiting--;
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
            const int expected = c->inside.load(std::memory_order_relaxed) - (int)elsewhere - c->waiting;
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
        if (rc) b.err = err_string();
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
    if (rc && !leader) err_string() = b.err;
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
                            first_err = err_string();
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
        err_string() = first_err;
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
    for (RayService *v : s->svc)
        if (v) { // (the service launches once per start, not per ray)
            l += v->starts.load(std::memory_order_relaxed);
            r += v->sum(&RayService::Caller::rays);
        }
    if (out_launches) *out_launches = l;
    if (out_rays) *out_rays = r;
    return TRX_OK;
}

// What the ray services of a scene have done so far (summed over the semantics words in use): calls answered, kernel starts,
// nanoseconds callers spent between posting a ray and reading its answer, and of that the GPU-side share - 100 MHz ticks
// from a ray's admission to its answer - with the walk's trips.  Any pointer may be NULL.
int trx_debug_service_stats(trx_scene *s, uint64_t *out_rays, uint64_t *out_starts, uint64_t *out_call_ns, uint64_t *out_walk_ticks,
                            uint64_t *out_walk_trips) {
    if (!s) return fail(TRX_ERR_INVALID, "null argument");
    uint64_t r = 0, st = 0, ns = 0, tk = 0, tr = 0;
    for (RayService *v : s->svc)
        if (v) {
            r += v->sum(&RayService::Caller::rays);
            st += v->starts.load(std::memory_order_relaxed);
            ns += v->sum(&RayService::Caller::call_ns);
            tk += v->sum(&RayService::Caller::walk_ticks);
            tr += v->sum(&RayService::Caller::walk_trips);
        }
    if (out_rays) *out_rays = r;
    if (out_starts) *out_starts = st;
    if (out_call_ns) *out_call_ns = ns;
    if (out_walk_ticks) *out_walk_ticks = tk;
    if (out_walk_trips) *out_walk_trips = tr;
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

} // extern "C"

// Traversable::traverse through the ray service: claim a slot, post the ray, spin for the answer.
static int traverse1_service(trx_scene *s, const trx_ray *ray, uint32_t sem, trx_rayhit *out) {
    RayService *v = s->svc[sem & 7u];
    if (!v) {
        std::lock_guard<std::mutex> lock(s->svc_mu);
        v = s->svc[sem & 7u];
        if (!v) {
            HIP_TRY(hipSetDevice(s->device));
            v = new (std::nothrow) RayService(s, sem);
            if (!v || !v->ok) {
                const std::string why = v ? v->init_err : "allocation failed";
                delete v;
                return fail(TRX_ERR_OOM, "ray service: %s", why.c_str());
            }
            s->svc[sem & 7u] = v;
        }
    }
    auto ensure_running = [&]() -> int {
        if (v->running.load(std::memory_order_acquire)) return TRX_OK;
        std::lock_guard<std::mutex> lock(v->mu);
        HIP_TRY(hipSetDevice(s->device));
        return v->start_locked();
    };
    int rc = ensure_running();
    if (rc) return rc;
    // a slot of one's own: the thread's last slot if it is free, else the next free one.  Callers are spread over the
    // workgroups first (position j of the order = workgroup j % kGroups, ray j / kGroups of it): walkers on different CUs
    // step their rays side by side, the rays of one walker step together.
    static std::atomic<uint32_t> next_thread{0};
    thread_local uint32_t my_pos = next_thread.fetch_add(1, std::memory_order_relaxed) % RayService::kSlots;
    auto slot_at = [](uint32_t j) { return (j % RayService::kGroups) * trx::kSvcRays + (j / RayService::kGroups) % trx::kSvcRays; };
    uint32_t k = slot_at(my_pos);
    for (uint32_t tries = 0;; tries++) {
        uint32_t expect = 0u;
        if (v->caller[k].busy.compare_exchange_strong(expect, 1u, std::memory_order_acquire)) break;
        my_pos = (my_pos + 1u) % RayService::kSlots;
        k = slot_at(my_pos);
        if (tries > RayService::kSlots) std::this_thread::yield(); // more callers than slots: wait for one
    }
    RayService::Caller &me = v->caller[k];
    uint32_t seq = me.seq + 1u;
    if (seq == 0u) seq = 1u;
    me.seq = seq;
    uint32_t *slot = v->ring + (size_t)k * kSvcSlotWords;
    const float tmax = ray->tmax;
    // three 16-byte stores, each whole on its own (the kernel takes the request once all three carry `seq`)
    auto bits = [](float f) { int32_t i; std::memcpy(&i, &f, 4); return i; };
    _mm_store_si128(reinterpret_cast<__m128i *>(slot), _mm_set_epi32((int)seq, bits(ray->origin[2]), bits(ray->origin[1]), bits(ray->origin[0])));
    _mm_store_si128(reinterpret_cast<__m128i *>(slot + 4), _mm_set_epi32((int)seq, bits(ray->direction[2]), bits(ray->direction[1]), bits(ray->direction[0])));
    _mm_store_si128(reinterpret_cast<__m128i *>(slot + 8), _mm_set_epi32((int)seq, 0, bits(tmax), bits(ray->tmin)));
    _mm_sfence();
    const volatile uint32_t *ans = slot + 16;
    const int64_t t0 = now_ns();
    // (more calling threads than cores: give the core away between looks.  Threads are counted once, when they first call.)
    static std::atomic<int> calling_threads{0};
    struct Registered {
        Registered() { calling_threads.fetch_add(1, std::memory_order_relaxed); }
        ~Registered() { calling_threads.fetch_sub(1, std::memory_order_relaxed); }
    };
    thread_local Registered registered;
    (void)registered;
    static const int cores = (int)std::max(1u, std::thread::hardware_concurrency());
    const bool crowded = calling_threads.load(std::memory_order_relaxed) > cores;
    // (two-level scenes: a second granule {instance, -, -, seq} beside the answer; the two land in either order)
    const volatile uint32_t *ans2 = slot + 20;
    const bool two = s->tlas;
    for (uint32_t spins = 0; ans[3] != seq || (two && ans2[3] != seq); spins++) {
        if (crowded) std::this_thread::yield();
        else cpu_relax();
        if ((spins & 0x3ffu) == 0x3ffu) {
            // (the service stops itself after 50 ms without callers; one that arrives at that very moment starts it again -
            // its request is still in the ring)
            rc = ensure_running();
            if (rc || now_ns() - t0 > RayService::kGiveUpNs) {
                me.busy.store(0u, std::memory_order_release);
                return rc ? rc : fail(TRX_ERR_NO_DEVICE, "the ray service did not answer within %lld s", (long long)(RayService::kGiveUpNs / 1000000000));
            }
        }
    }
    const __m128i a = _mm_load_si128(reinterpret_cast<const __m128i *>(slot + 16));
    alignas(16) uint32_t w[4], w2[4] = {0xFFFFFFFFu, 0u, 0u, seq};
    _mm_store_si128(reinterpret_cast<__m128i *>(w), a);
    if (two) _mm_store_si128(reinterpret_cast<__m128i *>(w2), _mm_load_si128(reinterpret_cast<const __m128i *>(slot + 20)));
    const int64_t t1 = now_ns();
    me.rays.fetch_add(1, std::memory_order_relaxed);
    me.walk_ticks.fetch_add((w[2] >> 1) & 0x7fffu, std::memory_order_relaxed);
    me.walk_trips.fetch_add(w[2] >> 16, std::memory_order_relaxed);
    me.call_ns.fetch_add((uint64_t)(t1 - t0), std::memory_order_relaxed);
    me.last_use_ns.store(t1, std::memory_order_relaxed);
    me.busy.store(0u, std::memory_order_release);
    if (w[3] != seq || w2[3] != seq) return fail(TRX_ERR_NO_DEVICE, "the ray service's answer was torn");
    if (w[2] & 1u) return fail(TRX_ERR_STACK_OVERFLOW, "a ray overflowed the %d-entry traversal stack (or the step cap)", kLdsStack + kSpillStack);
    trx_hit h;
    std::memcpy(&h.t, &w[0], 4);
    h.prim = w[1];
    to_rayhit(s, h, w2[0], out);
    return TRX_OK;
}
