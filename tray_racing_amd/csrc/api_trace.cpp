// api_trace.cpp - trace, count, bench and diagnostic entry points of include/trx.h / trx_dev.h over enqueue()
// (api_launch.cpp): device-resident forms, host-buffer twins with hipEvent timing (replaces src/timestamp.rs).
#include "api_internal.h"

extern "C" {

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

} // extern "C"

int trxapi::trace_rays_impl(trx_scene *s, const trx_ray *d_rays, uint64_t n, uint32_t sem, trx_hit *d_hits, hipStream_t stream, bool count,
                            SlotCounters **ctr, bool any_hit, uint32_t *d_inst, uint32_t *over_host, bool one_queue) {
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

extern "C" {

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
    bool count = true;
#ifdef TRX_DEV_TUNE
    if (getenv("TRX_HIST_NORMAL")) count = false; // (development builds: the histograms a NORMAL frame files under a tune word)
#endif
    rc = enqueue(s, p, kModePrimary, sem, count, nullptr, &ctr);
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

// The reference's frame loop, device-resident (src/rt_gpu/rt_gpu_software.rs:271-361: every frame a primary pass and the AO
// pass over its hits, --animate advancing the noise seed).  Serial: both passes of every frame on one stream, back to back -
// what trx_trace_primary_ao does per call, without the host in between.  Overlapped: the primary passes on stream A, the AO
// passes on stream B; AO(i) waits for primary(i), primary(i + 4) waits for AO(i) (four primary-hit buffers), so frame i's AO
// pass - whose last few hundred rays run alone on an almost idle GPU - overlaps a later frame's primary pass and the
// primary passes' own tails overlap AO passes.  Each stream keeps its
// launch slot and so its tile order (api_launch.cpp); the records are those of the serial loop.
int trx_frame_loop(trx_scene *s, const trx_view *view, uint32_t w, uint32_t h, uint32_t sem, uint32_t frame0, int animate,
                   float ao_eps, uint32_t n_frames, int overlap, trx_hit *out_primary, trx_hit *out_ao, float *out_ms) {
    if (!s || !view) return fail(TRX_ERR_INVALID, "null argument");
    if (n_frames == 0 || n_frames > 1000000u) return fail(TRX_ERR_INVALID, "n_frames %u outside 1..1000000", n_frames);
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu); // serialises users of the scene's scratch buffers
    HIP_TRY(hipSetDevice(s->device));
    const uint64_t n = (uint64_t)w * h;
    if (n == 0 || n > 0x7fffffffull) return fail(TRX_ERR_INVALID, "image %ux%u", w, h);
    FrameLoop &fl = s->loop;
    if (!fl.stream[0]) {
        for (int k = 0; k < 2; k++) HIP_TRY(hipStreamCreateWithFlags(&fl.stream[k], hipStreamNonBlocking));
        for (int k = 0; k < FrameLoop::kBuffers; k++) {
            HIP_TRY(hipEventCreateWithFlags(&fl.prim_done[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&fl.ao_done[k], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreate(&fl.t0));
        HIP_TRY(hipEventCreate(&fl.t1));
    }
    if (fl.records < n) {
        for (int k = 0; k < 2; k++) HIP_TRY(hipStreamSynchronize(fl.stream[k]));
        for (int k = 0; k < FrameLoop::kBuffers; k++) {
            if (fl.prim[k]) (void)hipFree(fl.prim[k]);
            if (fl.prim_inst[k]) (void)hipFree(fl.prim_inst[k]);
            fl.prim[k] = nullptr;
            fl.prim_inst[k] = nullptr;
        }
        if (fl.ao) (void)hipFree(fl.ao);
        if (fl.ao_inst) (void)hipFree(fl.ao_inst);
        fl.ao = nullptr;
        fl.ao_inst = nullptr;
        fl.records = 0;
        for (int k = 0; k < FrameLoop::kBuffers; k++) {
            HIP_TRY(hipMalloc(&fl.prim[k], n * sizeof(trx_hit)));
            if (s->tlas) HIP_TRY(hipMalloc(&fl.prim_inst[k], n * sizeof(uint32_t)));
        }
        HIP_TRY(hipMalloc(&fl.ao, n * sizeof(trx_hit)));
        if (s->tlas) HIP_TRY(hipMalloc(&fl.ao_inst, n * sizeof(uint32_t)));
        fl.records = n;
    }
    hipStream_t sa = fl.stream[0], sb = overlap ? fl.stream[1] : fl.stream[0];
    const trx_shard whole{0, 1, 0, 0};
    HIP_TRY(hipEventRecord(fl.t0, sa));
    int rc = TRX_OK;
    for (uint32_t i = 0; i < n_frames && rc == TRX_OK; i++) {
        const int b = (int)(i % (uint32_t)FrameLoop::kBuffers);
        if (i >= (uint32_t)FrameLoop::kBuffers && overlap) HIP_TRY(hipStreamWaitEvent(sa, fl.ao_done[b], 0)); // AO(i - 4) has read this buffer
        rc = trx_trace_primary_inst_dev(s, view, w, h, whole, sem, fl.prim[b], fl.prim_inst[b], sa);
        if (rc) break;
        if (overlap) {
            HIP_TRY(hipEventRecord(fl.prim_done[b], sa));
            HIP_TRY(hipStreamWaitEvent(sb, fl.prim_done[b], 0));
        }
        rc = trx_trace_ao_inst_dev(s, view, w, h, whole, sem, frame0 + (animate ? i : 0u), ao_eps, fl.prim[b], fl.prim_inst[b], fl.ao,
                                   fl.ao_inst, sb);
        if (rc) break;
        if (overlap) HIP_TRY(hipEventRecord(fl.ao_done[b], sb));
    }
    if (rc == TRX_OK && overlap) HIP_TRY(hipStreamWaitEvent(sa, fl.ao_done[(n_frames - 1u) % (uint32_t)FrameLoop::kBuffers], 0));
    if (rc == TRX_OK) HIP_TRY(hipEventRecord(fl.t1, sa));
    // (whatever happened, nothing of this call is left in flight when it returns)
    const hipError_t e0 = hipStreamSynchronize(fl.stream[0]), e1 = hipStreamSynchronize(fl.stream[1]);
    if (rc) return rc;
    HIP_TRY(e0);
    HIP_TRY(e1);
    if (out_ms) HIP_TRY(hipEventElapsedTime(out_ms, fl.t0, fl.t1));
    const int last = (int)((n_frames - 1u) % (uint32_t)FrameLoop::kBuffers);
    if (out_primary) HIP_TRY(hipMemcpy(out_primary, fl.prim[last], n * sizeof(trx_hit), hipMemcpyDeviceToHost));
    if (out_ao) HIP_TRY(hipMemcpy(out_ao, fl.ao, n * sizeof(trx_hit), hipMemcpyDeviceToHost));
    for (Slot &sl : s->slots) {
        if (!sl.ctr) continue;
        rc = read_overflow(s, sl.ctr);
        if (rc) return rc;
    }
    return TRX_OK;
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

} // extern "C"
