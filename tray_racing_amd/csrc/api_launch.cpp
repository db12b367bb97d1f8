// api_launch.cpp - the launch path behind every trace entry point: launch slots (a scene is re-entrant), kernel
// parameters, the tile-order state of a slot and the host side of the schedule tuner (replaces the dispatch of
// src/rt_gpu/rt_gpu_software.rs:271-361).
#include "api_internal.h"

namespace trxapi {

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
    if (pick < 0) { // the oldest slot no resident kernel sits on (at most kSlots - 1 are pinned: one per ray service)
        for (int i = 0; i < kSlots; i++)
            if (!s->slots[i].pinned && (pick < 0 || s->slots[i].last_use < s->slots[pick].last_use)) pick = i;
        if (pick < 0) return fail(TRX_ERR_INVALID, "every launch slot of the scene is held by a resident kernel");
    }
    Slot &slot = s->slots[pick];
    const bool same_stream = slot.used && slot.last_stream == stream;
    const uint32_t variant = g_variant.load(std::memory_order_relaxed);
    // tuning overrides (trx_set_kernel_variant): bits 8..12 waves per CU, bits 16..19 waves per workgroup
    uint32_t wpb = (variant >> 16) & 0x7u; // (bit 19: the tile-order feedback does not tune itself off, see below)
    // incoherent single-level passes (AO, explicit rays) run two waves to a workgroup, so that the second can hand its last rays to the first
    // when both are draining (kernels.hip, "drain"); an explicit 1 or 4 here switches that off
    // (two-level scenes: explicit rays only, see kMerge in kernels.hip)
    const bool merge_default = (wpb != 1 && wpb != 2 && wpb != 4) && mode != kModePrimary && mode != kModeFused && mode != kModeService &&
                               (!s->tlas || mode == kModeRays) && !count;
    if (wpb != 1 && wpb != 2 && wpb != 4) wpb = merge_default ? 2u : kDefaultWavesPerBlock;
    if (mode == kModeService) wpb = 2u; // a walker and a porter (kernels.h, kSvcRays); n_items = 64 per WAVE of the grid
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
    // (Round 6 tried an order for one-seed AO passes with mid-tile refills as well - first pass of a view natural, second
    // measuring with whole tiles, then replayed frozen: hairball-class pass 0.809 -> 0.799 ms, bistro-class 0.892 -> 0.895,
    // profiles/r06_ab_ao_tile_order.log - the longest rays take the whole pass wherever they start, as round 3 found; removed.)
    const bool lpt = mode != kModeRays && mode != kModeFused && mode != kModeService && p.refill_idle == 64u && !((variant >> 20) & 1u) &&
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
    if (mode == kModeService) pipe = false;
    HIP_TRY(launch_trace(p, mode, s->tlas, sem, count, pipe, grid, stream));
    HIP_TRY(hipEventRecord(slot.done, stream));
    slot.used = true;
    if (mode == kModeService) slot.pinned = true; // (until the service is stopped: RayService::stop_locked)
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

} // namespace trxapi
