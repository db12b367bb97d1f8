/* trx_dev.h - development surface of libtrx.so: diagnostics and tuning switches used by tools/, bench.py's legs and
 * the tests.  NOT part of the drop-in boundary: nothing here replaces an interface of tray_racing, INTEGRATION.md binds
 * none of it, and a host that only includes trx.h never sees it.  Results of every trace entry point are identical
 * whatever is set here (tests/test_gpu_parity.py::test_scheduling_variants_and_streams_do_not_change_results). */
#ifndef TRX_DEV_H
#define TRX_DEV_H

#include "trx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics: start / end wall-clock stamps (100 MHz ticks) of every persistent wave of one
 * primary frame: out_times[2*i], out_times[2*i+1].  Shows residency and the frame's tail. */
int trx_debug_wave_timeline(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                            uint32_t semantics, uint64_t *out_times, uint32_t max_waves,
                            uint32_t *out_waves);

/* Diagnostics: 8 words per wave: start, end (as above), then — only in libraries built with -DTRX_STAMPS,
 * zero otherwise — shader cycles spent in {refill, node fetch, node test, triangle phase, pop / bookkeeping}
 * and the number of loop trips. */
int trx_debug_wave_phases(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                          uint32_t semantics, uint64_t *out_records, uint32_t max_waves,
                          uint32_t *out_waves);
/* The same 8-word records for the AO pass of the frame (frame 0, eps 0.01; its primary pass runs first, unrecorded). */
int trx_debug_wave_timeline_ao(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                               uint32_t semantics, uint64_t *out_records, uint32_t max_waves,
                               uint32_t *out_waves);

/* Diagnostics (counting kernel): the compulsory footprint of one primary frame — how many distinct nodes were
 * fetched and distinct triangles tested (80 * nodes + 48 * tris + 8 * rays bytes). */
int trx_debug_footprint(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                        uint32_t semantics, uint64_t *out_nodes_touched, uint64_t *out_tris_touched);

/* Diagnostics (counting kernel): over the triangle phases of one primary frame, out_hist[0..15] = histogram of the
 * largest per-lane triangle count of the wave (15 = 15 or more), out_hist[16..31] = histogram of the wave's
 * (ray, triangle) pair total in units of 8, rounded up. */
int trx_debug_tri_histogram(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                            uint32_t semantics, uint32_t out_hist[32]);

/* Diagnostics: per 8x8 tile (row-major tile id) the wall-clock cost of the tile in a normal frame
 * (100 MHz ticks) and its wave-level iteration counts from a counting frame:
 * (node steps << 16) | triangle rounds.  n_tiles = ceil(w/8) * ceil(h/8). */
int trx_debug_tile_profile(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                           uint32_t semantics, uint32_t *out_cost, uint32_t *out_iters, uint32_t n_tiles);

/* Launches issued and rays served by trx_traverse1 on this scene so far: the ray services' kernel starts (a handful per
 * stretch of calls) plus, under TRX_TRAVERSE1_COMBINER=1, the combiner's launches (rays / launches = callers that shared
 * a launch on average).  Either pointer may be NULL. */
int trx_debug_traverse1_stats(trx_scene *scene, uint64_t *out_launches, uint64_t *out_rays);

/* The ray services of the scene (a resident kernel answers trx_traverse1, include/trx.h) so far: calls
 * answered, kernel starts, nanoseconds callers spent between posting a ray and reading its answer, the GPU-side share of
 * that in 100 MHz ticks (admission to answer) and the trips of the walks.  Any pointer may be NULL. */
int trx_debug_service_stats(trx_scene *scene, uint64_t *out_rays, uint64_t *out_starts, uint64_t *out_call_ns,
                            uint64_t *out_walk_ticks, uint64_t *out_walk_trips);

/* Measuring aid: the reference's CPU pixel loop over the literal Traversable::traverse (src/rt_cpu/rt_cpu.rs:35-57) -
 * `threads` host threads, thread k calls trx_traverse1 for rays k, k + threads, ... - with the loop's wall-clock seconds and
 * the launches its calls shared. */
int trx_debug_traverse1_threads(trx_scene *scene, const trx_ray *rays, uint64_t n_rays, uint32_t threads, uint32_t semantics,
                                trx_rayhit *out, double *out_seconds, uint64_t *out_launches);

/* Measuring aid: the no-locality rate of the node-fetch loop on THIS scene's buffers (bench.py's `fetch_vs_random`).  The tracer's
 * persistent grid at the tracer's occupancy, every lane fetching one uniformly random 80-byte node per step and, on average,
 * tris_per_node_x256 / 256 random 48-byte triangle records beside it, the next index a hash of what arrived (one step in
 * flight per wave, like a traversing wave) - and nothing else.  Returns nodes and triangle records fetched per second. */
int trx_debug_fetch_rate(trx_scene *scene, uint32_t steps, uint32_t tris_per_node_x256, double *out_nodes_per_s,
                         double *out_tris_per_s);

/* Measuring aid: the streaming ceiling of this GPU's memory - a float4 copy kernel over `bytes` (read + written bytes per
 * second, best of `reps` passes after two warm-up passes): bench.py's `roofline_hbm.peak_measured`. */
int trx_debug_copy_rate(int device, uint64_t bytes, uint32_t reps, double *out_bytes_per_s);

/* Kernel variant selection (tuning aid; 0 = default).  Returns the previous
 * value.  Variants compute identical results. */
uint32_t trx_set_kernel_variant(uint32_t variant);

#ifdef __cplusplus
}
#endif

#endif /* TRX_DEV_H */
