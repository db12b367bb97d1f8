/*
 * trx.h — C-ABI of the MI355X (gfx950) CWBVH traversal backend for tray_racing.
 *
 * This is the drop-in boundary for the ONE hot path of DGriffin91/tray_racing:
 * closest-hit traversal of rays through a CWBVH (80-byte nodes) with
 * Moeller-Trumbore triangle tests, for primary and AO rays.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the tray_racing checkout; `query.hlsl` / `query_tlas.hlsl` are short for
 * src/rt_gpu/rt_gpu_software_query.hlsl / rt_gpu_software_query_tlas.hlsl, other
 * bare shader and .rs names live under src/rt_gpu/ or src/rt_cpu/).  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.  All functions return 0 on success or a
 * negative trx_status; the message is available from trx_last_error()
 * (thread-local).  Nothing here aborts: the Rust shim `expect()`s on the
 * status to keep the reference's panic convention (src/main.rs:178-180).
 *
 * There is NO CPU fallback behind this ABI: every trace entry point fails
 * with TRX_ERR_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef TRX_H
#define TRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRX_ABI_VERSION 1

typedef enum trx_status {
    TRX_OK = 0,
    TRX_ERR_INVALID = -1,        /* bad argument (null pointer, size mismatch, bad enum) */
    TRX_ERR_NO_DEVICE = -2,      /* no usable HIP device / HIP call failed               */
    TRX_ERR_OOM = -3,            /* host or device allocation failed                     */
    TRX_ERR_STACK_OVERFLOW = -4, /* a ray exceeded the traversal stack (result invalid)  */
    TRX_ERR_FORMAT = -5,         /* node/triangle buffer fails structural validation     */
    TRX_ERR_IO = -6              /* scene / model file could not be read                 */
} trx_status;

/* ---- data formats -------------------------------------------------------- */

/* One CWBVH node = 80 bytes = uint4[5], exactly the layout the reference's
 * GPU path consumes (src/rt_gpu/rt_gpu_software_query.hlsl:40-43, decode at
 * :219-224,249,257-264,381-384; size asserts src/rt_gpu/mod.rs:70,105;
 * writer embree/src/bvh_embree_to_cwbvh.rs:172-185). */
#define TRX_NODE_BYTES 80u

/* Triangle input formats accepted by trx_scene_create. */
typedef enum trx_tri_format {
    /* obvhs RtCompressedTriangle, 24 B: f32 v0[3]; u32 e[3] with
     * e[k] = f16(e2[k]) | f16(e1[k]) << 16, e1 = v1 - v0, e2 = v2 - v0
     * (src/rt_gpu/rt_gpu_software_query.hlsl:45-49,75-85; src/rt_gpu/mod.rs:39-43,86). */
    TRX_TRI_F16_24 = 0,
    /* Raw obvhs `Triangle`, 36 B: f32 v0[3], v1[3], v2[3] (src/main.rs:515-519,540-544). */
    TRX_TRI_VERTS_36 = 1,
    /* f32 {v0[3], e1[3], e2[3]}, 36 B with e1 = v0 - v1, e2 = v2 - v0: the
     * f32 content of obvhs RtTriangle without padding / ng (src/rt_cpu/mod.rs:38-43). */
    TRX_TRI_EDGES_36 = 2
} trx_tri_format;

/* Bytes per triangle for a trx_tri_format (0 for an unknown value). */
uint32_t trx_tri_format_bytes(uint32_t tri_format);

/* Traversal semantics (bit set).  0 is the literal in-tree HLSL text.  The
 * reference's --cpu path lives in the un-vendored obvhs crate; the deviations
 * recalled for it are exposed as bits, and TRX_SEM_CPU is
 * the preset that combines them.  Every combination has an oracle twin. */
typedef enum trx_semantics {
    /* per-node IEEE divides, separate multiply and add for the slab planes,
     * tt <= t commits (last equal-t triangle wins):
     * src/rt_gpu/rt_gpu_software_query.hlsl:237-242,285-286,120 */
    TRX_SEM_HLSL = 0,
    /* node test multiplies by inv_dir = 1/dir computed once per ray with an IEEE
     * divide.  2^e * inv_dir is bit-identical to 2^e / dir (power-of-two scale);
     * (p - origin) * inv_dir may differ from (p - origin) / dir in the last ulp,
     * which can only change which boxes are culled at razor edges. */
    TRX_SEM_NODE_RCP = 1u << 0,
    /* commit a triangle only if tt < t (first equal-t triangle wins) */
    TRX_SEM_TIE_FIRST = 1u << 1,
    /* evaluate the slab planes q * adj_inv + adj_origin with one fused
     * multiply-add (the HLSL leaves contraction to the compiler) */
    TRX_SEM_NODE_FMA = 1u << 2,
    /* recalled obvhs CPU path: precomputed inv_direction, strict t < tmax */
    TRX_SEM_CPU = (1u << 0) | (1u << 1)
} trx_semantics;

/* ViewUniform mirror (src/main.rs:589-597, HLSL cbuffer
 * src/rt_gpu/rt_gpu_software.hlsl:30-38).  Matrices are column-major like glam
 * (m[c*4 + r]).  Filled by trx_view_from_camera or by the caller. */
typedef struct trx_view {
    float view_inv[16];
    float proj_inv[16];
    float eye[3];
    float exposure;
    uint32_t tlas_start; /* unused by the traced path; scene carries it */
    uint32_t _pad[3];
} trx_view;

/* Explicit ray, 32 B: mirrors obvhs Ray::new(origin, direction, tmin, tmax)
 * as used at src/rt_cpu/rt_cpu.rs:50-55,76. */
typedef struct trx_ray {
    float origin[3];
    float tmin;
    float direction[3];
    float tmax;
} trx_ray;

/* Compact hit record written by every kernel: 8 B per ray.
 * Miss: t = +inf, prim = 0xFFFFFFFF (RayHit::none(), callers test
 * hit.t < f32::MAX, src/rt_cpu/rt_cpu.rs:61,82).  `prim` indexes the
 * triangle buffer handed to trx_scene_create (i.e. primitive_indices order,
 * src/rt_gpu/mod.rs:38-48). */
typedef struct trx_hit {
    float t;
    uint32_t prim;
} trx_hit;

/* Mirror of obvhs RayHit {primitive_id, geometry_id, instance_id, t}
 * (fields: embree/src/embree_managed.rs:52-57) for the Traversable shim. */
typedef struct trx_rayhit {
    uint32_t primitive_id;
    uint32_t geometry_id;
    uint32_t instance_id;
    float t;
} trx_rayhit;

/* Which pixels of the w*h image this call (this GPU) traces: 8x8-pixel tiles
 * are numbered row-major; a call traces tiles with (tile % count) == index.
 * {0,1,0,0} = whole image.
 * layout 0 (TRX_LAYOUT_IMAGE): hit buffers are indexed by full-image pixel id
 * (y*w + x); pixels of other shards are left as they were.
 * layout 1 (TRX_LAYOUT_SHARD): hit buffers are compact, indexed by
 * local_tile*64 + (y&7)*8 + (x&7) with local_tile = tile / count; they hold
 * trx_shard_tiles() * 64 records (records of pixels outside the image are not
 * written).  This is the buffer a rank hands to the end-of-frame gather. */
#define TRX_LAYOUT_IMAGE 0u
#define TRX_LAYOUT_SHARD 1u
typedef struct trx_shard {
    uint32_t index;
    uint32_t count;
    uint32_t layout;
    uint32_t _pad;
} trx_shard;
/* Number of 8x8 tiles of a width x height image that belong to `shard`. */
uint32_t trx_shard_tiles(uint32_t width, uint32_t height, trx_shard shard);

/* Per-call counters (optional; pass NULL).  n_node / n_tri are the PROFILE_RT
 * counters of the reference (aabb_hit_count/8 and tri_hit_count,
 * src/rt_gpu/rt_gpu_software_query.hlsl:377-379,407-409), summed over rays. */
typedef struct trx_stats {
    uint64_t n_rays;
    uint64_t n_node;
    uint64_t n_tri;
    uint64_t n_hits;
    uint32_t max_stack;
    uint32_t overflow; /* number of rays that overflowed the stack */
    float kernel_ms;   /* hipEvent time of the traversal kernel(s) of this call */
    float _pad;
    /* wave-level executions of the node step / triangle test: n_node / (64 * n_wave_node)
     * is the SIMD efficiency of the node phase, likewise for triangles */
    uint64_t n_wave_node;
    uint64_t n_wave_tri;
} trx_stats;

typedef struct trx_scene trx_scene; /* opaque: device-resident nodes/tris/instances */

/* ---- errors / device ------------------------------------------------------ */

const char *trx_last_error(void);
uint32_t trx_abi_version(void);
/* Number of HIP devices visible (0 when none; never fails). */
int trx_device_count(void);
/* gcnArchName of a device into buf (e.g. "gfx950:sramecc+:xnack-"). */
int trx_device_name(int device, char *buf, size_t buf_len);

/* ---- scene upload ---------------------------------------------------------
 * Replaces the buffer upload half of rt_gpu_software::start
 * (src/rt_gpu/rt_gpu_software.rs:24-32 signature; :83-160 buffer creation),
 * fed by cwbvh_gpu_runner (src/rt_gpu/mod.rs:92-100 TLAS, :108-110 BLAS-only).
 *
 *  bvh_bytes        n_nodes * 80 bytes; all BLAS concatenated, TLAS last
 *  tri_bytes        n_tris * trx_tri_format_bytes(tri_format), in
 *                   primitive_indices order, primitive_base_idx pre-offset
 *  instance_offsets u32 BLAS node offset per TLAS primitive, or NULL / 0 when
 *                   there is no TLAS (the reference passes 16 zero bytes)
 *  tlas_start       node index of the TLAS root (0 and n_instances == 0 = BLAS only)
 *  device           HIP device ordinal
 * The inputs are copied; the caller keeps ownership. */
int trx_scene_create(const void *bvh_bytes, uint64_t n_nodes,
                     const void *tri_bytes, uint64_t n_tris, uint32_t tri_format,
                     const uint32_t *instance_offsets, uint32_t n_instances,
                     uint32_t tlas_start, int device, trx_scene **out);
void trx_scene_destroy(trx_scene *scene);
/* Bytes resident in HBM for this scene: nodes + 48-byte triangles + instances + what the launch slots used so far
 * hold (stack spill areas sized for the launched grids, tile-order lists). */
uint64_t trx_scene_device_bytes(const trx_scene *scene);
int trx_scene_device(const trx_scene *scene);

/* Optional: first triangle of each BLAS in the global triangle buffer
 * (n_blas + 1 entries, trx_flat.blas_tri_start) so trx_traverse1 can report
 * (geometry_id, local primitive_id) like the reference's CPU TLAS path
 * (src/cwbvh.rs:151-160). */
int trx_scene_set_geometry_ranges(trx_scene *scene, const uint32_t *blas_tri_start, uint32_t n_blas);

/* ---- entry nodes (TLAS scenes) -------------------------------------------------
 * In the reference a TLAS primitive is a whole BLAS: the walk enters at node 0 of the BLAS at instance_offsets[k]
 * (rt_gpu_software_query_tlas.hlsl:439-443).  entry_nodes[k] lets primitive k enter at another node of that BLAS
 * instead, so a TLAS can reference SUBTREES (re-braiding: a BLAS spanning the scene is opened into the pieces under its
 * root, trx_set_build_rebraid / trx_flat.instance_entry_nodes).  Several primitives may name the same BLAS; hit records
 * are unchanged (primitive ids are global, instance ids are TLAS primitives).  NULL / 0 restores node 0 everywhere; an
 * entry outside its BLAS is refused (TRX_ERR_FORMAT).  Buffers that come from the reference's host never need this. */
int trx_scene_set_instance_entry_nodes(trx_scene *scene, const uint32_t *entry_nodes, uint32_t n_instances);

/* ---- instance transforms (TLAS scenes) ----------------------------------------
 * The reference's TLAS path carries no transforms yet: "TODO transform ray according to the mesh transform"
 * (src/rt_gpu/rt_gpu_software_query_tlas.hlsl:409,433), "TODO Reset Ray to untransformed version" (:484), and
 * Traversable::get_instance_transform returns Mat4::default() (traversable/src/lib.rs:25-27, src/cwbvh.rs:163-165).
 * Here an instance is a TLAS primitive (entry k of instance_offsets): it may place its BLAS anywhere, and several
 * instances may share one BLAS.  object_to_world: n_instances column-major 4x4 affine matrices (glam Mat4 layout,
 * last row 0 0 0 1) in instance_offsets order; NULL / 0 restores identity.  On BLAS entry the ray is taken into
 * object space by the inverse (computed in double, rounded once to f32; origin as a point, direction as a vector,
 * NOT renormalised, so hit.t stays in world units); on exit the world ray is restored.  The TLAS node boxes handed
 * to trx_scene_create must bound the TRANSFORMED instances (trx_flat_build_instanced does that).  With no
 * transforms set the kernels behave exactly as before. */
int trx_scene_set_instance_transforms(trx_scene *scene, const float *object_to_world, uint32_t n_instances);
/* Traversable::get_instance_transform: the matrix given above, or identity. */
int trx_scene_get_instance_transform(const trx_scene *scene, uint32_t instance_id, float out_object_to_world[16]);
/* The world-to-object rows the kernels use: 3 rows of {m0 m1 m2 t}; x' = ((m0*x + m1*y) + m2*z) + t. */
int trx_scene_get_instance_world_to_object(const trx_scene *scene, uint32_t instance_id, float out_rows[12]);

/* ---- camera ---------------------------------------------------------------
 * ViewUniform::from_camera (src/main.rs:602-616): proj_inv =
 * inverse(perspective_infinite_reverse_rh(fov_deg->rad, w/h, 0.01)),
 * view_inv = inverse(look_at_rh(eye, look_at, +Y)). */
int trx_view_from_camera(const float eye[3], const float look_at[3], float fov_deg,
                         float width, float height, trx_view *out);

/* ---- tracing: device-resident outputs --------------------------------------
 * `stream` is a hipStream_t (NULL = the null stream).  d_* pointers are device
 * memory on the scene's device.  These calls only enqueue work.
 *
 * Scheduling state (never results) lives with the stream: a stream keeps a launch slot of
 * the scene, and an image pass remembers, per image geometry, in which order its 8x8 tiles
 * are best started (heaviest first, learnt from the previous pass of that geometry on that
 * stream).  A frame loop that keeps its passes on one stream - as the reference does on its
 * one queue - gets that for free from the second frame on, camera motion and cuts included;
 * the first pass of a geometry, and a caller that sprays frames over many streams, run in
 * natural order (bench.py: 0.52 ms against 0.42 ms).  Hits are identical either way.
 *
 * trx_trace_primary_dev replaces the timed dispatch of the reference
 * (src/rt_gpu/rt_gpu_software.rs:289-302) restricted to the primary ray of
 * src/rt_gpu/rt_gpu_software.hlsl:69-89: ray-gen in-kernel, closest hit,
 * one trx_hit per pixel at d_hits[y*w + x]. */
int trx_trace_primary_dev(trx_scene *scene, const trx_view *view, uint32_t width,
                          uint32_t height, trx_shard shard, uint32_t semantics,
                          trx_hit *d_hits, void *stream);

/* The same with the instance (TLAS primitive index, RayHit.instance_id; 0xFFFFFFFF for a miss or a scene without
 * TLAS) each hit was found in: d_inst has one u32 per record, indexed like d_hits; may be NULL. */
int trx_trace_primary_inst_dev(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                               trx_shard shard, uint32_t semantics, trx_hit *d_hits, uint32_t *d_inst, void *stream);

/* n_frames (1..TRX_MAX_BATCH_FRAMES) primary frames in ONE launch: frame f is traced with views[f] and its
 * records go to d_hits + f * frame_stride (in trx_hit records; frame_stride >= one frame's records for the
 * shard layout).  The reference renders frame after frame with one dispatch each
 * (src/rt_gpu/rt_gpu_software.rs:289-302); submitting several of them together gives the kernel's work queues
 * n_frames x the tiles to balance, which matters when a rank's share of one frame is small (tile shards on 8
 * GPUs: 4050 tiles for 4096 resident waves).  Results are identical to n_frames separate launches. */
#define TRX_MAX_BATCH_FRAMES 8
int trx_trace_primary_batch_dev(trx_scene *scene, const trx_view *views, uint32_t n_frames, uint32_t width,
                                uint32_t height, trx_shard shard, uint32_t semantics, trx_hit *d_hits,
                                uint64_t frame_stride, void *stream);

/* AO pass of src/rt_gpu/rt_gpu_software.hlsl:105-128 / src/rt_cpu/rt_cpu.rs:61-80:
 * for every pixel whose d_primary hit is valid, build the AO ray (normal from
 * the hit triangle flipped toward the viewer, origin = eye + d*t - d*ao_eps,
 * cosine-hemisphere direction from hash_noise(px, frame) /
 * hash_noise(px, frame + 1024)) and trace it (closest hit).  Pixels whose
 * primary ray missed get a miss record.  ao_eps: 0.0001 (GPU) / 0.01 (CPU).
 * The direction's sin / cos (sampling.hlsl:33-34 calls the platform's; on the CPU path Rust's f32::sin / cos are the C
 * library's sinf / cosf) are evaluated explicitly so that results reproduce bit for bit across machines: the binary64
 * algorithm the GNU C library (2.28 and later) publishes for sinf / cosf, rounded once to binary32 - measured identical to
 * glibc 2.35's sinf / cosf on x86_64 for every binary32 in [0, 2 pi] (tests/test_oracle.py; other glibc versions and
 * architectures pick other variants and may differ in the last bit), so on such a host the AO directions are the
 * reference CPU path's own.
 * Against a correctly rounded sin / cos (another C library) the AO hit's t moves in its last bits on a fraction of a
 * per cent of the rays. */
int trx_trace_ao_dev(trx_scene *scene, const trx_view *view, uint32_t width,
                     uint32_t height, trx_shard shard, uint32_t semantics, uint32_t frame,
                     float ao_eps, const trx_hit *d_primary, trx_hit *d_ao, void *stream);

/* AO pass with instance ids: d_primary_inst (from trx_trace_primary_inst_dev) is REQUIRED when the scene has
 * instance transforms — the hit triangle's normal lives in object space and is taken to world space by the
 * transpose of the instance's world-to-object matrix; d_ao_inst may be NULL. */
int trx_trace_ao_inst_dev(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height, trx_shard shard,
                          uint32_t semantics, uint32_t frame, float ao_eps, const trx_hit *d_primary,
                          const uint32_t *d_primary_inst, trx_hit *d_ao, uint32_t *d_ao_inst, void *stream);

/* The reference's whole pixel program as ONE launch (src/rt_gpu/rt_gpu_software.hlsl:47-144 is one dispatch: primary
 * ray, normal of the hit triangle, AO ray): d_primary / d_ao receive exactly what trx_trace_primary_inst_dev followed by
 * trx_trace_ao_inst_dev write (d_primary_inst / d_ao_inst may be NULL).  A lane whose primary ray ends in a hit becomes
 * that pixel's AO ray in place, so neither pass has a tail of its own and the primary hits never travel through memory. */
int trx_trace_frame_dev(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height, trx_shard shard,
                        uint32_t semantics, uint32_t frame, float ao_eps, trx_hit *d_primary, uint32_t *d_primary_inst,
                        trx_hit *d_ao, uint32_t *d_ao_inst, void *stream);

/* n_frames (1..TRX_MAX_BATCH_FRAMES) AO passes over ONE view and ONE primary hit buffer as one launch: pass f uses the
 * noise seed frame0 + f and writes its records at d_ao + f * frame_stride (d_ao_inst likewise, may be NULL;
 * d_primary_inst as for trx_trace_ao_inst_dev, NULL without instance transforms).  This is BASELINE.json's "4 spp":
 * the reference traces one AO ray per hit pixel per frame and varies the seed with the frame counter
 * (src/rt_cpu/rt_cpu.rs:95-97, src/rt_gpu/rt_gpu_software.rs:285-288), so 4 spp are frames 0..3 - submitted together
 * they share one launch's start-up and, above all, ONE drain (the end of an incoherent pass, where every wave finishes
 * its last rays at falling occupancy: two thirds of a hairball-class pass).  The seeds of a tile go to the same XCD.
 * Results are identical to n_frames separate trx_trace_ao_inst_dev launches. */
int trx_trace_ao_batch_dev(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height, trx_shard shard,
                           uint32_t semantics, uint32_t frame0, uint32_t n_frames, float ao_eps,
                           const trx_hit *d_primary, const uint32_t *d_primary_inst, trx_hit *d_ao, uint32_t *d_ao_inst,
                           uint64_t frame_stride, void *stream);

/* Batch form of Traversable::traverse (traversable/src/lib.rs:13-28):
 * n explicit rays -> n hits. */
int trx_trace_rays_dev(trx_scene *scene, const trx_ray *d_rays, uint64_t n_rays,
                       uint32_t semantics, trx_hit *d_hits, void *stream);

/* ... with the instance of every hit (d_inst: n u32, may be NULL). */
int trx_trace_rays_inst_dev(trx_scene *scene, const trx_ray *d_rays, uint64_t n_rays, uint32_t semantics,
                            trx_hit *d_hits, uint32_t *d_inst, void *stream);

/* Any-hit query — the reference's `intersects_bl_bvh` (query.hlsl:440-445; "Actual AO could use a faster
 * anyhit query", src/rt_cpu/rt_cpu.rs:78-79): one byte per ray, 1 if some triangle is hit inside
 * [tmin, tmax].  The traversal stops at the first accepted hit; until then it IS the closest-hit walk, so the
 * answer equals `trx_trace_rays*`'s hit / no hit exactly. */
int trx_trace_occluded_dev(trx_scene *scene, const trx_ray *d_rays, uint64_t n_rays,
                           uint32_t semantics, uint8_t *d_flags, void *stream);

/* Counting variant (PROFILE_RT): same traversal, also accumulates trx_stats.
 * Synchronous; d_hits may be NULL. */
int trx_count_primary(trx_scene *scene, const trx_view *view, uint32_t width,
                      uint32_t height, trx_shard shard, uint32_t semantics,
                      trx_hit *d_hits, trx_stats *stats);
int trx_count_ao(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                 trx_shard shard, uint32_t semantics, uint32_t frame, float ao_eps,
                 const trx_hit *d_primary, trx_hit *d_ao, trx_stats *stats);
int trx_count_rays(trx_scene *scene, const trx_ray *d_rays, uint64_t n_rays,
                   uint32_t semantics, trx_hit *d_hits, trx_stats *stats);

/* Status of the most recent trace on this scene (stack overflow is detected
 * in-kernel and latched in device memory).  Synchronises `stream`. */
int trx_scene_check(trx_scene *scene, void *stream);

/* Re-entrancy: the *_dev entry points and trx_traverse1 may be called concurrently on one scene
 * (launch slots); the synchronous entry points below, trx_count_* and trx_bench_primary share
 * per-scene scratch buffers and are serialised by a per-scene lock.
 * Process-wide settings: the builder knobs (trx_set_build_*) live behind one mutex and every build works on
 * the snapshot it took when it started, so setters and builds may run concurrently (a setter only affects
 * builds that start after it returns); trx_flat_build_params never touches them.  trx_set_kernel_variant is a
 * single atomic word read once per launch — a tuning aid: concurrent launches may see either value, results are
 * identical under every value. */

/* ---- tracing: host-buffer convenience (synchronous) --------------------------
 * What rt_gpu_software::start returns to its caller is a time in ms
 * (src/rt_gpu/rt_gpu_software.rs:376); out_ms is the hipEvent time of the
 * kernel(s) (src/timestamp.rs:62-74 equivalent).  out_hits: w*h (or n) host
 * records, may be NULL for timing only. */
int trx_trace_primary(trx_scene *scene, const trx_view *view, uint32_t width,
                      uint32_t height, uint32_t semantics, trx_hit *out_hits,
                      float *out_ms);
int trx_trace_primary_ao(trx_scene *scene, const trx_view *view, uint32_t width,
                         uint32_t height, uint32_t semantics, uint32_t frame, float ao_eps,
                         trx_hit *out_primary, trx_hit *out_ao, float *out_ms);
/* The frame loop of rt_gpu_software::start itself (src/rt_gpu/rt_gpu_software.rs:271-361: per RedrawRequested a
 * primary pass and the AO pass over its hits; --animate advances the noise seed, :285-288), n_frames frames without the
 * host in between.  overlap == 0: both passes of a frame back to back on one stream (what n_frames calls of
 * trx_trace_primary_ao enqueue).  overlap != 0: frame i's AO pass runs on a second stream under frame i + 1's primary
 * pass (two primary-hit buffers; same records).  out_primary / out_ao (host, may be NULL): the LAST frame's records;
 * out_ms: hipEvent time from the first launch to the end of the last pass (per frame: / n_frames). */
int trx_frame_loop(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height, uint32_t semantics,
                   uint32_t frame0, int animate, float ao_eps, uint32_t n_frames, int overlap, trx_hit *out_primary,
                   trx_hit *out_ao, float *out_ms);
int trx_trace_rays(trx_scene *scene, const trx_ray *rays, uint64_t n_rays,
                   uint32_t semantics, trx_hit *out_hits, float *out_ms);
/* The same with RayHit.instance_id per hit (u32 arrays, any of them may be NULL; all 0xFFFFFFFF without a TLAS).
 * These are the forms to use on scenes with instance transforms: the AO pass is fed the primary pass's ids. */
int trx_trace_primary_ao_inst(trx_scene *scene, const trx_view *view, uint32_t width, uint32_t height,
                              uint32_t semantics, uint32_t frame, float ao_eps, trx_hit *out_primary,
                              uint32_t *out_primary_inst, trx_hit *out_ao, uint32_t *out_ao_inst, float *out_ms);
int trx_trace_rays_inst(trx_scene *scene, const trx_ray *rays, uint64_t n_rays, uint32_t semantics,
                        trx_hit *out_hits, uint32_t *out_inst, float *out_ms);
/* Host-buffer form of trx_trace_occluded_dev. */
int trx_trace_occluded(trx_scene *scene, const trx_ray *rays, uint64_t n_rays, uint32_t semantics,
                       uint8_t *out_flags, float *out_ms);
/* Single-ray Traversable::traverse (traversable/src/lib.rs:13-28).  Thread-safe and meant to be called the way the
 * reference calls it - from every worker of a thread pool at once (src/rt_cpu/rt_cpu.rs:35-57).  No call launches
 * anything: the first call starts the scene's RAY SERVICE, a resident kernel that answers out of a ring of 64 slots in
 * pinned host memory (a caller claims a slot, writes its ray, spins on the slot's answer; single- and two-level scenes,
 * one service per semantics word in use) and that a watchdog thread stops 50 ms after the last call - until then a
 * device-wide synchronisation elsewhere in the process (hipDeviceSynchronize, hipFree) waits for it; trx_scene_destroy
 * stops it at once.  (TRX_TRAVERSE1_COMBINER=1 in the environment sends two-level scenes through round 5's path instead:
 * the callers inside this function at the same moment share one launch.)  A caller blocks for its ray's own walk plus
 * 4 us through host memory, so the rate is (concurrent callers) / (16-45 us): use trx_traverse_batch where the rays can
 * be had together.  primitive_id indexes the PERMUTED triangle
 * list of the hit BLAS (primitive_indices order), exactly like the reference, whose scene structs store their
 * triangles in that order (src/rt_cpu/mod.rs:38-43) and index them with RayHit.primitive_id
 * (src/cwbvh.rs:151-160,177-186); trx_flat.tri_source[blas_tri_start[g] + primitive_id] names the input
 * triangle (with pre-splitting several entries can name the same one). */
int trx_traverse1(trx_scene *scene, const trx_ray *ray, uint32_t semantics,
                  trx_rayhit *out);
/* The same answer for n rays in ONE launch: what rt_cpu::start's pixel loop (src/rt_cpu/rt_cpu.rs:35-92) should call
 * instead of n device round trips - generate the frame's rays, traverse them here, shade from the RayHits (which
 * carry geometry_id / instance_id exactly like trx_traverse1's).  Host buffers; out_ms (may be NULL) is the hipEvent
 * time of the traversal. */
int trx_traverse_batch(trx_scene *scene, const trx_ray *rays, uint64_t n_rays, uint32_t semantics,
                       trx_rayhit *out, float *out_ms);

/* Benchmark loop of the reference (warm-up dispatch + timestamp pair per
 * frame, min and mean over frames: src/rt_gpu/rt_gpu_software.rs:289-302,
 * 339-344,376).  Traces `frames` primary frames after `warmup` untimed ones
 * into an internal device buffer; returns min / mean kernel ms. */
int trx_bench_primary(trx_scene *scene, const trx_view *view, uint32_t width,
                      uint32_t height, uint32_t semantics, uint32_t warmup,
                      uint32_t frames, float *out_min_ms, float *out_mean_ms);

/* Development entry points (per-wave timelines, footprints, tile profiles, the kernel-variant tuning word) are NOT part
 * of the boundary a tray_racing host binds: they live in include/trx_dev.h. */

/* ---- multi-GPU: the hit-shard gather -----------------------------------------
 * tray_racing is single-GPU; these entry points are the north_star's "RCCL only for the final hit-buffer gather"
 * in a form a Rust / C host can drive: one process per GPU, rank r traces
 * trx_shard{r, N, TRX_LAYOUT_SHARD} of every frame of a batch straight into its block of a gather buffer laid out
 * [N][m][R] (R = trx_shard_tiles(w, h, {0, N}) * 64 records, the same on every rank), ONE in-place all-gather per
 * batch completes the m frames on every rank, and trx_assemble_frames de-interleaves them into row-major frames.
 * RCCL (librccl.so) is loaded on first use; a single-GPU host never touches it.  The environment variable
 * TRX_RCCL_LIBRARY, when set, names the library to load instead of the system's librccl.so; a library that cannot be
 * loaded makes every trx_comm_* call return TRX_ERR_NO_DEVICE with the loader's message. */
typedef struct trx_comm trx_comm;
/* 128 opaque bytes (ncclUniqueId): produced on rank 0, handed to the other ranks by the host's own means. */
int trx_comm_unique_id(void *out_id128);
/* Collective over all ranks (ncclCommInitRank).  device: the HIP device of this rank. */
int trx_comm_create(const void *id128, int rank, int world, int device, trx_comm **out);
void trx_comm_destroy(trx_comm *comm);
int trx_comm_world_size(const trx_comm *comm); /* as the communicator reports it */
/* In-place all-gather (ncclAllGather over xGMI) of `records_per_rank` hit records per rank, enqueued on `stream`:
 * d_flat holds world blocks of records_per_rank records, this rank's block (already written by its trace launches)
 * at rank * records_per_rank.  For a batch of m frames records_per_rank = m * R. */
int trx_gather_shards(trx_comm *comm, trx_hit *d_flat, uint64_t records_per_rank, void *stream);
/* The same gather to ONE rank (SURVEY 8(e)'s alternative: ncclSend / ncclRecv in one group): rank `root` receives the
 * other ranks' blocks into d_flat, every other rank sends its own block and receives nothing, so a host that
 * consumes the frames on one GPU moves 1/world of the all-gather's bytes.  Only the root may then assemble. */
int trx_gather_shards_root(trx_comm *comm, trx_hit *d_flat, uint64_t records_per_rank, int root, void *stream);
/* De-interleave a gathered [world][n_frames][records_per_frame] buffer into n_frames row-major width x height
 * frames (d_frames: n_frames * width * height records), enqueued on `stream`.  Needs no communicator: with
 * world == 1 it turns one shard-layout frame into an image-layout one. */
int trx_assemble_frames(const trx_hit *d_flat, uint64_t records_per_frame, uint32_t width, uint32_t height,
                        uint32_t world, uint32_t n_frames, trx_hit *d_frames, void *stream);

/* ---- host side: CWBVH construction (CPU) -------------------------------------
 * Stands in for obvhs build_cwbvh_from_tris / build_cwbvh (called at
 * src/cwbvh.rs:97,132), which the Rust host keeps doing in a real
 * integration.  Node encoding follows embree/src/bvh_embree_to_cwbvh.rs:85-186
 * and child ordering embree/src/bvh_embree.rs:284-349. */
typedef struct trx_bvh trx_bvh; /* opaque CwBvh {nodes, primitive_indices, total_aabb} */

/* verts: n_tris * 9 floats (v0,v1,v2).  max_prims_per_leaf 1..3
 * (src/main.rs:176-178).  threads <= 0: all cores. */
int trx_bvh_build_tris(const float *verts, uint64_t n_tris, uint32_t max_prims_per_leaf,
                       int threads, trx_bvh **out);
/* aabbs: n * 6 floats (min xyz, max xyz) — the TLAS build over BLAS AABBs
 * (src/cwbvh.rs:114,132). */
int trx_bvh_build_aabbs(const float *aabbs, uint64_t n, uint32_t max_prims_per_leaf,
                        int threads, trx_bvh **out);
/* SAH weights of the 8-wide collapse for subsequent builds (process-wide): cost of visiting a node
 * and of testing one primitive, the knobs behind the reference's --collapse-traversal-cost
 * (src/main.rs:158-163, swept by src/auto_tune.rs:20-28).  Defaults 1.0 / 0.3. */
int trx_set_build_costs(float traversal_cost, float prim_cost);
/* BVH2 reinsertion pass for subsequent builds (process-wide): per iteration the `batch_ratio`
 * fraction of the nodes with the largest area is re-placed where the tree's summed area drops the
 * most (Meister & Bittner 2018) — obvhs' `reinsertion_batch_ratio`, the reference's `-r`
 * (src/main.rs:113-118).  batch_ratio 0 switches the pass off.  Default 0.02 x 4 iterations.  Applied to
 * triangle builds only: over instance boxes (TLAS) it measured worse and is skipped. */
int trx_set_build_reinsertion(float batch_ratio, int iterations);
/* How the candidates of a reinsertion iteration are batched, for subsequent builds (process-wide).  0 (default): the
 * pipeline's own way - one at a time, each search on the tree the previous move left (trx_flat_build and the presets), or
 * batches of 128 searched concurrently (trx_flat_build_params).  1: ONE batch per iteration - every candidate searches
 * the tree the previous iteration left, the paper's formulation - and the searches run as a kernel on the build device
 * when one is set (trx_set_build_device; one thread per candidate), on the host cores otherwise: the same tree either
 * way.  An iteration gains less than with small batches (more searches are stale when their moves are applied), so give
 * the pass more of them (trx_set_build_reinsertion). */
int trx_set_build_reinsertion_batches(int whole_iterations);
/* The reference's --preset names (src/main.rs:125-131,563-570: "fastest_build" ... "very_slow_build", "" =
 * defaults) mapped onto this builder's knobs — SAH bins, exact-sweep threshold, reinsertion ratio and
 * iterations, pre-splitting (on from "slow_build"); "medium_build" is the default setting and "" restores it.
 * Process-wide, for subsequent builds. */
int trx_set_build_preset(const char *name);
/* Pre-splitting (obvhs `pre_split`, the reference's --split, src/main.rs:572): triangles whose boxes are mostly
 * empty are cut into several references with clipped, tighter boxes before the build, at most `extra_ratio` x
 * n_tris extra references (0 = off, the reference's default).  Such triangles then appear more than once in
 * the permuted triangle buffer of trx_flat (`tri_source` still names the input triangle).  Triangle builds
 * only; process-wide, for subsequent builds. */
int trx_set_build_split(float extra_ratio);
void trx_bvh_destroy(trx_bvh *bvh);
uint64_t trx_bvh_node_count(const trx_bvh *bvh);
uint64_t trx_bvh_prim_count(const trx_bvh *bvh);
const void *trx_bvh_nodes(const trx_bvh *bvh);                 /* node_count * 80 B */
const uint32_t *trx_bvh_primitive_indices(const trx_bvh *bvh); /* prim_count u32     */
void trx_bvh_total_aabb(const trx_bvh *bvh, float out6[6]);
double trx_bvh_build_seconds(const trx_bvh *bvh);

/* cwbvh_gpu_runner (src/rt_gpu/mod.rs:16-112) as one call: builds one BLAS per
 * object (or one flattened BLAS when use_tlas == 0, src/main.rs:300-308),
 * permutes triangles into primitive_indices order, offsets primitive_base_idx,
 * optionally builds the TLAS and concatenates, and returns the flat buffers
 * the GPU path consumes.  Output arrays are owned by the trx_flat and freed
 * with trx_flat_destroy. */
typedef struct trx_flat {
    void *bvh_bytes;
    uint64_t n_nodes;
    float *tri_verts;        /* n_tris * 9 floats, TRX_TRI_VERTS_36, permuted */
    uint64_t n_tris;
    uint32_t *instance_offsets;
    uint32_t n_instances;
    uint32_t tlas_start;
    uint32_t *tri_source;    /* n_tris: index of each permuted triangle in the input */
    uint32_t *blas_tri_start;/* n_blas + 1: first permuted triangle of each BLAS (geometry_id lookup) */
    uint32_t n_blas;
    double blas_build_s;
    double tlas_build_s;
    float *tri_boxes;        /* n_tris * 6 floats (min xyz, max xyz): the box each triangle entry was built with —
                              * its own, or the clipped part a pre-split reference covers (trx_set_build_split) */
    uint32_t *instance_source;   /* n_instances: which of the caller's objects (trx_flat_build) or instances
                                  * (trx_flat_build_instanced) TLAS primitive k is */
    float *instance_transforms;  /* n_instances * 16 (object-to-world, column-major) in TLAS-primitive order, ready for
                                  * trx_scene_set_instance_transforms; NULL when no transforms were given */
    uint32_t *instance_entry_nodes; /* n_instances, ready for trx_scene_set_instance_entry_nodes: the node of its BLAS at
                                     * which TLAS primitive k starts (re-braided TLAS, trx_set_build_rebraid); NULL when
                                     * every primitive is a whole BLAS (the reference's layout) */
} trx_flat;
int trx_flat_build(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects,
                   int use_tlas, uint32_t max_prims_per_leaf, int threads, trx_flat **out);
void trx_flat_destroy(trx_flat *flat);
/* One BLAS per object and a TLAS over n_instances INSTANCES of them (true instancing: several instances may name
 * the same object).  instance_object[k] = object of instance k; instance_object_to_world = n_instances column-major
 * affine 4x4 matrices (NULL = all identity).  TLAS boxes bound the transformed BLAS boxes.  The result's
 * instance_offsets / instance_transforms / instance_source are in TLAS-primitive order. */
int trx_flat_build_instanced(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects,
                             const uint32_t *instance_object, const float *instance_object_to_world,
                             uint32_t n_instances, uint32_t max_prims_per_leaf, int threads, trx_flat **out);

/* The reference's BvhBuildParams (src/main.rs:571-585), field for field, for callers that carry one around.
 * trx_flat_build_params runs the ploc_cwbvh pipeline with them: Morton sort at sort_precision bits (64 | 128,
 * "Unsupported sort precision" otherwise, src/main.rs:576-580), PLOC merging with ploc_search_distance (1..32)
 * places either side and distance 1 for the first search_depth_threshold rounds, reinsertion at
 * reinsertion_batch_ratio, optional pre_split, collapse to <= max_prims_per_leaf triangles per leaf at
 * collapse_traversal_cost.  post_collapse_reinsertion_batch_ratio_multiplier is "For BVH2 only" in the reference
 * (src/main.rs:119-123) and has no effect on a CWBVH build. */
typedef struct trx_build_params {
    uint32_t pre_split;                 /* --split */
    uint32_t ploc_search_distance;      /* --search-distance */
    uint32_t search_depth_threshold;    /* --search-depth-threshold */
    float reinsertion_batch_ratio;      /* -r */
    uint32_t sort_precision;            /* --sort-precision: 64 | 128 */
    uint32_t max_prims_per_leaf;        /* --max-prims-per-leaf: 1..3 for CWBVH */
    float post_collapse_reinsertion_batch_ratio_multiplier;
    float collapse_traversal_cost;      /* --collapse-traversal-cost */
} trx_build_params;
void trx_build_params_default(trx_build_params *params); /* the reference's command-line defaults */
/* Which stages of subsequent builds run on a HIP device, as kernels (process-wide): device >= 0 = that device, -1 = the
 * host cores only (default).  Objects of at least 32,768 primitives take the device stages; the many small BLASes of a
 * TLAS scene stay on the host cores.  Stages: the BVH2 stage of trx_flat_build_params (Morton sort + PLOC merge rounds);
 * the whole-iteration reinsertion pass (trx_set_build_reinsertion_batches(1)) with the tree resident on the device -
 * candidate selection, searches, the choice and application of the moves, the boxes; and, for every builder, the BVH2 -> CWBVH stage (cost table,
 * collapse, slot assignment, node encoding, primitive order).  Each device stage returns the very bytes its host twin
 * returns (same operations in the same order), so a build is the same build wherever it ran.  A device failure fails
 * the build (TRX_ERR_NO_DEVICE / TRX_ERR_OOM): nothing falls back silently. */
int trx_set_build_device(int device);
/* TLAS of trx_flat_build / trx_flat_build_params (process-wide): a BLAS whose box exceeds `area_fraction` of the scene
 * box's surface area is referenced through the subtrees under its root instead of as a whole, largest first, as long as
 * the opened node has inner children only; trx_flat.instance_entry_nodes then names each primitive's entry node.
 * Default 1/4096; 0 = never (every primitive a whole BLAS, as the reference builds it, src/cwbvh.rs:108-137).  Instanced
 * builds (trx_flat_build_instanced) are never re-braided. */
int trx_set_build_rebraid(float area_fraction);
int trx_flat_build_params(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects,
                          int use_tlas, const trx_build_params *params, int threads, trx_flat **out);

/* A preset's build with every stage on HIP device `device` (the ploc_cwbvh pipeline: Morton sort + PLOC rounds with the
 * reference's search parameters, src/main.rs:85-98; reinsertion in whole-iteration batches; collapse + encoding) and a
 * reinsertion budget per preset name (src/main.rs:565-570): bistro-class medium_build in 0.4 s against 0.96 s on 16 host
 * cores, walked with no more node visits.  device < 0: the same pipeline on the host cores (same bytes). */
int trx_flat_build_preset_device(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                                 const char *preset, uint32_t max_prims_per_leaf, int threads, int device, trx_flat **out);

/* ---- host side: scenes ----------------------------------------------------
 * The reference's assets are absent: seeded procedural
 * stand-ins with the triangle counts of README.md:27-34.  name is one of
 * "cornell" "demoscene" "kitchen" "bistro" "hairball" "san_miguel" "soup".
 * Returns a malloc'd vertex array (n_tris*9 floats) and per-object triangle
 * counts; free with trx_free. */
int trx_gen_scene(const char *name, uint64_t n_tris_target, uint64_t seed, float **out_verts,
                  uint64_t *out_n_tris, uint64_t **out_object_counts, uint32_t *out_n_objects);
/* Camera of assets/scenes/<name>.ron (eye, look_at, fov) for the stand-in. */
int trx_scene_camera(const char *name, float eye[3], float look_at[3], float *fov_deg);
/* load_meshs (src/main.rs:494-561): Wavefront OBJ (tri + quad fan, one object
 * per `o`) or the JSON triangle list. */
int trx_load_model(const char *path, float **out_verts, uint64_t *out_n_tris,
                   uint64_t **out_object_counts, uint32_t *out_n_objects);
/* A scene file of the reference (assets/scenes/<name>.ron: model_path, camera) as src/main.rs:259-298 reads it: the
 * camera, and the model loaded from the path the reference's rule gives (scene path and model path both relative: the
 * model is taken relative to the scene file's great-grandparent directory, src/main.rs:271-284). */
int trx_load_scene(const char *path, float **out_verts, uint64_t *out_n_tris, uint64_t **out_object_counts,
                   uint32_t *out_n_objects, float eye[3], float look_at[3], float *fov_deg);
void trx_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* TRX_H */
