/*
 * trx_oracle.h — CPU restatement of tray_racing's CWBVH closest-hit path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tray_racing_amd/ may include, link
 * or call this.  Users: tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for
 * this path, its CPU implementation lives in the un-vendored, un-pinned
 * `obvhs` git dependency (Cargo.toml:26-29, no rev; Cargo.lock git-ignored),
 * and neither Rust nor dxc exists in the build image, so the reference cannot
 * be executed.  This restatement follows the in-tree HLSL line by line and is
 * cross-checked against a BVH-independent brute-force query (orc_brute_*),
 * which pins it to the *specification* of the triangle test, not to outputs of
 * the reference.
 *
 * All paths below are relative to the tray_racing checkout.
 */
#ifndef TRX_ORACLE_H
#define TRX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same bit values as trx_semantics in include/trx.h */
#define ORC_SEM_HLSL 0u
#define ORC_SEM_NODE_RCP 1u
#define ORC_SEM_TIE_FIRST 2u
#define ORC_SEM_NODE_FMA 4u
#define ORC_SEM_CPU 3u

#define ORC_STACK_SIZE 64

typedef struct orc_hit {
    float t;
    uint32_t prim;
} orc_hit;

typedef struct orc_ray {
    float origin[3];
    float tmin;
    float direction[3];
    float tmax;
} orc_ray;

typedef struct orc_view {
    float view_inv[16];
    float proj_inv[16];
    float eye[3];
    float exposure;
    uint32_t tlas_start;
    uint32_t pad[3];
} orc_view;

/* Oracle scene: borrowed pointers.  tris = n_tris * 9 floats {v0, e1 = v0 - v1,
 * e2 = v2 - v0} (see orc_tris_from_*). */
typedef struct orc_scene {
    const uint32_t *nodes; /* n_nodes * 20 u32 (80 B) */
    uint64_t n_nodes;
    const float *tris;
    uint64_t n_tris;
    const uint32_t *instance_offsets;
    uint32_t n_instances;
    uint32_t tlas_start;
    /* optional: world-to-object transform of every TLAS primitive (instance), n_instances * 12 floats, three rows
     * of {m0 m1 m2 t}: the ray is taken into the instance's object space on BLAS entry and restored on exit
     * (the TODOs at query_tlas.hlsl:409,433,484; get_instance_transform, traversable/src/lib.rs:25-27).  NULL =
     * identity, the reference's behaviour. */
    const float *instance_w2o;
    /* optional: the node of its BLAS at which TLAS primitive k starts its walk (n_instances entries; re-braided
     * scenes, trx_scene_set_instance_entry_nodes).  NULL = node 0, the reference's rule (query_tlas.hlsl:443). */
    const uint32_t *instance_entry;
} orc_scene;

typedef struct orc_stats {
    uint64_t n_rays, n_node, n_tri, n_hits;
    uint32_t max_stack, overflow;
    double seconds; /* wall-clock of the frame loop */
    int threads;
    /* two-level scenes: node visits spent in the TLAS (n_node counts both levels) and BLAS (sub)trees entered */
    uint64_t n_tlas_node, n_inst_enter;
} orc_stats;

/* triangle format conversion to {v0,e1,e2} */
void orc_tris_from_verts(const float *verts, uint64_t n, float *out9);
void orc_tris_from_f16(const void *tri24, uint64_t n, float *out9);

/* camera: src/main.rs:602-616 */
void orc_view_from_camera(const float eye[3], const float look_at[3], float fov_deg, float width,
                          float height, orc_view *out);

/* primitives of the path (exported so tests can probe them one by one) */
uint32_t orc_octant_inv4(const float d[3]);
uint32_t orc_node_intersect(const float o[3], const float d[3], const float inv_d[3], uint32_t oct_inv4,
                            float max_distance, const uint32_t node[20], uint32_t sem);
int orc_intersect_tri(const float o[3], const float d[3], const float tri9[9], float tmin, float *t,
                      uint32_t sem);
void orc_primary_ray(const orc_view *view, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                     float o[3], float d[3]);
int orc_ao_ray(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t px,
               uint32_t py, orc_hit primary, uint32_t frame, float ao_eps, float o[3], float d[3]);
uint32_t orc_uhash(uint32_t a, uint32_t b);
float orc_hash_noise(uint32_t x, uint32_t y, uint32_t frame);
void orc_sincos(float theta, float *s, float *c);
void orc_set_ao_libm(int on); /* measurement aid: 1 = libm sinf / cosf, 2 = correctly rounded sin / cos in the AO direction */
/* binary32 values with bits lo..hi (step `stride`) whose orc_sincos differs from this platform's sinf / cosf */
uint64_t orc_sincos_libm_mismatches(uint32_t lo_bits, uint32_t hi_bits, uint32_t stride);
/* Node test over the eight children at once in AVX2 registers instead of the scalar loop: the same IEEE operations
 * per child, so every result is bit-identical (tests/test_oracle.py asserts it on the goldens and on whole frames);
 * it is what bench.py's cpu_baseline leg times, because the reference's CPU node test (obvhs) is SIMD as well.
 * Takes effect only on a CPU with AVX2 + FMA; orc_get_simd() says whether it did. */
void orc_set_simd(int on);
int orc_get_simd(void);

/* single ray through the CWBVH (BLAS-only when n_instances == 0) */
orc_hit orc_traverse(const orc_scene *s, const float o[3], const float d[3], float tmin, float tmax,
                     uint32_t sem, orc_stats *st);

/* the same with the TLAS primitive (instance) the hit was found in: *inst = index into instance_offsets, or
 * 0xFFFFFFFF for a miss / a scene without TLAS */
orc_hit orc_traverse_inst(const orc_scene *s, const float o[3], const float d[3], float tmin, float tmax,
                          uint32_t sem, orc_stats *st, uint32_t *inst);
int orc_ao_ray_inst(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                    orc_hit primary, uint32_t primary_inst, uint32_t frame, float ao_eps, float o[3], float d[3]);
/* point / direction through a 3x4 transform with the operation order both sides use:
 * ((m0*x + m1*y) + m2*z) [+ t] per row */
void orc_xform_point(const float m[12], const float p[3], float out[3]);
void orc_xform_dir(const float m[12], const float v[3], float out[3]);

/* frames; threads <= 0: all cores.  shard as trx_shard (8x8 tiles, tile % count == index);
 * pixels outside the shard are left untouched. */
void orc_trace_primary(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h,
                       uint32_t shard_index, uint32_t shard_count, uint32_t sem, int threads,
                       orc_hit *hits, orc_stats *st);
void orc_trace_ao(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t shard_index,
                  uint32_t shard_count, uint32_t sem, uint32_t frame, float ao_eps, int threads,
                  const orc_hit *primary, orc_hit *ao, orc_stats *st);
void orc_trace_rays(const orc_scene *s, const orc_ray *rays, uint64_t n, uint32_t sem, int threads,
                    orc_hit *hits, orc_stats *st);
/* instance-aware forms (inst arrays may be NULL; primary_inst is needed when instance_w2o is set) */
void orc_trace_primary_inst(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h,
                            uint32_t shard_index, uint32_t shard_count, uint32_t sem, int threads,
                            orc_hit *hits, uint32_t *inst, orc_stats *st);
void orc_trace_ao_inst(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t shard_index,
                       uint32_t shard_count, uint32_t sem, uint32_t frame, float ao_eps, int threads,
                       const orc_hit *primary, const uint32_t *primary_inst, orc_hit *ao, uint32_t *ao_inst,
                       orc_stats *st);
void orc_trace_rays_inst(const orc_scene *s, const orc_ray *rays, uint64_t n, uint32_t sem, int threads,
                         orc_hit *hits, uint32_t *inst, orc_stats *st);
/* the reference's whole CPU frame (src/rt_cpu/rt_cpu.rs:35-92): primary + AO + shade per pixel */
double orc_render_frame(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t sem,
                        uint32_t frame, float ao_eps, int threads, float *rgb);

/* per-ray node / triangle test counts of a primary frame */
void orc_count_primary_per_ray(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, uint32_t sem,
                               int threads, uint16_t *n_node, uint16_t *n_tri);

/* BVH-independent ground truth: every ray against every triangle in index
 * order with the same triangle test and tie rule. */
void orc_brute_rays(const float *tris9, uint64_t n_tris, const orc_ray *rays, uint64_t n, uint32_t sem,
                    int threads, orc_hit *hits);
void orc_brute_primary(const float *tris9, uint64_t n_tris, const orc_view *view, uint32_t w,
                       uint32_t h, uint32_t sem, int threads, orc_hit *hits);

/* structural check of a CWBVH against its triangles ("validate", src/cwbvh.rs:102-104).
 * verts: the permuted TRX_TRI_VERTS_36 triangles the nodes index.  Returns 0
 * when sound; otherwise a negative code and a message in err. */
/* boxes (optional, n_tris * 6 floats): per-primitive build boxes; given, leaf boxes must contain these instead of
 * whole triangles (pre-split references cover part of a triangle) */
int orc_validate(const orc_scene *s, const float *verts, const float *boxes, char *err, int err_len);

#ifdef __cplusplus
}
#endif
#endif
