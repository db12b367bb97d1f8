/* Plain C11 consumer of include/trx.h on a GPU: the flow of cwbvh_gpu_runner (src/rt_gpu/mod.rs:16-112) —
 * flat buffers in, one primary + AO frame out — printing a checksum tests/test_gpu_parity.py compares with the
 * oracle's frame.  usage: trace_frame <scene> <tris> <width> <height> <semantics> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "trx.h"

#define CHECK(call)                                                     \
    do {                                                                \
        if ((call) != TRX_OK) {                                         \
            fprintf(stderr, "%s: %s\n", #call, trx_last_error());       \
            return 1;                                                   \
        }                                                               \
    } while (0)

static uint64_t fnv(const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const uint32_t w = (uint32_t)atoi(argv[3]), h = (uint32_t)atoi(argv[4]), sem = (uint32_t)atoi(argv[5]);
    float *verts = NULL;
    uint64_t n_tris = 0, *counts = NULL;
    uint32_t n_objects = 0;
    CHECK(trx_gen_scene(argv[1], (uint64_t)atoll(argv[2]), 1, &verts, &n_tris, &counts, &n_objects));
    trx_flat *flat = NULL;
    CHECK(trx_flat_build(verts, counts, n_objects, 0, 3, 0, &flat));
    trx_scene *scene = NULL;
    CHECK(trx_scene_create(flat->bvh_bytes, flat->n_nodes, flat->tri_verts, flat->n_tris, TRX_TRI_VERTS_36, NULL, 0, 0, 0,
                           &scene));
    float eye[3], look[3], fov = 0.f, ms = 0.f;
    trx_view view;
    CHECK(trx_scene_camera(argv[1], eye, look, &fov));
    CHECK(trx_view_from_camera(eye, look, fov, (float)w, (float)h, &view));
    trx_hit *primary = (trx_hit *)malloc(sizeof(trx_hit) * w * h), *ao = (trx_hit *)malloc(sizeof(trx_hit) * w * h);
    if (!primary || !ao) return 3;
    CHECK(trx_trace_primary_ao(scene, &view, w, h, sem, 5, 0.01f, primary, ao, &ms));
    printf("%016llx %016llx %llu\n", (unsigned long long)fnv(primary, sizeof(trx_hit) * w * h),
           (unsigned long long)fnv(ao, sizeof(trx_hit) * w * h), (unsigned long long)flat->n_nodes);
    free(primary);
    free(ao);
    trx_scene_destroy(scene);
    trx_flat_destroy(flat);
    trx_free(verts);
    trx_free(counts);
    return 0;
}
