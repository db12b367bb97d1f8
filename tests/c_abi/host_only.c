/* Plain C11 consumer of include/trx.h: everything here works without a GPU.
 * Built and run by tests/test_abi.py (gcc -std=c11 -pedantic -Wall -Werror). */
#include <stdio.h>
#include <string.h>

#include "trx.h"

int main(void) {
    if (trx_abi_version() != TRX_ABI_VERSION) return 1;
    if (trx_tri_format_bytes(TRX_TRI_F16_24) != 24 || trx_tri_format_bytes(TRX_TRI_VERTS_36) != 36) return 2;
    trx_shard sh;
    memset(&sh, 0, sizeof sh);
    sh.index = 3;
    sh.count = 8;
    sh.layout = TRX_LAYOUT_SHARD;
    if (trx_shard_tiles(1920, 1080, sh) != 4050) return 3;

    float *verts = NULL;
    uint64_t n_tris = 0, *counts = NULL;
    uint32_t n_objects = 0;
    if (trx_gen_scene("cornell", 0, 1, &verts, &n_tris, &counts, &n_objects) != TRX_OK) return 4;
    trx_flat *flat = NULL;
    if (trx_flat_build(verts, counts, n_objects, 1, 3, 2, &flat) != TRX_OK) return 5;
    if (flat->n_tris != n_tris || flat->n_instances == 0 || flat->tlas_start == 0) return 6;

    float eye[3], look[3], fov = 0.f;
    trx_view view;
    if (trx_scene_camera("cornell", eye, look, &fov) != TRX_OK) return 7;
    if (trx_view_from_camera(eye, look, fov, 64.f, 64.f, &view) != TRX_OK) return 8;

    /* invalid input is reported through the status + trx_last_error, never a crash */
    if (trx_flat_build(verts, counts, n_objects, 0, 4, 0, &flat) == TRX_OK) return 9;
    if (strstr(trx_last_error(), "maximum of 3 primitives") == NULL) return 10;

    printf("objects %u triangles %llu nodes %llu instances %u devices %d\n", n_objects, (unsigned long long)n_tris,
           (unsigned long long)flat->n_nodes, flat->n_instances, trx_device_count());
    trx_flat_destroy(flat);
    trx_free(verts);
    trx_free(counts);
    return 0;
}
