/* The reference's CPU pixel loop over the LITERAL Traversable::traverse (src/rt_cpu/rt_cpu.rs:35-57: every worker of
 * a thread pool calls scene.traverse(ray) for its pixels, one ray per call) on the HIP backend: T host threads, thread k
 * takes the pixels i = k, k + T, k + 2T ... of the frame and calls trx_traverse1 for each.  Rays come from a file
 * (32-byte trx_ray records, written by the test from the oracle's primary-ray generator), RayHits go to a file
 * (16-byte trx_rayhit records); stdout: "<rays> <seconds> <launches>".
 * usage: traverse_threads <scene> <tris> <threads> <semantics> <rays.bin> <hits.bin> [semantics of the odd threads [tlas]]
 * (an eighth argument builds the scene two-level: the single-ray path of two-level scenes is the launch combiner, that of
 * single-level scenes the resident ray service) */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "trx.h"
#include "trx_dev.h"

#define CHECK(call)                                                     \
    do {                                                                \
        if ((call) != TRX_OK) {                                         \
            fprintf(stderr, "%s: %s\n", #call, trx_last_error());       \
            return 1;                                                   \
        }                                                               \
    } while (0)

struct job {
    trx_scene *scene;
    const trx_ray *rays;
    trx_rayhit *hits;
    uint64_t n, first, stride;
    uint32_t sem;
    int rc;
};

static void *worker(void *arg) {
    struct job *j = (struct job *)arg;
    for (uint64_t i = j->first; i < j->n; i += j->stride) {
        if (trx_traverse1(j->scene, &j->rays[i], j->sem, &j->hits[i]) != TRX_OK) {
            fprintf(stderr, "trx_traverse1: %s\n", trx_last_error());
            j->rc = 1;
            return NULL;
        }
    }
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 7) return 2;
    const int threads = atoi(argv[3]);
    const uint32_t sem = (uint32_t)atoi(argv[4]);
    const uint32_t sem_odd = argc > 7 ? (uint32_t)atoi(argv[7]) : sem; /* callers of mixed semantics at once */
    const int tlas = argc > 8 ? atoi(argv[8]) : 0;
    if (threads < 1 || threads > 4096) return 2;
    FILE *f = fopen(argv[5], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    const uint64_t n = (uint64_t)bytes / sizeof(trx_ray);
    trx_ray *rays = (trx_ray *)malloc((size_t)bytes);
    trx_rayhit *hits = (trx_rayhit *)calloc(n ? n : 1, sizeof(trx_rayhit));
    if (!rays || !hits || fread(rays, sizeof(trx_ray), n, f) != n) return 3;
    fclose(f);

    float *verts = NULL;
    uint64_t n_tris = 0, *counts = NULL;
    uint32_t n_objects = 0;
    CHECK(trx_gen_scene(argv[1], (uint64_t)atoll(argv[2]), 1, &verts, &n_tris, &counts, &n_objects));
    trx_flat *flat = NULL;
    CHECK(trx_flat_build(verts, counts, n_objects, tlas, 3, 0, &flat));
    trx_scene *scene = NULL;
    CHECK(trx_scene_create(flat->bvh_bytes, flat->n_nodes, flat->tri_verts, flat->n_tris, TRX_TRI_VERTS_36,
                           flat->n_instances ? flat->instance_offsets : NULL, flat->n_instances, flat->tlas_start, 0, &scene));
    if (flat->instance_entry_nodes && flat->n_instances)
        CHECK(trx_scene_set_instance_entry_nodes(scene, flat->instance_entry_nodes, flat->n_instances));
    /* one ray ahead of the clock: the first call creates the combiner (pinned buffers, streams) */
    if (n) CHECK(trx_traverse1(scene, &rays[0], sem, &hits[0]));

    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    struct job *jobs = (struct job *)malloc(sizeof(struct job) * (size_t)threads);
    if (!tid || !jobs) return 3;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < threads; k++) {
        jobs[k].scene = scene;
        jobs[k].rays = rays;
        jobs[k].hits = hits;
        jobs[k].n = n;
        jobs[k].first = (uint64_t)k;
        jobs[k].stride = (uint64_t)threads;
        jobs[k].sem = (k & 1) ? sem_odd : sem;
        jobs[k].rc = 0;
        if (pthread_create(&tid[k], NULL, worker, &jobs[k]) != 0) return 4;
    }
    int rc = 0;
    for (int k = 0; k < threads; k++) {
        pthread_join(tid[k], NULL);
        rc |= jobs[k].rc;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc) return 5;
    uint64_t launches = 0, served = 0;
    CHECK(trx_debug_traverse1_stats(scene, &launches, &served));
    f = fopen(argv[6], "wb");
    if (!f || fwrite(hits, sizeof(trx_rayhit), n, f) != n) return 3;
    fclose(f);
    printf("%llu %.6f %llu\n", (unsigned long long)n, (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec),
           (unsigned long long)launches);
    free(tid);
    free(jobs);
    free(rays);
    free(hits);
    trx_scene_destroy(scene);
    trx_flat_destroy(flat);
    trx_free(verts);
    trx_free(counts);
    return 0;
}
