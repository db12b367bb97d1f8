/* spec_sim.c - development aid (CPU, links the oracle): how many TRIPS does one ray's walk take when every trip may
 * test up to G pending nodes of the ray ahead of the walk (tests/analysis/spec_sim.py drives it)?
 *
 * The sequential walk (query.hlsl:328-438) visits its nodes in depth-first order; every hit bit of the current group and
 * of the stacked groups names a node that WILL be fetched and tested (nothing on the stack is culled later), and what a
 * node test computes per child - tmin = max(near planes, 1e-4), tfar = min(far planes) - does not depend on the ray's
 * current t: only the final `tmin <= min(tfar, t)` does.  So a test made ahead of time can be parked and filtered
 * with the t of the moment the walk reaches the node: same mask.  Model:
 *   pending = the nodes named so far, in depth-first order; `rec` = a parked test exists.
 *   a trip: the first G pending nodes without a record are fetched and tested (one memory round trip for all);
 *           then the walk consumes pending nodes from the front while they carry records.
 *   variant A ("full"): a record also carries the node's triangle tests (tt per triangle is t-independent as well), so
 *           consumption needs no memory at all;
 *   variant B ("nodes"): triangles are fetched when the walk gets there - consumption stops behind a node whose
 *           filtered mask names triangles (they travel with the next trip's node fetches).
 * Output per ray: node visits, trips of A and B for G = 1, 2, 4, 8.
 */
#include "../../oracle/trx_oracle.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t node;
    uint8_t rec;
} pend_t;

#define MAXP 4096

static uint32_t sim_one(const orc_scene *s, const orc_ray *ray, uint32_t sem, int G, int cap, int variant, uint32_t *n_nodes) {
    float o[3], d[3], inv[3];
    for (int k = 0; k < 3; k++) {
        o[k] = ray->origin[k];
        d[k] = ray->direction[k] == 0.0f ? 1.1920929e-7f : ray->direction[k];
        inv[k] = 1.0f / d[k];
    }
    const uint32_t oct4 = orc_octant_inv4(d);
    float t = ray->tmax;
    static __thread pend_t pend[MAXP];
    int np = 0; /* pend[np-1] is the FRONT (next to visit) */
    pend[np].node = 0;
    pend[np].rec = 0;
    np++;
    uint32_t trips = 0, visited = 0;
    int tri_blocked = 0; /* variant B: triangles of the last consumed node still to be fetched */
    uint32_t tri_base = 0, tri_bits = 0;
    for (;;) {
        /* ---- a trip: fetch + test up to G unrecorded nodes in depth-first order (records outstanding <= cap) */
        int outstanding = 0;
        for (int i = 0; i < np; i++) outstanding += pend[i].rec;
        int taken = 0;
        for (int i = np - 1; i >= 0 && taken < G && outstanding < cap; i--)
            if (!pend[i].rec) {
                pend[i].rec = 1;
                taken++;
                outstanding++;
            }
        trips++;
        if (tri_blocked) { /* B: the pending triangles arrive with this trip */
            for (uint32_t m = tri_bits; m;) {
                const int b = 31 - __builtin_clz(m);
                m &= ~(1u << b);
                orc_intersect_tri(o, d, s->tris + (size_t)(tri_base + b) * 9, ray->tmin, &t, sem);
            }
            tri_blocked = 0;
        }
        /* ---- consume */
        while (np > 0 && pend[np - 1].rec) {
            const uint32_t node = pend[--np].node;
            visited++;
            const uint32_t *n = s->nodes + (size_t)node * 20;
            const uint32_t mask = orc_node_intersect(o, d, inv, oct4, t, n, sem);
            const uint32_t imask = n[3] >> 24, child_base = n[4], prim_base = n[5];
            /* children go on the pending list so that the highest bit is visited first */
            for (int b = 24; b < 32; b++)
                if (mask & (1u << b)) {
                    const uint32_t slot = (uint32_t)(b - 24) ^ (oct4 & 0xffu);
                    const uint32_t child = child_base + (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u));
                    if (np < MAXP) {
                        pend[np].node = child;
                        pend[np].rec = 0;
                        np++;
                    }
                }
            const uint32_t tb = mask & 0x00ffffffu;
            if (tb) {
                if (variant == 0) {
                    for (uint32_t m = tb; m;) {
                        const int b = 31 - __builtin_clz(m);
                        m &= ~(1u << b);
                        orc_intersect_tri(o, d, s->tris + (size_t)(prim_base + b) * 9, ray->tmin, &t, sem);
                    }
                } else {
                    tri_blocked = 1;
                    tri_base = prim_base;
                    tri_bits = tb;
                    break;
                }
            }
        }
        if (np == 0 && !tri_blocked) break;
        if (trips > 100000u) break;
    }
    if (tri_blocked) trips++; /* the last triangles */
    *n_nodes = visited;
    return trips;
}

/* out: [n][9] = {nodes, A1, A2, A4, A8, B1, B2, B4, B8} */
void spec_sim(const orc_scene *s, const orc_ray *rays, uint64_t n, uint32_t sem, int cap, uint32_t *out) {
#pragma omp parallel for schedule(dynamic, 256)
    for (uint64_t i = 0; i < n; i++) {
        uint32_t nn = 0;
        for (int v = 0; v < 2; v++)
            for (int g = 0; g < 4; g++) out[i * 9 + 1 + v * 4 + g] = sim_one(s, rays + i, sem, 1 << g, cap, v, &nn);
        out[i * 9] = nn;
    }
}

/* the AO rays of a frame (one per primary hit), as explicit rays; returns their number */
uint64_t spec_ao_rays(const orc_scene *s, const orc_view *view, uint32_t w, uint32_t h, const orc_hit *primary, uint32_t frame,
                      float ao_eps, orc_ray *out) {
    uint64_t n = 0;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            orc_ray r;
            if (!orc_ao_ray(s, view, w, h, x, y, primary[(size_t)y * w + x], frame, ao_eps, r.origin, r.direction)) continue;
            r.tmin = 0.0f;
            r.tmax = 3.402823466e+38f;
            out[n++] = r;
        }
    return n;
}

/* ---- variant D: a prefetcher that walks AHEAD of the consumer -------------------------------------------------------
 * consumer  = the sequential walk, whose "fetch + test node" is a look-up in a table of parked tests (keyed by node index,
 *             direct-mapped, `cache` entries, newer tests overwrite); a miss makes that node candidate 0 of the next trip.
 * prefetcher = its own stack F of {node, tmin}: a tested node's children that pass with the t of the moment go on F
 *             (nearest last = on top), whether or not the consumer has reached the node; a trip pops up to G - 1 (or G)
 *             entries whose tmin still passes the current t and that are not parked yet, and tests them.
 * Nothing here can change the result: parked tests are t-independent, the consumer filters with its own t.
 * Returns trips; *tests = node tests made (>= nodes visited: some are wasted). */
typedef struct { uint32_t node; float tmin; } fent_t;
static uint32_t sim_deep(const orc_scene *s, const orc_ray *ray, uint32_t sem, int G, int cache, uint32_t *n_nodes, uint32_t *n_tests) {
    float o[3], d[3], inv[3];
    for (int k = 0; k < 3; k++) {
        o[k] = ray->origin[k];
        d[k] = ray->direction[k] == 0.0f ? 1.1920929e-7f : ray->direction[k];
        inv[k] = 1.0f / d[k];
    }
    const uint32_t oct4 = orc_octant_inv4(d);
    float t = ray->tmax;
    static __thread fent_t F[MAXP];
    static __thread uint32_t tag[1024];
    static __thread uint32_t cstack[MAXP]; /* consumer: pending nodes, DFS order (top = next) */
    for (int i = 0; i < cache; i++) tag[i] = 0xffffffffu;
    int nf = 0, nc = 0;
    cstack[nc++] = 0;
    F[nf++] = (fent_t){0, 0.0f};
    uint32_t trips = 0, visited = 0, tests = 0;
    for (;;) {
        /* consume while the next node's test is parked */
        while (nc > 0) {
            const uint32_t node = cstack[nc - 1];
            if (tag[node % (uint32_t)cache] != node) break;
            nc--;
            visited++;
            const uint32_t *n = s->nodes + (size_t)node * 20;
            const uint32_t mask = orc_node_intersect(o, d, inv, oct4, t, n, sem);
            const uint32_t imask = n[3] >> 24, child_base = n[4], prim_base = n[5];
            for (int b = 24; b < 32; b++)
                if (mask & (1u << b)) {
                    const uint32_t slot = (uint32_t)(b - 24) ^ (oct4 & 0xffu);
                    if (nc < MAXP) cstack[nc++] = child_base + (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u));
                }
            for (uint32_t m = mask & 0x00ffffffu; m;) {
                const int b = 31 - __builtin_clz(m);
                m &= ~(1u << b);
                orc_intersect_tri(o, d, s->tris + (size_t)(prim_base + b) * 9, ray->tmin, &t, sem);
            }
        }
        if (nc == 0) break;
        /* a trip: the consumer's blocked node first, then the prefetcher's stack */
        trips++;
        uint32_t cand[8];
        int ncand = 0;
        cand[ncand++] = cstack[nc - 1];
        while (ncand < G && nf > 0) {
            const fent_t e = F[--nf];
            if (e.tmin > t) continue;                               /* culled since it was found */
            if (tag[e.node % (uint32_t)cache] == e.node) continue;  /* parked already */
            int dup = 0;
            for (int k = 0; k < ncand; k++) dup |= cand[k] == e.node;
            if (!dup) cand[ncand++] = e.node;
        }
        for (int k = 0; k < ncand; k++) {
            const uint32_t node = cand[k];
            tests++;
            tag[node % (uint32_t)cache] = node;
            const uint32_t *n = s->nodes + (size_t)node * 20;
            const uint32_t mask = orc_node_intersect(o, d, inv, oct4, t, n, sem); /* children that pass with the t of this moment */
            const uint32_t imask = n[3] >> 24, child_base = n[4];
            for (int b = 24; b < 32; b++)
                if (mask & (1u << b)) {
                    const uint32_t slot = (uint32_t)(b - 24) ^ (oct4 & 0xffu);
                    if (nf < MAXP) F[nf++] = (fent_t){child_base + (uint32_t)__builtin_popcount(imask & ((1u << slot) - 1u)), 0.0f};
                }
        }
        if (trips > 100000u) break;
    }
    *n_nodes = visited;
    *n_tests = tests;
    return trips;
}

/* out: [n][1 + 2 * 4] = {nodes, trips G=2,4,8,16?..}: G in {2, 4, 8}, then tests for the same */
void spec_sim_deep(const orc_scene *s, const orc_ray *rays, uint64_t n, uint32_t sem, int cache, uint32_t *out) {
#pragma omp parallel for schedule(dynamic, 256)
    for (uint64_t i = 0; i < n; i++) {
        uint32_t nn = 0, nt = 0;
        for (int g = 0; g < 3; g++) {
            out[i * 7 + 1 + g] = sim_deep(s, rays + i, sem, 2 << g, cache, &nn, &nt);
            out[i * 7 + 4 + g] = nt;
        }
        out[i * 7] = nn;
    }
}
