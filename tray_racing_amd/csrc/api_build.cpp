// api_build.cpp - host side: builders, the flat-buffer assembly of cwbvh_gpu_runner (src/rt_gpu/mod.rs:16-112),
// scene generators and loaders.
#include "api_internal.h"

namespace {

// Builder settings for subsequent builds (trx_set_build_*): process-wide, guarded by g_build_mu; a build takes
// a snapshot when it starts, so concurrent builds and setters do not race.
struct BuildSettings {
    float traversal_cost = 1.0f, prim_cost = 0.3f;
    float reinsert_ratio = 0.02f;
    int reinsert_iters = 4;
    int sah_bins = 32;
    uint32_t sweep_max = 48;
    float pre_split = 0.0f;
    uint32_t ploc_distance = 0; // 0: binned-SAH BVH2; > 0: PLOC with this search distance
    uint32_t ploc_depth_threshold = 2, ploc_sort_bits = 64;
    int ploc_device = -1;       // >= 0: the PLOC stage of large builds runs on this HIP device (trx_set_build_device)
    bool reinsert_batched = false;
    bool reinsert_whole = false; // one batch per iteration (trx_set_build_reinsertion_batches)
    float rebraid_area = 1.0f / 4096.0f; // TLAS: open BLAS subtrees whose box exceeds this share of the scene box's area (0 = never)
};
BuildSettings g_build;
std::mutex g_build_mu;
BuildSettings build_settings() {
    std::lock_guard<std::mutex> lock(g_build_mu);
    return g_build;
}
BuildParams to_build_params(const BuildSettings &b, uint32_t max_prims, int threads) {
    BuildParams bp;
    bp.max_prims_per_leaf = max_prims;
    bp.threads = threads;
    bp.traversal_cost = b.traversal_cost;
    bp.prim_cost = b.prim_cost;
    bp.reinsertion_batch_ratio = b.reinsert_ratio;
    bp.reinsertion_iterations = b.reinsert_iters;
    bp.sah_bins = b.sah_bins;
    bp.sweep_max = b.sweep_max;
    bp.pre_split_ratio = b.pre_split;
    bp.ploc_search_distance = b.ploc_distance;
    bp.ploc_search_depth_threshold = b.ploc_depth_threshold;
    bp.ploc_sort_bits = b.ploc_sort_bits;
    bp.ploc_device = b.ploc_device;
    bp.reinsertion_batched = b.reinsert_batched;
    bp.reinsertion_whole_iterations = b.reinsert_whole;
    return bp;
}

} // namespace

extern "C" {

// ---- host side: builder ----------------------------------------------------------------------

int trx_bvh_build_tris(const float *verts, uint64_t n, uint32_t max_prims, int threads, trx_bvh **out) {
    if (!out || (n && !verts)) return fail(TRX_ERR_INVALID, "null argument");
    if (max_prims < 1 || max_prims > 3)
        return fail(TRX_ERR_INVALID, "CWBVH only supports a maximum of 3 primitives per leaf."); // src/main.rs:176-178
    if (n >= 0x7fffffffull) return fail(TRX_ERR_INVALID, "too many primitives");
    trx_bvh *b = new (std::nothrow) trx_bvh();
    if (!b) return fail(TRX_ERR_OOM, "host allocation failed");
    const BuildParams bp = to_build_params(build_settings(), max_prims, threads);
    try {
        build_cwbvh_from_tris(verts, n, bp, b->bvh);
    } catch (const std::runtime_error &e) { // the GPU build stage reports its own failures
        delete b;
        return fail(TRX_ERR_NO_DEVICE, "%s", e.what());
    } catch (const std::exception &) {
        delete b;
        return fail(TRX_ERR_OOM, "out of memory building the BVH");
    }
    *out = b;
    return TRX_OK;
}

int trx_bvh_build_aabbs(const float *aabbs, uint64_t n, uint32_t max_prims, int threads, trx_bvh **out) {
    if (!out || (n && !aabbs)) return fail(TRX_ERR_INVALID, "null argument");
    if (max_prims < 1 || max_prims > 3) return fail(TRX_ERR_INVALID, "CWBVH only supports a maximum of 3 primitives per leaf.");
    if (n >= 0x7fffffffull) return fail(TRX_ERR_INVALID, "too many primitives");
    trx_bvh *b = new (std::nothrow) trx_bvh();
    if (!b) return fail(TRX_ERR_OOM, "host allocation failed");
    BuildParams bp = to_build_params(build_settings(), max_prims, threads);
    bp.reinsertion_batch_ratio = 0.f; // boxes of instances: see trx_flat_build
    try {
        build_cwbvh_from_aabbs((const Aabb *)aabbs, n, bp, b->bvh);
    } catch (const std::runtime_error &e) { // the GPU build stage reports its own failures
        delete b;
        return fail(TRX_ERR_NO_DEVICE, "%s", e.what());
    } catch (const std::exception &) {
        delete b;
        return fail(TRX_ERR_OOM, "out of memory building the BVH");
    }
    *out = b;
    return TRX_OK;
}

int trx_set_build_costs(float traversal_cost, float prim_cost) {
    if (!(traversal_cost > 0.f) || !(prim_cost > 0.f)) return fail(TRX_ERR_INVALID, "costs must be positive");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.traversal_cost = traversal_cost;
    g_build.prim_cost = prim_cost;
    return TRX_OK;
}

int trx_set_build_device(int device) {
    if (device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n) return fail(TRX_ERR_NO_DEVICE, "no HIP device %d for the build stage", device);
    }
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.ploc_device = device < 0 ? -1 : device;
    return TRX_OK;
}

int trx_set_build_split(float extra_ratio) {
    if (!(extra_ratio >= 0.f) || extra_ratio > 4.f) return fail(TRX_ERR_INVALID, "split: extra reference ratio in [0, 4]");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.pre_split = extra_ratio;
    return TRX_OK;
}

int trx_set_build_rebraid(float area_fraction) {
    if (!(area_fraction >= 0.f) || area_fraction > 1.f) return fail(TRX_ERR_INVALID, "rebraid: area fraction in [0, 1]");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.rebraid_area = area_fraction;
    return TRX_OK;
}

int trx_set_build_preset(const char *name) {
    // {bins, sweep, reinsertion ratio, iterations, pre-split}: build time against tree quality, like the obvhs
    // presets (which switch pre_split on from slow_build upwards); "" restores the defaults
    struct Preset { const char *name; int bins; uint32_t sweep; float ratio; int iters; float split; };
    static const Preset presets[] = {
        {"fastest_build", 8, 0, 0.0f, 0, 0.0f},   {"very_fast_build", 16, 8, 0.01f, 1, 0.0f}, {"fast_build", 16, 24, 0.02f, 2, 0.0f},
        {"medium_build", 32, 48, 0.02f, 4, 0.0f}, {"slow_build", 32, 64, 0.05f, 6, 0.3f},     {"very_slow_build", 32, 64, 0.15f, 8, 0.3f},
        {"", 32, 48, 0.02f, 4, 0.0f},
    };
    if (!name) return fail(TRX_ERR_INVALID, "preset is null");
    for (const Preset &p : presets) {
        if (std::strcmp(name, p.name) == 0) {
            std::lock_guard<std::mutex> lock(g_build_mu);
            g_build.sah_bins = p.bins;
            g_build.sweep_max = p.sweep;
            g_build.reinsert_ratio = p.ratio;
            g_build.reinsert_iters = p.iters;
            g_build.pre_split = p.split;
            return TRX_OK;
        }
    }
    return fail(TRX_ERR_INVALID, "unknown preset '%s'", name);
}

int trx_set_build_reinsertion(float batch_ratio, int iterations) {
    if (!(batch_ratio >= 0.f) || batch_ratio > 1.f || iterations < 0)
        return fail(TRX_ERR_INVALID, "reinsertion: ratio in [0,1], iterations >= 0");
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.reinsert_ratio = batch_ratio;
    g_build.reinsert_iters = iterations;
    return TRX_OK;
}

int trx_set_build_reinsertion_batches(int whole_iterations) {
    std::lock_guard<std::mutex> lock(g_build_mu);
    g_build.reinsert_whole = whole_iterations != 0;
    return TRX_OK;
}

void trx_bvh_destroy(trx_bvh *b) { delete b; }
uint64_t trx_bvh_node_count(const trx_bvh *b) { return b ? b->bvh.nodes.size() : 0; }
uint64_t trx_bvh_prim_count(const trx_bvh *b) { return b ? b->bvh.primitive_indices.size() : 0; }
const void *trx_bvh_nodes(const trx_bvh *b) { return b ? b->bvh.nodes.data() : nullptr; }
const uint32_t *trx_bvh_primitive_indices(const trx_bvh *b) { return b ? b->bvh.primitive_indices.data() : nullptr; }
void trx_bvh_total_aabb(const trx_bvh *b, float out6[6]) {
    if (!b || !out6) return;
    std::memcpy(out6, b->bvh.total_aabb.mn, 12);
    std::memcpy(out6 + 3, b->bvh.total_aabb.mx, 12);
}
double trx_bvh_build_seconds(const trx_bvh *b) { return b ? b->bvh.build_seconds : 0.0; }

static int flat_build_impl(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                           uint32_t max_prims, int threads, const BuildSettings &settings, trx_flat **out,
                           const uint32_t *instance_object = nullptr, const float *instance_o2w = nullptr,
                           uint32_t n_instances = 0);

// BvhBuildParams of the reference (src/main.rs:571-585) for one build: the BVH2 comes from PLOC with the caller's
// search distance, depth threshold and Morton width, is optimised by the reinsertion pass at the caller's batch ratio
// and collapsed at the caller's traversal cost.  post_collapse_reinsertion_batch_ratio_multiplier is "For BVH2 only"
// in the reference's own words (src/main.rs:119-123): a CWBVH build has no such pass.
int trx_flat_build_params(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                          const trx_build_params *bp, int threads, trx_flat **out) {
    if (!bp) return fail(TRX_ERR_INVALID, "build params are null");
    if (bp->sort_precision != 64 && bp->sort_precision != 128) return fail(TRX_ERR_INVALID, "Unsupported sort precision");
    if (!(bp->reinsertion_batch_ratio >= 0.f) || !(bp->collapse_traversal_cost > 0.f))
        return fail(TRX_ERR_INVALID, "reinsertion_batch_ratio >= 0 and collapse_traversal_cost > 0 required");
    BuildSettings b = build_settings(); // bins / sweep threshold stay those of the current preset
    b.traversal_cost = bp->collapse_traversal_cost;
    // obvhs: 0..1 is the candidate ratio of one pass, above 1 the whole set is evaluated several times
    b.reinsert_ratio = std::min(bp->reinsertion_batch_ratio, 1.0f);
    b.reinsert_iters = bp->reinsertion_batch_ratio > 1.f ? (int)std::ceil(bp->reinsertion_batch_ratio)
                       : bp->reinsertion_batch_ratio > 0.f ? std::max(1, b.reinsert_iters) : 0; // passes: this library's (4)
    b.pre_split = bp->pre_split ? 0.3f : 0.0f;
    if (bp->ploc_search_distance < 1 || bp->ploc_search_distance > 32)
        return fail(TRX_ERR_INVALID, "ploc_search_distance %u outside 1..32", bp->ploc_search_distance);
    b.ploc_distance = bp->ploc_search_distance;
    b.ploc_depth_threshold = bp->search_depth_threshold;
    b.ploc_sort_bits = bp->sort_precision;
    b.reinsert_batched = true; // the parallel reinsertion pass, as in the reference's builder
    return flat_build_impl(verts, object_tri_counts, n_objects, use_tlas, bp->max_prims_per_leaf, threads, b, out);
}

// A preset's build with EVERY stage on the device (review item 8: the presets' own BVH2 stage is a top-down binned-SAH build
// on the host cores, 0.2 s of a 0.96 s medium_build, followed by 0.39 s of one-at-a-time reinsertion searches that are
// sequential by definition).  The device has no top-down builder; it has the whole ploc_cwbvh pipeline - Morton sort +
// PLOC rounds, the reinsertion pass in whole-iteration batches (selection, searches, moves, refit), collapse + encoding -
// so a preset built there is that pipeline with the reference's PLOC parameters (search distance 14, depth threshold 2,
// 64-bit codes, src/main.rs:85-98) and a reinsertion budget per preset name, chosen so that medium_build's tree is walked
// with no more node visits than the host preset's (bistro-class bench view: 17.46 against 17.82) in less than half its
// build time.  device < 0: the same pipeline on the host cores - its twin, byte for byte (tests/test_gpu_builder.py).
int trx_flat_build_preset_device(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                                 const char *preset, uint32_t max_prims, int threads, int device, trx_flat **out) {
    struct Budget { const char *name; float ratio; int iters; float split; };
    static const Budget budgets[] = {{"fastest_build", 0.0f, 0, 0.0f}, {"very_fast_build", 0.05f, 2, 0.0f}, {"fast_build", 0.10f, 4, 0.0f},
                                     {"medium_build", 0.15f, 8, 0.0f}, {"slow_build", 0.15f, 12, 0.3f},     {"very_slow_build", 0.30f, 16, 0.3f}};
    if (!preset) return fail(TRX_ERR_INVALID, "preset is null");
    const Budget *bu = nullptr;
    for (const Budget &b : budgets)
        if (std::strcmp(preset, b.name) == 0) bu = &b;
    if (!bu) return fail(TRX_ERR_INVALID, "unknown preset '%s'", preset);
    if (device >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n) return fail(TRX_ERR_NO_DEVICE, "no HIP device %d for the build", device);
    }
    BuildSettings b = build_settings(); // costs and re-braiding stay the caller's
    b.ploc_distance = 14;
    b.ploc_depth_threshold = 2;
    b.ploc_sort_bits = 64;
    b.ploc_device = device < 0 ? -1 : device;
    b.reinsert_ratio = bu->ratio;
    b.reinsert_iters = bu->iters;
    b.reinsert_whole = true;
    b.reinsert_batched = true;
    b.pre_split = bu->split;
    return flat_build_impl(verts, object_tri_counts, n_objects, use_tlas, max_prims, threads, b, out);
}

void trx_build_params_default(trx_build_params *bp) {
    if (!bp) return;
    // the defaults of the reference's command line (src/main.rs:85-124,158-163)
    bp->pre_split = 0;
    bp->ploc_search_distance = 14;
    bp->search_depth_threshold = 2;
    bp->reinsertion_batch_ratio = 0.15f;
    bp->sort_precision = 64;
    bp->max_prims_per_leaf = 3;
    bp->post_collapse_reinsertion_batch_ratio_multiplier = 0.0f;
    bp->collapse_traversal_cost = 1.0f;
}

// cwbvh_gpu_runner, src/rt_gpu/mod.rs:16-112
int trx_flat_build(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                   uint32_t max_prims, int threads, trx_flat **out) {
    return flat_build_impl(verts, object_tri_counts, n_objects, use_tlas, max_prims, threads, build_settings(), out);
}

// One BLAS per object and a TLAS over INSTANCES of them: instance k places object instance_object[k] with the
// affine object-to-world matrix instance_object_to_world + 16 k (column-major; NULL = identity for all).  The TLAS
// boxes bound the transformed BLAS boxes; trx_flat.instance_transforms / instance_source come back in
// TLAS-primitive order, ready for trx_scene_create + trx_scene_set_instance_transforms.
int trx_flat_build_instanced(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects,
                             const uint32_t *instance_object, const float *instance_object_to_world, uint32_t n_instances,
                             uint32_t max_prims, int threads, trx_flat **out) {
    if (!instance_object || n_instances == 0) return fail(TRX_ERR_INVALID, "no instances");
    for (uint32_t k = 0; k < n_instances; k++) {
        if (instance_object[k] >= n_objects) return fail(TRX_ERR_INVALID, "instance %u names object %u of %u", k, instance_object[k], n_objects);
        if (object_tri_counts && object_tri_counts[instance_object[k]] == 0) return fail(TRX_ERR_INVALID, "instance %u names an empty object", k);
    }
    return flat_build_impl(verts, object_tri_counts, n_objects, 1, max_prims, threads, build_settings(), out, instance_object,
                           instance_object_to_world, n_instances);
}

static int flat_build_impl(const float *verts, const uint64_t *object_tri_counts, uint32_t n_objects, int use_tlas,
                           uint32_t max_prims, int threads, const BuildSettings &settings, trx_flat **out,
                           const uint32_t *instance_object, const float *instance_o2w, uint32_t n_instances) {
    if (!out || !object_tri_counts || n_objects == 0) return fail(TRX_ERR_INVALID, "null argument");
    if (max_prims < 1 || max_prims > 3) return fail(TRX_ERR_INVALID, "CWBVH only supports a maximum of 3 primitives per leaf.");
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_objects; i++) total += object_tri_counts[i];
    if (total && !verts) return fail(TRX_ERR_INVALID, "verts is null");
    if (total >= 0x7fffffffull) return fail(TRX_ERR_INVALID, "too many triangles");
    const BuildParams bp = to_build_params(settings, max_prims, threads);
    try {
        // without --tlas everything is flattened into the first object (src/main.rs:300-308)
        std::vector<uint64_t> counts;
        std::vector<uint32_t> blas_of_object(n_objects, 0xFFFFFFFFu); // objects without triangles have no BLAS
        if (use_tlas) {
            for (uint32_t i = 0; i < n_objects; i++)
                if (object_tri_counts[i]) {
                    blas_of_object[i] = (uint32_t)counts.size();
                    counts.push_back(object_tri_counts[i]);
                }
            if (counts.empty()) counts.push_back(0);
        } else {
            counts.push_back(total);
        }
        std::vector<CwbvhNode> nodes;
        std::vector<uint32_t> blas_offset, blas_tri_start;
        std::vector<Aabb> blas_aabb;
        // The three arrays with one entry per triangle reference go straight into the buffers the caller receives (malloc:
        // no value-initialisation, no copy at the end - 250 MB each way on a 3.9 M triangle scene, a tenth of a second of
        // one core); their size is known once the BLASes are built.
        struct Grow { // entries filled so far / capacity, in triangle references
            float *tri = nullptr, *box = nullptr;
            uint32_t *src = nullptr;
            size_t n = 0, cap = 0;
            ~Grow() {
                std::free(tri);
                std::free(box);
                std::free(src);
            }
        } refs;
        double blas_s = 0.0, tlas_s = 0.0;
        // BLAS builds: large objects one after the other with every thread, the (many) small ones of a
        // TLAS scene concurrently with one thread each; assembly below stays in object order
        std::vector<CwBvh> built(counts.size());
        std::vector<uint64_t> firsts(counts.size());
        {
            uint64_t f0 = 0;
            for (size_t i = 0; i < counts.size(); i++) { firsts[i] = f0; f0 += counts[i]; }
            const auto t0 = std::chrono::steady_clock::now();
            int nthreads = threads > 0 ? threads : usable_threads();
            if (nthreads < 1) nthreads = 1;
            // objects the GPU stage would take (trx_set_build_device: >= kDevicePlocMinPrims primitives) are never
            // handed to the one-thread host pool
            const uint64_t kSmall = bp.ploc_device >= 0 ? std::min<uint64_t>(65536, kDevicePlocMinPrims - 1) : 65536;
            std::vector<size_t> small;
            for (size_t i = 0; i < counts.size(); i++) {
                if (counts[i] > kSmall || counts.size() == 1 || nthreads == 1)
                    build_cwbvh_from_tris(verts + firsts[i] * 9, counts[i], bp, built[i]);
                else
                    small.push_back(i);
            }
            if (!small.empty()) {
                BuildParams one = bp;
                one.threads = 1;
                one.ploc_device = -1; // the many small BLASes of a TLAS scene stay on the host cores
                std::atomic<size_t> next{0};
                std::mutex err_mu;
                std::exception_ptr first_error; // rethrown as it was (a builder failure is not an allocation failure)
                auto worker = [&]() {
                    try {
                        for (size_t k = next.fetch_add(1); k < small.size(); k = next.fetch_add(1)) {
                            const size_t i = small[k];
                            build_cwbvh_from_tris(verts + firsts[i] * 9, counts[i], one, built[i]);
                        }
                    } catch (...) {
                        std::lock_guard<std::mutex> g(err_mu);
                        if (!first_error) first_error = std::current_exception();
                        next.store(small.size()); // the other workers stop taking objects
                    }
                };
                std::vector<std::thread> pool;
                const int n = (int)std::min<size_t>((size_t)nthreads, small.size());
                for (int t = 0; t < n; t++) pool.emplace_back(worker);
                for (auto &th : pool) th.join();
                if (first_error) std::rethrow_exception(first_error);
            }
            blas_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        {
            size_t all = 0;
            for (const CwBvh &b : built) all += b.primitive_indices.size();
            refs.cap = std::max<size_t>(all, 1);
            refs.tri = (float *)std::malloc(refs.cap * 36);
            refs.box = (float *)std::malloc(refs.cap * 24);
            refs.src = (uint32_t *)std::malloc(refs.cap * 4);
            if (!refs.tri || !refs.box || !refs.src) throw std::bad_alloc();
        }
        uint64_t first = 0;
        for (size_t bi = 0; bi < counts.size(); bi++) {
            const uint64_t cnt = counts[bi];
            CwBvh &bvh = built[bi];
            const uint32_t tri_offset = (uint32_t)refs.n;
            blas_tri_start.push_back(tri_offset);
            // permute triangles into primitive_indices order (mod.rs:38-43); the entries are independent, so a large
            // BLAS is filled by every core
            {
                const size_t np = bvh.primitive_indices.size();
                const size_t r0 = refs.n;
                refs.n += np;
                auto fill = [&](size_t k0, size_t k1) {
                    for (size_t k = k0; k < k1; k++) {
                        const uint32_t pi = bvh.primitive_indices[k];
                        const float *v = verts + (first + pi) * 9;
                        std::memcpy(refs.tri + (r0 + k) * 9, v, 36);
                        refs.src[r0 + k] = (uint32_t)(first + pi);
                        // the box this entry was built with: the triangle's own, or its clipped part after pre-splitting
                        float *bx = refs.box + (r0 + k) * 6;
                        if (!bvh.primitive_boxes.empty()) {
                            for (int a = 0; a < 3; a++) { bx[a] = bvh.primitive_boxes[k].mn[a]; bx[3 + a] = bvh.primitive_boxes[k].mx[a]; }
                        } else {
                            for (int a = 0; a < 3; a++) {
                                bx[a] = std::min(v[a], std::min(v[3 + a], v[6 + a]));
                                bx[3 + a] = std::max(v[a], std::max(v[3 + a], v[6 + a]));
                            }
                        }
                    }
                };
                const int nt = (int)std::min<size_t>((size_t)std::max(1, threads > 0 ? threads : usable_threads()), np / 65536 + 1);
                if (nt <= 1) {
                    fill(0, np);
                } else {
                    std::vector<std::thread> pool;
                    for (int t = 0; t < nt; t++) pool.emplace_back(fill, np * t / nt, np * (t + 1) / nt);
                    for (auto &th : pool) th.join();
                }
            }
            // global triangle buffer: offset primitive_base_idx (mod.rs:44-48)
            for (CwbvhNode &n : bvh.nodes) n.primitive_base_idx += tri_offset;
            blas_offset.push_back((uint32_t)nodes.size());
            blas_aabb.push_back(bvh.total_aabb);
            nodes.insert(nodes.end(), bvh.nodes.begin(), bvh.nodes.end());
            first += cnt;
        }
        blas_tri_start.push_back((uint32_t)refs.n);
        std::vector<uint32_t> inst, inst_source, inst_entry;
        std::vector<float> inst_xf;
        uint32_t tlas_start = 0;
        if (use_tlas) {
            // what the TLAS is built over: one box per BLAS (the reference, src/cwbvh.rs:114), or one box per
            // instance = the BLAS box carried to world space by the instance's transform, padded by a few ulps of
            // its magnitude (the ray is taken to object space by the rounded INVERSE, which does not commute
            // exactly with transforming the box forward)
            std::vector<Aabb> tlas_boxes = blas_aabb;
            // Re-braiding (own builder only; the reference builds its TLAS over whole BLAS boxes, src/cwbvh.rs:114): a
            // BLAS whose box is large against the scene is referenced through the subtrees under its root instead -
            // repeatedly, largest box first, as long as the node has inner children only (a leaf child's triangles
            // could not be reached through any subtree) - so a floor or a shell that spans the scene stops making
            // every ray walk it from the root.  Each such TLAS primitive carries the node its walk starts at.
            std::vector<uint32_t> prim_blas, prim_entry;
            if (!instance_object && settings.rebraid_area > 0.f && counts.size() > 1) {
                auto area = [](const Aabb &b) {
                    const double dx = std::max(0.0, (double)b.mx[0] - b.mn[0]), dy = std::max(0.0, (double)b.mx[1] - b.mn[1]),
                                 dz = std::max(0.0, (double)b.mx[2] - b.mn[2]);
                    return 2.0 * (dx * dy + dy * dz + dz * dx);
                };
                Aabb scene_box = blas_aabb[0];
                for (const Aabb &b : blas_aabb)
                    for (int a = 0; a < 3; a++) { scene_box.mn[a] = std::min(scene_box.mn[a], b.mn[a]); scene_box.mx[a] = std::max(scene_box.mx[a], b.mx[a]); }
                const double limit = (double)settings.rebraid_area * area(scene_box);
                struct Item { double a; uint32_t blas, entry; Aabb box; };
                auto less = [](const Item &x, const Item &y) { return x.a < y.a || (x.a == y.a && (x.blas > y.blas || (x.blas == y.blas && x.entry > y.entry))); };
                std::priority_queue<Item, std::vector<Item>, decltype(less)> heap(less);
                std::vector<Item> final_items;
                for (uint32_t b = 0; b < (uint32_t)blas_aabb.size(); b++) heap.push(Item{area(blas_aabb[b]), b, 0u, blas_aabb[b]});
                // (measured on the san-miguel-class scene, three views, profiles/r03_rebraid_tlas_variants.log: 16 K / 64 K /
                // 160 K / 225 K primitives = 3.44 / 2.99 / 3.15 / 4.51 ms against 4.73 ms unopened; beyond ~160 K the
                // TLAS - a plain binned-SAH tree without the BLAS builder's reinsertion pass - becomes the worse upper tree)
                const size_t max_prims_tlas = blas_aabb.size() + 262144;
                // // (instance ids stay far below 2^24 triangle-group indices)
                while (!heap.empty()) {
                    Item it = heap.top();
                    heap.pop();
                    // (the BLAS nodes were moved into `nodes`; BLAS b starts at blas_offset[b])
                    const CwbvhNode &n = nodes[(size_t)blas_offset[it.blas] + it.entry];
                    bool openable = it.a > limit && n.imask != 0 && heap.size() + final_items.size() + 8 <= max_prims_tlas;
                    for (int sl = 0; sl < 8 && openable; sl++)
                        if (n.child_meta[sl] != 0 && (n.child_meta[sl] & 0x18) != 0x18) openable = false; // a leaf child
                    if (!openable) {
                        final_items.push_back(it);
                        continue;
                    }
                    uint32_t rank = 0;
                    for (int sl = 0; sl < 8; sl++) {
                        if (!((n.imask >> sl) & 1u)) continue;
                        // the child's quantised box, decoded exactly (24-bit origin + 8-bit step count x a power of two
                        // fits a double) and rounded outwards to f32, clipped to the box it was opened from
                        Aabb cb;
                        const uint8_t *qlo[3] = {n.child_min_x, n.child_min_y, n.child_min_z}, *qhi[3] = {n.child_max_x, n.child_max_y, n.child_max_z};
                        for (int a = 0; a < 3; a++) {
                            const double ex = std::ldexp(1.0, (int)n.e[a] - 127);
                            const double lo = (double)n.p[a] + qlo[a][sl] * ex, hi = (double)n.p[a] + qhi[a][sl] * ex;
                            float flo = (float)lo, fhi = (float)hi;
                            if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
                            if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
                            cb.mn[a] = std::max(flo, it.box.mn[a]);
                            cb.mx[a] = std::min(fhi, it.box.mx[a]);
                        }
                        heap.push(Item{area(cb), it.blas, n.child_base_idx + rank, cb});
                        rank++;
                    }
                }
                if (final_items.size() > blas_aabb.size()) {
                    // deterministic order: by BLAS, then entry node
                    std::sort(final_items.begin(), final_items.end(), [](const Item &x, const Item &y) { return x.blas < y.blas || (x.blas == y.blas && x.entry < y.entry); });
                    tlas_boxes.clear();
                    for (const Item &it : final_items) {
                        tlas_boxes.push_back(it.box);
                        prim_blas.push_back(it.blas);
                        prim_entry.push_back(it.entry);
                    }
                }
            }
            if (instance_object) {
                tlas_boxes.assign(n_instances, Aabb{});
                for (uint32_t k = 0; k < n_instances; k++) {
                    const Aabb &bb = blas_aabb[blas_of_object[instance_object[k]]];
                    Aabb wb;
                    for (int a = 0; a < 3; a++) { wb.mn[a] = 3.402823466e+38f; wb.mx[a] = -3.402823466e+38f; }
                    for (int c = 0; c < 8; c++) {
                        const float p[3] = {c & 1 ? bb.mx[0] : bb.mn[0], c & 2 ? bb.mx[1] : bb.mn[1], c & 4 ? bb.mx[2] : bb.mn[2]};
                        float q[3] = {p[0], p[1], p[2]};
                        if (instance_o2w) {
                            const float *m = instance_o2w + (size_t)k * 16;
                            for (int r = 0; r < 3; r++) q[r] = m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r];
                        }
                        for (int a = 0; a < 3; a++) { wb.mn[a] = std::min(wb.mn[a], q[a]); wb.mx[a] = std::max(wb.mx[a], q[a]); }
                    }
                    for (int a = 0; a < 3; a++) {
                        const float pad = 1e-5f * (std::max(std::fabs(wb.mn[a]), std::fabs(wb.mx[a])) + (wb.mx[a] - wb.mn[a])) + 1e-30f;
                        wb.mn[a] -= pad;
                        wb.mx[a] += pad;
                    }
                    tlas_boxes[k] = wb;
                }
            }
            // TLAS over the BLAS boxes (src/cwbvh.rs:114,132); instance table in TLAS
            // primitive order (mod.rs:72-78); TLAS nodes appended last (mod.rs:88-99)
            CwBvh tlas;
            // no reinsertion pass over instance boxes: the SAH's constant leaf cost misprices an instance
            // (a whole BLAS traversal), and the pass measured worse there (san-miguel-class stand-in:
            // 62.9 -> 67.4 node visits per ray with it, 62.7 with the pass in the BLASes only)
            BuildParams bpt = bp;
            bpt.reinsertion_batch_ratio = 0.f;
            // ... and an instance is dearer than a node visit: cost 3 instead of 0.3 keeps one instance per leaf
            // slot, each with its own quantised box (same scene: 62.7 -> 60.6 node visits per ray)
            bpt.prim_cost = std::max(bpt.prim_cost, 3.0f);
            build_cwbvh_from_aabbs(tlas_boxes.data(), tlas_boxes.size(), bpt, tlas);
            tlas_s = tlas.build_seconds;
            for (uint32_t pi : tlas.primitive_indices) {
                if (!prim_blas.empty()) { // re-braided: TLAS primitive pi is the subtree at node prim_entry[pi] of BLAS prim_blas[pi]
                    inst.push_back(blas_offset[prim_blas[pi]]);
                    inst_source.push_back(prim_blas[pi]);
                    inst_entry.push_back(prim_entry[pi]);
                    continue;
                }
                inst.push_back(blas_offset[instance_object ? blas_of_object[instance_object[pi]] : pi]);
                inst_source.push_back(pi);
                if (instance_o2w) inst_xf.insert(inst_xf.end(), instance_o2w + (size_t)pi * 16, instance_o2w + (size_t)pi * 16 + 16);
            }
            tlas_start = (uint32_t)nodes.size();
            nodes.insert(nodes.end(), tlas.nodes.begin(), tlas.nodes.end());
        }
        trx_flat *f = (trx_flat *)std::calloc(1, sizeof(trx_flat));
        if (!f) return fail(TRX_ERR_OOM, "host allocation failed");
        auto dup = [](const void *src, size_t bytes) -> void * {
            void *p = std::malloc(bytes ? bytes : 1);
            if (p && bytes) std::memcpy(p, src, bytes);
            return p;
        };
        f->n_nodes = nodes.size();
        f->bvh_bytes = dup(nodes.data(), nodes.size() * sizeof(CwbvhNode));
        f->n_tris = refs.n;
        f->tri_verts = refs.tri; // (handed over: see `refs`)
        refs.tri = nullptr;
        f->n_instances = (uint32_t)inst.size();
        f->instance_offsets = (uint32_t *)dup(inst.data(), inst.size() * 4);
        f->tlas_start = tlas_start;
        f->tri_source = refs.src;
        f->tri_boxes = refs.box;
        refs.src = nullptr;
        refs.box = nullptr;
        f->n_blas = (uint32_t)counts.size();
        f->blas_tri_start = (uint32_t *)dup(blas_tri_start.data(), blas_tri_start.size() * 4);
        f->blas_build_s = blas_s;
        f->tlas_build_s = tlas_s;
        f->instance_source = (uint32_t *)dup(inst_source.data(), inst_source.size() * 4);
        f->instance_transforms = inst_xf.empty() ? nullptr : (float *)dup(inst_xf.data(), inst_xf.size() * 4);
        f->instance_entry_nodes = inst_entry.empty() ? nullptr : (uint32_t *)dup(inst_entry.data(), inst_entry.size() * 4);
        if (!f->instance_source || (!inst_xf.empty() && !f->instance_transforms) || (!inst_entry.empty() && !f->instance_entry_nodes)) {
            trx_flat_destroy(f);
            return fail(TRX_ERR_OOM, "host allocation failed");
        }
        if (!f->bvh_bytes || !f->tri_verts || !f->instance_offsets || !f->tri_source || !f->blas_tri_start || !f->tri_boxes) {
            trx_flat_destroy(f);
            return fail(TRX_ERR_OOM, "host allocation failed");
        }
        *out = f;
    } catch (const std::runtime_error &e) { // the GPU build stage reports its own failures
        return fail(TRX_ERR_NO_DEVICE, "%s", e.what());
    } catch (const std::exception &) {
        return fail(TRX_ERR_OOM, "out of memory building the scene");
    }
    return TRX_OK;
}

void trx_flat_destroy(trx_flat *f) {
    if (!f) return;
    std::free(f->bvh_bytes);
    std::free(f->tri_verts);
    std::free(f->instance_offsets);
    std::free(f->tri_source);
    std::free(f->blas_tri_start);
    std::free(f->tri_boxes);
    std::free(f->instance_source);
    std::free(f->instance_transforms);
    std::free(f->instance_entry_nodes);
    std::free(f);
}

// ---- host side: scenes ---------------------------------------------------------------------------

static int export_mesh(std::vector<float> &verts, std::vector<uint64_t> &objects, float **out_verts,
                       uint64_t *out_n, uint64_t **out_counts, uint32_t *out_nobj) {
    uint64_t n = verts.size() / 9;
    float *v = (float *)std::malloc(std::max<size_t>(verts.size() * 4, 4));
    uint64_t *c = (uint64_t *)std::malloc(std::max<size_t>(objects.size() * 8, 8));
    if (!v || !c) {
        std::free(v);
        std::free(c);
        return fail(TRX_ERR_OOM, "host allocation failed");
    }
    if (!verts.empty()) std::memcpy(v, verts.data(), verts.size() * 4);
    if (!objects.empty()) std::memcpy(c, objects.data(), objects.size() * 8);
    *out_verts = v;
    *out_n = n;
    if (out_counts) *out_counts = c;
    else std::free(c);
    if (out_nobj) *out_nobj = (uint32_t)objects.size();
    return TRX_OK;
}

int trx_gen_scene(const char *name, uint64_t n_tris, uint64_t seed, float **out_verts, uint64_t *out_n,
                  uint64_t **out_counts, uint32_t *out_nobj) {
    if (!name || !out_verts || !out_n) return fail(TRX_ERR_INVALID, "null argument");
    std::vector<float> verts;
    std::vector<uint64_t> objects;
    try {
        if (!gen_scene(name, n_tris, seed, verts, objects)) return fail(TRX_ERR_INVALID, "unknown scene '%s'", name);
    } catch (const std::exception &) {
        return fail(TRX_ERR_OOM, "out of memory generating '%s'", name);
    }
    return export_mesh(verts, objects, out_verts, out_n, out_counts, out_nobj);
}

int trx_scene_camera(const char *name, float eye[3], float look_at[3], float *fov) {
    if (!name || !eye || !look_at || !fov) return fail(TRX_ERR_INVALID, "null argument");
    if (!scene_camera(name, eye, look_at, fov)) return fail(TRX_ERR_INVALID, "unknown scene '%s'", name);
    return TRX_OK;
}

int trx_load_model(const char *path, float **out_verts, uint64_t *out_n, uint64_t **out_counts, uint32_t *out_nobj) {
    if (!path || !out_verts || !out_n) return fail(TRX_ERR_INVALID, "null argument");
    std::vector<float> verts;
    std::vector<uint64_t> objects;
    try {
        if (!load_model(path, verts, objects)) return fail(TRX_ERR_IO, "Error while loading model file \"%s\"", path);
    } catch (const std::exception &) {
        return fail(TRX_ERR_OOM, "out of memory loading '%s'", path);
    }
    return export_mesh(verts, objects, out_verts, out_n, out_counts, out_nobj);
}

int trx_load_scene(const char *path, float **out_verts, uint64_t *out_n, uint64_t **out_counts, uint32_t *out_nobj, float eye[3],
                   float look_at[3], float *fov_deg) {
    if (!path || !out_verts || !out_n || !eye || !look_at || !fov_deg) return fail(TRX_ERR_INVALID, "null argument");
    std::string model;
    if (!parse_scene_ron(path, model, eye, look_at, fov_deg)) return fail(TRX_ERR_IO, "Failed to load config: %s", path);
    {   // (beyond the reference's rule: a model path that does not resolve from the working directory is tried next to the
        // scene file's great-grandparent directory, so that an ABSOLUTE scene path works from anywhere)
        FILE *probe = std::fopen(model.c_str(), "rb");
        if (probe) {
            std::fclose(probe);
        } else if (!model.empty() && model[0] != '/') {
            std::string base = path;
            for (int up = 0; up < 3; up++) {
                const size_t sl = base.find_last_of('/');
                base = sl == std::string::npos ? std::string() : base.substr(0, sl);
            }
            if (!base.empty()) model = base + "/" + model;
        }
    }
    return trx_load_model(model.c_str(), out_verts, out_n, out_counts, out_nobj);
}

void trx_free(void *p) { std::free(p); }

} // extern "C"

