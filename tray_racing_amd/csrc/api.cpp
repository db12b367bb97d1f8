// api.cpp - the C-ABI declared in include/trx.h: errors, devices, scene upload (replaces the buffer creation of
// src/rt_gpu/rt_gpu_software.rs:83-160) and its setters, the camera (src/main.rs:602-616).  The other entry points
// live in api_launch.cpp / api_trace.cpp / api_traverse.cpp / api_build.cpp (api_internal.h says which is where).
// There is deliberately no CPU traversal in this library.
#include "api_internal.h"

namespace {
thread_local std::string g_err;
} // namespace

namespace trxapi {
std::atomic<uint32_t> g_variant{0};
std::string &err_string() { return g_err; }
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
} // namespace trxapi

namespace trx {
// error reporting for the other translation units of the library (comm.cpp)
int fail_msg(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
} // namespace trx

namespace {

// Structural validation of untrusted node buffers: a kernel that walks a
// malformed tree would read out of bounds (and can take the GPU down), so
// every index the kernel can form is range-checked here once.
int validate_nodes(const CwbvhNode *nodes, uint64_t n_nodes, uint64_t n_tris, const uint32_t *inst,
                   uint32_t n_inst, uint32_t tlas_start) {
    std::vector<uint32_t> seg; // BLAS segment starts, sorted
    const bool tlas = n_inst > 0;
    if (tlas) {
        if (tlas_start >= n_nodes) return fail(TRX_ERR_FORMAT, "tlas_start %u >= n_nodes %llu", tlas_start, (unsigned long long)n_nodes);
        seg.assign(inst, inst + n_inst);
        for (uint32_t o : seg)
            if (o >= tlas_start) return fail(TRX_ERR_FORMAT, "instance offset %u not below tlas_start %u", o, tlas_start);
        std::sort(seg.begin(), seg.end());
        seg.erase(std::unique(seg.begin(), seg.end()), seg.end());
    }
    for (uint64_t i = 0; i < n_nodes; i++) {
        const CwbvhNode &n = nodes[i];
        uint64_t seg_begin = 0, seg_end = n_nodes;
        bool in_tlas = false;
        if (tlas) {
            if (i >= tlas_start) {
                seg_begin = tlas_start;
                in_tlas = true;
            } else {
                auto it = std::upper_bound(seg.begin(), seg.end(), (uint32_t)i);
                seg_begin = it == seg.begin() ? 0 : *(it - 1);
                seg_end = it == seg.end() ? tlas_start : *it;
            }
        }
        uint32_t inner = 0, prims = 0;
        for (int s = 0; s < 8; s++) {
            uint8_t m = n.child_meta[s];
            if (m == 0) continue;
            if ((m & 0x18) == 0x18) {
                inner++;
            } else {
                uint32_t bits = m >> 5, cnt = bits == 1 ? 1 : bits == 3 ? 2 : bits == 7 ? 3 : 0;
                if (!cnt) return fail(TRX_ERR_FORMAT, "node %llu slot %d: bad leaf meta 0x%02x", (unsigned long long)i, s, m);
                prims = std::max(prims, (uint32_t)(m & 0x1f) + cnt);
            }
        }
        if (inner != (uint32_t)__builtin_popcount(n.imask))
            return fail(TRX_ERR_FORMAT, "node %llu: imask 0x%02x disagrees with child_meta", (unsigned long long)i, n.imask);
        if (inner && (uint64_t)n.child_base_idx + inner > seg_end - seg_begin)
            return fail(TRX_ERR_FORMAT, "node %llu: children [%u,+%u) leave the BVH segment", (unsigned long long)i, n.child_base_idx, inner);
        if (prims > 24) return fail(TRX_ERR_FORMAT, "node %llu: more than 24 primitives", (unsigned long long)i);
        uint64_t limit = in_tlas ? n_inst : n_tris;
        if (prims && (uint64_t)n.primitive_base_idx + prims > limit)
            return fail(TRX_ERR_FORMAT, "node %llu: primitives [%u,+%u) out of range %llu", (unsigned long long)i,
                        n.primitive_base_idx, prims, (unsigned long long)limit);
    }
    return TRX_OK;
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            float f = (float)man * 5.9604644775390625e-8f;
            return sign ? -f : f;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// Any accepted input format -> 48-byte device record {v0, e1 = v0 - v1, e2 = v2 - v0, ng = e1 x e2 in the w lanes}.
void convert_tris(const void *src, uint64_t n, uint32_t fmt, TriDev *dst) {
    const uint8_t *b = (const uint8_t *)src;
    for (uint64_t i = 0; i < n; i++) {
        TriDev t;
        std::memset(&t, 0, sizeof(t));
        if (fmt == TRX_TRI_F16_24) {
            float v[3];
            uint32_t e[3];
            std::memcpy(v, b + 24 * i, 12);
            std::memcpy(e, b + 24 * i + 12, 12);
            for (int k = 0; k < 3; k++) {
                t.v0[k] = v[k];
                t.e1[k] = -half_to_float((uint16_t)(e[k] >> 16)); // stored e1 = v1 - v0
                t.e2[k] = half_to_float((uint16_t)(e[k] & 0xffff));
            }
        } else {
            float v[9];
            std::memcpy(v, b + 36 * i, 36);
            for (int k = 0; k < 3; k++) {
                t.v0[k] = v[k];
                if (fmt == TRX_TRI_VERTS_36) {
                    t.e1[k] = v[k] - v[3 + k];
                    t.e2[k] = v[6 + k] - v[k];
                } else {
                    t.e1[k] = v[3 + k];
                    t.e2[k] = v[6 + k];
                }
            }
        }
        t.ngx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
        t.ngy = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
        t.ngz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
        dst[i] = t;
    }
}

} // namespace

extern "C" {

const char *trx_last_error(void) { return g_err.c_str(); }
uint32_t trx_abi_version(void) { return TRX_ABI_VERSION; }

int trx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int trx_device_name(int device, char *buf, size_t buf_len) {
    if (!buf || !buf_len) return fail(TRX_ERR_INVALID, "null buffer");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buf_len, "%s", prop.gcnArchName);
    return TRX_OK;
}

uint32_t trx_tri_format_bytes(uint32_t f) {
    switch (f) {
    case TRX_TRI_F16_24: return 24;
    case TRX_TRI_VERTS_36: return 36;
    case TRX_TRI_EDGES_36: return 36;
    default: return 0;
    }
}

uint32_t trx_shard_tiles(uint32_t w, uint32_t h, trx_shard shard) {
    if (shard.count == 0) shard.count = 1;
    const uint64_t tiles = (uint64_t)((w + 7) / 8) * ((h + 7) / 8);
    if (shard.index >= shard.count || tiles <= shard.index) return 0;
    return (uint32_t)((tiles - shard.index + shard.count - 1) / shard.count);
}

uint32_t trx_set_kernel_variant(uint32_t variant) {
    return g_variant.exchange(variant);
}

int trx_scene_create(const void *bvh_bytes, uint64_t n_nodes, const void *tri_bytes, uint64_t n_tris,
                     uint32_t tri_format, const uint32_t *instance_offsets, uint32_t n_instances,
                     uint32_t tlas_start, int device, trx_scene **out) {
    if (!out) return fail(TRX_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!bvh_bytes || n_nodes == 0) return fail(TRX_ERR_INVALID, "empty node buffer");
    if (n_tris && !tri_bytes) return fail(TRX_ERR_INVALID, "tri_bytes is null");
    if (!trx_tri_format_bytes(tri_format)) return fail(TRX_ERR_INVALID, "unknown tri_format %u", tri_format);
    if (n_nodes >= 0xffffffffull || n_tris >= 0xffffffffull) return fail(TRX_ERR_INVALID, "buffer too large for u32 indices");
    if (n_instances && !instance_offsets) return fail(TRX_ERR_INVALID, "instance_offsets is null");
    if (!n_instances && tlas_start != 0) return fail(TRX_ERR_INVALID, "tlas_start without instances");
    int rc = validate_nodes((const CwbvhNode *)bvh_bytes, n_nodes, n_tris, instance_offsets, n_instances, tlas_start);
    if (rc) return rc;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(TRX_ERR_NO_DEVICE, "no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TRX_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));

    trx_scene *s = new (std::nothrow) trx_scene();
    if (!s) return fail(TRX_ERR_OOM, "host allocation failed");
    s->device = device;
    s->n_nodes = n_nodes;
    s->n_tris = n_tris;
    s->n_inst = n_instances;
    s->tlas_start = tlas_start;
    s->tlas = n_instances > 0;

    std::vector<TriDev> tris(std::max<uint64_t>(n_tris, 1));
    convert_tris(tri_bytes, n_tris, tri_format, tris.data());

    auto cleanup = [&](int code) {
        trx_scene_destroy(s);
        return code;
    };
#define HIP_TRY_S(expr)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return cleanup(fail(e_ == hipErrorOutOfMemory ? TRX_ERR_OOM : TRX_ERR_NO_DEVICE, "%s failed: %s", \
                                #expr, hipGetErrorString(e_)));                                              \
    } while (0)
    HIP_TRY_S(hipMalloc(&s->d_nodes, n_nodes * TRX_NODE_BYTES));
    HIP_TRY_S(hipMemcpy(s->d_nodes, bvh_bytes, n_nodes * TRX_NODE_BYTES, hipMemcpyHostToDevice));
    HIP_TRY_S(hipMalloc(&s->d_tris, tris.size() * sizeof(TriDev)));
    HIP_TRY_S(hipMemcpy(s->d_tris, tris.data(), tris.size() * sizeof(TriDev), hipMemcpyHostToDevice));
    uint32_t zero = 0;
    if (n_instances) s->h_inst.assign(instance_offsets, instance_offsets + n_instances);
    HIP_TRY_S(hipMalloc(&s->d_inst, std::max<uint32_t>(n_instances, 4) * sizeof(uint32_t)));
    HIP_TRY_S(hipMemcpy(s->d_inst, n_instances ? instance_offsets : &zero, (n_instances ? n_instances : 1) * sizeof(uint32_t),
                        hipMemcpyHostToDevice));
    HIP_TRY_S(hipEventCreate(&s->ev0));
    HIP_TRY_S(hipEventCreate(&s->ev1));
#undef HIP_TRY_S
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return cleanup(fail(TRX_ERR_NO_DEVICE, "hipGetDeviceProperties failed"));
        s->cu_count = prop.multiProcessorCount;
    }
    s->grid = trace_grid_size(device, 0, s->tlas, 0, false);
    {   // extent of the root node's quantisation frame (255 steps of 2^(e - 127) per axis): the scene's scale
        const CwbvhNode &root = ((const CwbvhNode *)bvh_bytes)[s->tlas ? tlas_start : 0];
        double d2 = 0.0;
        for (int k = 0; k < 3; k++) {
            const double ext = 255.0 * std::ldexp(1.0, (int)root.e[k] - 127);
            d2 += ext * ext;
        }
        s->scene_diag = (float)std::sqrt(d2);
    }
    {   // may the literal-division node test multiply e = 2^(byte - 127) by the ray's 1/d instead of dividing by d?  Exact
        // while the product stays normal: exponent bytes 0 (e = 0) or >= 21 with |1/d| >= 2^-20 (kernels.hip, pow2)
        bool ok = true;
        const CwbvhNode *nodes = (const CwbvhNode *)bvh_bytes;
        for (uint64_t i = 0; i < n_nodes && ok; i++)
            for (int k = 0; k < 3; k++) ok = ok && (nodes[i].e[k] == 0 || nodes[i].e[k] >= 21);
        // ... and may it compute (p - o) / d from the ray's 1/d by one correction step (kernels.hip, div_by_rcp)?  Every
        // component of every node's p is +0 or 2^-36 <= |p| <= 2^59 (no -0, no NaN, nothing tiny or enormous): with the
        // ray's own flag (origin components 0 or in the same range) p - o is then +0 or 2^-59 <= |p - o| <= 2^60
        bool org = ok;
        for (uint64_t i = 0; i < n_nodes && org; i++)
            for (int k = 0; k < 3; k++) {
                uint32_t bits;
                std::memcpy(&bits, &nodes[i].p[k], 4);
                const float a = std::fabs(nodes[i].p[k]);
                org = org && (bits == 0u || (a >= 0x1p-36f && a <= 0x1p59f));
            }
        s->exp_exact = ok ? (org ? 2u : 1u) : 0u;
    }
    if (s->grid <= 0) return cleanup(fail(TRX_ERR_NO_DEVICE, "could not size the persistent grid"));
    *out = s;
    return TRX_OK;
}

void trx_scene_destroy(trx_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (RayService *&v : s->svc) { // (resident kernels first: the device-wide synchronisation below would wait for them)
        delete v;
        v = nullptr;
    }
    (void)hipDeviceSynchronize();
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_tris) (void)hipFree(s->d_tris);
    if (s->d_inst) (void)hipFree(s->d_inst);
    if (s->d_inst_entry) (void)hipFree(s->d_inst_entry);
    if (s->d_scratch_a) (void)hipFree(s->d_scratch_a);
    if (s->d_scratch_b) (void)hipFree(s->d_scratch_b);
    if (s->d_scratch_ia) (void)hipFree(s->d_scratch_ia);
    if (s->d_scratch_ib) (void)hipFree(s->d_scratch_ib);
    if (s->d_scratch_rays) (void)hipFree(s->d_scratch_rays);
    if (s->d_wave_times) (void)hipFree(s->d_wave_times);
    if (s->d_inst_xform) (void)hipFree(s->d_inst_xform);
    for (Slot &sl : s->slots) {
        if (sl.ctr) (void)hipFree(sl.ctr);
        if (sl.spill) (void)hipFree(sl.spill);
        for (auto &o : sl.order)
            if (o.lists) (void)hipFree(o.lists);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    {
        FrameLoop &fl = s->loop;
        for (int k = 0; k < 2; k++)
            if (fl.stream[k]) (void)hipStreamDestroy(fl.stream[k]);
        for (int k = 0; k < FrameLoop::kBuffers; k++) {
            if (fl.prim_done[k]) (void)hipEventDestroy(fl.prim_done[k]);
            if (fl.ao_done[k]) (void)hipEventDestroy(fl.ao_done[k]);
            if (fl.prim[k]) (void)hipFree(fl.prim[k]);
            if (fl.prim_inst[k]) (void)hipFree(fl.prim_inst[k]);
        }
        if (fl.t0) (void)hipEventDestroy(fl.t0);
        if (fl.t1) (void)hipEventDestroy(fl.t1);
        if (fl.ao) (void)hipFree(fl.ao);
        if (fl.ao_inst) (void)hipFree(fl.ao_inst);
    }
    delete s->comb;
    delete s;
}

uint64_t trx_scene_device_bytes(const trx_scene *s) {
    if (!s) return 0;
    uint64_t bytes = s->n_nodes * TRX_NODE_BYTES + s->n_tris * sizeof(TriDev) + (uint64_t)s->n_inst * 4;
    // launch slots claimed so far: stack spill areas and tile-order lists
    std::lock_guard<std::mutex> lock(const_cast<trx_scene *>(s)->mu);
    for (const Slot &sl : s->slots) {
        bytes += (uint64_t)sl.spill_waves * kWaveScratch * sizeof(uint2);
        for (const auto &o : sl.order)
            if (o.lists) bytes += 2ull * (16 * kLptShards + (uint64_t)16 * kLptShards * (o.capacity / 2 + 64)) * sizeof(uint32_t);
        if (sl.ctr) bytes += sizeof(SlotCounters);
    }
    // trx_frame_loop's record buffers (four primary, one AO; instance ids beside them on two-level scenes)
    bytes += s->loop.records * (FrameLoop::kBuffers + 1) * (sizeof(trx_hit) + (s->tlas ? sizeof(uint32_t) : 0));
    return bytes;
}
int trx_scene_device(const trx_scene *s) { return s ? s->device : -1; }

int trx_scene_set_geometry_ranges(trx_scene *s, const uint32_t *blas_tri_start, uint32_t n_blas) {
    if (!s || (!blas_tri_start && n_blas)) return fail(TRX_ERR_INVALID, "null argument");
    s->blas_tri_start.assign(blas_tri_start, blas_tri_start + n_blas + (n_blas ? 1 : 0));
    return TRX_OK;
}

// The scene under a launch slot's frozen tile order changed (instances moved, entry nodes replaced): the next frame of
// every slot files a new order even if its camera has not moved - an order learnt for other geometry is only stale, never
// wrong, but a static camera would replay it for ever.  Call with s->mu held.
static void forget_tile_orders(trx_scene *s) {
    for (Slot &sl : s->slots)
        for (auto &o : sl.order) o.have_views = false;
}

// Entry nodes: TLAS primitive k starts its BLAS walk at node entry_nodes[k] of the BLAS at instance_offsets[k] instead
// of node 0, so one BLAS can be referenced as several subtrees (re-braiding: a BLAS whose box spans the scene no longer
// makes every ray enter it at the root).  Validated like the node buffer: an entry must lie inside its BLAS segment.
int trx_scene_set_instance_entry_nodes(trx_scene *s, const uint32_t *entry_nodes, uint32_t n) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu);
    HIP_TRY(hipSetDevice(s->device));
    uint32_t *fresh = nullptr;
    if (entry_nodes && n) {
        if (!s->tlas) return fail(TRX_ERR_INVALID, "entry nodes need a TLAS scene");
        if (n != s->n_inst) return fail(TRX_ERR_INVALID, "%u entry nodes for %u instances", n, s->n_inst);
        std::vector<uint32_t> seg(s->h_inst);
        std::sort(seg.begin(), seg.end());
        seg.erase(std::unique(seg.begin(), seg.end()), seg.end());
        for (uint32_t k = 0; k < n; k++) {
            auto it = std::upper_bound(seg.begin(), seg.end(), s->h_inst[k]);
            const uint32_t seg_end = it == seg.end() ? s->tlas_start : *it;
            if ((uint64_t)s->h_inst[k] + entry_nodes[k] >= seg_end)
                return fail(TRX_ERR_FORMAT, "instance %u: entry node %u leaves its BLAS [%u, %u)", k, entry_nodes[k], s->h_inst[k], seg_end);
        }
        HIP_TRY(hipMalloc(&fresh, (size_t)n * sizeof(uint32_t)));
        const hipError_t e = hipMemcpy(fresh, entry_nodes, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(TRX_ERR_NO_DEVICE, "upload of the entry nodes failed: %s", hipGetErrorString(e));
        }
    }
    uint32_t *old = nullptr;
    {   // swapped under the launch mutex, freed after every kernel enqueued before the swap has drained
        std::lock_guard<std::mutex> lock(s->mu);
        old = s->d_inst_entry;
        s->d_inst_entry = fresh;
        forget_tile_orders(s);
    }
    HIP_TRY(hipDeviceSynchronize());
    if (old) (void)hipFree(old);
    return TRX_OK;
}

// Instance transforms: the TODOs at query_tlas.hlsl:409,433,484 and Traversable::get_instance_transform
// (traversable/src/lib.rs:25-27).  object_to_world: n column-major 4x4 matrices (glam Mat4) in TLAS-primitive order.
int trx_scene_set_instance_transforms(trx_scene *s, const float *object_to_world, uint32_t n) {
    if (!s) return fail(TRX_ERR_INVALID, "null scene");
    std::lock_guard<std::recursive_mutex> host_lock(s->host_mu);
    HIP_TRY(hipSetDevice(s->device));
    // The new table is built and uploaded first, the pointer is swapped under the launch mutex (enqueue() reads it
    // there, so no launch can pick up a table that is about to be freed), and the old table is freed only after
    // every kernel enqueued before the swap has drained.
    auto swap_table = [&](float4 *fresh) -> int {
        float4 *old = nullptr;
        {
            std::lock_guard<std::mutex> lock(s->mu);
            old = s->d_inst_xform;
            s->d_inst_xform = fresh;
            forget_tile_orders(s);
        }
        HIP_TRY(hipDeviceSynchronize());
        if (old) (void)hipFree(old);
        return TRX_OK;
    };
    if (!object_to_world || n == 0) { // back to identity
        const int rc = swap_table(nullptr);
        if (rc) return rc;
        s->inst_o2w.clear();
        s->inst_w2o.clear();
        return TRX_OK;
    }
    if (!s->tlas) return fail(TRX_ERR_INVALID, "instance transforms need a TLAS scene");
    if (n != s->n_inst) return fail(TRX_ERR_INVALID, "%u transforms for %u instances", n, s->n_inst);
    std::vector<float> w2o((size_t)n * 12);
    for (uint32_t k = 0; k < n; k++) {
        const float *m = object_to_world + (size_t)k * 16; // m[c*4 + r]
        if (m[3] != 0.f || m[7] != 0.f || m[11] != 0.f || m[15] != 1.f)
            return fail(TRX_ERR_INVALID, "instance %u: transform is not affine (last row must be 0 0 0 1)", k);
        // inverse of the affine map in double, rounded once to f32
        const double a = m[0], b = m[4], c = m[8], d = m[1], e = m[5], f = m[9], g = m[2], h = m[6], i = m[10];
        const double tx = m[12], ty = m[13], tz = m[14];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return fail(TRX_ERR_INVALID, "instance %u: singular transform", k);
        const double r = 1.0 / det;
        const double inv[9] = {(e * i - f * h) * r, (c * h - b * i) * r, (b * f - c * e) * r,
                               (f * g - d * i) * r, (a * i - c * g) * r, (c * d - a * f) * r,
                               (d * h - e * g) * r, (b * g - a * h) * r, (a * e - b * d) * r};
        float *o = &w2o[(size_t)k * 12];
        for (int row = 0; row < 3; row++) {
            o[4 * row + 0] = (float)inv[3 * row + 0];
            o[4 * row + 1] = (float)inv[3 * row + 1];
            o[4 * row + 2] = (float)inv[3 * row + 2];
            o[4 * row + 3] = (float)-(inv[3 * row + 0] * tx + inv[3 * row + 1] * ty + inv[3 * row + 2] * tz);
        }
    }
    float4 *d = nullptr;
    HIP_TRY(hipMalloc(&d, w2o.size() * sizeof(float)));
    hipError_t e = hipMemcpy(d, w2o.data(), w2o.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(TRX_ERR_NO_DEVICE, "upload of the instance transforms failed: %s", hipGetErrorString(e));
    }
    const int rc_swap = swap_table(d);
    if (rc_swap) return rc_swap;
    s->inst_o2w.assign(object_to_world, object_to_world + (size_t)n * 16);
    s->inst_w2o.swap(w2o);
    return TRX_OK;
}

int trx_scene_get_instance_transform(const trx_scene *s, uint32_t instance_id, float out_object_to_world[16]) {
    if (!s || !out_object_to_world) return fail(TRX_ERR_INVALID, "null argument");
    if (instance_id >= std::max<uint32_t>(s->n_inst, 1)) return fail(TRX_ERR_INVALID, "instance %u of %u", instance_id, s->n_inst);
    if (s->inst_o2w.empty()) { // Mat4::default(): identity, like the reference (src/cwbvh.rs:163-165,189-191)
        for (int k = 0; k < 16; k++) out_object_to_world[k] = (k % 5 == 0) ? 1.f : 0.f;
        return TRX_OK;
    }
    std::memcpy(out_object_to_world, &s->inst_o2w[(size_t)instance_id * 16], 64);
    return TRX_OK;
}

int trx_scene_get_instance_world_to_object(const trx_scene *s, uint32_t instance_id, float out_rows[12]) {
    if (!s || !out_rows) return fail(TRX_ERR_INVALID, "null argument");
    if (instance_id >= std::max<uint32_t>(s->n_inst, 1)) return fail(TRX_ERR_INVALID, "instance %u of %u", instance_id, s->n_inst);
    if (s->inst_w2o.empty()) {
        static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        std::memcpy(out_rows, ident, 48);
        return TRX_OK;
    }
    std::memcpy(out_rows, &s->inst_w2o[(size_t)instance_id * 12], 48);
    return TRX_OK;
}

// ---- camera: ViewUniform::from_camera, src/main.rs:602-616 -----------------------------
static void mat4_inverse_f32(const float m[16], float out[16]) {
    // cofactor expansion (same scheme as glam's Mat4::inverse), evaluated in f32
    const float a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3], a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7];
    const float a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11], a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15];
    const float b00 = a00 * a11 - a01 * a10, b01 = a00 * a12 - a02 * a10, b02 = a00 * a13 - a03 * a10;
    const float b03 = a01 * a12 - a02 * a11, b04 = a01 * a13 - a03 * a11, b05 = a02 * a13 - a03 * a12;
    const float b06 = a20 * a31 - a21 * a30, b07 = a20 * a32 - a22 * a30, b08 = a20 * a33 - a23 * a30;
    const float b09 = a21 * a32 - a22 * a31, b10 = a21 * a33 - a23 * a31, b11 = a22 * a33 - a23 * a32;
    const float det = b00 * b11 - b01 * b10 + b02 * b09 + b03 * b08 - b04 * b07 + b05 * b06;
    const float r = 1.0f / det;
    out[0] = (a11 * b11 - a12 * b10 + a13 * b09) * r;
    out[1] = (a02 * b10 - a01 * b11 - a03 * b09) * r;
    out[2] = (a31 * b05 - a32 * b04 + a33 * b03) * r;
    out[3] = (a22 * b04 - a21 * b05 - a23 * b03) * r;
    out[4] = (a12 * b08 - a10 * b11 - a13 * b07) * r;
    out[5] = (a00 * b11 - a02 * b08 + a03 * b07) * r;
    out[6] = (a32 * b02 - a30 * b05 - a33 * b01) * r;
    out[7] = (a20 * b05 - a22 * b02 + a23 * b01) * r;
    out[8] = (a10 * b10 - a11 * b08 + a13 * b06) * r;
    out[9] = (a01 * b08 - a00 * b10 - a03 * b06) * r;
    out[10] = (a30 * b04 - a31 * b02 + a33 * b00) * r;
    out[11] = (a21 * b02 - a20 * b04 - a23 * b00) * r;
    out[12] = (a11 * b07 - a10 * b09 - a12 * b06) * r;
    out[13] = (a00 * b09 - a01 * b07 + a02 * b06) * r;
    out[14] = (a31 * b01 - a30 * b03 - a32 * b00) * r;
    out[15] = (a20 * b03 - a21 * b01 + a22 * b00) * r;
}

int trx_view_from_camera(const float eye[3], const float look_at[3], float fov_deg, float width, float height,
                         trx_view *out) {
    if (!eye || !look_at || !out) return fail(TRX_ERR_INVALID, "null argument");
    if (!(width > 0.f) || !(height > 0.f)) return fail(TRX_ERR_INVALID, "bad image size");
    std::memset(out, 0, sizeof(*out));
    const float aspect = width / height;
    const float fov = fov_deg * (3.14159265358979323846f / 180.0f);
    const float f = 1.0f / std::tan(0.5f * fov);
    float proj[16] = {0};
    proj[0] = f / aspect;
    proj[5] = f;
    proj[11] = -1.0f;
    proj[14] = 0.01f;
    mat4_inverse_f32(proj, out->proj_inv);
    float fw[3] = {look_at[0] - eye[0], look_at[1] - eye[1], look_at[2] - eye[2]};
    float fl = std::sqrt(fw[0] * fw[0] + fw[1] * fw[1] + fw[2] * fw[2]);
    if (!(fl > 0.f)) return fail(TRX_ERR_INVALID, "eye == look_at");
    for (float &x : fw) x /= fl;
    // s = normalize(f x up), up = +Y
    float s[3] = {fw[1] * 0.f - fw[2] * 1.f, fw[2] * 0.f - fw[0] * 0.f, fw[0] * 1.f - fw[1] * 0.f};
    float sl = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    if (!(sl > 0.f)) return fail(TRX_ERR_INVALID, "view direction parallel to +Y");
    for (float &x : s) x /= sl;
    float u[3] = {s[1] * fw[2] - s[2] * fw[1], s[2] * fw[0] - s[0] * fw[2], s[0] * fw[1] - s[1] * fw[0]};
    auto dot = [](const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    float view[16] = {s[0], u[0], -fw[0], 0, s[1], u[1], -fw[1], 0, s[2], u[2], -fw[2], 0,
                      -dot(s, eye), -dot(u, eye), dot(fw, eye), 1};
    mat4_inverse_f32(view, out->view_inv);
    std::memcpy(out->eye, eye, 12);
    return TRX_OK;
}

} // extern "C"
