"""Host-side mirror of the reference's CWBVH GPU path over the C-ABI.

Names follow the reference: `cwbvh_gpu_runner` (src/rt_gpu/mod.rs:16-112)
assembles the flat node / triangle / instance buffers, `Scene` owns what
rt_gpu_software::start uploaded (src/rt_gpu/rt_gpu_software.rs:24-32,83-160),
and `Scene.start` returns the frame time in ms like the reference does (:376).
"""
import ctypes as C

import numpy as np

from . import _lib as L

HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<u4")])
RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("direction", "<f4", 3), ("tmax", "<f4")])
RAYHIT_DTYPE = np.dtype([("primitive_id", "<u4"), ("geometry_id", "<u4"), ("instance_id", "<u4"), ("t", "<f4")])  # obvhs RayHit
MISS_PRIM = 0xFFFFFFFF


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- scenes / models ----------------------------------------------------------

def _take_mesh(verts_p, n, counts_p, nobj):
    lib = L.load()
    verts = np.ctypeslib.as_array(verts_p, shape=(max(n.value, 1), 9))[: n.value].copy()
    counts = np.ctypeslib.as_array(counts_p, shape=(max(nobj.value, 1),))[: nobj.value].copy()
    lib.trx_free(C.cast(verts_p, C.c_void_p))
    lib.trx_free(C.cast(counts_p, C.c_void_p))
    return verts, counts.astype(np.uint64)


def gen_scene(name, n_tris=0, seed=1):
    """Seeded procedural stand-in for one of the reference's absent assets."""
    lib = L.load()
    vp, cp = C.POINTER(C.c_float)(), C.POINTER(C.c_uint64)()
    n, nobj = C.c_uint64(), C.c_uint32()
    L.check(lib.trx_gen_scene(name.encode(), n_tris, seed, C.byref(vp), C.byref(n), C.byref(cp), C.byref(nobj)))
    return _take_mesh(vp, n, cp, nobj)


def load_meshs(path):
    """load_meshs (src/main.rs:494-561): OBJ or JSON -> (verts [n,9], triangles per object)."""
    lib = L.load()
    vp, cp = C.POINTER(C.c_float)(), C.POINTER(C.c_uint64)()
    n, nobj = C.c_uint64(), C.c_uint32()
    L.check(lib.trx_load_model(str(path).encode(), C.byref(vp), C.byref(n), C.byref(cp), C.byref(nobj)))
    return _take_mesh(vp, n, cp, nobj)


def load_scene(path):
    """A scene file of the reference (assets/scenes/*.ron) as src/main.rs:259-298 reads it: (verts [n,9], triangles per
    object, eye, look_at, fov) - trx_load_scene."""
    lib = L.load()
    vp, cp = C.POINTER(C.c_float)(), C.POINTER(C.c_uint64)()
    n, nobj = C.c_uint64(), C.c_uint32()
    eye, look, fov = (C.c_float * 3)(), (C.c_float * 3)(), C.c_float()
    L.check(lib.trx_load_scene(str(path).encode(), C.byref(vp), C.byref(n), C.byref(cp), C.byref(nobj), eye, look, C.byref(fov)))
    verts, counts = _take_mesh(vp, n, cp, nobj)
    return verts, counts, list(eye), list(look), fov.value


def copy_rate(device=0, nbytes=1 << 30, reps=5):
    """Streaming ceiling of the device's memory, bytes read + written per second by a float4 copy kernel (trx_debug_copy_rate)."""
    out = C.c_double()
    L.check(L.load().trx_debug_copy_rate(device, nbytes, reps, C.byref(out)))
    return out.value


def scene_camera(name):
    lib = L.load()
    eye, look, fov = (C.c_float * 3)(), (C.c_float * 3)(), C.c_float()
    L.check(lib.trx_scene_camera(name.encode(), eye, look, C.byref(fov)))
    return list(eye), list(look), fov.value


def view_from_camera(eye, look_at, fov_deg, width, height):
    """ViewUniform::from_camera (src/main.rs:602-616)."""
    lib = L.load()
    v = L.View()
    L.check(lib.trx_view_from_camera((C.c_float * 3)(*eye), (C.c_float * 3)(*look_at), fov_deg, float(width),
                                     float(height), C.byref(v)))
    return v


# ---- flat buffers (cwbvh_gpu_runner's assembly half) -----------------------------

class FlatScene:
    """The buffers rt_gpu_software::start receives: bvh_bytes, tri_bytes, instance_bytes, tlas_start."""

    def __init__(self, nodes, tri_verts, instance_offsets, tlas_start, tri_source, blas_tri_start,
                 blas_build_s=0.0, tlas_build_s=0.0, tri_boxes=None, instance_source=None, instance_transforms=None,
                 instance_entry=None):
        self.nodes = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 20)
        self.tri_verts = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
        self.instance_offsets = np.ascontiguousarray(instance_offsets, dtype=np.uint32)
        self.tlas_start = int(tlas_start)
        self.tri_source = np.ascontiguousarray(tri_source, dtype=np.uint32)
        self.blas_tri_start = np.ascontiguousarray(blas_tri_start, dtype=np.uint32)
        self.blas_build_s = blas_build_s
        self.tlas_build_s = tlas_build_s
        # box every triangle entry was built with (pre-split references cover only part of their triangle)
        self.tri_boxes = None if tri_boxes is None else np.ascontiguousarray(tri_boxes, dtype=np.float32).reshape(-1, 6)
        # TLAS primitive k = the caller's object / instance instance_source[k]; object-to-world 4x4 (column-major)
        # per TLAS primitive, or None = identity (the reference, src/cwbvh.rs:163-165)
        self.instance_source = None if instance_source is None else np.ascontiguousarray(instance_source, dtype=np.uint32)
        self.instance_transforms = None if instance_transforms is None else np.ascontiguousarray(
            instance_transforms, dtype=np.float32).reshape(-1, 16)
        # node of its BLAS at which TLAS primitive k starts (re-braided TLAS), or None = node 0 (the reference)
        self.instance_entry = None if instance_entry is None else np.ascontiguousarray(instance_entry, dtype=np.uint32)

    @property
    def n_nodes(self):
        return self.nodes.shape[0]

    @property
    def n_tris(self):
        return self.tri_verts.shape[0]

    @property
    def has_tlas(self):
        return self.instance_offsets.size > 0


def flat_build(verts, object_counts=None, use_tlas=False, max_prims_per_leaf=3, threads=0, traversal_cost=None,
               prim_cost=None, reinsertion=None, preset=None, split=None):
    """`reinsertion`: None keeps the process-wide setting; a float is the batch ratio (2 iterations),
    a (ratio, iterations) pair sets both (trx_set_build_reinsertion)."""
    lib = L.load()
    if preset is not None:   # the reference's --preset names (trx_set_build_preset); "" = defaults
        L.check(lib.trx_set_build_preset(preset.encode()))
    if split is not None:    # pre-splitting: extra triangle references as a fraction of n (trx_set_build_split)
        L.check(lib.trx_set_build_split(float(split)))
    if traversal_cost is not None or prim_cost is not None:
        L.check(lib.trx_set_build_costs(traversal_cost or 1.0, prim_cost or 0.3))
    if reinsertion is not None:
        ratio, iters = reinsertion if isinstance(reinsertion, (tuple, list)) else (reinsertion, 2)
        L.check(lib.trx_set_build_reinsertion(float(ratio), int(iters)))
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    if object_counts is None:
        object_counts = [verts.shape[0]]
    counts = np.ascontiguousarray(object_counts, dtype=np.uint64)
    if int(counts.sum()) != verts.shape[0]:
        raise L.TrxError(L.TRX_ERR_INVALID, "object counts do not add up to the triangle count")
    fp = C.POINTER(L.Flat)()
    L.check(lib.trx_flat_build(_ptr(verts), _ptr(counts), counts.size, 1 if use_tlas else 0, max_prims_per_leaf,
                               threads, C.byref(fp)))
    return _take_flat(lib, fp)


class _FlatOwner:
    """Keeps a trx_flat alive while numpy views of its buffers exist (trx_flat_destroy when the last one goes)."""

    def __init__(self, lib, fp):
        self.lib, self.fp = lib, fp

    def __del__(self):
        try:
            self.lib.trx_flat_destroy(self.fp)
        except Exception:   # interpreter shutdown
            pass


def _view(owner, address, dtype, count, shape):
    """A numpy array over `count` items of the library's memory at `address` (no copy: a 3.9 M triangle scene has
    280 MB of them); the array holds `owner`."""
    dtype = np.dtype(dtype)
    if count == 0 or not address:
        return np.zeros(shape, dtype=dtype)
    raw = (C.c_uint8 * (count * dtype.itemsize)).from_address(address)
    raw._trx_owner = owner
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def _take_flat(lib, fp):
    f = fp.contents
    owner = _FlatOwner(lib, fp)
    addr = lambda p: C.cast(p, C.c_void_p).value
    nodes = _view(owner, f.bvh_bytes, np.uint32, f.n_nodes * 20, (f.n_nodes, 20))
    tris = _view(owner, addr(f.tri_verts), np.float32, f.n_tris * 9, (f.n_tris, 9))
    inst = _view(owner, addr(f.instance_offsets), np.uint32, f.n_instances, (f.n_instances,))
    src = _view(owner, addr(f.tri_source), np.uint32, f.n_tris, (f.n_tris,))
    bts = _view(owner, addr(f.blas_tri_start), np.uint32, f.n_blas + 1, (f.n_blas + 1,))
    boxes = _view(owner, addr(f.tri_boxes), np.float32, f.n_tris * 6, (f.n_tris, 6))
    isrc = _view(owner, addr(f.instance_source), np.uint32, f.n_instances, (f.n_instances,))
    ixf = _view(owner, addr(f.instance_transforms), np.float32, f.n_instances * 16, (f.n_instances, 16)) if f.instance_transforms else None
    ient = _view(owner, addr(f.instance_entry_nodes), np.uint32, f.n_instances, (f.n_instances,)) if f.instance_entry_nodes else None
    return FlatScene(nodes, tris, inst, f.tlas_start, src, bts, f.blas_build_s, f.tlas_build_s, boxes, isrc, ixf, ient)


def flat_build_instanced(verts, object_counts, instance_object, object_to_world=None, max_prims_per_leaf=3, threads=0):
    """One BLAS per object, a TLAS over instances of them (trx_flat_build_instanced).  object_to_world: [n, 16]
    column-major affine matrices (glam Mat4), or None for identity."""
    lib = L.load()
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    counts = np.ascontiguousarray(object_counts, dtype=np.uint64)
    io = np.ascontiguousarray(instance_object, dtype=np.uint32)
    xf = None if object_to_world is None else np.ascontiguousarray(object_to_world, dtype=np.float32).reshape(-1, 16)
    if xf is not None and xf.shape[0] != io.size:
        raise L.TrxError(L.TRX_ERR_INVALID, "one transform per instance")
    fp = C.POINTER(L.Flat)()
    L.check(lib.trx_flat_build_instanced(_ptr(verts), _ptr(counts), counts.size, _ptr(io), _ptr(xf) if xf is not None else None,
                                         io.size, max_prims_per_leaf, threads, C.byref(fp)))
    return _take_flat(lib, fp)


def build_params(**fields):
    """trx_build_params with the reference's command-line defaults (src/main.rs:85-124), fields overridden by name."""
    lib = L.load()
    bp = L.BuildParams()
    lib.trx_build_params_default(C.byref(bp))
    for k, v in fields.items():
        if not hasattr(bp, k):
            raise AttributeError("BvhBuildParams has no field %r" % k)
        setattr(bp, k, v)
    return bp


def flat_build_params(verts, object_counts, params, use_tlas=False, threads=0):
    """cwbvh_gpu_runner's build half driven by a BvhBuildParams (src/main.rs:571-585) instead of process-wide knobs."""
    lib = L.load()
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    counts = np.ascontiguousarray(object_counts if object_counts is not None else [verts.shape[0]], dtype=np.uint64)
    fp = C.POINTER(L.Flat)()
    L.check(lib.trx_flat_build_params(_ptr(verts), _ptr(counts), counts.size, 1 if use_tlas else 0, C.byref(params), threads,
                                      C.byref(fp)))
    return _take_flat(lib, fp)


def flat_build_preset_device(verts, object_counts, preset="medium_build", device=0, use_tlas=False, max_prims_per_leaf=3, threads=0):
    """A preset's build with every stage on the device (trx_flat_build_preset_device); device < 0: its host twin."""
    lib = L.load()
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    counts = np.ascontiguousarray(object_counts if object_counts is not None else [verts.shape[0]], dtype=np.uint64)
    fp = C.POINTER(L.Flat)()
    L.check(lib.trx_flat_build_preset_device(_ptr(verts), _ptr(counts), counts.size, 1 if use_tlas else 0, preset.encode(),
                                             max_prims_per_leaf, threads, device, C.byref(fp)))
    return _take_flat(lib, fp)


def pack_tris_f16(tri_verts):
    """obvhs RtCompressedTriangle (24 B): v0 f32x3 + f16 edges, e2 in the low half (query.hlsl:75-85)."""
    v = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
    e1 = (v[:, 3:6] - v[:, 0:3]).astype(np.float16).view(np.uint16).astype(np.uint32)
    e2 = (v[:, 6:9] - v[:, 0:3]).astype(np.float16).view(np.uint16).astype(np.uint32)
    out = np.empty((v.shape[0], 6), dtype=np.uint32)
    out[:, 0:3] = v[:, 0:3].view(np.uint32)
    out[:, 3:6] = e2 | (e1 << 16)
    return out


# ---- device scene ---------------------------------------------------------------------

class Scene:
    """Device-resident CWBVH scene (what rt_gpu_software::start uploads)."""

    def __init__(self, flat, device=0, tri_format=L.TRI_VERTS_36, tri_bytes=None):
        lib = L.load()
        self._lib = lib
        self._h = C.c_void_p()
        self.flat = flat
        nodes = flat.nodes
        if tri_bytes is None:
            tri_bytes = flat.tri_verts
        tri_bytes = np.ascontiguousarray(tri_bytes)
        inst = flat.instance_offsets
        L.check(lib.trx_scene_create(_ptr(nodes), flat.n_nodes, _ptr(tri_bytes), flat.n_tris, tri_format,
                                     _ptr(inst) if inst.size else None, inst.size, flat.tlas_start, device,
                                     C.byref(self._h)))
        if flat.blas_tri_start.size > 1:
            L.check(lib.trx_scene_set_geometry_ranges(self._h, _ptr(flat.blas_tri_start),
                                                      flat.blas_tri_start.size - 1))
        if getattr(flat, "instance_transforms", None) is not None:
            self.set_instance_transforms(flat.instance_transforms)
        if getattr(flat, "instance_entry", None) is not None:
            L.check(lib.trx_scene_set_instance_entry_nodes(self._h, _ptr(flat.instance_entry), flat.instance_entry.size))

    def set_instance_transforms(self, object_to_world):
        """Object-to-world 4x4 (column-major) per TLAS primitive; None restores identity."""
        if object_to_world is None:
            L.check(self._lib.trx_scene_set_instance_transforms(self._h, None, 0))
            return
        xf = np.ascontiguousarray(object_to_world, dtype=np.float32).reshape(-1, 16)
        L.check(self._lib.trx_scene_set_instance_transforms(self._h, _ptr(xf), xf.shape[0]))

    def instance_transform(self, instance_id):
        """Traversable::get_instance_transform (traversable/src/lib.rs:25-27): object-to-world, column-major."""
        m = np.zeros(16, dtype=np.float32)
        L.check(self._lib.trx_scene_get_instance_transform(self._h, instance_id, _ptr(m)))
        return m

    def instance_world_to_object(self):
        """[n_instances, 12] world-to-object rows exactly as the kernels use them."""
        n = self.flat.instance_offsets.size
        out = np.zeros((n, 12), dtype=np.float32)
        for k in range(n):
            L.check(self._lib.trx_scene_get_instance_world_to_object(self._h, k, _ptr(out[k])))
        return out

    def close(self):
        if self._h:
            self._lib.trx_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def device_bytes(self):
        return self._lib.trx_scene_device_bytes(self._h)

    # host-buffer entry points -------------------------------------------------------
    def trace_primary(self, view, width, height, sem=L.SEM_HLSL):
        hits = np.empty(width * height, dtype=HIT_DTYPE)
        ms = C.c_float()
        L.check(self._lib.trx_trace_primary(self._h, C.byref(view), width, height, sem, _ptr(hits), C.byref(ms)))
        return hits, ms.value

    def trace_primary_ao(self, view, width, height, sem=L.SEM_HLSL, frame=0, ao_eps=0.01):
        prim = np.empty(width * height, dtype=HIT_DTYPE)
        ao = np.empty(width * height, dtype=HIT_DTYPE)
        ms = C.c_float()
        L.check(self._lib.trx_trace_primary_ao(self._h, C.byref(view), width, height, sem, frame, ao_eps, _ptr(prim),
                                               _ptr(ao), C.byref(ms)))
        return prim, ao, ms.value

    def frame_loop(self, view, width, height, sem=L.SEM_HLSL, frames=16, frame0=0, animate=True, ao_eps=0.01, overlap=False,
                   fetch=True):
        """The reference's frame loop, device-resident (trx_frame_loop): (total ms, last primary records, last AO records)."""
        prim = np.empty(width * height, dtype=HIT_DTYPE) if fetch else None
        ao = np.empty(width * height, dtype=HIT_DTYPE) if fetch else None
        ms = C.c_float()
        L.check(self._lib.trx_frame_loop(self._h, C.byref(view), width, height, sem, frame0, 1 if animate else 0, ao_eps, frames,
                                         int(overlap), _ptr(prim) if fetch else None, _ptr(ao) if fetch else None, C.byref(ms)))
        return ms.value, prim, ao

    def trace_primary_ao_inst(self, view, width, height, sem=L.SEM_HLSL, frame=0, ao_eps=0.01):
        """(primary, primary instance ids, ao, ao instance ids, ms): RayHit.instance_id beside every hit."""
        n = width * height
        prim, ao = np.empty(n, dtype=HIT_DTYPE), np.empty(n, dtype=HIT_DTYPE)
        pi, ai = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32)
        ms = C.c_float()
        L.check(self._lib.trx_trace_primary_ao_inst(self._h, C.byref(view), width, height, sem, frame, ao_eps, _ptr(prim),
                                                    _ptr(pi), _ptr(ao), _ptr(ai), C.byref(ms)))
        return prim, pi, ao, ai, ms.value

    def trace_rays_inst(self, rays, sem=L.SEM_HLSL):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.empty(rays.shape[0], dtype=HIT_DTYPE)
        inst = np.empty(rays.shape[0], dtype=np.uint32)
        ms = C.c_float()
        L.check(self._lib.trx_trace_rays_inst(self._h, _ptr(rays), rays.shape[0], sem, _ptr(hits), _ptr(inst), C.byref(ms)))
        return hits, inst, ms.value

    def trace_rays(self, rays, sem=L.SEM_HLSL):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.empty(rays.shape[0], dtype=HIT_DTYPE)
        ms = C.c_float()
        L.check(self._lib.trx_trace_rays(self._h, _ptr(rays), rays.shape[0], sem, _ptr(hits), C.byref(ms)))
        return hits, ms.value

    def trace_occluded(self, rays, sem=L.SEM_HLSL):
        """intersects_bl_bvh (query.hlsl:440-445) for a batch: uint8 flags, 1 = something is hit."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        flags = np.empty(rays.shape[0], dtype=np.uint8)
        ms = C.c_float()
        L.check(self._lib.trx_trace_occluded(self._h, _ptr(rays), rays.shape[0], sem, _ptr(flags), C.byref(ms)))
        return flags, ms.value

    def traverse(self, origin, direction, tmin=0.0, tmax=3.4028234663852886e38, sem=L.SEM_HLSL):
        """Traversable::traverse (traversable/src/lib.rs:17-21) for one ray."""
        ray = L.Ray((C.c_float * 3)(*origin), tmin, (C.c_float * 3)(*direction), tmax)
        out = L.RayHit()
        L.check(self._lib.trx_traverse1(self._h, C.byref(ray), sem, C.byref(out)))
        return out

    def traverse_threads(self, rays, threads=16, sem=L.SEM_HLSL):
        """The reference's CPU pixel loop over the literal traverse (src/rt_cpu/rt_cpu.rs:35-57): `threads` host threads
        (native ones, inside the library: trx_debug_traverse1_threads), thread k calls trx_traverse1 for rays k,
        k + threads, ...  Returns (RAYHIT_DTYPE array, seconds, launches the calls shared)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(rays.shape[0], dtype=RAYHIT_DTYPE)
        secs, launches = C.c_double(), C.c_uint64()
        L.check(self._lib.trx_debug_traverse1_threads(self._h, _ptr(rays), rays.shape[0], threads, sem, _ptr(out),
                                                      C.byref(secs), C.byref(launches)))
        return out, secs.value, int(launches.value)

    def service_stats(self):
        """The scene's ray services so far (trx_debug_service_stats): dict of rays, starts, us_per_call, gpu_us_per_call, trips_per_call."""
        v = [C.c_uint64() for _ in range(5)]
        L.check(self._lib.trx_debug_service_stats(self._h, *[C.byref(x) for x in v]))
        n = max(v[0].value, 1)
        return {"rays": v[0].value, "starts": v[1].value, "us_per_call": v[2].value / n * 1e-3, "gpu_us_per_call": v[3].value / n * 1e-2,
                "trips_per_call": v[4].value / n}

    def traverse_batch(self, rays, sem=L.SEM_HLSL):
        """Traversable::traverse for a whole batch in one launch: (RAYHIT_DTYPE array, ms)."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.empty(rays.shape[0], dtype=RAYHIT_DTYPE)
        ms = C.c_float()
        L.check(self._lib.trx_traverse_batch(self._h, _ptr(rays), rays.shape[0], sem, _ptr(out), C.byref(ms)))
        return out, ms.value

    def count_ao(self, view, width, height, d_primary, d_ao, sem=L.SEM_HLSL, frame=0, ao_eps=0.01, shard=(0, 1)):
        """Node steps / triangle tests of one AO pass over device-resident primary hits (counting kernel)."""
        st = L.Stats()
        L.check(self._lib.trx_count_ao(self._h, C.byref(view), width, height, L.Shard(*shard), sem, frame, ao_eps,
                                       C.c_void_p(d_primary), C.c_void_p(d_ao), C.byref(st)))
        return st

    def count_rays(self, d_rays, n, d_hits, sem=L.SEM_HLSL):
        """Node steps / triangle tests of one explicit-ray batch (counting kernel)."""
        st = L.Stats()
        L.check(self._lib.trx_count_rays(self._h, C.c_void_p(d_rays), n, sem, C.c_void_p(d_hits), C.byref(st)))
        return st

    def fetch_rate(self, tris_per_node=0.0, steps=400):
        """Measured ceiling of the node-fetch loop on this scene's buffers (trx_debug_fetch_rate): random 80-byte nodes
        and, tris_per_node per node on average, random 48-byte triangle records per second, nothing else running."""
        nps, tps = C.c_double(), C.c_double()
        L.check(self._lib.trx_debug_fetch_rate(self._h, steps, int(round(tris_per_node * 256)), C.byref(nps), C.byref(tps)))
        return nps.value, tps.value

    def count_primary(self, view, width, height, sem=L.SEM_HLSL, shard=(0, 1)):
        st = L.Stats()
        L.check(self._lib.trx_count_primary(self._h, C.byref(view), width, height, L.Shard(*shard), sem, None,
                                            C.byref(st)))
        return st

    def footprint(self, view, width, height, sem=L.SEM_HLSL):
        """(distinct nodes fetched, distinct triangles tested) by one primary frame."""
        n, t = C.c_uint64(), C.c_uint64()
        L.check(self._lib.trx_debug_footprint(self._h, C.byref(view), width, height, sem, C.byref(n), C.byref(t)))
        return n.value, t.value

    def bench_primary(self, view, width, height, sem=L.SEM_HLSL, warmup=1, frames=20):
        mn, mean = C.c_float(), C.c_float()
        L.check(self._lib.trx_bench_primary(self._h, C.byref(view), width, height, sem, warmup, frames, C.byref(mn),
                                            C.byref(mean)))
        return mn.value, mean.value

    # device-resident entry points (pointers are ints: tensor.data_ptr(), stream.cuda_stream) ----
    def trace_primary_dev(self, view, width, height, d_hits, sem=L.SEM_HLSL, shard=(0, 1), stream=0):
        L.check(self._lib.trx_trace_primary_dev(self._h, C.byref(view), width, height, L.Shard(*shard), sem,
                                                C.c_void_p(d_hits), C.c_void_p(stream)))

    def trace_primary_batch_dev(self, views, width, height, d_hits, frame_stride, sem=L.SEM_HLSL, shard=(0, 1),
                                stream=0):
        """len(views) frames (1..8) in one launch; frame f lands at d_hits + f*frame_stride records."""
        arr = (L.View * len(views))(*views)
        L.check(self._lib.trx_trace_primary_batch_dev(self._h, arr, len(views), width, height, L.Shard(*shard), sem,
                                                      C.c_void_p(d_hits), frame_stride, C.c_void_p(stream)))

    def trace_ao_dev(self, view, width, height, d_primary, d_ao, sem=L.SEM_HLSL, frame=0, ao_eps=0.01,
                     shard=(0, 1), stream=0):
        L.check(self._lib.trx_trace_ao_dev(self._h, C.byref(view), width, height, L.Shard(*shard), sem, frame, ao_eps,
                                           C.c_void_p(d_primary), C.c_void_p(d_ao), C.c_void_p(stream)))

    def trace_frame_dev(self, view, width, height, d_primary, d_ao, sem=L.SEM_HLSL, frame=0, ao_eps=0.01, shard=(0, 1),
                        stream=0, d_primary_inst=0, d_ao_inst=0):
        """Primary + AO of one frame in ONE launch (the reference's single dispatch)."""
        L.check(self._lib.trx_trace_frame_dev(self._h, C.byref(view), width, height, L.Shard(*shard), sem, frame, ao_eps,
                                              C.c_void_p(d_primary), C.c_void_p(d_primary_inst), C.c_void_p(d_ao),
                                              C.c_void_p(d_ao_inst), C.c_void_p(stream)))

    def trace_ao_batch_dev(self, view, width, height, d_primary, d_ao, frame_stride, n_frames, sem=L.SEM_HLSL, frame0=0,
                           ao_eps=0.01, shard=(0, 1), stream=0, d_primary_inst=0, d_ao_inst=0):
        """n_frames AO passes (seeds frame0 ..) over one view and one primary hit buffer in ONE launch."""
        L.check(self._lib.trx_trace_ao_batch_dev(self._h, C.byref(view), width, height, L.Shard(*shard), sem, frame0, n_frames,
                                                 ao_eps, C.c_void_p(d_primary), C.c_void_p(d_primary_inst), C.c_void_p(d_ao),
                                                 C.c_void_p(d_ao_inst), frame_stride, C.c_void_p(stream)))

    def trace_rays_dev(self, d_rays, n, d_hits, sem=L.SEM_HLSL, stream=0):
        L.check(self._lib.trx_trace_rays_dev(self._h, C.c_void_p(d_rays), n, sem, C.c_void_p(d_hits),
                                             C.c_void_p(stream)))

    def trace_occluded_dev(self, d_rays, n, d_flags, sem=L.SEM_HLSL, stream=0):
        L.check(self._lib.trx_trace_occluded_dev(self._h, C.c_void_p(d_rays), n, sem, C.c_void_p(d_flags),
                                                 C.c_void_p(stream)))

    def check(self, stream=0):
        L.check(self._lib.trx_scene_check(self._h, C.c_void_p(stream)))


def cwbvh_gpu_runner(verts, object_counts, width, height, camera, tlas=False, max_prims_per_leaf=3, benchmark=True,
                     frames=20, sem=L.SEM_HLSL, device=0, threads=0):
    """cwbvh_gpu_runner (src/rt_gpu/mod.rs:16-112) on the HIP backend: build the BLAS/TLAS,
    assemble the flat buffers, upload, trace, return (frame_ms, blas_build_s, tlas_build_ms)."""
    flat = flat_build(verts, object_counts, use_tlas=tlas, max_prims_per_leaf=max_prims_per_leaf, threads=threads)
    scene = Scene(flat, device=device)
    try:
        eye, look_at, fov = camera
        view = view_from_camera(eye, look_at, fov, width, height)
        mn, _mean = scene.bench_primary(view, width, height, sem=sem, warmup=1 if benchmark else 0, frames=frames)
        return mn, flat.blas_build_s, flat.tlas_build_s * 1000.0
    finally:
        scene.close()
