"""ctypes binding of liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under tray_racing_amd/ may import this module.
PARITY UNPINNED: see trx_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<u4")])
RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("direction", "<f4", 3), ("tmax", "<f4")])

SEM_HLSL, SEM_NODE_RCP, SEM_TIE_FIRST, SEM_NODE_FMA, SEM_CPU = 0, 1, 2, 4, 3
F32_MAX = 3.4028234663852886e38


class View(C.Structure):
    _fields_ = [("view_inv", C.c_float * 16), ("proj_inv", C.c_float * 16), ("eye", C.c_float * 3),
                ("exposure", C.c_float), ("tlas_start", C.c_uint32), ("pad", C.c_uint32 * 3)]


class SceneC(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint64), ("tris", C.c_void_p), ("n_tris", C.c_uint64),
                ("instance_offsets", C.c_void_p), ("n_instances", C.c_uint32), ("tlas_start", C.c_uint32),
                ("instance_w2o", C.c_void_p), ("instance_entry", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("n_rays", C.c_uint64), ("n_node", C.c_uint64), ("n_tri", C.c_uint64), ("n_hits", C.c_uint64),
                ("max_stack", C.c_uint32), ("overflow", C.c_uint32), ("seconds", C.c_double), ("threads", C.c_int),
                ("n_tlas_node", C.c_uint64), ("n_inst_enter", C.c_uint64)]


class HitC(C.Structure):
    _fields_ = [("t", C.c_float), ("prim", C.c_uint32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    P, u32, u64, i, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float
    VP, SP, STP = C.POINTER(View), C.POINTER(SceneC), C.POINTER(Stats)
    sigs = {
        "orc_tris_from_verts": (None, [P, u64, P]),
        "orc_set_ao_libm": (None, [i]),
        "orc_set_simd": (None, [i]),
        "orc_get_simd": (i, []),
        "orc_tris_from_f16": (None, [P, u64, P]),
        "orc_view_from_camera": (None, [P, P, f, f, f, VP]),
        "orc_octant_inv4": (u32, [P]),
        "orc_node_intersect": (u32, [P, P, P, u32, f, P, u32]),
        "orc_intersect_tri": (i, [P, P, P, f, P, u32]),
        "orc_primary_ray": (None, [VP, u32, u32, u32, u32, P, P]),
        "orc_ao_ray": (i, [SP, VP, u32, u32, u32, u32, HitC, u32, f, P, P]),
        "orc_uhash": (u32, [u32, u32]),
        "orc_hash_noise": (f, [u32, u32, u32]),
        "orc_sincos": (None, [f, P, P]),
        "orc_sincos_libm_mismatches": (C.c_uint64, [u32, u32, u32]),
        "orc_traverse": (HitC, [SP, P, P, f, f, u32, STP]),
        "orc_trace_primary": (None, [SP, VP, u32, u32, u32, u32, u32, i, P, STP]),
        "orc_trace_ao": (None, [SP, VP, u32, u32, u32, u32, u32, u32, f, i, P, P, STP]),
        "orc_trace_rays": (None, [SP, P, u64, u32, i, P, STP]),
        "orc_traverse_inst": (HitC, [SP, P, P, f, f, u32, STP, P]),
        "orc_ao_ray_inst": (i, [SP, VP, u32, u32, u32, u32, HitC, u32, u32, f, P, P]),
        "orc_xform_point": (None, [P, P, P]),
        "orc_xform_dir": (None, [P, P, P]),
        "orc_trace_primary_inst": (None, [SP, VP, u32, u32, u32, u32, u32, i, P, P, STP]),
        "orc_trace_ao_inst": (None, [SP, VP, u32, u32, u32, u32, u32, u32, f, i, P, P, P, P, STP]),
        "orc_trace_rays_inst": (None, [SP, P, u64, u32, i, P, P, STP]),
        "orc_render_frame": (C.c_double, [SP, VP, u32, u32, u32, u32, f, i, P]),
        "orc_brute_rays": (None, [P, u64, P, u64, u32, i, P]),
        "orc_brute_primary": (None, [P, u64, VP, u32, u32, u32, i, P]),
        "orc_validate": (i, [SP, P, P, C.c_char_p, i]),
        "orc_count_primary_per_ray": (None, [SP, VP, u32, u32, u32, i, P, P]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_simd(on):
    """The node test over 8 children at once in AVX2 registers (bit-identical results; what cpu_baseline times).
    Returns whether it took effect (needs AVX2 + FMA)."""
    lib = load()
    lib.orc_set_simd(1 if on else 0)
    return bool(lib.orc_get_simd())


def set_ao_libm(on):
    """Measurement aid (oracle only): AO directions from libm sinf / cosf (True or 1) or from a correctly rounded sin /
    cos (2) instead of the explicit evaluation."""
    load().orc_set_ao_libm(int(on))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def view_from_camera(eye, look_at, fov_deg, width, height):
    v = View()
    e = np.asarray(eye, dtype=np.float32)
    la = np.asarray(look_at, dtype=np.float32)
    load().orc_view_from_camera(_ptr(e), _ptr(la), fov_deg, float(width), float(height), C.byref(v))
    return v


def view_from_bytes(raw):
    """Reinterpret any 160-byte ViewUniform (e.g. the product's trx_view) as the oracle's view."""
    v = View()
    C.memmove(C.byref(v), bytes(raw), C.sizeof(View))
    return v


class Scene:
    """Oracle scene over the flat buffers (nodes [n,20] u32, tri_verts [n,9] f32)."""

    def __init__(self, nodes, tri_verts=None, instance_offsets=None, tlas_start=0, tri_f16=None, instance_w2o=None,
                 instance_entry=None):
        lib = load()
        self.nodes = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1, 20)
        if tri_f16 is not None:
            raw = np.ascontiguousarray(tri_f16, dtype=np.uint32).reshape(-1, 6)
            self.verts = None
            self.tris = np.empty((raw.shape[0], 9), dtype=np.float32)
            lib.orc_tris_from_f16(_ptr(raw), raw.shape[0], _ptr(self.tris))
        else:
            self.verts = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
            self.tris = np.empty_like(self.verts)
            lib.orc_tris_from_verts(_ptr(self.verts), self.verts.shape[0], _ptr(self.tris))
        self.inst = np.ascontiguousarray(instance_offsets if instance_offsets is not None else [], dtype=np.uint32)
        # world-to-object 3x4 (row-major, 12 floats) per TLAS primitive, or None = identity
        self.w2o = None if instance_w2o is None else np.ascontiguousarray(instance_w2o, dtype=np.float32).reshape(-1, 12)
        assert self.w2o is None or self.w2o.shape[0] == self.inst.size
        # entry node of every TLAS primitive inside its BLAS (re-braided scenes), or None = node 0
        self.entry = None if instance_entry is None else np.ascontiguousarray(instance_entry, dtype=np.uint32)
        assert self.entry is None or self.entry.size == self.inst.size
        self.c = SceneC(_ptr(self.nodes), self.nodes.shape[0], _ptr(self.tris), self.tris.shape[0],
                        _ptr(self.inst) if self.inst.size else None, self.inst.size, int(tlas_start),
                        _ptr(self.w2o) if self.w2o is not None else None,
                        _ptr(self.entry) if self.entry is not None else None)

    @classmethod
    def from_flat(cls, flat):
        return cls(flat.nodes, flat.tri_verts, flat.instance_offsets, flat.tlas_start,
                   instance_entry=getattr(flat, "instance_entry", None))

    def trace_primary(self, view, w, h, sem=SEM_HLSL, shard=(0, 1), threads=0, out=None):
        hits = out if out is not None else np.zeros(w * h, dtype=HIT_DTYPE)
        st = Stats()
        load().orc_trace_primary(C.byref(self.c), C.byref(view), w, h, shard[0], shard[1], sem, threads, _ptr(hits),
                                 C.byref(st))
        return hits, st

    def trace_ao(self, view, w, h, primary, sem=SEM_HLSL, frame=0, ao_eps=0.01, shard=(0, 1), threads=0):
        primary = np.ascontiguousarray(primary, dtype=HIT_DTYPE)
        ao = np.zeros(w * h, dtype=HIT_DTYPE)
        st = Stats()
        load().orc_trace_ao(C.byref(self.c), C.byref(view), w, h, shard[0], shard[1], sem, frame, ao_eps, threads,
                            _ptr(primary), _ptr(ao), C.byref(st))
        return ao, st

    def trace_rays(self, rays, sem=SEM_HLSL, threads=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        st = Stats()
        load().orc_trace_rays(C.byref(self.c), _ptr(rays), rays.shape[0], sem, threads, _ptr(hits), C.byref(st))
        return hits, st

    def trace_primary_inst(self, view, w, h, sem=SEM_HLSL, shard=(0, 1), threads=0):
        """(hits, instance ids, stats): instance = TLAS primitive index the hit was found in (0xFFFFFFFF = none)."""
        hits = np.zeros(w * h, dtype=HIT_DTYPE)
        inst = np.full(w * h, 0xFFFFFFFF, dtype=np.uint32)
        st = Stats()
        load().orc_trace_primary_inst(C.byref(self.c), C.byref(view), w, h, shard[0], shard[1], sem, threads, _ptr(hits),
                                      _ptr(inst), C.byref(st))
        return hits, inst, st

    def trace_ao_inst(self, view, w, h, primary, primary_inst, sem=SEM_HLSL, frame=0, ao_eps=0.01, shard=(0, 1), threads=0):
        primary = np.ascontiguousarray(primary, dtype=HIT_DTYPE)
        primary_inst = np.ascontiguousarray(primary_inst, dtype=np.uint32)
        ao = np.zeros(w * h, dtype=HIT_DTYPE)
        inst = np.full(w * h, 0xFFFFFFFF, dtype=np.uint32)
        st = Stats()
        load().orc_trace_ao_inst(C.byref(self.c), C.byref(view), w, h, shard[0], shard[1], sem, frame, ao_eps, threads,
                                 _ptr(primary), _ptr(primary_inst), _ptr(ao), _ptr(inst), C.byref(st))
        return ao, inst, st

    def trace_rays_inst(self, rays, sem=SEM_HLSL, threads=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        inst = np.full(rays.shape[0], 0xFFFFFFFF, dtype=np.uint32)
        st = Stats()
        load().orc_trace_rays_inst(C.byref(self.c), _ptr(rays), rays.shape[0], sem, threads, _ptr(hits), _ptr(inst),
                                   C.byref(st))
        return hits, inst, st

    def count_per_ray(self, view, w, h, sem=SEM_HLSL, threads=0):
        nn = np.zeros(w * h, dtype=np.uint16)
        nt = np.zeros(w * h, dtype=np.uint16)
        load().orc_count_primary_per_ray(C.byref(self.c), C.byref(view), w, h, sem, threads, _ptr(nn), _ptr(nt))
        return nn, nt

    def render_frame(self, view, w, h, sem=SEM_HLSL, frame=0, ao_eps=0.01, threads=0):
        return load().orc_render_frame(C.byref(self.c), C.byref(view), w, h, sem, frame, ao_eps, threads, None)

    def brute_primary(self, view, w, h, sem=SEM_HLSL, threads=0):
        hits = np.zeros(w * h, dtype=HIT_DTYPE)
        load().orc_brute_primary(_ptr(self.tris), self.tris.shape[0], C.byref(view), w, h, sem, threads, _ptr(hits))
        return hits

    def brute_rays(self, rays, sem=SEM_HLSL, threads=0):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        load().orc_brute_rays(_ptr(self.tris), self.tris.shape[0], _ptr(rays), rays.shape[0], sem, threads,
                              _ptr(hits))
        return hits

    def brute_rays_over(self, tri_verts, rays, sem=SEM_HLSL, threads=0):
        """Brute force over an arbitrary triangle list (n x 9 vertices), e.g. world-space copies of instances."""
        lib = load()
        v = np.ascontiguousarray(tri_verts, dtype=np.float32).reshape(-1, 9)
        tris = np.empty_like(v)
        lib.orc_tris_from_verts(_ptr(v), v.shape[0], _ptr(tris))
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        lib.orc_brute_rays(_ptr(tris), tris.shape[0], _ptr(rays), rays.shape[0], sem, threads, _ptr(hits))
        return hits

    def primary_rays(self, view, w, h):
        rays = np.zeros(w * h, dtype=RAY_DTYPE)
        o = np.zeros(3, dtype=np.float32)
        d = np.zeros(3, dtype=np.float32)
        lib = load()
        for i in range(w * h):
            lib.orc_primary_ray(C.byref(view), w, h, i % w, i // w, _ptr(o), _ptr(d))
            rays["origin"][i] = o
            rays["direction"][i] = d
        rays["tmin"] = 0.0
        rays["tmax"] = F32_MAX
        return rays

    def tri_t(self, ray_o, ray_d, prim, sem=SEM_HLSL):
        """t of one triangle test against (o, d) with the zero-direction fix applied, or None."""
        o = np.asarray(ray_o, dtype=np.float32)
        d = np.asarray(ray_d, dtype=np.float32).copy()
        d[d == 0.0] = np.float32(1.1920929e-7)
        t = np.array([F32_MAX], dtype=np.float32)
        tri = np.ascontiguousarray(self.tris[prim])
        ok = load().orc_intersect_tri(_ptr(o), _ptr(d), _ptr(tri), 0.0, _ptr(t), sem)
        return float(t[0]) if ok else None

    def validate(self, boxes=None):
        """boxes: FlatScene.tri_boxes of a pre-split build (entries cover only part of their triangle)."""
        if self.verts is None:
            raise ValueError("validate needs vertex-format triangles")
        err = C.create_string_buffer(256)
        if boxes is not None:
            boxes = np.ascontiguousarray(boxes, dtype=np.float32)
        rc = load().orc_validate(C.byref(self.c), _ptr(self.verts), _ptr(boxes) if boxes is not None else None, err, 256)
        return rc, err.value.decode()
