"""ctypes binding of libtrx.so (the C-ABI of include/trx.h).

Nothing in this package computes a traversal on the CPU: every trace call goes
through the HIP kernels in libtrx.so and raises TrxError when the library or a
gfx950 device is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRX_LIB") or os.path.join(_HERE, "libtrx.so")  # TRX_LIB: tuning builds

TRX_OK = 0
TRX_ERR_INVALID = -1
TRX_ERR_NO_DEVICE = -2
TRX_ERR_OOM = -3
TRX_ERR_STACK_OVERFLOW = -4
TRX_ERR_FORMAT = -5
TRX_ERR_IO = -6

TRI_F16_24 = 0
TRI_VERTS_36 = 1
TRI_EDGES_36 = 2

SEM_HLSL = 0
SEM_NODE_RCP = 1
SEM_TIE_FIRST = 2
SEM_NODE_FMA = 4
SEM_CPU = 3


class TrxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("trx error %d: %s" % (code, message))
        self.code = code


class View(C.Structure):
    _fields_ = [("view_inv", C.c_float * 16), ("proj_inv", C.c_float * 16), ("eye", C.c_float * 3),
                ("exposure", C.c_float), ("tlas_start", C.c_uint32), ("_pad", C.c_uint32 * 3)]


class Ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("tmin", C.c_float), ("direction", C.c_float * 3), ("tmax", C.c_float)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("prim", C.c_uint32)]


class RayHit(C.Structure):
    _fields_ = [("primitive_id", C.c_uint32), ("geometry_id", C.c_uint32), ("instance_id", C.c_uint32),
                ("t", C.c_float)]


class Shard(C.Structure):
    _fields_ = [("index", C.c_uint32), ("count", C.c_uint32), ("layout", C.c_uint32), ("_pad", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("n_rays", C.c_uint64), ("n_node", C.c_uint64), ("n_tri", C.c_uint64), ("n_hits", C.c_uint64),
                ("max_stack", C.c_uint32), ("overflow", C.c_uint32), ("kernel_ms", C.c_float), ("_pad", C.c_float),
                ("n_wave_node", C.c_uint64), ("n_wave_tri", C.c_uint64)]


class Flat(C.Structure):
    _fields_ = [("bvh_bytes", C.c_void_p), ("n_nodes", C.c_uint64), ("tri_verts", C.POINTER(C.c_float)),
                ("n_tris", C.c_uint64), ("instance_offsets", C.POINTER(C.c_uint32)), ("n_instances", C.c_uint32),
                ("tlas_start", C.c_uint32), ("tri_source", C.POINTER(C.c_uint32)),
                ("blas_tri_start", C.POINTER(C.c_uint32)), ("n_blas", C.c_uint32), ("blas_build_s", C.c_double),
                ("tlas_build_s", C.c_double), ("tri_boxes", C.POINTER(C.c_float)),
                ("instance_source", C.POINTER(C.c_uint32)), ("instance_transforms", C.POINTER(C.c_float)),
                ("instance_entry_nodes", C.POINTER(C.c_uint32))]


class BuildParams(C.Structure):
    """trx_build_params = the reference's BvhBuildParams (src/main.rs:571-585)."""
    _fields_ = [("pre_split", C.c_uint32), ("ploc_search_distance", C.c_uint32), ("search_depth_threshold", C.c_uint32),
                ("reinsertion_batch_ratio", C.c_float), ("sort_precision", C.c_uint32), ("max_prims_per_leaf", C.c_uint32),
                ("post_collapse_reinsertion_batch_ratio_multiplier", C.c_float), ("collapse_traversal_cost", C.c_float)]


_P = C.c_void_p
_u32, _u64, _i, _f = C.c_uint32, C.c_uint64, C.c_int, C.c_float

# name -> (restype, argtypes); every symbol include/trx.h declares
SIGNATURES = {
    "trx_last_error": (C.c_char_p, []),
    "trx_abi_version": (_u32, []),
    "trx_device_count": (_i, []),
    "trx_device_name": (_i, [_i, C.c_char_p, C.c_size_t]),
    "trx_tri_format_bytes": (_u32, [_u32]),
    "trx_scene_create": (_i, [_P, _u64, _P, _u64, _u32, _P, _u32, _u32, _i, C.POINTER(_P)]),
    "trx_scene_destroy": (None, [_P]),
    "trx_scene_device_bytes": (_u64, [_P]),
    "trx_scene_device": (_i, [_P]),
    "trx_scene_set_geometry_ranges": (_i, [_P, _P, _u32]),
    "trx_scene_set_instance_transforms": (_i, [_P, _P, _u32]),
    "trx_scene_get_instance_transform": (_i, [_P, _u32, _P]),
    "trx_scene_get_instance_world_to_object": (_i, [_P, _u32, _P]),
    "trx_view_from_camera": (_i, [C.POINTER(_f), C.POINTER(_f), _f, _f, _f, C.POINTER(View)]),
    "trx_trace_primary_dev": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _P, _P]),
    "trx_trace_primary_inst_dev": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _P, _P, _P]),
    "trx_trace_primary_batch_dev": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, Shard, _u32, _P, _u64, _P]),
    "trx_trace_ao_dev": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _u32, _f, _P, _P, _P]),
    "trx_trace_ao_inst_dev": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _u32, _f, _P, _P, _P, _P, _P]),
    "trx_trace_frame_dev": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _u32, _f, _P, _P, _P, _P, _P]),
    "trx_trace_ao_batch_dev": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _u32, _u32, _f, _P, _P, _P, _P, _u64, _P]),
    "trx_trace_rays_dev": (_i, [_P, _P, _u64, _u32, _P, _P]),
    "trx_trace_rays_inst_dev": (_i, [_P, _P, _u64, _u32, _P, _P, _P]),
    "trx_trace_occluded_dev": (_i, [_P, _P, _u64, _u32, _P, _P]),
    "trx_trace_occluded": (_i, [_P, _P, _u64, _u32, _P, C.POINTER(_f)]),
    "trx_count_primary": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _P, C.POINTER(Stats)]),
    "trx_count_ao": (_i, [_P, C.POINTER(View), _u32, _u32, Shard, _u32, _u32, _f, _P, _P, C.POINTER(Stats)]),
    "trx_count_rays": (_i, [_P, _P, _u64, _u32, _P, C.POINTER(Stats)]),
    "trx_scene_check": (_i, [_P, _P]),
    "trx_trace_primary": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _P, C.POINTER(_f)]),
    "trx_trace_primary_ao": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _u32, _f, _P, _P, C.POINTER(_f)]),
    "trx_frame_loop": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _u32, _i, _f, _u32, _i, _P, _P, C.POINTER(_f)]),
    "trx_trace_primary_ao_inst": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _u32, _f, _P, _P, _P, _P, C.POINTER(_f)]),
    "trx_trace_rays": (_i, [_P, _P, _u64, _u32, _P, C.POINTER(_f)]),
    "trx_trace_rays_inst": (_i, [_P, _P, _u64, _u32, _P, _P, C.POINTER(_f)]),
    "trx_traverse1": (_i, [_P, C.POINTER(Ray), _u32, C.POINTER(RayHit)]),
    "trx_debug_traverse1_stats": (_i, [_P, C.POINTER(_u64), C.POINTER(_u64)]),
    "trx_debug_fetch_rate": (_i, [_P, _u32, _u32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "trx_debug_copy_rate": (_i, [_i, _u64, _u32, C.POINTER(C.c_double)]),
    "trx_debug_service_stats": (_i, [_P] + [C.POINTER(_u64)] * 5),
    "trx_debug_traverse1_threads": (_i, [_P, _P, _u64, _u32, _u32, _P, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "trx_traverse_batch": (_i, [_P, _P, _u64, _u32, _P, C.POINTER(_f)]),
    "trx_bench_primary": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _u32, _u32, C.POINTER(_f), C.POINTER(_f)]),
    "trx_set_kernel_variant": (_u32, [_u32]),
    "trx_debug_tile_profile": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _P, _P, _u32]),
    "trx_debug_wave_timeline": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _P, _u32, C.POINTER(_u32)]),
    "trx_debug_footprint": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "trx_debug_tri_histogram": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _P]),
    "trx_debug_wave_phases": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _P, _u32, C.POINTER(_u32)]),
    "trx_debug_wave_timeline_ao": (_i, [_P, C.POINTER(View), _u32, _u32, _u32, _P, _u32, C.POINTER(_u32)]),
    "trx_shard_tiles": (_u32, [_u32, _u32, Shard]),
    "trx_comm_unique_id": (_i, [_P]),
    "trx_comm_create": (_i, [_P, _i, _i, _i, C.POINTER(_P)]),
    "trx_comm_destroy": (None, [_P]),
    "trx_comm_world_size": (_i, [_P]),
    "trx_gather_shards": (_i, [_P, _P, _u64, _P]),
    "trx_gather_shards_root": (_i, [_P, _P, _u64, _i, _P]),
    "trx_assemble_frames": (_i, [_P, _u64, _u32, _u32, _u32, _u32, _P, _P]),
    "trx_bvh_build_tris": (_i, [_P, _u64, _u32, _i, C.POINTER(_P)]),
    "trx_bvh_build_aabbs": (_i, [_P, _u64, _u32, _i, C.POINTER(_P)]),
    "trx_set_build_costs": (_i, [_f, _f]),
    "trx_set_build_reinsertion": (_i, [_f, _i]),
    "trx_set_build_reinsertion_batches": (_i, [_i]),
    "trx_set_build_preset": (_i, [C.c_char_p]),
    "trx_set_build_split": (_i, [_f]),
    "trx_set_build_device": (_i, [_i]),
    "trx_set_build_rebraid": (_i, [_f]),
    "trx_scene_set_instance_entry_nodes": (_i, [_P, _P, _u32]),
    "trx_bvh_destroy": (None, [_P]),
    "trx_bvh_node_count": (_u64, [_P]),
    "trx_bvh_prim_count": (_u64, [_P]),
    "trx_bvh_nodes": (_P, [_P]),
    "trx_bvh_primitive_indices": (_P, [_P]),
    "trx_bvh_total_aabb": (None, [_P, C.POINTER(_f)]),
    "trx_bvh_build_seconds": (C.c_double, [_P]),
    "trx_flat_build": (_i, [_P, _P, _u32, _i, _u32, _i, C.POINTER(C.POINTER(Flat))]),
    "trx_flat_build_instanced": (_i, [_P, _P, _u32, _P, _P, _u32, _u32, _i, C.POINTER(C.POINTER(Flat))]),
    "trx_flat_build_params": (_i, [_P, _P, _u32, _i, C.POINTER(BuildParams), _i, C.POINTER(C.POINTER(Flat))]),
    "trx_flat_build_preset_device": (_i, [_P, _P, _u32, _i, C.c_char_p, _u32, _i, _i, C.POINTER(C.POINTER(Flat))]),
    "trx_build_params_default": (None, [C.POINTER(BuildParams)]),
    "trx_flat_destroy": (None, [C.POINTER(Flat)]),
    "trx_gen_scene": (_i, [C.c_char_p, _u64, _u64, C.POINTER(C.POINTER(_f)), C.POINTER(_u64),
                           C.POINTER(C.POINTER(_u64)), C.POINTER(_u32)]),
    "trx_scene_camera": (_i, [C.c_char_p, C.POINTER(_f), C.POINTER(_f), C.POINTER(_f)]),
    "trx_load_model": (_i, [C.c_char_p, C.POINTER(C.POINTER(_f)), C.POINTER(_u64), C.POINTER(C.POINTER(_u64)),
                            C.POINTER(_u32)]),
    "trx_load_scene": (_i, [C.c_char_p, C.POINTER(C.POINTER(_f)), C.POINTER(_u64), C.POINTER(C.POINTER(_u64)),
                        C.POINTER(_u32), C.POINTER(_f), C.POINTER(_f), C.POINTER(_f)]),
    "trx_free": (None, [_P]),
}

_lib = None


def load():
    """Loads libtrx.so (after torch, so both share one HIP runtime) and binds every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TrxError(TRX_ERR_NO_DEVICE, "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)" % LIB_PATH)
    try:
        # torch bundles its own libamdhip64.so.7; importing it first makes the
        # dynamic loader hand the same runtime to libtrx.so, so device pointers
        # and streams are interchangeable between the two.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement of the ABI
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            # (an OLDER build under TRX_LIB, for A/B runs: diagnostics added since are simply absent; the product library
            # exports every symbol - tests/test_abi.py)
            if name.startswith("trx_debug_") and os.environ.get("TRX_LIB"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != TRX_OK:
        msg = load().trx_last_error()
        raise TrxError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc
