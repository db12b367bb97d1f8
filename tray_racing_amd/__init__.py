"""tray_racing_amd — MI355X (gfx950) backend for tray_racing's CWBVH traversal hot path.

`csrc/` holds the HIP kernels and the C-ABI (include/trx.h); `host.py` mirrors
the reference's `cwbvh_gpu_runner` / `rt_gpu_software::start` interface over it.
"""
from ._lib import (SEM_CPU, SEM_HLSL, SEM_NODE_FMA, SEM_NODE_RCP, SEM_TIE_FIRST, TRI_EDGES_36, TRI_F16_24,  # noqa: F401
                   TRI_VERTS_36, Hit, Ray, RayHit, Shard, Stats, TrxError, View, load)
from .host import (HIT_DTYPE, MISS_PRIM, RAY_DTYPE, RAYHIT_DTYPE, FlatScene, Scene, cwbvh_gpu_runner, flat_build, flat_build_instanced, flat_build_params, flat_build_preset_device, build_params, gen_scene,  # noqa: F401
                   load_meshs, load_scene, copy_rate, pack_tris_f16, scene_camera, view_from_camera)
