"""Profiling target for the rocprofv3 passes of profiles/: replays one BASELINE config's kernel a few times.
usage (after `--` of rocprofv3): python3 tools/prof_config.py <config> [frames]
configs: primary_bistro, primary_bistro_dense, ao_bistro, ao_hairball, ao4_hairball, primary_hairball, tlas_san_miguel_4k, rays_bistro"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

CONFIGS = {
    # name: (scene, mode, width, height, tlas)
    "primary_bistro": ("bistro", "primary", 1920, 1080, False),
    "primary_bistro_dense": ("bistro_dense", "primary", 1920, 1080, False),
    "primary_kitchen": ("kitchen", "primary", 1920, 1080, False),
    "primary_hairball": ("hairball", "primary", 1920, 1080, False),
    "ao_bistro": ("bistro", "ao", 1920, 1080, False),
    "ao_hairball": ("hairball", "ao", 1920, 1080, False),
    "ao4_hairball": ("hairball", "ao4", 1920, 1080, False),   # BASELINE configs[3]: 4 spp = AO frames 0..3 in ONE launch
    "tlas_san_miguel_4k": ("san_miguel", "primary", 3840, 2160, True),
    "rays_bistro": ("bistro", "rays", 1920, 1080, False),
}


def hemisphere_rays(flat, hits, view_eye, n, seed):
    """n uniformly random directions from points spread over the scene's bounding box (incoherent by construction)."""
    rng = np.random.default_rng(seed)
    v = flat.tri_verts.reshape(-1, 3)
    lo, hi = v.min(axis=0), v.max(axis=0)
    rays = np.zeros(n, dtype=T.RAY_DTYPE)
    rays["origin"] = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["tmin"] = 0.0
    rays["tmax"] = 3.4028234663852886e38
    return rays


def main():
    import torch
    cfg = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    scene_name, mode, w, h, tlas = CONFIGS[cfg]
    verts, counts = T.gen_scene(scene_name, 0, 1)
    flat = T.flat_build(verts, counts, use_tlas=tlas)
    eye, look, fov = T.scene_camera(scene_name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    st = sc.count_primary(view, w, h, sem=3)
    import hashlib
    info = {"config": cfg, "scene": scene_name, "mode": mode, "width": w, "height": h, "tlas": tlas,
            "tris": int(flat.n_tris), "nodes": int(flat.n_nodes),
            # the library that ran (tools/profile_summary.py refuses to mix libraries in one round's profile set)
            "lib_sha16": hashlib.sha256(open(L.LIB_PATH, "rb").read()).hexdigest()[:16]}
    if mode == "primary":
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=3, frames=frames)
        info.update(rays=w * h, n_node=int(st.n_node), n_tri=int(st.n_tri), ms_min=mn, ms_mean=mean)
    else:
        d_prim = torch.empty(w * h, dtype=torch.int64, device="cuda")
        sc.trace_primary_dev(view, w, h, d_prim.data_ptr(), sem=3)
        sc.check()
        if mode in ("ao", "ao4"):
            spp = 4 if mode == "ao4" else 1
            d_ao = torch.empty(spp * w * h, dtype=torch.int64, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for f in range(frames + 3):
                e0.record()
                if spp == 1:
                    sc.trace_ao_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=3, frame=f % 4, ao_eps=0.01)
                else:
                    sc.trace_ao_batch_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), w * h, spp, sem=3, frame0=spp * f, ao_eps=0.01)
                e1.record()
                torch.cuda.synchronize()
                if f >= 3:
                    best = min(best, e0.elapsed_time(e1))
            sc.check()
            hits = int((d_prim.cpu().numpy().view(T.HIT_DTYPE)["prim"] != 0xFFFFFFFF).sum())
            ast = L.Stats()
            L.check(L.load().trx_count_ao(sc.handle, C.byref(view), w, h, L.Shard(0, 1, 0, 0), 3, 0, 0.01,
                                          C.c_void_p(d_prim.data_ptr()), C.c_void_p(d_ao.data_ptr()), C.byref(ast)))
            # (the counts are those of ONE AO frame, seed 0; a batch of spp frames asks for about spp times as much)
            info.update(rays=hits * spp, n_node=int(ast.n_node) * spp, n_tri=int(ast.n_tri) * spp, ms_min=best, spp=spp)
        else:
            n = w * h
            rays = hemisphere_rays(flat, None, eye, n, 5)
            d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
            d_hits = torch.empty(n, dtype=torch.int64, device="cuda")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for f in range(frames + 3):
                e0.record()
                sc.trace_rays_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), sem=3)
                e1.record()
                torch.cuda.synchronize()
                if f >= 3:
                    best = min(best, e0.elapsed_time(e1))
            sc.check()
            rst = L.Stats()
            L.check(L.load().trx_count_rays(sc.handle, C.c_void_p(d_rays.data_ptr()), n, 3, C.c_void_p(d_hits.data_ptr()),
                                            C.byref(rst)))
            info.update(rays=n, n_node=int(rst.n_node), n_tri=int(rst.n_tri), ms_min=best, ray_bytes=32)
    print("PROF_CONFIG " + repr(info), flush=True)
    sc.close()


if __name__ == "__main__":
    main()
