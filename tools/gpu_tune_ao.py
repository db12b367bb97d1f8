"""[needs a development build: make -C tray_racing_amd/csrc KFLAGS=-DTRX_DEV_TUNE OUT=... and TRX_LIB pointing at it]
Times the AO pass and an explicit-ray pass for a list of TRX_TUNE development words / kernel-variant words
(tune:variant) on scenes (development aid).  usage: python tools/gpu_tune_ao.py bistro,hairball 0 1 0:0x14"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402
from tools.prof_config import hemisphere_rays  # noqa: E402

import torch  # noqa: E402

lib = L.load()
names = sys.argv[1].split(",")
words = [tuple(int(y, 0) for y in (x.split(":") + ["0"])[:2]) for x in sys.argv[2:]]
w, h = 1920, 1080
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=8, warm=3):
    best = 1e9
    for k in range(reps + warm):
        e0.record()
        fn(k)
        e1.record()
        torch.cuda.synchronize()
        if k >= warm:
            best = min(best, e0.elapsed_time(e1))
    return best


for name in names:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts, use_tlas=bool(int(os.environ.get("TLAS", "0"))))  # TLAS=1: the two-level kernels
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    d_prim = torch.empty(w * h, dtype=torch.int64, device="cuda")
    d_ao = torch.empty(w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, d_prim.data_ptr(), sem=3)
    sc.check()
    n = w * h
    rays = hemisphere_rays(flat, None, eye, n, 5)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.empty(n, dtype=torch.int64, device="cuda")
    ref_ao = ref_rays = None
    for rep in range(2):
        for tune, variant in words:
            os.environ["TRX_TUNE"] = str(tune)
            lib.trx_set_kernel_variant(variant)
            vary = int(os.environ.get("AO_FRAME_VARIES", "0"))  # 1: a new noise seed every pass, as a renderer would
            t_ao = timed(lambda k: sc.trace_ao_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=3, frame=(k if vary else 0), ao_eps=0.01))
            if vary:
                sc.trace_ao_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=3, frame=0, ao_eps=0.01)
            t_rays = timed(lambda k: sc.trace_rays_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), sem=3))
            sc.check()
            a, r = d_ao.cpu().numpy(), d_hits.cpu().numpy()
            if ref_ao is None:
                ref_ao, ref_rays = a.copy(), r.copy()
            print("%s tune 0x%x variant 0x%08x: AO %.3f ms  rays %.3f ms  same=%s" % (
                name, tune, variant, t_ao, t_rays, bool((a == ref_ao).all() and (r == ref_rays).all())), flush=True)
    os.environ["TRX_TUNE"] = "0"
    lib.trx_set_kernel_variant(0)
    sc.close()
