"""Incoherent passes (development aid): AO pass and a batch of random rays per scene, for a list of kernel-variant
words (e.g. 0x400 / 0x800 / 0xc00 / 0x1000 = 4 / 8 / 12 / 16 resident waves per CU), with a checksum of the hit
buffers so that variants can be compared for identical results.
usage: python tools/gpu_incoherent.py bistro,hairball 0 0x400 0x800 0xc00"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402
from tools.prof_config import hemisphere_rays  # noqa: E402

import torch  # noqa: E402

lib = L.load()
names = sys.argv[1].split(",")
words = [int(x, 0) for x in sys.argv[2:]] or [0]
w, h = 1920, 1080
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=8, warm=3):
    ts = []
    for k in range(reps + warm):
        e0.record()
        fn(k)
        e1.record()
        torch.cuda.synchronize()
        if k >= warm:
            ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)


for name in names:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    d_prim = torch.empty(w * h, dtype=torch.int64, device="cuda")
    d_ao = torch.empty(w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, d_prim.data_ptr(), sem=3)
    sc.check()
    n_ao = int((d_prim.cpu().numpy().view(T.HIT_DTYPE)["prim"] != 0xFFFFFFFF).sum())
    n = w * h
    rays = hemisphere_rays(flat, None, eye, n, 5)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.empty(n, dtype=torch.int64, device="cuda")
    for variant in words:
        lib.trx_set_kernel_variant(variant)
        ao_min, ao_mean = timed(lambda k: sc.trace_ao_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=3, frame=0, ao_eps=0.01))
        r_min, r_mean = timed(lambda k: sc.trace_rays_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), sem=3))
        sc.check()
        crc = zlib.crc32(d_hits.cpu().numpy().tobytes(), zlib.crc32(d_ao.cpu().numpy().tobytes()))
        print("%-12s variant 0x%08x: AO min %.4f mean %.4f ms (%d rays, %.0f Mrays/s) | random rays min %.4f mean %.4f ms (%.0f Mrays/s) | crc %08x" % (
            name, variant, ao_min, ao_mean, n_ao, n_ao / ao_min / 1e3, r_min, r_mean, n / r_min / 1e3, crc), flush=True)
    lib.trx_set_kernel_variant(0)
    sc.close()
