"""[development build in TRX_LIB: make KFLAGS=-DTRX_DEV_TUNE OUT=...]  Primary / AO / random-ray passes and the AO
pass's wave timeline per scene for a list of tune:variant words.  A word whose tune has bit 0x2000 (node stride 128)
re-creates the scene, since that switch is read at upload.
usage: python tools/gpu_tune3.py bistro,hairball 0 0x1000 0x2000 0x3000 0:0x20000000"""
import ctypes as C
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
words = [tuple(int(y, 0) for y in (x.split(":") + ["0"])[:2]) for x in sys.argv[2:]] or [(0, 0)]
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


def ao_timeline(sc, view):
    buf = np.zeros(8 * 8192, dtype=np.uint64)
    n = C.c_uint32()
    for _ in range(2):
        L.check(lib.trx_debug_wave_timeline_ao(sc.handle, C.byref(view), w, h, 3, buf.ctypes.data_as(C.c_void_p), 8192, C.byref(n)))
    t = buf[: 8 * n.value].reshape(-1, 8)[:, :2].astype(np.int64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
    total = end.max()
    ts = np.linspace(0, total, 11)
    return "pass %.0f us, wave end p50 %.0f p90 %.0f p99 %.0f us, alive at 0..100%%: %s" % (
        total, np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99),
        [int(((start <= x) & (end > x)).sum()) for x in ts])


for name in names:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    n = w * h
    rays = hemisphere_rays(flat, None, eye, n, 5)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.empty(n, dtype=torch.int64, device="cuda")
    d_prim = torch.empty(w * h, dtype=torch.int64, device="cuda")
    d_ao = torch.empty(w * h, dtype=torch.int64, device="cuda")
    sc, sc_stride = None, None
    for tune, variant in words:
        os.environ["TRX_TUNE"] = str(tune)
        stride = tune & 0x2000
        if sc is None or stride != sc_stride:
            if sc is not None:
                sc.close()
            sc = T.Scene(flat)
            sc_stride = stride
        lib.trx_set_kernel_variant(variant)
        p_min, p_mean = sc.bench_primary(view, w, h, sem=3, warmup=10, frames=30)
        sc.trace_primary_dev(view, w, h, d_prim.data_ptr(), sem=3)
        sc.check()
        n_ao = int((d_prim.cpu().numpy().view(T.HIT_DTYPE)["prim"] != 0xFFFFFFFF).sum())
        a_min, a_mean = timed(lambda k: sc.trace_ao_dev(view, w, h, d_prim.data_ptr(), d_ao.data_ptr(), sem=3, frame=0, ao_eps=0.01))
        r_min, r_mean = timed(lambda k: sc.trace_rays_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), sem=3))
        sc.check()
        crc = zlib.crc32(d_hits.cpu().numpy().tobytes(), zlib.crc32(d_ao.cpu().numpy().tobytes(), zlib.crc32(d_prim.cpu().numpy().tobytes())))
        print("%-12s tune 0x%04x variant 0x%08x: primary %.4f / %.4f | AO %.4f / %.4f (%.0f Mrays/s) | rays %.4f / %.4f | crc %08x" % (
            name, tune, variant, p_min, p_mean, a_min, a_mean, n_ao / a_min / 1e3, r_min, r_mean, crc), flush=True)
        print("%-12s    AO timeline: %s" % (name, ao_timeline(sc, view)), flush=True)
    os.environ["TRX_TUNE"] = "0"
    lib.trx_set_kernel_variant(0)
    sc.close()
