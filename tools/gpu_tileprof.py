"""Per-tile cost vs iteration counts: how long does a wave-level iteration take? (development aid)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
n = ((w + 7) // 8) * ((h + 7) // 8)
cost = np.zeros(n, dtype=np.uint32)
iters = np.zeros(n, dtype=np.uint32)
L.check(lib.trx_debug_tile_profile(sc.handle, C.byref(view), w, h, 3, cost.ctypes.data_as(C.c_void_p),
                                   iters.ctypes.data_as(C.c_void_p), n))
us = cost / 100.0
nn = (iters >> 16).astype(np.float64)
nt = (iters & 0xFFFF).astype(np.float64)
print("%s tiles %d | tile us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | sum %.0f us (/4096 waves = %.1f us)" % (
    name, n, us.mean(), np.percentile(us, 50), np.percentile(us, 90), np.percentile(us, 99), us.max(), us.sum(),
    us.sum() / 4096))
print("node iters: mean %.1f max %d | tri rounds: mean %.1f max %d" % (nn.mean(), nn.max(), nt.mean(), nt.max()))
A = np.stack([nn, nt, np.ones(n)], 1)
coef, *_ = np.linalg.lstsq(A, us, rcond=None)
print("least squares: tile_us = %.3f * node_iters + %.3f * tri_rounds + %.2f" % tuple(coef))
heavy = np.argsort(us)[-8:]
for i in heavy[::-1]:
    print("  tile %5d (x %3d y %3d): %.1f us, %d node iters, %d tri rounds -> %.2f us/iter" % (
        i, i % ((w + 7) // 8), i // ((w + 7) // 8), us[i], nn[i], nt[i], us[i] / max(nn[i] + nt[i], 1)))
sc.close()
