"""Where a persistent wave's cycles go (development aid; needs a -DTRX_STAMPS build in TRX_LIB)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
lib.trx_set_kernel_variant(int(os.environ.get("TRX_VARIANT", "0"), 0))
name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=10, frames=20)
print("%s: min %.3f ms mean %.3f ms (stamped build: slower than the product)" % (name, mn, mean))
buf = np.zeros(8 * 8192, dtype=np.uint64)
n = C.c_uint32()
for _ in range(6):
    L.check(lib.trx_debug_wave_phases(sc.handle, C.byref(view), w, h, 3, buf.ctypes.data_as(C.c_void_p), 8192, C.byref(n)))
r = buf[: 8 * n.value].reshape(-1, 8).astype(np.float64)
life_us = (r[:, 1] - r[:, 0]) / 100.0
names = ["refill", "node fetch", "node test", "triangle phase", "pop/bookkeeping"]
cyc = r[:, 2:7]
tot = cyc.sum()
iters = r[:, 7].sum()
print("waves %d, mean lifetime %.1f us, loop trips per wave %.0f, cycles per trip %.0f (clock %.2f GHz by lifetime)" % (
    n.value, life_us.mean(), iters / n.value, tot / max(iters, 1), cyc.sum(axis=1).mean() / life_us.mean() / 1e3))
for i, nm in enumerate(names):
    print("  %-16s %5.1f %%   %7.0f cycles per trip" % (nm, 100 * cyc[:, i].sum() / tot, cyc[:, i].sum() / max(iters, 1)))
sc.close()
