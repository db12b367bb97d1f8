"""Where a persistent wave's cycles go in the AO pass (development aid; needs a -DTRX_STAMPS build in TRX_LIB).
usage: python tools/gpu_phases_ao.py hairball,bistro [tune ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
w, h = 1920, 1080
for name in sys.argv[1].split(","):
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    for tune in [int(x, 0) for x in sys.argv[2:]] or [0]:
        os.environ["TRX_TUNE"] = str(tune)
        buf = np.zeros(8 * 8192, dtype=np.uint64)
        n = C.c_uint32()
        for _ in range(3):
            L.check(lib.trx_debug_wave_timeline_ao(sc.handle, C.byref(view), w, h, 3, buf.ctypes.data_as(C.c_void_p), 8192, C.byref(n)))
        r = buf[: 8 * n.value].reshape(-1, 8).astype(np.float64)
        life_us = (r[:, 1] - r[:, 0]) / 100.0
        t0 = r[:, 0].min()
        names = ["refill", "node fetch", "node test", "triangle phase", "pop/bookkeeping"]
        cyc = r[:, 2:7]
        tot = cyc.sum()
        iters = r[:, 7].sum()
        print("%s tune 0x%x (stamped build): pass %.0f us, waves %d, mean lifetime %.1f us, loop trips per wave %.0f (max %.0f), cycles per trip %.0f" % (
            name, tune, (r[:, 1].max() - t0) / 100.0, n.value, life_us.mean(), iters / n.value, r[:, 7].max(), tot / max(iters, 1)))
        for i, nm in enumerate(names):
            print("  %-16s %5.1f %%   %7.0f cycles per trip" % (nm, 100 * cyc[:, i].sum() / tot, cyc[:, i].sum() / max(iters, 1)))
        # the waves that set the tail: the 64 that end last
        last = np.argsort(r[:, 1])[-64:]
        c2 = r[last, 2:7]
        it2 = r[last, 7].sum()
        print("  last 64 waves: trips per wave %.0f, cycles per trip %.0f: %s" % (
            it2 / 64, c2.sum() / max(it2, 1), ", ".join("%s %.0f" % (nm, c2[:, i].sum() / max(it2, 1)) for i, nm in enumerate(names))))
    os.environ["TRX_TUNE"] = "0"
    sc.close()
