"""[needs a development build: make -C tray_racing_amd/csrc KFLAGS=-DTRX_DEV_TUNE OUT=... and TRX_LIB pointing at it]
Distinct nodes per wave-level node step (development aid; TRX_TUNE bit 8 re-purposes the pair-total histogram)."""
import ctypes as C
import os
import sys

import numpy as np

os.environ["TRX_TUNE"] = "0x100"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
w, h = (int(v) for v in os.environ.get("WH", "1920x1080").split("x"))
for name in sys.argv[1:] or ["bistro", "bistro_dense", "hairball", "kitchen"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    hist = np.zeros(32, dtype=np.uint32)
    L.check(lib.trx_debug_tri_histogram(sc.handle, C.byref(view), w, h, 3, hist.ctypes.data_as(C.c_void_p)))
    d = hist[16:].astype(np.float64)
    print("%s: %d wave-level node steps; distinct nodes per step: " % (name, d.sum()) +
          " ".join("%d:%.1f%%" % (i, 100 * d[i] / d.sum()) for i in range(16) if d[i]), flush=True)
    sc.close()
