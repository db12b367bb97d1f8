"""Children left by the packet test of a wave-uniform node step (development aid; round 5).
Needs a development build: make -C tray_racing_amd/csrc KFLAGS=-DTRX_DEV_TUNE OUT=$PWD/tuning_libs/dev.so BUILD=/tmp/build_dev
usage: TRX_LIB=tuning_libs/dev.so TRX_TUNE=0x100 TRX_HIST_NORMAL=1 python tools/gpu_cullhist.py [scene ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
w, h = 1920, 1080
for name in sys.argv[1:] or ["bistro", "kitchen", "hairball", "bistro_dense"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    hist = np.zeros(32, dtype=np.uint32)
    L.check(lib.trx_debug_tri_histogram(sc.handle, C.byref(view), w, h, 3, hist.ctypes.data_as(C.c_void_p)))
    k = hist[16:26].astype(np.float64)
    os.environ.pop("TRX_HIST_NORMAL")
    st = sc.count_primary(view, w, h, sem=3)
    os.environ["TRX_HIST_NORMAL"] = "1"
    n = k.sum()
    print("%s: %d wave-level node steps, %d wave-uniform (%.1f %%); children left by the packet test:" % (name, st.n_wave_node, n, 100.0 * n / max(st.n_wave_node, 1)))
    print("   " + "  ".join("%d: %.1f%%" % (i, 100.0 * k[i] / max(n, 1)) for i in range(10)) + "   (9 = no packet test)   mean %.2f" % ((k[:9] * np.arange(9)).sum() / max(k[:9].sum(), 1)))
    sc.close()
