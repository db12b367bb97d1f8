"""Shape of the triangle phases of a primary frame (development aid)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
w, h = 1920, 1080
for name in sys.argv[1:] or ["bistro"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    hist = np.zeros(32, dtype=np.uint32)
    L.check(lib.trx_debug_tri_histogram(sc.handle, C.byref(view), w, h, 3, hist.ctypes.data_as(C.c_void_p)))
    st = sc.count_primary(view, w, h, sem=3)
    n = hist[:16].sum()
    print("%s: %d triangle phases, %d wave-level node steps, %d triangle rounds, %.2f tris/ray" % (
        name, n, st.n_wave_node, st.n_wave_tri, st.n_tri / st.n_rays))
    print("  max per-lane count:  " + " ".join("%d:%.1f%%" % (i, 100.0 * hist[i] / n) for i in range(16) if hist[i]))
    print("  pair total (x8, up): " + " ".join("%d:%.1f%%" % (8 * i, 100.0 * hist[16 + i] / n) for i in range(16) if hist[16 + i]))
    # rounds needed by three schemes
    mx = np.arange(16)
    per_lane = (hist[:16] * mx).sum()
    per_lane_pk = (hist[:16] * ((mx + 1) // 2)).sum()
    coop = (hist[16:] * np.maximum(1, (np.arange(16) + 7) // 8)).sum()
    print("  rounds: per-lane %d, per-lane two-at-once %d, cooperative (64 pairs per round) ~%d" % (per_lane, per_lane_pk, coop))
    sc.close()
