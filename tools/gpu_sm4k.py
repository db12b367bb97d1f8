"""san-miguel-class 3840x2160 two-level primary + AO frame, a few repetitions (TRX_LIB=<library> to compare builds in separate processes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import tray_racing_amd as T
lib = T.load()
w, h = 3840, 2160
verts, counts = T.gen_scene("san_miguel", 0, 1)
flat = T.flat_build(verts, counts, use_tlas=True)
eye, look, fov = T.scene_camera("san_miguel")
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
for rep in range(2):
    for v in (0,):
        lib.trx_set_kernel_variant(v)
        ts = []
        for f in range(10):
            prim, ao, ms = sc.trace_primary_ao(view, w, h, sem=3, frame=f % 4, ao_eps=0.01)
            ts.append(ms)
        print("variant 0x%x: primary+AO min %.3f mean(last 6) %.3f ms" % (v, min(ts[3:]), np.mean(ts[4:])), flush=True)
