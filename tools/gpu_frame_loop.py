"""The reference's frame loop - primary pass then AO pass on one queue (trx_trace_primary_ao) - over many frames:
mean / best frame time once the per-kind tile orders and the schedule tuner have settled.
TRX_LIB=<other libtrx.so> compares builds; SCENES=a,b picks scenes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

T.load().trx_set_kernel_variant(int(os.environ.get("TRX_VARIANT", "0"), 0))
FRAMES = int(os.environ.get("FRAMES", "120"))
for name in os.environ.get("SCENES", "kitchen,bistro,bistro_dense,hairball").split(","):
    w, h = (3840, 2160) if name == "san_miguel" else (1920, 1080)
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts, use_tlas=(name == "san_miguel"))
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    ms = []
    for f in range(FRAMES):
        prim, ao, t = sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)
        ms.append(t)
    ms = np.array(ms)
    print("%-12s primary+AO  first %.3f  frames 4..11 best %.3f  last half mean %.3f best %.3f ms" % (
        name, ms[0], ms[4:12].min(), ms[FRAMES // 2:].mean(), ms[FRAMES // 2:].min()), flush=True)
