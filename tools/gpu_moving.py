"""Frame time of a camera that moves a fixed step per frame (the bench's moving-camera leg), per step size: every frame a
new view, the tile order on file from an earlier one.  usage: [TRX_LIB=...] python tools/gpu_moving.py [scene] [step_m ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
steps = [float(x) for x in sys.argv[2:]] or [0.0, 0.05, 0.5, 2.0]
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
sc = T.Scene(flat)
buf = torch.empty(w * h, dtype=torch.int64, device="cuda")
for i in range(150):   # clocks up, the slot past its first frames
    sc.trace_primary_dev(T.view_from_camera(eye, look, fov, w, h), w, h, buf.data_ptr(), sem=3)
torch.cuda.synchronize()
for step in steps:
    res = []
    for rep in range(3):
        evs = []
        for f in range(72):
            off = step * f
            v = T.view_from_camera((eye[0] + off, eye[1], eye[2]), (look[0] + off, look[1], look[2]), fov, w, h)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sc.trace_primary_dev(v, w, h, buf.data_ptr(), sem=3)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in evs][8:]
        res.append(sum(ts) / len(ts))
    print("%s step %.2f m/frame: mean of 64 frames %s ms" % (name, step, " ".join("%.4f" % x for x in res)), flush=True)
sc.check()
