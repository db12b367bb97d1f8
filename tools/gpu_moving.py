"""What does a moving camera cost against a static one AT THE SAME PLACE?  The bench line's moving-camera leg (5 cm per
frame, tile order learnt from the previous view) is compared frame by frame with the settled static-camera time at the
same view, so that the change of the scene along the path is not mistaken for the cost of a stale order."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

T.load().trx_set_kernel_variant(int(os.environ.get("TRX_VARIANT", "0"), 0))
name = os.environ.get("SCENE", "bistro")
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts, use_tlas=False)
eye, look, fov = T.scene_camera(name)
sc = T.Scene(flat)
buf = torch.empty(w * h, dtype=torch.int64, device="cuda")
STEP = float(os.environ.get("STEP", "0.05"))


ROT = float(os.environ.get("ROT", "0"))  # degrees of yaw per frame about the eye (with STEP=0: a turning camera)


def view_at(f):
    off = STEP * f
    a = np.radians(ROT * f)
    dx, dz = look[0] - eye[0], look[2] - eye[2]
    lx, lz = eye[0] + dx * np.cos(a) - dz * np.sin(a), eye[2] + dx * np.sin(a) + dz * np.cos(a)
    return T.view_from_camera((eye[0] + off, eye[1], eye[2]), (lx + off, look[1], lz), fov, w, h)


def run(views):
    evs = []
    for v in views:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        sc.trace_primary_dev(v, w, h, buf.data_ptr(), sem=3)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return np.array([a.elapsed_time(b) for a, b in evs])


N = int(os.environ.get("FRAMES", "96"))
run([view_at(0)] * 40)                        # settle (tuner decision included)
moving = run([view_at(f) for f in range(N)])
print("moving: " + " ".join("%.3f" % t for t in moving))
static = {}
for f in range(8, N, 12):
    t = run([view_at(f)] * 12)
    static[f] = t[6:].mean()
print("frame  moving(f-1..f+1 mean)  static  ratio")
for f, s in static.items():
    m = moving[f - 1:f + 2].mean()
    print("%5d  %.3f  %.3f  %.3f" % (f, m, s, m / s))
print("moving mean %.4f (frames 8..) | static mean at the sampled views %.4f" % (moving[8:].mean(), np.mean(list(static.values()))))
