"""First frames and moving cameras (development aid; a -DTRX_DEV_TUNE build in TRX_LIB lets the old natural-order first
frame be measured beside the probe-ordered one).  Per scene: the learnt order (static camera), every frame a camera cut
with the probe pass (variant bit 7), the same without the probe (TRX_TUNE 0x8000: what a first frame got before round
3), feedback off, and a camera that advances `step` metres per frame.
usage: python tools/gpu_first_frame.py bistro,bistro_dense,hairball,kitchen [tlas scene ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

import torch  # noqa: E402

lib = T.load()
for name in sys.argv[1].split(","):
    tlas = name.endswith("+tlas")
    name = name.replace("+tlas", "")
    w, h = (3840, 2160) if tlas else (1920, 1080)
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts, use_tlas=tlas)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    row = {}
    for tag, variant, tune in (("learnt", 0, 0), ("cut+probe", 0x80, 0), ("cut, no probe", 0x80, 0x8000), ("feedback off", 1 << 20, 0)):
        os.environ["TRX_TUNE"] = str(tune)
        lib.trx_set_kernel_variant(variant)
        best = (1e9, 1e9)
        for _ in range(3):
            mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=5, frames=30)
            best = min(best, (mean, mn))
        row[tag] = best
    os.environ["TRX_TUNE"] = "0"
    lib.trx_set_kernel_variant(0)
    buf = torch.empty(w * h, dtype=torch.int64, device="cuda")
    moving = {}
    for step in (0.05, 0.5):
        ev = []
        for f in range(72):
            off = step * f
            v = T.view_from_camera((eye[0] + off, eye[1], eye[2]), (look[0] + off, look[1], look[2]), fov, w, h)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            sc.trace_primary_dev(v, w, h, buf.data_ptr(), sem=3)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in ev][8:]
        moving[step] = sum(t) / len(t)
    print("%-14s %s | moving 5 cm/frame %.4f, 50 cm/frame %.4f ms" % (
        name + ("+tlas" if tlas else ""), " | ".join("%s mean %.4f min %.4f" % (k, v[0], v[1]) for k, v in row.items()),
        moving[0.05], moving[0.5]), flush=True)
    sc.close()
