"""Round 4: the reference-style frame (primary ray + one AO ray per hit pixel) as two launches against ONE launch
(trx_trace_frame_dev: a lane whose primary ray hits becomes the pixel's AO ray in place), per scene and per
kernel-variant word (bits 0..6: idle lanes that trigger a refill; bits 14..15: waiting lanes that trigger a conversion once
the queues are dry).  Checks that both ways write the same records.
usage: python tools/gpu_fused.py [scene ...]   VARIANTS=0,16,24,32,48,64 (default)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

w, h = 1920, 1080
lib = T.load()
variants = [int(v, 0) for v in os.environ.get("VARIANTS", "0,8,24,32,48,64").split(",")]
for name in sys.argv[1:] or ["bistro", "hairball", "kitchen"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    p2, a2 = (torch.zeros(w * h, dtype=torch.int64, device="cuda") for _ in range(2))
    p1, a1 = (torch.zeros(w * h, dtype=torch.int64, device="cuda") for _ in range(2))

    def timed(fn, reps=14, skip=4):
        ts = []
        for i in range(reps + skip):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn(i)
            b.record()
            torch.cuda.synchronize()
            if i >= skip:
                ts.append(a.elapsed_time(b))
        return min(ts), sum(ts) / len(ts)

    def two(i):
        sc.trace_primary_dev(view, w, h, p2.data_ptr(), sem=3)
        sc.trace_ao_dev(view, w, h, p2.data_ptr(), a2.data_ptr(), sem=3, frame=i % 4, ao_eps=0.01)

    t_min, t_mean = timed(two)
    print("%-9s two launches: %.3f ms min / %.3f mean" % (name, t_min, t_mean), flush=True)
    for v in variants:
        lib.trx_set_kernel_variant(v)

        def one(i):
            sc.trace_frame_dev(view, w, h, p1.data_ptr(), a1.data_ptr(), sem=3, frame=i % 4, ao_eps=0.01)
        f_min, f_mean = timed(one)
        lib.trx_set_kernel_variant(0)
        two(1)
        lib.trx_set_kernel_variant(v)
        one(1)
        lib.trx_set_kernel_variant(0)
        torch.cuda.synchronize()
        same = bool((p1 == p2).all()) and bool((a1 == a2).all())
        print("%-9s one launch, variant 0x%x: %.3f ms min / %.3f mean (%.3fx of two launches)%s" % (
            name, v, f_min, f_mean, f_mean / t_mean, "" if same else "   RECORDS DIFFER"), flush=True)
    sc.check()
    sc.close()
