"""Round 4: n AO frames (noise seeds frame0 .. frame0 + n - 1, configs[3]'s "4 spp") as n launches against ONE launch
(trx_trace_ao_batch_dev), per scene: hipEvent time of the n passes, rays per second over all of them.
usage: python tools/gpu_ao_batch.py [scene ...]   (default: hairball bistro kitchen)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

w, h = 1920, 1080
for name in sys.argv[1:] or ["hairball", "bistro", "kitchen"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    prim = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    ao = torch.zeros(8 * w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=3)
    torch.cuda.synchronize()
    n_ao = int(((prim & 0xffffffff) != 0x7f800000).sum().item())

    def timed(fn, reps=10, skip=3):
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

    for n in (1, 2, 4, 8):
        def separate(i, n=n):
            for f in range(n):
                sc.trace_ao_dev(view, w, h, prim.data_ptr(), ao.data_ptr() + 8 * f * w * h, sem=3, frame=4 * i + f, ao_eps=0.01)

        def batch(i, n=n):
            sc.trace_ao_batch_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), w * h, n, sem=3, frame0=4 * i, ao_eps=0.01)
        s_min, s_mean = timed(separate)
        b_min, b_mean = timed(batch)
        print("%-9s %d AO frame(s) of %d rays: %d launches %.3f ms min / %.3f mean = %.0f Mrays/s | one launch %.3f ms min / %.3f "
              "mean = %.0f Mrays/s (%.2fx)" % (name, n, n_ao, n, s_min, s_mean, n * n_ao / s_mean / 1e3, b_min, b_mean,
                                               n * n_ao / b_mean / 1e3, s_mean / b_mean), flush=True)
    sc.check()
    sc.close()
