"""Would sorting pay for a batch of truly incoherent rays (random origins in the scene's box, random directions)?
Times trx_trace_rays_dev on the batch as given, and sorted by direction octant + Morton code of the origin (torch.sort
stands in for the sorter; its own time is printed).  Development aid."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tools.prof_config import hemisphere_rays  # noqa: E402

dev = "cuda"


def time_rays(sc, rays, reps=10):
    n = rays.shape[0]
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for i in range(reps):
        e0.record()
        sc.trace_rays_dev(rays.data_ptr(), n, out.data_ptr(), sem=3)
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            best = min(best, e0.elapsed_time(e1))
    return best, out


def spread(x):
    x = (x | (x << 16)) & 0x030000FF
    x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3
    x = (x | (x << 2)) & 0x09249249
    return x


for name in sys.argv[1:] or ["bistro", "hairball"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    sc = T.Scene(flat)
    n = 1 << 21
    rays_np = hemisphere_rays(flat, None, None, n, 5)
    base = torch.from_numpy(rays_np.view(np.float32).reshape(n, 8).copy()).to(dev)
    t0, ref = time_rays(sc, base)
    print("%s: %d random rays as given %.3f ms (%.0f Mrays/s)" % (name, n, t0, n / t0 / 1e3), flush=True)
    o, d = base[:, 0:3], base[:, 4:7]
    lo, hi = o.min(0).values, o.max(0).values
    q = ((o - lo) / (hi - lo).clamp(min=1e-20) * 1023.0).long().clamp(0, 1023)
    morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    octant = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
    for label, key in (("Morton order of the origins", morton), ("octant, then Morton order", (octant << 30) | morton),
                       ("Morton order, then octant", (morton << 3) | octant)):
        order = torch.sort(key, stable=True).indices
        t, out = time_rays(sc, base[order].contiguous())
        same = bool((out == ref[order]).all())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o2 = torch.sort(key, stable=True).indices
            _ = base[o2]
        e1.record()
        torch.cuda.synchronize()
        print("   %-30s %.3f ms (x%.2f) same=%s | sort + gather %.3f ms" % (label, t, t0 / t, same, e0.elapsed_time(e1) / 5), flush=True)
    sc.close()
