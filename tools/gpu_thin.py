"""Round 4: thin waves (a dry wave down to eight rays gives every ray eight lanes) on and off (kernel-variant bit 28):
AO pass, 4-frame AO batch, the one-launch frame and 2 M random rays per scene; alternating repetitions.
usage: python tools/gpu_thin.py [scene ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

w, h = 1920, 1080
lib = T.load()
NO_THIN = 1 << 28
for name in sys.argv[1:] or ["hairball", "bistro", "kitchen"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    prim = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    ao = torch.zeros(4 * w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=3)
    torch.cuda.synchronize()
    n_ao = int(((prim & 0xffffffff) != 0x7f800000).sum().item())
    rng = np.random.default_rng(3)
    pts = flat.tri_verts.reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    rays = np.zeros(2_000_000, dtype=T.RAY_DTYPE)
    rays["origin"] = rng.uniform(lo, hi, size=(rays.size, 3)).astype(np.float32)
    d = rng.normal(size=(rays.size, 3)).astype(np.float32)
    rays["direction"] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["tmax"] = 3.0e38
    d_rays = torch.from_numpy(rays.view(np.uint8)).cuda()
    d_hits = torch.zeros(rays.size, dtype=torch.int64, device="cuda")
    passes = {
        "AO pass": lambda i: sc.trace_ao_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), sem=3, frame=i % 4, ao_eps=0.01),
        "AO x 4, one launch": lambda i: sc.trace_ao_batch_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), w * h, 4, sem=3, frame0=4 * i, ao_eps=0.01),
        "frame, one launch": lambda i: sc.trace_frame_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), sem=3, frame=i % 4, ao_eps=0.01),
        "2 M random rays": lambda i: sc.trace_rays_dev(d_rays.data_ptr(), rays.size, d_hits.data_ptr(), sem=3),
    }
    sweep = [int(x) for x in os.environ.get("THIN_SWEEP", "").split(",") if x]   # development build: TRX_THIN_MAX values
    if sweep:
        for label, fn in passes.items():
            ts = {v: [] for v in sweep}
            for i in range(3 + len(sweep) * 6):
                v = sweep[i % len(sweep)]
                os.environ["TRX_THIN_MAX"] = str(v)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn(i)
                b.record()
                torch.cuda.synchronize()
                if i >= 3:
                    ts[v].append(a.elapsed_time(b))
            print("%-9s %-20s %s" % (name, label, " | ".join("thin_max %2d: %.3f min %.3f mean" % (v, min(ts[v]), sum(ts[v]) / len(ts[v])) for v in sweep)), flush=True)
        sc.close()
        continue
    for label, fn in passes.items():
        ts = {0: [], NO_THIN: []}
        for i in range(3 + 2 * 8):
            v = NO_THIN if i & 1 else 0
            lib.trx_set_kernel_variant(v)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn(i)
            b.record()
            torch.cuda.synchronize()
            if i >= 3:
                ts[v].append(a.elapsed_time(b))
        lib.trx_set_kernel_variant(0)
        on, off = ts[0], ts[NO_THIN]
        print("%-9s %-20s thin waves on %.3f ms min / %.3f mean | off %.3f / %.3f | on / off %.3f  (%d AO rays)" % (
            name, label, min(on), sum(on) / len(on), min(off), sum(off) / len(off), (sum(on) / len(on)) / (sum(off) / len(off)), n_ao), flush=True)
    sc.check()
    sc.close()
