"""Round 6: the frame loop (trx_frame_loop) serial, overlapped (AO passes on a second stream, primaries running ahead as far as
four hit buffers allow) and GATED (overlap = 2: primary(i + 1) starts when AO(i) finds its queues dry); records compared.
usage: python tools/gpu_frame_overlap.py [scene ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import tray_racing_amd as T  # noqa: E402

for name in (sys.argv[1:] or ["bistro", "hairball", "kitchen"]):
    v, c = T.gen_scene(name, 0, 1)
    flat = T.flat_build(v, c, preset="medium_build")
    eye, look, fov = T.scene_camera(name)
    w, h = 1920, 1080
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    ref = None
    out = []
    for mode in (0, 1, 2, 0, 1, 2):
        sc.frame_loop(view, w, h, sem=3, frames=8, overlap=mode, fetch=False)
        t = min(sc.frame_loop(view, w, h, sem=3, frames=48, overlap=mode, fetch=False)[0] for _ in range(3)) / 48
        ms, prim, ao = sc.frame_loop(view, w, h, sem=3, frames=6, overlap=mode, fetch=True)
        if ref is None:
            ref = (prim.copy(), ao.copy())
        same = bool((prim.view(np.uint64) == ref[0].view(np.uint64)).all() and (ao.view(np.uint64) == ref[1].view(np.uint64)).all())
        out.append("%s %.4f%s" % (("serial", "overlapped", "gated")[mode], t, "" if same else " RECORDS DIFFER"))
    print(name, "ms per frame:", " | ".join(out), flush=True)
    sc.close()
