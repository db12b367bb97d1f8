"""Timings of every BASELINE.json config on the stand-in scenes (development aid / DESIGN.md table)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

T.load().trx_set_kernel_variant(int(os.environ.get("TRX_VARIANT", "0"), 0))  # tuning aid


def bench_ao(sc, view, w, h, frames, sem=3):
    best_p, best_f = 1e9, 1e9
    hits = 0
    for f in range(frames + 4):
        prim, ao, ms = sc.trace_primary_ao(view, w, h, sem=sem, frame=f % 4, ao_eps=0.01)
        if f >= 4:
            best_f = min(best_f, ms)
        hits = int(np.isfinite(prim["t"]).sum())
    return best_f, hits


for name, w, h, tlas in [("kitchen", 1920, 1080, False), ("bistro", 1920, 1080, False), ("bistro_dense", 1920, 1080, False), ("hairball", 1920, 1080, False),
                         ("san_miguel", 3840, 2160, True), ("demoscene", 512, 1080, False)]:
    verts, counts = T.gen_scene(name, 0, 1)
    t0 = time.time()
    flat = T.flat_build(verts, counts, use_tlas=tlas)
    tb = time.time() - t0
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    st = sc.count_primary(view, w, h, sem=3)
    mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
    line = "%s %dx%d tlas=%s: %d tris %d nodes build %.1fs | node/ray %.1f tri/ray %.1f maxsp %d | primary %.3f ms = %.0f Mrays/s" % (
        name, w, h, tlas, flat.n_tris, flat.n_nodes, tb, st.n_node / st.n_rays, st.n_tri / st.n_rays, st.max_stack, mn,
        w * h / mn / 1e3)
    fms, hits = bench_ao(sc, view, w, h, 8)
    line += " | primary+AO %.3f ms = %.0f Mrays/s (%d AO rays)" % (fms, (w * h + hits) / fms / 1e3, hits)
    print(line, flush=True)
    if name == "bistro":
        for sem, tag in [(0, "HLSL"), (3, "CPU"), (7, "CPU+FMA"), (4, "HLSL+FMA")]:
            mn, _ = sc.bench_primary(view, w, h, sem=sem, warmup=8, frames=30)
            print("   semantics %-8s %.3f ms" % (tag, mn), flush=True)
    sc.close()
