"""Sweep of the collapse's SAH weights (the knob behind the reference's --collapse-traversal-cost, which its --auto-tune
sweeps, src/auto_tune.rs:20-28): frame times per (traversal_cost, prim_cost) on the stand-in scenes (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
w, h = 1920, 1080
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bistro", "bistro_dense", "hairball", "kitchen"]
costs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.15, 0.3, 0.5, 0.8, 1.2]
for name in names:
    verts, counts = T.gen_scene(name, 0, 1)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    for pc in costs:
        L.check(lib.trx_set_build_costs(1.0, pc))
        flat = T.flat_build(verts, counts)
        sc = T.Scene(flat)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
        fms = min(sc.trace_primary_ao(view, w, h, sem=3, frame=f, ao_eps=0.01)[2] for f in range(4))
        print("%-12s prim_cost %.2f | %7d nodes | %5.2f nodes/ray %5.2f tris/ray | primary %.3f ms (mean %.3f) | primary+AO %.3f ms" % (
            name, pc, flat.n_nodes, st.n_node / st.n_rays, st.n_tri / st.n_rays, mn, mean, fms), flush=True)
        sc.close()
L.check(lib.trx_set_build_costs(1.0, 0.3))
