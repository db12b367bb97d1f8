"""Sweep of the SAH collapse weights / leaf size of the stand-in builder (the reference's
--collapse-traversal-cost x --max-prims-per-leaf auto-tune, src/auto_tune.rs:20-28)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
for max_prims in (3, 2, 1):
    for prim_cost in (0.15, 0.3, 0.5, 0.8, 1.2, 2.0):
        if max_prims == 1 and prim_cost != 0.3:
            continue
        flat = T.flat_build(verts, counts, max_prims_per_leaf=max_prims, traversal_cost=1.0, prim_cost=prim_cost)
        sc = T.Scene(flat)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=20)
        print("%s max_prims %d prim_cost %.2f: nodes %d | node/ray %.2f tri/ray %.2f | eff node %.2f tri %.2f | "
              "min %.3f ms mean %.3f ms  %.1f Mrays/s" % (
                  name, max_prims, prim_cost, flat.n_nodes, st.n_node / st.n_rays, st.n_tri / st.n_rays,
                  st.n_node / (64.0 * max(st.n_wave_node, 1)), st.n_tri / (64.0 * max(st.n_wave_tri, 1)), mn, mean,
                  w * h / mn / 1e3), flush=True)
        sc.close()
