"""Pre-splitting (--split) sweep: build, traversal counts and frame time per extra-reference ratio (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

for name in sys.argv[1:] or ["bistro"]:
    w, h = 1920, 1080
    verts, counts = T.gen_scene(name, 0, 1)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    for sp in (0.0, 0.1, 0.3, 1.0):
        t0 = time.time()
        flat = T.flat_build(verts, counts, split=sp)
        tb = time.time() - t0
        sc = T.Scene(flat)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
        prim, ao, fms = sc.trace_primary_ao(view, w, h, sem=3, frame=1, ao_eps=0.01)
        best = min(sc.trace_primary_ao(view, w, h, sem=3, frame=f, ao_eps=0.01)[2] for f in range(6))
        print("%s split %.1f: refs %d nodes %d build %.1fs | node/ray %.2f tri/ray %.2f | primary %.3f ms | primary+AO %.3f ms" % (
            name, sp, flat.n_tris, flat.n_nodes, tb, st.n_node / st.n_rays, st.n_tri / st.n_rays, mn, best), flush=True)
        sc.close()
    T.flat_build(verts[:1], split=0.0)
