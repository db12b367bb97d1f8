"""Build seconds, node visits and frame times per builder preset name (the reference's --preset values,
src/main.rs:125-131,563-570) of this library's own pipeline (development aid / DESIGN.md section 7 table)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

w, h = 1920, 1080
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["bistro"]):
    verts, counts = T.gen_scene(name, 0, 1)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    for preset in ("fastest_build", "very_fast_build", "fast_build", "medium_build", "slow_build", "very_slow_build"):
        t0 = time.time()
        flat = T.flat_build(verts, counts, preset=preset)
        tb = time.time() - t0
        sc = T.Scene(flat)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
        fms = min(sc.trace_primary_ao(view, w, h, sem=3, frame=f, ao_eps=0.01)[2] for f in range(4))
        print("%-12s %-16s | build %5.1f s | %7d nodes %8d triangle entries | %5.2f nodes/ray %5.2f tris/ray | primary %.3f ms (mean %.3f) | primary+AO %.3f ms" % (
            name, preset, tb, flat.n_nodes, flat.n_tris, st.n_node / st.n_rays, st.n_tri / st.n_rays, mn, mean, fms), flush=True)
        sc.close()
T.flat_build(verts[:4], preset="medium_build")
