"""Primary-frame time, node visits and build seconds per builder setting: this library's binned-SAH pipeline against
the ploc_cwbvh pipeline with the reference's parameters (development aid / DESIGN.md section 7 table)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bistro"]
w, h = 1920, 1080
for name in names:
    verts, counts = T.gen_scene(name, 0, 1)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    settings = [("binned SAH + reinsertion 0.02 x 4 (trx_flat_build, medium_build)", None)]
    for d, th, bits, r in [(14, 2, 64, 0.15), (14, 2, 64, 0.02), (14, 2, 64, 0.0), (2, 2, 64, 0.15), (32, 2, 64, 0.15), (14, 2, 128, 0.15),
                           (14, 0, 64, 0.15)]:
        settings.append(("ploc distance %d threshold %d bits %d, reinsertion %.2f" % (d, th, bits, r),
                         T.build_params(ploc_search_distance=d, search_depth_threshold=th, sort_precision=bits, reinsertion_batch_ratio=r)))
    for label, bp in settings:
        t0 = time.time()
        flat = T.flat_build(verts, counts) if bp is None else T.flat_build_params(verts, counts, bp)
        tb = time.time() - t0
        sc = T.Scene(flat)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
        _, _, fms = sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)
        fms = min(fms, sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)[2])
        print("%s | %-66s | build %5.1f s | %7d nodes | %5.2f nodes/ray %5.2f tris/ray | primary %.3f ms | primary+AO %.3f ms" % (
            name, label, tb, flat.n_nodes, st.n_node / st.n_rays, st.n_tri / st.n_rays, mn, fms), flush=True)
        sc.close()
