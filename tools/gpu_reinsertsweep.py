"""Frame times per reinsertion setting of this library's own (binned-SAH) pipeline (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
w, h = 1920, 1080
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bistro", "bistro_dense", "hairball"]
for name in names:
    verts, counts = T.gen_scene(name, 0, 1)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    for ratio, iters in ((0.0, 0), (0.02, 4), (0.05, 4), (0.1, 4), (0.1, 8), (0.25, 8)):
        L.check(lib.trx_set_build_reinsertion(ratio, iters))
        t0 = time.time()
        flat = T.flat_build(verts, counts)
        tb = time.time() - t0
        sc = T.Scene(flat)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
        fms = min(sc.trace_primary_ao(view, w, h, sem=3, frame=f, ao_eps=0.01)[2] for f in range(4))
        print("%-12s reinsertion %.2f x %d | build %5.1f s | %5.2f nodes/ray %5.2f tris/ray | primary %.3f ms (mean %.3f) | primary+AO %.3f ms" % (
            name, ratio, iters, tb, st.n_node / st.n_rays, st.n_tri / st.n_rays, mn, mean, fms), flush=True)
        sc.close()
L.check(lib.trx_set_build_reinsertion(0.02, 4))
