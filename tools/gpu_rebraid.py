"""TLAS re-braiding sweep (development aid): for each area fraction of trx_set_build_rebraid, TLAS primitives,
node visits / triangle tests per primary ray (counting kernel) and the frame time of the TLAS scene.
usage: python tools/gpu_rebraid.py san_miguel 3840 2160 0 0.0625 0.015625 0.00390625 0.0009765625"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

lib = T.load()
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
verts, counts = T.gen_scene(name, 0, 1)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
for frac in [float(x) for x in sys.argv[4:]]:
    lib.trx_set_build_rebraid(frac)
    t0 = time.time()
    flat = T.flat_build(verts, counts, use_tlas=True)
    build = time.time() - t0
    sc = T.Scene(flat)
    st = sc.count_primary(view, w, h, sem=3)
    mn, mean = min(sc.bench_primary(view, w, h, sem=3, warmup=5, frames=20)[::-1] for _ in range(2))[::-1]
    prim, ao, ms = sc.trace_primary_ao(view, w, h, sem=3, frame=0, ao_eps=0.01)
    print("%s %dx%d rebraid %.6f: %d TLAS primitives over %d objects, TLAS nodes %d, build %.1f s (tlas %.3f s) | nodes/ray %.1f tris/ray %.1f | "
          "primary min %.3f mean %.3f ms | primary+AO %.3f ms" % (
              name, w, h, frac, flat.instance_offsets.size, len(counts), flat.n_nodes - flat.tlas_start, build, flat.tlas_build_s,
              st.n_node / st.n_rays, st.n_tri / st.n_rays, mn, mean, ms), flush=True)
    sc.close()
lib.trx_set_build_rebraid(1.0 / 4096.0)
