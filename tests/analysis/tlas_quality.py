"""Oracle only (CPU): quality of the upper tree of a two-level scene - node visits per primary ray split into TLAS and BLAS
visits, BLAS (sub)trees entered and triangle tests per ray - per re-braiding area fraction.
usage: python tests/analysis/tlas_quality.py [scene] [scale] [fraction ...]   (default: san_miguel, image sides / 4)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tray_racing_amd as T  # noqa: E402
from oracle import binding as O  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "san_miguel"
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 4
fracs = [float(x) for x in sys.argv[3:]] or [0.0, 1.0 / 256, 1.0 / 1024, 1.0 / 4096, 1.0 / 16384]
w, h = 3840 // scale, 2160 // scale
lib = T.load()
verts, counts = T.gen_scene(name, 0, 1)
eye, look, fov = T.scene_camera(name)
views = [O.view_from_bytes(T.view_from_camera(eye, look, fov, w, h))]
# two more cameras: from the far corner looking back, and from above
views.append(O.view_from_bytes(T.view_from_camera([look[0], eye[1] + 2.0, look[2]], eye, fov, w, h)))
views.append(O.view_from_bytes(T.view_from_camera([eye[0], eye[1] + 12.0, eye[2]], look, fov, w, h)))
O.set_simd(True)
for fr in fracs:
    lib.trx_set_build_rebraid(fr)
    t0 = time.time()
    flat = T.flat_build(verts, counts, use_tlas=True)
    bt = time.time() - t0
    osc = O.Scene.from_flat(flat)
    row = []
    for v in views:
        _, st = osc.trace_primary(v, w, h, sem=3)
        row.append("%.1f (%.1f tlas + %.1f blas, %.2f entered, %.1f tris)" % (
            st.n_node / st.n_rays, st.n_tlas_node / st.n_rays, (st.n_node - st.n_tlas_node) / st.n_rays,
            st.n_inst_enter / st.n_rays, st.n_tri / st.n_rays))
    print("rebraid 1/%-6.0f %7d prims, tlas %6d nodes, build %.1f s (tlas %.0f ms): %s" % (
        1.0 / fr if fr else 0, flat.instance_offsets.size, flat.n_nodes - flat.tlas_start, bt, flat.tlas_build_s * 1e3, " | ".join(row)), flush=True)
lib.trx_set_build_rebraid(1.0 / 4096)
