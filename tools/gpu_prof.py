"""SIMD-efficiency and frame-size diagnostics for the traversal kernel (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
for name in sys.argv[1:] or ["bistro"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    sc = T.Scene(flat)
    for (w, h) in [(1920, 1080), (3840, 2160)]:
        view = T.view_from_camera(eye, look, fov, w, h)
        for variant in (64, 32):
            lib.trx_set_kernel_variant(variant)
            st = sc.count_primary(view, w, h, sem=3)
            mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=2, frames=10)
            print("%s %dx%d variant %d: %.3f ms %.1f Mrays/s | node/ray %.2f tri/ray %.2f | "
                  "node SIMD eff %.3f (wave node steps %d) tri SIMD eff %.3f (wave tri steps %d) | count kernel %.3f ms" % (
                      name, w, h, variant, mn, w * h / mn / 1e3, st.n_node / st.n_rays, st.n_tri / st.n_rays,
                      st.n_node / (64.0 * max(st.n_wave_node, 1)), st.n_wave_node,
                      st.n_tri / (64.0 * max(st.n_wave_tri, 1)), st.n_wave_tri, st.kernel_ms), flush=True)
    lib.trx_set_kernel_variant(0)
    sc.close()
