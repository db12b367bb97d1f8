"""Times a list of kernel-variant words on one scene (development aid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
name = sys.argv[1]
variants = [int(x, 0) for x in sys.argv[2:]]
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
for rep in range(2):
    for v in variants:
        lib.trx_set_kernel_variant(v)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
        print("%s variant 0x%08x: min %.3f ms mean %.3f ms %.1f Mrays/s" % (name, v, mn, mean, w * h / mn / 1e3), flush=True)
lib.trx_set_kernel_variant(0)
sc.close()
