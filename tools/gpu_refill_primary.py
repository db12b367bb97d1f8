"""Primary frames with mid-tile refills (development aid): trx_set_kernel_variant bits 0..6 = refill once this many lanes
idle (64 = whole tiles, the default for primary rays; below 64 the tile-order feedback is off).
usage: python tools/gpu_refill_primary.py hairball,bistro,bistro_dense,kitchen 0 48 32 24 16"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

lib = T.load()
w, h = 1920, 1080
for name in sys.argv[1].split(","):
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    out = []
    for v in [int(x, 0) for x in sys.argv[2:]]:
        lib.trx_set_kernel_variant(v)
        mn, mean = min(sc.bench_primary(view, w, h, sem=3, warmup=30, frames=30)[::-1] for _ in range(2))[::-1]
        out.append("refill %d: %.4f/%.4f" % (v if v else 64, mn, mean))
    lib.trx_set_kernel_variant(0)
    print("%-12s min/mean ms: %s" % (name, " | ".join(out)), flush=True)
    sc.close()
