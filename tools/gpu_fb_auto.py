"""Self-tuning of the tile-order feedback (development aid): primary frame times per scene with the feedback always on
(variant bit 19), always off (bit 20) and self-tuned (default), over enough frames for the slot to decide.
usage: python tools/gpu_fb_auto.py kitchen,bistro,bistro_dense,hairball"""
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
    out = []
    for tag, variant in (("always on", 1 << 19), ("always off", 1 << 20), ("self-tuned", 0)):
        sc = T.Scene(flat)
        lib.trx_set_kernel_variant(variant)
        res = [sc.bench_primary(view, w, h, sem=3, warmup=0 if k else 40, frames=40) for k in range(5)]
        out.append("%s: %s" % (tag, " ".join("%.4f/%.4f" % (mn, mean) for mn, mean in res)))
        lib.trx_set_kernel_variant(0)
        sc.close()
    print("%-12s min/mean ms over 5 x 40 frames (after 40): %s" % (name, " | ".join(out)), flush=True)
