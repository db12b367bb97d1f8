"""[needs a development build: make -C tray_racing_amd/csrc KFLAGS=-DTRX_DEV_TUNE OUT=... and TRX_LIB pointing at it]
Times a list of TRX_TUNE development words (and optionally kernel-variant words, as tune:variant) on scenes
(development aid).  usage: python tools/gpu_tune.py bistro,hairball 0 1 2 0x4:0x100000"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
names = sys.argv[1].split(",")
words = [tuple(int(y, 0) for y in (x.split(":") + ["0"])[:2]) for x in sys.argv[2:]]
w, h = 1920, 1080
for name in names:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    ref = None
    for rep in range(2):
        for tune, variant in words:
            os.environ["TRX_TUNE"] = str(tune)
            lib.trx_set_kernel_variant(variant)
            hits, _ = sc.trace_primary(view, w, h, sem=3)
            if ref is None:
                ref = hits.copy()
            same = bool((hits["prim"] == ref["prim"]).all() and (hits["t"].view("u4") == ref["t"].view("u4")).all())
            mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=8, frames=30)
            print("%s tune 0x%x variant 0x%08x: min %.3f ms mean %.3f ms %.1f Mrays/s same=%s" % (
                name, tune, variant, mn, mean, w * h / mn / 1e3, same), flush=True)
    os.environ["TRX_TUNE"] = "0"
    lib.trx_set_kernel_variant(0)
    sc.close()
