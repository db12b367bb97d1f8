"""Residency / tail diagnostics: per-wave lifetimes of one frame (development aid)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
variants = [int(x, 0) for x in sys.argv[2:]] or [0]
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)


def timeline(tag):
    buf = np.zeros(2 * 8192, dtype=np.uint64)
    n = C.c_uint32()
    for _ in range(10):  # let every launch slot learn the tile order
        L.check(lib.trx_debug_wave_timeline(sc.handle, C.byref(view), w, h, 3, buf.ctypes.data_as(C.c_void_p), 8192,
                                            C.byref(n)))
    t = buf[: 2 * n.value].reshape(-1, 2).astype(np.int64)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0  # us
    end = (t[:, 1] - t0) / 100.0
    life = end - start
    total = end.max()
    print("%s: %d waves, frame %.1f us | end p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f | mean lifetime %.1f us (%.0f%% of frame)" % (
        tag, n.value, total, np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99),
        end.max(), life.mean(), 100 * life.mean() / total), flush=True)
    wpb = int(os.environ.get("WPB", "1"))
    xcd = (np.arange(n.value) // wpb) % 8
    print("   per XCD (workgroup %% 8): end p50 / max = %s" % ["%.0f/%.0f" % (np.median(end[xcd == x]), end[xcd == x].max()) for x in range(8)], flush=True)
    late = np.argsort(end)[-6:]
    print("   last waves out: %s" % ["w%d xcd%d %.0f" % (i, xcd[i], end[i]) for i in late], flush=True)
    ts = np.linspace(0, total, 11)
    print("   alive waves at 0..100%% of the frame: %s" % [int(((start <= x) & (end > x)).sum()) for x in ts], flush=True)


for v in variants:
    lib.trx_set_kernel_variant(v)
    mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=10, frames=20)
    print("variant 0x%x: min %.3f ms mean %.3f ms %.1f Mrays/s" % (v, mn, mean, w * h / mn / 1e3), flush=True)
    timeline("   timeline")
lib.trx_set_kernel_variant(0)
sc.close()
