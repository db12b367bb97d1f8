"""What is the tail of a learnt-order frame made of?  Needs a -DTRX_TAIL_DIAG build (TRX_LIB=tuning_libs/tail.so):
per wave, the start of its LAST tile, that tile's position in the frame's order and the number of tiles it traced."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
lib.trx_set_kernel_variant(int(os.environ.get("TRX_VARIANT", "0x80000"), 0))
w, h = 1920, 1080
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
buf = np.zeros(8 * 8192, dtype=np.uint64)
n = C.c_uint32()
for _ in range(10):
    L.check(lib.trx_debug_wave_phases(sc.handle, C.byref(view), w, h, 3, buf.ctypes.data_as(C.c_void_p), 8192, C.byref(n)))
r = buf[: 8 * n.value].reshape(-1, 8).astype(np.int64)
t0 = r[:, 0].min()
start, end, last_t0, chunk, tiles = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, (r[:, 2] - t0) / 100.0, r[:, 3], r[:, 4]
frame = end.max()
n_tiles = ((w + 7) // 8) * ((h + 7) // 8)
last_len = end - last_t0
print("%s: frame %.1f us, %d waves, tiles per wave mean %.1f (min %d max %d)" % (name, frame, n.value, tiles.mean(), tiles.min(), tiles.max()))
print("last tile of a wave: duration p10 %.1f p50 %.1f p90 %.1f max %.1f us | order position p10 %.0f%% p50 %.0f%% p90 %.0f%%" % (
    np.percentile(last_len, 10), np.percentile(last_len, 50), np.percentile(last_len, 90), last_len.max(),
    100 * np.percentile(chunk, 10) / n_tiles, 100 * np.percentile(chunk, 50) / n_tiles, 100 * np.percentile(chunk, 90) / n_tiles))
late = np.argsort(end)[-12:]
print("the 12 last waves out: " + "; ".join("end %.0f last tile %.0f us at %.0f%% of the order, %d tiles" % (
    end[i], last_len[i], 100.0 * chunk[i] / n_tiles, tiles[i]) for i in late))
for lo, hi in ((0, 50), (50, 80), (80, 95), (95, 100)):
    a, b = np.percentile(end, lo), np.percentile(end, hi)
    m = (end >= a) & (end <= b)
    print("waves ending in p%d..p%d of end times (%.0f..%.0f us): last tile %.1f us mean, order position %.0f%% mean, %.1f tiles" % (
        lo, hi, a, b, last_len[m].mean(), 100 * chunk[m].mean() / n_tiles, tiles[m].mean()))

q = chunk % 8
late_m, early_m = last_t0 > np.percentile(last_t0, 90), end < np.percentile(end, 10)
print("queue of the last tile, waves whose last tile STARTED latest (top 10 %%): %s" % np.bincount(q[late_m], minlength=8).tolist())
print("queue of the last tile, waves that LEFT earliest (first 10 %%):            %s" % np.bincount(q[early_m], minlength=8).tolist())
for qq in range(8):
    m = q == qq
    print("  queue %d: last tiles of %4d waves, start p50 %.0f max %.0f us, position p50 %.0f%% min %.0f%%" % (
        qq, m.sum(), np.median(last_t0[m]), last_t0[m].max(), 100 * np.median(chunk[m]) / n_tiles, 100 * chunk[m].min() / n_tiles))

# the same frame's tiles: cost in the learnt order against the position they were given (diag build, TRX_TUNE bit 25)
os.environ["TRX_TUNE"] = str(0x2000000)
prof = []
for _ in range(2):
    cost = np.zeros(n_tiles, dtype=np.uint32)
    iters = np.zeros(n_tiles, dtype=np.uint32)
    L.check(lib.trx_debug_tile_profile(sc.handle, C.byref(view), w, h, 3, cost.ctypes.data_as(C.c_void_p),
                                       iters.ctypes.data_as(C.c_void_p), n_tiles))
    prof.append((cost >> 16, (cost & 0xffff) / 100.0, iters.copy()))
pos, us, it = prof[1]
print("learnt frame, tiles by position in the order: sum of tile time %.0f us (/ %d waves = %.1f us)" % (us.sum(), n.value, us.sum() / n.value))
edges = [0, 1, 3, 6, 12, 25, 40, 50, 60, 70, 80, 90, 100]
for a, b in zip(edges[:-1], edges[1:]):
    m = (pos >= a * n_tiles // 100) & (pos < b * n_tiles // 100)
    if m.any():
        print("  position %3d..%3d %%: %5d tiles, us mean %6.1f p10 %6.1f p50 %6.1f p90 %6.1f max %6.1f, share of all tile time %4.1f %%" % (
            a, b, m.sum(), us[m].mean(), np.percentile(us[m], 10), np.percentile(us[m], 50), np.percentile(us[m], 90), us[m].max(),
            100 * us[m].sum() / us.sum()))
k0 = np.floor(2 * np.log2(np.maximum(prof[0][1] * 100, 1))).astype(int)
k1 = np.floor(2 * np.log2(np.maximum(prof[1][1] * 100, 1))).astype(int)
d = k1 - k0
print("cost class of a tile, frame to frame (static camera): same %.1f %%, +-1 %.1f %%, off by 2 or more %.1f %%" % (
    100 * (d == 0).mean(), 100 * (abs(d) == 1).mean(), 100 * (abs(d) >= 2).mean()))
trips, pl, cw = (it >> 20).astype(float), ((it >> 10) & 1023).astype(float), (it & 1023).astype(float)
mid = (pos > 0.08 * n_tiles) & (pos < 0.6 * n_tiles) & (trips > 0)   # tiles run without issue priority, GPU full
A = np.stack([trips, pl, cw, np.ones(n_tiles)], 1)
coef, *_ = np.linalg.lstsq(A[mid], us[mid], rcond=None)
pred = A @ coef
print("mid-frame tiles: us = %.2f trips + %.2f per-lane rounds + %.2f cooperative rounds + %.1f; residual rms %.1f us (trips alone: %.1f us)" % (
    *coef, np.sqrt(np.mean((pred[mid] - us[mid]) ** 2)),
    np.sqrt(np.mean((np.polyval(np.polyfit(trips[mid], us[mid], 1), trips[mid]) - us[mid]) ** 2))))
print("per tile: trips mean %.1f max %d | per-lane rounds mean %.1f | cooperative rounds mean %.1f" % (trips.mean(), trips.max(), pl.mean(), cw.mean()))
us0 = prof[0][1]
fit = np.polyfit(trips[mid], us[mid], 1)
r0, r1 = us0 - np.polyval(fit, trips), us - np.polyval(fit, trips)
same_pos = prof[0][0] == pos
print("two consecutive learnt frames: %.1f %% of the tiles at the same position; residual (us - fit(trips)) correlation frame to frame %.2f (mid-frame tiles), %.2f (positions > 60 %%)" % (
    100 * same_pos.mean(), np.corrcoef(r0[mid], r1[mid])[0, 1], np.corrcoef(r0[pos > 0.6 * n_tiles], r1[pos > 0.6 * n_tiles])[0, 1]))
latep = pos > 0.6 * n_tiles
for name_, x in (("trips", trips), ("per-lane rounds", pl), ("cooperative rounds", cw)):
    print("  positions > 60 %%: correlation of tile time with %s %.2f" % (name_, np.corrcoef(x[latep], us[latep])[0, 1]))
worst = np.argsort(-(us * latep))[:10]
print("  slowest late tiles: " + "; ".join("%.0f us (%d trips, %d+%d rounds, pos %.0f%%, x %d y %d)" % (us[i], trips[i], pl[i], cw[i], 100.0 * pos[i] / n_tiles, i % ((w + 7) // 8), i // ((w + 7) // 8)) for i in worst))
sc.close()
