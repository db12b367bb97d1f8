"""Why does a moving camera lose part of the tile order's gain, and would dilating the cost classes recover it?
Measures the per-tile costs (trx_debug_tile_profile) of consecutive views STEP metres apart and list-schedules each
view's tiles on W waves (greedy, the next tile of the order to the first free wave) in several orders: the view's own
classes (what a static camera replays), the previous view's classes (what a moving camera replays), the previous view's
classes dilated over the 3x3 / 5x5 tile neighbourhood, and the natural order.  Development aid."""
import ctypes as C
import heapq
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
name = os.environ.get("SCENE", "bistro")
STEP = float(os.environ.get("STEP", "0.05"))
W = int(os.environ.get("WAVES", "4096"))
w, h = 1920, 1080
tx, ty = (w + 7) // 8, (h + 7) // 8
n = tx * ty
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
sc = T.Scene(flat)


def costs(f):
    off = STEP * f
    v = T.view_from_camera((eye[0] + off, eye[1], eye[2]), (look[0] + off, look[1], look[2]), fov, w, h)
    cost = np.zeros(n, dtype=np.uint32)
    iters = np.zeros(n, dtype=np.uint32)
    best = None
    for _ in range(3):   # the cheapest of three measurements per tile (contention noise)
        L.check(lib.trx_debug_tile_profile(sc.handle, C.byref(v), w, h, 3, cost.ctypes.data_as(C.c_void_p),
                                           iters.ctypes.data_as(C.c_void_p), n))
        best = cost.copy() if best is None else np.minimum(best, cost)
    return best


def classes(c):
    c = c.astype(np.int64) | 1
    msb = np.floor(np.log2(c)).astype(np.int64)
    kk = 2 * msb + np.where(msb > 0, (c >> np.maximum(msb - 1, 0)) & 1, 0)
    return np.clip(kk - 16, 0, 15)


def dilate(k, r):
    g = k.reshape(ty, tx)
    out = g.copy()
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            sh = np.full_like(g, 0)
            ys, yd = (slice(max(dy, 0), ty + min(dy, 0)), slice(max(-dy, 0), ty + min(-dy, 0)))
            xs, xd = (slice(max(dx, 0), tx + min(dx, 0)), slice(max(-dx, 0), tx + min(-dx, 0)))
            sh[yd, xd] = g[ys, xs]
            out = np.maximum(out, sh)
    return out.reshape(-1)


def makespan(order, c):
    free = [0.0] * W
    heapq.heapify(free)
    end = 0.0
    for t in order:
        s = heapq.heappop(free)
        e = s + c[t]
        heapq.heappush(free, e)
        end = max(end, e)
    return end / 100.0  # us


def order_of(k):
    return np.argsort(-k, kind="stable")


seq = [costs(f) for f in range(5)]
print("%s, %d tiles, %d waves, step %.3f m; simulated frame (us):" % (name, n, W, STEP))
print("view  sum/W   own-classes  own-exact  prev-classes  prev-dil3x3  prev-dil5x5  prev-blend  natural")
for f in range(1, 5):
    c, p = seq[f].astype(np.float64), seq[f - 1]
    kp = classes(p)
    blend = np.maximum(kp, dilate(kp, 1) - 1)   # neighbours count one class less
    print("%4d  %6.1f  %10.1f  %9.1f  %12.1f  %11.1f  %11.1f  %10.1f  %7.1f" % (
        f, c.sum() / W / 100.0, makespan(order_of(classes(seq[f])), c), makespan(np.argsort(-c), c),
        makespan(order_of(kp), c), makespan(order_of(dilate(kp, 1)), c), makespan(order_of(dilate(kp, 2)), c),
        makespan(order_of(blend), c), makespan(np.arange(n), c)))
    top = np.argsort(-c)[:200]
    print("      the view's 200 heaviest tiles: previous-view class rank percentile (median / worst) %.1f / %.1f" % (
        np.median([(kp >= kp[t]).mean() * 100 for t in top]), max((kp >= kp[t]).mean() * 100 for t in top)))
sc.close()
