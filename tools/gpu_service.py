"""Round 6: rays per second of the single-ray trx_traverse1 through the resident ray service (single-level scenes) under
1 / 4 / 16 / 64 host threads, against trx_traverse_batch on the same rays; and what the service's idle waves cost a
concurrent primary frame.  usage: python tools/gpu_service.py [scene] [tris] [tlas]   (TRX_TRAVERSE1_COMBINER=1: two-level scenes through round 5's launch combiner)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

scene_name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w, h = 1920, 1080
verts, counts = T.gen_scene(scene_name, tris, 1)
two_level = len(sys.argv) > 3 and sys.argv[3] == "tlas"   # python tools/gpu_service.py san_miguel 0 tlas
flat = T.flat_build(verts, counts, use_tlas=True) if two_level else T.flat_build(verts, counts, preset="medium_build")
eye, look, fov = T.scene_camera(scene_name)
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
rng = np.random.default_rng(11)
_last = [0.0, 0.0, 0.0, 0.0]


def gpu_side():
    """' | on the GPU x us, y trips per call' of the calls since the last look (trx_debug_service_stats)"""
    st = sc.service_stats()
    now = [st["rays"], st["us_per_call"] * st["rays"], st["gpu_us_per_call"] * st["rays"], st["trips_per_call"] * st["rays"]]
    d = [a - b for a, b in zip(now, _last)]
    _last[:] = now
    return " | on the GPU %.2f us, %.1f trips per call (%.2f us a trip)" % (d[2] / max(d[0], 1), d[3] / max(d[0], 1), d[2] / max(d[3], 1))


n = 16 * 2000
px = rng.integers(0, w * h, n)
fx = (px % w + 0.5) / w * 2.0 - 1.0
fy = 1.0 - (px // w + 0.5) / h * 2.0
fwd = np.array(look, dtype=np.float64) - np.array(eye, dtype=np.float64)
fwd /= np.linalg.norm(fwd)
right = np.cross(fwd, [0.0, 1.0, 0.0])
right /= np.linalg.norm(right)
up = np.cross(right, fwd)
th = np.tan(np.radians(fov) / 2.0)
dirs = fwd[None, :] + (fx * th * w / h)[:, None] * right[None, :] + (fy * th)[:, None] * up[None, :]
dirs /= np.linalg.norm(dirs, axis=1)[:, None]
rays = np.zeros(n, dtype=T.RAY_DTYPE)
rays["origin"] = np.array(eye, dtype=np.float32)
rays["direction"] = dirs.astype(np.float32)
rays["tmax"] = 3.4028234663852886e38
want, ms = sc.traverse_batch(rays, sem=3)
print("traverse_batch: %d rays, %.3f ms" % (n, ms), flush=True)
for threads in (1, 2, 4, 8, 16, 32, 64):
    m = n if threads >= 8 else n // 8
    sc.traverse_threads(rays[:256], threads=threads, sem=3)
    gpu_side()
    got, secs, starts = sc.traverse_threads(rays[:m], threads=threads, sem=3)
    print("threads %3d: %.4f Mrays/s, %.2f us per ray and thread, %d service starts, equal %s%s" % (
        threads, m / secs / 1e6, secs / m * threads * 1e6, starts, bool((got == want[:m]).all()), gpu_side()), flush=True)
# the reference's own pattern (src/rt_cpu/rt_cpu.rs:35: `(0..w*h).into_par_iter()` - rayon hands every worker a CONTIGUOUS run of
# pixel indices): thread k walks pixels k m .. (k + 1) m - 1 of the frame's middle rows, neighbours one after the other
def pixel_rays(idx):
    fx = (idx % w + 0.5) / w * 2.0 - 1.0
    fy = 1.0 - (idx // w + 0.5) / h * 2.0
    dd = fwd[None, :] + (fx * th * w / h)[:, None] * right[None, :] + (fy * th)[:, None] * up[None, :]
    dd /= np.linalg.norm(dd, axis=1)[:, None]
    r = np.zeros(len(idx), dtype=T.RAY_DTYPE)
    r["origin"] = np.array(eye, dtype=np.float32)
    r["direction"] = dd.astype(np.float32)
    r["tmax"] = 3.4028234663852886e38
    return r


for threads in (1, 4, 16):
    m = 2000
    runs = pixel_rays(np.arange(threads * m) + (h // 2) * w)
    dealt = np.empty_like(runs)
    for k in range(threads):
        dealt[k::threads] = runs[k * m:(k + 1) * m]   # (the helper gives thread k the rays k, k + threads, ...)
    want_r, _ = sc.traverse_batch(dealt, sem=3)
    sc.traverse_threads(dealt[:256], threads=threads, sem=3)
    gpu_side()
    got, secs, _ = sc.traverse_threads(dealt, threads=threads, sem=3)
    print("threads %3d, each a contiguous run of %d pixels (rayon's split): %.4f Mrays/s, %.2f us per ray and thread, equal %s%s" % (
        threads, m, len(dealt) / secs / 1e6, secs / len(dealt) * threads * 1e6, bool((got == want_r).all()), gpu_side()), flush=True)
# the same ray over and over from one thread: every node and triangle of its walk is in the nearest cache - what is left
# of a call is instruction issue of the walk plus the trip through host memory
same = np.repeat(rays[:1], 4000)
gpu_side()
got, secs, _ = sc.traverse_threads(same, threads=1, sem=3)
print("one thread, ONE ray 4000 times: %.2f us per call%s" % (secs / 4000 * 1e6, gpu_side()), flush=True)
two = np.tile(rays[:2], 2000)
got, secs, _ = sc.traverse_threads(two, threads=1, sem=3)
print("one thread, TWO rays alternating: %.2f us per call%s" % (secs / 4000 * 1e6, gpu_side()), flush=True)
for cycle in (16, 128, 1024):
    many = np.tile(rays[:cycle], 4096 // cycle)
    got, secs, _ = sc.traverse_threads(many, threads=1, sem=3)
    print("one thread, %d rays round and round: %.2f us per call%s" % (cycle, secs / len(many) * 1e6, gpu_side()), flush=True)
nul = same.copy()
nul["tmax"] = -1.0   # nothing lies in [0, -1]: the walk ends at the root (the trip through host memory alone)
got, secs, _ = sc.traverse_threads(nul, threads=1, sem=3)
print("one thread, a ray that ends at the root: %.2f us per call" % (secs / 4000 * 1e6), flush=True)
# idle service waves beside a primary frame: frame time with the service up (a caller thread keeps it alive) and down
import threading  # noqa: E402
time.sleep(0.2)
base = [sc.bench_primary(view, w, h, sem=3, warmup=3, frames=30) for _ in range(3)]
stop = False


def keep_alive():
    while not stop:
        sc.traverse(eye, dirs[0], sem=3)
        time.sleep(0.005)


t = threading.Thread(target=keep_alive)
t.start()
time.sleep(0.05)
withsvc = [sc.bench_primary(view, w, h, sem=3, warmup=3, frames=30) for _ in range(3)]
stop = True
t.join()
# ... and the other way round: what a busy GPU does to the callers (clocks up, SIMDs shared)
stop = False
busy_frames = [0]


def keep_busy():
    while not stop:
        sc.bench_primary(view, w, h, sem=3, warmup=0, frames=50)
        busy_frames[0] += 50


t = threading.Thread(target=keep_busy)
t.start()
time.sleep(0.1)
for threads in (1, 16):
    m = n if threads >= 8 else n // 8
    got, secs, starts = sc.traverse_threads(rays[:m], threads=threads, sem=3)
    print("threads %3d beside back-to-back primary frames: %.4f Mrays/s, %.2f us per ray and thread, equal %s" % (
        threads, m / secs / 1e6, secs / m * threads * 1e6, bool((got == want[:m]).all())), flush=True)
stop = True
t.join()
print("primary frame alone: min %.4f mean %.4f ms; with the ray service resident (one caller every 5 ms): min %.4f mean %.4f ms" % (
    min(b[0] for b in base), sum(b[1] for b in base) / 3, min(b[0] for b in withsvc), sum(b[1] for b in withsvc) / 3))
st = sc.service_stats()
print("ray service: %d rays, %d starts: %.2f us per call from post to answer (all thread counts above), of which %.2f us between admission "
      "and answer on the GPU (%.1f trips of the walk)" % (st["rays"], st["starts"], st["us_per_call"], st["gpu_us_per_call"], st["trips_per_call"]))
sc.close()
