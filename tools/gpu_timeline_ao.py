"""Per-wave lifetimes of one AO pass (trx_debug_wave_timeline_ao): how much of the pass is waves waiting for its longest rays?
usage: python tools/gpu_timeline_ao.py hairball bistro   (TRX_VARIANT=<kernel variant word>)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T
from tray_racing_amd import _lib as L
lib = L.load()
lib.trx_set_kernel_variant(int(os.environ.get("TRX_VARIANT", "0"), 0))
for name in sys.argv[1:]:
    w, h = 1920, 1080
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    buf = np.zeros(8 * 8192, dtype=np.uint64)
    n = C.c_uint32()
    for _ in range(6):
        L.check(lib.trx_debug_wave_timeline_ao(sc.handle, C.byref(view), w, h, 3, buf.ctypes.data_as(C.c_void_p), 8192, C.byref(n)))
    r = buf[: 8 * n.value].reshape(-1, 8).astype(np.int64)
    t0 = r[:, 0].min()
    start, end = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0
    print("%s AO: %d waves, pass %.1f us | end p10 %.1f p50 %.1f p90 %.1f max %.1f | mean lifetime %.1f us (%.0f%%)" % (
        name, n.value, end.max(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), end.max(), (end - start).mean(), 100 * (end - start).mean() / end.max()))
    ts = np.linspace(0, end.max(), 11)
    print("   alive at 0..100%%: %s" % [int(((start <= x) & (end > x)).sum()) for x in ts])
    if r[:, 5].any():  # -DTRX_TAIL_DIAG build: the moment each wave found the queues dry
        dry, alive, age, trips = (r[:, 5] - t0) / 100.0, r[:, 6], r[:, 7] & 0xffffffff, r[:, 7] >> 32
        left = end - dry
        print("   queues dry for a wave at p10 %.0f p50 %.0f p90 %.0f us | rays it still held: mean %.1f | oldest of them %.0f trips (mean) | time to finish them: p10 %.0f p50 %.0f p90 %.0f max %.0f us" % (
            np.percentile(dry, 10), np.percentile(dry, 50), np.percentile(dry, 90), alive.mean(), age.mean(),
            np.percentile(left, 10), np.percentile(left, 50), np.percentile(left, 90), left.max()))
        print("   trips per wave: mean %.0f -> %.2f us per trip over the wave's life; in the tail (after dry) the longest-living waves: %s" % (
            trips.mean(), ((end - start) / np.maximum(trips, 1)).mean(),
            ["%.0f us, %d rays, oldest %d" % (left[i], alive[i], age[i]) for i in np.argsort(-left)[:6]]))
        if r[:, 2].any():  # thin waves (round 4): when a wave gave its last rays eight lanes each
            thin = r[:, 2] != 0
            tt, rays_t, trips_t = (r[thin, 2] - t0) / 100.0, r[thin, 3], r[thin, 4]
            dur = end[thin] - tt
            thin_trips = np.maximum(trips[thin] - trips_t, 1)
            print("   thin: %d of %d waves | entered at p10 %.0f p50 %.0f p90 %.0f us with %.1f rays (mean) after %.0f trips | thin phase lasts "
                  "p50 %.0f p90 %.0f max %.0f us = %.0f trips (mean), %.2f us per trip (normal phase of the same waves: %.2f us per trip)" % (
                      thin.sum(), len(thin), np.percentile(tt, 10), np.percentile(tt, 50), np.percentile(tt, 90), rays_t.mean(), trips_t.mean(),
                      np.percentile(dur, 50), np.percentile(dur, 90), dur.max(), thin_trips.mean(), (dur / thin_trips).mean(),
                      ((tt - start[thin]) / np.maximum(trips_t, 1)).mean()))
            longest = np.argsort(-dur)[:6]
            print("   longest thin phases: %s" % ["%.0f us, %d trips, entered with %d rays" % (dur[i], thin_trips[i], rays_t[i]) for i in longest])
    sc.close()
