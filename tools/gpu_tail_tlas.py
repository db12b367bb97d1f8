import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import tray_racing_amd as T
from tray_racing_amd import _lib as L
lib = L.load()
cases = (("san_miguel", True, 3840, 2160), ("san_miguel", False, 3840, 2160), ("hairball", False, 1920, 1080))
if len(sys.argv) > 1:   # scene[:tlas[:WxH]] ...
    cases = []
    for a in sys.argv[1:]:
        f = a.split(":")
        wh = f[2].split("x") if len(f) > 2 else ("1920", "1080")
        cases.append((f[0], len(f) > 1 and f[1] == "1", int(wh[0]), int(wh[1])))
for name, tlas, w, h in cases:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts, use_tlas=tlas)
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
    print("%s tlas=%s AO %dx%d: %d waves, pass %.1f us | wave end p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
        name, tlas, w, h, n.value, end.max(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99), end.max()))
    ts = np.linspace(0, end.max(), 21)
    print("   alive at 0..100%% (5%% steps): %s" % [int(((start <= x) & (end > x)).sum()) for x in ts], flush=True)
    sc.close()
