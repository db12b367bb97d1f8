"""PCIe-inclusive rate of the host-buffer entry point (development aid / DESIGN.md section 8): trx_trace_primary copies the
16.6 MB hit buffer of a 1080p frame to pageable host memory after the kernel; wall clock per call against kernel time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

w, h = 1920, 1080
verts, counts = T.gen_scene("bistro", 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera("bistro")
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
for _ in range(5):
    sc.trace_primary(view, w, h, sem=3)
ts, ks = [], []
for _ in range(30):
    t0 = time.perf_counter()
    hits, ms = sc.trace_primary(view, w, h, sem=3)
    ts.append((time.perf_counter() - t0) * 1e3)
    ks.append(ms)
mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=5, frames=30)
print("host-buffer trx_trace_primary, 1080p bistro-class: wall clock per call min %.3f ms mean %.3f ms (%.0f Mrays/s at the mean); "
      "reported event time min %.3f ms; device-resident kernel min %.3f ms mean %.3f ms" % (
          min(ts), sum(ts) / len(ts), w * h / (sum(ts) / len(ts)) / 1e3, min(ks), mn, mean))
sc.close()
