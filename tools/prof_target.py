"""Profiling target: builds the bench workload once and replays a few frames (for rocprofv3 passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
sem = int(sys.argv[5]) if len(sys.argv) > 5 else 3
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts)
eye, look, fov = T.scene_camera(name)
view = T.view_from_camera(eye, look, fov, w, h)
sc = T.Scene(flat)
mn, mean = sc.bench_primary(view, w, h, sem=sem, warmup=2, frames=frames)
print("min %.3f ms mean %.3f ms  %.1f Mrays/s" % (mn, mean, w * h / mn / 1e3))
sc.close()
