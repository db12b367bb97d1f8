"""Development aid (tuning build -DTRX_SVC_PHASES, TRX_LIB=tuning_libs/svcph.so): cycles per step of the ray service walker's
trip - one caller, sixteen rays round and round (everything they touch is cached after the first round)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

w, h = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
verts, counts = T.gen_scene(name, 0, 1)
flat = T.flat_build(verts, counts, preset="medium_build")
eye, look, fov = T.scene_camera(name)
sc = T.Scene(flat)
rng = np.random.default_rng(11)
px = rng.integers(0, w * h, 16)
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
rays = np.zeros(16, dtype=T.RAY_DTYPE)
rays["origin"] = np.array(eye, dtype=np.float32)
rays["direction"] = dirs.astype(np.float32)
rays["tmax"] = 3.4028234663852886e38
got, secs, _ = sc.traverse_threads(np.tile(rays, 1000), threads=1, sem=3)
st = sc.service_stats()
print("calls %d, %.2f us per call, on the GPU %.2f us and %.1f trips per call" % (st["rays"], secs / 16000 * 1e6, st["gpu_us_per_call"], st["trips_per_call"]), flush=True)
sc.close()
