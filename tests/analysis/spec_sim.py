"""Development aid (CPU only, oracle side): trips of a lone ray's walk when up to G pending nodes are tested per trip
(tests/analysis/spec_sim.c has the model).  usage: python tests/analysis/spec_sim.py [scene] [tris] [width] [height]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tray_racing_amd as T  # noqa: E402
from oracle import binding as O  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "hairball"
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 0
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
so = "/tmp/spec_sim.so"
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-ffp-contract=off", os.path.join(ROOT, "tests", "analysis", "spec_sim.c"),
                       "-o", so, "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
O.load()
sim = C.CDLL(so)
verts, counts = T.gen_scene(scene, tris, 1)
flat = T.flat_build(verts, counts, use_tlas=False, preset="medium_build")
eye, look, fov = T.scene_camera(scene)
view = T.view_from_camera(eye, look, fov, w, h)
osc = O.Scene.from_flat(flat)
ov = O.view_from_bytes(view)
prim, st = osc.trace_primary(ov, w, h, sem=O.SEM_CPU)
rays = np.zeros(w * h, dtype=T.RAY_DTYPE)
sim.spec_ao_rays.restype = C.c_uint64
n = sim.spec_ao_rays(C.byref(osc.c), C.byref(ov), w, h, prim.ctypes.data_as(C.c_void_p), 0, C.c_float(0.01), rays.ctypes.data_as(C.c_void_p))
rays = rays[:n]
print("scene %s: %d tris, %d nodes, %d AO rays" % (scene, flat.n_tris, flat.n_nodes, n))
for cap in (2, 4, 8, 32):
    out = np.zeros((n, 9), dtype=np.uint32)
    sim.spec_sim(C.byref(osc.c), rays.ctypes.data_as(C.c_void_p), C.c_uint64(n), 3, cap, out.ctypes.data_as(C.c_void_p))
    nodes = out[:, 0]
    order = np.argsort(nodes)[::-1]
    print("records outstanding <= %d; mean nodes per ray %.1f, max %d" % (cap, nodes.mean(), nodes.max()))
    for name, sel in (("all rays", order), ("longest 300", order[:300]), ("longest 3000", order[:3000]), ("longest 30000", order[:30000])):
        o = out[sel].astype(np.float64)
        print("  %-14s nodes %7.1f | full records: trips G=1 %6.1f G=2 %6.1f G=4 %6.1f G=8 %6.1f | node records: G=1 %6.1f G=2 %6.1f G=4 %6.1f G=8 %6.1f" % (
            (name, o[:, 0].mean()) + tuple(o[:, k].mean() for k in range(1, 9))))
for cache in (64, 256):
    out = np.zeros((n, 7), dtype=np.uint32)
    sim.spec_sim_deep(C.byref(osc.c), rays.ctypes.data_as(C.c_void_p), C.c_uint64(n), 3, cache, out.ctypes.data_as(C.c_void_p))
    nodes = out[:, 0]
    order = np.argsort(nodes)[::-1]
    print("prefetcher ahead of the consumer, %d parked tests" % cache)
    for name, sel in (("all rays", order), ("longest 300", order[:300]), ("longest 3000", order[:3000]), ("longest 30000", order[:30000])):
        o = out[sel].astype(np.float64)
        print("  %-14s nodes %7.1f | trips G=2 %6.1f G=4 %6.1f G=8 %6.1f | node tests G=2 %6.1f G=4 %6.1f G=8 %6.1f" % (
            (name, o[:, 0].mean()) + tuple(o[:, k].mean() for k in range(1, 7))))
