"""AO-pass diagnostics: counts, SIMD efficiency, time (development aid)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

lib = L.load()
for name in sys.argv[1:] or ["bistro"]:
    w, h = 1920, 1080
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    eye, look, fov = T.scene_camera(name)
    view = T.view_from_camera(eye, look, fov, w, h)
    sc = T.Scene(flat)
    prim = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    ao = torch.zeros(w * h, dtype=torch.int64, device="cuda")
    sc.trace_primary_dev(view, w, h, prim.data_ptr(), sem=3)
    torch.cuda.synchronize()
    st = L.Stats()
    L.check(lib.trx_count_ao(sc.handle, C.byref(view), w, h, L.Shard(0, 1, 0, 0), 3, 0, 0.01, C.c_void_p(prim.data_ptr()),
                             C.c_void_p(ao.data_ptr()), C.byref(st)))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for variant in (64, 48, 32, 24, 16, 8, 4):
        lib.trx_set_kernel_variant(variant)
        best = 1e9
        for i in range(12):
            ev0.record()
            sc.trace_ao_dev(view, w, h, prim.data_ptr(), ao.data_ptr(), sem=3, frame=i % 4, ao_eps=0.01)
            ev1.record()
            torch.cuda.synchronize()
            if i >= 4:
                best = min(best, ev0.elapsed_time(ev1))
        print("   refill_idle %2d: %.3f ms" % (variant, best), flush=True)
    lib.trx_set_kernel_variant(0)
    print("%s AO: %d rays, node/ray %.1f tri/ray %.1f hits %d | node SIMD eff %.3f tri eff %.3f | wave node steps %d tri rounds %d | "
          "%.3f ms = %.0f Mrays/s" % (name, st.n_rays, st.n_node / max(st.n_rays, 1), st.n_tri / max(st.n_rays, 1), st.n_hits,
                                      st.n_node / (64.0 * max(st.n_wave_node, 1)), st.n_tri / (64.0 * max(st.n_wave_tri, 1)),
                                      st.n_wave_node, st.n_wave_tri, best, st.n_rays / best / 1e3), flush=True)
    sc.close()
