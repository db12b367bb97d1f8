"""Reinsertion with whole-iteration batches (round 5): build time, collapse cost and node visits per primary ray of the
bistro-class scene for the reference-default PLOC pipeline and for the medium_build preset - the pipeline's own batching
against one batch per iteration on the host cores and on the GPU (searches as a kernel), several iteration counts; the
host and device runs of one setting must produce the same bytes.
usage: python tools/gpu_reinsert.py [scene]"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import tray_racing_amd as T
    from tray_racing_amd import _lib as L
    lib = L.load()
    name = sys.argv[1] if len(sys.argv) > 1 else "bistro"
    verts, counts = T.gen_scene(name, 0, 1)
    eye, look, fov = T.scene_camera(name)
    w, h = 1920, 1080
    view = T.view_from_camera(eye, look, fov, w, h)

    def measure(label, build):
        t0 = time.time()
        flat = build()
        dt = time.time() - t0
        sc = T.Scene(flat)
        st = sc.count_primary(view, w, h, sem=3)
        mn, mean = sc.bench_primary(view, w, h, sem=3, warmup=3, frames=20)
        sc.close()
        digest = hashlib.sha256(flat.nodes.tobytes()).hexdigest()[:12]
        print("%-58s build %.2f s  nodes %7d  visits/ray %.2f  tris/ray %.2f  frame %.4f ms  %s" % (
            label, dt, flat.n_nodes, st.n_node / st.n_rays, st.n_tri / st.n_rays, mean, digest), flush=True)
        return digest

    def settings(whole, device, ratio=None, iters=None):
        L.check(lib.trx_set_build_preset(b"medium_build"))
        L.check(lib.trx_set_build_reinsertion_batches(1 if whole else 0))
        L.check(lib.trx_set_build_device(device))
        if ratio is not None:
            L.check(lib.trx_set_build_reinsertion(float(ratio), int(iters)))

    # ---- the reference-default PLOC pipeline (ratio 0.15 from the build parameters; iterations from the process-wide setting)
    settings(False, -1)
    measure("ploc default: batches of 128, host (round 4)", lambda: T.flat_build_params(verts, counts, T.build_params(), use_tlas=False))
    settings(False, 0)
    measure("ploc default: PLOC on the device, batches of 128", lambda: T.flat_build_params(verts, counts, T.build_params(), use_tlas=False))
    for iters in (4, 8, 12):
        settings(True, -1, 0.02, iters)
        a = measure("ploc default: whole iterations x %d, host" % iters, lambda: T.flat_build_params(verts, counts, T.build_params(), use_tlas=False))
        settings(True, 0, 0.02, iters)
        b = measure("ploc default: whole iterations x %d, device" % iters, lambda: T.flat_build_params(verts, counts, T.build_params(), use_tlas=False))
        print("   host tree == device tree: %s" % (a == b), flush=True)
    # ---- the medium_build preset (binned-SAH BVH2; one candidate at a time, ratio 0.02 x 4)
    settings(False, -1)
    measure("medium_build: one at a time (round 4)", lambda: T.flat_build(verts, counts, use_tlas=False))
    for ratio, iters in ((0.02, 4), (0.02, 8), (0.05, 8), (0.1, 8)):
        settings(True, 0, ratio, iters)
        measure("medium_build: whole iterations %.2f x %d, device" % (ratio, iters), lambda: T.flat_build(verts, counts, use_tlas=False))
    settings(False, -1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
