"""Stage times of the ploc_cwbvh build with the BVH2 stage on the host cores and on the GPU (development aid)."""
import os
import sys
import time

import numpy as np

os.environ["TRX_BUILD_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

lib = T.load()
for name in sys.argv[1:] or ["bistro"]:
    verts, _ = T.gen_scene(name, 0, 1)
    counts = np.array([verts.shape[0]], dtype=np.uint64)
    for ratio in (0.0, 0.15):
        bp = T.build_params(reinsertion_batch_ratio=ratio)
        for dev in (-1, 0):
            lib.trx_set_build_device(dev)
            t0 = time.time()
            flat = T.flat_build_params(verts, counts, bp)
            print("%s: %d triangles, reinsertion %.2f, BVH2 stage on %s: %.2f s in all, %d nodes" % (
                name, verts.shape[0], ratio, "the GPU" if dev >= 0 else "the host cores", time.time() - t0, flat.n_nodes), flush=True)
lib.trx_set_build_device(-1)
