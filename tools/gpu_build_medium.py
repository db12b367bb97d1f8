"""Stage times of the default (binned SAH, medium_build) build and of the flat assembly around it (development aid)."""
import os
import sys
import time

os.environ["TRX_BUILD_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402

for name in sys.argv[1:] or ["bistro"]:
    verts, counts = T.gen_scene(name, 0, 1)
    for rep in range(2):
        t0 = time.time()
        flat = T.flat_build(verts, counts, preset="medium_build")
        print("%s: medium_build %.2f s in all, %d nodes" % (name, time.time() - t0, flat.n_nodes), flush=True)
