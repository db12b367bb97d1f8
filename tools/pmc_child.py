"""Target of bench.py's live rocprofv3 --pmc passes: replays a few primary frames of a scene that the parent
dumped (flat buffers + view), so the counters are those of the very workload being benchmarked.
usage (under rocprofv3): python3 tools/pmc_child.py <scene.npz> [frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tray_racing_amd as T  # noqa: E402
from tray_racing_amd import _lib as L  # noqa: E402

z = np.load(sys.argv[1])
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
flat = T.FlatScene(z["nodes"], z["tri_verts"], z["instance_offsets"], int(z["tlas_start"]), np.zeros(0, np.uint32),
                   np.zeros(1, np.uint32), instance_entry=z["instance_entry"] if "instance_entry" in z.files else None)
view = L.View.from_buffer_copy(z["view"].tobytes())
w, h, sem = int(z["width"]), int(z["height"]), int(z["sem"])
sc = T.Scene(flat)
mn, mean = sc.bench_primary(view, w, h, sem=sem, warmup=4, frames=frames)
print("min %.4f ms mean %.4f ms" % (mn, mean))
sc.close()
