"""Any-hit against closest-hit on AO-like rays (development aid): time of the two queries over the same rays."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import tray_racing_amd as T  # noqa: E402

for name in sys.argv[1:] or ["bistro"]:
    verts, counts = T.gen_scene(name, 0, 1)
    flat = T.flat_build(verts, counts)
    sc = T.Scene(flat)
    rng = np.random.default_rng(3)
    n = 2_000_000
    pts = flat.tri_verts.reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    # hemisphere rays from points on random triangles (what an AO / shadow pass casts)
    tri = flat.tri_verts[rng.integers(0, flat.n_tris, n)].reshape(n, 3, 3)
    bary = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    origin = (tri * bary[:, :, None]).sum(1)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-30
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d *= np.sign((d * nrm).sum(1, keepdims=True))
    rays = np.zeros(n, dtype=T.RAY_DTYPE)
    rays["origin"] = origin + 1e-3 * nrm
    rays["direction"] = d
    rays["tmax"] = 3.4028234663852886e38
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    hits = torch.empty(n, dtype=torch.int64, device="cuda")
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    res = {}
    for what in ("closest", "any"):
        best = 1e9
        for i in range(10):
            ev0.record()
            if what == "closest":
                sc.trace_rays_dev(d_rays.data_ptr(), n, hits.data_ptr(), sem=3)
            else:
                sc.trace_occluded_dev(d_rays.data_ptr(), n, flags.data_ptr(), sem=3)
            ev1.record()
            torch.cuda.synchronize()
            if i >= 3:
                best = min(best, ev0.elapsed_time(ev1))
        res[what] = best
    sc.check()
    occ = (hits.view(torch.int32).reshape(-1, 2)[:, 1] != -1)
    print("%s: %d hemisphere rays, %.1f %% occluded | closest-hit %.3f ms (%.0f Mrays/s) | any-hit %.3f ms (%.0f Mrays/s) | "
          "flags agree: %s" % (name, n, 100.0 * occ.float().mean().item(), res["closest"], n / res["closest"] / 1e3,
                               res["any"], n / res["any"] / 1e3, bool((flags.bool() == occ).all())), flush=True)
    sc.close()
