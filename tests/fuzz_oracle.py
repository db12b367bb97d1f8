"""Randomised check of the oracle's CWBVH traversal against the BVH-independent brute force
(every ray against every triangle) on random scenes, builds and rays.  Test infrastructure;
not collected by pytest — tests/test_oracle.py runs a fixed-seed slice.

The two agree bit for bit except where the reference algorithm itself is BVH-dependent: the slab
test of rt_gpu_software_query.hlsl:281-291 rounds to nearest, so a node whose box face coincides
with its triangle (axis-aligned walls) can be culled by one ulp while it holds a hit a few ulps
closer than the current one, or — at a silhouette edge — the only hit.  The BVH answer is then
never CLOSER than brute force, and when both hit, further by a few ulps at most.

    python tests/fuzz_oracle.py --minutes 5 [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

SCENES = ("cornell", "kitchen", "bistro", "bistro_dense", "hairball", "san_miguel", "soup", "demoscene")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run(minutes=1.0, seed=1, max_cases=1 << 30, n_rays=3000, verbose=True):
    import tray_racing_amd as T
    from helpers import random_rays
    from oracle import binding as O
    rng = np.random.default_rng(seed)
    t_end = time.time() + 60.0 * minutes
    stats = dict(cases=0, rays=0, differing=0, missed=0, closer=0, worst_rel=0.0, tie_index=0)
    while time.time() < t_end and stats["cases"] < max_cases:
        name = SCENES[int(rng.integers(len(SCENES)))]
        n = 0 if name == "cornell" else int(rng.choice([1, 3, 40, 700, 3000]))
        s = int(rng.integers(1, 1 << 30))
        tlas, sem = bool(rng.integers(2)), int(rng.integers(8))
        verts, counts = T.gen_scene(name, n, s)
        split = float(rng.choice([0.0, 0.0, 0.3, 1.0]))   # pre-splitting: clipped references, triangles repeated
        flat = T.flat_build(verts, counts, use_tlas=tlas, max_prims_per_leaf=int(rng.integers(1, 4)), split=split)
        osc = O.Scene.from_flat(flat)
        if split == 0.0:   # the validator wants whole triangles inside their leaf boxes
            assert osc.validate()[0] == 0
        else:
            assert set(flat.tri_source.tolist()) == set(range(verts.shape[0]))
        rr = random_rays(T, flat, n_rays, s)
        got, _ = osc.trace_rays(rr, sem=sem)
        bf = osc.brute_rays(rr, sem=sem)
        # the node test clamps box entry to 1e-4 (rt_gpu_software_query.hlsl:275,288): nearer hits are invisible by design
        far = bf["t"] >= 2e-4
        d = np.flatnonzero((bits(got["t"]) != bits(bf["t"])) & far)
        stats["cases"] += 1
        stats["rays"] += int(far.sum())
        stats["tie_index"] += int(((got["prim"] != bf["prim"]) & far & (bits(got["t"]) == bits(bf["t"]))).sum())
        if d.size:
            g, b = got["t"][d].astype(np.float64), bf["t"][d].astype(np.float64)
            hit = np.isfinite(g)
            stats["differing"] += int(d.size)
            stats["missed"] += int((~hit).sum())
            stats["closer"] += int((g < b).sum())
            if hit.any():
                stats["worst_rel"] = max(stats["worst_rel"], float(((g[hit] - b[hit]) / b[hit]).max()))
            if verbose:
                i = d[0]
                print("differs: %s n=%d seed=%d tlas=%s sem=%d split=%.1f ray %d: bvh %s brute force %s" % (
                    name, n, s, tlas, sem, split, i, got[i], bf[i]), flush=True)
    T.flat_build(verts[:1], split=0.0)
    print(stats, flush=True)
    return stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    st = run(a.minutes, a.seed)
    sys.exit(1 if st["closer"] or st["worst_rel"] > 1e-6 else 0)
