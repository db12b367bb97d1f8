"""How exposed the results are to the two things nobody can check without running obvhs ("parity unpinned"):
(1) the TRX_SEM_CPU preset (NODE_RCP | TIE_FIRST) being a recollection of obvhs, while the only in-tree normative text
is the HLSL (TRX_SEM_HLSL): rays of each BASELINE frame whose (t, prim) differ between the two;
(2) the AO directions using an explicit sin / cos where the reference calls its platform's: AO rays whose hit differs
when this platform's libm sinf / cosf is used instead (since round 4 the explicit evaluation is glibc's own binary64
algorithm, so on a glibc host this column is 0), and when a correctly rounded sin / cos is (another C library).
Oracle only (CPU), full BASELINE sizes by default.  usage: python tests/analysis/semantics_exposure.py [--scale 4] [--json out]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tray_racing_amd as T  # noqa: E402
from oracle import binding as O  # noqa: E402

CONFIGS = [  # name, scene, width, height, tlas   (BASELINE.json configs[0..4])
    ("cfg0 demoscene 512x1080", "demoscene", 512, 1080, False),
    ("cfg1 kitchen-class 1920x1080", "kitchen", 1920, 1080, False),
    ("cfg2 bistro-class 1920x1080", "bistro", 1920, 1080, False),
    ("cfg3 hairball-class 1920x1080", "hairball", 1920, 1080, False),
    ("cfg4 san-miguel-class TLAS 3840x2160", "san_miguel", 3840, 2160, True),
]


def differ(a, b):
    return (a["t"].view(np.uint32) != b["t"].view(np.uint32)) | (a["prim"] != b["prim"])


def measure(scene, w, h, tlas, tris=0):
    verts, counts = T.gen_scene(scene, tris, 1)
    flat = T.flat_build(verts, counts, use_tlas=tlas)
    eye, look, fov = T.scene_camera(scene)
    view = O.view_from_bytes(T.view_from_camera(eye, look, fov, w, h))
    osc = O.Scene.from_flat(flat)
    p_cpu, st = osc.trace_primary(view, w, h, sem=O.SEM_CPU)
    p_hlsl, _ = osc.trace_primary(view, w, h, sem=O.SEM_HLSL)
    d = differ(p_cpu, p_hlsl)
    # relative difference of t where both hit
    both = np.isfinite(p_cpu["t"]) & np.isfinite(p_hlsl["t"]) & d
    rel = float(np.max(np.abs(p_cpu["t"][both] - p_hlsl["t"][both]) / p_hlsl["t"][both])) if both.any() else 0.0
    a_cpu, ast = osc.trace_ao(view, w, h, p_cpu, sem=O.SEM_CPU, frame=0, ao_eps=0.01)
    a_hlsl, _ = osc.trace_ao(view, w, h, p_cpu, sem=O.SEM_HLSL, frame=0, ao_eps=0.01)
    def against(mode):
        """AO frame with the AO directions from another sin / cos: (rays differing, ... another triangle, hit <-> miss,
        max rel. dt on the same triangle)."""
        O.set_ao_libm(mode)
        try:
            other, _ = osc.trace_ao(view, w, h, p_cpu, sem=O.SEM_CPU, frame=0, ao_eps=0.01)
        finally:
            O.set_ao_libm(0)
        dl = differ(a_cpu, other)
        flip = np.isfinite(a_cpu["t"]) != np.isfinite(other["t"])
        pc = dl & (a_cpu["prim"] != other["prim"])
        same = dl & ~pc & np.isfinite(a_cpu["t"]) & np.isfinite(other["t"])
        rel_s = float(np.max(np.abs(a_cpu["t"][same] - other["t"][same]) / a_cpu["t"][same])) if same.any() else 0.0
        return int(dl.sum()), int(pc.sum()), int(flip.sum()), rel_s
    libm = against(1)   # this platform's sinf / cosf (glibc: the explicit evaluation IS its algorithm -> 0 expected)
    cr = against(2)     # a correctly rounded sin / cos (what a CORE-MATH based C library returns)
    return {
        "tris": int(flat.n_tris), "primary_rays": int(w * h), "primary_hits": int(st.n_hits),
        "primary_differ_hlsl_vs_cpu": int(d.sum()),
        "primary_differ_prim_only_ties": int((d & (p_cpu["t"].view(np.uint32) == p_hlsl["t"].view(np.uint32))).sum()),
        "primary_max_rel_dt": rel,
        "ao_rays": int(ast.n_rays),
        "ao_differ_hlsl_vs_cpu": int(differ(a_cpu, a_hlsl).sum()),
        "ao_differ_libm_vs_explicit_sincos": libm[0],
        "ao_libm_prim_changes": libm[1],
        "ao_libm_max_rel_dt_same_triangle": libm[3],
        "ao_libm_hit_miss_flips": libm[2],
        "ao_differ_correctly_rounded_vs_explicit_sincos": cr[0],
        "ao_cr_prim_changes": cr[1],
        "ao_cr_max_rel_dt_same_triangle": cr[3],
        "ao_cr_hit_miss_flips": cr[2],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=1, help="divide image sides by this (CPU test suite uses 8)")
    ap.add_argument("--tris-scale", type=int, default=1, help="divide triangle counts by this")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    out = {}
    for label, scene, w, h, tlas in CONFIGS:
        tris = 0
        if args.tris_scale > 1:
            full = {"demoscene": 2 * 2048 * 2048, "kitchen": 56939, "bistro": 3872303, "hairball": 2880000, "san_miguel": 5075977}
            tris = max(2000, full[scene] // args.tris_scale)
        r = measure(scene, w // args.scale, h // args.scale, tlas, tris)
        out[label] = r
        print("%-40s primary %8d rays: %4d differ HLSL vs CPU preset (%d of them prim-only ties, max rel dt %.1e) | AO %8d rays: %4d differ "
              "HLSL vs CPU; vs this platform's libm sinf / cosf %d differ (%d other triangle, %d hit<->miss, max rel dt %.1e); vs a "
              "correctly rounded sin / cos %d differ (%d other triangle, %d hit<->miss, max rel dt on the same triangle %.1e)" % (
                  label, r["primary_rays"], r["primary_differ_hlsl_vs_cpu"], r["primary_differ_prim_only_ties"],
                  r["primary_max_rel_dt"], r["ao_rays"], r["ao_differ_hlsl_vs_cpu"], r["ao_differ_libm_vs_explicit_sincos"],
                  r["ao_libm_prim_changes"], r["ao_libm_hit_miss_flips"], r["ao_libm_max_rel_dt_same_triangle"],
                  r["ao_differ_correctly_rounded_vs_explicit_sincos"], r["ao_cr_prim_changes"], r["ao_cr_hit_miss_flips"],
                  r["ao_cr_max_rel_dt_same_triangle"]), flush=True)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
