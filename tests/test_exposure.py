"""What "parity unpinned" can cost, measured on the oracle (CPU): tools/semantics_exposure.py at reduced size.
The full-size counts are committed as profiles/r02_semantics_exposure.{txt,json} and quoted in DESIGN.md section 3."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("semantics_exposure", os.path.join(ROOT, "tools", "semantics_exposure.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("scene,w,h,tlas,tris", [("kitchen", 480, 270, False, 0), ("bistro", 480, 270, False, 200000),
                                                 ("san_miguel", 480, 270, True, 150000)])
def test_presets_and_libm_only_move_razor_edges(trx, orc, scene, w, h, tlas, tris):
    r = _tool().measure(scene, w, h, tlas, tris)
    n, na = r["primary_rays"], max(r["ao_rays"], 1)
    # HLSL text vs the recalled CPU preset: a handful of rays, nearly all exact ties between neighbouring triangles
    assert r["primary_differ_hlsl_vs_cpu"] < 2e-3 * n
    assert r["primary_max_rel_dt"] < 1e-5
    assert r["ao_differ_hlsl_vs_cpu"] < 1e-2 * na
    # libm sin / cos instead of the explicit evaluation: directions move by an ulp, so t moves in its last bits on a
    # few percent of the AO rays, inside the tolerance; another triangle only at razor edges; never hit <-> miss
    assert r["ao_differ_libm_vs_explicit_sincos"] < 0.25 * na
    assert r["ao_libm_max_rel_dt_same_triangle"] < 1e-4   # grazing hits amplify the ulp: see the full-size report
    assert r["ao_libm_prim_changes"] < 1e-3 * na
    assert r["ao_libm_hit_miss_flips"] <= 2


def test_committed_full_size_report_is_consistent():
    path = os.path.join(ROOT, "profiles", "r02_semantics_exposure.json")
    rep = json.load(open(path))
    assert len(rep) == 5
    for label, r in rep.items():
        assert r["primary_differ_hlsl_vs_cpu"] < 1e-3 * r["primary_rays"], label
        assert r["primary_max_rel_dt"] < 1e-5, label
        # AO rays under another platform's sin / cos: within 1e-5 on configs 0-3; the worst of the 8.3 M AO rays of the
        # 4K frame (a grazing hit) moves by 5.7e-5 — the north_star's 1e-5 on t is a statement about primary rays
        assert r["ao_libm_max_rel_dt_same_triangle"] < 1e-4, label
        assert r["ao_libm_hit_miss_flips"] == 0, label
