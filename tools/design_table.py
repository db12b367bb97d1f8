"""Prints the per-config table of DESIGN.md section 5 from profiles/rNN_traffic_<config>.json (tools/profile_summary.py).
usage: python tools/design_table.py [--round N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = int(sys.argv[sys.argv.index("--round") + 1]) if "--round" in sys.argv else 5
ROWS = [("primary_bistro", "bistro-class primary 1080p (configs[2])"), ("primary_bistro_dense", "dense bistro-class primary"),
        ("primary_hairball", "hairball-class primary"), ("ao_bistro", "bistro-class AO pass (1.94 M rays)"),
        ("ao_hairball", "hairball-class AO pass (0.59 M rays)"), ("ao4_hairball", "hairball-class 4 spp, one launch (configs[3])"),
        ("tlas_san_miguel_4k", "san-miguel-class TLAS primary 3840×2160 (configs[4], one GPU)"), ("rays_bistro", "2 M random rays, bistro-class")]
print("| config | kernel | ms | Mrays/s | nodes / tris per ray | requested GB/s | HBM GB/s measured (of 8 TB/s) | L1 / L2 hit | VALU issue | waitcnt / issue-stall |")
print("|---|---|---|---|---|---|---|---|---|---|")
for cfg, label in ROWS:
    path = os.path.join(ROOT, "profiles", "r%02d_traffic_%s.json" % (ROUND, cfg))
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    k = d["kernel"].split("k_trace")[1].split("(")[0].replace(" ", "")
    w = d["wave_cycle_split"]
    ms = d["kernel_ms_rocprof_stats_avg"]
    print("| %s | `k_trace%s` | %.3f | %s | %.1f / %.1f | %s | %s (%s %%) | %.2f / %.2f | %.2f | %.2f / %.2f |" % (
        label, k, ms, "{:,.0f}".format(d["rays_per_launch"] / ms / 1e3).replace(",", " "), d["nodes_per_ray"], d["tris_per_ray"],
        "{:,.0f}".format(d["requested_gbs"]).replace(",", " "), "{:,.0f}".format(d["hbm_gbs_measured"]).replace(",", " "),
        ("%.1f" % (100 * d["hbm_frac_of_8TBs"])).rstrip("0").rstrip("."), d["l1_hit_rate"], d["l2_hit_rate"], d["valu_issue_frac"],
        w["waitcnt"], w["issue_stall"]))
