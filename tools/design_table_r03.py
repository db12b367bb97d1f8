"""Prints the per-config table of DESIGN.md section 5 from profiles/r03_traffic_<config>.json (round-2 times in brackets)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = {"primary_bistro": "bistro-class primary 1080p (configs[2])", "primary_bistro_dense": "dense bistro-class primary 1080p",
         "primary_hairball": "hairball-class primary 1080p", "ao_bistro": "bistro-class AO pass (1.94 M rays)",
         "ao_hairball": "hairball-class AO pass (0.59 M rays; configs[3])",
         "tlas_san_miguel_4k": "san-miguel-class TLAS primary 3840x2160 (configs[4], one GPU; re-braided TLAS)",
         "rays_bistro": "2 M random rays through the bistro-class scene"}
r2 = {"primary_bistro": 0.462, "primary_bistro_dense": 0.826, "primary_hairball": 1.362, "ao_bistro": 1.066, "ao_hairball": 1.162,
      "tlas_san_miguel_4k": 4.698, "rays_bistro": 0.464}
for c in names:
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_traffic_%s.json" % c)))
    k = d["counters_per_launch"]
    kern = d["kernel"].split("k_trace")[1].split("(")[0].replace(" ", "")
    ws = d["wave_cycle_split"]
    print("| %s | `k_trace%s` | %.3f (%.3f) | %d | %.1f / %.1f | %d | %d (%d %%) | %.3f | %.2f | %.2f | %.2f | %.2f / %.2f | %d / %d |" % (
        names[c], kern, d["kernel_ms_hip_events_min"], r2[c], round(d["mrays_per_s"]), d["nodes_per_ray"], d["tris_per_ray"],
        round(d["requested_gbs"]), round(d["hbm_gbs_measured"]), round(100 * d["hbm_frac_of_8TBs"]), d["l1_hit_rate"], d["l2_hit_rate"],
        k["TA_TA_BUSY_sum"] / (256 * k["GRBM_GUI_ACTIVE"] / 8), d["valu_issue_frac"], ws["waitcnt"], ws["issue_stall"],
        round(k.get("VALUBusy", 0)), round(k.get("VALUUtilization", 0))))
