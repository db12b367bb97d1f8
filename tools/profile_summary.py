"""Summarises the output of tools/profile_set.sh into profiles/ (files rNN_*): the bench line, the rocprofv3 kernel
stats of the bench command, and per config one traffic_<config>.json (requested GB/s, measured HBM GB/s, L1 / L2 hit
rates, VALU issue, wave-cycle split) plus the counter CSV rows of the traversal kernel it was computed from.
(One script for every round: rounds 2-4 each carried a copy of it with the tag changed.)
usage: python tools/profile_summary.py --round N [gpurun_out/prof_rNN] [--dry]"""
import ast
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
ROUND = 6
if "--round" in argv:
    k = argv.index("--round")
    ROUND = int(argv[k + 1])
    del argv[k:k + 2]
args = [a for a in argv if not a.startswith("--")]
dry = "--dry" in argv
TAG = "r%02d" % ROUND
src = args[0] if args else os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
dst = os.path.join(ROOT, "profiles")
NODE_B, TRI_B, HIT_B = 80, 48, 8
SIMDS, CLOCK_GHZ, VALU_CYCLES = 1024, 2.4, 2.0
HBM_PEAK = 8000.0


def newest(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def kernel_rows(path):
    """Rows of the traversal kernels (k_trace<...>, non-counting) of one counter_collection.csv."""
    rows = list(csv.DictReader(open(path)))
    return [r for r in rows if "k_trace<" in r["Kernel_Name"] and not r["Kernel_Name"].rstrip(">").endswith("true")]


def summarise(cfg):
    log = os.path.join(src, cfg + ".stats.log")
    if not os.path.exists(log):
        return None
    info = None
    for line in open(log):
        if line.startswith("PROF_CONFIG "):
            info = ast.literal_eval(line[len("PROF_CONFIG "):].strip())
    if not info:
        return None
    ctr = defaultdict(lambda: [0.0, 0])
    durs = []
    kept = []
    kernel_name = None
    for i in range(1, 10):
        f = newest("%s/pmc%d/**/*counter_collection.csv" % (cfg, i))
        if not f:
            continue
        rows = kernel_rows(f)
        if not rows:
            continue
        # the dominant traversal kernel of the config = the one with the most dispatches
        names = defaultdict(int)
        for r in rows:
            names[r["Kernel_Name"]] += 1
        kernel_name = max(names, key=names.get)
        rows = [r for r in rows if r["Kernel_Name"] == kernel_name]
        first = min(int(r["Dispatch_Id"]) for r in rows)
        seen = {}
        for r in rows:
            if int(r["Dispatch_Id"]) < first + 2:   # warm-up launches (cold tile order, cold caches)
                continue
            c = ctr[r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
            seen[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            kept.append(r)
        if i == 1:
            durs = sorted(seen.values())
    m = {k: v[0] / v[1] for k, v in ctr.items() if v[1]}
    # kernel time without counters: the --stats pass
    ks = newest("%s/stats/**/*kernel_stats.csv" % cfg)
    stats_ms = None
    if ks:
        for r in csv.DictReader(open(ks)):
            if kernel_name and r["Name"] == kernel_name:
                stats_ms = float(r["AverageNs"]) * 1e-6
    ms = info.get("ms_min")
    rays = info["rays"]
    req = NODE_B * info.get("n_node", 0) + TRI_B * info.get("n_tri", 0) + (HIT_B + info.get("ray_bytes", 0)) * rays
    out = {
        "config": cfg, "kernel": kernel_name, "lib_sha16": info.get("lib_sha16"), "scene": info["scene"], "mode": info["mode"],
        "image": [info["width"], info["height"]], "tlas": info["tlas"], "tris": info["tris"], "nodes": info["nodes"],
        "rays_per_launch": rays,
        "nodes_per_ray": round(info.get("n_node", 0) / max(rays, 1), 2),
        "tris_per_ray": round(info.get("n_tri", 0) / max(rays, 1), 2),
        "kernel_ms_hip_events_min": round(ms, 4) if ms else None,
        "kernel_ms_rocprof_stats_avg": round(stats_ms, 4) if stats_ms else None,
        "mrays_per_s": round(rays / ms / 1e3, 1) if ms else None,
        "requested_bytes_per_launch": int(req),
        "requested_gbs": round(req / (ms * 1e-3) / 1e9, 1) if ms else None,
    }
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        hbm = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        out["hbm_bytes_per_launch"] = int(hbm)
        out["hbm_gbs_measured"] = round(hbm / (ms * 1e-3) / 1e9, 1) if ms else None
        out["hbm_frac_of_8TBs"] = round(hbm / (ms * 1e-3) / 1e9 / HBM_PEAK, 4) if ms else None
        out["FETCH_SIZE_KB"], out["WRITE_SIZE_KB"] = round(m["FETCH_SIZE"], 1), round(m["WRITE_SIZE"], 1)
        out["correction"] = "gfx950: FETCH_SIZE doubled (128-B fabric requests tallied at 64 B); separate --pmc passes"
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in m and "TCP_TCC_READ_REQ_sum" in m:
        out["l1_hit_rate"] = round(1.0 - m["TCP_TCC_READ_REQ_sum"] / max(m["TCP_TOTAL_CACHE_ACCESSES_sum"], 1), 4)
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
        out["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1), 4)
    if "SQ_INSTS_VALU" in m and ms:
        out["valu_wave_insts_per_launch"] = int(m["SQ_INSTS_VALU"])
        out["valu_issue_frac"] = round(m["SQ_INSTS_VALU"] / (ms * 1e-3) / 1e9 / (SIMDS * CLOCK_GHZ / VALU_CYCLES), 4)
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        out["wave_cycle_split"] = {k: round(m[c] / wc, 3) for k, c in (
            ("issuing", "SQ_ACTIVE_INST_ANY"), ("issuing_valu", "SQ_ACTIVE_INST_VALU"), ("waitcnt", "SQ_WAIT_ANY"),
            ("issue_stall", "SQ_WAIT_INST_ANY")) if c in m}
    if durs:
        out["kernel_ms_under_counters"] = round(sum(durs[: max(1, len(durs) // 2)]) / max(1, len(durs) // 2), 4)
    out["counters_per_launch"] = {k: round(v, 1) for k, v in sorted(m.items())}
    return out, kept


def main():
    results = []
    for cfg in ("primary_bistro", "primary_bistro_dense", "primary_hairball", "ao_bistro", "ao_hairball", "ao4_hairball",
                "tlas_san_miguel_4k", "rays_bistro"):
        r = summarise(cfg)
        if not r:
            continue
        out, rows = r
        results.append(out)
        print(json.dumps({k: v for k, v in out.items() if k != "counters_per_launch"}))
        if not dry:
            json.dump(out, open(os.path.join(dst, "%s_traffic_%s.json" % (TAG, cfg)), "w"), indent=1)
            if rows:
                w = csv.DictWriter(open(os.path.join(dst, "%s_pmc_%s.csv" % (TAG, cfg)), "w"), fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows)
            ks = newest("%s/stats/**/*kernel_stats.csv" % out["config"])
            if ks:
                shutil.copy(ks, os.path.join(dst, "%s_kernel_stats_%s.csv" % (TAG, out["config"])))
    if dry:
        return
    # The bench line of the round: copied only when it is NEWER than the committed one and comes from the library the
    # summarised configs ran (round 5 shipped a stale line: a later partial re-run of this script copied the bench line of
    # an older build over the final one).
    b = os.path.join(src, "bench_default.json")
    target = os.path.join(dst, "%s_bench_default.json" % TAG)
    libs = {r.get("lib_sha16") for r in results if r.get("lib_sha16")}
    if os.path.exists(b):
        line = open(b).read().strip().splitlines()[-1]
        lib = (json.loads(line).get("build") or {}).get("lib_sha16")
        if os.path.exists(target) and os.path.getmtime(b) <= os.path.getmtime(target):
            print("bench line NOT copied: %s is not newer than %s" % (b, target))
        elif libs and lib not in libs:
            print("bench line NOT copied: it ran library %s, the configs summarised here ran %s" % (lib, sorted(libs)))
        else:
            open(target, "w").write(line + "\n")
    if len(libs) > 1:
        print("WARNING: the summarised configs ran %d different libraries: %s" % (len(libs), sorted(libs)))
    ks = newest("stats_bench/**/*kernel_stats.csv")
    if ks and os.path.exists(b) and os.path.getmtime(ks) >= os.path.getmtime(b):
        shutil.copy(ks, os.path.join(dst, "%s_kernel_stats_bench.csv" % TAG))
    # the bench's live counters read profiles/traffic.json only as a fall-back; keep it in step with this round
    pb = [r for r in results if r["config"] == "primary_bistro"]
    if pb and "hbm_bytes_per_launch" in pb[0]:
        t = pb[0]
        json.dump({"hbm_bytes_per_launch": t["hbm_bytes_per_launch"], "kernel": t["kernel"],
                   "FETCH_SIZE_KB_per_launch": t["FETCH_SIZE_KB"], "WRITE_SIZE_KB_per_launch": t["WRITE_SIZE_KB"],
                   "correction": t["correction"], "algorithmic_bytes_per_launch": t["requested_bytes_per_launch"],
                   "valu_wave_insts_per_launch": t.get("valu_wave_insts_per_launch"),
                   "command": "tools/profile_set.sh (rocprofv3 --kernel-trace --pmc <group> -- python3 tools/prof_config.py primary_bistro 6)",
                   "round": ROUND}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    # table for DESIGN.md
    print("\n| config | kernel ms | Mrays/s | nodes/ray | tris/ray | requested GB/s | HBM GB/s (measured) | L1 hit | L2 hit | VALU issue | waitcnt / issue stall |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in results:
        ws = r.get("wave_cycle_split", {})
        print("| %s | %s | %s | %s | %s | %s | %s | %s | %s | %s | %s / %s |" % (
            r["config"], r["kernel_ms_hip_events_min"], r["mrays_per_s"], r["nodes_per_ray"], r["tris_per_ray"],
            r["requested_gbs"], r.get("hbm_gbs_measured"), r.get("l1_hit_rate"), r.get("l2_hit_rate"),
            r.get("valu_issue_frac"), ws.get("waitcnt"), ws.get("issue_stall")))


if __name__ == "__main__":
    main()
