"""Copies the output of tools/gpu_profiles.sh (gpurun_out/prof_rNN) into profiles/ and recomputes traffic.json.
usage: python tools/update_profiles.py [round=1]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]) if len(sys.argv) > 1 else 1
src = os.path.join(ROOT, "gpurun_out", "prof_r%02d" % rnd)
dst = os.path.join(ROOT, "profiles")
tag = "r%02d" % rnd
KERNEL = "k_trace<0, false, 1, false>"


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    assert f, pattern
    return f[-1]   # gpurun merges into gpurun_out/: older runs' files may still be there


for name in ("default", "streams1"):
    line = open(os.path.join(src, "bench_%s.json" % name)).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, "%s_bench_%s.json" % (tag, name)), "w").write(line + "\n")
shutil.copy(one("stats_default/**/*kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_default.csv" % tag))
shutil.copy(one("stats_s1/**/*kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_s1.csv" % tag))

avg = defaultdict(lambda: [0.0, 0])
groups = {"FETCH_SIZE": ["pmc_FETCH_SIZE"], "WRITE_SIZE": ["pmc_WRITE_SIZE"], "SQ": ["pmc_SQ_INSTS_VALU", "pmc_SQ_ACTIVE_INST_VALU"]}
for out_name, dirs in groups.items():
    rows, hdr = [], None
    for d in dirs:
        r = list(csv.reader(open(one(d + "/**/*counter_collection.csv"))))
        hdr = r[0]
        ki, ci, vi, di = hdr.index("Kernel_Name"), hdr.index("Counter_Name"), hdr.index("Counter_Value"), hdr.index("Dispatch_Id")
        body = [x for x in r[1:] if KERNEL in x[ki]]
        first = min(int(x[di]) for x in body)
        for x in body:
            rows.append(x)
            if int(x[di]) >= first + 2:   # the first launches of a process are warm-up
                avg[x[ci]][0] += float(x[vi])
                avg[x[ci]][1] += 1
    w = csv.writer(open(os.path.join(dst, "%s_pmc_%s.csv" % (tag, out_name)), "w"), quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(hdr)
    w.writerows(rows)
m = {k: v[0] / v[1] for k, v in avg.items()}
bench = json.loads(open(os.path.join(dst, "%s_bench_streams1.json" % tag)).read())
traffic = {
    "hbm_bytes_per_launch": int(round((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)),
    "kernel": "trx::k_trace<0,false,1,false> (primary, BLAS, TRX_SEM_CPU), bistro-class 1920x1080, one frame in flight",
    "FETCH_SIZE_KB_per_launch": round(m["FETCH_SIZE"], 1),
    "WRITE_SIZE_KB_per_launch": round(m["WRITE_SIZE"], 1),
    "correction": "gfx950: FETCH_SIZE counts 128-B fabric requests as 64 B, so it is doubled (MI355X_MICROARCH.md, HBM); "
                  "WRITE_SIZE is exact; the counters were collected in separate --pmc passes",
    "command": "rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 tools/prof_target.py bistro 10",
    "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
    "valu_wave_insts_per_launch": int(round(m["SQ_INSTS_VALU"])),
    "valu_note": "SQ_INSTS_VALU per launch; peak issue = 1024 SIMDs x 1 wave64 VALU instruction per 2 cycles x 2.4 GHz",
    "sq": {k: v for k, v in sorted(m.items()) if k.startswith("SQ_")},
    "round": rnd,
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
