#!/bin/bash
# Decode-once experiment (development build in tuning_libs/dev.so): SQ counters of the primary kernel with the wave-uniform
# decode path off (TRX_TUNE=0x80000) and on (0x40000), bistro- and kitchen-class frames.  Counters in their own pass.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_decode_once
rm -rf "$out"; mkdir -p "$out"
export TRX_LIB=tuning_libs/dev.so
for cfg in primary_bistro primary_kitchen; do
  for tune in 0x80000 0x40000; do
    export TRX_TUNE=$tune
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$out/${cfg}_$tune" -- python3 tools/prof_config.py $cfg 8 > "$out/${cfg}_$tune.log" 2>&1
    python3 - "$out/${cfg}_$tune" "$cfg" "$tune" <<'PY'
import csv, glob, sys
from collections import defaultdict
d, cfg, tune = sys.argv[1:4]
tot = defaultdict(lambda: [0.0, 0]); durs = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace<0" not in r["Kernel_Name"] or r["Kernel_Name"].rstrip(">").endswith("true"):
            continue
        c = tot[r["Counter_Name"]]; c[0] += float(r["Counter_Value"]); c[1] += 1
        durs[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
v = sorted(durs.values())
print("%s TRX_TUNE=%s: kernel ms (faster half, under counters) %.4f | per launch: %s" % (
    cfg, tune, sum(v[: max(1, len(v) // 2)]) / max(1, len(v) // 2), ", ".join("%s %.1fM" % (k, s / n / 1e6) for k, (s, n) in sorted(tot.items()))))
PY
  done
done
