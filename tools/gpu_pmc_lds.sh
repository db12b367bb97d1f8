#!/bin/bash
# LDS-side counters of the traversal kernels (run on the GPU box): instructions, bank conflicts, active cycles.
# One rocprofv3 --pmc pass per config, --kernel-trace only beside it; output under gpurun_out/pmc_lds.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_lds
rm -rf "$out"; mkdir -p "$out"
for cfg in ${CONFIGS:-primary_bistro ao_bistro ao_hairball}; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --output-format csv -d "$out/$cfg" -- python3 tools/prof_config.py $cfg 6 > "$out/$cfg.log" 2>&1
  tail -2 "$out/$cfg.log" | cut -c1-200
  python3 - "$out/$cfg" "$cfg" <<'PY'
import csv, glob, sys
tot = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace" not in row["Kernel_Name"]:
            continue
        import re
        c = tot.setdefault((re.search(r"k_trace<[^>]*>", row["Kernel_Name"]).group(0), row["Counter_Name"]), [0.0, 0])
        c[0] += float(row["Counter_Value"]); c[1] += 1
for (k, n), (s, c) in sorted(tot.items()):
    print("PMC_LDS %s %s %s per launch %.4g (%d launches)" % (sys.argv[2], k, n, s / c, c))
PY
done
