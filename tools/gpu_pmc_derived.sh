#!/bin/bash
# rocprofv3's derived busy metrics of the traversal kernels (run on the GPU box): VALUBusy (VALU pipelines busy, % of the
# kernel's duration), VALUUtilization (active lanes per VALU instruction, %), SALUBusy, MemUnitBusy.  One --pmc pass per
# config with --kernel-trace only beside it; output under gpurun_out/pmc_derived.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_derived
rm -rf "$out"; mkdir -p "$out"
for cfg in ${CONFIGS:-primary_bistro primary_bistro_dense primary_hairball ao_bistro ao_hairball tlas_san_miguel_4k rays_bistro}; do
  timeout 300 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy MemUnitBusy --output-format csv -d "$out/$cfg" -- python3 tools/prof_config.py $cfg 6 > "$out/$cfg.log" 2>&1
  python3 - "$out/$cfg" "$cfg" <<'PY'
import csv, glob, re, sys
want = {"primary": "k_trace<0", "ao": "k_trace<1", "tlas": "k_trace<0, true", "rays": "k_trace<2"}[sys.argv[2].split("_")[0]]
tot = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_trace<[^>]*>", row["Kernel_Name"])
        if not m or not m.group(0).startswith(want) or m.group(0).endswith("true>"):   # not the counting variant
            continue
        c = tot.setdefault((m.group(0), row["Counter_Name"]), [0.0, 0])
        c[0] += float(row["Counter_Value"]); c[1] += 1
print("PMC_DERIVED %-22s %s" % (sys.argv[2], "  ".join("%s %.1f" % (n, s / c) for (k, n), (s, c) in sorted(tot.items()))),
      "(%s, %d launches)" % (sorted(tot)[0][0] if tot else "no kernel rows", max([c for _, c in tot.values()] or [0])))
PY
done
