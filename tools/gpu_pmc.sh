#!/bin/bash
# One rocprofv3 --pmc pass per counter group over the bench workload (development aid).
# Every pass runs under `timeout`: an invalid counter group makes rocprofv3 abort and then hang in finalisation.
# usage (on the GPU box): bash tools/gpu_pmc.sh <outdir> "<CTR CTR ...>" ["<CTR ...>" ...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=$1; shift
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/g$i" -- python3 tools/prof_target.py bistro 10 > "$out.g$i.log" 2>&1
  tail -1 "$out.g$i.log"
done
python3 tools/pmc_summary.py "$out"
