#!/bin/bash
# The profile set of a round (run on the GPU box: ROUND=5 tools/profile_set.sh; output under gpurun_out/prof_rNN, summarised by
# tools/profile_summary.py --round N).  One script for every round: rounds 1-4 each carried a copy with the tag changed.
# Every rocprofv3 pass runs under `timeout`; the program itself follows `--` (no wrappers); counters in their own
# passes with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r$(printf %02d "${ROUND:-6}")
mkdir -p "$out"
if [ -z "$SKIP_BENCH" ]; then   # (a call is limited to 20 minutes: the seven configs go in three calls, the bench line in the first)
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_bench" -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-pmc --no-legs > "$out/stats_bench.log" 2>&1
fi
for cfg in ${CONFIGS:-primary_bistro primary_bistro_dense primary_hairball ao_bistro ao_hairball ao4_hairball tlas_san_miguel_4k rays_bistro}; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$cfg/stats" -- python3 tools/prof_config.py $cfg 10 > "$out/$cfg.stats.log" 2>&1
  grep PROF_CONFIG "$out/$cfg.stats.log"
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
             "FETCH_SIZE" "WRITE_SIZE" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
             "VALUBusy VALUUtilization SALUBusy"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/$cfg/pmc$i" -- python3 tools/prof_config.py $cfg 6 > "$out/$cfg.pmc$i.log" 2>&1
    tail -1 "$out/$cfg.pmc$i.log" | cut -c1-160
  done
done
python3 tools/profile_summary.py --round "${ROUND:-6}" "$out" --dry
