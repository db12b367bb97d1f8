#!/bin/bash
# Regenerates everything under profiles/ that comes from the GPU (run on the GPU box, output in gpurun_out/prof_r01).
# Every rocprofv3 pass runs under `timeout`; the program itself follows `--` (no wrappers).
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r01
rm -rf "$out"; mkdir -p "$out"
python3 bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python3 bench.py --streams 1 --no-cpu-baseline > "$out/bench_streams1.json" 2> "$out/bench_streams1.err"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_default" -- python3 bench.py --steps 200 --no-cpu-baseline > "$out/stats_default.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_s1" -- python3 bench.py --steps 200 --streams 1 --no-cpu-baseline > "$out/stats_s1.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/pmc_$name" -- python3 tools/prof_target.py bistro 10 > "$out/pmc_$name.log" 2>&1
  tail -1 "$out/pmc_$name.log" | cut -c1-200
done
find "$out" -name "*.csv" | head -40
tail -c 600 "$out/bench_default.json"
