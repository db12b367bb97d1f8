#!/bin/bash
# A/B of two builds under the counters (development aid): for each library in LIBS (names under tuning_libs/) and each
# config in CONFIGS, one rocprofv3 --pmc pass per counter group over tools/prof_config.py; summary by tools/pmc_ab_sum.py.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_ab
mkdir -p "$out"
for lib in ${LIBS:-c645 fix1}; do
  export TRX_LIB=tuning_libs/$lib.so
  for cfg in ${CONFIGS:-ao_hairball ao_bistro}; do
    i=0
    for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
               "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$out/$lib/$cfg/pmc$i" -- python3 tools/prof_config.py $cfg 6 > "$out/$lib.$cfg.pmc$i.log" 2>&1 || exit 1
    done
  done
done
python3 tools/pmc_ab_sum.py "$out"
