#!/bin/bash
# Profiles the bench command on the GPU box: kernel trace + stats, then PMC passes (separately).
# usage: tools_profile.sh <tag> [bench args...]
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-single-launch "$@" > $OUT/bench_trace.log 2>&1
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$N -- python3 bench.py --steps ${PMC_STEPS:-3} --warmup ${PMC_WARMUP:-1} --no-cpu-baseline --no-single-launch "$@" > $OUT/bench_pmc_$N.log 2>&1
done
find $OUT -name "*.csv" | head -50
