#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the headline hot path, one batch in flight (per-kernel durations
# and counters of single launches).  usage: tools/prof_hot.sh <tag> [bench args...]
set -u
TAG=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
COMMON="--streams 1 --steps 60 --warmup 10 --no-cpu-baseline --no-single-launch --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $COMMON "$@" > $OUT/bench_trace.log 2>&1
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$N -- python3 bench.py --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-single-launch --no-configs "$@" > $OUT/bench_pmc_$N.log 2>&1
done
python3 tools/pmc_summary.py $OUT $OUT/summary pendulum > /dev/null 2>&1
cat $OUT/summary.md
