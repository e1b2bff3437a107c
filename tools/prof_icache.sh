#!/bin/bash
# instruction-cache counters of a wavefront-kernel workload: tools/prof_icache.sh <workload> <batch>
set -u
W=$1; B=$2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_icache_$W
mkdir -p $OUT
ARGS="--workload $W --batch $B --streams 1 --steps 4 --warmup 1 --no-cpu-baseline --no-single-launch --no-configs"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 bench.py $ARGS > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wave_kernel" in k: agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   ", c, sum(v) / len(v))
PY
tail -3 $OUT/log.txt
