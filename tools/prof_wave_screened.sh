#!/bin/bash
# rocprofv3 kernel trace of a wavefront-kernel workload behind the screening pass (one batch in flight), plus one PMC
# pass for the L2/VALU picture of the wavefront kernel.  usage: tools/prof_wave_screened.sh <workload> <batch>
set -u
W=$1; B=$2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_wscr_$W
mkdir -p $OUT
ARGS="--workload $W --batch $B --streams 1 --steps 8 --warmup 2 --no-cpu-baseline --no-single-launch --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py $ARGS > $OUT/bench_pmc_tcc.log 2>&1
python3 - "$OUT" "$W" <<'PY'
import csv, glob, sys, collections, json
out, w = sys.argv[1], sys.argv[2]
lines = [f"# rocprofv3 summary: {w} behind the screening pass (one batch in flight)", "", "## kernel-trace --stats", "",
         "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        lines.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.1f} |")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wave_kernel" in k or "screen_kernel" in k:
            agg["wave_kernel" if "wave_kernel" in k else "screen_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines += ["", "## PMC (average per dispatch)", ""]
for k, d in agg.items():
    lines += [f"### `{k}`", "", "| counter | value |", "|---|---|"] + [f"| {c} | {sum(v) / len(v):.1f} |" for c, v in sorted(d.items())] + [""]
open(out + "/summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
