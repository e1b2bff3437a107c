#!/bin/bash
# rocprofv3 kernel trace + SQ / TCC counter passes of one wavefront-kernel workload at its bench size (one batch in
# flight).  usage: tools/prof_wave_r3.sh <tag> <workload> <batch> [extra bench args ...]
set -u
TAG=$1; W=$2; B=$3; shift 3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--workload $W --batch $B --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-single-launch --no-configs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/bench_pmc_sq2.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py $ARGS > $OUT/bench_pmc_tcc.log 2>&1 || exit 1
python3 - "$OUT" "$TAG" "$W $B $*" <<'PY'
import csv, glob, sys, collections
out, tag, what = sys.argv[1], sys.argv[2], sys.argv[3]
lines = [f"# rocprofv3 summary {tag}: bench.py --workload {what} (one batch in flight)", "", "## kernel-trace --stats", "",
         "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) >= 0.05:
            lines.append(f"| `{r['Name'][:120]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.1f} |")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
res = {}
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wave_kernel" in k or "screen_kernel" in k or "qp_tiers_kernel" in k:
            kk = "wave_kernel" if "wave_kernel" in k else ("qp_tiers_kernel" if "qp_tiers_kernel" in k else "screen_kernel")
            agg[kk][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[kk] = (k[:140], r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
lines += ["", "## PMC (average per dispatch)", ""]
for k, d in agg.items():
    lines += [f"### `{res[k][0]}`", "", f"VGPR {res[k][1]} AGPR {res[k][2]} SGPR {res[k][3]} LDS {res[k][4]} B, workgroup {res[k][5]}, grid {res[k][6]}", "",
              "| counter | value |", "|---|---|"] + [f"| {c} | {sum(v) / len(v):.1f} |" for c, v in sorted(d.items())] + [""]
open(out + "/summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
grep -h '"metric"' $OUT/bench_trace.log > $OUT/bench.json || true
