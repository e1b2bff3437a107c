#!/bin/bash
# rocprofv3 kernel trace + SQ / TCC / traffic counter passes of the AVI kernel on the game_avi configuration.
# usage: tools/prof_avi.sh <tag>
set -u
TAG=$1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/avi_run.py 10 > $OUT/bench_trace.log 2>&1 || exit 1
for PASS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PASS --output-format csv -d $OUT/pmc_$N -- python3 tools/avi_run.py 3 > $OUT/bench_pmc_$N.log 2>&1 || exit 1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
lines = [f"# rocprofv3 summary {tag}: tools/avi_run.py (game_avi, 1e6 points per call)", "", "## kernel-trace --stats", "",
         "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
for f in glob.glob(out + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) >= 0.05:
            lines.append(f"| `{r['Name'][:120]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.1f} |")
agg = collections.defaultdict(lambda: collections.defaultdict(list)); res = {}
for f in glob.glob(out + "/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lmpc::avi" in k:
            k = k.split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
lines += ["", "## PMC (average per dispatch)", ""]
tot_f = tot_w = 0.0
for k in sorted(agg):
    lines += [f"`{k}`: VGPR {res[k][0]} AGPR {res[k][1]} SGPR {res[k][2]} LDS {res[k][3]} B, workgroup {res[k][4]}, grid {res[k][5]}", "", "| counter | value |", "|---|---|"]
    lines += [f"| {c} | {sum(v) / len(v):.1f} |" for c, v in sorted(agg[k].items())]
    if "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
        fk, wk = sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"]), sum(agg[k]["WRITE_SIZE"]) / len(agg[k]["WRITE_SIZE"])
        tot_f += fk; tot_w += wk
        lines += ["", f"HBM traffic per dispatch (FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md): {(2 * fk + wk) * 1024 / 1e6:.1f} MB"]
    lines += [""]
lines += [f"HBM traffic of the chain per call: {(2 * tot_f + tot_w) * 1024 / 1e6:.1f} MB against 68 MB algorithmic"]
open(out + "/summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
