#!/bin/bash
# kernel traces of the game_avi chain for several option sets: tools/avi_trace.sh "<opts1>" "<opts2>" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
i=0
for O in "$@"; do
  i=$((i+1))
  export LMPC_BENCH_AVI_OPTIONS="$O"
  rm -rf gpurun_out/avi_trace_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/avi_trace_$i -- python3 tools/avi_run.py 10 > gpurun_out/avi_trace_$i.log 2>&1 || exit 1
  echo "== $O"
  python3 - gpurun_out/avi_trace_$i <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "lmpc" in r["Name"]:
            print("  %-60s calls %s avg %.1f us (min %.1f max %.1f)" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
