cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in pendulum_N50:200000 hybrid:100000; do
  name=${w%%:*}; n=${w##*:}
  extra=""; [ "$name" = "hybrid" ] && extra="--f32"
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name --output-format csv -- python3 bench.py --workload $name --batch $n --steps 4 --warmup 2 --no-cpu-baseline --no-configs --streams 1 $extra > gpurun_out/prof_$name.log 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/prof_$name/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "lmpc" in r["Name"]: print("$name", r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
