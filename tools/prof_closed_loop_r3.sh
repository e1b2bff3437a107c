#!/bin/bash
# rocprofv3 kernel statistics of the wavefront path's closed loop (pendulum N = 50, 2e5 scenarios x 100 steps, warm), both
# forms: gpurun -- 'bash tools/prof_closed_loop_r3.sh'  -> gpurun_out/r03_closed_loop_N50_{chain,gram}_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for form in chain gram; do
  g=0; [ $form = gram ] && g=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cl_$form -- python3 $R/tools/sim_bench.py --problem pendulum_N50 --n 200000 --steps 100 --async 0 --gram $g --reps 1 > $R/gpurun_out/prof_cl_$form.log 2>&1
  cp $(ls $R/gpurun_out/prof_cl_$form/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03_closed_loop_N50_${form}_kernel_stats.csv
  tail -1 $R/gpurun_out/prof_cl_$form.log
  head -6 $R/gpurun_out/r03_closed_loop_N50_${form}_kernel_stats.csv
done
