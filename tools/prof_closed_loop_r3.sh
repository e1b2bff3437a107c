#!/bin/bash
# rocprofv3 kernel statistics of the wavefront path's closed loop (pendulum N = 50, 2e5 scenarios x 100 steps, warm), both
# forms, the default loop (scenario-asynchronous rounds with run-ahead) and the step-synchronous one (--async 0):
# gpurun -- 'bash tools/prof_closed_loop_r3.sh'  -> gpurun_out/r03_closed_loop_N50_{chain,gram}[_sync]_kernel_stats.csv
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for form in chain gram; do for mode in rounds sync; do
  g=0; [ $form = gram ] && g=1
  a=1; sfx=""; [ $mode = sync ] && { a=0; sfx="_sync"; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cl_$form$sfx -- python3 $R/tools/sim_bench.py --problem pendulum_N50 --n 200000 --steps 100 --async $a --gram $g --reps 1 > $R/gpurun_out/prof_cl_$form$sfx.log 2>&1
  cp $(ls $R/gpurun_out/prof_cl_$form$sfx/*/*kernel_stats.csv | head -1) $R/gpurun_out/r03_closed_loop_N50_${form}${sfx}_kernel_stats.csv
  tail -1 $R/gpurun_out/prof_cl_$form$sfx.log | cut -c1-110
  head -5 $R/gpurun_out/r03_closed_loop_N50_${form}${sfx}_kernel_stats.csv | cut -c1-90,330-420
done; done
