#!/bin/bash
# round-3 profile set: headline (one batch in flight) + the wavefront-kernel workloads at their bench sizes, both forms
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tools/profile_bench.sh r03_s1 --streams 1 --no-configs > gpurun_out/prof_r03_s1.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/prof_r03_s1 gpurun_out/r03_s1 pendulum > gpurun_out/r03_s1.traffic.json || exit 1
tools/prof_wave_r3.sh r03_wave_3in_1e6_chain mass_spring_3in 1000000 > gpurun_out/pw1.log 2>&1 || exit 1
tools/prof_wave_r3.sh r03_wave_3in_1e6_gram mass_spring_3in 1000000 --opt gram_scan=1 > gpurun_out/pw2.log 2>&1 || exit 1
tools/prof_wave_r3.sh r03_wave_hybrid_f32_1e5_chain hybrid 100000 --f32 > gpurun_out/pw3.log 2>&1 || exit 1
tools/prof_wave_r3.sh r03_wave_hybrid_f32_1e5_gram hybrid 100000 --f32 --opt gram_scan=1 > gpurun_out/pw4.log 2>&1 || exit 1
tools/prof_wave_r3.sh r03_wave_N100_chain pendulum_N100 200000 > gpurun_out/pw5.log 2>&1 || exit 1
tools/prof_wave_r3.sh r03_wave_N100_gram pendulum_N100 200000 --opt gram_scan=1 > gpurun_out/pw6.log 2>&1 || exit 1
tools/prof_wave_r3.sh r03_wave_N125_gram pendulum_N125 200000 --opt gram_scan=1 > gpurun_out/pw7.log 2>&1 || exit 1
echo done
