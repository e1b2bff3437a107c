#!/bin/bash
export TMPDIR=/tmp
# usage: profile_wave.sh <workload> <batch>   (extra bench args in $EXTRA)
OUT=gpurun_out/prof_wave
mkdir -p $OUT
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  N=$(echo $PASS | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 --workload $1 --batch $2 $EXTRA > $OUT/log_$N.txt 2>&1
done
python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_wave/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'wave_kernel' in r['Kernel_Name'] or 'lane_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            vg=(r['VGPR_Count'],r['SGPR_Count'],r['LDS_Block_Size'],r['Workgroup_Size'],r['Grid_Size'])
print(vg)
for k,v in sorted(agg.items()): print(k, sum(v)/len(v))
PY
