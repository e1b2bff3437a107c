#!/bin/bash
# PMC counters of the row kernel (four problems per wavefront): tools/profile_row.sh <name> <N> [row_kernel option, default 1] [f32]
# Two counter passes (kernel trace only beside them) over tools/row_pmc_run.py; summary on stdout.
export TMPDIR=/tmp
NAME=${1:-mass_spring_3in}; N=${2:-1000000}; RK=${3:-1}; F32=${4:-f64}
OUT=gpurun_out/prof_row_$RK
rm -rf $OUT; mkdir -p $OUT
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  T=$(echo $PASS | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$T -- python3 tools/row_pmc_run.py $NAME $N $RK $F32 > $OUT/log_$T.txt 2>&1
done
python3 - "$OUT" "$N" <<'PY'
import csv, glob, collections, sys
out, N = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(list); meta = {}
for f in glob.glob(out + '/pmc_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'row_kernel' in k or 'wave_kernel' in k:
            key = 'row' if 'row_kernel' in k else 'wave'
            agg[(key, r['Counter_Name'])].append(float(r['Counter_Value']))
            meta[key] = (r['VGPR_Count'], r.get('Accum_VGPR_Count', '?'), r['SGPR_Count'], r['LDS_Block_Size'], r['Workgroup_Size'], r['Grid_Size'])
for key in sorted(meta):
    print(f"== {key}_kernel: VGPR {meta[key][0]} AGPR {meta[key][1]} SGPR {meta[key][2]} LDS {meta[key][3]} B, workgroup {meta[key][4]}, grid {meta[key][5]}")
    for (k2, c), v in sorted(agg.items()):
        if k2 == key:
            print(f"   {c:24s} {sum(v)/len(v):16.0f}   per problem {sum(v)/len(v)/N:10.1f}")
PY
