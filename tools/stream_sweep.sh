#!/bin/bash
# batches in flight x streaming wavefronts per workgroup of the one-launch kernel, cold HBM (bench.py's timed region)
for st in 1 2 3 4 6; do
  for ns in 3 4; do
    echo -n "streams=$st fast_nstr=$ns  "
    python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
import linearmpc_jl_amd as lmpc
W = bench.Workload(torch, lmpc, "pendulum", bench.BATCH, torch.device("cuda", 0), 0, 0, $st, options={"fast_nstr": $ns, "lane_block": 64})
W.timed(300, 10)
el = min(W.timed(900, 10) for _ in range(2))
print("%.2f us/step = %.4g solves/s" % (1e6 * el / 900, bench.BATCH * 900 / el))
PY
  done
done
