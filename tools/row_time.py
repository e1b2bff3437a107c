"""Time of one batched solve on the row kernel (for A/B builds: LMPC_HIP_LIB=...): python tools/row_time.py [name] [N]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import linearmpc_jl_amd as lmpc
from conftest import load_golden
name = sys.argv[1] if len(sys.argv) > 1 else "mass_spring_3in"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
g = load_golden(name)
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=int(g["nu"]) if "nu" in g else None)
qp.set_option("row_kernel", 1)
th = torch.from_numpy(bench.make_theta(name, N, 77)).cuda()
x, ef = qp.solve_device(th); torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); qp.solve_device(th); b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
print(f"{os.environ.get('LMPC_HIP_LIB', 'default')}: {name} {best:.3f} ms per {N}; checksum {float(x.sum()):.12e} flags {int((ef == 1).sum())}")
