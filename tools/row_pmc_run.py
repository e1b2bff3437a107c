"""Three calls of the batched solve on a resident batch (for the counter passes of tools/profile_row.sh)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import linearmpc_jl_amd as lmpc
from conftest import load_golden
name, N, rk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f32 = len(sys.argv) > 4 and sys.argv[4] == "f32"
g = load_golden(name)
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=int(g["nu"]) if "nu" in g else None,
                              **({"settings": lmpc.default_settings_f32()} if f32 else {}))
qp.set_option("row_kernel", rk)
th = torch.from_numpy(bench.make_theta(name, N, 77).astype(np.float32 if f32 else np.float64)).cuda()
for _ in range(3):
    qp.solve_device(th)
torch.cuda.synchronize()
print("done")
