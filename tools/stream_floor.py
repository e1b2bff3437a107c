"""One-launch kernel on batches with a given share of points that need iterations: the stream's own floor."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import linearmpc_jl_amd as lmpc
dev = torch.device("cuda:0")
g = bench.make_problem("pendulum")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
N = 1000000
base = [bench.make_theta("pendulum", N, 100 + i) for i in range(6)]
for scale in (0.01, 0.5, 0.75, 1.0):
    ths = [torch.from_numpy(np.ascontiguousarray(b * scale)).to(dev) for b in base]
    xs = [torch.empty((N, 1), dtype=torch.float64, device=dev) for _ in range(6)]
    fs = [torch.empty(N, dtype=torch.int32, device=dev) for _ in range(6)]
    its = torch.empty(N, dtype=torch.int32, device=dev)
    for k in range(12):
        qp.solve_device(ths[k % 6], x=xs[k % 6], exitflag=fs[k % 6])
    qp.solve_device(ths[0], x=xs[0], exitflag=fs[0], iters=its)
    torch.cuda.synchronize()
    hard = float((its > 1).float().mean())
    qp.profile(True)
    for k in range(60):
        qp.solve_device(ths[k % 6], x=xs[k % 6], exitflag=fs[k % 6])
    torch.cuda.synchronize()
    cnt, ms, _, _ = qp.profile_read()
    qp.profile(False)
    print("theta x %.2f: share that needs iterations %.4f, one call %.2f us (%d calls, cold)" % (scale, hard, 1e3 * ms, cnt), flush=True)
