#!/usr/bin/env python3
"""Where does the iterating (lane) kernel's time go?  Runs the headline batch with the solver's
iteration limit set to 1, 2, 3, ... : the kernel then stops every queued problem after that many
iterations, so the differences are the marginal cost of one more iteration over the whole work list and
limit 1 is the fixed part (launch, work-list and theta loads, b = Dth theta, outputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import linearmpc_jl_amd as lmpc
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "pendulum"
name = "pendulum" if wl.startswith("pendulum") else wl
g = bench.make_problem(name)
dev = torch.device("cuda", 0)
theta = torch.from_numpy(bench.make_theta(name, 1_000_000, 1234, wl == "pendulum_hard")).to(dev)
x = torch.empty((1_000_000, int(g["nu"])), dtype=torch.float64, device=dev)
f = torch.empty(1_000_000, dtype=torch.int32, device=dev)
for lim in (1, 2, 3, 4, 5, 6, 8, 10000):
    s = lmpc.default_settings()
    s.iter_limit = lim
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"],
                                  nout=int(g["nu"]), settings=s)
    for _ in range(10):
        qp.solve_device(theta, x=x, exitflag=f)
    torch.cuda.synchronize()
    qp.profile(True)
    for _ in range(100):
        qp.solve_device(theta, x=x, exitflag=f)
    torch.cuda.synchronize()
    n, whole, scr, it = qp.profile_read()
    print(f"iter_limit {lim:6d}: screen {scr*1e3:7.2f} us  iterate {it*1e3:7.2f} us   unsolved {(f < 1).float().mean().item():.4f}")

# floor: an empty work list (theta = 0: every problem is finished by the screening pass) -- what one
# launch of the iterating kernel costs between two events when no workgroup has anything to do
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=int(g["nu"]))
tz = torch.zeros_like(theta)
for _ in range(10):
    qp.solve_device(tz, x=x, exitflag=f)
torch.cuda.synchronize()
qp.profile(True)
for _ in range(100):
    qp.solve_device(tz, x=x, exitflag=f)
torch.cuda.synchronize()
n, whole, scr, it = qp.profile_read()
print(f"empty work list  : screen {scr*1e3:7.2f} us  iterate {it*1e3:7.2f} us")
