#!/usr/bin/env python3
"""Device time of the secondary entry points at 10^6 problems (pendulum): the generated controller's
three functions, form_parameter, the closed loop with a reference trajectory.  Spots kernels that are far
from what their bytes should cost."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import linearmpc_jl_amd as lmpc
import bench
from oracle import mpc2mpqp as omm, observer as oobs

g = bench.make_problem("pendulum")
p = omm.pendulum()
dev = torch.device("cuda", 0)
N = 1_000_000
mpc = lmpc.MPC(lmpc.MPQP(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"]), nx=4, nu=1, nr=2, nuprev=1)
ctl = lmpc.GeneratedController(mpc)
kf = oobs.kalman_filter(p.F, p.G, p.C, Q=1e2 * np.array([1e-3, 1, 1e-3, 1]), R=[1, 0.1])
ctl.set_observer(*kf.codegen_arrays(), 4, 1, 0, 2)
th = torch.from_numpy(bench.make_theta("pendulum", N, 1234)).to(dev)
x = th[:, :4].contiguous(); r = th[:, 4:6].contiguous(); u = th[:, 6:7].contiguous()
y = (x @ torch.from_numpy(p.C).to(dev).T).contiguous()
flags = torch.empty(N, dtype=torch.int32, device=dev)
qp = ctl.model


def timeit(name, fn, reps=20, nbytes=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    extra = f"  ({nbytes / us / 1e6:.2f} TB/s of {nbytes / 1e6:.0f} MB)" if nbytes else ""
    print(f"{name:44s} {us:9.1f} us{extra}")


uu = u.clone()
timeit("solve_device (screen + lane)", lambda: qp.solve_device(th, exitflag=flags), nbytes=68e6)
timeit("compute_control_device (update_parameter+solve)", lambda: qp.compute_control_device(uu, x, r, exitflag=flags), nbytes=68e6 + 112e6)
xs = x.clone()
timeit("predict_state_device", lambda: qp.predict_state(xs, u), nbytes=72e6)
timeit("correct_state_device", lambda: qp.correct_state(xs, y), nbytes=80e6)
rt = torch.from_numpy(np.tile(np.array([[1.0], [0.0]]), (1, 30))).to(dev)
timeit("form_parameter_device (shared trajectory)", lambda: qp.form_parameter_device(x, r=rt[:, :1].contiguous(), uprev=u), nbytes=96e6)
F, G = np.ascontiguousarray(p.F), np.ascontiguousarray(p.G)
t0 = time.perf_counter()
out = qp.simulate_ref(x.cpu().numpy()[:200000], 50, F, G, np.tile(np.array([[1.0], [0.0]]), (1, 60)), preview=0)
print(f"simulate_ref 2e5 scenarios x 50 steps incl. host copies: {1e3 * (time.perf_counter() - t0):.1f} ms")
