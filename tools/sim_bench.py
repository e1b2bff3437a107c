#!/usr/bin/env python3
"""Closed-loop batch simulation throughput (lmpc_simulate_device): N scenarios x T steps of
[form theta -> solve -> plant step] on the device.  Prints scenario-steps per second."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import linearmpc_jl_amd as lmpc
from linearmpc_jl_amd._cabi import lib, check
import ctypes

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warm", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "pendulum.npz")))
sys.path.insert(0, ROOT)
from oracle import mpc2mpqp as omm
prob = omm.pendulum()
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
rng = np.random.default_rng(0)
N, T = a.n, a.steps
dev = torch.device("cuda", 0)
x0 = torch.from_numpy(rng.uniform([-5, -5, -0.3, -2], [5, 5, 0.3, 2], (N, 4))).to(dev)
r = torch.from_numpy(np.hstack([rng.uniform(-5, 5, (N, 1)), np.zeros((N, 1))])).to(dev)
F = np.ascontiguousarray(prob.F); G = np.ascontiguousarray(prob.G)
vp = ctypes.c_void_p
fm = torch.empty(N, dtype=torch.int32, device=dev)
for rep in range(a.reps):
    x = x0.clone(); up = torch.zeros((N, 1), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    check(lib().lmpc_simulate_device(qp._h, N, T, 4, 2, 1, vp(F.ctypes.data), vp(G.ctypes.data), vp(x.data_ptr()),
                                     vp(r.data_ptr()), vp(up.data_ptr()), None, None, vp(fm.data_ptr()), a.warm,
                                     vp(torch.cuda.current_stream(dev).cuda_stream)), qp._h)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N} T={T} warm={a.warm}: {N*T/(t2-t0):.3e} scenario-steps/s, {1e6*(t2-t0)/T:.1f} us/step "
          f"(host enqueue {1e6*(t1-t0)/T:.1f} us/step), min flag {int(fm.min())}")
