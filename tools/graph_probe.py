#!/usr/bin/env python3
"""Does the hot path survive HIP graph capture?  Captures K steps (screen + iterate kernels on one
stream) with torch.cuda.graph, replays, compares results and per-step time with eager launches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import linearmpc_jl_amd as lmpc
import bench

g = bench.make_problem("pendulum")
dev = torch.device("cuda", 0)
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
theta = torch.from_numpy(bench.make_theta("pendulum", 1_000_000, 1234)).to(dev)
x = torch.empty((1_000_000, 1), dtype=torch.float64, device=dev)
f = torch.empty(1_000_000, dtype=torch.int32, device=dev)
for _ in range(5):
    qp.solve_device(theta, x=x, exitflag=f)
torch.cuda.synchronize()
x_ref, f_ref = x.clone(), f.clone()
K = 20
s = torch.cuda.Stream()
gr = torch.cuda.CUDAGraph()
x.zero_(); f.zero_()
torch.cuda.synchronize()
with torch.cuda.graph(gr, stream=s):
    for _ in range(K):
        qp.solve_device(theta, x=x, exitflag=f)
torch.cuda.synchronize()
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
print("graph results identical:", bool(torch.equal(x, x_ref) and torch.equal(f, f_ref)))
t0 = time.perf_counter()
R = 50
for _ in range(R):
    gr.replay()
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / (R * K)
t0 = time.perf_counter()
for _ in range(R * K):
    qp.solve_device(theta, x=x, exitflag=f)
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / (R * K)
print(f"one stream: graph {tg*1e6:.2f} us/step, eager {te*1e6:.2f} us/step")
