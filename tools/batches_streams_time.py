"""Headline workload: calls of nb batches (lmpc_solve_batches_device) issued round-robin on ns streams, cold HBM
(rotating batches); wall time between a device synchronize and the next: python tools/batches_streams_time.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import linearmpc_jl_amd as lmpc
from conftest import load_golden
g = load_golden("pendulum")
N, NROT = 1_000_000, 24
dev = torch.device("cuda:0")
ths = [torch.from_numpy(bench.make_theta("pendulum", N, 7919 * r + 1234)).to(dev) for r in range(NROT)]
xs = [torch.empty((N, 1), dtype=torch.float64, device=dev) for _ in range(NROT)]
fs = [torch.empty(N, dtype=torch.int32, device=dev) for _ in range(NROT)]
for ns, nb in [(1, 1), (3, 1), (1, 8), (2, 4), (3, 4), (3, 8), (2, 8), (4, 2), (6, 1)]:
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    # one handle per stream (a handle's calls are stream-ordered on its own counters)
    qps = [lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1) for _ in range(ns)]
    groups = [list(range(i, i + nb)) for i in range(0, NROT, nb) if i + nb <= NROT]
    calls = []
    for k, gset in enumerate(groups):
        q, st = qps[k % ns], streams[k % ns].cuda_stream
        calls.append(q.bind_device_batches([ths[i] for i in gset], [xs[i] for i in gset], [fs[i] for i in gset], st) if nb > 1
                     else q.bind_device_call(ths[gset[0]], xs[gset[0]], fs[gset[0]], st))
    for c in calls: c()
    torch.cuda.synchronize()
    reps = 4 * len(calls)
    t0 = time.perf_counter()
    for r in range(reps):
        calls[r % len(calls)]()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per = el / (reps * nb)
    print(f"{ns} stream(s) x {nb} batch(es) per call: {per*1e6:.2f} us per batch, {N/per:.3e} solves/s, {68*N/per/8e12:.3f} of 8 TB/s", flush=True)
