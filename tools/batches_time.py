"""Headline workload, several batches per call (lmpc_solve_batches_device) against one call per batch, on ONE stream,
cold HBM (rotating batches): python tools/batches_time.py [nb ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import linearmpc_jl_amd as lmpc
from conftest import load_golden
g = load_golden("pendulum")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
N, NROT = 1_000_000, 12
dev = torch.device("cuda:0")
ths = [torch.from_numpy(bench.make_theta("pendulum", N, 7919 * r + 1234)).to(dev) for r in range(NROT)]
xs = [torch.empty((N, 1), dtype=torch.float64, device=dev) for _ in range(NROT)]
fs = [torch.empty(N, dtype=torch.int32, device=dev) for _ in range(NROT)]
st = torch.cuda.current_stream(dev).cuda_stream
for nb in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6]:
    groups = [list(range(i, i + nb)) for i in range(0, NROT, nb) if i + nb <= NROT]
    calls = [qp.bind_device_batches([ths[i] for i in gset], [xs[i] for i in gset], [fs[i] for i in gset], st) if nb > 1
             else qp.bind_device_call(ths[gset[0]], xs[gset[0]], fs[gset[0]], st) for gset in groups]
    for c in calls: c()
    torch.cuda.synchronize()
    reps = 40
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(reps):
        calls[r % len(calls)]()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{nb} batch(es) per call: {ms*1e3:.2f} us per call, {ms*1e3/nb:.2f} us per batch, {nb*N/(ms*1e-3):.3e} solves/s, "
          f"{68*nb*N/(ms*1e-3)/1e9:.0f} GB/s = {68*nb*N/(ms*1e-3)/8e12:.3f} of 8 TB/s", flush=True)
