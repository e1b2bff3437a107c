"""One call over N = k x 10^6 contiguous pendulum points (what a virtual batch of k batches would cost)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import linearmpc_jl_amd as lmpc
from conftest import load_golden
g = load_golden("pendulum")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
for k in (1, 2, 3, 4, 6):
    N = k * 1_000_000
    nrot = max(2, 12 // k)
    ths = [torch.from_numpy(bench.make_theta("pendulum", N, 7919 * r + 1234)).to(dev) for r in range(nrot)]
    x = torch.empty((N, 1), dtype=torch.float64, device=dev); f = torch.empty(N, dtype=torch.int32, device=dev)
    calls = [qp.bind_device_call(t, x, f, st) for t in ths]
    for c in calls: c()
    torch.cuda.synchronize()
    reps = 30
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for r in range(reps): calls[r % nrot]()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"N = {k}e6 in one call: {ms*1e3:.2f} us, {ms*1e3/k:.2f} us per 1e6, {68*N/(ms*1e-3)/8e12:.3f} of 8 TB/s", flush=True)
    del ths, x, f, calls
