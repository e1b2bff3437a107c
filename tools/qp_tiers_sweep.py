#!/usr/bin/env python3
"""Which path is fastest for small problems with general rows: the lane kernels (screen + lane), the wavefront kernel
behind its screening pass, or the wavefront kernel behind the tiers pass?  Random problems, 1e6 points each."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402
from test_gpu_parity import _random_qp  # noqa: E402

dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)
N = 1_000_000
for n, mg, amp in ((4, 20, 3.0), (4, 56, 3.0), (6, 30, 3.0), (6, 58, 3.0), (8, 24, 3.0), (8, 56, 3.0), (10, 20, 3.0), (10, 40, 3.0),
                   (10, 54, 3.0), (12, 30, 3.0), (12, 52, 3.0), (8, 56, 1.0), (12, 30, 1.0)):
    H, f, f_theta, A, bu, bl, W, sense = _random_qp(rng, n, mg, 6)
    qp = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, bu, bl, W, sense, nout=1)
    default = qp.kernel_name
    th = torch.from_numpy(np.ascontiguousarray(rng.uniform(-amp, amp, (N, 6)))).to(dev)
    xb = torch.empty((N, 1), dtype=torch.float64, device=dev); fb = torch.empty(N, dtype=torch.int32, device=dev)
    itb = torch.empty(N, dtype=torch.int32, device=dev)
    res = {}
    for label, opts in (("lane", {"wave": 0, "qp_tiers": 0}), ("tiers+lane", {"wave": 0, "qp_tiers": 1}), ("wave", {"wave": 1, "qp_tiers": 0}), ("tiers+wave", {"wave": 1, "qp_tiers": 1})):
        try:
            for k, v in opts.items():
                qp.set_option(k, v)
        except lmpc.LmpcError:
            continue
        for _ in range(2):
            qp.solve_device(th, x=xb, exitflag=fb, iters=itb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            qp.solve_device(th, x=xb, exitflag=fb)
        torch.cuda.synchronize()
        res[label] = 1e3 * (time.perf_counter() - t0) / 4
    it = itb.cpu().numpy(); ef = fb.cpu().numpy()
    print(f"n {n:2d} m {n + mg:2d} amp {amp}: default {default:22s} mean iters {it.mean():5.2f} solved {np.mean(ef >= 1):.2f}  ms per 1e6: "
          + "  ".join(f"{k} {v:7.3f}" for k, v in res.items()), flush=True)
    qp.close()

# the reference's mass_spring example on all four paths
import bench  # noqa: E402
g = bench.make_problem("mass_spring")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
th = torch.from_numpy(bench.make_theta("mass_spring", N, 1234, False)).to(dev)
xb = torch.empty((N, 1), dtype=torch.float64, device=dev); fb = torch.empty(N, dtype=torch.int32, device=dev)
ref = None
for label, opts in (("lane", {"wave": 0, "qp_tiers": 0}), ("tiers+lane", {"wave": 0, "qp_tiers": 1}), ("wave", {"wave": 1, "qp_tiers": 0}), ("tiers+wave", {"wave": 1, "qp_tiers": 1})):
    for k, v in opts.items():
        qp.set_option(k, v)
    for _ in range(2):
        qp.solve_device(th, x=xb, exitflag=fb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        qp.solve_device(th, x=xb, exitflag=fb)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 4
    cur = (xb.cpu().numpy().copy(), fb.cpu().numpy().copy())
    same = ref is None or (np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1]))
    ref = ref or cur
    print(f"mass_spring {label}: {ms:.3f} ms per 1e6, identical to the first: {same}", flush=True)
