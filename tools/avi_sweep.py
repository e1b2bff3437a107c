#!/usr/bin/env python3
"""AVI kernel: resident wavefronts per CU (size of the scratch footprint) against throughput, game_avi 1e6 points."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, linearmpc_jl_amd as lmpc
g = bench.make_problem("game_kat")
rng = np.random.default_rng(1234)
N = 1_000_000
th = torch.from_numpy(np.ascontiguousarray(np.hstack([rng.uniform(-30, 30, (N, 4)), rng.uniform(-1, 1, (N, 2))]))).to("cuda:0")
ref = None
for lds, w in ((0, 16), (0, 8), (0, 24)):
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=2)
    qp.set_option("avi_waves", w)
    x, ef = qp.solve_device(th); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): qp.solve_device(th, x=x, exitflag=ef)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    if ref is None: ref = x.clone()
    print(f"avi_lds {lds} avi_waves {w:2d}: {1e3 * dt:.3f} ms per 1e6, identical {bool(torch.equal(x, ref))}", flush=True)
    qp.close()
