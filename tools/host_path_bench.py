#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point (lmpc_solve_batch): what a Julia caller with
Theta in host memory sees.  Never the headline `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linearmpc_jl_amd as lmpc
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "pendulum.npz")))
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
rng = np.random.default_rng(0)
for N in (10_000, 100_000, 1_000_000, 4_000_000):
    th = np.ascontiguousarray(np.hstack([rng.uniform([-5, -5, -0.3, -2], [5, 5, 0.3, 2], (N, 4)),
                                         rng.uniform(-5, 5, (N, 1)), np.zeros((N, 1)), rng.uniform(-2, 2, (N, 1))]))
    qp.solve(th, want_iters=False, want_active=False)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        x, ef, _, _ = qp.solve(th, want_iters=False, want_active=False)
    dt = (time.perf_counter() - t0) / reps
    print(f"N={N}: {N/dt:.3e} solves/s, {1e3*dt:.3f} ms per call, {68*N/dt/1e9:.1f} GB/s over PCIe (68 B per solve)")
