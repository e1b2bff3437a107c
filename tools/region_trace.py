#!/usr/bin/env python3
"""One-enqueue region discovery (lmpc_discover_regions_device) on the bench's sample, for rocprofv3 / option sweeps:
tools/region_trace.py [region_blocks ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402
from linearmpc_jl_amd import explicit  # noqa: E402

dev = torch.device("cuda", 0)
g = bench.make_problem("pendulum")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], device=0)
lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0]); ub = np.array([20.0] * 4 + [20.0, 0.0] + [2.0])
first = explicit.discover_regions_device(qp, lb, ub, 1_000_000, seed=4)
theta = first["theta"]
sampler = explicit.DeviceRegionSampler(qp, 1_000_000, capacity=1024)
for lf, rb in [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]] or [(1, 0)]:
    qp.set_option("region_lockfree", lf)
    qp.set_option("region_blocks", rb)
    sampler.run(theta); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(20):
        m, c, f, s = sampler.run(theta)
    dt = (time.perf_counter() - t0) / 20
    same = np.array_equal(m, first["masks"]) and np.array_equal(c, first["counts"]) and np.array_equal(f, first["first_index"])
    print(f"lockfree {lf} region_blocks {rb}: {1e3 * dt:.4f} ms per 1e6 samples, {len(m)} sets, identical {same}", flush=True)
