#!/usr/bin/env python3
"""Where the FIRST call on a fresh handle spends its time: reserve, calls 1 .. 4 (each synchronised), with and without
the wavefront path's probe.  usage: tools/first_call.py [workload ...]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
for wl in (sys.argv[1:] or ["pendulum", "pendulum_hard", "mass_spring", "pendulum_N50", "mass_spring_3in"]):
    for opts in ({}, {"wave_probe": 0}):
        t0 = time.perf_counter()
        w = bench.Workload(torch, lmpc, wl, 1_000_000, dev, 0, 0, 1, options=opts)
        torch.cuda.synchronize(dev)
        t_setup = time.perf_counter() - t0
        ts = []
        for k in range(5):
            t0 = time.perf_counter()
            w.launch(k)
            torch.cuda.synchronize(dev)
            ts.append(1e3 * (time.perf_counter() - t0))
        print(f"{wl:18s} {str(opts):18s} kernel {w.kernel:24s} setup+buffers {t_setup:6.2f} s; calls (ms): " + " ".join(f"{t:8.3f}" for t in ts), flush=True)
        del w
