#!/usr/bin/env python3
"""Same-process timing of the headline hot path for one library build (LMPC_HIP_LIB selects it):
one call at a time (HIP events per kernel) and three batches in flight, each on cache-resident and on
rotating (cold HBM) batches, with the chip kept busy right before every measurement (clocks up).

usage: LMPC_HIP_LIB=path/to/lib.so python tools/hot_ab.py [--workload pendulum] [--reps 2] [--opt name=value ...]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="pendulum")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--f32", action="store_true")
    a = ap.parse_args()
    import torch
    import linearmpc_jl_amd as lmpc
    dev = torch.device("cuda", 0)
    opts = {k: int(v) for k, v in (o.split("=") for o in a.opt)}
    W3 = bench.Workload(torch, lmpc, a.workload, bench.BATCH, dev, 0, 0, 3, f32=a.f32, options={**opts, "lane_block": 64})
    W1 = bench.Workload(torch, lmpc, a.workload, bench.BATCH, dev, 0, 0, 1, f32=a.f32, options=opts)
    tag = os.path.basename(os.environ.get("LMPC_HIP_LIB", "in-tree"))
    for rep in range(a.reps):
        for resident in (True, False):
            W1.timed(300, 10, resident)                       # clocks up
            solo = W1.single_launch(200, resident)
            W3.timed(300, 10, resident)
            el = W3.timed(a.steps, 10, resident)
            print(f"{tag:24s} {'resident' if resident else 'cold    '} single: call {1e3*solo[1]:6.2f} us = screen "
                  f"{1e3*solo[2]:6.2f} + iterate {1e3*solo[3]:6.2f} | 3 in flight: {1e6*el/a.steps:6.2f} us/step "
                  f"= {bench.BATCH*a.steps/el:.4g} solves/s", flush=True)


if __name__ == "__main__":
    main()
