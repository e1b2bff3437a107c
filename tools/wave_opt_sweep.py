"""Same-box sweep of launch options of the wavefront kernel on one workload.
    python tools/wave_opt_sweep.py workload[:batch][:f32] "k=v,k=v" "k=v" ...
"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import linearmpc_jl_amd as lmpc

dev = torch.device("cuda:0")
parts = sys.argv[1].split(":")
name = parts[0]
n = int(parts[1]) if len(parts) > 1 else 200000
f32 = len(parts) > 2 and parts[2] == "f32"
for spec in sys.argv[2:]:
    opts = {}
    for kv in spec.split(","):
        if kv:
            k, _, v = kv.partition("=")
            opts[k] = int(v)
    W = bench.Workload(torch, lmpc, name, n, dev, 0, 0, 1, f32=f32, rotate=False, options=opts)
    sec = W.timed(3, 1, nstreams=1) / 3
    print(f"{name:16s} {'f32' if f32 else 'f64'} N={n:8d} {spec:50s} {sec*1e3:9.3f} ms {n/sec:.3e}/s", flush=True)
    W.close()
