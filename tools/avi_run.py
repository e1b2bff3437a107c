#!/usr/bin/env python3
"""The game_avi configuration of bench.py by itself (for rocprofv3: tools/prof_avi.sh)."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
r = bench.avi_config(torch, lmpc, torch.device("cuda", 0), 0, 1_000_000, steps, 2, False)
print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "verified", "mean_iterations", "kernel", "roofline")}))
