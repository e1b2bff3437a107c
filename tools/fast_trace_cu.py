#!/usr/bin/env python3
"""Where do the one-launch kernel's stream-end times differ: between the workgroups of ONE CU, between CUs, between
XCDs?  Needs a -DLMPC_FAST_TRACE build (LMPC_HIP_LIB); diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import linearmpc_jl_amd as lmpc
g = bench.make_problem("pendulum")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
for o in sys.argv[1:]:
    k, v = o.split("="); qp.set_option(k, int(v))
ths = [torch.from_numpy(bench.make_theta("pendulum", 1000000, 1234 + i)).cuda() for i in range(6)]
for i in range(12):
    qp.solve_device(ths[i % 6])
torch.cuda.synchronize()
os.environ["LMPC_FAST_TRACE_FILE"] = "/tmp/fast_trace.bin"
for rep in range(3):
    qp.solve_device(ths[rep])
    torch.cuda.synchronize()
    t = np.fromfile("/tmp/fast_trace.bin", dtype=np.int64).reshape(-1, 4, 8)
    t0 = t[:, :, 0][t[:, :, 0] > 0].min()
    hw = (t[:, 0, 7] >> 16) & 0xffffffff
    xcc = (t[:, 0, 7] >> 48) & 0xff
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    se_t = (t[:, :, 1].astype(float) - t0) / 100.0           # stream end per wave (0 for the solver role)
    wg_end = se_t.max(axis=1)
    wend = ((t[:, :, 6].astype(float) - t0) / 100.0).max(axis=1)
    ncu = len(np.unique(cuid))
    per_cu = {c: wg_end[cuid == c] for c in np.unique(cuid)}
    cu_mean = np.array([v.mean() for v in per_cu.values()]); cu_spread = np.array([v.max() - v.min() for v in per_cu.values()])
    per_x = [wg_end[xcc == x] for x in np.unique(xcc)]
    print(f"rep {rep}: {len(wg_end)} workgroups on {ncu} CUs ({np.bincount(np.bincount(cuid.astype(int))[np.bincount(cuid.astype(int)) > 0]).tolist()} CUs with 0,1,2,3.. WGs)")
    print(f"   stream end per WG: med {np.median(wg_end):.2f} max {wg_end.max():.2f} us; kernel end max {wend.max():.2f}")
    print(f"   per-CU mean of stream end: min {cu_mean.min():.2f} med {np.median(cu_mean):.2f} max {cu_mean.max():.2f}; spread INSIDE a CU: med {np.median(cu_spread):.2f} max {cu_spread.max():.2f}")
    print("   per-XCD mean / max of stream end: " + " ".join(f"{v.mean():.1f}/{v.max():.1f}" for v in per_x))
    wgs = np.array([len(v) for v in per_cu.values()])
    for k in np.unique(wgs):
        print(f"   CUs with {k} WGs: mean stream end {cu_mean[wgs == k].mean():.2f} us ({(wgs == k).sum()} CUs)")
