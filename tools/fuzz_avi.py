#!/usr/bin/env python3
"""Soak of the register-resident variational chain (avi_tiers -> avi_lane -> avi) against the CPU checker: random
box-constrained problems n = 2 .. 8 (random skew part up to 10x the symmetric part, nearly dependent bounds scaling,
wide and narrow boxes, theta scales that drive 0 .. n bounds active), every answer compared bit for bit (x, exit flag,
iteration count, active set).  usage: tools/fuzz_avi.py [trials] [points] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402
from oracle import avi as oavi  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
points = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2026
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
bad = 0
stats = {"problems": 0, "points": 0, "removals": 0, "handed_down": 0, "max_iter": 0}
for trial in range(trials):
    n = int(rng.integers(2, 9)); nth = int(rng.integers(1, 10)); nout = int(rng.integers(1, n + 1))
    B = rng.normal(size=(n, n)) * rng.uniform(0.2, 3)
    K = rng.normal(size=(n, n)) * rng.uniform(0.0, 10.0)
    H = B @ B.T + rng.uniform(1e-3, 1.0) * np.eye(n) + (K - K.T)
    sc = 10.0 ** rng.uniform(-2, 2, n)
    H = H * np.outer(sc, sc)
    bu = rng.uniform(0.01, 3, n) / sc; bl = -rng.uniform(0.01, 3, n) / sc
    if rng.random() < 0.2:
        bl[rng.random(n) < 0.5] = -1e30
    try:
        qp = lmpc.BatchedQP.from_mpqp(H, rng.normal(size=n) * sc, rng.normal(size=(n, nth)) * sc[:, None], np.zeros((0, n)), bu, bl,
                                      rng.normal(size=(n, nth)) * 0.3 / sc[:, None], np.zeros(n, np.int32), nout=nout)
    except lmpc.LmpcError as e:
        print("trial", trial, "setup refused:", e)
        continue
    if not qp.kernel_name.startswith("avi_tiers"):
        print("trial", trial, "kernel", qp.kernel_name)
        qp.close(); continue
    pk = qp.avi_pack()
    P = oavi.AVI(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["ML"], pk["MR"], pk["G"], pk["du"], pk["dl"],
                 pk["Dth"], pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"])).contiguous()
    th = np.ascontiguousarray(rng.normal(size=(points, nth)) * 10.0 ** rng.uniform(-1, 1.5))
    xo, efo, ito, acto = oavi.solve_batch(P, th)
    t = torch.from_numpy(th).to(dev)
    qp.set_option("avi_tiers_first", int(rng.integers(-1, 4)))
    it = torch.full((points,), -77, dtype=torch.int32, device=dev)
    act = torch.full((points, qp.words), -1, dtype=torch.int64, device=dev)
    x, ef = qp.solve_device(t, iters=it, active=act)
    torch.cuda.synchronize()
    efg, itg, actg, xg = ef.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy().view(np.uint64), x.cpu().numpy()
    ok = efo >= 1
    same = (np.array_equal(efg, efo) and np.array_equal(itg, ito) and np.array_equal(actg[ok], acto.view(np.uint64)[ok])
            and np.array_equal(xg[ok], xo[ok]))
    if not same:
        bad += 1
        print("MISMATCH trial", trial, "n", n, "flags", int((efg != efo).sum()), "iters", int((itg != ito).sum()),
              "x", int((xg[ok] != xo[ok]).any(axis=1).sum()), flush=True)
    na = np.array([bin(int(a)).count("1") for a in acto.view(np.uint64)[:, 0]])
    stats["problems"] += 1; stats["points"] += points
    stats["removals"] += int(((ito - 1 - na) > 0).sum()); stats["handed_down"] += int((efo < 1).sum())
    stats["max_iter"] = max(stats["max_iter"], int(ito.max()))
    qp.close()
print("fuzz_avi:", stats, "mismatching problems:", bad)
sys.exit(1 if bad else 0)
