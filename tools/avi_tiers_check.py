#!/usr/bin/env python3
"""Register-resident chain of the variational kernel (avi_tiers -> avi) against the generic kernel alone and the CPU
checker, bit for bit (x, flags, iteration counts, active sets), then timings of the chain per first-pass depth."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402
from oracle import avi as oavi  # noqa: E402

dev = torch.device("cuda", 0)
g = bench.make_problem("game_kat")
nout = int(g["nu"])
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout, device=0)
print("kernel", qp.kernel_name)
pk = qp.avi_pack()
P = oavi.AVI(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["ML"], pk["MR"], pk["G"], pk["du"], pk["dl"],
             pk["Dth"], pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"])).contiguous()
rng = np.random.default_rng(5)
ok = True
for name, th in (("wide", np.hstack([rng.uniform(-30, 30, (200_000, 4)), rng.uniform(-1, 1, (200_000, 2))])),
                 ("fixture", g["theta"]),
                 ("huge", np.hstack([rng.uniform(-300, 300, (50_000, 4)), rng.uniform(-3, 3, (50_000, 2))])),
                 ("one", np.hstack([rng.uniform(-30, 30, (1, 4)), rng.uniform(-1, 1, (1, 2))])),
                 ("ragged", np.hstack([rng.uniform(-30, 30, (4097, 4)), rng.uniform(-1, 1, (4097, 2))]))):
    th = np.ascontiguousarray(th)
    xo, efo, ito, acto = oavi.solve_batch(P, th)
    t = torch.from_numpy(th).to(dev)
    for first in (0, 1, 2, 3):
        qp.set_option("avi_tiers", 1)
        qp.set_option("avi_tiers_first", first)
        for rep in range(2):          # (twice: the counters' hand-over between calls)
            it = torch.full((len(th),), -77, dtype=torch.int32, device=dev)
            act = torch.full((len(th), qp.words), -1, dtype=torch.int64, device=dev)
            x, ef = qp.solve_device(t, iters=it, active=act)
            torch.cuda.synchronize()
            same = (np.array_equal(x.cpu().numpy(), xo) and np.array_equal(ef.cpu().numpy(), efo) and
                    np.array_equal(it.cpu().numpy(), ito) and np.array_equal(act.cpu().numpy().view(np.uint64), acto.view(np.uint64)))
            ok &= bool(same)
            print(name, "first", first, "rep", rep, "identical to the checker:", same, flush=True)
    qp.set_option("avi_tiers", 0)
    it = torch.full((len(th),), -77, dtype=torch.int32, device=dev)
    x, ef = qp.solve_device(t, iters=it)
    torch.cuda.synchronize()
    print(name, "generic alone identical:", np.array_equal(x.cpu().numpy(), xo) and np.array_equal(it.cpu().numpy(), ito), flush=True)
print("ALL IDENTICAL" if ok else "MISMATCH")

rng = np.random.default_rng(1234)
th = np.ascontiguousarray(np.hstack([rng.uniform(-30, 30, (1_000_000, 4)), rng.uniform(-1, 1, (1_000_000, 2))]))
ts = [torch.from_numpy(np.roll(th, r, axis=0).copy()).to(dev) for r in range(6)]
xb = torch.empty((1_000_000, nout), dtype=torch.float64, device=dev)
fb = torch.empty(1_000_000, dtype=torch.int32, device=dev)
for mode in ((0, 2), (1, 0), (1, 1), (1, 2), (1, 3)):
    qp.set_option("avi_tiers", mode[0])
    qp.set_option("avi_tiers_first", mode[1])
    for k in range(3):
        qp.solve_device(ts[k % 6], x=xb, exitflag=fb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        qp.solve_device(ts[k % 6], x=xb, exitflag=fb)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 20
    qp.profile(True)
    for k in range(5):
        qp.solve_device(ts[k % 6], x=xb, exitflag=fb)
    torch.cuda.synchronize()
    pr = qp.profile_read()
    qp.profile(False)
    print("tiers", mode, "ms per 1e6: %.4f" % (el * 1e3), "profile", pr, flush=True)
