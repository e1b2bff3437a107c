#!/usr/bin/env python3
"""The tiers pass in front of the wavefront kernel (small n, many rows: lmpc_qp_tiers_kernel.hpp) against the wavefront
path alone and the CPU checker, bit for bit (x on solved points, flags, iteration counts, active sets), then timings."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402
from conftest import oracle_ldp_from  # noqa: E402
from oracle import ldp as oldp  # noqa: E402

dev = torch.device("cuda", 0)
g = bench.make_problem("mass_spring")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1, device=0)
print("kernel", qp.kernel_name)
L = oracle_ldp_from(qp.ldp())
ok_all = True
rng = np.random.default_rng(3)
for name, th in (("bench", bench.make_theta("mass_spring", 100_000, 1234, False)), ("golden", g["theta"]),
                 ("narrow", rng.uniform(-1.5, 1.5, (50_000, 12))), ("one", rng.uniform(-4, 4, (1, 12))),
                 ("ragged", rng.uniform(-4, 4, (4097, 12)))):
    th = np.ascontiguousarray(th)
    xo, efo, ito, acto = oldp.solve_batch(L, th)
    t = torch.from_numpy(th).to(dev)
    for tiers in (2, 0, 1):
        qp.set_option("qp_tiers", tiers)
        it = torch.full((len(th),), -77, dtype=torch.int32, device=dev)
        act = torch.full((len(th), qp.words), -1, dtype=torch.int64, device=dev)
        x, ef = qp.solve_device(t, iters=it, active=act)
        torch.cuda.synchronize()
        efg, itg, actg, xg = ef.cpu().numpy(), it.cpu().numpy(), act.cpu().numpy().view(np.uint64), x.cpu().numpy()
        okp = efo >= 1
        same = (np.array_equal(efg, efo) and np.array_equal(itg, ito) and np.array_equal(actg, acto.view(np.uint64)) and
                np.array_equal(xg[okp], xo[okp]))
        fails_x = np.array_equal(xg[~okp], xo[~okp])
        ok_all &= bool(same)
        print(name, "tiers", tiers, "identical:", same, "(x of failed points identical too:", fails_x, ") flags", dict(zip(*np.unique(efo, return_counts=True))), flush=True)
        if not same:
            bad = np.flatnonzero((efg != efo) | (itg != ito) | (actg != acto.view(np.uint64)).any(axis=1))
            print("  first mismatches", bad[:8], efg[bad[:8]], efo[bad[:8]], itg[bad[:8]], ito[bad[:8]])
print("ALL IDENTICAL" if ok_all else "MISMATCH")

th = bench.make_theta("mass_spring", 1_000_000, 1234, False)
ts = [torch.from_numpy(np.roll(th, r, axis=0).copy()).to(dev) for r in range(4)]
xb = torch.empty((1_000_000, 1), dtype=torch.float64, device=dev)
fb = torch.empty(1_000_000, dtype=torch.int32, device=dev)
for tiers in (0, 2, 1):          # (0: never, 2: always, 1: the handle measures one call each way and goes with the faster)
    qp.set_option("qp_tiers", tiers)
    for k in range(3):
        qp.solve_device(ts[k % 4], x=xb, exitflag=fb)
    torch.cuda.synchronize()
    qp.profile(True)
    t0 = time.perf_counter()
    for k in range(10):
        qp.solve_device(ts[k % 4], x=xb, exitflag=fb)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / 10
    pr = qp.profile_read()
    qp.profile(False)
    print("qp_tiers", tiers, "ms per 1e6: %.3f" % (el * 1e3), "profile (calls, total, first kernel, second)", pr, flush=True)
