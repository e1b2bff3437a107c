#!/usr/bin/env python3
"""Randomized differential run of branch and bound (rows flagged BINARY): random hybrid problems -- n = 6 .. 40 simple
bounds of which 3 .. 10 binary, 0 .. 30 general rows (so that the search's nodes REMOVE rows: the lazy snapshots of a
node's factor have to be saved), one-sided rows -- wavefront kernel against the CPU oracle's search on the same pack:
exit flags, summed iteration counts, active sets, x bit for bit; binary64 / binary32 and n-chain / Gram-scan form in
turn.  Every fourth trial at 4 400 points: one pass against two passes at a forced small first capacity (overflow to
the second pass) and the default split, identical arrays.  usage: tools/fuzz_bnb.py [trials] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import linearmpc_jl_amd as lmpc
from conftest import oracle_ldp_from
from oracle import ldp as oldp

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
stats = {"f64": 0, "f32": 0, "gram": 0, "two_pass": 0, "nodes_iters": 0, "solved": 0, "infeasible": 0}
t0 = time.time()
for trial in range(trials):
    n = int(rng.integers(6, 41))
    mg = int(rng.integers(0, 31))
    nth = int(rng.integers(1, 9))
    nb = int(rng.integers(3, min(n, 10) + 1))
    Hh = rng.standard_normal((n, n)); H = Hh @ Hh.T + n * np.eye(n)
    A = rng.standard_normal((mg, n)); m = n + mg
    scale = rng.choice([0.5, 1.0, 2.0])
    bu = scale * rng.uniform(0.5, 2.0, m); bl = -scale * rng.uniform(0.5, 2.0, m)
    bu[n:] *= 3.0; bl[n:] *= 3.0                      # general rows looser: most points stay feasible
    W = 0.3 * rng.standard_normal((m, nth)); W[:n] = 0.0
    f_theta = rng.standard_normal((n, nth)) * n ** 0.5
    sense = np.zeros(m, np.int32)
    sense[rng.choice(n, nb, replace=False)] = 16
    for j in range(n, m):
        if rng.random() < 0.15: bl[j] = -1e30
    f32 = trial % 2 == 1
    gram = trial % 4 >= 2
    big = trial % 4 == 3
    st = lmpc.default_settings_f32() if f32 else None
    N = 4400 if big else 193
    theta = rng.uniform(-1.5, 1.5, (N, nth))
    thq = theta.astype(np.float32) if f32 else theta
    def make(opts):
        qp = lmpc.BatchedQP.from_mpqp(H, np.zeros(n), f_theta, A, bu, bl, W, sense, nout=min(n, 3), settings=st)
        if gram: qp.set_option("gram_scan", 1)
        for k, v in opts.items(): qp.set_option(k, v)
        return qp
    qp = make({"wave_two_pass": 0})
    solve = (lambda q: q.solve_f32(thq)) if f32 else (lambda q: q.solve(thq))
    x, ef, it, act = solve(qp)
    L = oracle_ldp_from(qp.ldp())
    so = oldp.default_settings_f32() if f32 else oldp.default_settings()
    so.mode = 1 if gram else 0
    nchk = min(N, 193)
    xo, efo, ito, acto = oldp.solve_batch(L, thq[:nchk], so, dtype=np.float32 if f32 else np.float64)
    keep = ef[:nchk] != -7
    ok = np.array_equal(ef[:nchk][keep], efo[keep]) and np.array_equal(it[:nchk][keep], ito[keep]) and \
        np.array_equal(act[:nchk][keep], acto[keep]) and np.array_equal(x[:nchk][keep], xo[keep])
    stats["f32" if f32 else "f64"] += 1; stats["gram"] += int(gram)
    stats["nodes_iters"] += int(it.sum()); stats["solved"] += int((ef >= 1).sum()); stats["infeasible"] += int((ef == -1).sum())
    if ok and big:
        cap = n + 1
        for opts in ({}, {"wave_two_pass": 1, "wave_cap1": max(8, min(cap - 4, nb + 2))}):
            q2 = make(opts)
            r2 = solve(q2)
            ok = ok and all(np.array_equal(a, b) for a, b in zip((x, ef, it, act), r2))
            q2.close()
            stats["two_pass"] += 1
    qp.close()
    if not ok:
        bad += 1
        print(f"MISMATCH trial {trial}: n={n} mg={mg} nb={nb} f32={f32} gram={gram} big={big} seed={seed}", flush=True)
    if (trial + 1) % 20 == 0:
        print(f"trial {trial + 1}: {bad} mismatches, {stats}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {trials} trials, {bad} mismatches, {stats}")
sys.exit(1 if bad else 0)
