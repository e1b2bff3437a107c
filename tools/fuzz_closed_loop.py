#!/usr/bin/env python3
"""Randomized differential run of the wavefront path's closed loops: random controllers (general rows, soft rows, 1-3
inputs, state dimension 2-8, working-set capacities from 10 to 64 and beyond), random stable plants, scenario counts that
are no multiple of a wavefront -- lmpc_simulate against the CPU checker's closed loop bit for bit, in every execution
order: scenario-asynchronous rounds with run-ahead (default), step-synchronous with the kept factorisation, both with a
first pass forced at 24 rows, the mask start, cold.  usage: tools/fuzz_closed_loop.py [trials] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import linearmpc_jl_amd as lmpc
from conftest import oracle_ldp_from
from oracle import ldp as oldp

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad = 0
legs = 0
t0 = time.time()
for trial in range(trials):
    nx = int(rng.integers(2, 9)); nu = int(rng.integers(1, 4)); nr = int(rng.integers(0, 3)); nup = int(rng.integers(0, nu + 1))
    nth = nx + nr + nup
    n = int(rng.integers(max(nu, 3), 45))
    mg = int(rng.integers(10, 140))
    nsoft = int(rng.integers(0, mg + 1)) if rng.random() < 0.6 else 0
    Hh = rng.standard_normal((n, n)); H = Hh @ Hh.T + n * np.eye(n)
    A = rng.standard_normal((mg, n)); m = n + mg
    scale = rng.choice([0.3, 1.0, 3.0])
    bu = scale * rng.uniform(0.5, 2.0, m); bl = -scale * rng.uniform(0.5, 2.0, m)
    W = 0.3 * rng.standard_normal((m, nth)); W[:n] = 0.0
    if nsoft and rng.random() < 0.5:
        W[n:, 0] = np.abs(W[n:, 0]) * rng.uniform(0.5, 2.0)      # one state pushes many soft rows over their bounds
    f_theta = 0.6 * rng.standard_normal((n, nth))
    sense = np.zeros(m, np.int32)
    if nsoft:
        sense[n + rng.choice(mg, nsoft, replace=False)] = 8
    gram = int(rng.random() < 0.5)
    try:
        qp = lmpc.BatchedQP.from_mpqp(H, np.zeros(n), f_theta, A, bu, bl, W, sense, nout=nu)
    except lmpc.LmpcError:
        continue
    qp.set_option("wave", 1)
    if not qp.kernel_name.endswith("wave"):
        continue
    qp.set_option("gram_scan", gram)
    Fm = rng.standard_normal((nx, nx)); Fm *= rng.uniform(0.5, 0.97) / np.abs(np.linalg.eigvals(Fm)).max()
    Gm = 0.5 * rng.standard_normal((nx, nu))
    N, T = int(rng.integers(100, 700)), int(rng.integers(5, 30))
    x0 = rng.uniform(-2, 2, (N, nx)) * rng.choice([0.3, 1.0, 3.0])
    r = rng.uniform(-1, 1, (N, nr)) if nr else None
    L = oracle_ldp_from(qp.ldp())
    so = oldp.default_settings(); so.mode = gram
    refs = {}
    def ref(w):
        if w not in refs:
            refs[w] = oldp.simulate(L, x0, T, Fm, Gm, r=r, warm=w, settings=so)
        return refs[w]
    cases = [({"sim_async": 1}, True, 2), ({"sim_async": 0}, True, 2), ({"sim_async": 1, "wave_two_pass": 1}, True, 2),
             ({"sim_async": 0, "wave_two_pass": 1}, True, 2), ({"sim_async": 0, "sim_keep_factor": 0}, True, 1),
             ({"sim_async": 1}, False, False), ({"sim_async": 1, "wave_two_pass": 1}, False, False)]
    for opts, warm, owarm in cases:
        for k_, v_ in {"sim_async": 1, "wave_two_pass": -1, "sim_keep_factor": 1, "sim_fused": 1}.items():
            qp.set_option(k_, v_)
        for k_, v_ in opts.items():
            qp.set_option(k_, v_)
        out = qp.simulate(x0, T, Fm, Gm, r=r, warm=warm)
        rf = ref(owarm)
        ok = all(np.array_equal(out[key], rf[key]) for key in ("U", "X", "x", "flag_min"))
        legs += 1
        if not ok:
            bad += 1
            print(f"MISMATCH trial {trial}: nx={nx} nu={nu} nr={nr} nup={nup} n={n} mg={mg} nsoft={nsoft} gram={gram} N={N} T={T} "
                  f"opts={opts} warm={warm}", flush=True)
    if trial % 20 == 19:
        print(f"trial {trial + 1}: {bad} mismatches in {legs} closed loops, {time.time() - t0:.0f} s", flush=True)
    qp.close()
print(f"done: {trials} trials, {legs} closed loops, {bad} mismatches")
sys.exit(1 if bad else 0)
