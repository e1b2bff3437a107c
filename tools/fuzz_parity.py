#!/usr/bin/env python3
"""Long randomized differential run: random problem shapes / row kinds / bounds, HIP path vs CPU oracle on the
same pack -- exit flags, iteration counts, active sets bit for bit, x to 1e-10; cold and warm; f64 and (every
fourth trial, wavefront-kernel shapes) f32; n up to 100 (two variable slots per lane).  Also prints the MARGINAL
case report (oracle.ldp.marginal_report): the points whose terminal decision sits inside a tolerance band, where
libdaqp may legitimately end on another active set.  usage: tools/fuzz_parity.py [trials] [seed] [gram]
("gram": the wavefront kernel's Gram-scan form against the oracle's mode 1, plus its agreement with the n-chain form
on every solved point: flags, iteration counts, active sets, and the largest |dx|).  Wavefront-kernel trials also
re-solve in two passes at random first-pass capacities, alternating with one pass on the same handle: identical arrays."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import linearmpc_jl_amd as lmpc
from conftest import oracle_ldp_from
from oracle import ldp as oldp

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gram = len(sys.argv) > 3 and sys.argv[3] == "gram"
gram_vs_chain = {"solved_points": 0, "decisions_differ": 0, "max_dx_hard": 0.0, "max_dx_soft": 0.0, "failed_flag_differs": 0}
rng = np.random.default_rng(seed)
bad = 0
stats = {"lane": 0, "wave": 0, "refused": 0, "f32": 0}
marg = {"points": 0, "primal_marginal": 0, "dual_marginal": 0, "trials_with_marginal_points": 0}
flags_seen = {}
t0 = time.time()


def copy_settings(s):
    so = oldp.Settings()
    for f, _ in so._fields_:
        setattr(so, f, getattr(s, f, 0))
    return so


for trial in range(trials):
    n = int(rng.integers(1, 40)) if rng.random() < 0.3 else int(rng.integers(1, 15))
    if trial % 25 == 24:
        n = int(rng.integers(64, 101))             # long horizons: two variable slots per lane
    mg = int(rng.integers(0, 120)) if rng.random() < 0.3 else int(rng.integers(0, 46))
    if trial % 10 == 9:
        mg = int(rng.integers(120, 380))           # 3 .. 6 constraint slots per lane
    ms = n if rng.random() < 0.7 else 0
    nth = int(rng.integers(0, 20))
    if ms + mg == 0:
        mg = 1
    nsoft = int(rng.integers(0, mg + 1)) if (mg and rng.random() < 0.4) else 0
    Hh = rng.standard_normal((n, n)); H = Hh @ Hh.T + n * np.eye(n)
    A = rng.standard_normal((mg, n))
    m = ms + mg
    scale = rng.choice([0.3, 1.0, 3.0])
    bu = scale * rng.uniform(0.5, 2.0, m); bl = -scale * rng.uniform(0.5, 2.0, m)
    W = 0.3 * rng.standard_normal((m, nth)); W[:ms] = 0.0
    f_theta = rng.standard_normal((n, nth))
    sense = np.zeros(m, np.int32)
    if nsoft:
        sense[ms + rng.choice(mg, nsoft, replace=False)] = 8
    for j in range(m):
        r = rng.random()
        if sense[j] == 0 and r < 0.08: bu[j] = 1e30
        elif sense[j] == 0 and r < 0.16: bl[j] = -1e30
        elif sense[j] == 0 and r < 0.20: bu[j], bl[j], sense[j] = 1e30, -1e30, 4
        elif sense[j] == 0 and j >= ms and r < 0.24 and n >= 3:
            v = rng.uniform(-0.2, 0.2); bu[j], bl[j], sense[j] = v, v, 5
    if (sense == 5).sum() > max(n - 1, 0):
        sense[sense == 5] = 0
    bnb = trial % 5 == 4 and ms == n and n <= 10
    if bnb:                                        # hybrid: some simple bounds become BINARY rows
        sense[:ms][sense[:ms] != 0] = 0
        bu[:ms] = np.where(bu[:ms] > 1e20, 1.0, bu[:ms]); bl[:ms] = np.where(bl[:ms] < -1e20, -1.0, bl[:ms])
        nb = int(rng.integers(1, min(n, 6) + 1))
        sense[rng.choice(ms, nb, replace=False)] = 16
        stats["bnb"] = stats.get("bnb", 0) + 1
    f32 = trial % 4 == 3
    st = lmpc.default_settings_f32() if f32 else None
    try:
        qp = lmpc.BatchedQP.from_mpqp(H, np.zeros(n), f_theta, A, bu, bl, W, sense, nout=min(n, 3), settings=st)
    except lmpc.LmpcError as e:
        stats["refused"] += 1
        assert e.code in (-1, -6, -103), e
        continue
    theta = rng.uniform(-2, 2, (193, nth))
    L = oracle_ldp_from(qp.ldp())
    so = copy_settings(st) if f32 else oldp.default_settings()
    if gram and (f32 or qp.kernel_name.endswith("wave")):     # (the lane kernels have one form)
        qp.set_option("gram_scan", 1)
        so.mode = 1
    try:
        if f32:
            x, ef, it, act = qp.solve_f32(theta.astype(np.float32))
            xo, efo, ito, acto = oldp.solve_batch(L, theta.astype(np.float32), so, dtype=np.float32)
            tol = 1e-5
            stats["f32"] += 1
        else:
            x, ef, it, act = qp.solve(theta)
            xo, efo, ito, acto = oldp.solve_batch(L, theta, so)
            tol = 1e-10
            stats["wave" if qp.kernel_name.endswith("wave") else "lane"] += 1
    except lmpc.LmpcError as e:
        assert e.code == -103 and f32, e          # binary32 needs the wavefront kernel
        continue
    keep = ef != -7                                # working-set capacity: the oracle has no such limit
    ok = np.array_equal(ef[keep], efo[keep]) and np.array_equal(it[keep], ito[keep]) and np.array_equal(act[keep], acto[keep])
    ok = ok and (np.abs(x[keep] - xo[keep]).max() <= tol if keep.any() else True)
    if ok and not f32 and not bnb and (ef >= 1).sum() >= 8:
        sel = ef >= 1
        xw, efw, itw, actw = qp.solve(theta[sel][:64], warm=act[sel][:64])
        xq, efq, itq, actq = oldp.solve_batch(L, theta[sel][:64], so, warm=act[sel][:64])
        ok = np.array_equal(efw, efq) and np.array_equal(itw, itq) and np.array_equal(actw, actq) and np.abs(xw - xq).max() <= tol
    if ok and not bnb and (f32 or qp.kernel_name.endswith("wave")):
        # the wavefront kernel in two passes, forced at a random first-pass capacity, alternating with one pass on the same
        # handle: never visible in a result
        for tp, c1 in ((1, int(rng.choice([8, 12, 16, 24, 32, 48]))), (0, 0), (1, int(rng.choice([8, 16, 24, 40]))), (1, 24), (0, 0)):
            qp.set_option("wave_two_pass", tp)
            if tp:
                qp.set_option("wave_cap1", c1)
            x2, ef2, it2, act2 = qp.solve_f32(theta.astype(np.float32)) if f32 else qp.solve(theta)
            same = np.array_equal(ef2, ef) and np.array_equal(it2, it) and np.array_equal(act2, act) and np.array_equal(x2, x)
            if not same:
                ok = False
                print(f"two-pass differs in trial {trial}: two_pass={tp} cap1={c1}", flush=True)
        qp.set_option("wave_two_pass", -1)
        stats["two_pass_legs"] = stats.get("two_pass_legs", 0) + 5
    for k, c in zip(*np.unique(ef, return_counts=True)):
        flags_seen[int(k)] = flags_seen.get(int(k), 0) + int(c)
    if gram and not f32 and qp.kernel_name.endswith("wave"):
        qp.set_option("gram_scan", 0)
        x0, ef0, it0, act0 = qp.solve(theta)
        s0 = ef0 >= 1
        gram_vs_chain["solved_points"] += int(s0.sum())
        dd = (ef[s0] != ef0[s0]) | (it[s0] != it0[s0]) | (act[s0] != act0[s0]).any(axis=1)
        gram_vs_chain["decisions_differ"] += int(dd.sum())
        gram_vs_chain["failed_flag_differs"] += int((ef[~s0] != ef0[~s0]).sum())
        same = np.flatnonzero(s0)[~dd]
        if len(same):
            key = "max_dx_soft" if nsoft else "max_dx_hard"
            gram_vs_chain[key] = max(gram_vs_chain[key], float(np.abs(x[same] - x0[same]).max()))
        if dd.any():
            print(f"gram/chain decisions differ in trial {trial} (n={n} m={m} nsoft={nsoft}): {int(dd.sum())} points", flush=True)
    if not f32 and not bnb:
        rep = oldp.marginal_report(L, theta)
        marg["points"] += rep["solved"]
        marg["primal_marginal"] += rep["primal_marginal"]; marg["dual_marginal"] += rep["dual_marginal"]
        if rep["primal_marginal"] or rep["dual_marginal"]:
            marg["trials_with_marginal_points"] += 1
            print(f"marginal points in trial {trial} (n={n} m={m}): primal {rep['primal_marginal_first']} dual {rep['dual_marginal_first']}", flush=True)
    if not ok:
        bad += 1
        print(f"MISMATCH trial {trial}: n={n} ms={ms} mg={mg} nth={nth} nsoft={nsoft} f32={f32} kernel={qp.kernel_name}", flush=True)
    if trial % 100 == 99:
        print(f"trial {trial + 1}: {bad} mismatches, {stats}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {trials} trials, {bad} mismatches, {stats}, exit flags {flags_seen}")
print(f"marginal cases among the solved f64 points: {marg}")
if gram:
    print(f"Gram-scan form against the n-chain form on the same handle: {gram_vs_chain}")
sys.exit(1 if bad else 0)
