"""Randomized parity of the row kernel's plain solves (binary64 and binary32) against the wavefront kernel and the oracle:
random problems with simple bounds, general rows, soft rows, one-sided and immutable rows over all instantiations
(n <= 64, m <= 160).  Usage: python tools/fuzz_row.py [trials] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import linearmpc_jl_amd as lmpc  # noqa: E402
from conftest import oracle_ldp_from  # noqa: E402


def random_problem(rng):
    cls = int(rng.integers(0, 3))
    n = int(rng.integers(3, 17)) if cls == 0 else (int(rng.integers(17, 33)) if cls == 1 else int(rng.integers(33, 61)))
    mg = int(rng.integers(1, (48 if cls == 0 else 64 if cls == 1 else 96) - 0))
    nth = int(rng.integers(1, 14))
    nsoft = int(rng.integers(0, min(mg, 4) + 1))
    Hh = rng.standard_normal((n, n))
    H = Hh @ Hh.T + n * np.eye(n)
    A = rng.standard_normal((mg, n))
    m = n + mg
    bu = rng.uniform(0.2, 2.0, m)
    bl = -rng.uniform(0.2, 2.0, m)
    W = 0.4 * rng.standard_normal((m, nth))
    W[:n] = 0.0
    sense = np.zeros(m, np.int32)
    if nsoft:
        sense[n + rng.choice(mg, nsoft, replace=False)] = 8
    for j in rng.choice(m, int(rng.integers(0, 4)), replace=False):      # one-sided / free rows
        if sense[j] == 0:
            if rng.random() < 0.5:
                bl[j] = -1e30
            else:
                bu[j], bl[j], sense[j] = 1e30, -1e30, 4
    return H, np.zeros(n), rng.standard_normal((n, nth)), A, bu, bl, W, sense


def run_trial(rng, f32, N):
    from oracle import ldp as oldp
    H, f, fth, A, bu, bl, W, sense = random_problem(rng)
    s = lmpc.default_settings_f32() if f32 else None
    qp = lmpc.BatchedQP.from_mpqp(H, f, fth, A, bu, bl, W, sense, nout=min(3, H.shape[0]), **({"settings": s} if f32 else {}))
    theta = rng.uniform(-1, 1, (N, W.shape[1])) * rng.choice([0.5, 1.5, 4.0])
    theta = theta.astype(np.float32 if f32 else np.float64)
    th_d = torch.from_numpy(theta).cuda()
    out = {}
    for mode in (0, 1):
        qp.set_option("row_kernel", mode)
        it_d = torch.empty(N, dtype=torch.int32, device="cuda")
        ac_d = torch.zeros((N, qp.words), dtype=torch.int64, device="cuda")
        x_d, ef_d = qp.solve_device(th_d, iters=it_d, active=ac_d)
        torch.cuda.synchronize()
        qp.check()
        out[mode] = (x_d.cpu().numpy(), ef_d.cpu().numpy(), it_d.cpu().numpy(), ac_d.cpu().numpy())
    same = all(np.array_equal(out[0][q], out[1][q], equal_nan=(q == 0)) for q in range(4))
    sel = np.arange(0, N, max(1, N // 40))
    if f32:
        so = oldp.Settings()
        for fl, _ in so._fields_:
            setattr(so, fl, getattr(s, fl, 0))
        xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel], so, dtype=np.float32)
    else:
        xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel])
    x, ef, it, ac = out[1]
    # (a working set that wants more than n + 2 + #soft rows ends with exit flag -7 on the kernels and has
    # no room in the oracle's arrays either; such points -- none seen so far -- are counted, not compared)
    lim = (ef[sel] == -7) & (efo < 0)
    keep = ~lim
    ok = (np.array_equal(ef[sel][keep], efo[keep]) and np.array_equal(it[sel][keep], ito[keep])
          and np.array_equal(ac[sel].view(np.uint64)[keep], acto[keep]) and np.array_equal(x[sel][keep], xo[keep], equal_nan=True))
    return same, ok, H.shape[0], len(bu), qp.kernel_name, float((ef >= 1).mean()), float(it.mean()), int(lim.sum())


def main(trials=60, seed=1):
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(trials):
        f32 = t % 2 == 1
        same, ok, n, m, kn, solved, its, nlim = run_trial(rng, f32, 3000)
        if not (same and ok):
            bad += 1
        print(f"trial {t:3d} {'f32' if f32 else 'f64'} n={n:2d} m={m:3d} [{kn}] solved {solved:.2f} iterations {its:6.1f}: "
              f"{'identical' if same else 'DIFFERENT from the wavefront kernel'}, {'oracle ok' if ok else 'ORACLE MISMATCH'}"
              + (f" ({nlim} sampled points at the capacity limit)" if nlim else ""), flush=True)
    print("OK" if not bad else f"FAILED ({bad} trials)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
