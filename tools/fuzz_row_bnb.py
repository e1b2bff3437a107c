"""Randomized parity of the row kernel's branch and bound (binary32; --f64 / --mixed: binary64) against the wavefront kernel: random hybrid problems
(binaries among the simple bounds, general rows, soft rows; infeasible assignments and infeasible problems) in both
instantiations -- up to 16 rows (n <= 15) and up to 48 rows (n <= 47, taken on request as the only pass) -- every output
compared bit for bit; a sample of each trial against the binary32 oracle.
Usage: python tools/fuzz_row_bnb.py [trials] [seed] [--f64 | --mixed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import linearmpc_jl_amd as lmpc  # noqa: E402
from conftest import oracle_ldp_from  # noqa: E402


def random_hybrid(rng, big):
    n = int(rng.integers(17, 44)) if big else int(rng.integers(3, 14))
    mg = int(rng.integers(0, min(60 - n, 20))) if big else int(rng.integers(0, min(30 - n, 10)))
    nth = int(rng.integers(1, 7))
    nsoft = int(rng.integers(0, min(mg, 3) + 1))
    nbin = int(rng.integers(1, 9 if big else 6))
    Hh = rng.standard_normal((n, n))
    H = Hh @ Hh.T + n * np.eye(n)
    A = rng.standard_normal((mg, n))
    m = n + mg
    bu = rng.uniform(0.3, 2.0, m)
    bl = -rng.uniform(0.3, 2.0, m)
    W = 0.3 * rng.standard_normal((m, nth))
    W[:n] = 0.0
    sense = np.zeros(m, np.int32)
    if nsoft:
        sense[n + rng.choice(mg, nsoft, replace=False)] = 8
    sense[rng.choice(n, min(nbin, n), replace=False)] |= 16
    return H, np.zeros(n), rng.standard_normal((n, nth)), A, bu, bl, W, sense


def run_trial(rng, big, N, s, so_cls, f64=False):
    H, f, fth, A, bu, bl, W, sense = random_hybrid(rng, big)
    qp = lmpc.BatchedQP.from_mpqp(H, f, fth, A, bu, bl, W, sense, **({} if f64 else {"settings": s}))
    theta = (rng.uniform(-1, 1, (N, W.shape[1])) * rng.choice([0.3, 1.0, 2.5])).astype(np.float64 if f64 else np.float32)
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
    from oracle import ldp as oldp
    sel = np.arange(0, N, max(1, N // 40))
    if f64:
        xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel])
    else:
        so = so_cls()
        for fl, _ in so._fields_:
            setattr(so, fl, getattr(s, fl, 0))
        xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel], so, dtype=np.float32)
    x, ef, it, ac = out[1]
    ok = (np.array_equal(ef[sel], efo) and np.array_equal(it[sel], ito) and np.array_equal(ac[sel].view(np.uint64), acto)
          and np.array_equal(x[sel], xo, equal_nan=True))
    return same, ok, H.shape[0], len(bu), int((sense & 16).astype(bool).sum()), float((ef >= 1).mean()), float(it.mean())


def main(trials=40, seed=1):
    from oracle import ldp as oldp
    rng = np.random.default_rng(seed)
    s = lmpc.default_settings_f32()
    bad = 0
    for t in range(trials):
        big = t % 2 == 1
        f64 = "--f64" in sys.argv or ("--mixed" in sys.argv and t % 4 >= 2)
        same, ok, n, m, nb, solved, its = run_trial(rng, big, 1500, s, oldp.Settings, f64)
        if not (same and ok):
            bad += 1
        print(f"trial {t:3d} {'f64' if f64 else 'f32'} n={n:2d} m={m:2d} binaries={nb} solved {solved:.2f} iterations {its:7.1f}: "
              f"{'identical' if same else 'DIFFERENT from the wavefront kernel'}, {'oracle ok' if ok else 'ORACLE MISMATCH'}", flush=True)
    print("OK" if not bad else f"FAILED ({bad} trials)")
    return 1 if bad else 0


if __name__ == "__main__":
    _a = [a for a in sys.argv[1:] if not a.startswith("--")]
    sys.exit(main(int(_a[0]) if len(_a) > 0 else 40, int(_a[1]) if len(_a) > 1 else 1))
