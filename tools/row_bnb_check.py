"""Branch and bound on the row kernel (four searches per wavefront, binary32) against the wavefront kernel and the
binary32 oracle, and its time.  Usage: python tools/row_bnb_check.py [name] [N] [--no-oracle] [--f64]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import bench  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402
from conftest import load_golden, oracle_ldp_from  # noqa: E402
from row_check import timed  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "satellite4"
    N = int(args[1]) if len(args) > 1 else 4000
    g = load_golden(name)
    f64 = "--f64" in sys.argv
    s = None if f64 else lmpc.default_settings_f32()
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"],
                                  **({} if f64 else {"settings": s}))
    rng = np.random.default_rng(5)
    base = g["theta"]
    theta = base[rng.integers(0, len(base), N)] * rng.uniform(0.2, 1.6, (N, 1)) + rng.normal(0, 0.01, (N, base.shape[1]))
    theta[: len(base)] = base[:N]
    if name == "satellite20" and N >= 50000:
        theta = bench.make_theta(name, N, 77)
    theta = theta.astype(np.float64 if f64 else np.float32)
    th_d = torch.from_numpy(theta).cuda()
    out = {}
    for mode in (0, 1):
        qp.set_option("row_kernel", mode)
        it_d = torch.empty(N, dtype=torch.int32, device="cuda")
        ac_d = torch.zeros((N, qp.words), dtype=torch.int64, device="cuda")
        x_d, ef_d = qp.solve_device(th_d, iters=it_d, active=ac_d)
        torch.cuda.synchronize()
        qp.check()
        out[mode] = (x_d.cpu().numpy(), ef_d.cpu().numpy(), it_d.cpu().numpy(), ac_d.cpu().numpy().view(np.uint64))
    names = ("x", "exitflag", "iters", "active")
    bad = 0
    for q in range(4):
        a, b = out[0][q], out[1][q]
        same = np.array_equal(a, b, equal_nan=True) if q == 0 else np.array_equal(a, b)
        if not same:
            rows = np.nonzero((a != b).reshape(N, -1).any(axis=1))[0]
            print(f"MISMATCH {names[q]}: {len(rows)} of {N} problems, first {rows[:8]}")
            for r in rows[:4]:
                print("   wave:", out[0][1][r], out[0][2][r], out[0][0][r][:3], " row:", out[1][1][r], out[1][2][r], out[1][0][r][:3])
            bad += 1
    print("row kernel vs wavefront kernel:", "IDENTICAL" if not bad else "DIFFERENT",
          "| flags", dict(zip(*np.unique(out[1][1], return_counts=True))), "| mean iterations", out[1][2].mean())
    if "--no-oracle" not in sys.argv:
        from oracle import ldp as oldp
        sel = np.arange(0, N, max(1, N // 300))
        if f64:
            xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel])
        else:
            so = oldp.Settings()
            for f, _ in so._fields_:
                setattr(so, f, getattr(s, f, 0))
            xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel], so, dtype=np.float32)
        x, ef, it, ac = out[1]
        ok = (np.array_equal(ef[sel], efo) and np.array_equal(it[sel], ito) and np.array_equal(ac[sel], acto)
              and np.abs(x[sel] - xo).max() == 0.0)
        print("row kernel vs oracle on", len(sel), "points:", "IDENTICAL" if ok else "DIFFERENT")
        bad += 0 if ok else 1
    if N >= 50000:
        for mode in (0, 1):
            qp.set_option("row_kernel", mode)
            print(f"row_kernel={mode}: {timed(qp, th_d):.3f} ms per {N} problems")
    print("OK" if not bad else "FAILED")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
