"""Row kernel (four problems per wavefront, lmpc_row_kernel.hpp) against the wavefront kernel and the oracle, and its
time per 10^6 problems.  Usage: python tools/row_check.py [name] [N] [--no-oracle] [--first] [--f32]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import bench  # noqa: E402
import linearmpc_jl_amd as lmpc  # noqa: E402
from conftest import load_golden, oracle_ldp_from  # noqa: E402


def timed(qp, th_d, reps=3):
    qp.solve_device(th_d)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        qp.solve_device(th_d)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "mass_spring_3in"
    N = int(args[1]) if len(args) > 1 else 40000
    g = load_golden(name)
    f32 = "--f32" in sys.argv
    st = lmpc.default_settings_f32() if f32 else None
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"],
                                  nout=int(g["nu"]) if "--first" in sys.argv and "nu" in g else None,
                                  **({"settings": st} if f32 else {}))
    theta = bench.make_theta(name, N, 77)
    if f32:
        theta = theta.astype(np.float32)
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
        sel = np.arange(0, N, max(1, N // 2000))
        if f32:
            so = oldp.Settings()
            for fl, _ in so._fields_:
                setattr(so, fl, getattr(st, fl, 0))
            xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel], so, dtype=np.float32)
        else:
            xo, efo, ito, acto = oldp.solve_batch(oracle_ldp_from(qp.ldp()), theta[sel])
        x, ef, it, ac = out[1]
        ok = (np.array_equal(ef[sel], efo) and np.array_equal(it[sel], ito) and np.array_equal(ac[sel], acto)
              and np.abs(x[sel] - xo).max() == 0.0)
        print("row kernel vs oracle on", len(sel), "points:", "IDENTICAL" if ok else "DIFFERENT")
        bad += 0 if ok else 1
    if N >= 100000:
        for mode in (0, 1):
            qp.set_option("row_kernel", mode)
            print(f"row_kernel={mode}: {timed(qp, th_d):.3f} ms per {N} problems")
    print("OK" if not bad else "FAILED")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
