"""Generates tests/golden/game_kat.npz: the reference's game-theoretic MPC test problem
(/root/reference/test/runtests.jl:1337-1358 -- two players, one input each, non-symmetric H, handed to DAQP with
is_avi = true by /root/reference/src/setup.jl:11-13) as the condensing restatement builds it (oracle/mpc2mpqp.py,
following /root/reference/src/mpc2mpqp.jl:900-950), seeded parameter points, and the AVI oracle's answers for them.

Run from the repo root:  python tests/golden/make_game.py

Cross-checks before anything is written: (a) the closed loop of the reference's test (x0 = [10,10], r = [10,0],
N = 500) ends at y = [10, 0] within the test's atol 1e-4; (b) every 10th answer equals a projection iteration
x <- clip(x - tau (H x + f(theta)), bl, bu) run to a fixed point (an independent method: no active sets, no
factorisation) to 1e-9; (c) every answer's KKT certificate (oracle.avi.kkt_residual) is below 1e-8.
Julia/DAQP cannot run in this image, so the fixture is not an output of the reference itself.
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import avi as oavi  # noqa: E402
from oracle import mpc2mpqp as omm  # noqa: E402


def projection_fixed_point(H, fth, lo, hi, iters=400000):
    Hs = (H + H.T) / 2
    tau = np.linalg.eigvalsh(Hs)[0] / np.linalg.norm(H, 2) ** 2
    x = np.zeros(len(fth))
    for _ in range(iters):
        xn = np.clip(x - tau * (H @ x + fth), lo, hi)
        if np.abs(xn - x).max() < 1e-15:
            break
        x = xn
    return x


def main():
    p = omm.game_kat()
    q = omm.mpc2mpqp(p)
    assert not q.is_symmetric and np.linalg.eigvalsh((q.H + q.H.T) / 2)[0] > 0
    P = oavi.qp2avi(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    P2 = oavi.qp2avi(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=p.nu)
    sim = oavi.simulate(P2, [[10.0, 10.0]], 500, p.F, p.G, r=[[10.0, 0.0]], uprev=[[0.0, 0.0]])
    y_end = sim["X"][499, 0]                      # ys[:, end] of the reference's Simulation: the state at step N
    assert abs(y_end[0] - 10.0) < 1e-4 and abs(y_end[1]) < 1e-4 and sim["flag_min"][0] == 1, y_end
    rng = np.random.default_rng(1234)
    N = 4000
    # states / references wide enough that 0 .. 6 bounds are active, previous inputs inside the bounds
    theta = np.hstack([rng.uniform(-30, 30, (N, 2)), rng.uniform(-30, 30, (N, 2)), rng.uniform(-1, 1, (N, 2))])
    theta[0] = omm.form_parameter(p, [10.0, 10.0], [10.0, 0.0], [0.0, 0.0])
    X, ef, it, act = oavi.solve_batch(P, theta)
    assert (ef == 1).all()
    for i in range(0, N, 10):
        xp = projection_fixed_point(q.H, q.f + q.f_theta @ theta[i], q.bl, q.bu)
        assert np.abs(xp - X[i]).max() < 1e-9, (i, np.abs(xp - X[i]).max())
    for i in range(N):
        st, pv, sg = oavi.kkt_residual(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, theta[i], X[i], act[i])
        assert st < 1e-8 and pv < 2e-6 and sg < 1e-8, (i, st, pv, sg)
    nact = np.array([bin(int(a)).count("1") for a in act[:, 0]])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "game_kat.npz"),
                        H=q.H, f=q.f, f_theta=q.f_theta, A=q.A, bu=q.bu, bl=q.bl, W=q.W, senses=q.senses,
                        nu=p.nu, nx=p.nx, F=p.F, G=p.G, theta=theta, X=X, exitflag=ef, iters=it, active=act,
                        y_end=y_end, u_first=sim["U"][0, 0], ML=P.ML, MR=P.MR, Gram=P.G, du=P.du0, dl=P.dl0, Dth=P.Dth)
    print("game_kat: n=%d m=%d nth=%d, %d points, active-set sizes %s, mean iterations %.2f, y_end=%s"
          % (q.n, q.m, q.nth, N, np.bincount(nact).tolist(), it.mean(), y_end))


if __name__ == "__main__":
    main()
