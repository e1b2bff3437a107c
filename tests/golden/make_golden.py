"""Generates the committed golden fixtures (tests/golden/*.npz).

Run from the repo root:  python tests/golden/make_golden.py

Each fixture holds (a) the mpQP a LinearMPC.jl user would hand over for one of the benchmark
problems (built by oracle/mpc2mpqp.py, the restatement of /root/reference/src/mpc2mpqp.jl),
(b) the LDP pack (oracle/ldp.py restating /root/reference/src/codegen.jl:239-280), (c) seeded
parameter points and (d) the oracle's answers for them: X*, exit flag, iterations, active-set
mask.  Before anything is written the answers are cross-checked by independent means:

  * pendulum: brute-force KKT enumeration over all 3^5 active-set patterns (unique optimum of a
    strictly convex QP), and the reference's own known answer u = 1.7612519326
    (/root/reference/test/runtests.jl:62-66);
  * mass-spring: KKT residuals of every "optimal" answer and an LP feasibility check
    (scipy.optimize.linprog) of every "infeasible" answer;
  * soft_doc: the documentation's worked example with soft output bounds, known answer u = -1
    (/root/reference/docs/src/manual/simple.md:98-107);
  * prestab: K2, nominal and LQR-prestabilised controllers give the same input
    (/root/reference/test/runtests.jl:119-136);
  * preprocessing: the reference's K4 known answer (/root/reference/test/runtests.jl:1306-1318);
  * satellite4 / satellite20: hybrid MPC with binary thrusters (/root/reference/src/mpc_examples.jl:533-546,
    test/runtests.jl:820-834).  Np=4: every answer equals the best of the 2^8 binary assignments,
    each solved as a plain linear system (no solver involved).  Np=20: the closed loop reaches the
    reference and every binary input sits on a bound (the reference's own assertions).

  * refcond_kat: reference condensation (/root/reference/test/runtests.jl:669-733): the condensed
    controller's first move equals the uncondensed preview controller's on the test's trajectory, and
    a constant trajectory condenses to itself.

  * refprev_full_kat: reference preview without condensation (/root/reference/test/runtests.jl:627-667), every
    fourth answer checked by KKT enumeration.

  * dist_preview_kat: disturbance preview (/root/reference/test/runtests.jl:735-774): theta = [x; r; vec(d_traj)];
    a constant trajectory reproduces the non-preview controller's answers to 1e-12.

`python tests/golden/make_golden.py name ...` rewrites only the named fixtures.

Julia/DAQP cannot run in this image, so no fixture is an output of the reference itself; they pin
the oracle against regressions and carry the reference's known answers.
"""
import itertools
import os
import sys

import numpy as np
from scipy.optimize import linprog

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from oracle import ldp as oldp  # noqa: E402
from oracle import mpc2mpqp as omm  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def kkt_enumerate_box(q, theta):
    """Exact solution of a box-constrained strictly convex QP by enumerating active sets."""
    n = q.n
    f = q.f + q.f_theta @ theta
    for pat in itertools.product((0, 1, -1), repeat=n):
        act = [i for i in range(n) if pat[i]]
        free = [i for i in range(n) if not pat[i]]
        x = np.zeros(n)
        for i in act:
            x[i] = q.bu[i] if pat[i] == 1 else q.bl[i]
        if free:
            x[free] = np.linalg.solve(q.H[np.ix_(free, free)], -(f[free] + q.H[np.ix_(free, act)] @ x[act]))
        g = q.H @ x + f
        if np.any(x > q.bu + 1e-9) or np.any(x < q.bl - 1e-9):
            continue
        if all((-g[i] >= -1e-9) if pat[i] == 1 else (-g[i] <= 1e-9) for i in act):
            return x
    raise RuntimeError("no KKT point found")


def kkt_residuals(q, theta, x, act_words):
    m = q.m
    Afull = np.vstack([np.eye(q.n)[:q.ms], q.A])
    up = [j for j in range(m) if (int(act_words[j >> 6]) >> (j & 63)) & 1]
    lo = [j for j in range(m) if (int(act_words[(m + j) >> 6]) >> ((m + j) & 63)) & 1]
    bu, bl = q.bu + q.W @ theta, q.bl + q.W @ theta
    g = q.H @ x + q.f + q.f_theta @ theta
    E = Afull[up + lo]
    lam = np.linalg.lstsq(E.T, -g, rcond=None)[0] if len(up + lo) else np.zeros(0)
    stat = np.abs(g + E.T @ lam).max() if lam.size else np.abs(g).max()
    pfeas = max((Afull @ x - bu).max(), (bl - Afull @ x).max())
    sign = min([lam[i] for i in range(len(up))] + [-lam[len(up) + i] for i in range(len(lo))] + [0.0])
    return stat, pfeas, sign


def lp_feasible(q, theta):
    Afull = np.vstack([np.eye(q.n)[:q.ms], q.A])
    bu, bl = q.bu + q.W @ theta, q.bl + q.W @ theta
    res = linprog(np.zeros(q.n), A_ub=np.vstack([Afull, -Afull]), b_ub=np.concatenate([bu, -bl]),
                  bounds=[(None, None)] * q.n, method="highs")
    return res.status == 0


ONLY = set(sys.argv[1:])


def save(name, q, L, theta, X, ef, it, act, extra=None):
    if ONLY and name not in ONLY:
        return
    d = dict(H=q.H, f=q.f, f_theta=q.f_theta, A=q.A, bu=q.bu, bl=q.bl, W=q.W, senses=q.senses,
             nu=q.nu, nx=q.nx, M=L.M, du=L.du0, dl=L.dl0, Dth=L.Dth, Rout=L.Rout, x0=L.x0, Xth=L.Xth,
             theta=theta, X=X, exitflag=ef, iters=it, active=act)
    d.update(extra or {})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(f"{name}: N={theta.shape[0]} flags={dict(zip(*np.unique(ef, return_counts=True)))} "
          f"max iters={it.max()}")


def pendulum_theta(rng, N, hard=False):
    """SURVEY.md 8(d) sampling; `hard` draws from the example's +-20 ParameterRange
    (/root/reference/src/mpc_examples.jl:128-134)."""
    if hard:
        x = rng.uniform(-20, 20, (N, 4))
        r = np.stack([rng.uniform(-20, 20, N), np.zeros(N)], 1)
    else:
        x = rng.uniform([-5, -5, -0.3, -2], [5, 5, 0.3, 2], (N, 4))
        r = np.stack([rng.uniform(-5, 5, N), np.zeros(N)], 1)
    return np.hstack([x, r, rng.uniform(-2, 2, (N, 1))])


def main():
    rng = np.random.default_rng(1234)   # seed mirrors /root/reference/test/runtests.jl:8

    # ---- pendulum (BASELINE configs 1-2)
    prob = omm.pendulum()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    theta = np.vstack([pendulum_theta(rng, 1536), pendulum_theta(rng, 512, hard=True)])
    theta[0] = omm.form_parameter(prob, [5.0, 5.0, 0.0, 0.0])
    # up to the 10^4 points SURVEY.md section 8(c) asks for (their own generator: the shared one above feeds
    # every later fixture, whose points must not move)
    rng_more = np.random.default_rng(20240)
    theta = np.vstack([theta, pendulum_theta(rng_more, 6000), pendulum_theta(rng_more, 1952, hard=True)])
    X, ef, it, act = oldp.solve_batch(L, theta)
    assert abs(X[0, 0] - 1.7612519326) < 1e-9, X[0]
    assert np.all(ef == 1)
    for i in range(0, theta.shape[0], 8):
        xe = kkt_enumerate_box(q, theta[i])
        assert np.abs(xe - X[i]).max() < 2e-6, (i, xe, X[i])   # DAQP stops within primal_tol 1e-6
    save("pendulum", q, L, theta, X, ef, it, act, dict(K1_x=[5.0, 5.0, 0, 0], K1_u=1.7612519326))

    # ---- mass-spring chain nm=6 (BASELINE config 3; reference example has one input)
    prob = omm.mass_spring()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    theta = np.vstack([rng.uniform(-4, 4, (384, 12)), rng.uniform(-2, 2, (384, 12)),
                       rng.uniform(-1, 1, (256, 12))])
    X, ef, it, act = oldp.solve_batch(L, theta)
    for i in range(theta.shape[0]):
        if ef[i] == 1:
            stat, pf, sg = kkt_residuals(q, theta[i], X[i], act[i])
            assert stat < 1e-8 and pf < 2e-6 and sg > -1e-9, (i, stat, pf, sg)
        elif i % 4 == 0:
            assert not lp_feasible(q, theta[i]), i
    save("mass_spring", q, L, theta, X, ef, it, act)

    # ---- BASELINE config 3 as worded (3 inputs): synthetic variant, n = 30 -> wavefront kernel
    prob = omm.mass_spring_3in()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    theta = np.vstack([rng.uniform(-4, 4, (192, 12)), rng.uniform(-2, 2, (192, 12)), rng.uniform(-1, 1, (128, 12))])
    X, ef, it, act = oldp.solve_batch(L, theta)
    nchk = 0
    for i in range(theta.shape[0]):
        if ef[i] == 1 and nchk < 120:
            stat, pf, sg = kkt_residuals(q, theta[i], X[i], act[i])
            assert stat < 1e-7 and pf < 2e-6 and sg > -1e-8, (i, stat, pf, sg)
            nchk += 1
        elif ef[i] == -1 and i % 8 == 0:
            assert not lp_feasible(q, theta[i]), i
    save("mass_spring_3in", q, L, theta, X, ef, it, act)

    # ---- K8: documentation example with SOFT output bounds (docs/src/manual/simple.md:60-107)
    prob = omm.doc_simple_soft()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    theta = np.hstack([rng.uniform(-1, 2, (768, 2)), rng.uniform(0, 1, (768, 2)), rng.uniform(-3, 3, (768, 1))])
    theta[0] = omm.form_parameter(prob, [0.5, 1.0], r=[0.0, 0.0])
    X, ef, it, act = oldp.solve_batch(L, theta)
    assert ef[0] >= 1 and abs(X[0, 0] + 1.0) < 1e-6, (ef[0], X[0])      # "u = -1", simple.md:107
    nsoft = int(np.sum((q.senses & 8) != 0))
    assert nsoft > 0 and np.any(ef == 2)
    save("soft_doc", q, L, theta, X, ef, it, act, dict(K8_x=[0.5, 1.0], K8_u=-1.0))

    # ---- K2: prestabilising feedback (test/runtests.jl:119-136): same input as the nominal controller
    pn, pp = omm.prestab_kat(False), omm.prestab_kat(True)
    qn, qq = omm.mpc2mpqp(pn), omm.mpc2mpqp(pp)
    Ln = oldp.qp2ldp(qn.H, qn.f, qn.f_theta, qn.A, qn.bu, qn.bl, qn.W, qn.senses, nout=1)
    Lp = oldp.qp2ldp(qq.H, qq.f, qq.f_theta, qq.A, qq.bu, qq.bl, qq.W, qq.senses, nout=1, K=pp.gain())
    theta = np.hstack([rng.uniform(-0.2, 0.2, (255, 2)), rng.uniform(-1, 1, (255, 1)), np.zeros((255, 1))])
    theta = np.vstack([omm.form_parameter(pn, [0.0, 0.0], r=[1.0, 0.0])[None], theta])
    Xn, efn, _, _ = oldp.solve_batch(Ln, theta)
    X, ef, it, act = oldp.solve_batch(Lp, theta)
    assert efn[0] == 1 and ef[0] == 1 and abs(Xn[0, 0] - X[0, 0]) < 1e-10          # runtests.jl:134
    assert np.linalg.cond(qq.H) < np.linalg.cond(qn.H)                             # runtests.jl:135
    both = (efn >= 1) & (ef >= 1)
    assert np.abs(Xn[both] - X[both]).max() < 1e-8
    save("prestab", qq, Lp, theta, X, ef, it, act, dict(K=pp.gain(), u_nominal=Xn[:, 0], ef_nominal=efn))

    # ---- K4: preprocessing folds Au-only rows into the simple bounds
    q = omm.mpc2mpqp(omm.preprocessing_kat())
    assert q.A.shape[0] == 0 and np.all(q.bu == 0.9) and np.all(q.bl == -0.5)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    theta = rng.uniform(-1, 1, (256, q.nth))
    X, ef, it, act = oldp.solve_batch(L, theta)
    save("preprocessing_kat", q, L, theta, X, ef, it, act)

    # ---- hybrid MPC: satellite with two on/off thrusters (branch and bound over BINARY rows)
    rng = np.random.default_rng(4321)
    prob = omm.satellite(4)
    q = omm.mpc2mpqp(prob)
    assert np.sum((q.senses & 16) != 0) == 8 and q.A.shape[0] == 0
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    theta = np.hstack([rng.uniform(-0.4, 0.4, (256, 1)), rng.uniform(-1, 1, (256, 2)),
                       rng.uniform(-0.5, 0.5, (256, 1)), np.zeros((256, 2))])
    X, ef, it, act = oldp.solve_batch(L, theta)
    assert np.all(ef == 1)
    bins = np.flatnonzero(q.senses & 16)
    free = np.setdiff1d(np.arange(q.n), bins)
    Hff, Hfb = q.H[np.ix_(free, free)], q.H[np.ix_(free, bins)]
    for i in range(theta.shape[0]):
        fth = q.f + q.f_theta @ theta[i]
        bestv, bestx = np.inf, None
        for combo in itertools.product((0, 1), repeat=len(bins)):
            xb = np.where(combo, q.bu[bins], q.bl[bins]) + q.W[bins] @ theta[i]
            xf = np.linalg.solve(Hff, -(fth[free] + Hfb @ xb))     # u1 is unbounded: plain stationarity
            x = np.zeros(q.n)
            x[bins], x[free] = xb, xf
            v = 0.5 * x @ q.H @ x + fth @ x
            if v < bestv:
                bestv, bestx = v, x
        vi = 0.5 * X[i] @ q.H @ X[i] + fth @ X[i]
        assert vi <= bestv + 1e-7 * max(1.0, abs(bestv)), (i, vi, bestv)
        assert np.all(np.minimum(np.abs(X[i, bins] - q.bu[bins]), np.abs(X[i, bins] - q.bl[bins])) < 1e-9)
    save("satellite4", q, L, theta, X, ef, it, act)

    prob = omm.satellite(20)
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    # closed loop of test/runtests.jl:820-834 with a constant reference 0.5 (no preview)
    x = np.zeros(3)
    r = np.array([0.5, 0.0, 0.0])
    traj_theta, traj_u = [], []
    for k in range(40):
        th = omm.form_parameter(prob, x, r=r)
        U, e, _, _ = oldp.solve_batch(L, th[None])
        assert e[0] == 1
        u = U[0, :3]
        for b in prob.binary_controls:
            assert min(abs(u[b] - prob.umin[b]), abs(u[b] - prob.umax[b])) < 1e-5, (k, u)
        traj_theta.append(th)
        traj_u.append(u)
        x = prob.F @ x + prob.G @ u
    assert abs(x[0] - 0.5) < 1e-3, x                                  # runtests.jl:829
    theta = np.vstack([np.array(traj_theta),
                       np.hstack([rng.uniform(-0.3, 0.3, (88, 1)), rng.uniform(-0.5, 0.5, (88, 2)),
                                  rng.uniform(-0.5, 0.5, (88, 1)), np.zeros((88, 2))])])
    X, ef, it, act = oldp.solve_batch(L, theta)
    bins = np.flatnonzero(q.senses & 16)
    ok = ef == 1
    assert ok.sum() >= 100
    assert np.all(np.minimum(np.abs(X[ok][:, bins] - q.bu[bins]), np.abs(X[ok][:, bins] - q.bl[bins])) < 1e-9)
    save("satellite20", q, L, theta, X, ef, it, act,
         dict(F=prob.F, G=prob.G, closed_loop_u=np.array(traj_u), closed_loop_xT=x))

    # ---- the reference's hybrid test AS WORDED (test/runtests.jl:820-834): reference_preview = true,
    # rs = [zeros(1,5) 0.5*ones(1,15); zeros(2,20)], 20 closed-loop steps from x0 = 0; at step k the
    # controller sees rs[:, k+1 .. k+Np] held at the last column (simulation.jl:69-73,101,128-134)
    prob = omm.satellite(20)
    prob.reference_preview = True
    q = omm.mpc2mpqp(prob)
    assert (q.n, q.m, q.nth) == (60, 60, 3 + 3 * 20)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    Nsim = 20
    rs = np.vstack([np.hstack([np.zeros((1, 5)), 0.5 * np.ones((1, Nsim - 5))]), np.zeros((2, Nsim))])
    x = np.zeros(3)
    ths, us, ys = [], [], []
    for k in range(Nsim):
        ys.append(x.copy())
        prev = np.stack([rs[:, min(k + 1 + i, Nsim - 1)] for i in range(prob.Np)], 1)    # get_preview(rs, k+1, Np)
        th = omm.form_parameter(prob, x, r=prev)
        U, e, _, _ = oldp.solve_batch(L, th[None])
        assert e[0] == 1
        u = U[0, :3]
        for b in prob.binary_controls:                                                   # runtests.jl:831-834
            assert min(abs(u[b] - prob.umin[b]), abs(u[b] - prob.umax[b])) < 1e-5, (k, u)
        ths.append(th)
        us.append(u)
        x = prob.F @ x + prob.G @ u
    assert abs(ys[-1][0] - 0.5) < 1e-3, ys[-1]                                           # runtests.jl:829
    rng2 = np.random.default_rng(99)
    extra_th = []
    for _ in range(44):                                  # more points: random states, random step references
        xr = np.hstack([rng2.uniform(-0.2, 0.2), rng2.uniform(-0.3, 0.3, 2)])
        step = rng2.integers(0, 20)
        rr = np.zeros((3, 20))
        rr[0, step:] = rng2.uniform(-0.4, 0.4)
        extra_th.append(omm.form_parameter(prob, xr, r=rr))
    theta = np.vstack([np.array(ths), np.array(extra_th)])
    X, ef, it, act = oldp.solve_batch(L, theta)
    save("satellite20_preview", q, L, theta, X, ef, it, act,
         dict(F=prob.F, G=prob.G, rs=rs, closed_loop_u=np.array(us), closed_loop_y=np.array(ys)))

    # ---- "Codegen Reference Preview - Full" (runtests.jl:627-667): the reference compares Julia with its generated
    # C to 1e-10 on this controller and this trajectory; here the same controller pins mpQP-setup == LDP-setup ==
    # oracle, and the generated-controller entry point fed with vec(r_traj)
    prob = omm.make_mpc([[1, 1], [0, 1]], [[0], [1]], np.eye(2), Np=5, Nc=5, Q=[1.0, 1.0], R=[0.1],
                        umin=[-2.0], umax=[2.0])
    prob.reference_preview = True
    q = omm.mpc2mpqp(prob)
    assert prob.parameter_dims() == (2, 10, 0, 0, 0)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    r_traj = np.array([[0.0, 0.5, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0, 0.0]])
    rngp = np.random.default_rng(627)
    ths = [omm.form_parameter(prob, [0.0, 0.0], r=r_traj)]
    assert np.array_equal(ths[0][2:], r_traj.T.reshape(-1))       # theta = [x; vec(r_traj)] (column by column)
    for _ in range(255):
        ths.append(omm.form_parameter(prob, rngp.uniform(-3, 3, 2), r=rngp.uniform(-2, 2, (2, 5))))
    theta = np.array(ths)
    X, ef, it, act = oldp.solve_batch(L, theta)
    assert np.all(ef == 1)
    for i in range(0, 256, 4):
        xe = kkt_enumerate_box(q, theta[i])
        assert np.abs(xe - X[i]).max() < 2e-6
    save("refprev_full_kat", q, L, theta, X, ef, it, act, dict(r_traj=r_traj, u_julia_equals_c=X[0, :1]))

    # ---- the reference's published benchmark class: pendulum, Np = Nc = N, input + state constraints
    # (docs/src/manual/benchmark.md:4-16; oracle/mpc2mpqp.py::pendulum_benchmark).  theta: the closed loops of the
    # example's two scenarios (mpc_examples.jl:136-139: x0 = [0, 0, 0.15, 0] with r = 0, and x0 = 0 with r = [1, 0];
    # 200 steps of Ts = 0.01 each) -- the points a closed-loop benchmark visits -- plus perturbed copies
    for N in (50, 75, 100, 125):
        name = f"pendulum_N{N}"
        if ONLY and name not in ONLY:
            continue
        prob = omm.pendulum_benchmark(N)
        q = omm.mpc2mpqp(prob)
        L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
        ths = []
        for x0, r in (([0, 0, 0.15, 0], [0.0, 0.0]), ([0, 0, 0, 0], [1.0, 0.0])):
            x, up = np.array(x0, float), np.zeros(1)
            for k in range(200):
                th = omm.form_parameter(prob, x, r=r, uprev=up)
                ths.append(th)
                U, e, _, _ = oldp.solve_batch(L, th[None])
                assert e[0] >= 1
                up = U[0, :1].copy()
                x = prob.F @ x + prob.G @ up
        rngb = np.random.default_rng(1000 + N)
        base = np.array(ths)
        pert = base[rngb.integers(0, len(base), 368)] + rngb.normal(size=(368, 7)) * [0.02, 0.05, 0.005, 0.05, 0.02, 0.0, 0.05]
        theta = np.vstack([base, pert])
        X, ef, it, act = oldp.solve_batch(L, theta)
        assert np.all(ef >= 1)
        if ONLY and name not in ONLY:
            continue
        d = dict(H=q.H, f=q.f, f_theta=q.f_theta, A=q.A, bu=q.bu, bl=q.bl, W=q.W, senses=q.senses, nu=q.nu, nx=q.nx,
                 theta=theta, X=X, exitflag=ef, iters=it, active=act, F=prob.F, G=prob.G, n_closed_loop=len(base))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(f"{name}: n={q.n} m={q.m} N={theta.shape[0]} flags={dict(zip(*np.unique(ef, return_counts=True)))} "
              f"mean iters={it.mean():.1f} max iters={it.max()}")

    # ---- K5: closed-loop end values the reference's tests assert
    # (a) "x0 uncertainty" runtests.jl:1067-1074: x1 -> 0.4 (1e-6); soft output bounds, tightened
    prob = omm.x0_uncertainty_kat()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    x, ths = np.zeros(2), []
    for k in range(1000):                               # Simulation's default N
        th = omm.form_parameter(prob, x, r=[0.5])
        U, e, _, _ = oldp.solve_batch(L, th[None])
        assert e[0] >= 1
        ths.append(th)
        x = prob.F @ x + prob.G @ U[0, :1]
    assert abs(x[0] - 0.4) < 1e-6, x
    rngk = np.random.default_rng(77)
    theta = np.vstack([np.array(ths[:150]), np.array(ths[150::25]),
                       np.hstack([rngk.uniform(-0.6, 0.6, (64, 1)), rngk.uniform(-0.5, 0.5, (64, 1)),
                                  rngk.uniform(-0.6, 0.6, (64, 1))])])
    X, ef, it, act = oldp.solve_batch(L, theta)
    save("x0unc_kat", q, L, theta, X, ef, it, act, dict(F=prob.F, G=prob.G, x_end=x))

    # (b) "Set offset" runtests.jl:1320-1327: us[end] = 10.5, ys[end] = 1.5
    prob = omm.offset_kat()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    x, up, ths, us, ys = np.zeros(1), prob.uprev0.copy(), [], [], []
    for k in range(50):
        ys.append(prob.C @ x + prob.h_offset)
        th = omm.form_parameter(prob, x, r=[1.5], uprev=up)
        U, e, _, _ = oldp.solve_batch(L, th[None])
        assert e[0] == 1
        up = U[0, :1].copy()
        ths.append(th)
        us.append(up)
        x = prob.F @ x + prob.G @ up + prob.f_offset
    assert abs(us[-1][0] - 10.5) < 1e-7 and abs(ys[-1][0] - 1.5) < 1e-7, (us[-1], ys[-1])
    theta = np.vstack([np.array(ths), np.hstack([rngk.uniform(-3, 3, (78, 1)), rngk.uniform(0, 3, (78, 1)),
                                                 rngk.uniform(5, 15, (78, 1))])])
    X, ef, it, act = oldp.solve_batch(L, theta)
    save("offset_kat", q, L, theta, X, ef, it, act, dict(us=np.array(us), ys=np.array(ys)))

    # (c) "Unconstrained" runtests.jl:1329-1335: move_block!([2,2,2,24]), ys[end] = 5.0
    prob = omm.moveblock_kat()
    q = omm.mpc2mpqp(prob)
    assert (q.n, q.m, prob.Nc) == (4, 0, 7)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    x, ths, ys = np.zeros(1), [], []
    for k in range(20):
        ys.append(prob.C @ x)
        th = omm.form_parameter(prob, x, r=[5.0])
        U, e, _, _ = oldp.solve_batch(L, th[None])
        assert e[0] == 1
        ths.append(th)
        x = prob.F @ x + prob.G @ U[0, :1]
    assert abs(ys[-1][0] - 5.0) < 5e-8 * 5.0, ys[-1]                  # Julia's isapprox default
    theta = np.vstack([np.array(ths), rngk.uniform(-5, 5, (108, 2))])
    X, ef, it, act = oldp.solve_batch(L, theta)
    save("moveblock_kat", q, L, theta, X, ef, it, act, dict(F=prob.F, G=prob.G, ys=np.array(ys)))

    # ---- reference condensation (runtests.jl:669-733): the generated controller takes the ny x Np
    # trajectory and collapses it with traj2setpoint (codegen/mpc_update_parameter.c:9-16)
    prob = omm.refcond_kat()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    T2S = prob.traj2setpoint
    rt = np.array([[0.0, 0.5, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0, 0.0]])
    th0 = omm.form_parameter(prob, [0.0, 0.0], r=rt)
    # the uncondensed preview controller gives the same first move on the test's trajectory (the
    # condensation weights the first move 1e6), and a constant trajectory condenses to itself
    prob_full = omm.refcond_kat()
    prob_full.reference_condensation = False
    qf = omm.mpc2mpqp(prob_full)
    Lf = oldp.qp2ldp(qf.H, qf.f, qf.f_theta, qf.A, qf.bu, qf.bl, qf.W, qf.senses, nout=qf.n)
    u_c = oldp.solve_batch(L, th0[None])[0][0, 0]
    u_f = oldp.solve_batch(Lf, omm.form_parameter(prob_full, [0.0, 0.0], r=rt)[None])[0][0, 0]
    assert abs(u_c - u_f) < 1e-5, (u_c, u_f)
    assert np.allclose(T2S @ np.tile([0.7, -0.2], 5), [0.7, -0.2], atol=1e-9)
    rngc = np.random.default_rng(31)
    states = np.vstack([np.zeros((1, 2)), rngc.uniform(-2, 2, (127, 2))])
    trajs = np.vstack([rt.T.reshape(1, -1), np.cumsum(rngc.uniform(-0.5, 0.5, (127, 5, 2)), 1).reshape(127, -1)])
    theta = np.array([omm.form_parameter(prob, states[i], r=trajs[i].reshape(5, 2).T) for i in range(128)])
    X, ef, it, act = oldp.solve_batch(L, theta)
    save("refcond_kat", q, L, theta, X, ef, it, act,
         dict(traj2setpoint=T2S, state=states, reference=trajs, u_full_preview=u_f))

    # ---- disturbance preview (runtests.jl:735-774): theta = [x; r; vec(d_traj)], the generated controller
    # takes the nd x Np trajectory as its `disturbance` argument
    prob = omm.disturbance_preview_kat(True)
    q = omm.mpc2mpqp(prob)
    assert q.nth == 7
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    # a constant trajectory is the non-preview controller's constant disturbance (same H, same optimum)
    prob0 = omm.disturbance_preview_kat(False)
    q0 = omm.mpc2mpqp(prob0)
    L0 = oldp.qp2ldp(q0.H, q0.f, q0.f_theta, q0.A, q0.bu, q0.bl, q0.W, q0.senses, nout=q0.n)
    rngd = np.random.default_rng(41)
    for _ in range(50):
        xx, rr, dd = rngd.uniform(-1, 1, 2), rngd.uniform(-1, 1, 1), rngd.uniform(-0.3, 0.3, 1)
        ua = oldp.solve_batch(L, omm.form_parameter(prob, xx, r=rr, d=np.tile(dd[:, None], (1, 4)))[None])[0]
        ub = oldp.solve_batch(L0, omm.form_parameter(prob0, xx, r=rr, d=dd)[None])[0]
        assert np.abs(ua - ub).max() < 1e-12
    states = np.vstack([np.zeros((1, 2)), rngd.uniform(-1, 1, (127, 2))])
    refs = np.vstack([np.zeros((1, 1)), rngd.uniform(-1, 1, (127, 1))])
    dtraj = np.vstack([np.array([[0.0, 1.0, 1.0, 1.0]]), rngd.uniform(-0.4, 0.4, (127, 4))])
    theta = np.hstack([states, refs, dtraj])
    assert np.array_equal(theta[0], omm.form_parameter(prob, [0, 0], r=[0.0], d=np.array([[0.0, 1, 1, 1]])))
    X, ef, it, act = oldp.solve_batch(L, theta)
    save("dist_preview_kat", q, L, theta, X, ef, it, act, dict(state=states, reference=refs, disturbance=dtraj))


if __name__ == "__main__":
    main()
