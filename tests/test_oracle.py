"""CPU tests of the parity oracle against the reference's known answers and the committed
golden vectors."""
import itertools

import numpy as np
import pytest

from conftest import load_golden, oracle_ldp_from
from oracle import ldp as oldp
from oracle import mpc2mpqp as omm


def test_K1_invpend_known_answer():
    # /root/reference/test/runtests.jl:62-66: compute_control(invpend, [5,5,0,0]) = 1.7612519326 (tol 1e-6)
    prob = omm.pendulum()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=prob.nu)
    X, ef, it, act = oldp.solve_batch(L, omm.form_parameter(prob, [5.0, 5, 0, 0])[None])
    assert ef[0] == 1
    assert abs(X[0, 0] - 1.7612519326) < 1e-6
    assert int(act[0, 0]) == 0b00010            # U_2 at its upper bound, nothing else


def test_pendulum_problem_shape():
    # n=5, m=5 simple bounds, theta=[x(4); r(2); u_prev(1)]
    q = omm.mpc2mpqp(omm.pendulum())
    assert (q.n, q.m, q.ms, q.nth) == (5, 5, 5, 7)
    assert np.allclose(q.H, q.H.T) and np.all(np.linalg.eigvalsh(q.H) > 0)
    assert not np.any(q.W) and not np.any(q.senses)
    assert np.all(q.bu == 2.0) and np.all(q.bl == -2.0)


def test_mass_spring_problem_shape():
    # 54 state-constraint rows generated, one dropped by remove_redundant (mpc2mpqp.jl:742)
    q = omm.mpc2mpqp(omm.mass_spring())
    assert (q.n, q.m, q.ms, q.nth) == (10, 63, 10, 12)


def test_K4_preprocessing_folds_bounds():
    # /root/reference/test/runtests.jl:1306-1318
    q = omm.mpc2mpqp(omm.preprocessing_kat())
    assert q.A.shape[0] == 0
    assert np.all(q.bu == 0.9 * np.ones(10))
    assert np.all(q.bl == -0.5 * np.ones(10))


def _enumerate_box(q, theta):
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
            return x, pat
    raise AssertionError("no KKT point")


def test_oracle_vs_kkt_enumeration_pendulum():
    prob = omm.pendulum()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.n)
    rng = np.random.default_rng(7)
    theta = np.hstack([rng.uniform(-20, 20, (60, 4)), rng.uniform(-20, 20, (60, 1)), np.zeros((60, 1)),
                       rng.uniform(-2, 2, (60, 1))])
    X, ef, _, act = oldp.solve_batch(L, theta)
    assert np.all(ef == 1)
    for i in range(theta.shape[0]):
        xe, pat = _enumerate_box(q, theta[i])
        assert np.abs(xe - X[i]).max() < 2e-6          # the solver stops within primal_tol = 1e-6


def test_K8_doc_example_with_soft_output_bounds():
    # /root/reference/docs/src/manual/simple.md:60-107: "the optimal control action at x=[0.5,1] with
    # r=[0,0] is u=-1"; output bounds are SOFT (setup.jl:94), rho_soft = 1/soft_weight (setup.jl:26)
    prob = omm.doc_simple_soft()
    q = omm.mpc2mpqp(prob)
    assert np.sum((q.senses & omm.SOFT) != 0) == 17 and q.n == 10
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    X, ef, _, _ = oldp.solve_batch(L, omm.form_parameter(prob, [0.5, 1.0], r=[0.0, 0.0])[None])
    assert ef[0] >= 1 and abs(X[0, 0] + 1.0) < 1e-6
    # states that cannot keep the soft output bounds are still solved (exit flag 2 = soft optimal),
    # and the violation stays small: y1 = x1 after one step exceeds its bound by O(rho_soft * lam)
    g = load_golden("soft_doc")
    i = int(np.flatnonzero(g["exitflag"] == 2)[0])
    X, ef, _, _ = oldp.solve_batch(L, g["theta"][i][None])
    assert ef[0] == 2


@pytest.mark.parametrize("name", ["pendulum", "mass_spring", "mass_spring_3in", "preprocessing_kat", "soft_doc", "prestab",
                                  "satellite4", "satellite20", "refcond_kat", "dist_preview_kat"])
def test_oracle_reproduces_golden(name):
    g = load_golden(name)
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    X, ef, it, act = oldp.solve_batch(L, g["theta"])
    assert np.array_equal(ef, g["exitflag"])
    assert np.array_equal(it, g["iters"])
    assert np.array_equal(act, g["active"])
    ok = ef >= 1
    assert np.abs(X[ok] - g["X"][ok]).max() <= 1e-12


@pytest.mark.parametrize("name", ["pendulum", "mass_spring", "mass_spring_3in", "preprocessing_kat", "soft_doc",
                                  "satellite4", "satellite20", "refcond_kat", "dist_preview_kat"])
def test_golden_pack_matches_restated_transform(name):
    g = load_golden(name)
    L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=g["H"].shape[0])
    for k, a in (("M", L.M), ("du", L.du0), ("dl", L.dl0), ("Dth", L.Dth), ("Rout", L.Rout), ("Xth", L.Xth)):
        assert np.allclose(a, g[k], rtol=0, atol=1e-12), k


def test_golden_solutions_satisfy_kkt_mass_spring():
    g = load_golden("mass_spring")
    H, A, n = g["H"], g["A"], g["H"].shape[0]
    ms = g["bu"].size - A.shape[0]
    Afull = np.vstack([np.eye(n)[:ms], A])
    m = g["bu"].size
    checked = 0
    for i in np.flatnonzero(g["exitflag"] == 1)[:200]:
        th, x, a = g["theta"][i], g["X"][i], g["active"][i]
        up = [j for j in range(m) if (int(a[j >> 6]) >> (j & 63)) & 1]
        lo = [j for j in range(m) if (int(a[(m + j) >> 6]) >> ((m + j) & 63)) & 1]
        bu, bl = g["bu"] + g["W"] @ th, g["bl"] + g["W"] @ th
        grad = H @ x + g["f"] + g["f_theta"] @ th
        E = Afull[up + lo]
        lam = np.linalg.lstsq(E.T, -grad, rcond=None)[0] if len(up + lo) else np.zeros(0)
        assert (np.abs(grad + E.T @ lam).max() if lam.size else np.abs(grad).max()) < 1e-8
        assert max((Afull @ x - bu).max(), (bl - Afull @ x).max()) < 2e-6
        assert all(lam[k] > -1e-9 for k in range(len(up))) and all(lam[len(up) + k] < 1e-9 for k in range(len(lo)))
        checked += 1
    assert checked > 50


def test_warm_start_reaches_same_solution():
    # K6 (/root/reference/test/runtests.jl:85-117): cold vs warm start agree (|du| < 1e-9)
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    Xw, efw, itw, actw = oldp.solve_batch(L, g["theta"], warm=g["active"])
    assert np.array_equal(efw, g["exitflag"])
    assert np.abs(Xw - g["X"]).max() < 1e-9
    assert np.array_equal(actw, g["active"])
    assert itw.max() <= 2 and itw.mean() < g["iters"].mean()


def test_equality_and_immutable_rows():
    # sense flags of mpc2mpqp.jl:868-885: an EQUALITY row is active from the start and never leaves,
    # an IMMUTABLE (both bounds infinite) row is never looked at
    rng = np.random.default_rng(3)
    n, nth = 4, 3
    Hh = rng.standard_normal((n, n)); H = Hh @ Hh.T + n * np.eye(n)
    A = rng.standard_normal((3, n))
    bu = np.array([1.0, 1.0, 1e30, 1e30, 0.3, 1e30, 0.5])
    bl = np.array([-1.0, -1.0, -1e30, -1e30, 0.3, -1e30, -0.5])
    sense = np.array([0, 0, 4, 4, 5, 4, 0], np.int32)
    W = np.zeros((7, nth)); W[4] = [0.1, 0, 0]
    f_theta = rng.standard_normal((n, nth))
    L = oldp.qp2ldp(H, np.zeros(n), f_theta, A, bu, bl, W, sense, nout=n)
    theta = rng.uniform(-1, 1, (50, nth))
    X, ef, it, act = oldp.solve_batch(L, theta)
    assert np.all(ef == 1)
    for i in range(50):
        assert abs(A[0] @ X[i] - (0.3 + 0.1 * theta[i, 0])) < 1e-9          # equality holds
        assert (int(act[i, 0]) >> 4) & 1                                   # and is in the working set
        assert np.all(X[i, :2] <= 1 + 1e-6) and np.all(X[i, :2] >= -1 - 1e-6)
        assert abs(A[2] @ X[i]) <= 0.5 + 1e-6


def test_K6_closed_loop_cold_equals_warm():
    # /root/reference/test/runtests.jl:85-117: 100 closed-loop steps from x = [5,5,0,0], cold vs warm
    # started solves give the same inputs (|du| < 1e-9); plant here = the linear prediction model
    prob = omm.pendulum()
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=prob.nu)
    x0 = np.array([[5.0, 5.0, 0.0, 0.0], [0.0, 0.0, 0.15, 0.0], [-3.0, 2.0, 0.1, -1.0]])
    r = np.zeros((3, 2))
    cold = oldp.simulate(L, x0, 100, prob.F, prob.G, r=r, warm=False)
    wrm = oldp.simulate(L, x0, 100, prob.F, prob.G, r=r, warm=True)
    assert np.all(cold["flag_min"] >= 1) and np.all(wrm["flag_min"] >= 1)
    assert np.abs(cold["U"] - wrm["U"]).max() < 1e-9
    # warm == 2: the factorisation itself is kept between two steps (libdaqp's workspace under DAQP_WARMSTART,
    # /root/reference/codegen/mpc_update_qp.c:44-54), in both forms of the solver -- the same criterion
    for mode in (0, 1):
        so = oldp.default_settings(); so.mode = mode
        kept = oldp.simulate(L, x0, 100, prob.F, prob.G, r=r, warm=2, settings=so)
        assert np.all(kept["flag_min"] >= 1) and np.abs(cold["U"] - kept["U"]).max() < 1e-9
    assert abs(cold["U"][0, 0, 0] - 1.7612519326) < 1e-6          # first move of scenario 0 is K1
    assert np.abs(cold["U"]).max() <= 2 + 1e-6                     # |u| <= 2 along the whole run
    assert abs(cold["x"][0, 1]) < 0.5 * abs(x0[0, 1])              # the cart is being braked


def test_K2_prestabilising_feedback_equals_nominal():
    # /root/reference/test/runtests.jl:119-136: |u_nom - u_prestab| < 1e-10 at x = 0, r = [1, 0] and the
    # prestabilised Hessian is better conditioned; with K != 0 the input bounds become general rows
    pn, pp = omm.prestab_kat(False), omm.prestab_kat(True)
    qn, qq = omm.mpc2mpqp(pn), omm.mpc2mpqp(pp)
    assert (qn.ms, qq.ms, qq.m, qq.n) == (30, 0, 30, 30)
    assert np.linalg.cond(qq.H) < np.linalg.cond(qn.H)
    Ln = oldp.qp2ldp(qn.H, qn.f, qn.f_theta, qn.A, qn.bu, qn.bl, qn.W, qn.senses, nout=1)
    Lp = oldp.qp2ldp(qq.H, qq.f, qq.f_theta, qq.A, qq.bu, qq.bl, qq.W, qq.senses, nout=1, K=pp.gain())
    th = omm.form_parameter(pn, [0.0, 0.0], r=[1.0, 0.0])[None]
    un, efn, _, _ = oldp.solve_batch(Ln, th)
    up, efp, _, _ = oldp.solve_batch(Lp, th)
    assert efn[0] == 1 and efp[0] == 1 and abs(un[0, 0] - up[0, 0]) < 1e-10


def test_K3_generalized_parameters_in_constraints():
    # /root/reference/test/runtests.jl:1250-1268: parameter dims (1,1,0,0,1); u = 1.0 at p = 0, 0.25 at p = 0.75
    prob = omm.generalized_parameter_kat()
    assert prob.parameter_dims() == (1, 1, 0, 0, 1)
    q = omm.mpc2mpqp(prob)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    for par, expect in ((0.0, 1.0), (0.75, 0.25)):
        X, ef, _, _ = oldp.solve_batch(L, omm.form_parameter(prob, [0.0], r=[0.0], par=[par])[None])
        assert ef[0] == 1 and abs(X[0, 0] - expect) < 1e-6


def _best_binary_assignment(g, theta):
    """Hybrid problem with unbounded continuous inputs: best of all binary assignments, each a
    plain linear solve (no QP solver involved)."""
    import itertools
    H, bu, bl, W, n = g["H"], g["bu"], g["bl"], g["W"], g["H"].shape[0]
    bins = np.flatnonzero(g["senses"] & 16)
    free = np.setdiff1d(np.arange(n), bins)
    fth = g["f"] + g["f_theta"] @ theta
    best = np.inf
    for combo in itertools.product((0, 1), repeat=len(bins)):
        xb = np.where(combo, bu[bins], bl[bins]) + W[bins] @ theta
        xf = np.linalg.solve(H[np.ix_(free, free)], -(fth[free] + H[np.ix_(free, bins)] @ xb))
        x = np.zeros(n)
        x[bins], x[free] = xb, xf
        best = min(best, 0.5 * x @ H @ x + fth @ x)
    return best


def test_hybrid_branch_and_bound_finds_best_binary_assignment():
    # binaries of the satellite example (mpc_examples.jl:533-546), horizon 4: 2^8 assignments
    g = load_golden("satellite4")
    assert g["A"].shape[0] == 0 and np.sum((g["senses"] & 16) != 0) == 8
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    rng = np.random.default_rng(77)
    theta = np.hstack([rng.uniform(-0.4, 0.4, (40, 1)), rng.uniform(-1, 1, (40, 2)),
                       rng.uniform(-0.5, 0.5, (40, 1)), np.zeros((40, 2))])
    X, ef, it, act = oldp.solve_batch(L, theta)
    assert np.all(ef == 1)
    bins = np.flatnonzero(g["senses"] & 16)
    for i in range(len(theta)):
        fth = g["f"] + g["f_theta"] @ theta[i]
        v = 0.5 * X[i] @ g["H"] @ X[i] + fth @ X[i]
        best = _best_binary_assignment(g, theta[i])
        assert v <= best + 1e-7 * max(1.0, abs(best))
        assert np.all(np.minimum(np.abs(X[i, bins] - g["bu"][bins]), np.abs(X[i, bins] - g["bl"][bins])) < 1e-9)
        # the mask names every binary row as active at one of its bounds
        m = len(g["bu"])
        for j in bins:
            up = (int(act[i, j >> 6]) >> (j & 63)) & 1
            lo = (int(act[i, (m + j) >> 6]) >> ((m + j) & 63)) & 1
            assert up + lo == 1


def test_hybrid_closed_loop_reaches_reference_with_binaries_on_bounds():
    # test/runtests.jl:820-834 (constant reference instead of the preview the reference test uses)
    prob = omm.satellite(20)
    g = load_golden("satellite20")
    L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=3)
    out = oldp.simulate(L, np.zeros((1, 3)), 40, prob.F, prob.G, r=np.array([[0.5, 0.0, 0.0]]))
    assert out["flag_min"][0] == 1
    assert abs(out["x"][0, 0] - 0.5) < 1e-3                               # runtests.jl:829
    U = out["U"][:, 0, :]
    for b in prob.binary_controls:                                        # runtests.jl:831-834
        assert np.all((np.abs(U[:, b] - prob.umin[b]) < 1e-5) | (np.abs(U[:, b] - prob.umax[b]) < 1e-5))
    assert np.abs(U - g["closed_loop_u"]).max() < 1e-9


def test_f32_oracle_tracks_the_f64_oracle():
    # binary32 build of the same source (oracle/daqp_ldp_oracle_f32.c): same outcome as binary64 on
    # well-conditioned problems, x to single-precision accuracy, identical active sets where both solve
    for name, nout in (("pendulum", 1), ("satellite4", 12)):
        g = load_golden(name)
        L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
        th = g["theta"][:1500]
        X, ef, it, act = oldp.solve_batch(L, th)
        Xf, eff, itf, actf = oldp.solve_batch(L, th, dtype=np.float32)
        assert Xf.dtype == np.float32
        ok = (ef >= 1) & (eff >= 1)
        assert ok.mean() > 0.98
        assert np.abs(X[ok] - Xf[ok]).max() < 1e-3
        assert (act[ok] == actf[ok]).all(axis=1).mean() > 0.99


def test_reference_preview_condensing():
    # mpc2mpqp.jl:535-577 ref_preview_cost: with a constant reference the preview problem has the same
    # optimum as the plain one; theta grows from ny to ny*Np reference entries
    for mk in (omm.pendulum, lambda: omm.satellite(6)):
        p0, p1 = mk(), mk()
        p1.reference_preview = True
        q0, q1 = omm.mpc2mpqp(p0), omm.mpc2mpqp(p1)
        assert q1.nth == q0.nth + p0.ny * (p0.Np - 1)
        assert np.allclose(q0.H, q1.H) and np.array_equal(q0.bu, q1.bu)
        L0 = oldp.qp2ldp(q0.H, q0.f, q0.f_theta, q0.A, q0.bu, q0.bl, q0.W, q0.senses, nout=q0.n)
        L1 = oldp.qp2ldp(q1.H, q1.f, q1.f_theta, q1.A, q1.bu, q1.bl, q1.W, q1.senses, nout=q1.n)
        rng = np.random.default_rng(3)
        for _ in range(6):
            x, r, up = rng.uniform(-.3, .3, p0.nx), rng.uniform(-.5, .5, p0.ny), rng.uniform(-1, 1, p0.nu)
            X0, e0, _, _ = oldp.solve_batch(L0, omm.form_parameter(p0, x, r=r, uprev=up)[None])
            X1, e1, _, _ = oldp.solve_batch(L1, omm.form_parameter(p1, x, r=r, uprev=up)[None])
            assert e0[0] == e1[0] == 1 and np.abs(X0 - X1).max() < 1e-9


def test_hybrid_preview_fixture_is_reproduced():
    # the reference's hybrid test as worded (runtests.jl:820-834): fixture answers from the oracle
    g = load_golden("satellite20_preview")
    L = oracle_ldp_from({k: g[k] for k in ("M", "du", "dl", "Dth", "Rout", "x0", "Xth", "senses")} | {"ms": 60})
    X, ef, it, act = oldp.solve_batch(L, g["theta"][:24])
    assert np.array_equal(ef, g["exitflag"][:24]) and np.abs(X - g["X"][:24]).max() < 1e-9
    assert abs(g["closed_loop_y"][-1, 0] - 0.5) < 1e-3


def test_K5_closed_loop_end_values():
    # end values the reference's closed-loop tests assert, through the restated
    # condensing (offsets, move blocking, constraint tightening) and the oracle solver
    # (a) runtests.jl:1067-1074 "x0 uncertainty": x1 -> 0.4 within 1e-6 (soft output bound 0.5 - |C| dx0)
    p = omm.x0_uncertainty_kat()
    q = omm.mpc2mpqp(p)
    assert (q.n, q.m, q.nth) == (25, 49, 3) and int(((q.senses & 8) != 0).sum()) == 24
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    x = np.zeros(2)
    for _ in range(400):
        U, e, _, _ = oldp.solve_batch(L, omm.form_parameter(p, x, r=[0.5])[None])
        assert e[0] >= 1
        x = p.F @ x + p.G @ U[0]
    assert abs(x[0] - 0.4) < 1e-6
    # (b) runtests.jl:1320-1327 "Set offset": us[end] = 10.5, ys[end] = 1.5
    p = omm.offset_kat()
    q = omm.mpc2mpqp(p)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    x, up = np.zeros(1), p.uprev0.copy()
    for _ in range(50):
        y = p.C @ x + p.h_offset
        U, e, _, _ = oldp.solve_batch(L, omm.form_parameter(p, x, r=[1.5], uprev=up)[None])
        up = U[0].copy()
        x = p.F @ x + p.G @ up + p.f_offset
    assert abs(up[0] - 10.5) < 1e-7 and abs(y[0] - 1.5) < 1e-7
    # (c) runtests.jl:1329-1335 "Unconstrained": move_block!([2,2,2,24]) -> four moves, ys[end] = 5.0
    p = omm.moveblock_kat()
    q = omm.mpc2mpqp(p)
    assert p.move_blocks == [[2, 2, 2, 4]] and p.Nc == 7 and (q.n, q.m) == (4, 0)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    x = np.zeros(1)
    for _ in range(20):
        y = p.C @ x
        U, e, _, _ = oldp.solve_batch(L, omm.form_parameter(p, x, r=[5.0])[None])
        x = p.F @ x + p.G @ U[0]
    assert abs(y[0] - 5.0) < 5e-8 * 5.0


@pytest.mark.parametrize("name", ["x0unc_kat", "offset_kat", "moveblock_kat"])
def test_K5_fixtures_are_reproduced(name):
    g = load_golden(name)
    L = oracle_ldp_from({k: g[k] for k in ("M", "du", "dl", "Dth", "Rout", "x0", "Xth", "senses")} |
                        {"ms": int(g["bu"].size - g["A"].shape[0])})
    X, ef, it, act = oldp.solve_batch(L, g["theta"])
    assert np.array_equal(ef, g["exitflag"]) and np.array_equal(it, g["iters"]) and np.array_equal(act, g["active"])
    assert np.abs(X - g["X"]).max() < 1e-9


def _preview_sim(prev, N=20):
    rt = np.zeros((2, N)); rt[0, 10:] = 1.0
    p = omm.preview_sim_kat(prev)
    q = omm.mpc2mpqp(p)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    x, us, ys = np.array([1.0, 0.0]), [], []
    for k in range(N):
        ys.append(p.C @ x)
        r = np.stack([rt[:, min(k + 1 + i, N - 1)] for i in range(p.Np)], 1) if prev else rt[:, k]
        U, e, _, _ = oldp.solve_batch(L, omm.form_parameter(p, x, r=r)[None])
        assert e[0] >= 1
        us.append(U[0])
        x = p.F @ x + p.G @ U[0]
    return np.array(us).T, np.array(ys).T, rt


def test_reference_preview_simulation_assertions():
    # /root/reference/test/runtests.jl:276-327: preview changes the inputs, lowers the tracking error
    # (ratio < 0.9) and both loops meet the reference at the end (1e-3)
    up, yp, rt = _preview_sim(True)
    un, yn, _ = _preview_sim(False)
    assert np.linalg.norm(up - un) > 1e-1
    ep, en = yp - rt, yn - rt
    assert np.linalg.norm(ep) / np.linalg.norm(en) < 0.9
    assert np.linalg.norm(ep[:, -1]) < 1e-3 and np.linalg.norm(en[:, -1]) < 1e-3


def test_reference_condensation_restated():
    """/root/reference/test/runtests.jl:669-733 (reference_preview + reference_condensation) and
    src/mpc2mpqp.jl:550-569: theta carries ONE setpoint = traj2setpoint * vec(r_traj); the test's own
    check of mpc_update_parameter is theta == [x; condense_reference(r_traj); u]."""
    from oracle import mpc2mpqp as omm
    g = load_golden("refcond_kat")
    p = omm.refcond_kat()
    q = omm.mpc2mpqp(p)
    assert q.nth == 4 and p.traj2setpoint.shape == (2, 10)          # x(2) + one setpoint(2), no u_prev (Rr = 0)
    assert np.abs(p.traj2setpoint - g["traj2setpoint"]).max() < 1e-9
    rt = np.array([[0.0, 0.5, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0, 0.0]])
    th = omm.form_parameter(p, [0.0, 0.0], r=rt)
    assert np.abs(th - np.concatenate([[0, 0], p.traj2setpoint @ rt.T.reshape(-1)])).max() < 1e-12
    assert np.abs(th - g["theta"][0]).max() < 1e-9
    # a constant trajectory condenses to itself, and then the controller is the plain tracking one
    assert np.abs(p.traj2setpoint @ np.tile([0.3, -0.1], 5) - [0.3, -0.1]).max() < 1e-9
    p2 = omm.refcond_kat()
    p2.reference_preview = p2.reference_condensation = False
    q2 = omm.mpc2mpqp(p2)
    assert np.abs(q.f_theta - q2.f_theta).max() < 1e-12 and np.abs(q.H - q2.H).max() == 0
    # on the test's trajectory the first move equals the uncondensed preview controller's
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    u = oldp.solve_batch(L, th[None])[0][0, 0]
    assert abs(u - float(g["u_full_preview"])) < 1e-5


def test_observer_restated():
    """/root/reference/src/observer.jl:53-72,104-123 and codegen/mpc_observer.c: the Kalman gain solves the
    filter Riccati equation, predict!/correct! equal the generated C loops on the generated arrays
    (the reference's own check, runtests.jl:936-947, to 1e-9)."""
    from oracle import observer as oobs
    from oracle import mpc2mpqp as omm
    p = omm.pendulum(Np=100, Nc=100)
    kf = oobs.kalman_filter(p.F, p.G, p.C, Q=1e2 * np.array([1e-3, 1, 1e-3, 1]), R=[1, 0.1])
    nx, nu, nd, ny = kf.dims
    assert (nx, nu, nd, ny) == (4, 1, 0, 2)
    # K = P C'(C P C' + R)^-1 with P the stabilising solution: F (I - K C) is a stable matrix
    assert np.abs(np.linalg.eigvals(kf.F @ (np.eye(nx) - kf.K @ kf.C))).max() < 1.0
    dyn, meas, kt = kf.codegen_arrays()
    assert dyn.size == nx * (1 + nx + nu + nd) and meas.size == ny * (1 + nx + nd) and kt.size == ny * nx
    rng = np.random.default_rng(3)
    for _ in range(20):
        x, u, y = rng.standard_normal(nx), rng.standard_normal(nu), rng.standard_normal(ny)
        x1 = kf.predict(x, u)
        assert np.linalg.norm(oobs.c_predict(dyn, x, u, None, nx, nu, nd) - x1) < 1e-9
        assert np.linalg.norm(oobs.c_correct(meas, kt, x1, y, None, nx, ny, nd) - kf.correct(x1, y)) < 1e-9
    # with a measured disturbance (runtests.jl:951-987 "Observer + disturbance")
    kd = oobs.kalman_filter([[1, 1], [0, 1.0]], [[0], [1.0]], [[1.0, 0]], Gd=[[0.5], [1.0]], Dd=[[0.1]],
                            f_offset=[0.1, -0.2], h_offset=[0.3], Q=[1.0, 1], R=[1e-2])
    dyn, meas, kt = kd.codegen_arrays()
    x, u, y, d = rng.standard_normal(2), rng.standard_normal(1), rng.standard_normal(1), rng.standard_normal(1)
    assert np.linalg.norm(oobs.c_predict(dyn, x, u, d, 2, 1, 1) - kd.predict(x, u, d)) < 1e-9
    assert np.linalg.norm(oobs.c_correct(meas, kt, x, y, d, 2, 1, 1) - kd.correct(x, y, d)) < 1e-9


def _observer_disturbance_loop(seed, solve):
    """Simulation of /root/reference/test/runtests.jl:951-961 (src/simulation.jl:93-113 with an observer and a
    constant measured disturbance): y_meas = x1 + d2 + noise -> correct!(y, d) -> compute_control(xhat; r = 0, d)
    -> predict!(u, d) -> x+ = F x + G u + Gd d.  Returns the outputs y = C x + Dd d."""
    from oracle import observer as oobs
    p = omm.observer_disturbance_kat()
    kf = oobs.kalman_filter(p.F, p.G, p.C, Gd=p.Gd, Dd=p.Dd, Q=[1.0, 1], R=[1e-2])
    rng = np.random.default_rng(seed)
    x, xh, d, ys = np.array([1.0, 0.0]), np.array([1.0, 0.0]), np.array([1.0, 1.0]), []
    for _ in range(100):
        ym = np.array([x[0] + d[1] + 0.01 * rng.standard_normal()])
        ys.append((p.C @ x + p.Dd @ d)[0])
        xh = kf.correct(xh, ym, d)
        u = solve(omm.form_parameter(p, xh, r=[0.0], d=d))
        xh = kf.predict(xh, u, d)
        x = p.F @ x + p.G @ u + p.Gd @ d
    return np.array(ys)


def test_measured_disturbance_with_observer_reference_assertion():
    # "Observer + disturbance": the controller knows d through theta = [x; r; d] (extended system,
    # mpc2mpqp.jl:664-669), the Kalman filter through Gd / Dd; the reference asserts |mean(ys[end-20:end])| < 1e-2
    p = omm.observer_disturbance_kat()
    q = omm.mpc2mpqp(p)
    assert (q.n, q.m, q.nth) == (10, 0, 5)
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)

    def solve(th):
        U, ef, _, _ = oldp.solve_batch(L, th[None])
        assert ef[0] >= 1
        return U[0, :1]

    for seed in range(5):
        ys = _observer_disturbance_loop(seed, solve)
        assert abs(np.mean(ys[-21:])) < 1e-2
    # without the disturbance in theta the offset stays: the same loop with d hidden from the controller
    def solve_blind(th):
        th = th.copy(); th[3:5] = 0.0
        return solve(th)
    assert abs(np.mean(_observer_disturbance_loop(0, solve_blind)[-21:])) > 0.1


def test_disturbance_preview_restated():
    """/root/reference/test/runtests.jl:735-774 and src/mpc2mpqp.jl:48-66,579-604: theta = [x; r; vec(d_traj)]
    (the test's own check of mpc_update_parameter), and a constant trajectory is the non-preview controller."""
    g = load_golden("dist_preview_kat")
    p = omm.disturbance_preview_kat(True)
    q = omm.mpc2mpqp(p)
    assert q.nth == 7 and np.abs(q.f_theta - g["f_theta"]).max() < 1e-12
    th = omm.form_parameter(p, [0.0, 0.0], r=[0.0], d=np.array([[0.0, 1.0, 1.0, 1.0]]))
    assert np.array_equal(th, np.array([0, 0, 0, 0, 1.0, 1, 1]))
    p0 = omm.disturbance_preview_kat(False)
    q0 = omm.mpc2mpqp(p0)
    assert q0.nth == 4 and np.abs(q.H - q0.H).max() == 0
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=1)
    L0 = oldp.qp2ldp(q0.H, q0.f, q0.f_theta, q0.A, q0.bu, q0.bl, q0.W, q0.senses, nout=1)
    rng = np.random.default_rng(2)
    for _ in range(20):
        x, r, d = rng.uniform(-1, 1, 2), rng.uniform(-1, 1, 1), rng.uniform(-0.3, 0.3, 1)
        a = oldp.solve_batch(L, omm.form_parameter(p, x, r=r, d=np.tile(d[:, None], (1, 4)))[None])[0]
        b = oldp.solve_batch(L0, omm.form_parameter(p0, x, r=r, d=d)[None])[0]
        assert np.abs(a - b).max() < 1e-12


def test_parameter_preview_restated():
    """/root/reference/test/runtests.jl:1136-1155 and :1270-1304 (src/mpc2mpqp.jl:125-145,162,478-508): with
    parameter_preview theta carries one generalised parameter per predicted step."""
    p2 = omm.make_mpc([[1, 1], [0, 1]], [[0], [1]], np.eye(2), Np=5, Nc=3, Q=[1.0, 1.0], R=[0.1], umin=[-2.0], umax=[2.0])
    p2.Eu = np.array([[1.0]])
    p2.parameter_preview = True
    assert p2.parameter_dims() == (2, 2, 0, 0, 5)                                    # :1147
    q2 = omm.mpc2mpqp(p2)
    L2 = oldp.qp2ldp(q2.H, q2.f, q2.f_theta, q2.A, q2.bu, q2.bl, q2.W, q2.senses, nout=1)
    th_c = omm.form_parameter(p2, [-1.0, 0.0], r=[0.0, 0.0], par=np.array([1.0]))
    th_p = omm.form_parameter(p2, [-1.0, 0.0], r=[0.0, 0.0], par=np.array([[1.0, 0.0, 0.0, 0.0, 0.0]]))
    assert np.array_equal(th_c[4:], np.ones(5)) and np.array_equal(th_p[4:], [1.0, 0, 0, 0, 0])   # :1148-1149
    uc = oldp.solve_batch(L2, th_c[None])[0][0]
    up = oldp.solve_batch(L2, th_p[None])[0][0]
    assert np.linalg.norm(uc - up) > 1e-3                                           # :1153
    # the explicit-preview codegen problem has a closed form: Q = 0 decouples the moves, u_k = clip(2 p_k, 0, 2)
    p = omm.parameter_preview_kat()
    q = omm.mpc2mpqp(p)
    assert q.nth == 5
    L = oldp.qp2ldp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=3)
    rng = np.random.default_rng(4)
    P = rng.uniform(-0.5, 1.5, (200, 3))
    theta = np.hstack([np.zeros((200, 2)), P])
    X, ef, _, _ = oldp.solve_batch(L, theta)
    assert np.all(ef == 1) and np.abs(X - np.clip(2 * P, 0, 2)).max() < 1e-12


@pytest.mark.parametrize("name", ["soft_doc", "x0unc_kat", "pendulum_N50", "pendulum_N50_active"])
def test_soft_rows_equal_the_explicit_slack_qp(name):
    """Pins the SOFT-row convention to the reference's own definition.  /root/reference/src/utils.jl:329-364
    (make_singlesided) writes a soft row out as an explicit QP: one slack per soft row entering both sides of the row
    with the coefficient -norm_factors[i] (the row's norm in least-distance coordinates) and the cost soft_weight I,
    soft_weight = 1 / rho_soft (/root/reference/src/setup.jl:26).  That QP -- n + #soft variables, hard rows only --
    is built by tests/golden/make_soft_explicit.py and solved by the oracle's HARD path; the SOFT-flag path
    (rho_soft added to the pivot of a soft row in normalised-row units, oracle/daqp_ldp_oracle.c ldl_add) must give
    the same U.  Conditioning: the slack weight 1 / rho amplifies rounding, observed 1e-8 at rho = 1e-3 and 1e-5 at
    1e-6 (ratio = the ratio of the weights); a WRONG unit (slack measured in the un-normalised row, nf = 1) is off
    by > 1e-3 at either weight, which the last assertion shows, so the bound discriminates."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_soft_explicit",
                                                  os.path.join(os.path.dirname(__file__), "golden", "make_soft_explicit.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    # ("pendulum_N50_active": the reference's benchmark class, N = 50, from starts BEYOND its soft output bounds --
    # 23 of its 24 points end soft-optimal with slacks up to 2e2 in normalised-row units; the closed-loop sample of
    # "pendulum_N50" barely touches those rows)
    g = load_golden(name.replace("_active", ""))
    kat = load_golden("soft_explicit_kat")
    th = kat[f"{name}_theta"]
    n = g["H"].shape[0]
    for rho, tol in ((1e-3, 1e-7), (1e-6, 1e-4)):
        Us, efs = mk.solve_soft(g, th, rho)
        assert np.all(efs >= 1)
        # (a) against the committed explicit-slack answers
        Uk = kat[f"{name}_U_rho{rho:g}"]
        assert np.all(kat[f"{name}_flag_rho{rho:g}"] >= 1)
        assert np.abs(Us - Uk).max() <= tol, (rho, np.abs(Us - Uk).max())
        # (b) ... which this checkout reproduces from the reference's definition
        U, eps, ef = mk.solve_explicit(g, th, 1.0 / rho)
        assert np.all(ef >= 1) and np.abs(U - Uk).max() <= tol
        # exit flag 2 (soft optimal) <=> rho sum lam_i^2 > primal_tol, and the explicit slack of a soft row in
        # normalised-row units is eps_i = rho lam_i: the same quantity from the explicit QP is sum eps_i^2 / rho
        ssq = (eps ** 2).sum(axis=1) / rho
        assert np.all(ssq[efs == 1] <= 1.5e-6) and np.all(ssq[efs == 2] > 0.67e-6)
    # discriminating power: the same explicit QP with the slack in UN-normalised row units is a different problem
    if name != "pendulum_N50":                       # (its sampled points barely touch the soft rows)
        q = mk.explicit_slack_qp(g, 1e3)
        A = q["A"].copy()
        A[:, n:] = np.sign(A[:, n:])                 # nf = 1
        L = oldp.qp2ldp(q["H"], q["f"], q["f_theta"], A, q["bu"], q["bl"], q["W"], q["senses"], nout=n)
        Xw, efw, _, _ = oldp.solve_batch(L, th)
        Us, efs = mk.solve_soft(g, th, 1e-3)
        assert np.abs(Xw - Us)[(efw >= 1) & (efs == 2)].max() > 1e-3


@pytest.mark.parametrize("name,xtol", [("pendulum", 1e-10), ("mass_spring", 1e-10), ("mass_spring_3in", 1e-9),
                                       ("preprocessing_kat", 1e-10), ("soft_doc", 1e-4), ("pendulum_N75", 1e-4),
                                       ("x0unc_kat", 1e-6)])
def test_gram_scan_twin_takes_the_same_decisions(name, xtol):
    """The oracle's mode 1 (the checker of the wavefront kernel's Gram-scan form: row values from Gram columns, dual
    objective from the factorisation, pairwise 64-leaf trees for the append's dot products) against mode 0 (the
    n-chain form, what libdaqp does per iteration) on the committed fixtures: on every solvable point the same exit
    flag, iteration count and final active set, x to rounding (the 1 / rho_soft penalty amplifies it on SOFT rows);
    and mode 1 reproduces the committed answers like mode 0 does.  On infeasible, nearly dependent problems the two
    forms may end with different FAILURE flags (-1 / -2: the guard against a broken working set looks at different
    roundings), never on different sides of 'solved'."""
    g = load_golden(name)
    L = oracle_ldp_from(g) if "M" in g else oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"],
                                                         g["senses"], nout=int(g["nu"]))
    th = np.asarray(g["theta"], float)[:1500]
    s0, s1 = oldp.default_settings(), oldp.default_settings()
    s1.mode = 1
    X0, e0, i0, a0 = oldp.solve_batch(L, th, s0)
    X1, e1, i1, a1 = oldp.solve_batch(L, th, s1)
    ok = e0 >= 1
    assert np.array_equal(ok, e1 >= 1)
    assert np.array_equal(e0[ok], e1[ok]) and np.array_equal(i0[ok], i1[ok]) and np.array_equal(a0[ok], a1[ok])
    assert np.abs(X0[ok] - X1[ok]).max() <= xtol
    assert set(np.unique(e1[~ok])) <= {-1, -2}
    assert np.array_equal(e1[ok], np.asarray(g["exitflag"])[:1500][ok])
    # warm start from the final sets: the same sets come back in one iteration-equivalent
    Xw, ew, iw, aw = oldp.solve_batch(L, th[ok][:200], s1, warm=a1[ok][:200])
    assert np.array_equal(ew, e1[ok][:200]) and np.array_equal(aw, a1[ok][:200])
    # binary32 build of the same mode
    s32 = oldp.default_settings_f32(); s32.mode = 1
    Xf, ef, _, _ = oldp.solve_batch(L, th[:300].astype(np.float32), s32, dtype=np.float32)
    Xc, ec, _, _ = oldp.solve_batch(L, th[:300].astype(np.float32), oldp.default_settings_f32(), dtype=np.float32)
    assert Xf.dtype == np.float32 and ((ef >= 1) == (ec >= 1)).mean() > 0.9


@pytest.mark.parametrize("name", ["soft_doc", "x0unc_kat", "pendulum_N50", "pendulum_N50_active"])
def test_soft_path_solution_is_a_kkt_point_of_the_explicit_slack_qp(name):
    """A certificate that involves no solver at all.  For the reference's explicit-slack QP (utils.jl:329-364: slack
    eps_i per soft row with coefficient -nf_i on both sides, cost soft_weight I) the KKT conditions at a point U are
        eps_i = max(0, violation_i) / nf_i                      (the slack a violated soft row needs, zero otherwise)
        mu_i  = soft_weight * eps_i / nf_i                       (stationarity in eps_i: the multiplier is DETERMINED)
        H U + f(theta) + sum_soft +-A_i' mu_i + sum_hard-active +-A_j' mu_j = 0,  mu_j >= 0   (stationarity in U)
    so for the U the SOFT-flag path returns, the residual  -(H U + f) - (soft terms)  must lie in the cone of the
    active HARD rows' normals: a non-negative least-squares problem whose optimum must be ~0, and every hard row must
    hold.  Checked at rho_soft = 1e-3 (the same semantics as 1e-6, without its 1e6 amplification of rounding)."""
    from scipy.optimize import nnls
    g = load_golden(name.replace("_active", ""))
    kat = load_golden("soft_explicit_kat")
    th = kat[f"{name}_theta"][:40]
    rho = 1e-3
    w = 1.0 / rho
    H, f, fth = np.asarray(g["H"], float), np.asarray(g["f"], float).ravel(), np.asarray(g["f_theta"], float)
    n = H.shape[0]
    A = np.asarray(g["A"], float).reshape(-1, n)
    bu, bl, W = np.asarray(g["bu"], float), np.asarray(g["bl"], float), np.asarray(g["W"], float)
    m = bu.size
    ms = m - A.shape[0]
    A0 = np.vstack([np.eye(n)[:ms], A])
    soft = (np.asarray(g["senses"]) & 8) != 0
    imm = (np.asarray(g["senses"]) & 4) != 0
    Rl = np.linalg.cholesky((H + H.T) / 2)
    nf = np.linalg.norm(np.linalg.solve(Rl, A0.T).T, axis=1)
    L = oldp.qp2ldp(H, f, fth, A, bu, bl, W, g["senses"], nout=n)
    s = oldp.default_settings(); s.rho_soft = rho
    U, ef, _, _ = oldp.solve_batch(L, th, s)
    assert np.all(ef >= 1)
    worst = 0.0
    for k in range(len(th)):
        u = U[k]
        up = bu + W @ th[k]; lo = bl + W @ th[k]
        Au = A0 @ u
        vu, vl = Au - up, lo - Au                                   # > 0: violated
        hard = ~soft & ~imm
        scale = 1.0 + np.abs(Au)
        assert (vu[hard] <= 2e-6 * scale[hard] * np.maximum(nf[hard], 1)).all() and (vl[hard] <= 2e-6 * scale[hard] * np.maximum(nf[hard], 1)).all()
        grad = H @ u + f + fth.reshape(n, -1) @ th[k]
        res = -grad
        cols = [A0[j] for j in np.flatnonzero(hard & (np.abs(vu) <= 1e-5 * np.maximum(nf, 1)))] + \
               [-A0[j] for j in np.flatnonzero(hard & (np.abs(vl) <= 1e-5 * np.maximum(nf, 1)))]
        for i in np.flatnonzero(soft):
            # clearly violated soft row: its multiplier is determined by the violation; a soft row ON its bound --
            # within the solver's primal_tol (1e-6 in normalised-row units = 1e-6 nf_i here), where "violated" and
            # "satisfied" are the same answer and w v / nf^2 would turn the tolerance into a force -- may carry any
            # multiplier >= 0: it joins the cone
            tv = 2e-6 * nf[i]
            if vu[i] > tv: res -= A0[i] * (w * vu[i] / nf[i] ** 2)
            elif vu[i] >= -tv: cols.append(A0[i])
            if vl[i] > tv: res += A0[i] * (w * vl[i] / nf[i] ** 2)
            elif vl[i] >= -tv: cols.append(-A0[i])
        if cols:
            _, rn = nnls(np.array(cols).T, res)
        else:
            rn = np.linalg.norm(res)
        worst = max(worst, rn / (1.0 + np.linalg.norm(grad)))
    assert worst <= 1e-5, worst


def test_closed_loop_with_a_kept_factorisation_soft_rows_and_many_scenarios():
    """warm == 2 on the benchmark class (pendulum N = 50, soft state rows) and the doc example with soft rows:
    the kept working set continues through removals, soft rows and scenarios at rest, and ends on the cold loop's
    inputs; iterations per step drop against the cold loop."""
    for name in ("pendulum_N50", "soft_doc"):
        g = load_golden(name)
        L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"].astype(np.int32), nout=1)
        rng = np.random.default_rng(5)
        if name == "soft_doc":
            prob = omm.doc_simple_soft()
            F, G = prob.F, prob.G
            N, T = 60, 25
            x0 = rng.uniform(-0.2, 0.7, (N, 2)); r = np.tile([1.0, 0.0], (N, 1)); r[N // 2:] = [0.3, 0.0]
        else:
            F, G = g["F"], g["G"]
            N, T = 40, 30
            base = g["theta"][:int(g["n_closed_loop"])]
            pick = base[rng.integers(0, len(base), N)]
            x0, r = pick[:, :4].copy(), pick[:, 4:6].copy()
        for mode in (0, 1):
            so = oldp.default_settings(); so.mode = mode
            cold = oldp.simulate(L, x0, T, F, G, r=r, warm=False, settings=so)
            mask = oldp.simulate(L, x0, T, F, G, r=r, warm=True, settings=so)
            kept = oldp.simulate(L, x0, T, F, G, r=r, warm=2, settings=so)
            ok = (cold["flag_min"] >= 1) & (kept["flag_min"] >= 1) & (mask["flag_min"] >= 1)
            assert ok.mean() > 0.9, (name, mode)
            # the same optimum up to the tolerances: a soft row inside the primal_tol band (1e-6) may end on either
            # side of it depending on where the iterations started (2e-7 in u on the N = 50 problem)
            assert np.abs(mask["U"][:, ok] - kept["U"][:, ok]).max() < 1e-5, (name, mode)
            assert np.abs(cold["U"][:, ok] - kept["U"][:, ok]).max() < 1e-5, (name, mode)


def test_kept_factorisation_through_the_64_row_limit_on_the_cpu():
    """oracle_simulate warm = 2 where working sets cross 64 rows (the wavefront kernel's capacity, which the twin
    mirrors): a step that wants more rows is re-solved from the mask in the n-chain form and nothing is kept after it;
    the loop stays on the optimum of the cold loop (soft rows: up to the primal tolerance band) in both forms."""
    rng = np.random.default_rng(5)
    n, mg, nth = 6, 150, 2
    Hh = rng.standard_normal((n, n)); H = Hh @ Hh.T + n * np.eye(n)
    A = rng.standard_normal((mg, n)); m = n + mg
    bu = rng.uniform(0.5, 2.0, m); bl = -rng.uniform(0.5, 2.0, m)
    W = 0.3 * rng.standard_normal((m, nth)); W[:n] = 0.0
    W[n:, 0] = np.abs(W[n:, 0]) + 0.5
    f_theta = rng.standard_normal((n, nth))
    sense = np.zeros(m, np.int32); sense[n:] = 8
    L = oldp.qp2ldp(H, np.zeros(n), f_theta, A, bu, bl, W, sense, nout=2)
    N, T = 40, 25
    x0 = np.vstack([rng.uniform(-1, 1, (10, 2)), np.hstack([rng.uniform(20, 60, (N - 10, 1)), rng.uniform(-1, 1, (N - 10, 1))])])
    Fm = np.array([[0.8, 0.05], [0.0, 0.7]])
    Gm = 0.05 * rng.standard_normal((2, 2))
    _, _, _, act0 = oldp.solve_batch(L, x0)
    nact0 = np.array([sum(bin(int(w)).count("1") for w in row) for row in act0])
    assert (nact0 > 64).sum() >= 20
    for mode in (0, 1):
        so = oldp.default_settings(); so.mode = mode
        cold = oldp.simulate(L, x0, T, Fm, Gm, warm=False, settings=so)
        kept = oldp.simulate(L, x0, T, Fm, Gm, warm=2, settings=so)
        mask = oldp.simulate(L, x0, T, Fm, Gm, warm=1, settings=so)
        assert (cold["flag_min"] >= 1).all() and (kept["flag_min"] >= 1).all() and (mask["flag_min"] >= 1).all()
        assert np.abs(kept["U"] - cold["U"]).max() < 1e-5 and np.abs(mask["U"] - cold["U"]).max() < 1e-5


# ------------------------------------------------------------------ variational objective (is_avi)
def test_game_theoretic_mpc_is_pinned_by_the_reference_test():
    """/root/reference/test/runtests.jl:1337-1358: two players on a double integrator; mpQP.H is not symmetric
    (:1349) and the closed loop from x0 = [10,10] with r = [10,0], N = 500, ends at y = [10, 0] (atol 1e-4, :1353-1354).
    The condensing restatement (mpc2mpqp.jl:900-950) + the AVI oracle reproduce both; the fixture holds the same."""
    from oracle import avi as oavi
    p = omm.game_kat()
    q = omm.mpc2mpqp(p)
    assert not q.is_symmetric and not np.allclose(q.H, q.H.T)
    assert (q.n, q.m, q.ms, q.nth) == (6, 6, 6, 6)                     # move blocks [1,1,8] x 2 inputs; theta = [x; r; uprev]
    assert omm.mpc2mpqp(omm.pendulum()).is_symmetric
    P = oavi.qp2avi(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=p.nu)
    sim = oavi.simulate(P, [[10.0, 10.0]], 500, p.F, p.G, r=[[10.0, 0.0]], uprev=[[0.0, 0.0]])
    y_end = sim["X"][499, 0]                                           # ys[:, end]: the state at the last step
    assert abs(y_end[0] - 10.0) < 1e-4 and abs(y_end[1] - 0.0) < 1e-4 and sim["flag_min"][0] == 1
    g = load_golden("game_kat")
    assert np.allclose(g["H"], q.H, rtol=0, atol=1e-12) and np.allclose(g["y_end"], y_end, atol=1e-12)
    # warm starts change the path, not the answer (unique solution)
    simw = oavi.simulate(P, [[10.0, 10.0]], 500, p.F, p.G, r=[[10.0, 0.0]], uprev=[[0.0, 0.0]], warm=True)
    assert np.abs(simw["X"] - sim["X"]).max() < 1e-9


def test_avi_oracle_against_the_golden_vectors_and_a_projection_iteration():
    """The AVI oracle reproduces the committed answers bit for bit from the committed pack; a sample of them is
    re-derived by a method that shares nothing with it: the fixed point of x <- clip(x - tau (Hx + f(theta)), bl, bu)."""
    from oracle import avi as oavi
    g = load_golden("game_kat")
    P = oavi.qp2avi(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=6)
    X, ef, it, act = oavi.solve_batch(P, g["theta"])
    assert np.array_equal(ef, g["exitflag"]) and np.array_equal(act, g["active"]) and np.array_equal(it, g["iters"])
    assert np.abs(X - g["X"]).max() <= 1e-12
    H = g["H"]
    tau = np.linalg.eigvalsh((H + H.T) / 2)[0] / np.linalg.norm(H, 2) ** 2
    nact = np.array([bin(int(a)).count("1") for a in act[:, 0]])
    for i in [int(np.nonzero(nact == k)[0][0]) for k in range(7)]:    # one point per active-set size 0 .. 6
        fth = g["f"] + g["f_theta"] @ g["theta"][i]
        x = np.zeros(6)
        for _ in range(300000):
            xn = np.clip(x - tau * (H @ x + fth), g["bl"], g["bu"])
            if np.abs(xn - x).max() < 1e-15:
                break
            x = xn
        assert np.abs(x - X[i]).max() < 1e-9, (i, nact[i])


def test_avi_oracle_kkt_and_infeasibility_on_random_problems():
    """Random non-symmetric problems with simple, general, one-sided and SOFT rows: every answer flagged solved
    satisfies the KKT conditions of the variational inequality (stationarity, primal feasibility of hard rows,
    multiplier signs; oracle.avi.kkt_residual -- dense algebra, nothing of the solver's recursions), every answer
    flagged infeasible has an empty hard constraint set (LP), and nothing ends on the iteration limit -- the plain
    QP loop cycles on such problems (no dual objective), the principal-pivoting step rule does not."""
    from scipy.optimize import linprog
    from oracle import avi as oavi
    rng = np.random.default_rng(11)
    S_ = oldp.default_settings()
    S_.rho_soft = 1e-2                       # (keeps the certificate's slack / rho step well conditioned)
    nsolved = ninf = 0
    for trial in range(60):
        n = int(rng.integers(2, 9)); mg = int(rng.integers(0, 12)); ms = int(rng.integers(0, n + 1)); nth = int(rng.integers(1, 5))
        if ms + mg == 0:
            ms = 1
        B = rng.normal(size=(n, n)); K = rng.normal(size=(n, n)) * rng.uniform(0, 2)
        H = B @ B.T + 0.3 * np.eye(n) + (K - K.T)
        f, fth = rng.normal(size=n), rng.normal(size=(n, nth))
        A = rng.normal(size=(mg, n)); m = ms + mg
        bu, bl = rng.uniform(0.1, 2, m), -rng.uniform(0.1, 2, m)
        bl[rng.random(m) < 0.2] = -1e30
        W = rng.normal(size=(m, nth)) * 0.3
        sense = np.zeros(m, np.int32); sense[ms:][rng.random(mg) < 0.3] = 8
        P = oavi.qp2avi(H, f, fth, A, bu, bl, W, sense, nout=n)
        th = rng.normal(size=(32, nth)) * rng.uniform(0.5, 4)
        X, ef, it, act = oavi.solve_batch(P, th, S_)
        assert set(np.unique(ef)) <= {1, 2, -1}, np.unique(ef)
        assert it.max() < 200
        Aext = np.vstack([np.eye(n)[:ms], A]); hard = (sense & 8) == 0
        sc = max(1.0, np.abs(H).max())
        for i in range(32):
            if ef[i] >= 1:
                st, pv, sg = oavi.kkt_residual(H, f, fth, A, bu, bl, W, sense, th[i], X[i], act[i], rho_soft=1e-2)
                assert st < 1e-7 * sc and pv < 2e-6 and sg < 1e-6 * sc, (trial, i, st, pv, sg)
                nsolved += 1
            elif i % 4 == 0:
                b = W @ th[i]
                r = linprog(np.zeros(n), A_ub=np.vstack([Aext[hard], -Aext[hard]]),
                            b_ub=np.concatenate([(bu + b)[hard], -(bl + b)[hard]]), bounds=[(None, None)] * n, method="highs")
                assert r.status == 2, (trial, i, r.status)
                ninf += 1
    assert nsolved > 1000 and ninf > 20


def test_proximal_point_iterations_reach_a_kkt_point_of_the_semidefinite_problem():
    """DAQP's eps_prox mode (a symmetric positive SEMIdefinite H; without it setup! answers -5,
    /root/reference/src/setup.jl:18-19) as the oracle states it: x_{k+1} = argmin 1/2 x'Hx + f'x + eps/2 |x - x_k|^2
    over the rows, until |x_{k+1} - x_k| < eta.  Rank-deficient H, every variable bounded (so the QP is bounded),
    general rows: the limit must satisfy the KKT conditions of the ORIGINAL problem -- stationarity with multipliers of
    the right sign recovered by least squares on the final active set, primal feasibility -- which certifies optimality
    of a convex QP without reference to any solver; on top, no feasible point near it has a smaller objective.
    (A strictly convex problem with a vanishing regularisation is no usable reference: the Cholesky factor of
    H + 1e-7 I is so ill-conditioned that the QP oracle's normalised-row tolerance lets its answers violate the rows
    by 1e-2.)"""
    from oracle import avi as oavi
    rng = np.random.default_rng(3)
    eps, eta = 1e-4, 1e-9
    nsolved = 0
    for trial in range(40):
        n = int(rng.integers(3, 9)); r = int(rng.integers(1, n)); mg = int(rng.integers(0, 8)); nth = int(rng.integers(1, 4))
        B = rng.normal(size=(n, r)); H = B @ B.T                      # rank r < n
        f, fth = rng.normal(size=n), rng.normal(size=(n, nth))
        A = rng.normal(size=(mg, n)); m = n + mg
        bu, bl = rng.uniform(0.5, 2, m), -rng.uniform(0.5, 2, m)
        W = rng.normal(size=(m, nth)) * 0.2; W[:n] = 0
        sense = np.zeros(m, np.int32)
        P, px = oavi.qp2prox(H, f, fth, A, bu, bl, W, sense, nout=n, eps=eps)
        th = rng.normal(size=(16, nth))
        X, ef, it, act = oavi.prox_solve_batch(P, px, th, eps, eta)
        assert np.all(ef == 1) and it.max() < 2000
        Aext = np.vstack([np.eye(n), A])
        for i in range(16):
            g = H @ X[i] + f + fth @ th[i]
            up = np.array([(int(act[i][j >> 6]) >> (j & 63)) & 1 for j in range(m)], bool)
            lo = np.array([(int(act[i][(m + j) >> 6]) >> ((m + j) & 63)) & 1 for j in range(m)], bool)
            rows = np.nonzero(up | lo)[0]
            if len(rows):
                mu = np.linalg.lstsq(Aext[rows].T, -g, rcond=None)[0]
                res = np.abs(g + Aext[rows].T @ mu).max()
                assert np.max(np.where(up[rows], -mu, mu)) < 1e-7, (trial, i)
            else:
                res = np.abs(g).max()
            ax = Aext @ X[i]
            assert res < 1e-7 and np.maximum(ax - (bu + W @ th[i]), (bl + W @ th[i]) - ax).max() < 2e-6, (trial, i, res)
            obj = lambda z: 0.5 * z @ H @ z + (f + fth @ th[i]) @ z
            for _ in range(8):
                z = X[i] + 0.05 * rng.normal(size=n)
                az = Aext @ z
                if np.maximum(az - (bu + W @ th[i]), (bl + W @ th[i]) - az).max() <= 0:
                    assert obj(z) >= obj(X[i]) - 1e-9, (trial, i)
            nsolved += 1
    assert nsolved == 640
