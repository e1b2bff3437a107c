"""GPU parity tests of the wavefront kernel's Gram-scan form (`lmpc_set_option("gram_scan", 1)`).

The Gram-scan form takes the same decisions as the n-chain form but sums three quantities in a different order
(row values from Gram columns, the dual objective from the factorisation, pairwise lane trees for the dot
products of a row append), so its checker is the oracle's mode 1 (oracle/daqp_ldp_oracle.c "Gram-scan form"):
bit-identical exit flags, iteration counts, active sets and x.  Against the n-chain form (mode 0, what libdaqp
does per iteration: /root/reference/codegen/mpc_update_qp.c:48) the two agree on every solvable point in flag,
iteration count and active set and to <= 1e-10 in x on hard-constrained problems; with SOFT rows the 1/rho_soft
penalty amplifies the rounding of either form, the bound asserted there is 1e-4.
"""
import numpy as np
import pytest

from conftest import load_golden, oracle_ldp_from
from test_gpu_parity import _qp_from_golden, _random_qp, _copy_settings, lmpc  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _gram_settings(rho=None, f32=False):
    from oracle import ldp as oldp
    s = oldp.default_settings_f32() if f32 else oldp.default_settings()
    s.mode = 1
    if rho is not None:
        s.rho_soft = rho
    return s


def _compare_gram(qp, theta, warm=None, settings=None, f32=False):
    """gram_scan on: the GPU against the oracle's mode 1, bit for bit."""
    from oracle import ldp as oldp
    L = oracle_ldp_from(qp.ldp())
    qp.set_option("gram_scan", 1)
    s = settings if settings is not None else _gram_settings(f32=f32)
    assert s.mode == 1
    if f32:
        theta = np.asarray(theta, np.float32)
        x, ef, it, act = qp.solve_f32(theta, warm=warm)
        xo, efo, ito, acto = oldp.solve_batch(L, theta, s, warm=warm, dtype=np.float32)
    else:
        x, ef, it, act = qp.solve(theta, warm=warm)
        xo, efo, ito, acto = oldp.solve_batch(L, theta, s, warm=warm)
    assert np.array_equal(ef, efo), np.flatnonzero(ef != efo)[:10]
    assert np.array_equal(it, ito), np.flatnonzero(it != ito)[:10]
    assert np.array_equal(act, acto)
    assert np.array_equal(x, xo), np.abs(x - xo).max()
    return x, ef, it, act


def _against_chain_form(qp, theta, x, ef, it, act, xtol):
    """... and against the n-chain form on the same handle: same answers wherever a solution exists."""
    qp.set_option("gram_scan", 0)
    x0, ef0, it0, act0 = qp.solve(theta)
    qp.set_option("gram_scan", 1)
    ok = ef0 >= 1
    assert np.array_equal(ef >= 1, ok)
    assert np.array_equal(ef[ok], ef0[ok]) and np.array_equal(it[ok], it0[ok]) and np.array_equal(act[ok], act0[ok])
    if ok.any():
        assert np.abs(x[ok] - x0[ok]).max() <= xtol, np.abs(x[ok] - x0[ok]).max()


@pytest.mark.parametrize("name", ["pendulum", "mass_spring", "mass_spring_3in", "preprocessing_kat", "x0unc_kat"])
def test_gram_form_hard_rows(lmpc, name):
    g = load_golden(name)
    qp = _qp_from_golden(lmpc, g)
    qp.set_option("wave", 1)
    assert qp.kernel_name.endswith("wave")
    x, ef, it, act = _compare_gram(qp, g["theta"])
    _against_chain_form(qp, g["theta"], x, ef, it, act, 1e-6 if name == "x0unc_kat" else TOL)
    ok = ef >= 1
    _compare_gram(qp, g["theta"][ok], warm=act[ok])


@pytest.mark.parametrize("name,rho", [("soft_doc", None), ("soft_doc", 1e-3)])
def test_gram_form_soft_rows(lmpc, name, rho):
    g = load_golden(name)
    s = lmpc.default_settings()
    if rho is not None:
        s.rho_soft = rho
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], settings=s)
    x, ef, it, act = _compare_gram(qp, g["theta"], settings=_gram_settings(rho))
    assert (ef == 2).any() and (ef == 1).any()
    _against_chain_form(qp, g["theta"], x, ef, it, act, 1e-4)


@pytest.mark.parametrize("packed,level,nwv", [(0, 0, 2), (0, 1, 8), (1, 0, 8), (1, 1, 2)])
def test_gram_form_layouts_and_staging_levels_agree(lmpc, packed, level, nwv):
    for name in ("mass_spring_3in", "soft_doc", "satellite20"):
        g = load_golden(name)
        qp = _qp_from_golden(lmpc, g)
        qp.set_option("gram_scan", 1)
        ref = qp.solve(g["theta"][:300])
        qp.set_option("wave_packed", packed)
        qp.set_option("wave_level", level)
        qp.set_option("wave_nwv", nwv)
        out = qp.solve(g["theta"][:300])
        for a, b in zip(ref, out):
            assert np.array_equal(a, b), name


@pytest.mark.parametrize("N", [50, 75, 100, 125])
def test_gram_form_reference_benchmark_class(lmpc, N):
    """docs/src/manual/benchmark.md:4-16: n = N, 3N - 2 rows, 2N - 2 of them soft -- where the Gram-scan form pays:
    |W| Gram columns per iteration instead of n columns of M'."""
    g = load_golden(f"pendulum_N{N}")
    theta = g["theta"]
    qp = _qp_from_golden(lmpc, g, 1)
    assert qp.kernel_name.endswith("wave")
    for scr in (1, 0):
        qp.set_option("screen_wave", scr)
        x, ef, it, act = _compare_gram(qp, theta)
    assert np.all(ef >= 1)
    _against_chain_form(qp, theta, x, ef, it, act, 1e-4)
    qt = _qp_from_golden(lmpc, g)                                 # whole trajectory + warm start
    xt, eft, itt, actt = _compare_gram(qt, theta[:200])
    assert np.array_equal(xt[:, 0], x[:200, 0])
    _compare_gram(qt, theta[:200], warm=actt)
    q32 = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1,
                                   settings=lmpc.default_settings_f32())
    _compare_gram(q32, theta[:256], f32=True)


@pytest.mark.parametrize("n,mg,nth,nsoft,seed", [(20, 30, 6, 0, 0), (30, 90, 8, 10, 1), (63, 100, 5, 0, 2),
                                                 (12, 200, 4, 40, 3), (3, 5, 2, 2, 4),
                                                 (8, 300, 3, 0, 11), (20, 480, 5, 60, 12), (40, 984, 4, 100, 13),
                                                 (70, 100, 5, 0, 16), (30, 270, 6, 30, 17), (100, 273, 7, 0, 20),
                                                 (127, 60, 4, 0, 21), (90, 200, 5, 20, 22)])
def test_gram_form_random_problems(lmpc, n, mg, nth, nsoft, seed):
    rng = np.random.default_rng(seed)
    H, f, f_theta, A, bu, bl, W, sense = _random_qp(rng, n, mg, nth, nsoft)
    big = mg > 128
    qp = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, (2 if big else 1) * bu, (2 if big else 1) * bl,
                                  (0.3 if big else 1.0) * W, sense, nout=min(n, 4))
    assert qp.kernel_name.endswith("wave")
    theta = rng.uniform(-2, 2, (400, nth))
    x, ef, it, act = _compare_gram(qp, theta)
    assert (ef >= 1).mean() > 0.05
    _against_chain_form(qp, theta, x, ef, it, act, 1e-4 if nsoft else 1e-9)
    ok = ef >= 1
    _compare_gram(qp, theta[ok][:65], warm=act[ok][:65])
    if nsoft == 0:
        s32 = lmpc.default_settings_f32()
        qf = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, (2 if big else 1) * bu, (2 if big else 1) * bl,
                                      (0.3 if big else 1.0) * W, sense, nout=min(n, 4), settings=s32)
        so = _copy_settings(lmpc, s32)
        so.mode = 1
        _compare_gram(qf, theta, settings=so, f32=True)


@pytest.mark.parametrize("name", ["satellite4", "satellite20", "satellite20_preview"])
def test_gram_form_hybrid_branch_and_bound(lmpc, name):
    g = load_golden(name)
    qp = _qp_from_golden(lmpc, g)
    x, ef, it, act = _compare_gram(qp, g["theta"])
    assert np.all(ef == 1)
    bins = np.flatnonzero(g["senses"] & 16)
    assert np.all(np.minimum(np.abs(x[:, bins] - g["bu"][bins]), np.abs(x[:, bins] - g["bl"][bins])) < 1e-9)
    assert np.abs(x - g["X"]).max() <= 1e-8
    q32 = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"],
                                   settings=lmpc.default_settings_f32())
    _compare_gram(q32, g["theta"], f32=True)


def test_gram_form_working_sets_beyond_the_64_lanes(lmpc):
    # a point whose working set outgrows the 64 lanes is re-solved by the one-problem-per-thread kernel, which is
    # the n-chain form; the oracle's mode 1 does the same (mode 0 from scratch)
    rng = np.random.default_rng(5)
    n, mg, nth = 6, 150, 2
    H, f, f_theta, A, bu, bl, W, sense = _random_qp(rng, n, mg, nth, nsoft=mg)
    W[n:, 0] = np.abs(W[n:, 0]) + 0.5
    qp = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, bu, bl, W, sense, nout=2)
    theta = np.vstack([rng.uniform(-1, 1, (200, nth)),
                       np.hstack([rng.uniform(30, 60, (56, 1)), rng.uniform(-1, 1, (56, 1))])])
    x, ef, it, act = _compare_gram(qp, theta)
    nact = np.array([sum(bin(int(w)).count("1") for w in row) for row in act])
    assert (nact > 64).sum() >= 20 and (ef >= 1).all()


def test_gram_form_closed_loop(lmpc):
    from oracle import ldp as oldp
    from oracle import mpc2mpqp as omm
    prob = omm.doc_simple_soft()
    g = load_golden("soft_doc")
    qp = _qp_from_golden(lmpc, g, 1)
    qp.set_option("gram_scan", 1)
    L = oracle_ldp_from(qp.ldp())
    rng = np.random.default_rng(9)
    N, T = 400, 10
    x0 = rng.uniform(0, 0.5, (N, 2)); x0[0] = 0.0
    r = np.tile([1.0, 0.0], (N, 1))
    for warm in (False, True):
        # (the wavefront path's warm loop keeps every scenario's factorisation: the oracle's warm == 2)
        ref = oldp.simulate(L, x0, T, prob.F, prob.G, r=r, warm=2 if warm else False, settings=_gram_settings())
        out = qp.simulate(x0, T, prob.F, prob.G, r=r, warm=warm)
        assert np.array_equal(out["flag_min"], ref["flag_min"])
        assert np.array_equal(out["U"], ref["U"]) and np.array_equal(out["X"], ref["X"])


@pytest.mark.parametrize("name,gram", [("pendulum_N50", 0), ("pendulum_N50", 1), ("soft_doc", 0), ("soft_doc", 1)])
def test_wave_path_closed_loop_asynchronous_rounds(lmpc, name, gram):
    """Closed loop on the wavefront-kernel path (soft rows): scenario-asynchronous rounds (round 3) against the
    lock-step loop and the oracle's closed loop, bit for bit -- trajectories, final states, smallest flags; warm and
    cold; in both forms of the kernel (the oracle's mode follows the form)."""
    from oracle import ldp as oldp
    from oracle import mpc2mpqp as omm
    g = load_golden(name)
    qp = _qp_from_golden(lmpc, g, 1)
    qp.set_option("gram_scan", gram)
    L = oracle_ldp_from(qp.ldp())
    rng = np.random.default_rng(17)
    if name == "soft_doc":
        prob = omm.doc_simple_soft()
        F, G = prob.F, prob.G
        N, T = 3000, 25
        x0 = rng.uniform(-0.2, 0.7, (N, 2)); x0[0] = 0.0
        r = np.tile([1.0, 0.0], (N, 1)); r[N // 2:] = [0.3, 0.0]
    else:
        F, G = g["F"], g["G"]
        N, T = 4000, 40
        base = g["theta"][:int(g["n_closed_loop"])]
        pick = base[rng.integers(0, len(base), N)] + rng.normal(size=(N, 7)) * [0.02, 0.05, 0.005, 0.05, 0.02, 0.0, 0.0]
        x0, r = pick[:, :4].copy(), pick[:, 4:6].copy()
    so = _gram_settings() if gram else oldp.default_settings()
    # one step per visit of the wavefront kernel and the mask-based warm start: the oracle's warm = 1
    qp.set_option("sim_keep_factor", 0)
    qp.set_option("sim_run_ahead", 0)
    for warm in (True, False):
        ref = oldp.simulate(L, x0, T, F, G, r=r, warm=warm, settings=so)
        outs = []
        for asyn in (2, 0):                        # 2: asynchronous rounds on the wavefront-kernel path too
            qp.set_option("sim_async", asyn)
            outs.append(qp.simulate(x0, T, F, G, r=r, warm=warm))
        for out in outs:
            assert np.array_equal(out["flag_min"], ref["flag_min"])
            assert np.array_equal(out["U"], ref["U"]) and np.array_equal(out["X"], ref["X"])
            assert np.array_equal(out["x"], ref["x"])
        assert (ref["flag_min"] >= 1).mean() > 0.5
    # run-ahead (default): a scenario's consecutive steps that need iterations stay inside the wavefront kernel, warm on
    # the factor as it stands -- the kept factorisation of the step-synchronous loop without the trip through memory:
    # the oracle's warm = 2; cold, every step starts from nothing either way
    qp.set_option("sim_keep_factor", 1)
    qp.set_option("sim_run_ahead", 1)
    for warm, owarm in ((True, 2), (False, False)):
        ref = oldp.simulate(L, x0, T, F, G, r=r, warm=owarm, settings=so)
        for asyn in (2, 0):
            qp.set_option("sim_async", asyn)
            out = qp.simulate(x0, T, F, G, r=r, warm=warm)
            for key in ("U", "X", "x", "flag_min"):
                assert np.array_equal(out[key], ref[key]), (warm, asyn, key)


@pytest.mark.parametrize("name,gram", [("pendulum_N50", 0), ("pendulum_N50", 1), ("soft_doc", 0), ("soft_doc", 1)])
def test_wave_path_closed_loop_keeps_the_factorisation(lmpc, name, gram):
    """The wavefront path's warm closed loop (default): every scenario's final working set stays on the device with
    its L and D, in its order, and the next step continues from it (what DAQP_WARMSTART means for libdaqp's workspace,
    /root/reference/codegen/mpc_update_qp.c:44-54).  Bit for bit the oracle's warm == 2 in both forms; the same
    optimum as the mask-based warm start and as the cold loop (K6's 1e-9 criterion, test/runtests.jl:85-117); fewer
    iterations than the mask-based start never hurts correctness, so only the results are compared."""
    from oracle import ldp as oldp
    from oracle import mpc2mpqp as omm
    g = load_golden(name)
    qp = _qp_from_golden(lmpc, g, 1)
    qp.set_option("gram_scan", gram)
    L = oracle_ldp_from(qp.ldp())
    rng = np.random.default_rng(23)
    if name == "soft_doc":
        prob = omm.doc_simple_soft()
        F, G = prob.F, prob.G
        N, T = 3000, 25
        x0 = rng.uniform(-0.2, 0.7, (N, 2)); x0[0] = 0.0
        r = np.tile([1.0, 0.0], (N, 1)); r[N // 2:] = [0.3, 0.0]
    else:
        F, G = g["F"], g["G"]
        N, T = 4000, 40
        base = g["theta"][:int(g["n_closed_loop"])]
        pick = base[rng.integers(0, len(base), N)] + rng.normal(size=(N, 7)) * [0.02, 0.05, 0.005, 0.05, 0.02, 0.0, 0.0]
        x0, r = pick[:, :4].copy(), pick[:, 4:6].copy()
    so = _gram_settings() if gram else oldp.default_settings()
    ref = oldp.simulate(L, x0, T, F, G, r=r, warm=2, settings=so)
    out = qp.simulate(x0, T, F, G, r=r, warm=True)
    again = qp.simulate(x0, T, F, G, r=r, warm=True)       # (the kept states of the first run are not reused)
    for o in (out, again):
        assert np.array_equal(o["flag_min"], ref["flag_min"])
        assert np.array_equal(o["U"], ref["U"]) and np.array_equal(o["X"], ref["X"]) and np.array_equal(o["x"], ref["x"])
    assert (ref["flag_min"] >= 1).mean() > 0.5
    qp.set_option("sim_keep_factor", 0)
    mask = qp.simulate(x0, T, F, G, r=r, warm=True)
    cold = qp.simulate(x0, T, F, G, r=r, warm=False)
    ok = (ref["flag_min"] >= 1) & (mask["flag_min"] >= 1) & (cold["flag_min"] >= 1)
    assert ok.mean() > 0.5
    # the same optimum up to the tolerances: a soft row inside the primal_tol band (1e-6) may end on either side of it,
    # depending on where the iterations started (4e-7 in u among 4000 x 40 steps of the N = 50 problem)
    assert np.abs(out["U"][:, ok] - mask["U"][:, ok]).max() < 1e-5 and np.abs(out["U"][:, ok] - cold["U"][:, ok]).max() < 1e-5


@pytest.mark.parametrize("gram", [0, 1])
def test_wave_path_closed_loop_random_shapes_keep_and_mask_start(lmpc, gram):
    """Random controllers on the wavefront path -- general rows, soft rows, one to three inputs, with and without a
    previous-control block, up to three constraint slots per lane, scenario counts that are no multiple of a
    wavefront -- closed loop against the oracle bit for bit: the kept factorisation (oracle warm = 2), the mask start
    (`sim_keep_factor` 0, oracle warm = 1), the cold loop; fused plant step and the unfused loop agree bit for bit."""
    from oracle import ldp as oldp
    from test_gpu_parity import _random_qp
    rng = np.random.default_rng(1234 + gram)
    shapes = [(2, 1, 0, 1, 6, 20, 4), (4, 1, 2, 1, 8, 60, 0), (3, 2, 1, 2, 10, 30, 6), (6, 2, 1, 0, 12, 130, 10),
              (5, 3, 0, 3, 9, 45, 3), (8, 1, 2, 1, 20, 70, 12), (4, 1, 0, 0, 30, 100, 0)]
    for nx, nu, nr, nup, n, mg, nsoft in shapes:
        nth = nx + nr + nup
        H, f, f_theta, A, bu, bl, W, sense = _random_qp(rng, n, mg, nth, nsoft=nsoft)
        f_theta *= 0.6
        qp = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, bu, bl, W, sense, nout=nu)
        qp.set_option("wave", 1)
        assert qp.kernel_name.endswith("wave")
        qp.set_option("gram_scan", gram)
        Fm = rng.standard_normal((nx, nx))
        Fm *= 0.9 / np.abs(np.linalg.eigvals(Fm)).max()
        Gm = 0.5 * rng.standard_normal((nx, nu))
        N, T = 333, 20
        x0 = rng.uniform(-2, 2, (N, nx))
        r = rng.uniform(-1, 1, (N, nr)) if nr else None
        L = oracle_ldp_from(qp.ldp())
        so = _gram_settings() if gram else oldp.default_settings()
        for keep, owarm, warm in ((1, 2, True), (0, 1, True), (1, False, False)):
            qp.set_option("sim_keep_factor", keep)
            ref = oldp.simulate(L, x0, T, Fm, Gm, r=r, warm=owarm, settings=so)
            outs = []
            for fused in (1, 0):
                qp.set_option("sim_fused", fused)
                outs.append(qp.simulate(x0, T, Fm, Gm, r=r, warm=warm))
            qp.set_option("sim_fused", 1)
            for out in outs:
                for key in ("U", "X", "x", "flag_min"):
                    assert np.array_equal(out[key], ref[key]), (nx, nu, nr, nup, n, mg, nsoft, keep, warm, key)
            assert (outs[0]["U"] != 0).any()


@pytest.mark.parametrize("gram", [0, 1])
def test_wave_path_closed_loop_kept_factor_through_the_64_row_limit(lmpc, gram):
    """Closed loop whose working sets cross the wavefront kernel's 64 rows: scenarios start with 60..150 soft rows active
    and decay into the small-working-set regime.  A step that wants more than 64 rows is re-solved by the slow path
    from the mask (n-chain arithmetic) and keeps nothing; the step after it starts from that mask; later steps keep
    their factorisation again.  The oracle's warm = 2 takes the same decisions (`solve_one_keep`)."""
    from oracle import ldp as oldp
    from test_gpu_parity import _random_qp
    rng = np.random.default_rng(5)
    n, mg, nth = 6, 150, 2
    H, f, f_theta, A, bu, bl, W, sense = _random_qp(rng, n, mg, nth, nsoft=mg)
    W[n:, 0] = np.abs(W[n:, 0]) + 0.5                    # theta_0 >> 0 pushes every soft row over its bound
    qp = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, bu, bl, W, sense, nout=2)
    qp.set_option("gram_scan", gram)
    assert qp.kernel_name.endswith("wave")
    N, T = 150, 30
    x0 = np.vstack([rng.uniform(-1, 1, (60, 2)), np.hstack([rng.uniform(20, 60, (N - 60, 1)), rng.uniform(-1, 1, (N - 60, 1))])])
    Fm = np.array([[0.8, 0.05], [0.0, 0.7]])
    Gm = 0.05 * rng.standard_normal((2, 2))
    L = oracle_ldp_from(qp.ldp())
    so = _gram_settings() if gram else oldp.default_settings()
    ref = oldp.simulate(L, x0, T, Fm, Gm, warm=2, settings=so)
    # the run really crosses the limit, in both directions
    th = np.hstack([x0])
    _, _, _, act0 = oldp.solve_batch(L, th, so)
    nact0 = np.array([sum(bin(int(w)).count("1") for w in row) for row in act0])
    assert (nact0 > 64).sum() >= 30 and (nact0 <= 64).sum() >= 30
    xT = ref["x"]
    _, _, _, actT = oldp.solve_batch(L, xT, so)
    assert max(sum(bin(int(w)).count("1") for w in row) for row in actT) <= 64
    for fused in (1, 0):
        qp.set_option("sim_fused", fused)
        out = qp.simulate(x0, T, Fm, Gm, warm=True)
        for key in ("U", "X", "x", "flag_min"):
            assert np.array_equal(out[key], ref[key]), (fused, key)
    assert (ref["flag_min"] >= 1).all()
    # the mask start through the same run (until round 3 the wavefront kernel overwrote the mask of a point it handed to
    # the slow path with its cut-off working set -- in a closed loop that mask is what the slow path starts from)
    qp.set_option("sim_keep_factor", 0)
    ref1 = oldp.simulate(L, x0, T, Fm, Gm, warm=1, settings=so)
    out1 = qp.simulate(x0, T, Fm, Gm, warm=True)
    for key in ("U", "X", "x", "flag_min"):
        assert np.array_equal(out1[key], ref1[key]), key


@pytest.mark.parametrize("gram", [0, 1])
def test_two_passes_of_the_wavefront_kernel(lmpc, gram):
    """A first pass at a smaller working-set capacity (more wavefronts resident, M' staged again), a second one at the
    full capacity for the points that outgrew it, the slow path behind that: decided by the handle's own statistics
    (`lmpc_wave_stats`), and never visible in a result.  (a) the N = 50 benchmark problem: after a few calls the handle
    runs its batches at 24 rows first; every call equals the single-pass result and the oracle bit for bit, cold and
    warm.  (b) a problem whose working sets spread from 0 to 150 rows, first pass forced: points finished by the first
    pass, by the second, and by the slow path in one batch, all as the oracle has them."""
    from oracle import ldp as oldp
    from test_gpu_parity import _random_qp
    so = _gram_settings() if gram else oldp.default_settings()
    g = load_golden("pendulum_N50")
    qp = _qp_from_golden(lmpc, g, 1)
    qp.set_option("gram_scan", gram)
    rng = np.random.default_rng(3)
    theta = np.tile(g["theta"], (40, 1))[:8192] + rng.normal(size=(8192, 7)) * [0.02, 0.05, 0.005, 0.05, 0.02, 0.0, 0.0]
    L = oracle_ldp_from(qp.ldp())
    xo, efo, ito, acto = oldp.solve_batch(L, theta, so)
    qp.set_option("wave_two_pass", 0)
    x1, ef1, it1, act1 = qp.solve(theta)
    assert qp.wave_stats()["first_pass_rows"] == 0
    qp.set_option("wave_two_pass", -1)
    outs = [qp.solve(theta) for _ in range(4)]
    st = qp.wave_stats()
    assert st["problems"] > 1000 and st["within_24"] >= 0.97 * st["problems"] and st["first_pass_rows"] == 24
    for x, ef, it, act in outs:
        assert np.array_equal(x, x1) and np.array_equal(ef, ef1) and np.array_equal(it, it1) and np.array_equal(act, act1)
    assert np.array_equal(ef1, efo) and np.array_equal(it1, ito) and np.array_equal(act1, acto) and np.abs(x1 - xo).max() <= 1e-10
    ok = efo >= 1
    xw, efw, itw, actw = qp.solve(theta[ok], warm=acto[ok])
    xq, efq, itq, actq = oldp.solve_batch(L, theta[ok], so, warm=acto[ok])
    assert np.array_equal(efw, efq) and np.array_equal(itw, itq) and np.array_equal(actw, actq) and np.abs(xw - xq).max() <= 1e-10
    # (b)
    rng = np.random.default_rng(5)
    n, mg, nth = 6, 150, 2
    H, f, f_theta, A, bu, bl, W, sense = _random_qp(rng, n, mg, nth, nsoft=mg)
    W[n:, 0] = np.abs(W[n:, 0]) + 0.5
    qb = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, bu, bl, W, sense, nout=2)
    qb.set_option("gram_scan", gram)
    qb.set_option("wave_two_pass", 1)                     # forced, at the default 24 rows
    thb = np.vstack([rng.uniform(-1, 1, (3000, nth)), np.hstack([rng.uniform(0, 2.5, (6000, 1)), rng.uniform(-1, 1, (6000, 1))]),
                     np.hstack([rng.uniform(5, 60, (1000, 1)), rng.uniform(-1, 1, (1000, 1))])])
    Lb = oracle_ldp_from(qb.ldp())
    xo, efo, ito, acto = oldp.solve_batch(Lb, thb, so)
    nact = np.array([sum(bin(int(w)).count("1") for w in row) for row in acto])
    assert (nact <= 20).sum() > 500 and ((nact > 24) & (nact <= 60)).sum() > 500 and (nact > 64).sum() > 500
    for _ in range(2):
        x, ef, it, act = qb.solve(thb)
        assert np.array_equal(ef, efo) and np.array_equal(it, ito) and np.array_equal(act, acto)
        assert np.abs(x - xo).max() <= 1e-10
    assert qb.wave_stats()["first_pass_rows"] == 24 and qb.wave_stats()["within_48"] < qb.wave_stats()["problems"]
    # calls in one pass and in two, alternating on the same handle (the counters of the first pass's overflow list are
    # cleared by the call before: a one-pass call has to do that too -- tools/fuzz_closed_loop.py found it did not)
    for tp in (0, 1, 0, 0, 1, 1, 0, 1):
        qb.set_option("wave_two_pass", tp)
        x, ef, it, act = qb.solve(thb)
        assert np.array_equal(ef, efo) and np.array_equal(it, ito) and np.array_equal(act, acto), tp
        assert np.abs(x - xo).max() <= 1e-10


def test_closed_loop_run_ahead_is_exact_after_the_handle_has_statistics(lmpc):
    """A handle that has already solved batches (its statistics would send a plain batch through a first pass at a small
    capacity) runs its closed loops with run-ahead in ONE pass: a step that outgrew a first pass would have to restart
    from the factor of the step before, which run-ahead keeps nowhere.  Working sets here pass 24 and 32 rows during
    the transient; trajectories equal the checker's bit for bit on the first and on later calls."""
    from oracle import ldp as oldp
    from test_gpu_parity import _random_qp
    rng = np.random.default_rng(77)
    nx, nu, n, mg, nsoft = 4, 1, 40, 90, 90
    H, f, f_theta, A, bu, bl, W, sense = _random_qp(rng, n, mg, nx, nsoft=nsoft)
    W[n:, 0] = np.abs(W[n:, 0])
    f_theta *= 0.6
    qp = lmpc.BatchedQP.from_mpqp(H, f, f_theta, A, bu, bl, W, sense, nout=nu)
    qp.set_option("gram_scan", 1)
    assert qp.kernel_name.endswith("wave")
    Fm = np.diag([0.9, 0.8, 0.7, 0.6]) + 0.02 * rng.standard_normal((4, 4))
    Gm = 0.3 * rng.standard_normal((4, 1))
    N, T = 6000, 15
    x0 = rng.uniform(-1, 1, (N, nx)); x0[:, 0] = rng.uniform(0, 3, N)
    L = oracle_ldp_from(qp.ldp())
    so = _gram_settings()
    # batches first: statistics with most working sets small
    th = rng.uniform(-0.3, 0.3, (8192, nx))
    for _ in range(4):
        qp.solve(th)
    ref = oldp.simulate(L, x0, T, Fm, Gm, warm=2, settings=so)
    _, _, _, act0 = oldp.solve_batch(L, x0, so)
    nact0 = np.array([sum(bin(int(w)).count("1") for w in row) for row in act0])
    assert (nact0 > 32).sum() > 200 and (nact0 <= 24).sum() > 200
    for _ in range(2):
        out = qp.simulate(x0, T, Fm, Gm, warm=True)
        for key in ("U", "X", "x", "flag_min"):
            assert np.array_equal(out[key], ref[key]), key
    assert (ref["flag_min"] >= 1).mean() > 0.9
    # the step-synchronous loop DOES run in two passes (forced here, 24 rows first): a point that outgrows the first
    # pass leaves its kept state alone, a kept working set larger than the first pass's capacity goes straight on,
    # and the second pass restores it -- same trajectories
    qp.set_option("sim_async", 0)
    qp.set_option("wave_two_pass", 1)
    for fused in (1, 0):
        qp.set_option("sim_fused", fused)
        out = qp.simulate(x0, T, Fm, Gm, warm=True)
        for key in ("U", "X", "x", "flag_min"):
            assert np.array_equal(out[key], ref[key]), (fused, key)
    assert qp.wave_stats()["first_pass_rows"] == 24


@pytest.mark.parametrize("gram", [0, 1])
def test_wave_path_closed_loop_edge_sizes(lmpc, gram):
    """One scenario, one step, fewer scenarios than a wavefront has lanes, a horizon of two: the wavefront path's default
    closed loop (rounds with run-ahead) and the step-synchronous one against the checker's warm = 2, bit for bit."""
    from oracle import ldp as oldp
    g = load_golden("pendulum_N50")
    qp = _qp_from_golden(lmpc, g, 1)
    qp.set_option("gram_scan", gram)
    L = oracle_ldp_from(qp.ldp())
    so = _gram_settings() if gram else oldp.default_settings()
    base = g["theta"][:int(g["n_closed_loop"])]
    for N, T in ((1, 1), (1, 40), (3, 2), (65, 2), (130, 7)):
        pick = base[(np.arange(N) * 7) % len(base)]
        x0, r = pick[:, :4].copy(), pick[:, 4:6].copy()
        ref = oldp.simulate(L, x0, T, g["F"], g["G"], r=r, warm=2, settings=so)
        for asyn in (1, 0):
            qp.set_option("sim_async", asyn)
            out = qp.simulate(x0, T, g["F"], g["G"], r=r, warm=True)
            for key in ("U", "X", "x", "uprev", "flag_min"):
                assert np.array_equal(out[key], ref[key]), (N, T, asyn, key)
