"""CPU tests of the host logic and of the C-ABI surface (no compute calls: no GPU here)."""
import ctypes
import re
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import ldp as oldp


@pytest.fixture(scope="module")
def lmpc():
    import linearmpc_jl_amd as mod
    return mod


def test_library_exports_every_declared_symbol(lmpc):
    hdr = open(os.path.join(ROOT, "include", "lmpc_hip.h")).read()
    declared = set(re.findall(r"\b(lmpc_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"lmpc_handle", "lmpc_settings"}
    assert declared == set(lmpc.SYMBOLS), declared ^ set(lmpc.SYMBOLS)
    L = lmpc.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.lmpc_abi_version() == 2          # 2: lmpc_settings grew eps_prox / eta_prox (round 4)


def test_public_header_is_self_contained_c(tmp_path):
    # the boundary is a C ABI: the header must compile on its own as plain C (a maintainer's cgo / ccall / ctypes
    # generator reads nothing else) -- it used size_t without <stddef.h> until round 3
    import shutil, subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "hdr.c"
    src.write_text('#include "lmpc_hip.h"\nint main(void) { return 0; }\n')
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_default_settings_match_reference_docs(lmpc):
    # /root/reference/docs/src/manual/solver.md:49-56
    s = lmpc.default_settings()
    assert (s.primal_tol, s.dual_tol, s.progress_tol, s.cycle_tol, s.iter_limit, s.rho_soft) == \
        (1e-6, 1e-12, 1e-6, 10, 10000, 1e-6)
    so = oldp.default_settings()
    for name, _ in lmpc.Settings._fields_:
        if name in ("eps_prox", "eta_prox"):          # (the oracle takes the proximal settings as arguments)
            continue
        assert getattr(s, name) == getattr(so, name), name


@pytest.mark.parametrize("name", ["pendulum", "mass_spring", "mass_spring_3in", "preprocessing_kat", "soft_doc"])
def test_host_transform_matches_oracle_qp2ldp(lmpc, name):
    # library's C++ QP->LDP (lmpc_transform) vs the numpy restatement of codegen.jl:239-280
    g = load_golden(name)
    n = g["H"].shape[0]
    for nout in (1, n):
        t = lmpc.transform(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
        L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
        for k, ref in (("M", L.M), ("du", L.du0), ("dl", L.dl0), ("Dth", L.Dth), ("Rout", L.Rout),
                       ("x0", L.x0), ("Xth", L.Xth)):
            assert np.allclose(t[k], ref, rtol=1e-12, atol=1e-12), (name, k)
        assert np.allclose(np.linalg.norm(t["M"], axis=1), 1.0)


def test_transform_with_prestabilising_feedback_and_linear_term(lmpc):
    rng = np.random.default_rng(0)
    n, nth, nx, nout = 4, 5, 3, 2
    Hh = rng.standard_normal((n, n)); H = Hh @ Hh.T + np.eye(n)
    f = rng.standard_normal(n); f_theta = rng.standard_normal((n, nth))
    A = rng.standard_normal((3, n)); bu = np.ones(5); bl = -np.ones(5); W = rng.standard_normal((5, nth))
    K = rng.standard_normal((nout, nx))
    t = lmpc.transform(H, f, f_theta, A, bu, bl, W, None, nout=nout, K=K, nx=nx)
    L = oldp.qp2ldp(H, f, f_theta, A, bu, bl, W, np.zeros(5, np.int32), nout=nout, K=K)
    for k, ref in (("M", L.M), ("du", L.du0), ("dl", L.dl0), ("Dth", L.Dth), ("Rout", L.Rout), ("x0", L.x0), ("Xth", L.Xth)):
        assert np.allclose(t[k], ref, rtol=1e-11, atol=1e-11), k


def test_transform_error_codes(lmpc):
    g = load_golden("pendulum")
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.transform(-g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"])
    assert e.value.code == -5 and "positive definite" in str(e.value)
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.transform(g["H"], g["f"], g["f_theta"], g["A"], g["bl"], g["bu"], g["W"], g["senses"])
    assert e.value.code == -1


def test_no_cpu_fallback(lmpc, has_gpu):
    if has_gpu:
        pytest.skip("a GPU is present")
    g = load_golden("pendulum")
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"])
    assert e.value.code == -101


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "linearmpc.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle/" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_form_parameter_layout(lmpc):
    # /root/reference/src/explicit.jl:54-63: theta = [x; r; d; uprev; p]
    g = load_golden("pendulum")
    mpc = lmpc.MPC(lmpc.MPQP(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"]),
                   nx=4, nu=1, nr=2, nuprev=1)
    assert mpc.get_parameter_dims() == (4, 2, 0, 1, 0)
    th = mpc.form_parameter([1, 2, 3, 4], r=[5, 6], uprev=[7])
    assert np.array_equal(th, [1, 2, 3, 4, 5, 6, 7])
    assert np.array_equal(mpc.form_parameter([1, 2, 3, 4]), [1, 2, 3, 4, 0, 0, 0])   # r, uprev default 0
    with pytest.raises(ValueError):
        mpc.form_parameter([1, 2, 3])
    with pytest.raises(ValueError):
        mpc.form_parameter([1, 2, 3, 4], r=[1.0])
    TH = mpc.form_parameter_batch(np.ones((3, 4)), R=[5, 6], Uprev=np.arange(3)[:, None])
    assert TH.shape == (3, 7) and np.array_equal(TH[:, 4:6], [[5, 6]] * 3) and np.array_equal(TH[:, 6], [0, 1, 2])


def test_shard_bounds(lmpc):
    for n, w in [(10, 3), (1_000_000, 8), (5, 8), (0, 2), (64, 1)]:
        spans = [lmpc.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == lmpc.shard_counts(n, w)


def test_region_discovery_bookkeeping(lmpc):
    # sampling-based critical-region discovery, host logic driven by the oracle here.
    # With 5 box-constrained moves there are at most 3^5 optimal active sets; on the example's +-20
    # range ~48 of them are ever optimal (the count saturates: 43 @6e4, 45 @5e5, 48 @3e6 samples).
    # (The "> 100" of /root/reference/test/runtests.jl:199-204 counts ASCertain's partition by
    # active-set SEQUENCE, a different object.)
    from conftest import oracle_ldp_from
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0])
    ub = np.array([20.0] * 4 + [20.0, 0.0] + [2.0])
    theta = lmpc.explicit.sample_range(lb, ub, 60000, seed=1)
    assert theta.shape == (60000, 7) and np.all(theta[:, 5] == 0.0)
    out = lmpc.explicit.discover_regions(lambda th: oldp.solve_batch(L, th), theta)
    assert 40 <= len(out["masks"]) <= 243 and out["counts"].sum() == out["n_solved"] == 60000
    assert out["counts"][0] >= out["counts"][-1]
    # the affine law of a region reproduces the implicit solution inside it
    # (explicit == implicit, /root/reference/test/runtests.jl:178-183, :319, :381 to 1e-10)
    X, ef, it, act = oldp.solve_batch(L, theta[:400])
    for i in range(0, 400, 7):
        Fz, gz = lmpc.explicit.affine_law(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], act[i])
        assert np.abs(Fz @ theta[i] + gz - X[i]).max() < 1e-8
    up, lo = lmpc.explicit.mask_to_sets(np.array([0b0000100010], np.uint64), 5)
    assert up == [1] and lo == [0]


def test_preview_formatting_follows_the_reference():
    # utils.jl:78-261: vector -> repeated over the horizon; trajectory -> flattened column by column,
    # cut at Np or padded with its last column; wrong sizes raise with the reference's messages
    import linearmpc_jl_amd as lmpc
    q = lmpc.MPQP(np.eye(2), np.zeros(2), np.zeros((2, 11)), np.zeros((0, 2)), np.ones(2), -np.ones(2),
                  np.zeros((2, 11)), np.zeros(2, np.int32))
    mpc = lmpc.MPC(q, nx=1, nu=1, nr=2 * 3, nd=1 * 3, np_=1, Np=3, reference_preview=True, disturbance_preview=True)
    assert np.array_equal(mpc.format_reference([1.0, 2.0]), [1, 2, 1, 2, 1, 2])
    assert np.array_equal(mpc.format_reference(np.array([[1.0, 3, 5, 7], [2, 4, 6, 8]])), [1, 2, 3, 4, 5, 6])
    assert np.array_equal(mpc.format_reference(np.array([[1.0, 3], [2, 4]])), [1, 2, 3, 4, 3, 4])
    assert np.array_equal(mpc.format_reference(None), np.zeros(6))
    with pytest.raises(ValueError, match="must match number of outputs"):
        mpc.format_reference([1.0, 2.0, 3.0])
    with pytest.raises(ValueError, match="must have 2 rows"):
        mpc.format_reference(np.zeros((3, 4)))
    assert np.array_equal(mpc.format_disturbance([0.5]), [0.5, 0.5, 0.5])
    assert np.array_equal(mpc.format_disturbance(np.array([[1.0, 2.0]])), [1, 2, 2])
    assert np.array_equal(mpc.format_affine_parameters(np.array([[9.0, 8.0]])), [9.0])      # no parameter preview
    th = mpc.form_parameter([7.0], r=[1.0, 2.0], d=[0.5], p=[4.0])
    assert np.array_equal(th, [7, 1, 2, 1, 2, 1, 2, 0.5, 0.5, 0.5, 4])
    plain = lmpc.MPC(q, nx=1, nu=1, nr=2, nd=0, nuprev=0, np_=0)
    assert np.array_equal(plain.format_reference(np.array([[1.0, 3], [2, 4]])), [1, 2])    # first column
    with pytest.raises(ValueError, match="must match number of outputs"):
        plain.format_reference([1.0])


def test_c_client_links_and_runs(tmp_path):
    """A plain C99 program (examples/abi_check.c) compiled with gcc against include/lmpc_hip.h and linked to
    the shared library: the header is valid C (not only C++), every symbol it uses links with C linkage, the
    host-only transform runs, and without a GPU setup refuses loudly (with one it solves and checks)."""
    import shutil
    import subprocess
    import linearmpc_jl_amd as lmpc
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = tmp_path / "abi_check"
    libdir = os.path.dirname(lmpc.LIB_PATH)
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "abi_check.c"), "-o", str(exe), "-L", libdir, "-llmpc_hip",
                    f"-Wl,-rpath,{libdir}", "-lm"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "ok" in out.stdout


def test_multi_device_partition(lmpc):
    # lmpc_multi_partition: contiguous shards, remainder on the leading devices
    for N, nd in [(10, 1), (10, 3), (1_000_003, 8), (5, 8), (0, 4)]:
        off = lmpc.MultiQP.partition(N, nd)
        assert off[0] == 0 and off[-1] == N and len(off) == nd + 1
        sizes = np.diff(off)
        assert sizes.min() >= 0 and sizes.max() - sizes.min() <= 1
        assert np.all(np.diff(sizes) <= 0)          # the larger shards come first
    assert lmpc.MultiQP.partition(10, 3) == list(np.cumsum([0] + lmpc.shard_counts(10, 3)))


def test_multi_setup_refused_without_gpu(lmpc, has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.MultiQP.from_mpqp(np.eye(2), np.zeros(2), np.zeros((2, 1)), np.zeros((0, 2)), np.ones(2), -np.ones(2),
                               np.zeros((2, 1)))
    assert e.value.code == -101


def test_setup_keywords_of_the_reference(lmpc, has_gpu):
    """setup.jl:11-13: DAQP.setup(...; break_points = mpQP.break_points, is_avi = !mpQP.is_symmetric).  The C entry
    lmpc_setup_ex carries both; what each combination answers WITHOUT a device (argument checks and the host
    transform run before the GPU is touched)."""
    H = np.array([[2.0, 0.5], [0.0, 2.0]])                      # not symmetric, symmetric part positive definite
    args = (np.zeros(2), np.zeros((2, 1)), np.zeros((0, 2)), np.ones(2), -np.ones(2), np.zeros((2, 1)))
    with pytest.raises(lmpc.LmpcError) as e:                    # the QP -> LDP transform does not apply to it
        lmpc.transform(H, *args)
    assert e.value.code == -103
    # prioritised constraints: refused loudly, by the library, whatever else is asked
    q = lmpc.MPQP(np.eye(2), *args, np.zeros(2, np.int32), break_points=np.array([1, 2], np.int32))
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.MPC(q, nx=1, nu=1).setup()
    assert e.value.code == -103 and "break_points" in str(e.value)
    # is_avi = 0 with a non-symmetric H: the caller contradicts itself
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.BatchedQP.from_mpqp(H, *args, is_avi=False)
    assert e.value.code == -100
    # H + H' not positive definite: DAQP's non-convex flag
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.BatchedQP.from_mpqp(np.array([[1.0, 3.0], [0.0, 1.0]]), *args, is_avi=True)
    assert e.value.code == -5
    # a proper variational problem passes every host-side check: without a GPU the answer is NOGPU, not UNSUPPORTED
    q2 = lmpc.MPQP(H, *args, np.zeros(2, np.int32), is_symmetric=False)
    if not has_gpu:
        with pytest.raises(lmpc.LmpcError) as e:
            lmpc.MPC(q2, nx=1, nu=1).setup()
        assert e.value.code == -101
        with pytest.raises(lmpc.LmpcError) as e:                # lmpc_setup decides is_avi from H by itself
            lmpc.BatchedQP.from_mpqp(H, *args)
        assert e.value.code == -101


def test_semidefinite_hessian_needs_eps_prox(lmpc, has_gpu):
    """A merely positive semidefinite H: DAQP.setup answers -5 ("Nonconvex objective", /root/reference/src/setup.jl:18-19)
    unless the eps_prox setting is positive; lmpc_setup the same -- with eps_prox > 0 the host-side checks pass and the
    answer without a GPU is NOGPU.  The settings struct carries eps_prox / eta_prox behind DAQP's documented six
    (ABI version 2); an indefinite H is refused either way."""
    assert lmpc.lib().lmpc_abi_version() >= 2
    s = lmpc.default_settings()
    assert s.eps_prox == 0.0 and s.eta_prox == 1e-6
    B = np.array([[1.0], [1.0], [0.5]])
    H = B @ B.T                                                   # rank 1 of 3
    args = (np.zeros(3), np.zeros((3, 1)), np.zeros((0, 3)), np.ones(3), -np.ones(3), np.zeros((3, 1)))
    with pytest.raises(lmpc.LmpcError) as e:
        lmpc.BatchedQP.from_mpqp(H, *args)
    assert e.value.code == -5
    s.eps_prox = 1e-4
    if not has_gpu:
        with pytest.raises(lmpc.LmpcError) as e:
            lmpc.BatchedQP.from_mpqp(H, *args, settings=s)
        assert e.value.code == -101
    with pytest.raises(lmpc.LmpcError) as e:                      # indefinite: still non-convex
        lmpc.BatchedQP.from_mpqp(H - 0.5 * np.eye(3), *args, settings=s)
    assert e.value.code == -5


def test_host_transform_of_a_variational_problem_matches_the_oracle(lmpc):
    """lmpc_transform_avi (what an is_avi setup precomputes on the host) against the oracle's qp2avi on the
    reference's game-theoretic test problem (test/runtests.jl:1337-1358) and on a random problem with general rows."""
    from oracle import avi as oavi, mpc2mpqp as omm
    p = omm.game_kat()
    q = omm.mpc2mpqp(p)
    assert not q.is_symmetric and q.n == 6 and q.m == 6 and q.nth == 6
    rng = np.random.default_rng(3)
    n, mg, ms, nth = 5, 7, 3, 4
    B = rng.normal(size=(n, n)); K = rng.normal(size=(n, n))
    Hr = B @ B.T + np.eye(n) + (K - K.T)
    cases = [(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, 2, None, 0),
             (Hr, rng.normal(size=n), rng.normal(size=(n, nth)), rng.normal(size=(mg, n)), rng.uniform(1, 2, ms + mg),
              -rng.uniform(1, 2, ms + mg), rng.normal(size=(ms + mg, nth)), np.zeros(ms + mg, np.int32), 2,
              rng.normal(size=(2, 2)), 2)]
    for (H, f, fth, A, bu, bl, W, sn, nout, Kfb, nx) in cases:
        got = lmpc.transform_avi(H, f, fth, A, bu, bl, W, sn, nout=nout, K=Kfb, nx=nx)
        ref = oavi.qp2avi(H, f, fth, A, bu, bl, W, sn, nout=nout, K=Kfb)
        for k, r in (("ML", ref.ML), ("MR", ref.MR), ("G", ref.G), ("du", ref.du0), ("dl", ref.dl0), ("Dth", ref.Dth),
                     ("Rout", ref.Rout), ("x0", ref.x0), ("Xth", ref.Xth)):
            assert np.allclose(got[k], r, rtol=1e-10, atol=1e-11), k
        assert np.allclose(np.diag(got["G"]), 1.0, atol=1e-12) and not np.allclose(got["G"], got["G"].T)


# ------------------------------------------------------------------ reference-held vectors, verbatim
def _fmt_mpc(lmpc, nx, ny, nd, Np, rp=False, dp=False):
    nr = ny * (Np if rp else 1)
    ndd = nd * (Np if dp else 1)
    nth = nx + nr + ndd
    q = lmpc.MPQP(np.eye(1), np.zeros(1), np.zeros((1, nth)), np.zeros((0, 1)), np.ones(1), -np.ones(1),
                  np.zeros((1, nth)), np.zeros(1, np.int32))
    return lmpc.MPC(q, nx=nx, nu=1, nr=nr, nd=ndd, Np=Np, reference_preview=rp, disturbance_preview=dp)


def test_reference_formatting_vectors_verbatim(lmpc):
    """/root/reference/test/runtests.jl:1401-1428 "Reference and disturbance formatting helpers", number for
    number: MPC([1 1; 0 1], [0; 1]; C = I, Np = 4), then the scalar plant with Gd = 1."""
    mpc = _fmt_mpc(lmpc, 2, 2, 0, 4, rp=True)
    assert np.allclose(mpc.format_reference([1.0, 2.0]), np.tile([1.0, 2.0], 4))                               # :1406
    assert np.allclose(mpc.format_reference(np.array([[1.0, 2, 3, 4, 5], [10.0, 20, 30, 40, 50]])),
                       [1.0, 10.0, 2.0, 20.0, 3.0, 30.0, 4.0, 40.0])                                          # :1407-1408
    assert np.allclose(mpc.format_reference(np.array([[1.0, 2.0], [10.0, 20.0]])),
                       [1.0, 10.0, 2.0, 20.0, 2.0, 20.0, 2.0, 20.0])                                          # :1409-1410
    for bad in ([1.0], np.ones((1, 2)), 1.0):                                                                 # :1411-1413
        with pytest.raises(ValueError):
            mpc.format_reference(bad)
    plain = _fmt_mpc(lmpc, 2, 2, 0, 4, rp=False)
    assert np.allclose(plain.format_reference(np.array([[7.0, 8.0, 9.0], [1.0, 2.0, 3.0]])), [7.0, 1.0])      # :1415-1416
    dmpc = _fmt_mpc(lmpc, 1, 1, 1, 4, dp=True)
    assert np.allclose(dmpc.format_disturbance([3.0]), [3.0, 3.0, 3.0, 3.0])                                   # :1421
    assert np.allclose(dmpc.format_disturbance(np.array([[1.0, 2.0]])), [1.0, 2.0, 2.0, 2.0])                  # :1422
    for bad in ([1.0, 2.0], np.ones((2, 2))):                                                                  # :1423-1424
        with pytest.raises(ValueError):
            dmpc.format_disturbance(bad)
    dplain = _fmt_mpc(lmpc, 1, 1, 1, 4, dp=False)
    assert np.allclose(dplain.format_disturbance(np.array([[7.0, 8.0, 9.0]])), [7.0])                          # :1427


def test_reference_parameter_dims_verbatim():
    """get_parameter_dims as the reference's tests assert it: runtests.jl:253-257 (reference preview),
    :370-374 (disturbance preview), :1147-1150 (generalised parameter preview with its formatting)."""
    from oracle import mpc2mpqp as omm
    F, G = omm.zoh(np.array([[0.0, 1.0], [10.0, 0.0]]), np.array([[0.0], [1.0]]), 0.1)
    p = omm.make_mpc(F, G, np.eye(2), Np=5, Nc=3, Q=[1.0, 1.0], R=[0.1], Rr=[0.1], umin=[-20.0], umax=[20.0], Ts=0.1)
    p.reference_preview = True
    assert p.parameter_dims()[:4] == (2, 2 * 5, 0, 1)                                                          # :253-257
    assert omm.mpc2mpqp(p).nth == 2 + 10 + 1
    p = omm.make_mpc([[1.0, 1.0], [0.0, 1.0]], [[0.0], [1.0]], [[1.0, 0.0]], Np=5, Nc=5, Q=[10.0], R=[0.1],
                     umin=[-0.5], umax=[0.5], Gd=[[0.0], [1.0]])
    p.disturbance_preview = True
    assert p.parameter_dims()[:4] == (2, 1, 5, 0)                                                              # :370-374
    assert omm.mpc2mpqp(p).nth == 8
    p = omm.make_mpc([[1, 1], [0, 1]], [[0], [1]], np.eye(2), Np=5, Nc=3, Q=[1.0, 1.0], R=[0.1], umin=[-2.0], umax=[2.0])
    p.Eu = np.array([[1.0]])
    p.parameter_preview = True
    assert p.parameter_dims() == (2, 2, 0, 0, 5)                                                               # :1147-1148
    th = omm.form_parameter(p, [0.0, 0.0], par=np.array([0.25]))
    assert np.array_equal(th[4:], np.full(5, 0.25))                                                            # :1149
    th = omm.form_parameter(p, [0.0, 0.0], par=np.array([[0.25, 0.5]]))
    assert np.array_equal(th[4:], [0.25, 0.5, 0.5, 0.5, 0.5])                                                  # :1150


def test_reference_move_block_structure_verbatim():
    """/root/reference/test/runtests.jl:138-176 "Move blocking" on the `aircraft` example (Np = 10, two inputs):
    the padded / clipped / per-input block vectors and the number of decision variables."""
    from oracle import mpc2mpqp as omm
    p = omm.aircraft(10)
    p.move_block([])
    assert len(omm.mpc2mpqp(p).f) == 10 * p.nu                                                                 # :144-146
    p.move_block([1, 1, 2, 3, 3])
    assert len(omm.mpc2mpqp(p).f) == 5 * p.nu                                                                  # :148-150
    for arg, want in (([1, 1], [[1, 9], [1, 9]]),                                                              # :153-155 pad
                      ([2, 3, 3, 6, 8, 9], [[2, 3, 3, 2], [2, 3, 3, 2]]),                                      # :158-160 clip
                      (2, [[2, 2, 2, 2, 2], [2, 2, 2, 2, 2]]),                                                 # :162-164
                      (3, [[3, 3, 3, 1], [3, 3, 3, 1]]),                                                       # :166-168
                      ([[1, 2, 3], [4, 2]], [[1, 2, 7], [4, 6]]),                                              # :170-172
                      ([[1, 2, 3, 15, 20], [2]], [[1, 2, 3, 4], [10]])):                                       # :174-176
        p.move_block(arg)
        q = omm.mpc2mpqp(p)
        assert p.move_blocks == want
        assert len(q.f) == sum(len(mb) for mb in want) and q.H.shape == (len(q.f), len(q.f))


def test_marginal_case_report_finds_a_constructed_marginal_point():
    """oracle.ldp.marginal_report: a parameter point placed 1e-7 inside the boundary where
    a bound becomes active is counted as primal-marginal; generic points are not."""
    from oracle import ldp as oldp
    g = load_golden("pendulum")
    L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
    th_in = np.zeros(7)                                   # at the origin no bound is active
    th_out = g["theta"][0]                                # K1: the second move sits on its upper bound
    lo, hi = 0.0, 1.0
    for _ in range(80):                                   # bisect for the point where the active set changes
        mid = 0.5 * (lo + hi)
        act = oldp.solve_batch(L, (th_in + mid * (th_out - th_in))[None])[3]
        if act.any():
            hi = mid
        else:
            lo = mid
    rep = oldp.marginal_report(L, np.vstack([th_in + (lo - 1e-9) * (th_out - th_in), th_in, th_out]))
    assert rep["primal_marginal"] == 1 and rep["primal_marginal_first"] == [0] and rep["dual_marginal"] == 0


def test_bench_helpers_cover_every_workload():
    """bench.py's synthetic inputs and bookkeeping (no GPU): every workload's theta has the width of its fixture,
    the rotation defeats the 256 MiB Infinity Cache, the marginal report runs on a bench batch."""
    import bench
    for w, name in (("pendulum", "pendulum"), ("mass_spring_3in", "mass_spring_3in"), ("hybrid", "satellite20"),
                    ("soft_doc", "soft_doc"), ("pendulum_N50", "pendulum_N50"), ("pendulum_N125", "pendulum_N125")):
        g = bench.make_problem(name)
        th = bench.make_theta(name, 257, 5, hard=False)
        assert th.shape == (257, g["f_theta"].shape[1]) and th.flags.c_contiguous and np.isfinite(th).all()
    assert bench.make_theta("pendulum", 64, 1, hard=True).std() > bench.make_theta("pendulum", 64, 1).std()
    assert bench.make_theta("mass_spring", 9, 1).shape == (9, bench.make_problem("mass_spring")["f_theta"].shape[1])
    assert np.abs(bench.make_theta("mass_spring_3in", 64, 1, hard="feasible")).max() <= 1.5
    per = bench.algorithmic_bytes(7, 1)
    assert per == 68
    nrot = bench.rotation_depth(1_000_000, per)
    assert nrot * 1_000_000 * per > 1.25 * bench.L3_BYTES and nrot <= 8
    assert bench.rotation_depth(100, per) == 64           # tiny batches: the cap (everything is cache-resident anyway)


def test_bench_line_is_compact_and_strict_json():
    """The driver keeps only a tail of stdout: the bench line must stay a few KB whatever the full report holds
    (round 3's line grew to 37 KB and was not parsed).  Built here from a full report of every configuration with
    padded notes; strict JSON (no NaN / Infinity), the contract's keys present, under the limit."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_final.json")))
    full["roofline"]["frac"] = float("nan")                # a non-finite number must not reach the line as NaN
    full["config"]["notes"] = "x" * 50000                   # whatever the full report grows to
    for c in full["configs"].values():
        c["verification"] = {"blob": list(range(3000))}
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT and len(line) < 4096 + 2048
    obj = json.loads(line, parse_constant=lambda s: (_ for _ in ()).throw(ValueError(s)))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in obj, k
    assert obj["roofline"]["frac"] is None and obj["roofline"]["peak"] == 8000.0
    assert set(obj["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(obj["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert "model" not in obj["config"] and obj["config"]["workload"].startswith("pendulum")
    assert set(obj["configs"]) == set(full["configs"])
    # a report with very many configurations still yields a line under the limit (configs dropped, said so)
    full["configs"] = {f"c{i}": dict(full["configs"]["pendulum_N50"]) for i in range(200)}
    line2 = bench.compact_line(full)
    assert len(line2) < bench.LINE_LIMIT and "truncated" in json.loads(line2)


def test_settings_layout_is_the_same_in_every_binding(lmpc):
    """`lmpc_settings` as the C header declares it, as the ctypes mirror declares it and as the (never executed) Julia glue
    declares it: same fields, same order, same types -- a drift in the Julia struct would be silent otherwise."""
    import ctypes
    import re
    from linearmpc_jl_amd._cabi import Settings
    hdr = open(os.path.join(ROOT, "include", "lmpc_hip.h")).read()
    body = re.search(r"typedef struct lmpc_settings \{(.*?)\} lmpc_settings;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = [(m.group(2), m.group(1)) for m in re.finditer(r"\b(double|int32_t)\s+(\w+)\s*;", body)]
    py_fields = [(n, {ctypes.c_double: "double", ctypes.c_int32: "int32_t"}[t]) for n, t in Settings._fields_]
    assert c_fields == py_fields and len(c_fields) == 10
    assert ctypes.sizeof(Settings) == 72 and Settings.cycle_tol.offset == 48 and Settings.iter_limit.offset == 52 \
        and Settings.eps_prox.offset == 56
    jl = open(os.path.join(ROOT, "integration", "LmpcHipExt.jl")).read()
    sbody = re.search(r"struct LmpcSettings[^\n]*\n(.*?)\nend", jl, re.S).group(1)
    jl_fields = [(m.group(1), {"Cdouble": "double", "Cint": "int32_t"}[m.group(2)]) for m in re.finditer(r"(\w+)::(Cdouble|Cint)", sbody)]
    assert jl_fields == c_fields
    # ... and the constructor passes the fields in that order
    ctor = re.search(r"return LmpcSettings\((.*?)\)\nend", jl, re.S).group(1)
    assert [a.strip().split(".")[-1] for a in ctor.replace("\n", " ").split(",")] == [n for n, _ in c_fields]


def test_row_kernel_column_order_tables_are_valid_layouts():
    """The row kernel's factor layouts with a searched column order (lmpc_row_kernel.hpp rowp_cbm: 31 rows, and 48 rows
    with every column padded by one entry): no two entries share an address, everything lies inside rowp_size, and the
    16 columns of each slot of positions start at 16 different offsets modulo 16 (what makes column accesses free of LDS
    bank conflicts)."""
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "linearmpc.jl_amd", "csrc", "lmpc_row_kernel.hpp")).read()

    def table(name):
        m = re.search(r"constexpr int " + name + r"\[\d+\] = \{([^}]*)\}", src)
        assert m, name
        return [int(v) for v in m.group(1).replace("\n", " ").split(",")]

    for name, rows, lrow, size in (("k31", 31, 31, None), ("k48", 48, 49, 1291), ("k44", 44, 45, 1095), ("k16", 16, 17, 171)):
        cbm = table(name)
        assert len(cbm) == rows - 1
        if size is None:                                    # rowp_cb(capp, capp - 1) of the unpadded layout
            q, r = (rows - 1) >> 2, (rows - 1) & 3
            size = 4 * q * rows - 8 * q * (q - 1) + r * (rows - 4 * q)
        used = set()
        for t, c in enumerate(cbm):
            p0 = t & ~3
            for p in range(p0, lrow):                        # column t holds rows p0(t) .. lrow-1 at cbm(t) + p
                a = c + p
                assert 0 <= a < size and a not in used, (name, t, p)
                used.add(a)
        for g in range((rows - 1 + 15) // 16):
            starts = [cbm[t] % 16 for t in range(16 * g, min(16 * g + 16, rows - 1))]
            assert len(set(starts)) == len(starts), (name, g)
