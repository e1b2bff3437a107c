"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the QP -> LDP transform and a ctypes
front end to the C oracle solver (oracle/daqp_ldp_oracle.c).

Nothing in the shipped package may import this; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do.

`qp2ldp` follows /root/reference/src/codegen.jl:239-280 (`qp2ldp`) line by line:

    R      = chol((H+H')/2)                         :242   (upper factor, H = R'R)
    Mext   = [I_nb ; A] / R.U                        :243
    Vth    = R.L \\ f_theta ;  v = R.L \\ f           :244-245
    Dth    = W + Mext*Vth ;  du/dl = bu/bl + Mext*v  :246-249
    rows of Mext, Dth, du, dl divided by |Mext_i|    :252-264
    Uth_offset = -(H \\ f_theta)[1:nout,:]            :269-270
    u_offset   = -(H \\ f)[1:nout]                    :272-273

and adds what the batched backend needs on top: the first `nout` rows of R^-1 (un-normalised)
so that the primal solution is recovered as  x = R^-1 u + u_offset + Uth_offset*theta
(codegen/mpc_update_qp.c:14-22, where `uscaling[i]*xstar[i]` is the same product), and the
prestabilising-feedback correction `Uth_offset[:, :nx] -= K` (codegen.jl:157, utils.jl:48-49).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


@dataclass
class LDP:
    n: int
    m: int
    ms: int
    nth: int
    nout: int
    M: np.ndarray       # m x n  normalised rows
    du0: np.ndarray     # m
    dl0: np.ndarray     # m
    Dth: np.ndarray     # m x nth
    Rout: np.ndarray    # nout x n
    x0: np.ndarray      # nout
    Xth: np.ndarray     # nout x nth
    sense: np.ndarray   # m int32
    scale: np.ndarray   # m row norms before normalisation

    def contiguous(self):
        for name in ("M", "du0", "dl0", "Dth", "Rout", "x0", "Xth", "scale"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.float64))
        self.sense = np.ascontiguousarray(self.sense, dtype=np.int32)
        return self


def qp2ldp(H, f, f_theta, A, bu, bl, W, sense, nout, K=None) -> LDP:
    H = np.asarray(H, float)
    n = H.shape[0]
    f = np.asarray(f, float).reshape(n)
    f_theta = np.asarray(f_theta, float).reshape(n, -1)
    nth = f_theta.shape[1]
    bu = np.asarray(bu, float).reshape(-1)
    bl = np.asarray(bl, float).reshape(-1)
    m = bu.size
    A = np.asarray(A, float).reshape(-1, n)
    ms = m - A.shape[0]
    W = np.asarray(W, float).reshape(m, nth)
    Rl = np.linalg.cholesky((H + H.T) / 2)          # lower; R.U = Rl.T
    Rinv = np.linalg.solve(Rl.T, np.eye(n))         # R^-1 (upper triangular)
    Mext = np.vstack([np.eye(n)[:ms], A]) @ Rinv
    Vth = np.linalg.solve(Rl, f_theta)
    v = np.linalg.solve(Rl, f)
    Dth = W + Mext @ Vth
    shift = Mext @ v
    du, dl = bu + shift, bl + shift
    scale = np.linalg.norm(Mext, axis=1)
    nzr = scale > 0
    Mext[nzr] /= scale[nzr, None]
    Dth[nzr] /= scale[nzr, None]
    du[nzr] /= scale[nzr]
    dl[nzr] /= scale[nzr]
    Xth = -np.linalg.solve(H, f_theta)[:nout]
    x0 = -np.linalg.solve(H, f)[:nout]
    if K is not None:
        K = np.atleast_2d(np.asarray(K, float))
        Xth = Xth.copy()
        Xth[:K.shape[0], :K.shape[1]] -= K
    return LDP(n, m, ms, nth, nout, Mext, du, dl, Dth, Rinv[:nout].copy(), x0, Xth,
               np.asarray(sense, np.int32).reshape(m), scale).contiguous()


# ---------------------------------------------------------------- ctypes front end
class _CLdp(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("m", ctypes.c_int32), ("ms", ctypes.c_int32),
                ("nth", ctypes.c_int32), ("nout", ctypes.c_int32),
                ("M", ctypes.c_void_p), ("du0", ctypes.c_void_p), ("dl0", ctypes.c_void_p),
                ("Dth", ctypes.c_void_p), ("Rout", ctypes.c_void_p), ("x0", ctypes.c_void_p),
                ("Xth", ctypes.c_void_p), ("sense", ctypes.c_void_p)]


class Settings(ctypes.Structure):
    """Field order = oracle_settings in daqp_ldp_oracle.c; defaults = solver.md:49-56."""
    _fields_ = [("primal_tol", ctypes.c_double), ("dual_tol", ctypes.c_double),
                ("zero_tol", ctypes.c_double), ("progress_tol", ctypes.c_double),
                ("fval_bound", ctypes.c_double), ("rho_soft", ctypes.c_double),
                ("cycle_tol", ctypes.c_int32), ("iter_limit", ctypes.c_int32),
                ("mode", ctypes.c_int32), ("pad_", ctypes.c_int32)]      # mode 1: Gram-scan form (twin of the GRAM kernels)


_lib = None


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ("daqp_ldp_oracle.c", "daqp_ldp_oracle_f32.c", "daqp_avi_oracle.c", "Makefile")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


def use_native():
    """Switch to a -O3 -march=native build of the same sources, compiled on THIS machine (the CPU baseline
    leg of bench.py: SURVEY.md section 8d asks for the host's own instruction set).  Same arithmetic
    contract, so results do not change.  Returns the flags description; falls back to the portable
    build if the compiler is missing."""
    global _lib, _LIB_PATH
    native = os.path.join(_HERE, "_build", "liboracle_native.so")
    try:
        subprocess.run(["make", "-C", _HERE, "-s", "native"], check=True, capture_output=True)
    except (OSError, subprocess.CalledProcessError):
        return "-O3 -march=x86-64-v3"
    _LIB_PATH = native
    _lib = None
    lib()
    return "-O3 -march=native"


def lib():
    global _lib
    if _lib is None:
        if not _LIB_PATH.endswith("_native.so"):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_active_words.restype = ctypes.c_int
        _lib.oracle_solve_batch.restype = None
        _lib.oracle_solve_batch_f32.restype = None
    return _lib


def default_settings() -> Settings:
    s = Settings()
    lib().oracle_default_settings(ctypes.byref(s))
    return s


def default_settings_f32() -> Settings:
    """Tolerances that make sense in binary32 (this build's choice; libdaqp's own single-precision
    defaults live in a header outside the reference tree)."""
    s = Settings()
    lib().oracle_default_settings_f32(ctypes.byref(s))
    return s


def active_words(m):
    return (2 * m + 63) // 64


def solve_batch(ldp: LDP, theta, settings: Settings | None = None, warm=None, dtype=np.float64):
    """Solve every row of theta (N x nth).  Returns X (N x nout), exitflag, iters, active.

    dtype=np.float32 runs the binary32 build of the oracle on the pack rounded to binary32 (what the
    library's f32 path does with its binary64 pack)."""
    L = lib()
    ldp.contiguous()
    f32 = np.dtype(dtype) == np.float32
    theta = np.asarray(theta, dtype)
    theta = np.ascontiguousarray(theta.reshape(-1, ldp.nth) if ldp.nth else theta.reshape(len(theta), 0))
    N = theta.shape[0]
    nw = active_words(ldp.m)
    X = np.empty((N, ldp.nout), dtype)
    ef = np.empty(N, np.int32)
    it = np.empty(N, np.int32)
    act = np.zeros((N, nw), np.uint64)
    arrs = [np.ascontiguousarray(a, dtype=dtype) for a in (ldp.M, ldp.du0, ldp.dl0, ldp.Dth, ldp.Rout, ldp.x0, ldp.Xth)]
    c = _CLdp(ldp.n, ldp.m, ldp.ms, ldp.nth, ldp.nout, *(a.ctypes.data for a in arrs), ldp.sense.ctypes.data)
    s = settings if settings is not None else (default_settings_f32() if f32 else default_settings())
    wptr = None
    if warm is not None:
        warm = np.ascontiguousarray(np.asarray(warm, np.uint64).reshape(N, nw))
        wptr = ctypes.c_void_p(warm.ctypes.data)
    (L.oracle_solve_batch_f32 if f32 else L.oracle_solve_batch)(ctypes.byref(c), ctypes.byref(s), ctypes.c_int64(N),
                         ctypes.c_void_p(theta.ctypes.data), wptr,
                         ctypes.c_void_p(X.ctypes.data), ctypes.c_void_p(ef.ctypes.data),
                         ctypes.c_void_p(it.ctypes.data), ctypes.c_void_p(act.ctypes.data))
    return X, ef, it, act


def runner(ldp: LDP, theta, settings: Settings | None = None, dtype=np.float64):
    """Timing front end (bench.py's cpu_baseline leg): all arguments marshalled ONCE, returns
    run(reps) that makes `reps` bare C calls over the same batch -- a few microseconds of Python per
    call instead of solve_batch's array set-up, which matters when a thread's slice is small."""
    L = lib()
    ldp.contiguous()
    f32 = np.dtype(dtype) == np.float32
    theta = np.ascontiguousarray(np.asarray(theta, dtype).reshape(-1, ldp.nth))
    N = theta.shape[0]
    nw = active_words(ldp.m)
    X = np.empty((N, ldp.nout), dtype)
    ef = np.empty(N, np.int32)
    it = np.empty(N, np.int32)
    arrs = [np.ascontiguousarray(a, dtype=dtype) for a in (ldp.M, ldp.du0, ldp.dl0, ldp.Dth, ldp.Rout, ldp.x0, ldp.Xth)]
    c = _CLdp(ldp.n, ldp.m, ldp.ms, ldp.nth, ldp.nout, *(a.ctypes.data for a in arrs), ldp.sense.ctypes.data)
    s = settings if settings is not None else (default_settings_f32() if f32 else default_settings())
    fn = L.oracle_solve_batch_repeat_f32 if f32 else L.oracle_solve_batch_repeat
    fn.restype = None
    args = (ctypes.byref(c), ctypes.byref(s), ctypes.c_int64(N), ctypes.c_void_p(theta.ctypes.data),
            ctypes.c_void_p(X.ctypes.data), ctypes.c_void_p(ef.ctypes.data), ctypes.c_void_p(it.ctypes.data))
    keep = (theta, X, ef, it, arrs, c, s)

    def run(reps=1):
        fn(*args, ctypes.c_int32(int(reps)))          # ONE C call: the interpreter lock is released for all of it
        return keep[1]
    run.N = N
    return run


def simulate(ldp: LDP, x0, T, F, G, r=None, uprev=None, settings: Settings | None = None, warm=True,
             dtype=np.float64):
    """Closed loop on the CPU oracle (oracle_simulate): returns dict(x, U (T,N,nu), X (T+1,N,nx), uprev, flag_min).
    dtype=np.float32: the binary32 build of the oracle on the pack and the plant rounded to binary32."""
    Lb = lib()
    ldp.contiguous()
    f32 = np.dtype(dtype) == np.float32
    F = np.ascontiguousarray(np.atleast_2d(np.asarray(F, np.float64)).astype(dtype))
    nx = F.shape[0]
    nu = ldp.nout
    G = np.ascontiguousarray(np.asarray(G, np.float64).reshape(nx, nu).astype(dtype))
    x = np.ascontiguousarray(np.array(np.asarray(x0, dtype).reshape(-1, nx), copy=True))
    N = x.shape[0]
    nr = 0 if r is None else np.asarray(r).reshape(N, -1).shape[1]
    nup = ldp.nth - nx - nr
    rr = None if nr == 0 else np.ascontiguousarray(np.asarray(r, dtype).reshape(N, nr))
    up = np.ascontiguousarray(np.zeros((N, max(nup, 1)), dtype) if uprev is None else
                              np.array(np.asarray(uprev, dtype).reshape(N, nup), copy=True))
    U = np.empty((T, N, nu), dtype)
    X = np.empty((T + 1, N, nx), dtype)
    fm = np.empty(N, np.int32)
    arrs = [np.ascontiguousarray(a, dtype=dtype) for a in (ldp.M, ldp.du0, ldp.dl0, ldp.Dth, ldp.Rout, ldp.x0, ldp.Xth)]
    c = _CLdp(ldp.n, ldp.m, ldp.ms, ldp.nth, ldp.nout, *(a.ctypes.data for a in arrs), ldp.sense.ctypes.data)
    s = settings if settings is not None else (default_settings_f32() if f32 else default_settings())
    fn = Lb.oracle_simulate_f32 if f32 else Lb.oracle_simulate
    fn.restype = None
    vp = ctypes.c_void_p
    fn(ctypes.byref(c), ctypes.byref(s), ctypes.c_int64(N), ctypes.c_int32(T), ctypes.c_int32(nx),
       ctypes.c_int32(nr), ctypes.c_int32(nup), vp(F.ctypes.data), vp(G.ctypes.data), vp(x.ctypes.data),
       vp(rr.ctypes.data) if rr is not None else None, vp(up.ctypes.data), vp(U.ctypes.data),
       vp(X.ctypes.data), vp(fm.ctypes.data), ctypes.c_int32(int(warm) if isinstance(warm, int) and not isinstance(warm, bool) else int(bool(warm))))
    return dict(x=x, U=U, X=X, uprev=up[:, :nup], flag_min=fm)


def marginal_report(ldp: LDP, theta, settings: Settings | None = None, dual_band=1e-9, max_list=20):
    """Classify the MARGINAL parameter points of a sample ("hard parts"): points
    where a terminal decision of the dual active-set method sits inside its tolerance band, so that
    another correct implementation (libdaqp itself) may legitimately stop on a different active set and
    differ from this oracle by up to O(primal_tol):

      primal-marginal  an INACTIVE row's slack at the terminal iterate is below primal_tol (the row is
                       within the band in which "violated" and "satisfied" are the same answer);
      dual-marginal    an ACTIVE row's multiplier is within `dual_band` of zero (the row could as
                       well have been left out).

    Slacks and multipliers are recomputed here from the final active set by a dense KKT solve in
    numpy (independent of the solver's recursions).  Returns counts and the first `max_list` indices."""
    s = settings if settings is not None else default_settings()
    theta = np.asarray(theta, float)
    theta = np.ascontiguousarray(theta.reshape(-1, ldp.nth) if ldp.nth else theta.reshape(len(theta), 0))
    X, ef, it, act = solve_batch(ldp, theta, s)
    m, n = ldp.m, ldp.n
    B = theta @ ldp.Dth.T                                  # shifts b_j per problem
    up_bits = np.zeros((len(theta), m), bool)
    lo_bits = np.zeros((len(theta), m), bool)
    for j in range(m):
        up_bits[:, j] = (act[:, j >> 6] >> np.uint64(j & 63)) & np.uint64(1)
        jj = m + j
        lo_bits[:, j] = (act[:, jj >> 6] >> np.uint64(jj & 63)) & np.uint64(1)
    imm = (ldp.sense & 4) != 0
    soft = (ldp.sense & 8) != 0
    ok = ef >= 1
    primal_list, dual_list = [], []
    n_primal = n_dual = 0
    keys = np.ascontiguousarray(act).view([("", act.dtype)] * act.shape[1]).reshape(-1)
    for key in np.unique(keys[ok]):
        idx = np.nonzero(ok & (keys == key))[0]
        i0 = idx[0]
        rows = np.nonzero(up_bits[i0] | lo_bits[i0])[0]
        lower = lo_bits[i0][rows]
        u = np.zeros((len(idx), n))
        if len(rows):
            Mw = ldp.M[rows]
            d = np.where(lower, ldp.dl0[rows] + B[np.ix_(idx, rows)], ldp.du0[rows] + B[np.ix_(idx, rows)])
            K = Mw @ Mw.T + np.diag(np.where(soft[rows], s.rho_soft, 0.0))
            lam = -np.linalg.solve(K, d.T).T               # u = -M_W' lam
            u = -lam @ Mw
            free = ~imm[rows]
            dm = (np.abs(lam[:, free]) < dual_band).any(axis=1) if free.any() else np.zeros(len(idx), bool)
            n_dual += int(dm.sum())
            for i in idx[dm][:max(0, max_list - len(dual_list))]:
                dual_list.append(int(i))
        inact = np.ones(m, bool)
        inact[rows] = False
        inact &= ~imm
        if inact.any():
            Mu = u @ ldp.M[inact].T
            su = (ldp.du0[inact] + B[np.ix_(idx, np.nonzero(inact)[0])]) - Mu
            sl = Mu - (ldp.dl0[inact] + B[np.ix_(idx, np.nonzero(inact)[0])])
            pm = (np.minimum(su, sl) < s.primal_tol).any(axis=1)
            n_primal += int(pm.sum())
            for i in idx[pm][:max(0, max_list - len(primal_list))]:
                primal_list.append(int(i))
    return {"sample": int(len(theta)), "solved": int(ok.sum()),
            "primal_marginal": n_primal, "dual_marginal": n_dual,
            "primal_tol": float(s.primal_tol), "dual_band": float(dual_band),
            "primal_marginal_first": sorted(primal_list), "dual_marginal_first": sorted(dual_list),
            "note": "points where libdaqp may legitimately end on a different active set (difference in u* up to "
                    "O(primal_tol)); everywhere else a strictly convex QP has one optimum and one active set"}
