"""TEST INFRASTRUCTURE ONLY -- the affine-variational-inequality mode of the online path: numpy restatement of the
host transform and a ctypes front end to oracle/daqp_avi_oracle.c.

Nothing in the shipped package may import this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

A controller with several objectives (/root/reference/src/setup.jl:137-151, mpc2mpqp.jl:900-950) has a non-symmetric
H and is set up with is_avi = true (/root/reference/src/setup.jl:11-13); the online call is the same `solve`
(/root/reference/src/utils.jl:268-283).  `qp2avi` is the counterpart of qp2ldp (codegen.jl:239-280) for that case:
no Cholesky factor exists, so the working coordinates are u = x - x_unc(theta) with x_unc = -H^-1 (f + f_theta theta):

    ML   = [I_ms ; A]                     rows scaled by s_j = 1/sqrt(ML_j H^-1 ML_j')   (so that G_jj = 1)
    MR_j = (H^-1 ML_j')'                  u = -sum_{j in W} MR_j lam_j
    G    = ML MR'                         m x m, not symmetric, G + G' positive definite on independent rows
    Dth  = s (W + [I;A] H^-1 f_theta),  du/dl = s (bu/bl + [I;A] H^-1 f)
    x    = Rout u + x0 + Xth theta,  Rout = I[:nout], x0 = -(H^-1 f)[:nout], Xth = -(H^-1 f_theta)[:nout] (- K)
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass

import numpy as np

from . import ldp as _ldp
from .ldp import Settings, active_words, default_settings


@dataclass
class AVI:
    n: int
    m: int
    ms: int
    nth: int
    nout: int
    ML: np.ndarray
    MR: np.ndarray
    G: np.ndarray
    du0: np.ndarray
    dl0: np.ndarray
    Dth: np.ndarray
    Rout: np.ndarray
    x0: np.ndarray
    Xth: np.ndarray
    sense: np.ndarray
    scale: np.ndarray

    def contiguous(self):
        for name in ("ML", "MR", "G", "du0", "dl0", "Dth", "Rout", "x0", "Xth", "scale"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.float64))
        self.sense = np.ascontiguousarray(self.sense, dtype=np.int32)
        return self


def qp2avi(H, f, f_theta, A, bu, bl, W, sense, nout, K=None) -> AVI:
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
    if np.linalg.eigvalsh((H + H.T) / 2)[0] <= 0:
        raise ValueError("H + H' is not positive definite")
    Hinv = np.linalg.inv(H)
    ML = np.vstack([np.eye(n)[:ms], A])
    MR = ML @ Hinv.T                                   # row j = (H^-1 ML_j')'
    Dth = W + ML @ (Hinv @ f_theta)
    shift = ML @ (Hinv @ f)
    du, dl = bu + shift, bl + shift
    g = np.einsum("ij,ij->i", ML, MR)
    scale = np.sqrt(np.where(g > 0, g, 1.0))
    scale[~(g > 0)] = 0.0
    nz = scale > 0
    ML[nz] /= scale[nz, None]; MR[nz] /= scale[nz, None]
    Dth[nz] /= scale[nz, None]; du[nz] /= scale[nz]; dl[nz] /= scale[nz]
    G = ML @ MR.T
    Xth = -(Hinv @ f_theta)[:nout]
    x0 = -(Hinv @ f)[:nout]
    if K is not None:
        K = np.atleast_2d(np.asarray(K, float))
        Xth = Xth.copy()
        Xth[:K.shape[0], :K.shape[1]] -= K
    return AVI(n, m, ms, nth, nout, ML, MR, G, du, dl, Dth, np.eye(n)[:nout].copy(), x0, Xth,
               np.asarray(sense, np.int32).reshape(m), scale).contiguous()


class _CAvi(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("m", ctypes.c_int32), ("ms", ctypes.c_int32), ("nth", ctypes.c_int32),
                ("nout", ctypes.c_int32)] + [(k, ctypes.c_void_p) for k in
                                              ("ML", "MR", "G", "du0", "dl0", "Dth", "Rout", "x0", "Xth", "sense",
                                               "Hinv", "x0f", "Xthf", "Kth")]


def _cavi(p: AVI, prox=None):
    p.contiguous()
    extra = (None, None, None, None) if prox is None else tuple(np.ascontiguousarray(prox[k], dtype=np.float64).ctypes.data
                                                                  for k in ("Hinv", "x0f", "Xthf", "Kth"))
    return _CAvi(p.n, p.m, p.ms, p.nth, p.nout, *(getattr(p, k).ctypes.data for k in
                                                   ("ML", "MR", "G", "du0", "dl0", "Dth", "Rout", "x0", "Xth", "sense")), *extra)


def solve_batch(p: AVI, theta, settings: Settings | None = None, warm=None):
    """Every row of theta (N x nth) -> X (N x nout), exitflag, iters, active masks (oracle_avi_solve_batch)."""
    L = _ldp.lib()
    theta = np.ascontiguousarray(np.asarray(theta, np.float64).reshape(-1, p.nth))
    N = theta.shape[0]
    nw = active_words(p.m)
    X = np.empty((N, p.nout)); ef = np.empty(N, np.int32); it = np.empty(N, np.int32)
    act = np.zeros((N, nw), np.uint64)
    s = settings if settings is not None else default_settings()
    wptr = None
    if warm is not None:
        warm = np.ascontiguousarray(np.asarray(warm, np.uint64).reshape(N, nw))
        wptr = ctypes.c_void_p(warm.ctypes.data)
    c = _cavi(p)
    L.oracle_avi_solve_batch.restype = None
    vp = ctypes.c_void_p
    L.oracle_avi_solve_batch(ctypes.byref(c), ctypes.byref(s), ctypes.c_int64(N), vp(theta.ctypes.data), wptr,
                             vp(X.ctypes.data), vp(ef.ctypes.data), vp(it.ctypes.data), vp(act.ctypes.data))
    return X, ef, it, act


def qp2prox(H, f, f_theta, A, bu, bl, W, sense, nout, eps, K=None):
    """The pack of the proximal-point mode (eps_prox > 0, H symmetric positive SEMIdefinite): the AVI pack of the
    regularised Hessian H + eps I plus what the outer iteration needs -- (H + eps I)^-1, the full-length affine map
    of the unconstrained optimum, and the feedback term of the outputs.  Returns (AVI, dict of the extra arrays)."""
    H = np.asarray(H, float)
    n = H.shape[0]
    if not np.allclose(H, H.T, rtol=1e-9, atol=0):
        raise ValueError("proximal iterations need a symmetric H")
    Ht = H + eps * np.eye(n)
    f = np.asarray(f, float).reshape(n)
    f_theta = np.asarray(f_theta, float).reshape(n, -1)
    P = qp2avi(Ht, f, f_theta, A, bu, bl, W, sense, nout=nout, K=K)
    Hinv = np.linalg.inv(Ht)
    Kth = np.zeros((nout, f_theta.shape[1]))
    if K is not None:
        K = np.atleast_2d(np.asarray(K, float))
        Kth[:K.shape[0], :K.shape[1]] -= K
    return P, {"Hinv": Hinv, "x0f": -(Hinv @ f), "Xthf": -(Hinv @ f_theta), "Kth": Kth}


def prox_solve_batch(p: AVI, prox, theta, eps, eta=1e-6, settings: Settings | None = None):
    """oracle_avi_prox_solve_batch: X (N x nout), exitflag, summed iterations, active masks of the last subproblem."""
    L = _ldp.lib()
    theta = np.ascontiguousarray(np.asarray(theta, np.float64).reshape(-1, p.nth))
    N = theta.shape[0]
    nw = active_words(p.m)
    X = np.empty((N, p.nout)); ef = np.empty(N, np.int32); it = np.empty(N, np.int32)
    act = np.zeros((N, nw), np.uint64)
    s = settings if settings is not None else default_settings()
    keep = {k: np.ascontiguousarray(prox[k], dtype=np.float64) for k in ("Hinv", "x0f", "Xthf", "Kth")}
    c = _cavi(p, keep)
    vp = ctypes.c_void_p
    L.oracle_avi_prox_solve_batch.restype = None
    L.oracle_avi_prox_solve_batch(ctypes.byref(c), ctypes.byref(s), ctypes.c_double(eps), ctypes.c_double(eta),
                                  ctypes.c_int64(N), vp(theta.ctypes.data), vp(X.ctypes.data), vp(ef.ctypes.data),
                                  vp(it.ctypes.data), vp(act.ctypes.data))
    return X, ef, it, act


def simulate(p: AVI, x0, T, F, G, r=None, uprev=None, settings: Settings | None = None, warm=False):
    """Closed loop (oracle_avi_simulate): dict(x, U (T,N,nu), X (T+1,N,nx), uprev, flag_min)."""
    L = _ldp.lib()
    F = np.ascontiguousarray(np.atleast_2d(np.asarray(F, np.float64)))
    nx, nu = F.shape[0], p.nout
    G = np.ascontiguousarray(np.asarray(G, np.float64).reshape(nx, nu))
    x = np.ascontiguousarray(np.array(np.asarray(x0, np.float64).reshape(-1, nx), copy=True))
    N = x.shape[0]
    nr = 0 if r is None else np.asarray(r).reshape(N, -1).shape[1]
    nup = p.nth - nx - nr
    rr = None if nr == 0 else np.ascontiguousarray(np.asarray(r, np.float64).reshape(N, nr))
    up = np.ascontiguousarray(np.zeros((N, max(nup, 1))) if uprev is None else
                              np.array(np.asarray(uprev, np.float64).reshape(N, nup), copy=True))
    U = np.empty((T, N, nu)); X = np.empty((T + 1, N, nx)); fm = np.empty(N, np.int32)
    s = settings if settings is not None else default_settings()
    c = _cavi(p)
    vp = ctypes.c_void_p
    L.oracle_avi_simulate.restype = None
    L.oracle_avi_simulate(ctypes.byref(c), ctypes.byref(s), ctypes.c_int64(N), ctypes.c_int32(T), ctypes.c_int32(nx),
                          ctypes.c_int32(nr), ctypes.c_int32(nup), vp(F.ctypes.data), vp(G.ctypes.data), vp(x.ctypes.data),
                          vp(rr.ctypes.data) if rr is not None else None, vp(up.ctypes.data), vp(U.ctypes.data),
                          vp(X.ctypes.data), vp(fm.ctypes.data), ctypes.c_int32(int(bool(warm))))
    return dict(x=x, U=U, X=X, uprev=up[:, :nup], flag_min=fm)


def kkt_residual(H, f, f_theta, A, bu, bl, W, sense, theta, x, active, rho_soft=1e-6):
    """Independent certificate for ONE solved point (full x, n entries): with the rows of `active` on their bounds,
    recover the multipliers by least squares from H x + f(theta) + A_W' mu = 0 and return
    (stationarity residual, worst primal violation over hard rows, worst multiplier sign violation).  SOFT rows may be
    violated: their multiplier is their slack / rho_soft (scaled units), fixed before the least-squares step."""
    H = np.asarray(H, float); n = H.shape[0]
    A = np.asarray(A, float).reshape(-1, n)
    m = np.size(bu); ms = m - A.shape[0]
    Aext = np.vstack([np.eye(n)[:ms], A])
    th = np.asarray(theta, float)
    g = H @ x + np.asarray(f, float) + np.asarray(f_theta, float) @ th
    up = np.array([(int(active[j >> 6]) >> (j & 63)) & 1 for j in range(m)], bool)
    lo = np.array([(int(active[(m + j) >> 6]) >> ((m + j) & 63)) & 1 for j in range(m)], bool)
    rows = [int(j) for j in np.nonzero(up | lo)[0]]
    bsh = np.asarray(W, float).reshape(m, th.size) @ th
    ax = Aext @ x
    bub, blb = np.asarray(bu, float) + bsh, np.asarray(bl, float) + bsh
    sense = np.asarray(sense)
    soft = (sense & 8) != 0
    mu = np.zeros(m)
    # SOFT rows of the working set: the slack IS the multiplier, (row value - bound) * s_j^2 / rho_soft with the
    # row scale s_j = 1 / sqrt(a_j H^-1 a_j')  (rho_soft is measured in the scaled row's units, as in the QP mode)
    Hinv = np.linalg.inv(H) if any(soft[j] for j in rows) else None     # (a semidefinite H has none: only soft rows need it)
    for j in rows:
        if soft[j]:
            s2 = 1.0 / float(Aext[j] @ Hinv @ Aext[j])
            mu[j] = (ax[j] - (bub[j] if up[j] else blb[j])) * s2 / rho_soft
    hard_rows = [j for j in rows if not soft[j]]
    res = g + Aext.T @ mu
    if hard_rows:
        mh = np.linalg.lstsq(Aext[hard_rows].T, -res, rcond=None)[0]
        mu[hard_rows] = mh
        res = res + Aext[hard_rows].T @ mh
    stat = float(np.abs(res).max())
    sign = max([0.0] + [float(-mu[j] if up[j] else mu[j]) for j in rows if not (sense[j] & 4)])
    hard = ~soft & ((sense & 4) == 0)
    viol = np.maximum(ax - bub, blb - ax)
    return stat, float(viol[hard].max() if hard.any() else 0.0), float(sign)
