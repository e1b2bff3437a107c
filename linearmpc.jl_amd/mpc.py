"""Host-side mirror of the reference's operator interface for the online path.

Julia is not available in the build image, so the layer a LinearMPC.jl maintainer would write in
Julia is mirrored here in Python with the reference's names, argument meaning
and error behaviour, so that the parity tests read like test/runtests.jl:

    MPQP                         /root/reference/src/types.jl:75-98   (data contract)
    MPC.setup()                  /root/reference/src/setup.jl:7-29    (`setup!`)
    MPC.form_parameter           /root/reference/src/explicit.jl:54-63
    MPC.solve(theta)             /root/reference/src/utils.jl:268-283
    MPC.compute_control          /root/reference/src/utils.jl:43-51
    MPC.compute_control_trajectory  /root/reference/src/utils.jl:62-70
    MPC.solve_batch / compute_control_batch   -- the new, batched entry points

The condensing step (mpc2mpqp) is NOT part of this package: the mpQP arrives ready-made from the
LinearMPC.jl host (or, in tests, from the oracle's restatement of it).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ._cabi import Settings, default_settings
from .solver import BatchedQP


@dataclass
class MPQP:
    """types.jl:75-98.  Arrays are converted to Julia's column-major layout at the C boundary."""
    H: np.ndarray
    f: np.ndarray
    f_theta: np.ndarray
    A: np.ndarray
    bu: np.ndarray
    bl: np.ndarray
    W: np.ndarray
    senses: np.ndarray
    H_theta: np.ndarray | None = None
    prio: np.ndarray | None = None
    break_points: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    has_binaries: bool = False
    is_symmetric: bool = True


class MPC:
    """The fields of `mutable struct MPC` (types.jl:108-157) that the online path reads."""

    def __init__(self, mpqp: MPQP, nx, nu, nr=0, nd=0, nuprev=0, np_=0, K=None,
                 soft_weight=1e6, device=0):
        self.mpQP = mpqp
        self.nx, self.nu, self.nr, self.nd, self.nuprev, self.np = nx, nu, nr, nd, nuprev, np_
        self.K = np.zeros((nu, nx)) if K is None else np.asarray(K, float).reshape(nu, nx)
        self.uprev = np.zeros(nu)
        self.settings = default_settings()
        self.settings.rho_soft = 1.0 / soft_weight          # setup.jl:26
        self.device = device
        self.mpqp_issetup = False
        self.opt_model: BatchedQP | None = None             # stands where DAQP.Model stands
        self._ctrl_model: BatchedQP | None = None           # nout = nu, K folded in (batched path)

    # setup.jl:7-29
    def setup(self):
        q = self.mpQP
        self.opt_model = BatchedQP.from_mpqp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses,
                                             nout=q.H.shape[0], settings=self.settings,
                                             device=self.device)
        self._ctrl_model = None
        self.mpqp_issetup = True
        return self

    def get_parameter_dims(self):
        return self.nx, self.nr, self.nd, self.nuprev, self.np

    # explicit.jl:54-63 (constant reference / disturbance / parameter; previews are formatted on
    # the LinearMPC.jl host before theta reaches this boundary)
    def form_parameter(self, x, r=None, d=None, uprev=None, p=None):
        x = np.asarray(x, float).reshape(-1)
        if x.size != self.nx:
            raise ValueError(f"State vector must have length {self.nx}")
        r = np.zeros(self.nr) if r is None else np.asarray(r, float).reshape(-1)
        if r.size != self.nr:
            raise ValueError(f"Reference vector length ({r.size}) must match number of outputs ({self.nr})")
        d = np.zeros(self.nd) if d is None else np.asarray(d, float).reshape(-1)
        if d.size != self.nd:
            raise ValueError(f"Disturbance vector must have length {self.nd}")
        up = self.uprev[:self.nuprev] if uprev is None else np.asarray(uprev, float).reshape(-1)[:self.nuprev]
        p = np.zeros(self.np) if p is None else np.asarray(p, float).reshape(-1)
        if p.size != self.np:
            raise ValueError(f"Generalized parameters must have length {self.np}")
        return np.concatenate([x, r, d, up, p])

    # utils.jl:268-283
    def solve(self, theta):
        """-> (xdaqp, fval, exitflag, info) like DAQP.solve; the reference reads x and exitflag only."""
        if not self.mpqp_issetup:
            self.setup()
        theta = np.asarray(theta, float).reshape(-1)
        x, ef, it, act = self.opt_model.solve(theta[None, :])
        q = self.mpQP
        fth = q.f + q.f_theta @ theta
        fval = 0.5 * x[0] @ q.H @ x[0] + fth @ x[0]
        info = {"iterations": int(it[0]), "active": act[0].copy()}
        return x[0].copy(), float(fval), int(ef[0]), info

    # utils.jl:43-51
    def compute_control(self, x, r=None, d=None, uprev=None, p=None, check=True):
        theta = self.form_parameter(x, r, d, uprev, p)
        udaqp, _, exitflag, _ = self.solve(theta)
        if check:
            assert exitflag >= 1, f"solver exit flag {exitflag}"
        self.uprev = udaqp[:self.nu] - self.K @ theta[:self.nx]
        return self.uprev.copy()

    # utils.jl:62-70
    def compute_control_trajectory(self, x, r=None, d=None, uprev=None, p=None, check=True):
        theta = self.form_parameter(x, r, d, uprev, p)
        udaqp, _, exitflag, _ = self.solve(theta)
        if check:
            assert exitflag >= 1, f"solver exit flag {exitflag}"
        self.uprev = udaqp[:self.nu] - self.K @ theta[:self.nx]
        return udaqp

    # ---------------------------------------------------------------- batched entry points
    def control_model(self) -> BatchedQP:
        """Handle whose outputs are the first nu entries of U* minus K x (utils.jl:48-49 folded in)."""
        if self._ctrl_model is None:
            q = self.mpQP
            self._ctrl_model = BatchedQP.from_mpqp(
                q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=self.nu,
                K=self.K if np.any(self.K) else None, nx=self.nx, settings=self.settings,
                device=self.device)
        return self._ctrl_model

    def form_parameter_batch(self, X, R=None, D=None, Uprev=None, P=None):
        X = np.asarray(X, float).reshape(-1, self.nx)
        N = X.shape[0]

        def blk(a, w):
            if w == 0:
                return np.zeros((N, 0))
            if a is None:
                return np.zeros((N, w))
            a = np.asarray(a, float)
            return np.broadcast_to(a.reshape(-1, w) if a.ndim > 1 else a.reshape(1, w), (N, w))

        return np.ascontiguousarray(np.hstack([X, blk(R, self.nr), blk(D, self.nd),
                                               blk(Uprev, self.nuprev), blk(P, self.np)]))

    def solve_batch(self, Theta):
        """Batched `solve`: (X* (N x n), exitflag, iterations, active-set masks)."""
        if not self.mpqp_issetup:
            self.setup()
        return self.opt_model.solve(Theta)

    def compute_control_batch(self, X, R=None, D=None, Uprev=None, P=None, check=True):
        """Batched `compute_control`; stateless (uprev is an input, nothing is stored)."""
        Theta = self.form_parameter_batch(X, R, D, Uprev, P)
        U, ef, _, _ = self.control_model().solve(Theta, want_iters=False, want_active=False)
        if check:
            assert np.all(ef >= 1), f"{int(np.sum(ef < 1))} problems did not solve (min flag {ef.min()})"
        return U, ef
