"""TEST INFRASTRUCTURE ONLY -- numpy restatement of LinearMPC.jl's condensing step.

This module is part of the parity oracle.  Nothing in the shipped package may
import it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

It turns a linear MPC description (model, weights, bounds, general constraints)
into the dense multi-parametric QP

    min_U  1/2 U'HU + (f + f_theta th)'U    s.t.  bl + W th <= [I_ms 0; A] U <= bu + W th

exactly the way the reference builds it on the Julia host, so that the fixtures
fed to the HIP path are the matrices a LinearMPC.jl user would hand over.  Only
the features the benchmark / known-answer problems exercise are restated: reference preview
(mpc2mpqp.jl:535-577, :70-92) and reference condensation (:550-569, traj2setpoint), disturbance preview
(:48-66, :579-604) and measured disturbances as states (:664-669), generalised parameters with preview
(:125-145, :478-508), move blocking (:830-857, setup.jl:202-248 incl. scalar and per-input blocks), constant
offsets in dynamics and outputs (:683-688, :517-530, :393-398), x0-uncertainty tightening (robust.jl:1-29),
binary inputs, the variational (game-theoretic) objective of several players (:900-950, setup.jl:137-151), whose
non-symmetric H makes the problem an affine variational inequality (setup.jl:13 is_avi).  NOT restated:
prioritised constraints / break points beyond the stable sort (:859-866, :890-892), invariant-set terminal
constraints.
Reference lines followed, all under /root/reference/src/:

    zoh                     utils.jl:291-295
    Model(A,B,Ts)           model.jl:78-90
    get_parameter_dims      mpc2mpqp.jl:147-164
    create_extended_system  mpc2mpqp.jl:649-690
    create_extended_cost    mpc2mpqp.jl:692-731
    state_predictor         mpc2mpqp.jl:20-46
    create_objective        mpc2mpqp.jl:407-533
    create_variational_objective  mpc2mpqp.jl:900-950   (set_objective!(mpc, uids; ...) setup.jl:137-151)
    create_controlbounds    mpc2mpqp.jl:206-245
    create_general_constraints mpc2mpqp.jl:249-354
    create_constraints      mpc2mpqp.jl:358-402
    remove_redundant        mpc2mpqp.jl:733-773
    remove_duplicate        mpc2mpqp.jl:775-828
    MPQP(obj,constraints)   mpc2mpqp.jl:868-899
    example problems        mpc_examples.jl:104-141 (invpend), :241-286 (mass_spring),
                            README.md:39-52 (pendulum on a cart), docs/src/manual/simple.md:60-83
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
from scipy.linalg import expm, block_diag

# DAQP constraint-sense bit flags ([EXT] DAQPBase constants used at mpc2mpqp.jl:875-884)
ACTIVE, LOWER, IMMUTABLE, SOFT, BINARY = 1, 2, 4, 8, 16
EQUALITY = ACTIVE + IMMUTABLE


def zoh(A, B, Ts):
    """utils.jl:291-295 -- exact zero-order-hold discretisation via one matrix exponential."""
    A = np.atleast_2d(np.asarray(A, float))
    B = np.asarray(B, float).reshape(A.shape[0], -1)
    nx, nu = B.shape
    blk = np.zeros((nx + nu, nx + nu))
    blk[:nx, :nx] = A * Ts
    blk[:nx, nx:] = B * Ts
    E = expm(blk)
    return E[:nx, :nx].copy(), E[:nx, nx:].copy()


def _as_weight(w, n):
    """utils.jl:297-299 matrixify: scalar -> w*I, vector -> diag, matrix -> itself."""
    w = np.asarray(w, float)
    if w.ndim == 0:
        return np.eye(n) * float(w)
    if w.ndim == 1:
        return np.diag(w)
    return w.copy()


@dataclass
class GeneralConstraint:
    """types.jl:4-18 -- lb <= Ax x_k + Au u_k <= ub for the (1-based) time steps in ks."""
    Ax: np.ndarray
    Au: np.ndarray
    lb: np.ndarray
    ub: np.ndarray
    ks: Sequence[int]
    soft: bool = False
    prio: int = 0
    Ap: Optional[np.ndarray] = None      # generalised-parameter term  + Ap p_k  (types.jl:11)


@dataclass
class MPCProblem:
    """The subset of `mutable struct MPC` (types.jl:108-157) that the restated path reads."""
    F: np.ndarray
    G: np.ndarray
    C: np.ndarray
    Np: int
    Nc: int
    Q: np.ndarray
    R: np.ndarray
    Rr: np.ndarray
    umin: np.ndarray = field(default_factory=lambda: np.zeros(0))
    umax: np.ndarray = field(default_factory=lambda: np.zeros(0))
    constraints: List[GeneralConstraint] = field(default_factory=list)
    reference_tracking: bool = True
    preprocess: bool = True
    Ts: float = -1.0
    K: Optional[np.ndarray] = None          # prestabilising feedback u = v - K x (setup.jl:186-199)
    Eu: Optional[np.ndarray] = None         # affine input cost (Eu p + eu)'u_k  (setup.jl:136-150)
    eu: Optional[np.ndarray] = None
    reference_preview: bool = False         # settings.reference_preview (types.jl:56,67): r is ny x Np in theta
    reference_condensation: bool = False    # settings.reference_condensation (types.jl:54,65): the trajectory is
    traj2setpoint: Optional[np.ndarray] = None   # collapsed to one setpoint by traj2setpoint (set by mpc2mpqp)
    f_offset: Optional[np.ndarray] = None   # x+ = F x + G u + f_offset   (model.jl:20, setup.jl:516-531)
    h_offset: Optional[np.ndarray] = None   # y  = C x + h_offset         (model.jl:30)
    move_blocks: Optional[list] = None      # per input: block lengths (setup.jl:202-248)
    x0_uncertainty: Optional[np.ndarray] = None   # mpc.dx0 (setup.jl:293-296): constraints tightened by |Ax| dx0
    parameter_preview: bool = False         # settings.parameter_preview (types.jl:58,69): p is np_base x Np in theta
    disturbance_preview: bool = False       # settings.disturbance_preview (types.jl:57,68): d is nd x Np in theta
    Gd: Optional[np.ndarray] = None         # measured disturbance: x+ = F x + G u + Gd d (model.jl:17,70)
    Dd: Optional[np.ndarray] = None         #                       y  = C x + Dd d       (model.jl:28)
    binary_controls: Sequence[int] = ()     # 0-based inputs restricted to {umin, umax} (setup.jl:277-281)
    Nc_binary: int = -1                     # "binary control horizon" (-1 = whole control horizon)
    objectives: list = field(default_factory=list)   # mpc.objectives (types.jl:159): one (weights, uids) per player

    def add_objective(self, uids, Q=None, R=None, Rr=None, S=None, Qf=None):
        """setup.jl:137-151 set_objective!(mpc, uids; ...): the cost of the player that owns inputs `uids` (0-based
        here).  Unset weights are zero (not the MPC's defaults), Qf defaults to Q; mpc.weights.Rr[uids,uids] is
        set too so that get_parameter_dims keeps u_prev in theta (:149)."""
        uids = [int(u) for u in uids]
        nui = len(uids)
        w = {"Q": np.zeros((self.ny, self.ny)) if Q is None else _as_weight(Q, self.ny),
             "R": np.zeros((nui, nui)) if R is None else _as_weight(R, nui),
             "Rr": np.zeros((nui, nui)) if Rr is None else _as_weight(Rr, nui),
             "S": np.zeros((self.nx, nui)) if S is None else np.asarray(S, float).reshape(self.nx, nui)}
        w["Qf"] = w["Q"].copy() if Qf is None else _as_weight(Qf, self.ny)
        self.Rr[np.ix_(uids, uids)] = w["Rr"]
        self.objectives.append((w, uids))
        return self

    def np_base(self):
        """utils.jl:207-216 get_affine_parameter_base_dim (largest column count among Eu / Ap)."""
        dims = [0 if self.Eu is None else np.atleast_2d(self.Eu).shape[1]]
        dims += [0 if c.Ap is None else np.atleast_2d(c.Ap).shape[1] for c in self.constraints]
        return max(dims)

    @property
    def nd(self):
        return 0 if self.Gd is None else np.atleast_2d(self.Gd).shape[1]

    def has_f_offset(self):
        return self.f_offset is not None and np.any(np.asarray(self.f_offset) != 0)

    def set_offset(self, xo=None, uo=None, fo=None, ho=None):
        """setup.jl:516-531 set_offset!: f_offset = fo - F xo - G uo, h_offset = ho - C xo."""
        xo = np.zeros(self.nx) if xo is None else np.asarray(xo, float)
        uo = np.zeros(self.nu) if uo is None else np.asarray(uo, float)
        fo = np.zeros(self.nx) if fo is None else np.asarray(fo, float)
        ho = np.zeros(self.ny) if ho is None else np.asarray(ho, float)
        self.f_offset = fo - self.F @ xo - self.G @ uo
        self.h_offset = ho - self.C @ xo
        self.uprev0 = uo.copy()                  # mpc.uprev .= uo
        return self

    @staticmethod
    def _format_move_block(block, Np):
        """setup.jl:236-248 format_move_block: padded or clipped so that the blocks add up to Np."""
        block = [int(b) for b in block]
        if not block:
            return []
        tot = sum(block)
        if tot < Np:
            block[-1] += Np - tot
        elif tot > Np:
            acc, i = 0, 0
            while True:
                acc += block[i]
                if acc >= Np:
                    break
                i += 1
            block = block[:i + 1]
            block[-1] += Np - acc
        return block

    def move_block(self, block):
        """setup.jl:202-234 move_block!: nothing / empty -> no blocking (Nc = Np); a number -> constant block
        size (Np // block + 1 blocks, then clipped); one vector -> the same blocks for every input; a list of
        vectors -> one per input.  Nc = (largest sum of all blocks but the last) + 1."""
        if block is None or (not np.isscalar(block) and len(block) == 0):
            self.move_blocks = []
            self.Nc = self.Np
            return self
        if np.isscalar(block):
            block = [] if block <= 0 else [int(block)] * (self.Np // int(block) + 1)
            return self.move_block(block)
        if np.isscalar(block[0]):
            block = [list(block) for _ in range(self.nu)]
        if len(block) != self.nu:
            raise ValueError("Need to have blocks for every control input")
        self.move_blocks = [self._format_move_block(mb, self.Np) for mb in block]
        self.Nc = max(sum(mb[:-1]) for mb in self.move_blocks) + 1
        return self

    def gain(self):
        return np.zeros((self.nu, self.nx)) if self.K is None else np.asarray(self.K, float).reshape(self.nu, self.nx)

    def set_prestabilizing_feedback(self, K=None):
        """setup.jl:186-199: given K, or the infinite-horizon LQR gain from the discrete Riccati
        equation with weights C'QC and R+Rr (`ared`)."""
        if K is None:
            from scipy.linalg import solve_discrete_are
            Qx = self.C.T @ self.Q @ self.C
            Ru = self.R + self.Rr
            Pm = solve_discrete_are(self.F, self.G, Qx, Ru)
            K = np.linalg.solve(Ru + self.G.T @ Pm @ self.G, self.G.T @ Pm @ self.F)
        self.K = np.asarray(K, float).reshape(self.nu, self.nx)
        return self

    @property
    def nx(self):
        return self.F.shape[0]

    @property
    def nu(self):
        return self.G.shape[1]

    @property
    def ny(self):
        return self.C.shape[0]

    # mpc2mpqp.jl:147-164
    def parameter_dims(self):
        nr = self.ny if self.reference_tracking else 0
        if self.reference_preview and not self.reference_condensation and nr > 0:
            nr = nr * self.Np                    # mpc2mpqp.jl:154-156: one reference per predicted step
        nuprev = self.nu if np.any(self.Rr != 0) else 0
        nd = self.nd * self.Np if (self.disturbance_preview and self.nd > 0) else self.nd     # :157-160
        npp = self.np_base() * (self.Np if self.parameter_preview else 1)                     # :162
        return self.nx, nr, nd, nuprev, npp

    def add_constraint(self, Ax=None, Au=None, lb=(), ub=(), ks=None, soft=False, prio=0, Ap=None):
        """setup.jl:57-79 add_constraint! (default ks = 2:Np, missing side = +-1e30)."""
        lb = np.atleast_1d(np.asarray(lb, float))
        ub = np.atleast_1d(np.asarray(ub, float))
        m = max(lb.size, ub.size)
        if m == 0 or (Ax is None and Au is None):
            return
        ub = np.concatenate([ub, 1e30 * np.ones(m - ub.size)])
        lb = np.concatenate([lb, -1e30 * np.ones(m - lb.size)])
        Ax = np.zeros((m, self.nx)) if Ax is None else np.atleast_2d(np.asarray(Ax, float))
        Au = np.zeros((m, self.nu)) if Au is None else np.atleast_2d(np.asarray(Au, float))
        ks = list(range(2, self.Np + 1)) if ks is None else list(ks)
        Ap = None if Ap is None else np.atleast_2d(np.asarray(Ap, float))
        self.constraints.append(GeneralConstraint(Ax, Au, lb, ub, ks, soft, prio, Ap))


def make_mpc(F, G, C=None, Np=10, Nc=None, Q=None, R=None, Rr=None, umin=(), umax=(),
             reference_tracking=True, Ts=-1.0, Gd=None, Dd=None):
    """MPC(F,G;...) + set_objective! + set_bounds! (types.jl:159-172, setup.jl:36-46,:136-150).

    Default weights follow MPCWeights(nu,nx,nr) (types.jl:34-37): Q=I_ny, R=I_nu, Rr=0.
    """
    F = np.atleast_2d(np.asarray(F, float))
    G = np.asarray(G, float).reshape(F.shape[0], -1)
    nx, nu = G.shape
    C = np.eye(nx) if C is None else np.atleast_2d(np.asarray(C, float))
    ny = C.shape[0]
    Nc = Np if Nc is None else Nc
    Q = np.eye(ny) if Q is None else _as_weight(Q, ny)
    R = np.eye(nu) if R is None else _as_weight(R, nu)
    Rr = np.zeros((nu, nu)) if Rr is None else _as_weight(Rr, nu)
    p = MPCProblem(F, G, C, Np, Nc, Q, R, Rr,
                   np.atleast_1d(np.asarray(umin, float)), np.atleast_1d(np.asarray(umax, float)),
                   [], reference_tracking, True, Ts)
    if Gd is not None:
        p.Gd = np.asarray(Gd, float).reshape(nx, -1)
        p.Dd = np.zeros((ny, p.Gd.shape[1])) if Dd is None else np.asarray(Dd, float).reshape(ny, -1)
    return p


# --------------------------------------------------------------------------- prediction
def state_predictor(F, G, Np, Nc):
    """mpc2mpqp.jl:20-46.  X = Phi x0 + Gam U with u_k = u_Nc held for k > Nc."""
    nx, nu = G.shape
    Gam = np.zeros(((Np + 1) * nx, Nc * nu))
    Phi = np.zeros(((Np + 1) * nx, nx))
    Phi[:nx] = np.eye(nx)
    Fpow, FG = F.copy(), G.copy()
    for i in range(1, Nc + 1):
        for j in range(0, Nc - i + 1):
            Gam[(i + j) * nx:(i + j + 1) * nx, j * nu:(j + 1) * nu] = FG
        Phi[i * nx:(i + 1) * nx] = Fpow
        if i == Nc:
            break
        Fpow = Fpow @ F
        FG = F @ FG
    for i in range(Nc + 1, Np + 1):
        Gam[i * nx:(i + 1) * nx] = F @ Gam[(i - 1) * nx:i * nx]
        Gam[i * nx:(i + 1) * nx, -nu:] += G
        Phi[i * nx:(i + 1) * nx] = F @ Phi[(i - 1) * nx:i * nx]
    return Phi, Gam


def extended_system(p: MPCProblem):
    """mpc2mpqp.jl:649-690 (no disturbance preview)."""
    nx, nr, nd, nuprev, _ = p.parameter_dims()
    nu, ny = p.nu, p.ny
    K = p.gain()
    F, G, C = p.F - p.G @ K, p.G.copy(), p.C.copy()
    if nr > 0 and not p.reference_preview:       # reference rides along as constant states (:656-662;
        F = block_diag(F, np.eye(ny))            # with preview it is no state, see ref_preview_cost)
        G = np.vstack([G, np.zeros((ny, nu))])
        C = np.hstack([C, -np.eye(ny)])
    if nd > 0 and not p.disturbance_preview:     # measured disturbance as constant states (:664-669)
        F = block_diag(F, np.eye(nd))
        F[:nx, -nd:] = np.asarray(p.Gd, float).reshape(nx, nd)
        G = np.vstack([G, np.zeros((nd, nu))])
        Dd = np.zeros((ny, nd)) if p.Dd is None else np.asarray(p.Dd, float).reshape(ny, nd)
        C = np.hstack([C, Dd])
    if nuprev > 0:                               # previous input as a state, du as an output
        F = block_diag(F, np.zeros((nu, nu)))
        F[-nu:, :nx] = -K
        G = np.vstack([G, np.eye(nu)])
        nye, nxe = C.shape
        C = np.vstack([np.hstack([C, np.zeros((nye, nu))]),
                       np.hstack([K, np.zeros((nu, nxe - nx)), np.eye(nu)])])
    if np.any(p.R != 0) and np.any(K != 0):      # u'Ru with u = v - Kx: K x becomes an output (:679-681)
        C = np.vstack([C, np.hstack([K, np.zeros((nu, C.shape[1] - nx))])])
    if p.has_f_offset():                         # constant 1 as the last state (:683-688)
        F = block_diag(F, np.eye(1))
        F[:nx, -1] = np.asarray(p.f_offset, float)
        G = np.vstack([G, np.zeros((1, nu))])
        C = np.hstack([C, np.zeros((C.shape[0], 1))])
    return F, G, C


def extended_cost(p: MPCProblem):
    """mpc2mpqp.jl:692-731.  Returns Q, R, S, Qf of the extended system."""
    nx, nr, nd, nuprev, _ = p.parameter_dims()
    nu, ny = p.nu, p.ny
    K = p.gain()
    Q, R, Rr = p.Q.copy(), p.R.copy(), p.Rr.copy()
    Qf = Q.copy()                                # Qf, Qfx unset => terminal weight = Q (:694)
    S = np.zeros((nx, nu))
    if nr > 0 and not p.reference_preview:       # :701-703
        S = np.vstack([S, np.zeros((ny, nu))])
    if nd > 0 and not p.disturbance_preview:     # :705-707
        S = np.vstack([S, np.zeros((nd, nu))])
    if nuprev > 0:
        Q = block_diag(Q, Rr)
        Qf = block_diag(Qf, np.zeros((nu, nu)))
        S = np.vstack([S, -Rr])
        S[:nx] -= K.T @ Rr
        R = R + Rr
    if np.any(R != 0) and np.any(K != 0):        # :718-724
        Q = block_diag(Q, p.R)
        Qf = block_diag(Qf, np.zeros((nu, nu)))
        S[:nx] -= K.T @ p.R
    if p.has_f_offset():                         # :726-728
        S = np.vstack([S, np.zeros((1, nu))])
    return Q, R, S, Qf


def dense_objective(p: MPCProblem, F, Phi, Gam, C, Q, R, S, Qf):
    """mpc2mpqp.jl:407-533 (branches for preview / offsets / binaries / affine params unused)."""
    N, Nc = p.Np, p.Nc
    nxe, nue = Gam.shape[0] // (N + 1), R.shape[0]
    posQ = np.flatnonzero(np.diag(Q) > 0)
    Qp, Cp = Q[np.ix_(posQ, posQ)], C[posQ]
    posQf = np.flatnonzero(np.diag(Qf) > 0)
    Qfp, Cf = Qf[np.ix_(posQf, posQf)], C[posQf]

    H = np.kron(np.eye(Nc), R)
    H[-nue:, -nue:] += (N - Nc) * R              # held input after the control horizon
    CQC = np.kron(np.eye(N + 1), Cp.T @ Qp @ Cp)
    CQC[-nxe:, -nxe:] = Cf.T @ Qfp @ Cf
    H = H + Gam.T @ CQC @ Gam
    f_theta = Gam.T @ CQC @ Phi
    H_theta = Phi.T @ CQC @ Phi
    if np.any(S != 0):
        Stot = np.vstack([np.kron(np.eye(Nc), S), np.zeros(((N - Nc + 1) * nxe, Nc * nue))])
        Stot[Nc * nxe:N * nxe, -nue:] = np.tile(S, (N - Nc, 1))
        GS = Gam.T @ Stot
        H = H + GS + GS.T
        f_theta = f_theta + Stot.T @ Phi
    if p.reference_tracking and p.reference_preview:
        # ref_preview_cost (mpc2mpqp.jl:535-577, no condensation): (C x_k - r_k)'Q(C x_k - r_k) with a
        # reference per step; the cross term puts Fr = -Gam' kron(I, C'Q) (first block dropped: the
        # reference at k = 0 meets no input) into f_theta right behind the state columns
        ny, nxp = p.ny, p.nx
        C_full, Q_full, Qf_full = C[:ny], Q[:ny, :ny], Qf[:ny, :ny]
        CQ = np.kron(np.eye(N + 1), C_full.T @ Q_full)
        CQ[-C_full.shape[1]:, -ny:] = C_full.T @ Qf_full
        Fr = (-Gam.T @ CQ)[:, ny:]
        Hr = np.kron(np.eye(N), Q_full)
        Hr[-ny:, -ny:] = Qf_full
        nrp = ny * N
        if p.reference_condensation:
            # mpc2mpqp.jl:550-569: the trajectory enters through ONE setpoint s = traj2setpoint r_traj,
            # chosen so that W H^-1 Fr Is s is the least-squares match of W H^-1 Fr r_traj (W weights the
            # first control move 1e6: its accuracy matters most); Is repeats the setpoint over the horizon
            Is = np.tile(np.eye(ny), (N, 1))
            Wc = np.eye(H.shape[0])
            Wc[np.arange(p.nu), np.arange(p.nu)] = 1e6
            WinvHFr = Wc @ np.linalg.solve(H, Fr)
            p.traj2setpoint = np.linalg.lstsq(WinvHFr @ Is, WinvHFr, rcond=None)[0]
            Fr = Fr @ Is
            Hr = Is.T @ Hr @ Is
            nrp = ny
        f_theta = np.hstack([f_theta[:, :nxp], Fr, f_theta[:, nxp:]])
        tail = H_theta.shape[0] - nxp
        H_theta = np.block([[H_theta[:nxp, :nxp], np.zeros((nxp, nrp)), H_theta[:nxp, nxp:]],
                            [np.zeros((nrp, nxp)), Hr, np.zeros((nrp, tail))],
                            [H_theta[nxp:, :nxp], np.zeros((tail, nrp)), H_theta[nxp:, nxp:]]])
    if p.disturbance_preview and p.nd > 0:
        # disturbance_preview_cost (mpc2mpqp.jl:579-604): d_k enters the predicted states through
        # Psi (disturbance_predictor, :48-58: x_k collects Gd d_0 .. Gd d_{k-1}) and the outputs through Dd;
        # the cross term with U puts Fd = (C Gam)' Qy Yd into f_theta behind the state and reference columns
        ny, nxp = p.ny, p.nx
        nd0 = p.nd
        nrp = p.parameter_dims()[1]
        C_full, Q_full, Qf_full = C[:ny], Q[:ny, :ny], Qf[:ny, :ny]
        E = np.vstack([np.asarray(p.Gd, float).reshape(nxp, nd0), np.zeros((nxe - nxp, nd0))])
        Psi = np.zeros(((N + 1) * nxe, N * nd0))
        for k in range(1, N + 1):
            Psi[k * nxe:(k + 1) * nxe] = F @ Psi[(k - 1) * nxe:k * nxe]
            Psi[k * nxe:(k + 1) * nxe, (k - 1) * nd0:k * nd0] += E
        CY = np.kron(np.eye(N), C_full)
        Dd = np.zeros((ny, nd0)) if p.Dd is None else np.asarray(p.Dd, float).reshape(ny, nd0)
        Gy = CY @ Gam[nxe:]
        Yd = CY @ Psi[nxe:] + np.kron(np.eye(N), Dd)
        Qy = np.kron(np.eye(N), Q_full)
        Qy[-ny:, -ny:] = Qf_full
        Fd, Hd = Gy.T @ Qy @ Yd, Yd.T @ Qy @ Yd
        split = nxp + nrp
        tail = H_theta.shape[0] - split
        ndp = N * nd0
        f_theta = np.hstack([f_theta[:, :split], Fd, f_theta[:, split:]])
        H_theta = np.block([[H_theta[:split, :split], np.zeros((split, ndp)), H_theta[:split, split:]],
                            [np.zeros((ndp, split)), Hd, np.zeros((ndp, tail))],
                            [H_theta[split:, :split], np.zeros((tail, ndp)), H_theta[split:, split:]]])
    f = np.zeros(H.shape[0])
    # generalised-parameter cost on the inputs (mpc2mpqp.jl:478-508): f += Umap' eu, f_theta gets
    # one column block Umap' (Eu stacked) for p (constant over the horizon, no preview)
    npb = p.np_base()
    Umap = np.kron(np.vstack([np.eye(Nc), np.zeros((N - Nc, Nc))]), np.eye(p.nu))
    if p.eu is not None:
        f = f + Umap.T @ np.tile(np.asarray(p.eu, float).reshape(p.nu), N)
    if npb > 0:
        Eu = np.zeros((p.nu, npb)) if p.Eu is None else np.atleast_2d(np.asarray(p.Eu, float))
        # stage_parameter_matrix (:145): one parameter for the whole horizon, or one per predicted step
        stage = np.kron(np.eye(N), Eu) if p.parameter_preview else np.tile(Eu, (N, 1))
        Fp = Umap.T @ stage
        npp = stage.shape[1]
        f_theta = np.hstack([f_theta, Fp])
        nthc = H_theta.shape[0]
        H_theta = np.block([[H_theta, np.zeros((nthc, npp))], [np.zeros((npp, nthc)), np.zeros((npp, npp))]])
    # regularisation of binary inputs (mpc2mpqp.jl:510-515): u^2 - (umin+umax) u is constant on
    # {umin, umax}, so it does not move the optimum but keeps H positive definite when R = 0
    if len(p.binary_controls):
        fb = np.zeros(p.nu)
        bc = list(p.binary_controls)
        fb[bc] = (p.umax[bc] + p.umin[bc]) / 2
        fbin = np.tile(fb, Nc)
        f = f - fbin
        H = H + np.diag((fbin != 0).astype(float))
    if p.has_f_offset():                         # collapse the constant state into f (:517-521)
        assert npb == 0
        f = f + f_theta[:, -1]
        f_theta = f_theta[:, :-1]
        H_theta = H_theta[:-1, :-1]
    if p.reference_tracking and p.h_offset is not None and np.any(p.h_offset):   # r - h_offset (:523-530)
        nrp = p.parameter_dims()[1]
        ho = np.tile(p.h_offset, N) if (p.reference_preview and not p.reference_condensation) \
            else np.asarray(p.h_offset, float)
        f = f - f_theta[:, p.nx:p.nx + nrp] @ ho
    return (H + H.T) / 2, f, f_theta, H_theta


def extended_cost_player(p: MPCProblem, w, uids):
    """mpc2mpqp.jl:692-731 create_extended_cost(mpc, weights; uids): one player's Q, R, S, Qf on the extended
    system; S has one column per input of the player."""
    nx, nr, nd, nuprev, _ = p.parameter_dims()
    nu, ny, nui = p.nu, p.ny, len(uids)
    K = p.gain()
    Q, R, Rr, S = w["Q"].copy(), w["R"].copy(), w["Rr"].copy(), w["S"].copy()
    Qf = Q.copy() if not np.any(w["Qf"]) else w["Qf"].copy()                  # :694 (Qfx is not restated)
    if nr > 0 and not p.reference_preview:
        S = np.vstack([S, np.zeros((ny, nui))])
    if p.nd > 0 and not p.disturbance_preview:
        S = np.vstack([S, np.zeros((p.nd, nui))])
    if nuprev > 0:
        Rrfull = np.zeros((nu, nu))
        Rrfull[np.ix_(uids, uids)] = Rr
        Q = block_diag(Q, Rrfull)
        Qf = block_diag(Qf, np.zeros((nu, nu)))
        S = np.vstack([S, -Rrfull[:, uids]])
        S[:nx] -= K[uids].T @ Rr
        R = R + Rr
    if np.any(R != 0) and np.any(K != 0):
        Rfull = np.zeros((nu, nu))
        Rfull[np.ix_(uids, uids)] = w["R"]
        Q = block_diag(Q, Rfull)
        Qf = block_diag(Qf, np.zeros((nu, nu)))
        S[:nx] -= K[uids].T @ w["R"]
    if p.has_f_offset():
        S = np.vstack([S, np.zeros((1, nui))])
    return Q, R, S, Qf


def variational_objective(p: MPCProblem, Phi, Gam, C):
    """mpc2mpqp.jl:900-950 create_variational_objective: block row i of H is the gradient of player i's cost with
    respect to its OWN inputs, Gam_i' CQC_i [Gam_1 ... Gam_P] (+ R_i on its own block) -- not symmetric unless the
    players' costs agree; f = 0, f_theta row block i = Gam_i' CQC_i Phi (+ S terms)."""
    N, Nc, nu = p.Np, p.Nc, p.nu
    ws = [extended_cost_player(p, w, uids) for w, uids in p.objectives]
    uids = [u for _, u in p.objectives]
    flat = sorted(u for us in uids for u in us)
    if flat != list(range(nu)):
        raise ValueError("The controls have to be fully partitioned")
    nU, nxe = Gam.shape[1], Phi.shape[1]
    # (Julia: vec(uids .+ (0:nu:nU-nu)') -- all inputs of the player at step 0, then step 1, ...)
    Uids = [np.asarray([u + k for k in range(0, nU - nu + 1, nu) for u in us]) for us in uids]
    Gs = [Gam[:, U] for U in Uids]
    H = np.zeros((nU, nU))
    f_theta = np.zeros((nU, nxe))
    for i, (Q, R, S, Qf) in enumerate(ws):
        nui = len(uids[i])
        CQC = block_diag(np.kron(np.eye(N), C.T @ Q @ C), C.T @ Qf @ C)
        for j in range(len(ws)):
            H[np.ix_(Uids[i], Uids[j])] = Gs[i].T @ CQC @ Gs[j]
            if i == j:
                H[np.ix_(Uids[i], Uids[i])] += np.kron(np.eye(Nc), R)
                last = Uids[i][-nui:]
                H[np.ix_(last, last)] += (N - Nc) * R
        f_theta[Uids[i]] = Gs[i].T @ CQC @ Phi
        Stot = np.vstack([np.kron(np.eye(Nc), S), np.zeros(((N - Nc + 1) * nxe, Nc * nui))])
        Stot[Nc * nxe:N * nxe, -nui:] = np.tile(S, (N - Nc, 1))
        GS = Gs[i].T @ Stot
        H[np.ix_(Uids[i], Uids[i])] += GS + GS.T
        f_theta[Uids[i]] += Stot.T @ Phi
    return H, np.zeros(nU), f_theta, np.zeros((0, 0))


def apply_move_block(p, H, f, f_theta, cons):
    """mpc2mpqp.jl:830-857: U = T V with one column of T per block; bounds of dropped moves go."""
    A, bu, bl, W, soft, prio = cons
    nu, Nc = p.nu, p.Nc
    nub = p.umax.size
    nUold, nUnew = nu * Nc, sum(len(mb) for mb in p.move_blocks)
    T = np.zeros((nUold, nUnew))
    counter = list(range(nu))
    keep, new_id = [], 0
    for ps in range(max(len(mb) for mb in p.move_blocks)):
        for iu, mb in enumerate(p.move_blocks):
            if len(mb) <= ps:
                continue
            block = mb[ps] if len(mb) != ps + 1 else 1
            T[counter[iu]:counter[iu] + nu * (block - 1) + 1:nu, new_id] = 1
            if counter[iu] < nub * Nc:
                keep.append(counter[iu])
            counter[iu] += nu * block
            new_id += 1
    keep = keep + list(range(nub * Nc, bu.size))
    K = p.gain()
    Anew = (A[keep] @ T) if np.any(K != 0) else (A @ T)
    binary = getattr(p, "_binary_rows", np.zeros(0, bool))
    if binary.size:
        p._binary_rows = binary[keep]
    return T.T @ H @ T, T.T @ f, T.T @ f_theta, (Anew, bu[keep], bl[keep], W[keep], soft[keep], prio[keep])


def dense_constraints(p: MPCProblem, Phi, Gam):
    """mpc2mpqp.jl:358-402 -> (A, bu, bl, W, issoft, prio); simple bounds first."""
    nx, nr, nd, nuprev, npb = p.parameter_dims()
    nu, Np, Nc = p.nu, p.Np, p.Nc
    prev = p.reference_preview and nr > 0
    dprev = p.disturbance_preview and nd > 0
    nxe = nx + (0 if prev else nr) + (0 if dprev else nd) + nuprev   # previewed blocks are no states (:210-211)
    if p.has_f_offset():
        nxe += 1                                     # constant offset in the dynamics (:212)
    n = Gam.shape[1]

    def with_ref_block(Wm, Wd=None):                 # insert_preview_parameter_blocks (:70-92), Wr = 0
        if prev:
            Wm = np.hstack([Wm[:, :nx], np.zeros((Wm.shape[0], nr)), Wm[:, nx:]])
        if dprev:                                    # the disturbance block sits behind state and reference
            at = nx + nr
            Wdd = np.zeros((Wm.shape[0], nd)) if Wd is None else Wd
            Wm = np.hstack([Wm[:, :at], Wdd, Wm[:, at:]])
        return Wm

    if p.umax.size:
        # create_controlbounds (:206-245): with K = 0 simple bounds on U, otherwise general rows
        # (I - Kfull Gam) V <= b + Kfull Phi x0  for u_k = v_k - K x_k
        K = p.gain()
        bu, bl = np.tile(p.umax, Nc), np.tile(p.umin, Nc)
        if np.any(K != 0):
            Kfull = np.kron(np.eye(Nc), np.hstack([K, np.zeros((nu, nxe - nx))]))
            A = np.eye(Nc * nu) - Kfull @ Gam[:Nc * nxe, :Nc * nu]
            W = Kfull @ Phi[:Nc * nxe]
        else:
            A = np.zeros((0, n))
            W = np.zeros((Nc * nu, nxe))
        W = with_ref_block(W)
        soft = np.zeros(n, bool)
        prio = np.zeros(n, int)
        if npb > 0:
            W = np.hstack([W, np.zeros((W.shape[0], npb))])
        # binary controls (mpc2mpqp.jl:368-373): flag their bound rows over the binary horizon
        single = np.zeros(nu, bool)
        single[list(p.binary_controls)] = True
        binary = np.tile(single, Nc)
        if p.Nc_binary >= 0:
            binary[p.Nc_binary * nu:] = False
    else:
        A, bu, bl = np.zeros((0, n)), np.zeros(0), np.zeros(0)
        W, soft, prio = np.zeros((0, nxe + (nr if prev else 0) + (nd if dprev else 0) + npb)), np.zeros(0, bool), np.zeros(0, int)
        binary = np.zeros(0, bool)
    if p.constraints:
        eyeX = np.eye(Np + 1)
        eyeU = np.vstack([np.eye(Nc), np.zeros((1 + Np - Nc, Nc))])
        Ax_rows, Au_rows, ubs, lbs, softs, prios, Wp_rows = [], [], [], [], [], [], []
        for c in p.constraints:
            mi = c.Au.shape[0]
            kmax = Np + 1 if not np.any(c.Au) else Np
            ks = [k for k in c.ks if k <= kmax]
            sel = [k - 1 for k in ks]
            pad = np.zeros((mi, nxe - nx))
            Au_rows.append(np.kron(eyeU[sel], c.Au))
            Ax_rows.append(np.kron(eyeX[sel], np.hstack([c.Ax - c.Au @ p.gain(), pad])))
            ubk, lbk = np.tile(c.ub, len(ks)), np.tile(c.lb, len(ks))
            if p.x0_uncertainty is not None and np.any(p.x0_uncertainty):
                # constraint_tightening (robust.jl:1-29) with wmin = wmax = 0: every row loses
                # sum_j |Ax_ij dx0_j| on both sides, from k = 2 on (mpc2mpqp.jl:298-305)
                acc = np.abs((c.Ax - c.Au @ p.gain()) * np.asarray(p.x0_uncertainty, float)[None, :]).sum(axis=1)
                t = np.concatenate([acc if k >= 2 else np.zeros(mi) for k in ks])
                ubk, lbk = ubk - t, lbk + t
            ubs.append(ubk)
            lbs.append(lbk)
            if npb > 0:                          # parameter_preview_direct (:125-143): W[:, p] = -Ap
                nb0 = p.np_base()
                Ap = np.zeros((mi, nb0)) if c.Ap is None else c.Ap
                if p.parameter_preview:          # stage k reads the parameter of its own step (held at Np)
                    Wp = np.zeros((mi * len(ks), nb0 * Np))
                    for i_, k_ in enumerate(ks):
                        col = min(k_, Np) - 1
                        Wp[i_ * mi:(i_ + 1) * mi, col * nb0:(col + 1) * nb0] = -Ap
                    Wp_rows.append(Wp)
                else:
                    Wp_rows.append(np.tile(-Ap, (len(ks), 1)))
            softs.append(np.full(mi * len(ks), c.soft))
            prios.append(np.full(mi * len(ks), c.prio, int))
        Axt, Aut = np.vstack(Ax_rows), np.vstack(Au_rows)
        A = np.vstack([A, Axt @ Gam + Aut])
        Wd = None
        if dprev:                                # :348: Wd = -Axtot Psi (no direct Ad terms here)
            nd0 = p.nd
            Fx = extended_system(p)[0]
            E = np.vstack([np.asarray(p.Gd, float).reshape(nx, nd0), np.zeros((nxe - nx, nd0))])
            Psi = np.zeros(((Np + 1) * nxe, Np * nd0))
            for k in range(1, Np + 1):
                Psi[k * nxe:(k + 1) * nxe] = Fx @ Psi[(k - 1) * nxe:k * nxe]
                Psi[k * nxe:(k + 1) * nxe, (k - 1) * nd0:k * nd0] += E
            Wd = -Axt @ Psi
        Wg = with_ref_block(-Axt @ Phi, Wd)
        if npb > 0:
            Wg = np.hstack([Wg, np.vstack(Wp_rows)])
        W = np.vstack([W, Wg])
        bu = np.concatenate([bu] + ubs)
        bl = np.concatenate([bl] + lbs)
        soft = np.concatenate([soft] + softs)
        prio = np.concatenate([prio] + prios)
        binary = np.concatenate([binary, np.zeros(sum(len(x) for x in softs), bool)])
    p._binary_rows = binary                      # carried next to the constraint tuple (no row is dropped
    if p.has_f_offset() and W.shape[0]:          # or reordered among the simple bounds afterwards)
        col = nxe - 1 + (nr if prev else 0) + (nd if dprev else 0)   # collapse the constant state into the bounds (:393-398)
        bu, bl = bu + W[:, col], bl + W[:, col]
        W = np.delete(W, col, axis=1)
    elif p.has_f_offset():
        W = np.zeros((0, W.shape[1] - 1))
    return A, bu, bl, W, soft, prio


def sort_by_priority(A, bu, bl, W, soft, prio):
    """mpc2mpqp.jl:859-866 (stable sort of the general rows by prio)."""
    ns = bu.size - A.shape[0]
    order = np.argsort(prio[ns:], kind="stable")
    full = np.concatenate([np.arange(ns), order + ns])
    return A[order], bu[full], bl[full], W[full], soft[full], prio[full]


def remove_redundant(A, bu, bl, W, soft, prio):
    """mpc2mpqp.jl:733-773: drop ~zero rows, fold single-entry rows into simple bounds."""
    A, bu, bl, W = A.copy(), bu.copy(), bl.copy(), W.copy()
    ns = bu.size - A.shape[0]
    keep = list(range(ns))
    scale = [1.0] * ns
    for i in range(A.shape[0]):
        a = A[i]
        idx = ns + i
        nrm = np.linalg.norm(a)
        if nrm <= 1e-10:
            continue
        nz = np.flatnonzero(np.abs(a) > 1e-12)
        first = nz[0]
        if a[first] < 0:                         # unique half-plane orientation
            A[i] = -a + 0.0
            a = A[i]
            bu[idx], bl[idx] = -bl[idx], -bu[idx]
            W[idx] = -W[idx] + 0.0
        if nz.size == 1 and first < ns and prio[first] == prio[idx] and soft[first] == soft[idx] \
                and not np.any(W[idx] - W[first]):
            bu[first] = min(bu[first], bu[idx] / nrm)
            bl[first] = max(bl[first], bl[idx] / nrm)
            continue
        keep.append(idx)
        scale.append(1.0 / nrm)
    if len(keep) < bu.size:
        keep = np.asarray(keep, int)
        scale = np.asarray(scale)
        A = A[keep[ns:] - ns] * scale[ns:, None]
        bu, bl, W = bu[keep] * scale, bl[keep] * scale, W[keep] * scale[:, None]
        soft, prio = soft[keep], prio[keep]
    return A, bu, bl, W, soft, prio


def remove_duplicate(A, bu, bl, W, soft, prio):
    """mpc2mpqp.jl:775-828: merge general rows equal to 6 decimals (tightest bounds win)."""
    ns = bu.size - A.shape[0]
    ext = np.hstack([A, W[ns:], soft[ns:, None].astype(float), prio[ns:, None].astype(float)])
    groups, order = {}, []
    for i in range(ext.shape[0]):
        key = tuple(np.round(ext[i], 6) + 0.0)
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(i)
    if len(order) == A.shape[0]:
        return A, bu, bl, W, soft, prio
    firsts = np.array([groups[k][0] for k in order], int)
    A2 = A[firsts]
    bu2 = np.concatenate([bu[:ns], [bu[ns + np.array(groups[k])].min() for k in order]])
    bl2 = np.concatenate([bl[:ns], [bl[ns + np.array(groups[k])].max() for k in order]])
    full = np.concatenate([np.arange(ns), firsts + ns])
    return A2, bu2, bl2, W[full], soft[full], prio[full]


@dataclass
class MPQP:
    """types.jl:75-98 data contract (column-major Float64 on the Julia side; C-order here)."""
    H: np.ndarray
    f: np.ndarray
    H_theta: np.ndarray
    f_theta: np.ndarray
    A: np.ndarray
    bu: np.ndarray
    bl: np.ndarray
    W: np.ndarray
    senses: np.ndarray
    prio: np.ndarray
    nu: int = 1
    nx: int = 0
    is_symmetric: bool = True            # types.jl:91: isapprox(H, H', rtol = 1e-9); false -> DAQP's is_avi (setup.jl:13)

    @property
    def n(self):
        return self.H.shape[0]

    @property
    def m(self):
        return self.bu.size

    @property
    def ms(self):
        return self.bu.size - self.A.shape[0]

    @property
    def nth(self):
        return self.f_theta.shape[1]


def mpc2mpqp(p: MPCProblem) -> MPQP:
    """mpc2mpqp.jl:612-647."""
    F, G, C = extended_system(p)
    Phi, Gam = state_predictor(F, G, p.Np, p.Nc)
    if not p.objectives:                         # normal, single-objective MPC
        Q, R, S, Qf = extended_cost(p)
        H, f, f_theta, H_theta = dense_objective(p, F, Phi, Gam, C, Q, R, S, Qf)
    else:
        H, f, f_theta, H_theta = variational_objective(p, Phi, Gam, C)
    cons = dense_constraints(p, Phi, Gam)
    if p.move_blocks:
        H, f, f_theta, cons = apply_move_block(p, H, f, f_theta, cons)
    cons = sort_by_priority(*cons)
    if p.preprocess:
        cons = remove_redundant(*cons)
        cons = remove_duplicate(*cons)
    A, bu, bl, W, soft, prio = cons
    senses = np.zeros(bu.size, np.int32)         # mpc2mpqp.jl:868-885
    for i in range(bu.size):
        if bu[i] > 1e20 and bl[i] < -1e20:
            senses[i] = IMMUTABLE
        elif abs(bu[i] - bl[i]) < 1e-12:
            senses[i] = EQUALITY
    senses[soft.astype(bool)] += SOFT
    brow = getattr(p, "_binary_rows", np.zeros(0, bool))
    if brow.any():                               # mpc2mpqp.jl:884 (binary rows are simple bounds here)
        nsimple = bu.size - A.shape[0]
        senses[:nsimple][brow[:nsimple]] += BINARY
    bu = np.clip(bu, -1e30, 1e30)
    bl = np.clip(bl, -1e30, 1e30)
    sym = bool(np.linalg.norm(H - H.T) <= 1e-9 * max(np.linalg.norm(H), np.linalg.norm(H.T)))   # mpc2mpqp.jl:897
    return MPQP(H, f, H_theta, f_theta, A, bu, bl, W, senses, prio.astype(np.int32), p.nu, p.nx, sym)


# --------------------------------------------------------------------------- named problems
def pendulum(Np=50, Nc=5) -> MPCProblem:
    """README.md:39-52 == mpc_examples.jl:104-141 linearised at the origin (M=m=1, l=.5, damp=10)."""
    A = np.array([[0, 1, 0, 0], [0, -10, 9.81, 0], [0, 0, 0, 1], [0, -20, 39.24, 0.0]])
    B = 100 * np.array([[0.0], [1.0], [0.0], [2.0]])
    C = np.array([[1.0, 0, 0, 0], [0, 0, 1.0, 0]])
    Ts = 0.01
    F, G = zoh(A, B, Ts)
    return make_mpc(F, G, C, Np=Np, Nc=Nc, Q=[1.2 ** 2, 1.0], R=[0.0], Rr=[1.0],
                    umin=[-2.0], umax=[2.0], Ts=Ts)


def pendulum_benchmark(N=50, soft=True) -> MPCProblem:
    """The reference's published benchmark problem class (docs/src/manual/benchmark.md:4): the inverted pendulum
    on a cart with prediction and control horizons swept TOGETHER, Np = Nc = N in {50, 75, 100, 125}, "both input
    and state constraints imposed".  The benchmark's own script lives in another repository
    (darnstrom/lmpc-codegen-benchmark) and is not part of the reference tree, so the state constraints here are
    this build's stand-in: output bounds |cart position| <= 1.5, |angle| <= 0.2 on steps 2..N, soft as
    set_bounds!(ymin, ymax) makes them by default (setup.jl:94).  n = N variables, 3N - 2 rows."""
    p = pendulum(Np=N, Nc=N)
    p.add_constraint(Ax=p.C, lb=[-1.5, -0.2], ub=[1.5, 0.2], ks=range(2, N + 1), soft=soft)
    return p


def mass_spring(nm=6, Np=10, Nc=10, kappa=1.0, lam=0.0) -> MPCProblem:
    """mpc_examples.jl:241-286: chain of nm masses, force on mass 1, |pos| <= 4 for k = 2..Nc."""
    nx = 2 * nm
    off = np.ones(nm - 1)
    Fx = np.diag(kappa * off, 1) + np.diag(kappa * off, -1) - 2 * kappa * np.eye(nm)
    Fv = np.diag(lam * off, 1) + np.diag(lam * off, -1) - 2 * lam * np.eye(nm)
    A = np.block([[np.zeros((nm, nm)), np.eye(nm)], [Fx, Fv]])
    B = np.zeros((nx, 1))
    B[nm, 0] = 1.0
    F, G = zoh(A, B, 0.5)
    p = make_mpc(F, G, np.eye(nx), Np=Np, Nc=Nc, Q=100 * np.ones(nx), R=[1.0], Rr=[0.0],
                 umin=[-0.5], umax=[0.5], reference_tracking=False, Ts=0.5)
    p.add_constraint(Ax=np.hstack([np.eye(nm), np.zeros((nm, nm))]), lb=-4 * np.ones(nm),
                     ub=4 * np.ones(nm), ks=range(2, Nc + 1))
    return p


def mass_spring_3in(nm=6, Np=10, Nc=10) -> MPCProblem:
    """BASELINE.json config 3 as worded ("12 states / 3 inputs, Nc=10, ~60 ineq"): the reference's
    mass_spring example (mpc_examples.jl:241-286) has ONE input, so this is a SYNTHETIC variant of it
    with forces on masses 1, 3 and 5 (B defined here, everything else as in the example)."""
    p = mass_spring(nm, Np, Nc)
    nx = 2 * nm
    off = np.ones(nm - 1)
    Fx = np.diag(off, 1) + np.diag(off, -1) - 2 * np.eye(nm)
    A = np.block([[np.zeros((nm, nm)), np.eye(nm)], [Fx, np.zeros((nm, nm))]])
    B = np.zeros((nx, 3))
    for c, mass in enumerate((0, 2, 4)):
        B[nm + mass, c] = 1.0
    F, G = zoh(A, B, 0.5)
    q = make_mpc(F, G, np.eye(nx), Np=Np, Nc=Nc, Q=100 * np.ones(nx), R=[1.0, 1.0, 1.0], Rr=np.zeros(3),
                 umin=[-0.5] * 3, umax=[0.5] * 3, reference_tracking=False, Ts=0.5)
    q.add_constraint(Ax=np.hstack([np.eye(nm), np.zeros((nm, nm))]), lb=-4 * np.ones(nm),
                     ub=4 * np.ones(nm), ks=range(2, Nc + 1))
    return q


def aircraft(Np=10, Nc=2) -> MPCProblem:
    """A two-input fixture for the move-blocking SIZE checks (test/runtests.jl:138-176 run them on the `aircraft`
    example), built from that example's plant (src/mpc_examples.jl:174-205: A, B, C, Ts, scaling, weights, |u| <=
    0.5).  It is NOT the reference's example problem: the measured-disturbance feedthrough Dd is left out and the
    output bound is a soft bound on the second output over steps 2..Np, where the reference bounds both outputs at
    step 2 only (`ks = 2:2`).  The move-block structure the test counts depends on nu and the blocks alone; no
    golden or parity vector is generated from this fixture."""
    A = np.array([[-0.0151, -60.5651, 0, -32.174], [-0.0001, -1.3411, 0.9929, 0],
                  [0.00018, 43.2541, -0.86939, 0], [0, 0, 1, 0]])
    B = np.array([[-2.516, -13.136], [-0.1689, -0.2514], [-17.251, -1.5766], [0, 0]])
    C = np.array([[0, 1.0, 0, 0], [0, 0, 0, 1]]) / np.array([[1.0], [200.0]])
    F, G = zoh(A, B, 0.05)
    p = make_mpc(F, 50 * G, C, Np=Np, Nc=Nc, Q=[100.0, 100.0], R=[0.0, 0.0], Rr=[0.01, 0.01],
                 umin=[-0.5, -0.5], umax=[0.5, 0.5], Ts=0.05)
    p.add_constraint(Ax=C[1:2], lb=[-0.5], ub=[0.5], ks=range(2, Np + 1), soft=True)
    return p


def preprocessing_kat() -> MPCProblem:
    """test/runtests.jl:1306-1318 (K4): two Au-only constraints fold into the simple bounds."""
    F, G = zoh(np.array([[0, 1.0], [10, 0]]), np.array([[0.0], [1.0]]), 0.1)
    p = make_mpc(F, G, np.eye(2), Np=10, umin=[-1.0], umax=[1.0], Ts=0.1)
    p.add_constraint(Au=[[-1.0]], lb=[-0.9], ub=[1.5], ks=range(1, 11))
    p.add_constraint(Au=[[1.0]], lb=[-0.5], ub=[2.0], ks=range(1, 11))
    return p


def prestab_kat(prestabilize: bool) -> MPCProblem:
    """test/runtests.jl:119-136 (K2): unstable double-integrator-like plant, |u| <= 1, Np = 30; the
    nominal controller and the one with LQR prestabilising feedback give the same input."""
    F, G = zoh(np.array([[0, 1.0], [10, 0]]), np.array([[0.0], [1.0]]), 0.1)
    p = make_mpc(F, G, np.eye(2), Np=30, umin=[-1.0], umax=[1.0], Ts=0.1)
    if prestabilize:
        p.set_prestabilizing_feedback()
    return p


def satellite(Np=20, Nc=None) -> MPCProblem:
    """mpc_examples.jl:533-546 / example/hybrid.jl: attitude control with one continuous and two
    on/off (binary) thrusters: u1 free, u2 in {0, 1}, u3 in {-1, 0}."""
    A = np.array([[0.0, 1, 0], [0, 0, 0], [0, 0, 0]])
    B = np.array([[0.0, 0, 0], [2.5, 1, 1], [-10, 0, 0]])
    F, G = zoh(A, B, 0.1)
    p = make_mpc(F, G, np.eye(3), Np=Np, Nc=Np if Nc is None else Nc, Q=[0.5e4, 1e-2, 1e-1],
                 R=[10.0, 10.0, 10.0], Rr=np.zeros(3), umin=[-np.inf, 0.0, -1.0], umax=[np.inf, 1.0, 0.0], Ts=0.1)
    p.binary_controls = (1, 2)
    return p


def doc_simple_soft() -> MPCProblem:
    """docs/src/manual/simple.md:60-83 (K8): soft output bounds (set_bounds! ymin/ymax defaults to
    soft=true, ks=2:Np, setup.jl:94) plus one hard general constraint; u = -1 at x=[0.5,1], r=0."""
    F = np.array([[1.0, 0.5], [0.0, 1.0]])
    G = np.array([[0.0], [1.0]])
    C = np.array([[1.0, 0.0], [1.0, 1.0]])
    p = make_mpc(F, G, C, Np=10, Nc=10, Q=[1.0, 1.0], R=[0.0], Rr=[1.0], umin=[-3.0], umax=[3.0])
    p.add_constraint(Ax=C, lb=[0.0, 0.0], ub=[1.0, 2.0], ks=range(2, 11), soft=True)
    p.add_constraint(Ax=[[2.0, -1.0]], lb=[-1.0], ub=[2.0], ks=range(2, 11))
    return p


def generalized_parameter_kat() -> MPCProblem:
    """test/runtests.jl:1250-1268 (K3): scalar integrator, cost (Eu p + eu)'u with Eu = -1, eu = -0.1,
    constraint u + p <= 1 at every step: u = 1.0 for p = 0, u = 0.25 for p = 0.75."""
    p = make_mpc([[1.0]], [[1.0]], [[1.0]], Np=4, Nc=4, Q=[0.0], R=[1e-6], umin=[0.0], umax=[2.0])
    p.Eu = np.array([[-1.0]])
    p.eu = np.array([-0.1])
    p.add_constraint(Au=[[1.0]], Ap=[[1.0]], ub=[1.0], ks=range(1, 5))
    return p


def format_reference_preview(r, ny, Np):
    """utils.jl:84-111: a vector is repeated over the horizon; a trajectory (ny x T) is flattened
    column by column, cut at Np columns or padded with its last column."""
    r = np.asarray(r, float)
    if r.ndim == 1:
        if r.size != ny:
            raise ValueError(f"Reference vector length ({r.size}) must match number of outputs ({ny})")
        return np.tile(r, Np)
    if r.shape[0] != ny:
        raise ValueError(f"Reference matrix must have {ny} rows (number of outputs)")
    if r.shape[1] >= Np:
        return r[:, :Np].T.reshape(-1)
    ext = np.hstack([r, np.tile(r[:, -1:], (1, Np - r.shape[1]))])
    return ext.T.reshape(-1)


def x0_uncertainty_kat() -> MPCProblem:
    """test/runtests.jl:1067-1074 "x0 uncertainty": double integrator, |u| <= 0.2, soft output bounds
    |y| <= 0.5 on k = 2..Np (set_bounds! ymin/ymax: soft, setup.jl:94), dx0 = 0.1: tracking r = 0.5
    stops at the tightened bound, x1 -> 0.4."""
    p = make_mpc([[1, 0.1], [0, 1]], [[0.005], [0.1]], [[1.0, 0.0]], Np=25, umin=[-0.2], umax=[0.2], Ts=0.1)
    p.add_constraint(Ax=p.C, lb=[-0.5], ub=[0.5], ks=range(2, 26), soft=True)
    p.x0_uncertainty = 0.1 * np.ones(2)
    return p


def preview_sim_kat(preview: bool) -> MPCProblem:
    """test/runtests.jl:276-327 "Reference Preview Simulation": discrete double integrator, Np = 5,
    Nc = 3, |u| <= 2, soft output bounds, Q = I, R = 0.1; a step in r1 at k = 10."""
    p = make_mpc([[1, 1], [0, 1]], [[0], [1]], np.eye(2), Np=5, Nc=3, Q=[1.0, 1.0], R=[0.1],
                 umin=[-2.0], umax=[2.0])
    p.add_constraint(Ax=p.C, lb=[-1.0, -0.5], ub=[1.0, 0.5], ks=range(2, 6), soft=True)
    p.reference_preview = preview
    return p


def refcond_kat() -> MPCProblem:
    """test/runtests.jl:669-733 "Codegen Reference Preview - Condensed": double integrator, C = I,
    Np = Nc = 5, |u| <= 2, Q = I, R = 0.1, reference_preview + reference_condensation; the test feeds
    r_traj = [0 .5 1 1 1; 0 0 0 0 0] at x = 0 through Julia, generated C and explicit MPC and asserts
    they agree to 1e-10, and that mpc_update_parameter gives theta = [x; traj2setpoint r_traj; u]."""
    p = make_mpc([[1, 1], [0, 1]], [[0], [1]], np.eye(2), Np=5, Nc=5, Q=[1.0, 1.0], R=[0.1],
                 umin=[-2.0], umax=[2.0])
    p.reference_preview = True
    p.reference_condensation = True
    return p


def observer_disturbance_kat() -> MPCProblem:
    """test/runtests.jl:951-961 "Observer + disturbance": double integrator, Gd = [1 0; 0 0], Dd = [0 1],
    C = [1 0], every default (Np = Nc = 10, Q = I, R = I, no bounds): theta = [x(2); r(1); d(2)].  With the
    Kalman filter (Q = [1,1], R = 1e-2) and d = [1,1] the measured output settles at 0 (|mean| < 1e-2)."""
    return make_mpc([[1, 1], [0, 1]], [[0], [1]], [[1.0, 0.0]], Np=10, Gd=[[1, 0], [0, 0]], Dd=[[0, 1]])


def disturbance_preview_kat(preview=True) -> MPCProblem:
    """test/runtests.jl:735-774 "Codegen Disturbance Preview": double integrator, Gd = [0; 1], C = [1 0],
    Np = Nc = 4, |u| <= 0.5, Q = 10, R = 0.1, disturbance_preview: theta = [x(2); r(1); vec(d_traj)(4)]."""
    p = make_mpc([[1, 1], [0, 1]], [[0], [1]], [[1.0, 0.0]], Np=4, Nc=4, Q=[10.0], R=[0.1],
                 umin=[-0.5], umax=[0.5], Gd=[[0], [1]])
    p.disturbance_preview = preview
    return p


def parameter_preview_kat() -> MPCProblem:
    """test/runtests.jl:1270-1304 "Generalized Parameter Codegen for Explicit Preview": scalar integrator,
    Np = Nc = 3, 0 <= u <= 2, Q = 0, R = 1, Eu = -2, parameter_preview: theta = [x; r; p_0 p_1 p_2].  With
    Q = 0 the moves decouple: u_k = argmin 1/2 u^2 - 2 p_k u = 2 p_k clipped to [0, 2]."""
    p = make_mpc([[1.0]], [[1.0]], [[1.0]], Np=3, Nc=3, Q=[0.0], R=[1.0], umin=[0.0], umax=[2.0])
    p.Eu = np.array([[-2.0]])
    p.parameter_preview = True
    return p


def offset_kat() -> MPCProblem:
    """test/runtests.jl:1320-1327 "Set offset": first-order plant, uo = 10, ho = 0.5; the closed loop
    with r = 1.5 settles at u = 10.5, y = 1.5."""
    p = make_mpc([[0.778800783]], [[1.0]], [[0.44239843385]], Np=10, Q=[1.0], R=[0.0], Rr=[0.1])
    return p.set_offset(uo=[10.0], ho=[0.5])


def moveblock_kat() -> MPCProblem:
    """test/runtests.jl:1329-1335 "Unconstrained": move_block!([2,2,2,24]) on Np = 10 (clipped to
    [2,2,2,4], Nc = 7, four moves), Q = 1, R = Rr = 0, no bounds; y reaches r = 5."""
    p = make_mpc([[0.77880078307]], [[1.0]], [[2.211992169]], Np=10, Q=[1.0], R=[0.0], Rr=[0.0], Ts=100.0)
    return p.move_block([2, 2, 2, 24])


def game_kat() -> MPCProblem:
    """test/runtests.jl:1337-1358 "Game-theoretic MPC": double integrator with two inputs, one player per input
    (player 1 tracks x1, player 2 tracks x2, both Rr = 1e3), |u| <= 1, move blocks [1,1,8]; the closed loop from
    x0 = [10,10] with r = [10,0] ends at y = [10, 0] (atol 1e-4) and mpQP.H is not symmetric."""
    p = make_mpc([[1, 0.1], [0, 1]], [[0, 0], [1, 1]], np.eye(2), Np=10, umin=[-1.0, -1.0], umax=[1.0, 1.0])
    p.add_objective([0], Q=[1.0, 0.0], Rr=1e3)
    p.add_objective([1], Q=[0.0, 1.0], Rr=1e3)
    return p.move_block([1, 1, 8])


def form_parameter(p: MPCProblem, x, r=None, uprev=None, par=None, d=None):
    """explicit.jl:54-63: theta = [x; r; d; uprev; p] (r, d, uprev, p default 0)."""
    nx, nr, nd, nuprev, npb = p.parameter_dims()
    x = np.asarray(x, float).reshape(nx)
    if p.reference_preview and r is not None and nr > 0:
        r = format_reference_preview(np.asarray(r, float), p.ny, p.Np)
        if p.reference_condensation:             # utils.jl:141-146 condense_reference
            r = p.traj2setpoint @ r
    r = np.zeros(nr) if r is None else np.asarray(r, float).reshape(-1)[:nr]
    u = np.zeros(nuprev) if uprev is None else np.asarray(uprev, float).reshape(-1)[:nuprev]
    if p.parameter_preview and par is not None and npb > 0:
        par = np.asarray(par, float)
        nb0 = p.np_base()
        if par.ndim == 1 and par.size == nb0:                                   # utils.jl:219-261
            par = np.tile(par, p.Np)
        elif par.ndim == 2:
            par = format_reference_preview(par, nb0, p.Np)
    pp = np.zeros(npb) if par is None else np.asarray(par, float).reshape(-1)[:npb]
    if p.disturbance_preview and d is not None and nd > 0:
        d = format_reference_preview(np.asarray(d, float), p.nd, p.Np)     # utils.jl:149-170: same tiling / padding
    dd = np.zeros(nd) if d is None else np.asarray(d, float).reshape(-1)[:nd]
    return np.concatenate([x, r, dd, u, pp])
