"""Host-side mirror of the reference's operator interface for the online path.

Julia is not available in the build image, so the layer a LinearMPC.jl maintainer would write in
Julia is mirrored here in Python with the reference's names, argument meaning
and error behaviour, so that the parity tests read like test/runtests.jl:

    MPQP                         /root/reference/src/types.jl:75-98   (data contract)
    MPC.setup()                  /root/reference/src/setup.jl:7-29    (`setup!`)
    MPC.form_parameter           /root/reference/src/explicit.jl:54-63
    MPC.format_reference / format_disturbance / format_affine_parameters
                                 /root/reference/src/utils.jl:78-261  (preview tiling and padding)
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
                 soft_weight=1e6, device=0, Np=1, reference_preview=False, disturbance_preview=False,
                 parameter_preview=False, reference_condensation=False, traj2setpoint=None):
        self.mpQP = mpqp
        self.nx, self.nu, self.nr, self.nd, self.nuprev, self.np = nx, nu, nr, nd, nuprev, np_
        # MPCSettings (types.jl:54-69): with a preview the block of theta holds Np columns
        self.Np = int(Np)
        self.reference_preview = bool(reference_preview)
        self.disturbance_preview = bool(disturbance_preview)
        self.parameter_preview = bool(parameter_preview)
        # settings.reference_condensation (types.jl:54,65): the Np-column reference trajectory is
        # collapsed to ONE setpoint by mpc.traj2setpoint (ny x ny*Np, computed by the condensing step,
        # mpc2mpqp.jl:550-566), so the block of theta is ny wide again
        self.reference_condensation = bool(reference_condensation)
        self.traj2setpoint = None if traj2setpoint is None else np.asarray(traj2setpoint, float)
        self.ny = nr // self.Np if (self.reference_preview and not self.reference_condensation) else nr   # model.ny
        self.nd_base = nd // self.Np if self.disturbance_preview else nd       # model.nd
        self.np_base = np_ // self.Np if self.parameter_preview else np_       # utils.jl:204-217
        self.K = np.zeros((nu, nx)) if K is None else np.asarray(K, float).reshape(nu, nx)
        self.uprev = np.zeros(nu)
        self.settings = default_settings()
        self.settings.rho_soft = 1.0 / soft_weight          # setup.jl:26
        self.device = device
        self.mpqp_issetup = False
        self.opt_model: BatchedQP | None = None             # stands where DAQP.Model stands
        self._ctrl_model: BatchedQP | None = None           # nout = nu, K folded in (batched path)
        # what the handles were built from / last given -- the glue's cache key (integration/LmpcHipExt.jl
        # `_model_for`): the mpQP OBJECT (setup! makes a new one, setup.jl:9) and the solver settings
        self._opt_mpqp = None
        self._ctrl_mpqp = None
        self._pushed_settings = None
        self.opt_model_eps_prox = 0.0

    # setup.jl:7-29
    def setup(self):
        """`setup!`: DAQP.setup(model, H, f, A, bu, bl, senses; break_points = mpQP.break_points,
        is_avi = !mpQP.is_symmetric) (setup.jl:11-13) -> lmpc_setup_ex with the same two keywords.  A non-empty
        break_points (prioritised constraints) is refused by the library with LMPC_ERR_UNSUPPORTED -- loudly, never by
        solving a different problem; a non-symmetric H sets the handle up for the variational inequality."""
        q = self.mpQP
        old = self.opt_model
        self.opt_model = BatchedQP.from_mpqp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses,
                                             nout=q.H.shape[0], settings=self.settings, device=self.device,
                                             break_points=np.asarray(q.break_points, np.int32),
                                             is_avi=not q.is_symmetric)
        if old is not None:
            old.close()
        if self._ctrl_model is not None:
            self._ctrl_model.close()
        self._ctrl_model = None
        self._opt_mpqp = q
        self._pushed_settings = self._settings_key()
        self.opt_model_eps_prox = self.settings.eps_prox
        self.mpqp_issetup = True
        return self

    def _settings_key(self):
        return tuple(getattr(self.settings, k) for k, _ in self.settings._fields_)

    def _model_for(self) -> BatchedQP:
        """The handle `solve` runs on -- the logic of the Julia glue's `_model_for` (integration/LmpcHipExt.jl), kept
        identical here so that the test-suite exercises it: (1) an MPC whose data changed (`set_*!` sets
        mpqp_issetup = false, /root/reference/src/setup.jl:36-160) is set up again first, as `solve` does
        (utils.jl:269); (2) a handle built from ANOTHER mpQP object than the MPC now holds is stale -- `setup!`
        always makes a new mpQP (setup.jl:9), whoever called it -- and is rebuilt, the old one freed; (3) the solver
        settings on the model are compared with what the handle was last given and pushed if they differ
        (`DAQP.settings(mpc.opt_model, Dict(...))` after the first solve)."""
        if not self.mpqp_issetup or self.opt_model is None or self._opt_mpqp is not self.mpQP:
            self.setup()
        key = self._settings_key()
        if key != self._pushed_settings and self.settings.eps_prox != self.opt_model_eps_prox:
            self.setup()                      # eps_prox is part of the factorisation: a new handle, like DAQP's own setup
        if key != self._pushed_settings:
            for model in (self.opt_model, self._ctrl_model):
                if model is not None:
                    model.set_settings(self.settings)
            self._pushed_settings = key
        return self.opt_model

    def get_parameter_dims(self):
        return self.nx, self.nr, self.nd, self.nuprev, self.np

    @staticmethod
    def _tile(a, w, Np, what, unit):
        """The common body of utils.jl:84-111 / :149-170 / :240-258: a vector of the base width is
        repeated over the horizon, a (w x T) trajectory is flattened column by column, cut at Np
        columns or padded with its last column."""
        a = np.asarray(a, float)
        if a.ndim == 1:
            if a.size != w:
                raise ValueError(f"{what} vector length ({a.size}) must match number of {unit} ({w})")
            return np.tile(a, Np)
        if a.ndim != 2:
            raise ValueError(f"{what} must be a vector or matrix")
        if a.shape[0] != w:
            raise ValueError(f"{what} matrix must have {w} rows (number of {unit})")
        if a.shape[1] < Np:
            a = np.hstack([a, np.tile(a[:, -1:], (1, Np - a.shape[1]))])
        return a[:, :Np].T.reshape(-1)

    @staticmethod
    def _single(a, w, what, unit):
        a = np.asarray(a, float)
        if a.ndim == 1:
            if a.size != w:
                raise ValueError(f"{what} vector length ({a.size}) must match number of {unit} ({w})")
            return a
        if a.ndim == 2:                                   # first column in non-preview mode
            if a.shape[0] != w:
                raise ValueError(f"{what} matrix must have {w} rows (number of {unit})")
            return a[:, 0].copy()
        raise ValueError(f"{what} must be a vector or matrix")

    # utils.jl:136-146
    def condense_reference(self, r):
        if self.reference_condensation:
            return self.traj2setpoint @ np.asarray(r, float).reshape(-1)
        return r

    # utils.jl:78-133
    def format_reference(self, r):
        if self.nr == 0:                                  # reference_tracking off
            return np.zeros(0)
        if r is None:
            r = np.zeros(self.ny)
        if np.size(r) == 0:
            return np.zeros(0)
        if self.reference_preview:
            return self.condense_reference(self._tile(r, self.ny, self.Np, "Reference", "outputs"))
        return self._single(r, self.ny, "Reference", "outputs")

    # utils.jl:141-201
    def format_disturbance(self, d):
        if self.nd_base == 0:
            return np.zeros(0)
        if d is None:
            d = np.zeros(self.nd_base)
        if np.size(d) == 0:
            return np.zeros(0)
        if self.disturbance_preview:
            return self._tile(d, self.nd_base, self.Np, "Disturbance", "disturbances")
        return self._single(d, self.nd_base, "Disturbance", "disturbances")

    # utils.jl:219-261
    def format_affine_parameters(self, p):
        if self.np == 0:
            return np.zeros(0)
        if p is None:
            return np.zeros(self.np)
        p = np.asarray(p, float)
        if p.ndim == 1 and p.size == self.np_base:
            return np.tile(p, self.Np) if self.parameter_preview else p.copy()
        if p.ndim == 1 and p.size == self.np:
            return p.copy()
        if p.ndim == 2:
            if p.shape[0] != self.np_base:
                raise ValueError(f"Generalized parameter matrix must have {self.np_base} rows")
            if not self.parameter_preview:
                return p[:, 0].copy()
            return self._tile(p, self.np_base, self.Np, "Generalized parameter", "parameters")
        raise ValueError("Generalized parameters must be a vector or matrix")

    # explicit.jl:54-63
    def form_parameter(self, x, r=None, d=None, uprev=None, p=None):
        x = np.asarray(x, float).reshape(-1)
        if x.size != self.nx:
            raise ValueError(f"State vector must have length {self.nx}")
        r = self.format_reference(r)
        d = self.format_disturbance(d)
        if d.size != self.nd:
            raise ValueError(f"Disturbance vector must have length {self.nd}")
        up = self.uprev[:self.nuprev] if uprev is None else np.asarray(uprev, float).reshape(-1)[:self.nuprev]
        p = self.format_affine_parameters(p)
        if p.size != self.np:
            raise ValueError(f"Generalized parameters must have length {self.np}")
        return np.concatenate([x, r, d, up, p])

    # docs/src/manual/solver.md:19-22 -- DAQP.settings(mpc.opt_model, Dict(:iter_limit => 2000, ...))
    def solver_settings(self, **changes):
        """The mirror of `DAQP.settings(mpc.opt_model[, Dict])`: without arguments the settings in force, with
        keyword arguments (primal_tol, dual_tol, zero_tol, progress_tol, fval_bound, rho_soft, cycle_tol,
        iter_limit) they are changed on the MPC and on every handle it has set up -- the batched backend reads the
        user's solver settings, it does not assume DAQP's defaults."""
        for k, v in changes.items():
            if not hasattr(self.settings, k):
                raise KeyError(f"unknown solver setting {k}")
            setattr(self.settings, k, type(getattr(self.settings, k))(v))
        if changes and "eps_prox" not in changes:
            for model in (self.opt_model, self._ctrl_model):
                if model is not None:
                    model.set_settings(self.settings)
            self._pushed_settings = self._settings_key()
        # (eps_prox: the next solve sets the handles up again, _model_for)
        return {k: getattr(self.settings, k) for k, _ in self.settings._fields_}

    # utils.jl:268-283
    def solve(self, theta):
        """-> (xdaqp, fval, exitflag, info) like DAQP.solve: ONE parameter vector through `lmpc_solve_one`, the
        call the Julia glue's `LinearMPC.solve(mpc::MPC, θ::AbstractVector)` makes (integration/LmpcHipExt.jl), so
        that compute_control / compute_control_trajectory / a Simulation loop run on it unchanged.  fval =
        1/2 x'Hx + (f + f_theta θ)'x is formed on the host from x* (the reference reads x* and exitflag only,
        utils.jl:45-48)."""
        model = self._model_for()
        theta = np.asarray(theta, float).reshape(-1)
        x, flag = model.solve_one(theta)
        q = self.mpQP
        fth = q.f + q.f_theta @ theta
        fval = 0.5 * x @ q.H @ x + fth @ x
        info = {"exitflag": int(flag), "status": "Solved" if flag >= 1 else "Failed"}
        return x.copy(), float(fval), int(flag), info

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
        self._model_for()                         # (stale-handle and settings checks; drops a stale _ctrl_model)
        if self._ctrl_model is None or self._ctrl_mpqp is not self.mpQP:
            q = self.mpQP
            if self._ctrl_model is not None:
                self._ctrl_model.close()
            self._ctrl_model = BatchedQP.from_mpqp(
                q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=self.nu,
                K=self.K if np.any(self.K) else None, nx=self.nx, settings=self.settings,
                device=self.device, break_points=np.asarray(q.break_points, np.int32), is_avi=not q.is_symmetric)
            self._ctrl_mpqp = q
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
        return self._model_for().solve(Theta)

    def compute_control_batch(self, X, R=None, D=None, Uprev=None, P=None, check=True):
        """Batched `compute_control`; stateless (uprev is an input, nothing is stored)."""
        Theta = self.form_parameter_batch(X, R, D, Uprev, P)
        U, ef, _, _ = self.control_model().solve(Theta, want_iters=False, want_active=False)
        if check:
            assert np.all(ef >= 1), f"{int(np.sum(ef < 1))} problems did not solve (min flag {ef.min()})"
        return U, ef


class GeneratedController:
    """What `LinearMPC.codegen(mpc; warm_start)` produces, batched: the counterpart of the generated
    `int mpc_compute_control(c_float* control, c_float* state, c_float* reference, c_float* disturbance
    [, c_float* affine_parameter])` (reference src/codegen.jl:1-17,139-218, codegen/mpc_update_qp.c:29-54,
    codegen/mpc_update_parameter.c) behind `lmpc_set_parameter_layout` + `lmpc_compute_control`.

    The LDP arrays the generator writes (Dth, du, dl, Uth_offset incl. -K, u_offset: codegen.jl:140-141,
    183-189) are what the handle's pack holds; N_STATE ... N_AFFINE_PARAMETER come from the MPC's
    parameter dimensions, N_PREVIEW_HORIZON / traj2setpoint from settings.reference_condensation."""

    def __init__(self, mpc: MPC, warm_start=False):
        self.mpc = mpc
        self.warm_start = bool(warm_start)
        self.model = mpc.control_model()
        cond = mpc.reference_preview and mpc.reference_condensation
        self.model.set_parameter_layout(mpc.nx, mpc.nr, mpc.nd, mpc.nuprev, mpc.np,
                                        preview_horizon=mpc.Np if cond else 0,
                                        traj2setpoint=mpc.traj2setpoint if cond else None)

    def set_observer(self, plant_dynamics, measurement_function, k_transpose, nx, nu, nd, ny):
        """The observer part of the generated code (`codegen(mpc.state_observer, ...)`, codegen.jl:209-212):
        the three arrays as src/observer.jl:139-141 writes them."""
        self.model.set_observer(plant_dynamics, measurement_function, k_transpose, nx, nu, nd, ny)

    def mpc_predict_state(self, state, control, disturbance=None):
        """generated `mpc_predict_state(state, control, disturbance)` for N scenarios, state in place."""
        return self.model.predict_state(state, control, disturbance)

    def mpc_correct_state(self, state, measurement, disturbance=None):
        """generated `mpc_correct_state(state, measurement, disturbance)` for N scenarios, state in place."""
        return self.model.correct_state(state, measurement, disturbance)

    def mpc_compute_control(self, control, state, reference=None, disturbance=None, affine_parameter=None):
        """control: (N, nu) float64, in = previous control, out = u* (in place); returns exit flags (N,).
        reference: (N, nr) -- or (N, ny*Np), each an ny x Np trajectory column by column, when the
        controller condenses references.  None stands for the C caller's NULL."""
        return self.model.compute_control(control, state, reference, disturbance, affine_parameter,
                                          warm=self.warm_start)
