"""`BatchedQP`: Python face of one `lmpc_handle` (include/lmpc_hip.h).

It stands where `mpc.opt_model` (a DAQPBase.Model) stands in the reference
(/root/reference/src/types.jl:141, setup.jl:11-13): set up once from the mpQP, then solved for
many parameter points.  All arithmetic happens in the HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _cabi
from ._cabi import Block, LmpcError, Observer, ParamLayout, Settings, check, lib

_vp = ctypes.c_void_p


def _f64(a, order="C"):
    # aligned + requested order; no OWNDATA requirement (it would copy every reshaped view, 56 MB per
    # call for a 10^6-point theta)
    return np.require(np.asarray(a, dtype=np.float64), requirements=["A", order[0]])


def _ptr(a):
    return _vp(a.ctypes.data) if a is not None and a.size else None


def transform(H, f, f_theta, A, bu, bl, W, senses=None, nout=None, K=None, nx=0):
    """Host-only QP -> LDP transform (`lmpc_transform`; reference codegen.jl:239-280 `qp2ldp`).

    Returns a dict with row-major M, du, dl, Dth, Rout, x0, Xth (+ dims)."""
    H = _f64(H, "F")
    n = H.shape[0]
    f_theta = _f64(np.asarray(f_theta, float).reshape(n, -1), "F")
    nth = f_theta.shape[1]
    bu = _f64(np.asarray(bu, float).reshape(-1))
    bl = _f64(np.asarray(bl, float).reshape(-1))
    m = bu.size
    A = _f64(np.asarray(A, float).reshape(-1, n), "F")
    ms = m - A.shape[0]
    W = _f64(np.asarray(W, float).reshape(m, nth), "F")
    f = _f64(np.zeros(n) if f is None else np.asarray(f, float).reshape(n))
    nout = n if nout is None else int(nout)
    sense = np.ascontiguousarray(np.zeros(m, np.int32) if senses is None else senses, dtype=np.int32)
    Kf = None if K is None else _f64(np.asarray(K, float).reshape(nout, -1), "F")
    out = dict(M=np.empty((m, n)), du=np.empty(m), dl=np.empty(m), Dth=np.empty((m, nth)),
               Rout=np.empty((nout, n)), x0=np.empty(nout), Xth=np.empty((nout, nth)))
    rc = lib().lmpc_transform(n, m, ms, nth, nout, _ptr(H), _ptr(f), _ptr(f_theta), _ptr(A),
                              _ptr(bu), _ptr(bl), _ptr(W), _ptr(sense), _ptr(Kf) if Kf is not None else None,
                              int(nx if K is not None else 0),
                              *[_ptr(out[k]) for k in ("M", "du", "dl", "Dth", "Rout", "x0", "Xth")])
    check(rc)
    out.update(n=n, m=m, ms=ms, nth=nth, nout=nout, sense=sense)
    return out


def transform_avi(H, f, f_theta, A, bu, bl, W, senses=None, nout=None, K=None, nx=0):
    """Host-only transform of a NON-symmetric problem (`lmpc_transform_avi`; what an is_avi setup precomputes):
    dict with row-major ML, MR, G, du, dl, Dth, Rout, x0, Xth (+ dims)."""
    H = _f64(H, "F")
    n = H.shape[0]
    f_theta = _f64(np.asarray(f_theta, float).reshape(n, -1), "F")
    nth = f_theta.shape[1]
    bu = _f64(np.asarray(bu, float).reshape(-1))
    bl = _f64(np.asarray(bl, float).reshape(-1))
    m = bu.size
    A = _f64(np.asarray(A, float).reshape(-1, n), "F")
    ms = m - A.shape[0]
    W = _f64(np.asarray(W, float).reshape(m, nth), "F")
    f = _f64(np.zeros(n) if f is None else np.asarray(f, float).reshape(n))
    nout = n if nout is None else int(nout)
    sense = np.ascontiguousarray(np.zeros(m, np.int32) if senses is None else senses, dtype=np.int32)
    Kf = None if K is None else _f64(np.asarray(K, float).reshape(nout, -1), "F")
    out = dict(ML=np.empty((m, n)), MR=np.empty((m, n)), G=np.empty((m, m)), du=np.empty(m), dl=np.empty(m),
               Dth=np.empty((m, nth)), Rout=np.empty((nout, n)), x0=np.empty(nout), Xth=np.empty((nout, nth)))
    rc = lib().lmpc_transform_avi(n, m, ms, nth, nout, _ptr(H), _ptr(f), _ptr(f_theta), _ptr(A),
                                  _ptr(bu), _ptr(bl), _ptr(W), _ptr(sense), _ptr(Kf) if Kf is not None else None,
                                  int(nx if K is not None else 0),
                                  *[_ptr(out[k]) for k in ("ML", "MR", "G", "du", "dl", "Dth", "Rout", "x0", "Xth")])
    check(rc)
    out.update(n=n, m=m, ms=ms, nth=nth, nout=nout, sense=sense)
    return out


def _dev_arg(t, name, dtype, numel, device, optional=True):
    """Validate a CUDA tensor handed to a *_device entry point as a raw pointer: the kernels write
    through it with a fixed element size and a dense layout, so a wrong dtype, a strided view or a
    tensor on another GPU would silently corrupt memory.  Returns the pointer (or None)."""
    if t is None:
        if optional:
            return None
        raise ValueError(f"{name} is required")
    if not t.is_cuda or t.device.index != device:
        raise ValueError(f"{name} must be a CUDA tensor on cuda:{device}")
    if t.dtype != dtype:
        raise ValueError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if t.numel() != numel:
        raise ValueError(f"{name} must hold {numel} elements, got {t.numel()}")
    return _vp(t.data_ptr())


def _mask_dtypes():
    import torch
    return (torch.int64, torch.uint64) if hasattr(torch, "uint64") else (torch.int64,)


class BatchedQP:
    """One condensed-MPC QP structure resident on one GPU, solved for batches of theta."""

    def __init__(self, handle, device):
        self._h = handle
        self.device = device
        dims = (ctypes.c_int32 * 6)()
        check(lib().lmpc_get_dims(self._h, ctypes.byref(dims)), self._h)
        self.n, self.m, self.ms, self.nth, self.nout, self.words = (int(v) for v in dims)

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_mpqp(cls, H, f, f_theta, A, bu, bl, W, senses=None, nout=None, K=None, nx=0,
                  settings: Settings | None = None, device=0, break_points=None, is_avi=None):
        """DAQP.setup + DAQP.settings equivalent (reference setup.jl:11-13,26).  `break_points` / `is_avi` are the two
        keywords the reference passes to DAQP.setup (mpQP.break_points, !mpQP.is_symmetric): given, the call goes
        through lmpc_setup_ex; left None, lmpc_setup decides is_avi from H the way the reference does."""
        H = _f64(H, "F")
        n = H.shape[0]
        f_theta = _f64(np.asarray(f_theta, float).reshape(n, -1), "F")
        nth = f_theta.shape[1]
        bu = _f64(np.asarray(bu, float).reshape(-1))
        bl = _f64(np.asarray(bl, float).reshape(-1))
        m = bu.size
        A = _f64(np.asarray(A, float).reshape(-1, n), "F")
        ms = m - A.shape[0]
        W = _f64(np.asarray(W, float).reshape(m, nth), "F")
        f = _f64(np.zeros(n) if f is None else np.asarray(f, float).reshape(n))
        nout = n if nout is None else int(nout)
        sense = np.ascontiguousarray(np.zeros(m, np.int32) if senses is None else senses, dtype=np.int32)
        Kf = None if K is None else _f64(np.asarray(K, float).reshape(nout, -1), "F")
        h = _vp()
        sp = ctypes.cast(ctypes.pointer(settings), _vp) if settings is not None else None
        if break_points is None and is_avi is None:
            rc = lib().lmpc_setup(ctypes.byref(h), n, m, ms, nth, nout, _ptr(H), _ptr(f), _ptr(f_theta),
                                  _ptr(A), _ptr(bu), _ptr(bl), _ptr(W), _ptr(sense),
                                  _ptr(Kf) if Kf is not None else None, int(nx if K is not None else 0), sp, int(device))
        else:
            bp = np.ascontiguousarray(np.zeros(0, np.int32) if break_points is None else break_points, dtype=np.int32)
            rc = lib().lmpc_setup_ex(ctypes.byref(h), n, m, ms, nth, nout, _ptr(H), _ptr(f), _ptr(f_theta),
                                     _ptr(A), _ptr(bu), _ptr(bl), _ptr(W), _ptr(sense),
                                     _ptr(Kf) if Kf is not None else None, int(nx if K is not None else 0), sp,
                                     _ptr(bp) if bp.size else None, int(bp.size), int(bool(is_avi)), int(device))
        check(rc)
        return cls(h, device)

    @classmethod
    def from_ldp(cls, M, du, dl, Dth, Rout, x0, Xth, sense=None, ms=0,
                 settings: Settings | None = None, device=0):
        """Setup from the arrays the reference's code generator emits (codegen.jl:183-189)."""
        M = _f64(M)
        m, n = M.shape
        Dth = _f64(np.asarray(Dth, float).reshape(m, -1))
        nth = Dth.shape[1]
        Rout = _f64(np.asarray(Rout, float).reshape(-1, n))
        nout = Rout.shape[0]
        du, dl, x0 = _f64(du), _f64(dl), _f64(x0)
        Xth = _f64(np.asarray(Xth, float).reshape(nout, nth))
        sense = np.ascontiguousarray(np.zeros(m, np.int32) if sense is None else sense, dtype=np.int32)
        h = _vp()
        rc = lib().lmpc_setup_ldp(ctypes.byref(h), n, m, int(ms), nth, nout, _ptr(M), _ptr(du), _ptr(dl),
                                  _ptr(Dth), _ptr(Rout), _ptr(x0), _ptr(Xth), _ptr(sense),
                                  ctypes.cast(ctypes.pointer(settings), _vp) if settings is not None else None,
                                  int(device))
        check(rc)
        return cls(h, device)

    # ------------------------------------------------------------------ inspection
    @property
    def kernel_name(self) -> str:
        return lib().lmpc_kernel_name(self._h).decode()

    def wave_stats(self):
        """Working-set statistics of the wavefront kernel as the handle last saw them (lmpc_wave_stats): dict with the
        problems finished, how many of them stayed within 24 / 32 / 48 rows, and the first-pass capacity of the next
        call (0 = one pass)."""
        out = (ctypes.c_ulonglong * 5)()
        check(lib().lmpc_wave_stats(self._h, out), self._h)
        return {"problems": int(out[0]), "within_24": int(out[1]), "within_32": int(out[2]), "within_48": int(out[3]),
                "first_pass_rows": int(out[4])}

    def ldp(self):
        """The constant pack the kernels use (row-major) -- what tests hand to the oracle."""
        out = dict(M=np.empty((self.m, self.n)), du=np.empty(self.m), dl=np.empty(self.m),
                   Dth=np.empty((self.m, self.nth)), Rout=np.empty((self.nout, self.n)),
                   x0=np.empty(self.nout), Xth=np.empty((self.nout, self.nth)))
        sense = np.zeros(self.m, np.int32)
        check(lib().lmpc_get_ldp(self._h, *[_ptr(out[k]) for k in ("M", "du", "dl", "Dth", "Rout", "x0", "Xth")],
                                 _ptr(sense)), self._h)
        out.update(sense=sense, n=self.n, m=self.m, ms=self.ms, nth=self.nth, nout=self.nout)
        return out

    @property
    def is_avi(self) -> bool:
        """The handle was set up for a non-symmetric H (variational objective; reference setup.jl:13 is_avi)."""
        return bool(lib().lmpc_is_avi(self._h))

    def avi_pack(self):
        """The constant pack of a variational-inequality handle: ldp() (M = the scaled rows ML) plus MR and the full
        non-symmetric Gram matrix G -- what tests hand to the oracle's AVI solver."""
        out = self.ldp()
        out["ML"] = out["M"]
        out["MR"] = np.empty((self.m, self.n))
        out["G"] = np.empty((self.m, self.m))
        check(lib().lmpc_get_avi(self._h, _ptr(out["MR"]), _ptr(out["G"])), self._h)
        return out

    def prox_pack(self):
        """The extra arrays of a proximal-point handle (eps_prox > 0): (H + eps I)^-1, the full-length affine map of
        the unconstrained optimum, the outputs' feedback term -- what tests hand to the oracle next to avi_pack()."""
        out = dict(Hinv=np.empty((self.n, self.n)), x0f=np.empty(self.n), Xthf=np.empty((self.n, self.nth)),
                   Kth=np.empty((self.nout, self.nth)))
        lib().lmpc_get_prox.argtypes = [_vp] * 5
        lib().lmpc_get_prox.restype = ctypes.c_int
        check(lib().lmpc_get_prox(self._h, *[_ptr(out[k]) for k in ("Hinv", "x0f", "Xthf", "Kth")]), self._h)
        return out

    def set_settings(self, settings: Settings):
        check(lib().lmpc_set_settings(self._h, ctypes.byref(settings)), self._h)

    # ------------------------------------------------------------------ solving
    def solve(self, theta, warm=None, want_iters=True, want_active=True):
        """Host arrays in, host arrays out (`lmpc_solve_batch`)."""
        theta = _f64(np.asarray(theta, float).reshape(-1, self.nth) if self.nth else np.zeros((len(theta), 0)))
        N = theta.shape[0]
        # outputs are touched before the call: a device-to-host copy into pages the OS has not mapped
        # yet runs at a fraction of the PCIe rate (measured: 22 ms instead of 1.3 ms for 10^6 problems)
        x = np.zeros((N, self.nout))
        ef = np.zeros(N, np.int32)
        it = np.zeros(N, np.int32) if want_iters else None
        act = np.zeros((N, self.words), np.uint64) if want_active else None
        for a_ in (x, ef, it, act):
            if a_ is not None:
                a_.fill(0)
        w = None
        if warm is not None:
            w = np.ascontiguousarray(np.asarray(warm, np.uint64).reshape(N, self.words))
        check(lib().lmpc_solve_batch(self._h, N, _ptr(theta), _ptr(x), _ptr(ef),
                                     _ptr(it) if it is not None else None,
                                     _ptr(act) if act is not None else None,
                                     _ptr(w) if w is not None else None), self._h)
        return x, ef, it, act

    def solve_f32(self, theta, warm=None, want_iters=True, want_active=True):
        """Binary32 solve (`lmpc_solve_batch_f32`; reference codegen.jl:19 float_type="float"):
        float32 host arrays in and out, wavefront kernel."""
        theta = np.ascontiguousarray(np.asarray(theta, np.float32).reshape(-1, self.nth) if self.nth
                                     else np.zeros((len(theta), 0), np.float32))
        N = theta.shape[0]
        x = np.empty((N, self.nout), np.float32)
        ef = np.empty(N, np.int32)
        it = np.empty(N, np.int32) if want_iters else None
        act = np.zeros((N, self.words), np.uint64) if want_active else None
        w = None
        if warm is not None:
            w = np.ascontiguousarray(np.asarray(warm, np.uint64).reshape(N, self.words))
        check(lib().lmpc_solve_batch_f32(self._h, N, _ptr(theta), _ptr(x), _ptr(ef),
                                         _ptr(it) if it is not None else None,
                                         _ptr(act) if act is not None else None,
                                         _ptr(w) if w is not None else None), self._h)
        return x, ef, it, act

    def solve_one(self, theta):
        """DAQP.solve shape for one theta: (x*, exitflag) (`lmpc_solve_one`)."""
        theta = _f64(np.asarray(theta, float).reshape(self.nth))
        x = np.empty(self.nout)
        rc = lib().lmpc_solve_one(self._h, _ptr(theta), _ptr(x))
        if rc <= -100:
            raise LmpcError(rc, _cabi.last_error(self._h))
        return x, rc

    def solve_device(self, theta, x=None, exitflag=None, iters=None, active=None, warm=None, stream=None):
        """Device-resident batch (`lmpc_solve_batch_device`): torch CUDA tensors in and out, the
        launch is enqueued on `stream` (default: torch's current stream) and NOT synchronised."""
        import torch
        if not theta.is_cuda or theta.dtype not in (torch.float64, torch.float32) or not theta.is_contiguous():
            raise ValueError("theta must be a contiguous float64 (or float32: binary32 path) CUDA tensor of shape (N, nth)")
        if theta.device.index != self.device:
            raise ValueError("theta lives on a different GPU than this handle")
        N = theta.shape[0]
        dev = theta.device
        if x is None:
            x = torch.empty((N, self.nout), dtype=theta.dtype, device=dev)
        if x.dtype != theta.dtype:
            raise ValueError("x and theta must have the same dtype")
        if exitflag is None:
            exitflag = torch.empty(N, dtype=torch.int32, device=dev)
        if theta.dim() != 2 or theta.shape[1] != self.nth:
            raise ValueError(f"theta must have shape (N, {self.nth})")
        # `stream` is a raw hipStream_t: tensors this call allocated live on torch's current stream, so the
        # caller who passes a foreign stream must keep x / exitflag alive until that stream has finished
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        fn = lib().lmpc_solve_batch_device if theta.dtype == torch.float64 else lib().lmpc_solve_batch_f32_device

        def mask(t, name):
            if t is not None and t.dtype not in _mask_dtypes():
                raise ValueError(f"{name} must be a 64-bit integer tensor")
            return _dev_arg(t, name, t.dtype if t is not None else None, N * self.words, self.device)

        check(fn(
            self._h, N, _vp(theta.data_ptr()), _dev_arg(x, "x", theta.dtype, N * self.nout, self.device, False),
            _dev_arg(exitflag, "exitflag", torch.int32, N, self.device, False),
            _dev_arg(iters, "iters", torch.int32, N, self.device), mask(active, "active"), mask(warm, "warm"),
            _vp(st)), self._h)
        return x, exitflag

    def bind_device_call(self, theta, x, exitflag, stream):
        """`lmpc_solve_batch_device` with every argument validated and marshalled ONCE: returns a callable that
        enqueues the same call again (a loop over a fixed set of resident batches: the per-call Python work --
        tensor checks, ctypes conversions -- is ~5 us, a third of the 18 us a batch takes on the GPU)."""
        import torch
        if not (theta.is_cuda and theta.dtype == torch.float64 and theta.is_contiguous() and theta.dim() == 2
                and theta.shape[1] == self.nth and theta.device.index == self.device):
            raise ValueError("theta must be a contiguous float64 CUDA tensor of shape (N, nth) on this handle's GPU")
        N = int(theta.shape[0])
        args = (self._h, ctypes.c_int64(N), _vp(theta.data_ptr()),
                _dev_arg(x, "x", torch.float64, N * self.nout, self.device, False),
                _dev_arg(exitflag, "exitflag", torch.int32, N, self.device, False), None, None, None, _vp(stream))
        fn = lib().lmpc_solve_batch_device
        h = self._h
        keep = (theta, x, exitflag)

        def call():
            rc = fn(*args)
            if rc != _cabi.LMPC_OK:
                check(rc, h)
            return keep
        return call

    def bind_device_batches(self, thetas, xs, exitflags, stream):
        """`lmpc_solve_batches_device`: several resident batches of the same size in ONE call (on the handles the
        one-launch kernel covers: one kernel launch for up to eight of them).  Arguments validated and marshalled once;
        returns a callable that enqueues the same call again."""
        import torch
        nb = len(thetas)
        if not (nb == len(xs) == len(exitflags) and nb >= 1):
            raise ValueError("as many x and exitflag tensors as theta tensors")
        N = int(thetas[0].shape[0])
        for t in thetas:
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.dim() == 2 and t.shape == (N, self.nth)
                    and t.device.index == self.device):
                raise ValueError("every theta must be a contiguous float64 CUDA tensor of shape (N, nth) on this handle's GPU")
        P = ctypes.c_void_p * nb
        tt = P(*[t.data_ptr() for t in thetas])
        xx = P(*[_dev_arg(x, "x", torch.float64, N * self.nout, self.device, False).value for x in xs])
        ff = P(*[_dev_arg(f, "exitflag", torch.int32, N, self.device, False).value for f in exitflags])
        args = (self._h, ctypes.c_int32(nb), ctypes.c_int64(N), ctypes.cast(tt, ctypes.c_void_p), ctypes.cast(xx, ctypes.c_void_p),
                ctypes.cast(ff, ctypes.c_void_p), _vp(stream))
        fn = lib().lmpc_solve_batches_device
        h = self._h
        keep = (list(thetas), list(xs), list(exitflags), tt, xx, ff)

        def call():
            rc = fn(*args)
            if rc != _cabi.LMPC_OK:
                check(rc, h)
            return keep
        return call

    def solve_batches_device(self, thetas, xs=None, exitflags=None, stream=None):
        """Several resident batches in one call; returns (xs, exitflags).  Enqueued on `stream` (default: torch's current
        stream), not synchronised."""
        import torch
        dev = torch.device("cuda", self.device)
        N = int(thetas[0].shape[0])
        xs = xs or [torch.empty((N, self.nout), dtype=torch.float64, device=dev) for _ in thetas]
        exitflags = exitflags or [torch.empty(N, dtype=torch.int32, device=dev) for _ in thetas]
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        self.bind_device_batches(thetas, xs, exitflags, st)()
        return xs, exitflags

    def simulate(self, x0, T, F, G, r=None, uprev=None, warm=True, want_x=True):
        """Batched closed loop (`lmpc_simulate`): N scenarios, T steps of solve + x <- F x + G u.

        Returns dict(x=final states, U=(T,N,nu), X=(T+1,N,nx) or None, uprev, flag_min)."""
        F = _f64(np.atleast_2d(F))
        nx = F.shape[0]
        nu = self.nout
        G = _f64(np.asarray(G, float).reshape(nx, nu))
        x = _f64(np.array(np.asarray(x0, float).reshape(-1, nx), copy=True))
        N = x.shape[0]
        nr = 0 if r is None else np.asarray(r).reshape(N, -1).shape[1]
        nup = self.nth - nx - nr
        if nup < 0 or nup > nu:
            raise ValueError("theta = [x; r; uprev] does not match this handle")
        rr = None if nr == 0 else _f64(np.asarray(r, float).reshape(N, nr))
        up = None
        if nup:
            up = _f64(np.zeros((N, nup)) if uprev is None else np.array(np.asarray(uprev, float).reshape(N, nup), copy=True))
        U = np.empty((T, N, nu))
        X = np.empty((T + 1, N, nx)) if want_x else None
        fm = np.empty(N, np.int32)
        check(lib().lmpc_simulate(self._h, N, int(T), nx, nr, nup, _ptr(F), _ptr(G), _ptr(x), _ptr(rr) if rr is not None else None,
                                  _ptr(up) if up is not None else None, _ptr(U), _ptr(X) if X is not None else None,
                                  _ptr(fm), int(bool(warm))), self._h)
        return dict(x=x, U=U, X=X, uprev=up, flag_min=fm)

    def simulate_f32(self, x0, T, F, G, r=None, uprev=None, warm=True, want_x=True):
        """Binary32 closed loop (`lmpc_simulate_f32`): float32 arrays in and out, wavefront kernel."""
        F = _f64(np.atleast_2d(F))
        nx = F.shape[0]
        nu = self.nout
        G = _f64(np.asarray(G, float).reshape(nx, nu))
        x = np.ascontiguousarray(np.array(np.asarray(x0, np.float32).reshape(-1, nx), copy=True))
        N = x.shape[0]
        nr = 0 if r is None else np.asarray(r).reshape(N, -1).shape[1]
        nup = self.nth - nx - nr
        if nup < 0 or nup > nu:
            raise ValueError("theta = [x; r; uprev] does not match this handle")
        rr = None if nr == 0 else np.ascontiguousarray(np.asarray(r, np.float32).reshape(N, nr))
        up = None
        if nup:
            up = np.ascontiguousarray(np.zeros((N, nup), np.float32) if uprev is None
                                      else np.array(np.asarray(uprev, np.float32).reshape(N, nup), copy=True))
        U = np.empty((T, N, nu), np.float32)
        X = np.empty((T + 1, N, nx), np.float32) if want_x else None
        fm = np.empty(N, np.int32)
        check(lib().lmpc_simulate_f32(self._h, N, int(T), nx, nr, nup, _ptr(F), _ptr(G), _ptr(x),
                                      _ptr(rr) if rr is not None else None, _ptr(up) if up is not None else None,
                                      _ptr(U), _ptr(X) if X is not None else None, _ptr(fm), int(bool(warm))), self._h)
        return dict(x=x, U=U, X=X, uprev=up, flag_min=fm)

    # ------------------------------------------------------------------ theta on the device, previews
    @staticmethod
    def _block(t, H=0, k0=0):
        """torch CUDA tensor (w,), (w, T) [shared] or (N, w, T) [per scenario] -> lmpc_block.
        The C side wants each matrix column by column (Julia layout), i.e. the (T, w) transpose."""
        import torch
        if t is None:
            return None, None
        if t.dim() == 1:
            t = t.reshape(-1, 1)
        per = t.dim() == 3
        keep = t.transpose(-1, -2).contiguous().to(torch.float64)          # (.., T, w): column after column
        w, T = int(t.shape[-2]), int(t.shape[-1])
        return Block(keep.data_ptr(), w * T if per else 0, w, T, int(k0), int(H)), keep

    def form_parameter_device(self, x, r=None, d=None, uprev=None, p=None, r_preview=0, d_preview=0,
                              p_preview=0, k0=0, theta=None, stream=None):
        """Batched form_parameter (`lmpc_form_parameter_device`; reference explicit.jl:54-63 with
        utils.jl:78-261 formatting).  x: (N, nx) CUDA tensor; r/d/p: (w,), (w, T) shared or (N, w, T);
        *_preview = horizon Np (0 = constant block); k0 = first trajectory column taken."""
        import torch
        N, nx = x.shape
        dev = x.device
        br, kr = self._block(r, r_preview, k0)
        bd, kd = self._block(d, d_preview, k0)
        bp, kp = self._block(p, p_preview, k0)
        nup = 0 if uprev is None else int(uprev.shape[1])
        if theta is None:
            theta = torch.empty((N, self.nth), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        check(lib().lmpc_form_parameter_device(
            self._h, N, _vp(theta.data_ptr()), _vp(x.data_ptr()), nx,
            ctypes.byref(br) if br is not None else None, ctypes.byref(bd) if bd is not None else None,
            _vp(uprev.data_ptr()) if uprev is not None else None, nup,
            ctypes.byref(bp) if bp is not None else None, _vp(st)), self._h)
        theta._lmpc_keep = (kr, kd, kp)          # the launch is asynchronous: keep the sources alive
        return theta

    def simulate_ref(self, x0, T, F, G, r, preview=0, uprev=None, warm=False):
        """Closed loop with a reference trajectory (`lmpc_simulate_ref_device`; reference
        simulation.jl:69-73,93-113).  r: (ny, Tr) shared or (N, ny, Tr); preview = Np for
        settings.reference_preview, 0 otherwise.  Host arrays in and out."""
        import torch
        dev = torch.device("cuda", self.device)
        F = _f64(np.atleast_2d(F))
        nx = F.shape[0]
        nu = self.nout
        G = _f64(np.asarray(G, float).reshape(nx, nu))
        x = torch.from_numpy(_f64(np.array(np.asarray(x0, float).reshape(-1, nx), copy=True))).to(dev)
        N = x.shape[0]
        rt = torch.from_numpy(np.asarray(r, float)).to(dev)
        br, keep = self._block(rt, preview, 0)
        nup = self.nth - nx - br.w * (preview if preview > 0 else 1)
        up = None
        if nup:
            up = torch.from_numpy(_f64(np.zeros((N, nup)) if uprev is None else np.asarray(uprev, float).reshape(N, nup))).to(dev)
        U = torch.empty((T, N, nu), dtype=torch.float64, device=dev)
        X = torch.empty((T + 1, N, nx), dtype=torch.float64, device=dev)
        fm = torch.empty(N, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        check(lib().lmpc_simulate_ref_device(self._h, N, int(T), nx, ctypes.byref(br), nup, _ptr(F), _ptr(G),
                                             _vp(x.data_ptr()), _vp(up.data_ptr()) if up is not None else None,
                                             _vp(U.data_ptr()), _vp(X.data_ptr()), _vp(fm.data_ptr()),
                                             int(bool(warm)), _vp(st)), self._h)
        torch.cuda.synchronize(dev)
        return dict(x=x.cpu().numpy(), U=U.cpu().numpy(), X=X.cpu().numpy(),
                    uprev=None if up is None else up.cpu().numpy(), flag_min=fm.cpu().numpy())

    # ------------------------------------------------------------------ generated-controller entry point
    def set_parameter_layout(self, nx, nr=0, nd=0, nuprev=0, np_=0, preview_horizon=0, traj2setpoint=None):
        """`lmpc_set_parameter_layout`: N_STATE, N_REFERENCE, N_DISTURBANCE, N_CONTROL_PREV,
        N_AFFINE_PARAMETER of the generated header (reference codegen.jl:154-165); with reference
        condensation N_PREVIEW_HORIZON and mpc.traj2setpoint (nr x nr*Np; passed as the generator
        writes it, column by column)."""
        t2s = None
        if preview_horizon:
            t2s = _f64(np.asarray(traj2setpoint, float).reshape(nr, nr * preview_horizon), "F")
        lay = ParamLayout(int(nx), int(nr), int(nd), int(nuprev), int(np_), int(preview_horizon),
                          t2s.ctypes.data if t2s is not None else None)
        check(lib().lmpc_set_parameter_layout(self._h, ctypes.byref(lay)), self._h)
        self._layout = (int(nx), int(nr), int(nd), int(nuprev), int(np_), int(preview_horizon))

    def compute_control(self, control, state, reference=None, disturbance=None, affine_parameter=None, warm=False):
        """`lmpc_compute_control`: the generated `mpc_compute_control(control, state, reference,
        disturbance[, affine_parameter])` (reference codegen/mpc_update_qp.c:29-54) for N problems.
        `control` (N x nu) is read (previous control) and overwritten (u*), as in the C function;
        returns the exit flags."""
        nx, nr, nd, nup, npp, nph = self._layout
        if not (isinstance(control, np.ndarray) and control.dtype == np.float64 and control.flags.c_contiguous
                and control.flags.writeable):
            raise TypeError("control must be a writeable C-contiguous float64 array (it is updated in place)")
        N = control.size // self.nout if self.nout else 0
        control = control.reshape(N, self.nout)

        def arr(a, w):
            if a is None or w == 0:
                return None
            return _f64(np.asarray(a, float).reshape(N, w))

        st, rf = arr(state, nx), arr(reference, nr * (nph if nph else 1))
        ds, pr = arr(disturbance, nd), arr(affine_parameter, npp)
        ef = np.empty(N, np.int32)
        check(lib().lmpc_compute_control(self._h, N, _ptr(control), _ptr(st), _ptr(rf) if rf is not None else None,
                                         _ptr(ds) if ds is not None else None, _ptr(pr) if pr is not None else None,
                                         _ptr(ef), int(bool(warm))), self._h)
        return ef

    def compute_control_device(self, control, state, reference=None, disturbance=None, affine_parameter=None,
                               exitflag=None, warm=False, stream=None):
        """`lmpc_compute_control_device`: CUDA tensors in place, enqueued on the current (or given) stream."""
        import torch
        nx, nr, nd, nup, npp, nph = self._layout
        N = int(control.shape[0])
        dev = control.device
        if exitflag is None:
            exitflag = torch.empty(N, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        f64, d = torch.float64, self.device
        check(lib().lmpc_compute_control_device(
            self._h, N, _dev_arg(control, "control", f64, N * self.nout, d, False),
            _dev_arg(state, "state", f64, N * nx, d, nx == 0),
            _dev_arg(reference, "reference", f64, N * nr * (nph if nph else 1), d),
            _dev_arg(disturbance, "disturbance", f64, N * nd, d),
            _dev_arg(affine_parameter, "affine_parameter", f64, N * npp, d),
            _dev_arg(exitflag, "exitflag", torch.int32, N, d, False), int(bool(warm)), _vp(st)), self._h)
        return exitflag

    def compute_control_observer_device(self, control, observer_state, n_measured=0, reference=None,
                                        measured_disturbance=None, affine_parameter=None, exitflag=None,
                                        warm=False, stream=None):
        """`lmpc_compute_control_observer_device`: generated `mpc_compute_control_observer` of an offset-free
        observer for N scenarios (CUDA tensors; control in place)."""
        import torch
        nx, nr, nd, nup, npp, nph = self._layout
        N = int(control.shape[0])
        dev = control.device
        if exitflag is None:
            exitflag = torch.empty(N, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
        f64, d = torch.float64, self.device
        nm = int(n_measured)
        check(lib().lmpc_compute_control_observer_device(
            self._h, N, _dev_arg(control, "control", f64, N * self.nout, d, False),
            _dev_arg(observer_state, "observer_state", f64, N * (nx + nd - nm), d, False), nm,
            _dev_arg(reference, "reference", f64, N * nr * (nph if nph else 1), d),
            _dev_arg(measured_disturbance, "measured_disturbance", f64, N * nm, d),
            _dev_arg(affine_parameter, "affine_parameter", f64, N * npp, d),
            _dev_arg(exitflag, "exitflag", torch.int32, N, d, False), int(bool(warm)), _vp(st)), self._h)
        return exitflag

    # ------------------------------------------------------------------ generated state observer
    def set_observer(self, plant_dynamics, measurement_function, k_transpose, nx, nu, nd, ny):
        """`lmpc_set_observer`: MPC_PLANT_DYNAMICS, MPC_MEASUREMENT_FUNCTION, K_TRANSPOSE_OBSERVER as the
        reference's code generator writes them (src/observer.jl:124-141)."""
        dyn = _f64(np.asarray(plant_dynamics, float).reshape(-1))
        mea = _f64(np.asarray(measurement_function, float).reshape(-1))
        kt = _f64(np.asarray(k_transpose, float).reshape(-1))
        assert dyn.size == nx * (1 + nx + nu + nd) and mea.size == ny * (1 + nx + nd) and kt.size == ny * nx
        o = Observer(int(nx), int(nu), int(nd), int(ny), dyn.ctypes.data, mea.ctypes.data, kt.ctypes.data)
        check(lib().lmpc_set_observer(self._h, ctypes.byref(o)), self._h)
        self._obs = (int(nx), int(nu), int(nd), int(ny))

    def _observer_call(self, fn_host, fn_dev, state, other, w_other, disturbance, stream):
        if getattr(self, "_obs", None) is None:           # the library says so (LMPC_ERR_BADARG)
            check(fn_host(self._h, 1, None, None, None), self._h)
        nx, nu, nd, ny = self._obs
        if isinstance(state, np.ndarray):
            if not (state.dtype == np.float64 and state.flags.c_contiguous and state.flags.writeable):
                raise TypeError("state must be a writeable C-contiguous float64 array (it is updated in place)")
            N = state.size // nx
            w_other = self._obs[w_other]                  # index into (nx, nu, nd, ny): control or measurement width
            oth = _f64(np.asarray(other, float).reshape(N, w_other)) if w_other else None
            dd = _f64(np.asarray(disturbance, float).reshape(N, nd)) if (disturbance is not None and nd) else None
            check(fn_host(self._h, N, _ptr(state), _ptr(oth) if oth is not None else None,
                          _ptr(dd) if dd is not None else None), self._h)
            return state
        import torch
        N = int(state.shape[0])
        st = torch.cuda.current_stream(state.device).cuda_stream if stream is None else stream
        p = lambda t: _vp(t.data_ptr()) if t is not None else None
        check(fn_dev(self._h, N, p(state), p(other), p(disturbance), _vp(st)), self._h)
        return state

    def predict_state(self, state, control, disturbance=None, stream=None):
        """`lmpc_predict_state[_device]` = generated mpc_predict_state for N scenarios; `state` (N x nx,
        numpy or CUDA tensor) is updated in place."""
        return self._observer_call(lib().lmpc_predict_state, lib().lmpc_predict_state_device, state, control,
                                   1, disturbance, stream)

    def correct_state(self, state, measurement, disturbance=None, stream=None):
        """`lmpc_correct_state[_device]` = generated mpc_correct_state for N scenarios, in place."""
        return self._observer_call(lib().lmpc_correct_state, lib().lmpc_correct_state_device, state, measurement,
                                   3, disturbance, stream)

    # ------------------------------------------------------------------ profiling
    def profile(self, enable=True):
        check(lib().lmpc_profile(self._h, int(bool(enable))), self._h)

    def profile_read(self):
        """(launches, avg ms whole call, avg ms screening kernel, avg ms iterating kernel) since the
        last read; HIP events recorded on the launch stream."""
        ms = (ctypes.c_double * 3)()
        cnt = lib().lmpc_profile_read(self._h, ctypes.byref(ms))
        return cnt, ms[0], ms[1], ms[2]

    def reserve(self, n, stream=None):
        """`lmpc_reserve`: allocate now what the first device call on a batch of n problems would allocate lazily."""
        lib().lmpc_reserve.argtypes = [_vp, ctypes.c_int64, _vp]
        lib().lmpc_reserve.restype = ctypes.c_int
        check(lib().lmpc_reserve(self._h, int(n), _vp(int(stream)) if stream else None), self._h)

    def release_scratch(self):
        """`lmpc_release_scratch`: give the staging / scratch buffers back (they grow with the largest batch)."""
        check(lib().lmpc_release_scratch(self._h), self._h)

    def set_option(self, name: str, value: int):
        check(lib().lmpc_set_option(self._h, name.encode(), int(value)), self._h)

    def distinct_active_sets_device(self, active, exitflag=None, capacity=65536, stream=None):
        """`lmpc_distinct_active_sets_device`: the distinct rows of a solved batch's `active` tensor (N x words int64
        CUDA tensor, as `solve_device(..., active=...)` fills it), reduced ON the device; only the distinct sets come
        back.  Returns (masks (R x words uint64), counts (R), first_index (R)) as numpy arrays sorted by decreasing
        count, ties by first index -- the order `explicit.unique_active_sets` gives."""
        import torch
        dev = active.device
        N = int(active.shape[0])
        if not (active.is_cuda and active.is_contiguous() and active.dtype in _mask_dtypes() and
                active.numel() == N * self.words):
            raise ValueError("active must be a contiguous (N, words) 64-bit integer CUDA tensor")
        if exitflag is not None and not (exitflag.is_cuda and exitflag.dtype == torch.int32 and exitflag.numel() == N):
            raise ValueError("exitflag must be an int32 CUDA tensor of length N")
        st = 0 if stream is None else int(stream)
        while True:
            masks = torch.empty((capacity, self.words), dtype=torch.int64, device=dev)
            counts = torch.empty(capacity, dtype=torch.int64, device=dev)
            first = torch.empty(capacity, dtype=torch.int64, device=dev)
            nset = torch.zeros(1, dtype=torch.int32, device=dev)
            torch.cuda.current_stream(dev).synchronize()         # (buffers above come from torch's stream)
            check(lib().lmpc_distinct_active_sets_device(
                self._h, N, _vp(active.data_ptr()), _vp(exitflag.data_ptr()) if exitflag is not None else None,
                int(capacity), _vp(masks.data_ptr()), _vp(counts.data_ptr()), _vp(first.data_ptr()),
                _vp(nset.data_ptr()), _vp(st) if st else None), self._h)
            over = lib().lmpc_distinct_active_sets_overflowed(self._h, _vp(st) if st else None)
            if over < 0:
                check(over, self._h)
            if over == 0:
                break
            if over == 2:
                raise LmpcError(-102, "lmpc_distinct_active_sets_device: table slot never published")
            capacity *= 4                                        # more regions than room: once more with more
        R = int(nset.item())
        m = masks[:R].cpu().numpy().view(np.uint64)
        c = counts[:R].cpu().numpy()
        f = first[:R].cpu().numpy()
        order = np.lexsort((f, -c))
        return m[order], c[order], f[order]

    def check(self):
        """`lmpc_check`: wait for the handle's GPU, then raise if a kernel reported an internal failure
        (problems with exit flag -8) since the last check."""
        check(lib().lmpc_check(self._h), self._h)

    def close(self):
        if self._h:
            lib().lmpc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiQP:
    """One condensed-MPC QP structure replicated on several GPUs of this process (`lmpc_multi`,
    include/lmpc_hip.h): one call solves a batch across all of them -- the shape of the reference's
    caller, one process with one Theta (/root/reference/src/utils.jl:268-283)."""

    def __init__(self, handle):
        self._hm = handle
        self.ndev = int(lib().lmpc_multi_devices(self._hm))
        self.parts = [BatchedQP(lib().lmpc_multi_handle(self._hm, i), None) for i in range(self.ndev)]
        for q in self.parts:
            q.close = lambda: None                      # the multi handle owns them
        q0 = self.parts[0]
        self.n, self.m, self.ms, self.nth, self.nout, self.words = q0.n, q0.m, q0.ms, q0.nth, q0.nout, q0.words

    @classmethod
    def from_mpqp(cls, H, f, f_theta, A, bu, bl, W, senses=None, nout=None, K=None, nx=0,
                  settings: Settings | None = None, devices=None):
        """`lmpc_setup_multi`; devices = None: every visible GPU."""
        H = _f64(H, "F")
        n = H.shape[0]
        f_theta = _f64(np.asarray(f_theta, float).reshape(n, -1), "F")
        nth = f_theta.shape[1]
        bu = _f64(np.asarray(bu, float).reshape(-1))
        bl = _f64(np.asarray(bl, float).reshape(-1))
        m = bu.size
        A = _f64(np.asarray(A, float).reshape(-1, n), "F")
        ms = m - A.shape[0]
        W = _f64(np.asarray(W, float).reshape(m, nth), "F")
        f = _f64(np.zeros(n) if f is None else np.asarray(f, float).reshape(n))
        nout = n if nout is None else int(nout)
        sense = np.ascontiguousarray(np.zeros(m, np.int32) if senses is None else senses, dtype=np.int32)
        Kf = None if K is None else _f64(np.asarray(K, float).reshape(nout, -1), "F")
        devs = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        h = _vp()
        rc = lib().lmpc_setup_multi(ctypes.byref(h), n, m, ms, nth, nout, _ptr(H), _ptr(f), _ptr(f_theta),
                                    _ptr(A), _ptr(bu), _ptr(bl), _ptr(W), _ptr(sense),
                                    _ptr(Kf) if Kf is not None else None, int(nx if K is not None else 0),
                                    ctypes.cast(ctypes.pointer(settings), _vp) if settings is not None else None,
                                    _ptr(devs) if devs is not None else None, 0 if devs is None else int(devs.size))
        check(rc)
        return cls(h)

    def _check(self, rc):
        if rc != _cabi.LMPC_OK:
            raise LmpcError(rc, (lib().lmpc_multi_last_error(self._hm) or b"").decode())

    def set_option(self, name: str, value: int):
        """`lmpc_multi_set_option` ("transport": 0 RCCL pairs, 1 event-ordered peer copies)."""
        lib().lmpc_multi_set_option.argtypes = [_vp, ctypes.c_char_p, ctypes.c_int]
        lib().lmpc_multi_set_option.restype = ctypes.c_int
        self._check(lib().lmpc_multi_set_option(self._hm, name.encode(), int(value)))

    @staticmethod
    def partition(N, ndev):
        """`lmpc_multi_partition`: shard d = [off[d], off[d+1])."""
        off = (ctypes.c_int64 * (ndev + 1))()
        lib().lmpc_multi_partition(int(N), int(ndev), off)
        return [int(v) for v in off]

    def solve(self, theta, warm=None, want_iters=True, want_active=True):
        """`lmpc_solve_batch_multi`: host arrays in and out, the batch split over the GPUs."""
        theta = _f64(np.asarray(theta, float).reshape(-1, self.nth) if self.nth else np.zeros((len(theta), 0)))
        N = theta.shape[0]
        x = np.zeros((N, self.nout))
        ef = np.zeros(N, np.int32)
        it = np.zeros(N, np.int32) if want_iters else None
        act = np.zeros((N, self.words), np.uint64) if want_active else None
        w = None if warm is None else np.ascontiguousarray(np.asarray(warm, np.uint64).reshape(N, self.words))
        self._check(lib().lmpc_solve_batch_multi(self._hm, N, _ptr(theta), _ptr(x), _ptr(ef),
                                                 _ptr(it) if it is not None else None,
                                                 _ptr(act) if act is not None else None,
                                                 _ptr(w) if w is not None else None))
        return x, ef, it, act

    def solve_device(self, thetas, gather=True):
        """`lmpc_solve_batch_multi_device`: thetas[d] = (N_d, nth) float64 CUDA tensor on device d.
        Returns (x shards, exit-flag shards, x gathered on the first device or None, flags gathered or None)."""
        import torch
        assert len(thetas) == self.ndev
        xs = [torch.empty((t.shape[0], self.nout), dtype=torch.float64, device=t.device) for t in thetas]
        fs = [torch.empty(t.shape[0], dtype=torch.int32, device=t.device) for t in thetas]
        for t in thetas:
            if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.dim() == 2 and t.shape[1] == self.nth):
                raise ValueError("every shard must be a contiguous float64 CUDA tensor of shape (N_d, nth)")
            torch.cuda.synchronize(t.device)            # the library launches on its own streams
        ntot = sum(int(t.shape[0]) for t in thetas)
        xr = torch.empty((ntot, self.nout), dtype=torch.float64, device=thetas[0].device) if gather else None
        fr = torch.empty(ntot, dtype=torch.int32, device=thetas[0].device) if gather else None
        arr = lambda ts: (ctypes.c_void_p * self.ndev)(*[t.data_ptr() for t in ts])
        nd = (ctypes.c_int64 * self.ndev)(*[int(t.shape[0]) for t in thetas])
        self._check(lib().lmpc_solve_batch_multi_device(
            self._hm, ctypes.cast(nd, _vp), ctypes.cast(arr(thetas), _vp), ctypes.cast(arr(xs), _vp),
            ctypes.cast(arr(fs), _vp), _vp(xr.data_ptr()) if gather else None, _vp(fr.data_ptr()) if gather else None))
        return xs, fs, xr, fr

    def close(self):
        if self._hm:
            for q in self.parts:
                q._h = None
            lib().lmpc_free_multi(self._hm)
            self._hm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
