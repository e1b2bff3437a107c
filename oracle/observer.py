"""CPU restatement of the reference's state observer for the parity tests (TEST INFRASTRUCTURE: only
tests/ may import it; the product path never does).

    KalmanFilter(F, G, C; Q, R)      /root/reference/src/observer.jl:53-72   (steady-state gain from the DARE)
    predict! / correct!              /root/reference/src/observer.jl:104-123
    generated arrays                 /root/reference/src/observer.jl:124-141 (MPC_PLANT_DYNAMICS,
                                     MPC_MEASUREMENT_FUNCTION, K_TRANSPOSE_OBSERVER)
    mpc_predict_state / mpc_correct_state   /root/reference/codegen/mpc_observer.c:1-28

The reference solves the Riccati equation with MatrixEquations.ared (not vendored); scipy's
solve_discrete_are solves the same equation (unique stabilising solution).
"""
from dataclasses import dataclass

import numpy as np
from scipy.linalg import solve_discrete_are


def _weight(w, n):
    w = np.asarray(w, float)
    return np.diag(w) if w.ndim == 1 else w.reshape(n, n)


@dataclass
class KalmanFilter:
    F: np.ndarray
    G: np.ndarray
    Gd: np.ndarray
    f_offset: np.ndarray
    C: np.ndarray
    Dd: np.ndarray
    h_offset: np.ndarray
    K: np.ndarray

    @property
    def dims(self):
        return self.F.shape[0], self.G.shape[1], self.Gd.shape[1], self.C.shape[0]

    # observer.jl:104-108
    def predict(self, x, u, d=None):
        x = self.F @ x + self.G @ u + self.f_offset
        if d is not None and self.Gd.shape[1]:
            x = x + self.Gd @ d
        return x

    # observer.jl:114-118
    def correct(self, x, y, d=None):
        inov = y - self.C @ x - self.h_offset
        if d is not None and self.Dd.shape[1]:
            inov = inov - self.Dd @ d
        return x + self.K @ inov

    # observer.jl:139-141: what the code generator writes
    def codegen_arrays(self):
        dyn = np.hstack([self.f_offset[:, None], self.F, self.G, self.Gd])
        meas = np.hstack([self.h_offset[:, None], self.C, self.Dd])
        return np.ascontiguousarray(dyn).reshape(-1), np.ascontiguousarray(meas).reshape(-1), \
            np.ascontiguousarray(self.K.T).reshape(-1)


def kalman_filter(F, G, C, Gd=None, Dd=None, f_offset=None, h_offset=None, Q=None, R=None) -> KalmanFilter:
    """observer.jl:53-72: P = ared(F', C', R, Q), K = P C' (C P C' + R)^-1."""
    F = np.atleast_2d(np.asarray(F, float))
    nx = F.shape[0]
    G = np.asarray(G, float).reshape(nx, -1)
    C = np.atleast_2d(np.asarray(C, float))
    ny = C.shape[0]
    Gd = np.zeros((nx, 0)) if Gd is None else np.asarray(Gd, float).reshape(nx, -1)
    Dd = np.zeros((ny, 0)) if Dd is None else np.asarray(Dd, float).reshape(ny, -1)
    f_offset = np.zeros(nx) if f_offset is None else np.asarray(f_offset, float).reshape(nx)
    h_offset = np.zeros(ny) if h_offset is None else np.asarray(h_offset, float).reshape(ny)
    Q = np.eye(nx) if Q is None else _weight(Q, nx)
    R = np.eye(ny) if R is None else _weight(R, ny)
    P = solve_discrete_are(F.T, C.T, Q, R)
    K = P @ C.T @ np.linalg.inv(C @ P @ C.T + R)
    return KalmanFilter(F, G, Gd, f_offset, C, Dd, h_offset, K)


def c_predict(dyn, x, u, d, nx, nu, nd):
    """mpc_predict_state (codegen/mpc_observer.c:1-11), loop for loop."""
    xo = np.array(x, float)
    out = np.empty(nx)
    disp = 0
    for i in range(nx):
        acc = dyn[disp]; disp += 1
        for j in range(nx):
            acc += dyn[disp] * xo[j]; disp += 1
        for j in range(nu):
            acc += dyn[disp] * u[j]; disp += 1
        for j in range(nd):
            acc += dyn[disp] * d[j]; disp += 1
        out[i] = acc
    return out


def c_correct(meas, kt, x, y, d, nx, ny, nd):
    """mpc_correct_state (codegen/mpc_observer.c:13-24), loop for loop."""
    xo = np.array(x, float)
    out = xo.copy()
    dc = dk = 0
    for j in range(ny):
        inno = y[j] - meas[dc]; dc += 1
        for i in range(nx):
            inno -= meas[dc] * xo[i]; dc += 1
        for i in range(nd):
            inno -= meas[dc] * d[i]; dc += 1
        for i in range(nx):
            out[i] += kt[dk] * inno; dk += 1
    return out
