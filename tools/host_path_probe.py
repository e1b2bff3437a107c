import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linearmpc_jl_amd as lmpc
from linearmpc_jl_amd._cabi import lib
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "pendulum.npz")))
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
rng = np.random.default_rng(0)
N = 1_000_000
th = np.ascontiguousarray(np.hstack([rng.uniform([-5, -5, -0.3, -2], [5, 5, 0.3, 2], (N, 4)),
                                     rng.uniform(-5, 5, (N, 1)), np.zeros((N, 1)), rng.uniform(-2, 2, (N, 1))]))
x = np.zeros((N, 1)); ef = np.zeros(N, np.int32)
vp = ctypes.c_void_p
for rep in range(5):
    t0 = time.perf_counter()
    rc = lib().lmpc_solve_batch(qp._h, N, vp(th.ctypes.data), vp(x.ctypes.data), vp(ef.ctypes.data), None, None, None)
    dt = time.perf_counter() - t0
    print(f"C call only: {1e3*dt:.3f} ms  ({N/dt:.3e} solves/s)")
for mode in ("fresh np.empty", "fresh np.empty + fill(0)", "fresh np.zeros"):
    for rep in range(3):
        t0 = time.perf_counter()
        if mode == "fresh np.empty":
            x2 = np.empty((N, 1)); ef2 = np.empty(N, np.int32)
        elif mode == "fresh np.zeros":
            x2 = np.zeros((N, 1)); ef2 = np.zeros(N, np.int32)
        else:
            x2 = np.empty((N, 1)); ef2 = np.empty(N, np.int32); x2.fill(0); ef2.fill(0)
        t1 = time.perf_counter()
        rc = lib().lmpc_solve_batch(qp._h, N, vp(th.ctypes.data), vp(x2.ctypes.data), vp(ef2.ctypes.data), None, None, None)
        dt = time.perf_counter() - t0
        print(f"{mode}: alloc {1e3*(t1-t0):.3f} ms, total {1e3*dt:.3f} ms")
