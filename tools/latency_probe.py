import os, sys, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import linearmpc_jl_amd as lmpc
from linearmpc_jl_amd._cabi import lib
vp = ctypes.c_void_p
for name, nout in (("pendulum", 1), ("soft_doc", 1), ("satellite20", 3)):
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
    th = np.ascontiguousarray(g["theta"][0]); x = np.zeros(nout)
    for _ in range(20):
        lib().lmpc_solve_one(qp._h, vp(th.ctypes.data), vp(x.ctypes.data))
    t0 = time.perf_counter(); n = 300
    for _ in range(n):
        lib().lmpc_solve_one(qp._h, vp(th.ctypes.data), vp(x.ctypes.data))
    print(f"{name}: lmpc_solve_one {1e6*(time.perf_counter()-t0)/n:.1f} us per call ({qp.kernel_name})")

# the generated controller's call (codegen/mpc_update_qp.c:29-54), one state per call, host arrays
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "pendulum.npz")))
q = lmpc.MPQP(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"])
mpc = lmpc.MPC(q, nx=4, nu=1, nr=2, nuprev=1)
gc = lmpc.GeneratedController(mpc)
control = np.zeros((1, 1)); state = np.array([[0.5, 0.1, 0.05, 0.0]]); ref = np.array([[1.0, 0.0]])
for _ in range(20):
    gc.mpc_compute_control(control, state, ref)
t0 = time.perf_counter(); n = 300
for _ in range(n):
    gc.mpc_compute_control(control, state, ref)
print(f"pendulum: mpc_compute_control (N = 1, host arrays) {1e6*(time.perf_counter()-t0)/n:.1f} us per call")
