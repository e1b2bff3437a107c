#!/usr/bin/env python3
"""Closed-loop batch simulation throughput (lmpc_simulate_device): N scenarios x T steps of
[form theta -> solve -> plant step] on the device.  Prints scenario-steps per second."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import linearmpc_jl_amd as lmpc
from linearmpc_jl_amd._cabi import lib, check
import ctypes

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warm", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--async", dest="asyn", type=int, default=1, help="scenario-asynchronous closed loop (1; 2 = also on the wavefront-kernel path) or lock-step (0)")
ap.add_argument("--blind", type=int, default=-1, help="rounds enqueued between two counter reads (library default if < 0)")
ap.add_argument("--traj", type=int, default=0, help="also record the U (T x N x nu) and X ((T+1) x N x nx) trajectories")
ap.add_argument("--small", type=int, default=1, help="all-in-registers streaming kernel (1) or the general one (0)")
ap.add_argument("--fused", type=int, default=1, help="plant step inside the solve kernels (1) or as its own kernel (0)")
ap.add_argument("--groups", type=int, default=1,
                help="split the scenarios into this many groups, each with its own handle, HIP stream and host thread "
                     "(scenarios are independent: the groups' kernels overlap on the chip)")
ap.add_argument("--problem", default="pendulum",
                help="golden fixture: pendulum (lane kernels) or pendulum_N50/75/100/125 (long horizon with state "
                     "constraints: wavefront kernel, lock-step loop)")
ap.add_argument("--screen-wave", type=int, default=1, help="screening pass in front of the wavefront kernel")
ap.add_argument("--settled", type=float, default=0.0, help="fraction of the scenarios that start at rest on their reference (no step of theirs needs iterations)")
ap.add_argument("--gram", type=int, default=0, help="wavefront kernel: Gram-scan form (lmpc_set_option gram_scan)")
ap.add_argument("--wave-cap", type=int, default=0, help="wavefront kernel: working-set rows held per problem (lmpc_set_option wave_cap; 0 = library default)")
ap.add_argument("--two-pass", type=int, default=-1, help="wavefront kernel: first pass at a smaller working-set capacity (lmpc_set_option wave_two_pass; -1 = by the handle's statistics)")
ap.add_argument("--keep", type=int, default=1, help="wavefront path, warm: keep every scenario's factorisation between two steps (lmpc_set_option sim_keep_factor)")
a = ap.parse_args()
g = dict(np.load(os.path.join(ROOT, "tests", "golden", a.problem + ".npz")))
sys.path.insert(0, ROOT)
if "F" in g:
    class prob: F, G = g["F"], g["G"]
else:
    from oracle import mpc2mpqp as omm
    prob = omm.pendulum()
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
print("kernel:", qp.kernel_name)
if qp.kernel_name.endswith("wave"):
    qp.set_option("screen_wave", a.screen_wave)
    qp.set_option("gram_scan", a.gram)
    qp.set_option("sim_keep_factor", a.keep)
    if a.wave_cap: qp.set_option("wave_cap", a.wave_cap)
    qp.set_option("wave_two_pass", a.two_pass)
qp.set_option("sim_fused", a.fused)
qp.set_option("sim_async", a.asyn)
qp.set_option("sim_small", a.small)
if a.blind >= 0:
    qp.set_option("sim_blind", a.blind)
rng = np.random.default_rng(0)
N, T = a.n, a.steps
dev = torch.device("cuda", 0)
if "n_closed_loop" in g:      # the benchmark class: start from points one closed loop visits, perturbed
    base = g["theta"][:int(g["n_closed_loop"])]
    pick = base[rng.integers(0, len(base), N)] + rng.normal(size=(N, 7)) * [0.02, 0.05, 0.005, 0.05, 0.02, 0.0, 0.0]
    if a.settled > 0:
        ns = int(a.settled * N)
        pick[:ns, :4] = 0.0; pick[:ns, 4:6] = 0.0; pick[:ns, 6] = 0.0
    x0 = torch.from_numpy(np.ascontiguousarray(pick[:, :4])).to(dev)
    r = torch.from_numpy(np.ascontiguousarray(pick[:, 4:6])).to(dev)
    up0 = torch.from_numpy(np.ascontiguousarray(pick[:, 6:7])).to(dev)
else:
    x0 = torch.from_numpy(rng.uniform([-5, -5, -0.3, -2], [5, 5, 0.3, 2], (N, 4))).to(dev)
    r = torch.from_numpy(np.hstack([rng.uniform(-5, 5, (N, 1)), np.zeros((N, 1))])).to(dev)
    up0 = torch.zeros((N, 1), dtype=torch.float64, device=dev)
F = np.ascontiguousarray(prob.F); G = np.ascontiguousarray(prob.G)
vp = ctypes.c_void_p
fm = torch.empty(N, dtype=torch.int32, device=dev)
Ut = torch.empty((T, N, 1), dtype=torch.float64, device=dev) if a.traj else None
Xt = torch.empty((T + 1, N, 4), dtype=torch.float64, device=dev) if a.traj else None
for rep in range(a.reps):
    x = x0.clone(); up = up0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    check(lib().lmpc_simulate_device(qp._h, N, T, 4, 2, 1, vp(F.ctypes.data), vp(G.ctypes.data), vp(x.data_ptr()),
                                     vp(r.data_ptr()), vp(up.data_ptr()), vp(Ut.data_ptr()) if a.traj else None,
                                     vp(Xt.data_ptr()) if a.traj else None, vp(fm.data_ptr()), a.warm,
                                     vp(torch.cuda.current_stream(dev).cuda_stream)), qp._h)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N} T={T} warm={a.warm}: {N*T/(t2-t0):.3e} scenario-steps/s, {1e6*(t2-t0)/T:.1f} us/step "
          f"(host enqueue {1e6*(t1-t0)/T:.1f} us/step), min flag {int(fm.min())}"
          + (f", wave stats {qp.wave_stats()}" if qp.kernel_name.endswith("wave") else ""))


if a.groups > 1:
    import threading
    Gn = a.groups
    qps = [lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
           for _ in range(Gn)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(Gn)]
    bounds = np.linspace(0, N, Gn + 1).astype(int)
    for rep in range(a.reps):
        xs = [x0[bounds[i]:bounds[i + 1]].clone() for i in range(Gn)]
        rs = [r[bounds[i]:bounds[i + 1]].contiguous() for i in range(Gn)]
        ups = [torch.zeros((bounds[i + 1] - bounds[i], 1), dtype=torch.float64, device=dev) for i in range(Gn)]
        fms = [torch.empty(bounds[i + 1] - bounds[i], dtype=torch.int32, device=dev) for i in range(Gn)]
        torch.cuda.synchronize()

        def run(i):
            n_i = int(bounds[i + 1] - bounds[i])
            check(lib().lmpc_simulate_device(qps[i]._h, n_i, T, 4, 2, 1, vp(F.ctypes.data), vp(G.ctypes.data),
                                             vp(xs[i].data_ptr()), vp(rs[i].data_ptr()), vp(ups[i].data_ptr()), None, None,
                                             vp(fms[i].data_ptr()), a.warm, vp(streams[i].cuda_stream)), qps[i]._h)

        t0 = time.perf_counter()
        ths = [threading.Thread(target=run, args=(i,)) for i in range(Gn)]
        for t_ in ths:
            t_.start()
        for t_ in ths:
            t_.join()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"N={N} T={T} warm={a.warm} groups={Gn}: {N*T/(t2-t0):.3e} scenario-steps/s, {1e6*(t2-t0)/T:.1f} us/step, "
              f"min flag {min(int(f_.min()) for f_ in fms)}")
