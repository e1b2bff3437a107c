#!/usr/bin/env python3
"""Lane kernel or wavefront kernel for the mid-size problems (n = 6 .. 12)?  Pendulum with Nc = n input
bounds (boxed instantiations) and the mass-spring chain with output bounds (general instantiation),
both kernels on the same batch; prints solves/s."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import linearmpc_jl_amd as lmpc
from oracle import mpc2mpqp as omm

dev = torch.device("cuda", 0)
rng = np.random.default_rng(7)


def run(q, theta, wave, reps=5):
    qp = lmpc.BatchedQP.from_mpqp(q.H, q.f, q.f_theta, q.A, q.bu, q.bl, q.W, q.senses, nout=q.nu)
    if wave:
        qp.set_option("wave", 1)
    th = torch.from_numpy(theta).to(dev)
    x = torch.empty((len(theta), q.nu), dtype=torch.float64, device=dev)
    f = torch.empty(len(theta), dtype=torch.int32, device=dev)
    for _ in range(2):
        qp.solve_device(th, x=x, exitflag=f)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        qp.solve_device(th, x=x, exitflag=f)
    torch.cuda.synchronize()
    return len(theta) * reps / (time.perf_counter() - t0), qp.kernel_name, float((f >= 1).float().mean())


N = 400_000
for nc in (6, 8, 10, 12):
    q = omm.mpc2mpqp(omm.pendulum(Np=50, Nc=nc))
    for scale, tag in ((5.0, "mild"), (20.0, "hard")):
        x = rng.uniform(-scale, scale, (N, 4)) * np.array([1, 1, 0.06 if scale == 5 else 1, 0.4 if scale == 5 else 1])
        theta = np.ascontiguousarray(np.hstack([x, rng.uniform(-scale, scale, (N, 1)), np.zeros((N, 1)), rng.uniform(-2, 2, (N, 1))]))
        a = run(q, theta, False)
        b = run(q, theta, True)
        print(f"pendulum Nc={nc:2d} {tag}: {a[1]:14s} {a[0]:.3e}   wave {b[0]:.3e}   solved {a[2]:.2f}")
for nm, np_ in ((4, 10), (6, 10), (6, 8)):
    q = omm.mpc2mpqp(omm.mass_spring(nm=nm, Np=np_, Nc=np_))
    theta = np.ascontiguousarray(rng.uniform(-4, 4, (N, 2 * nm)))
    a = run(q, theta, False)
    b = run(q, theta, True)
    print(f"mass-spring nm={nm} Np={np_} (n={q.n}, m={q.m}): {a[1]:14s} {a[0]:.3e}   wave {b[0]:.3e}   solved {a[2]:.2f}")
