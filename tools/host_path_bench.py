#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (what a LinearMPC.jl caller with Theta in host
memory gets): lmpc_solve_batch on one GPU for several chunk sizes, pinned in place or not, and
lmpc_solve_batch_multi over every visible GPU.  Outputs are preallocated and touched."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import linearmpc_jl_amd as lmpc

g = bench.make_problem("pendulum")
N = 1_000_000
theta = bench.make_theta("pendulum", N, 1234)
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)


import ctypes
L = lmpc.lib()
x = np.zeros((N, 1)); ef = np.zeros(N, np.int32)          # preallocated, pages touched
vp = lambda a: ctypes.c_void_p(a.ctypes.data)


def call_single():
    rc = L.lmpc_solve_batch(qp._h, N, vp(theta), vp(x), vp(ef), None, None, None)
    assert rc == 1, rc


def timeit(fn, reps=12):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts)


qp.set_option("host_threads", 0)
t = timeit(call_single)
print(f"lmpc_solve_batch   pageable, one chunk, one thread:      {1e3*t:7.3f} ms per 1e6 = {N/t:.3e} solves/s", flush=True)
qp.set_option("host_threads", 1)
for chunk in (262144, 131072, 65536, 32768, 16384):
    qp.set_option("host_chunk", chunk)
    t = timeit(call_single)
    print(f"lmpc_solve_batch   pageable, two host threads, last chunk {chunk:7d}: {1e3*t:7.3f} ms per 1e6 = {N/t:.3e} solves/s", flush=True)
import mmap


def paged(a):          # page-aligned copy with pages of its own
    b = np.frombuffer(mmap.mmap(-1, a.nbytes), dtype=a.dtype).reshape(a.shape)
    b[...] = a
    return b


theta, x, ef = paged(theta), paged(x), paged(ef)
qp.set_option("host_chunk", 32768)
for a_ in (theta, x, ef):
    assert L.lmpc_pin_host(vp(a_), a_.nbytes) == 1
t = timeit(call_single)
print(f"lmpc_solve_batch   arrays pinned once by the caller:    {1e3*t:7.3f} ms per 1e6 = {N/t:.3e} solves/s", flush=True)
for a_ in (theta, x, ef):
    L.lmpc_unpin_host(vp(a_))
mq = lmpc.MultiQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
def call_multi():
    rc = L.lmpc_solve_batch_multi(mq._hm, N, vp(theta), vp(x), vp(ef), None, None, None)
    assert rc == 1, rc


t = timeit(call_multi)
print(f"lmpc_solve_batch_multi over {mq.ndev} device(s): {1e3*t:7.3f} ms per 1e6 = {N/t:.3e} solves/s")
