#!/usr/bin/env python3
"""Probe: which sequences of pageable copies / hipHostRegister on overlapping pages upset the HIP runtime."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import linearmpc_jl_amd as lmpc
case = sys.argv[1]
g = bench.make_problem("pendulum")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
L = lmpc.lib()
N = 30011
buf = np.zeros(N * 7 + N + N // 2 + 64)           # one allocation: theta | x | ef share pages at their seams
theta = buf[:N * 7].reshape(N, 7); theta[...] = bench.make_theta("pendulum", N, 1)
x = buf[N * 7:N * 8].reshape(N, 1)
ef = buf[N * 8:N * 8 + N // 2 + 1].view(np.int32)[:N]
vp = lambda a: ctypes.c_void_p(a.ctypes.data)
call = lambda: L.lmpc_solve_batch(qp._h, N, vp(theta), vp(x), vp(ef), None, None, None)
if case == "pageable_then_register":
    qp.set_option("host_register", 0); print("pageable", call(), flush=True)
    qp.set_option("host_register", 1); print("registered", call(), flush=True)
elif case == "register_twice":
    qp.set_option("host_register", 1); print(call(), call(), flush=True)
elif case == "register_then_pageable":
    qp.set_option("host_register", 1); print("registered", call(), flush=True)
    qp.set_option("host_register", 0); print("pageable", call(), flush=True)
elif case == "torch_then_register":
    import torch
    t = torch.from_numpy(buf).cuda(); back = t.cpu(); buf2 = back.numpy()
    print("torch roundtrip ok", flush=True)
    qp.set_option("host_register", 1); print("registered", call(), flush=True)
print("done", case, flush=True)
