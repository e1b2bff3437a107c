"""Per-phase shader-clock shares of the wavefront kernel (Gram-scan form, binary64), from a -DLMPC_WAVE_TRACE build
of that translation unit (LMPC_HIP_LIB=_ab/lib_wtrace.so).  usage: python tools/wave_trace.py workload[:batch] ..."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import linearmpc_jl_amd as lmpc
L = lmpc.lib()
names = ["setup (b = Dth theta)", "y re-sweep after a removal", "backward sweep (lam*)", "blocking test", "fval + Gram scan",
         "violation test + selection", "row append (ldl_add)", "row removal (ldl_remove)", "singular direction + its blocking",
         "exit: primal step, outputs, clear", "iterations", "problems"]
dev = torch.device("cuda:0")
for spec in sys.argv[1:] or ["mass_spring_3in:200000", "pendulum_N100:200000", "soft_doc:200000"]:
    name, _, n = spec.partition(":")
    n = int(n or 200000)
    W = bench.Workload(torch, lmpc, name, n, dev, 0, 0, 1, rotate=False, options={"gram_scan": 1})
    out = (ctypes.c_ulonglong * 16)()
    W.timed(1, 1, nstreams=1)
    L.lmpc_debug_wave_trace(out, 1)
    sec = W.timed(2, 0, nstreams=1) / 2
    L.lmpc_debug_wave_trace(out, 1)
    v = np.array(list(out), float)
    tot = v[:10].sum()
    print(f"== {name} N={n}: {sec*1e3:.2f} ms per call with the stamps; {v[11]/2:.0f} problems in the wavefront kernel per call, "
          f"{v[10]/max(v[11],1):.1f} iterations each, {tot/max(v[10],1):.0f} stamped cycles per iteration")
    for k in range(10):
        print(f"   {names[k]:38s} {100*v[k]/tot:5.1f} %   {v[k]/max(v[10],1):8.0f} cycles per iteration")
    W.close()
