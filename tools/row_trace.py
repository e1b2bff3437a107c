"""Per-phase shader-clock shares of the row kernel from a -DLMPC_ROW_TRACE build:
   tools/ab_row_build.sh trace -DLMPC_ROW_TRACE && LMPC_HIP_LIB=linearmpc.jl_amd/lib/ab/lib_trace.so python tools/row_trace.py [name] [N]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import linearmpc_jl_amd as lmpc
from conftest import load_golden
L = lmpc.lib()
names = ["take a problem (b = Dth theta)", "stationary point (sweeps)", "blocking test", "primal step", "soft slack + constraint scan",
         "violation test + selection", "row append", "removal: compaction + shifts", "removal: rank-one update", "outputs, clean-up"]
name = sys.argv[1] if len(sys.argv) > 1 else "mass_spring_3in"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
g = load_golden(name)
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=int(g["nu"]) if "nu" in g else None)
qp.set_option("row_kernel", 1)
th = torch.from_numpy(bench.make_theta(name, N, 77)).cuda()
out = (ctypes.c_ulonglong * 32)()
qp.solve_device(th); torch.cuda.synchronize()
L.lmpc_debug_row_trace(out, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); qp.solve_device(th); b.record(); torch.cuda.synchronize()
L.lmpc_debug_row_trace(out, 1)
v = np.array(list(out), float)
tot = v[:10].sum() + v[15]
trips = max(v[10], 1)
print(f"== {name} N={N}: {a.elapsed_time(b):.2f} ms per call with the stamps; {v[13]:.0f} problems, {trips:.0f} wavefront trips "
      f"({4*trips/max(v[13],1):.1f} per problem x 4 rows), {100*v[11]/trips:.0f} % with an append phase, {100*v[12]/trips:.0f} % with a removal phase; "
      f"{tot/trips:.0f} stamped cycles per trip")
for k in range(10):
    print(f"   {names[k]:38s} {100*v[k]/tot:5.1f} %   {v[k]/trips:8.0f} cycles per trip")
print(f"   {'loop overhead':38s} {100*v[15]/tot:5.1f} %   {v[15]/trips:8.0f} cycles per trip")
