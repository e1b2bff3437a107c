"""Per-phase shader-clock shares of the row kernel's branch and bound (binary32) from a -DLMPC_ROW_TRACE build:
   tools/ab_row_bnb_build.sh trace -DLMPC_ROW_TRACE && LMPC_HIP_LIB=linearmpc.jl_amd/lib/ab/lib_trace.so python tools/row_bnb_trace.py [N]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import linearmpc_jl_amd as lmpc
from conftest import load_golden
L = lmpc.lib()
names = ["take a problem (b = Dth theta)", "stationary point (sweeps)", "blocking test", "primal step", "soft slack + constraint scan",
         "violation test + selection", "gather for the append", "removal: snapshots, compaction, shifts", "removal: rank-one + row append", "outputs, clean-up"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = load_golden("satellite20")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], settings=lmpc.default_settings_f32())
qp.set_option("row_kernel", 1)
th = torch.from_numpy(bench.make_theta("satellite20", N, 77).astype(np.float32)).cuda()
out = (ctypes.c_ulonglong * 32)()
it = torch.empty(N, dtype=torch.int32, device="cuda")
qp.solve_device(th, iters=it); torch.cuda.synchronize()
L.lmpc_debug_row_trace(out, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); qp.solve_device(th, iters=it); b.record(); torch.cuda.synchronize()
L.lmpc_debug_row_trace(out, 1)
v = np.array(list(out), float)
tot = v[:10].sum() + v[14] + v[15]
trips = max(v[10], 1)
print(f"== satellite20 f32 N={N}: {a.elapsed_time(b):.2f} ms per call with the stamps; {v[13]:.0f} problems, {trips:.0f} wavefront trips "
      f"({4*trips/max(v[13],1):.1f} per problem x 4 rows; {it.double().mean().item():.1f} iterations per problem), {100*v[11]/trips:.0f} % with an add phase, "
      f"{100*v[12]/trips:.0f} % with a removal phase; {tot/trips:.0f} stamped cycles per trip")
for k in range(10):
    print(f"   {names[k]:42s} {100*v[k]/tot:5.1f} %   {v[k]/trips:8.0f} cycles per trip")
print(f"   {'search controller (node ends)':42s} {100*v[14]/tot:5.1f} %   {v[14]/trips:8.0f} cycles per trip")
print(f"   {'loop overhead':42s} {100*v[15]/tot:5.1f} %   {v[15]/trips:8.0f} cycles per trip")
print(f"   per trip: {v[16]/trips:.2f} branchings ({v[22]/max(v[16],1):.0f} cycles each), pop loops {v[23]/trips:.0f} cycles, "
      f"{v[17]/trips:.2f} clean / {v[18]/trips:.2f} dirty restores ({v[20]/max(v[17]+v[18],1):.0f} cycles each), "
      f"{v[19]/trips:.2f} lazy saves ({v[21]/max(v[19],1):.0f} cycles each)")
print(f"   restores with a copy from memory: {v[24]/max(v[18],1):.0f} cycles each; without: {v[25]/max(v[26],1):.0f} cycles each ({v[26]/trips:.2f} per trip)")
print(f"   rank-one updates: {v[27]/trips:.0f} cycles per trip, {v[27]/max(v[12],1):.0f} per removal phase ({v[28]/max(v[12],1):.1f} columns from the lowest removed row to the largest working set)")
