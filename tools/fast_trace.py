#!/usr/bin/env python3
"""Timeline of fast_kernel's wavefronts from a -DLMPC_FAST_TRACE build (LMPC_HIP_LIB) -- diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
import linearmpc_jl_amd as lmpc
g = bench.make_problem("pendulum")
qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
scale = 1.0
for o in sys.argv[1:]:
    k, v = o.split("=")
    if k == "scale": scale = float(v)      # scale=0.5: no point of the batch needs iterations (the stream alone)
    else: qp.set_option(k, int(v))
ths = [torch.from_numpy(bench.make_theta("pendulum", 1000000, 1234 + i) * scale).cuda() for i in range(6)]
for i in range(12):
    x, ef = qp.solve_device(ths[i % 6])
torch.cuda.synchronize()
os.environ["LMPC_FAST_TRACE_FILE"] = "/tmp/fast_trace.bin"
x, ef = qp.solve_device(ths[0])
torch.cuda.synchronize()
t = np.fromfile("/tmp/fast_trace.bin", dtype=np.int64).reshape(-1, 4, 8).astype(np.float64)
# rows by ROLE (role = (wave + workgroup) & 3; roles 0..2 stream, role 3 solves from the start)
t = np.stack([np.roll(t[b], b & 3, axis=0) for b in range(t.shape[0])])
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
us = lambda a: (a - t0) / 100.0
print("workgroups", t.shape[0])
st = us(t[:, :, 0]); print("wave start      us: min %.2f med %.2f max %.2f" % (st.min(), np.median(st), st.max()))
se = us(t[:, :3, 1]); print("stream end      us: min %.2f med %.2f max %.2f" % (se.min(), np.median(se), se.max()))
dur = (t[:, :3, 1] - t[:, :3, 0]) / 100.0; print("stream duration us: min %.2f med %.2f max %.2f" % (dur.min(), np.median(dur), dur.max()))
c1 = t[:, :, 2]; m = c1 > 0; print("first claim     us: min %.2f med %.2f max %.2f (waves with a claim: %d)" % (us(c1[m]).min(), np.median(us(c1[m])), us(c1[m]).max(), m.sum()))
en = us(t[:, :, 6]); print("wave end        us: min %.2f med %.2f max %.2f" % (en.min(), np.median(en), en.max()))
npass = t[:, :, 7]; print("passes per wave: solver %.2f streamers %.2f ; per workgroup %.2f" % (npass[:, 3].mean(), npass[:, :3].mean(), npass.sum(1).mean()))
wgend = us(t[:, :, 6]).max(1); wgst = st.min(1)
print("workgroup life  us: med %.2f max %.2f ; last stream end -> wg end: med %.2f max %.2f" % (np.median(wgend - wgst), (wgend - wgst).max(), np.median(wgend - se.max(1)), (wgend - se.max(1)).max()))
p2 = t[:, 3, 3]; m2 = p2 > 0
print("solver wave: claim1 -> claim2 us: med %.2f" % np.median((p2[m2] - t[:, 3, 2][m2]) / 100.0))
s_ = t[:, 3, :]; ok = (s_[:, 2] > 0) & (s_[:, 4] > 0) & (s_[:, 5] > 0) & (s_[:, 3] > 0)
print("solver wave, first pass: claim -> records loaded and shifts formed %.2f us, -> tiers done %.2f us, -> next claim %.2f us (medians)" % (
    np.median((s_[ok, 4] - s_[ok, 2]) / 100.0), np.median((s_[ok, 5] - s_[ok, 4]) / 100.0), np.median((s_[ok, 3] - s_[ok, 5]) / 100.0)))
