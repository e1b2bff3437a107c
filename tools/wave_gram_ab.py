"""Same-box A/B of the wavefront kernel's two forms (lmpc_set_option "gram_scan" 0 | 1) on the wavefront-kernel
workloads of bench.py: device-resident batches, one stream, one call at a time.

    python tools/wave_gram_ab.py [workload[:batch][:f32] ...]
"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import linearmpc_jl_amd as lmpc

dev = torch.device("cuda:0")
specs = sys.argv[1:] or ["mass_spring_3in:1000000", "hybrid:100000:f32", "pendulum_N50:200000", "pendulum_N75:200000",
                         "pendulum_N100:200000", "pendulum_N125:200000", "soft_doc:200000"]
for spec in specs:
    parts = spec.split(":")
    name = parts[0]
    n = int(parts[1]) if len(parts) > 1 else 200000
    f32 = len(parts) > 2 and parts[2] == "f32"
    row = []
    outs = []
    for gram in (0, 1):
        opts = {"gram_scan": gram}
        if os.environ.get("LMPC_WAVE_CAP"):          # working-set rows held per problem (lmpc_set_option wave_cap)
            opts["wave_cap"] = int(os.environ["LMPC_WAVE_CAP"])
        W = bench.Workload(torch, lmpc, name, n, dev, 0, 0, 1, f32=f32, rotate=False, options=opts)
        sec = W.timed(3, 1, nstreams=1) / 3
        torch.cuda.synchronize()
        outs.append((W.xbuf[0].cpu().numpy().copy(), W.fbuf[0].cpu().numpy().copy()))
        row.append((sec, n / sec))
        W.close()
    (x0, f0), (x1, f1) = outs
    ok = f0 >= 1
    dx = np.abs(x0[ok] - x1[ok]).max() if ok.any() else 0.0
    print(f"{name:16s} {'f32' if f32 else 'f64'} N={n:8d}  chain {row[0][0]*1e3:9.3f} ms {row[0][1]:.3e}/s | gram {row[1][0]*1e3:9.3f} ms "
          f"{row[1][1]:.3e}/s  x{row[1][1]/row[0][1]:.2f} | solved {ok.mean():.3f} flags differ {(f0 != f1).sum()} "
          f"(of them solved-status flips {((f0 >= 1) != (f1 >= 1)).sum()}) max|dx| {dx:.2e}", flush=True)
