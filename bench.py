#!/usr/bin/env python3
"""Headline benchmark: condensed-MPC QP solves per second, pendulum Nc=5, batch 1e6 (f64).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (lmpc_solve_batch_device: constraint shift, dual
active-set solve, primal recovery) over one batch of 1e6 synthetic parameter points per GPU,
already resident in HBM.  With N > 1 every rank (one process per GPU) owns its own 1e6-point
shard (weak scaling).  The solve needs no data-path collective: shards are independent and their
results stay on the GPU that produced them; the one exchange step -- an RCCL gather of the
solutions and exit flags to rank 0 over xGMI (every rank sends on its own link) -- happens once,
after the last step, inside the timed region (--gather step all-gathers after every step,
overlapped with the next solve; at 12 MB per rank and step that exchange is ~10x longer than the
30 us solve it follows, so it is not the default).  Consecutive steps are
independent batches; by default three of them are kept in flight on three HIP streams (each with
its own solver handle), which lets the streaming pass of one batch overlap the latency-bound
iterating pass of another (--streams 1 serialises them).

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 1_000_000
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_problem(name):
    """mpQP of the benchmark problem.  The condensing step is not part of the timed path (it stays
    on the LinearMPC.jl host); the committed golden fixture carries the matrices."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz")))
    return g


def make_theta(name, n, seed, hard=False):
    """Synthetic parameter points, SURVEY.md section 8(d) / BASELINE.md sampling."""
    rng = np.random.default_rng(seed)
    if name == "pendulum":
        if hard:      # the example's +-20 ParameterRange (reference mpc_examples.jl:128-134)
            x = rng.uniform(-20, 20, (n, 4))
            r = rng.uniform(-20, 20, (n, 1))
        else:
            x = rng.uniform([-5, -5, -0.3, -2], [5, 5, 0.3, 2], (n, 4))
            r = rng.uniform(-5, 5, (n, 1))
        return np.ascontiguousarray(np.hstack([x, r, np.zeros((n, 1)), rng.uniform(-2, 2, (n, 1))]))
    if name in ("mass_spring", "mass_spring_3in"):
        return np.ascontiguousarray(rng.uniform(-4, 4, (n, 12)))
    if name == "satellite20":   # hybrid MPC, theta = [x(3); r(3)] (reference mpc_examples.jl:533-546, runtests.jl:820-834)
        return np.ascontiguousarray(np.hstack([rng.uniform(-0.3, 0.3, (n, 1)), rng.uniform(-0.5, 0.5, (n, 2)),
                                               rng.uniform(-0.5, 0.5, (n, 1)), np.zeros((n, 2))]))
    if name == "soft_doc":      # docs example with soft output bounds (reference docs/src/manual/simple.md:60-83)
        return np.ascontiguousarray(np.hstack([rng.uniform(-1, 2, (n, 2)), rng.uniform(0, 1, (n, 2)),
                                               rng.uniform(-3, 3, (n, 1))]))
    raise ValueError(name)


def algorithmic_bytes(nth, nout, real_bytes=8):
    # read theta (8*nth) + write x (8*nout) + write exit flag (4); SURVEY.md section 8(d)
    return real_bytes * nth + real_bytes * nout + 4


def cpu_baseline(g, theta, nout, min_seconds=10.0, f32=False, all_cores_seconds=5.0):
    """Single-thread CPU oracle (the restated DAQP algorithm) on a bounded sample of the same batch
    (about 10-20 s of CPU work); a reported baseline, not the product path."""
    from oracle import ldp as oldp
    dt_ = np.float32 if f32 else np.float64
    L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
    probe = min(1000, theta.shape[0])
    t0 = time.perf_counter()
    oldp.solve_batch(L, theta[:probe], dtype=dt_)
    per = (time.perf_counter() - t0) / probe
    # sample: the leading rows of the batch, sized so that one pass takes ~2 s at most
    ns = int(min(theta.shape[0], max(probe, 2.0 / max(per, 1e-9))))
    sample = theta[:ns]
    t0 = time.perf_counter()
    passes = 0
    while True:
        oldp.solve_batch(L, sample, dtype=dt_)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or passes >= 200:
            break
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {"value": passes * ns / dt, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": f"{passes} passes over the first {ns} points of the same batch, 1 thread of "
                     f"{os.cpu_count()} ({model}), oracle/daqp_ldp_oracle.c ({'binary32' if f32 else 'binary64'} build) -O2 -mfma"}
    # the same oracle on all host cores this process may use: the sample cut
    # into one contiguous slice per thread (the C call releases the GIL), ~5 s
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 64))
    if ncores > 1 and all_cores_seconds > 0:
        from concurrent.futures import ThreadPoolExecutor
        big = theta[:min(theta.shape[0], max(ns, 4096 * ncores))]
        parts = [p_ for p_ in np.array_split(big, ncores) if len(p_)]
        with ThreadPoolExecutor(len(parts)) as pool:
            list(pool.map(lambda p_: oldp.solve_batch(L, p_[:256], dtype=dt_), parts))   # spin the threads up
            t0 = time.perf_counter()
            passes = 0
            while True:
                list(pool.map(lambda p_: oldp.solve_batch(L, p_, dtype=dt_), parts))
                passes += 1
                dta = time.perf_counter() - t0
                if dta >= all_cores_seconds or passes >= 200:
                    break
        out["all_cores"] = {"value": passes * big.shape[0] / dta, "unit": "solves/s", "cores": len(parts),
                            "sample": f"{passes} passes over the first {big.shape[0]} points, one contiguous slice per thread"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="pendulum", choices=["pendulum", "pendulum_hard", "mass_spring", "mass_spring_3in", "soft_doc", "hybrid"])
    ap.add_argument("--f32", action="store_true",
                    help="binary32 path (lmpc_solve_batch_f32_device; wavefront kernel; reference codegen float_type=float)")
    ap.add_argument("--wave", action="store_true", help="force the wavefront-per-QP kernel (diagnostic)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", default="final", choices=["final", "step", "none"],
                    help="N > 1: RCCL gather of x and exit flags to rank 0 once after the last step (default), "
                         "an all-gather after every step (overlapped with the next solve), or never")
    ap.add_argument("--no-single-launch", action="store_true",
                    help="skip the one-call-at-a-time section after the timed region (profiling runs: keeps the "
                         "kernel trace to the launches of the timed region)")
    ap.add_argument("--no-screen", action="store_true", help="iterating kernel only (diagnostic)")
    ap.add_argument("--ablate", type=int, default=0, help="timing-only ablation bits for the screening kernel")
    ap.add_argument("--lane-per", type=int, default=0, help="work-list workgroups per shard (tuning)")
    ap.add_argument("--lane-tier", type=int, default=-1, help="lane kernel: first-tier capacity on (1) / off (0) (tuning)")
    ap.add_argument("--lane-block", type=int, default=0, help="lane-kernel workgroup size (tuning)")
    ap.add_argument("--wave-level", type=int, default=-1, help="wave kernel: LDS staging level 0..3 (tuning)")
    ap.add_argument("--wave-nwv", type=int, default=0, help="wave kernel: wavefronts per workgroup (tuning)")
    ap.add_argument("--wave-cap", type=int, default=0, help="wave kernel: wavefronts per CU of the grid (tuning)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches kept in flight per GPU (each has its own handle and HIP stream)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import linearmpc_jl_amd as lmpc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # test hooks (single-GPU box): LMPC_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
    # LMPC_BENCH_BACKEND=gloo swaps RCCL for gloo so the N > 1 control flow can be exercised there
    if os.environ.get("LMPC_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("LMPC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    name = "pendulum" if args.workload.startswith("pendulum") else ("satellite20" if args.workload == "hybrid" else args.workload)
    hard = args.workload == "pendulum_hard"
    g = make_problem(name)
    nout = int(g["nu"])
    nstreams = max(1, min(8, args.streams))
    tdt = torch.float32 if args.f32 else torch.float64
    qps = [lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"],
                                    g["senses"], nout=nout, device=local_rank,
                                    settings=lmpc.default_settings_f32() if args.f32 else None) for _ in range(nstreams)]
    qp = qps[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    stream_handles = [s_.cuda_stream for s_ in streams]
    if args.no_screen:
        for q_ in qps:
            q_.set_option("screen", 0)
    if args.wave:
        for q_ in qps:
            q_.set_option("wave", 1)
    if args.lane_per:
        for q_ in qps:
            q_.set_option("lane_per", args.lane_per)
    if args.wave_cap:
        for q_ in qps:
            q_.set_option("wave_cap", args.wave_cap)
    if args.wave_level >= 0:
        for q_ in qps:
            q_.set_option("wave_level", args.wave_level)
    if args.wave_nwv:
        for q_ in qps:
            q_.set_option("wave_nwv", args.wave_nwv)
    # several batches in flight: 64-lane workgroups for the iterating kernel (its wavefronts then spread
    # over the CUs independently of each other; +3 % over the library's stand-alone choice of 256,
    # which is the better one with a single batch in flight: tools/block_sweep.sh)
    if not args.lane_block and nstreams > 1 and not args.f32 and not args.wave and args.workload.startswith("pendulum"):
        args.lane_block = 64
    if args.lane_block:
        for q_ in qps:
            q_.set_option("lane_block", args.lane_block)
    if args.lane_tier >= 0:
        for q_ in qps:
            q_.set_option("lane_tier", args.lane_tier)
    if args.ablate:
        for q_ in qps:
            q_.set_option("ablate", args.ablate)
    n_local = args.batch
    theta_h = make_theta(name, n_local, 1234 + rank, hard)
    theta = torch.from_numpy(theta_h).to(dev).to(tdt)

    # double-buffered outputs so the gather of step k overlaps the solve of step k+1
    nbuf = max(2, nstreams)
    xbuf = [torch.empty((n_local, nout), dtype=tdt, device=dev) for _ in range(nbuf)]
    fbuf = [torch.empty(n_local, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    do_gather = world > 1 and args.gather == "step"
    final_gather = world > 1 and args.gather == "final"
    if final_gather:
        # gather to rank 0: every rank sends its shard straight to the root over its own xGMI link
        # (ncclSend/ncclRecv pairs), N-1 links in parallel -- a ring all-gather would move N-1 shards
        # through every link and nobody but the root reads them
        xfin = torch.empty((world * n_local, nout), dtype=tdt, device=dev) if rank == 0 else None
        ffin = torch.empty(world * n_local, dtype=torch.int32, device=dev) if rank == 0 else None

    gather_impl = {"mode": "gather"}

    def gather_to_root(xs, fs):
        if gather_impl["mode"] == "gather":
            dist.gather(xs, list(xfin.split(n_local)) if rank == 0 else None, dst=0)
            dist.gather(fs, list(ffin.split(n_local)) if rank == 0 else None, dst=0)
        else:                                    # fallback chosen during warm-up, see below
            dist.all_gather_into_tensor(xfin, xs)
            dist.all_gather_into_tensor(ffin, fs)
    if do_gather:
        xall = [torch.empty((world * n_local, nout), dtype=tdt, device=dev) for _ in range(nbuf)]
        fall = [torch.empty(world * n_local, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    pending = [None] * nbuf

    def step(k):
        b = k % nbuf
        # step k runs on stream k % nstreams with that stream's own handle (work list, counters):
        # consecutive steps are independent batches, so the streaming pass of one overlaps the
        # latency-bound iterating pass of the other
        sidx = k % nstreams
        if not do_gather:                        # single GPU: raw stream handle, no context switch
            qps[sidx].solve_device(theta, x=xbuf[b], exitflag=fbuf[b], stream=stream_handles[sidx])
            return
        with torch.cuda.stream(streams[sidx]):
            if pending[b] is not None:           # buffer b is free once its gather has finished
                for w in pending[b]:
                    w.wait()
                pending[b] = None
            qps[sidx].solve_device(theta, x=xbuf[b], exitflag=fbuf[b])
            if do_gather:
                pending[b] = (dist.all_gather_into_tensor(xall[b], xbuf[b], async_op=True),
                              dist.all_gather_into_tensor(fall[b], fbuf[b], async_op=True))

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                for w in pending[b]:
                    w.wait()
                pending[b] = None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    drain()
    if final_gather:
        # untimed: the first all-gather of this shape sets up RCCL's channels and buffers
        torch.cuda.synchronize(dev)
        try:
            gather_to_root(xbuf[0], fbuf[0])
            torch.cuda.synchronize(dev)
            ok = torch.ones(1, device=dev)
        except RuntimeError:                     # a backend without gather: every rank takes all shards instead
            ok = torch.zeros(1, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            gather_impl["mode"] = "all_gather"
            xfin = torch.empty((world * n_local, nout), dtype=tdt, device=dev)
            ffin = torch.empty(world * n_local, dtype=torch.int32, device=dev)
            gather_to_root(xbuf[0], fbuf[0])
    fence()
    # Device time over the timed region: ONE pair of HIP events per launch stream brackets all of
    # that stream's launches (events between every two kernels cost ~12 % throughput: each is an extra
    # packet the queue has to retire in order).  Per-kernel durations come from the single-launch
    # section below, which records events around every kernel.
    per_call_events = os.environ.get("LMPC_BENCH_CALL_EVENTS") == "1"
    for q_ in qps:
        q_.profile(per_call_events)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    t0 = time.perf_counter()
    for e_, s_ in zip(ev0, streams):
        e_.record(s_)
    for k in range(args.steps):
        step(k)
    for e_, s_ in zip(ev1, streams):
        e_.record(s_)
    enqueue_s = time.perf_counter() - t0          # host time to issue all steps (diagnostic)
    drain()
    if final_gather and args.steps:
        # the one exchange step of the sharded job: every rank receives all shards' solutions
        torch.cuda.synchronize(dev)
        lastb = (args.steps - 1) % nbuf
        gather_to_root(xbuf[lastb], fbuf[lastb])
    fence()
    elapsed = time.perf_counter() - t0
    prof = [q_.profile_read() for q_ in qps] if per_call_events else []
    for q_ in qps:
        q_.profile(False)
    # average device time of one launch on its stream = event span of the stream / its launches
    calls_on = [len(range(i_, args.steps, nstreams)) for i_ in range(nstreams)]
    span_ms = [ev0[i_].elapsed_time(ev1[i_]) for i_ in range(nstreams)]
    stream_call_ms = sum(span_ms) / max(sum(calls_on), 1)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    flags = fbuf[(args.steps - 1) % nbuf].cpu().numpy() if args.steps else np.zeros(0, np.int32)
    # the same call with NOTHING else on the chip (outside the timed region): one launch after the
    # other on one stream, HIP events on that stream -- the per-launch duration rocprofv3's kernel
    # trace reports; with several batches in flight the launches above overlap and a single launch
    # no longer owns the GPU
    solo = None
    if rank == 0 and args.steps and not args.no_single_launch:
        nsolo = int(min(200, max(20, args.steps)))
        torch.cuda.synchronize(dev)
        qp.profile(True)
        for _ in range(nsolo):
            qp.solve_device(theta, x=xbuf[0], exitflag=fbuf[0], stream=stream_handles[0])
        torch.cuda.synchronize(dev)
        solo = qp.profile_read()
        qp.profile(False)
    # distribution of the work (untimed extra solve): throughput depends on how many iterations the
    # batch needs, so the histograms travel with the number
    it_d = torch.empty(n_local, dtype=torch.int32, device=dev)
    ac_d = torch.zeros((n_local, qp.words), dtype=torch.int64, device=dev)
    qp.solve_device(theta, x=xbuf[0], exitflag=fbuf[0], iters=it_d, active=ac_d)
    torch.cuda.synchronize(dev)
    it_hist = torch.bincount(it_d.clamp(max=31).to(torch.int64), minlength=2).cpu().tolist()
    bits = ac_d.cpu().numpy().view(np.uint64)
    nact = np.zeros(n_local, np.int64)
    for w_ in range(bits.shape[1]):
        v_ = bits[:, w_].copy()
        while v_.any():
            nact += (v_ & np.uint64(1)).astype(np.int64)
            v_ >>= np.uint64(1)
    act_hist = np.bincount(nact, minlength=1).tolist()
    mean_it = float(it_d.to(torch.float64).mean().item())
    # algorithmic FP64 work per solve: per iteration the scan 2mn, the primal
    # step 2|W|n and the triangular solves 2|W|^2 (final |W| as a stand-in), plus the affine maps
    flop_est = float(np.mean(it_d.cpu().numpy() * (2.0 * qp.m * qp.n) +
                             it_d.cpu().numpy() * (2.0 * nact * qp.n + 2.0 * nact * nact))
                     + 2.0 * qp.m * qp.nth + 2.0 * nout * qp.nth)
    if rank == 0:
        total = world * n_local * args.steps
        value = total / elapsed
        bytes_per = algorithmic_bytes(qp.nth, nout, 4 if args.f32 else 8)
        # One batch in flight: algorithmic bytes of a call / its device time (HIP events on the launch
        # stream).  Several batches in flight: their launches overlap on the chip, a single launch no
        # longer owns it, so the bytes one step moves are divided by the wall time one step takes
        # (about kernel_ms / batches_in_flight; kernel_ms stays in the record for the rocprof check).
        step_ms = 1e3 * elapsed / max(args.steps, 1)
        kern_ms, screen_ms, iterate_ms = (solo[1], solo[2], solo[3]) if solo else (0.0, 0.0, 0.0)
        nlaunch = args.steps
        dur_ms = stream_call_ms if nstreams == 1 else step_ms
        achieved = (bytes_per * n_local) / (dur_ms * 1e-3) / 1e9 if dur_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        out = {
            "metric": "condensed-MPC QP solves/sec (batch 1e6 params), pendulum Nc=5"
                      if args.workload == "pendulum" else f"condensed-MPC QP solves/sec ({args.workload})",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.f32 else "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: "
                       + ("inverted pendulum on cart, 4 states / 1 input, Np=50 Nc=5 "
                          "(n=5 vars, 5 two-sided input bounds, theta=[x;r;u_prev] nth=7), "
                          if name == "pendulum" else ("mass-spring chain nm=6, Np=Nc=10 (n=10, m=63, nth=12), "
                                                      if name == "mass_spring" else f"{name} (n={qp.n}, m={qp.m}, nth={qp.nth}), "))
                       + f"{n_local} parameter points per GPU, cold start, first move u0 returned",
                       "batch_per_gpu": n_local, "kernel": "wave" if args.f32 else qp.kernel_name, "batches_in_flight": nstreams,
                       "gather": ("all_gather(x, exitflag) over RCCL after every step, overlapped" if do_gather
                                  else (gather_impl["mode"] + "(x, exitflag) to rank 0 over RCCL once, after the last step") if final_gather
                                  else "none"),
                       "solved_fraction": float((flags >= 1).mean()) if flags.size else None,
                       "iterations_hist": it_hist, "mean_iterations": mean_it,
                       "active_set_size_hist": act_hist},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "stream_call_ms": stream_call_ms,
                         "kernel_ms": kern_ms, "screen_kernel_ms": screen_ms,
                         "iterate_kernel_ms": iterate_ms, "launches_timed": nlaunch,
                         "duration_used_ms": dur_ms, "host_enqueue_ms_per_step": 1e3 * enqueue_s / max(args.steps, 1),
                         "algorithmic_bytes_per_solve": bytes_per,
                         "fp64_flop_per_solve_est": flop_est,
                         "fp64_tflops_est": flop_est * value / 1e12,
                         "solves_per_s_per_cu": value / world / 256.0},
        }
        if solo is not None and solo[0] > 0 and solo[1] > 0:
            ach1 = (bytes_per * n_local) / (solo[1] * 1e-3) / 1e9
            out["roofline"]["single_launch"] = {
                "launches_timed": solo[0], "kernel_ms": solo[1], "screen_kernel_ms": solo[2],
                "iterate_kernel_ms": solo[3], "achieved": ach1, "frac": ach1 / HBM_PEAK_GBS,
                "note": "one call at a time on one stream (after the timed region): HIP-event duration of a "
                        "single launch, the figure rocprofv3 --kernel-trace reports per kernel"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g, theta_h, nout, f32=args.f32)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
