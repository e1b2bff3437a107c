#!/usr/bin/env python3
"""Headline benchmark: condensed-MPC QP solves per second, pendulum Nc=5, batch 1e6 (f64).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (lmpc_solve_batch_device: constraint shift, dual
active-set solve, primal recovery) over one batch of 1e6 synthetic parameter points per GPU,
already resident in HBM.  Consecutive steps are DIFFERENT batches: the bench rotates through
enough distinct theta / output buffers that more than 256 MiB passes between two uses of a line,
so no step is served from the 256 MiB Infinity Cache ("cold HBM"; the cache-resident figures --
one theta buffer reused every step -- are reported next to it, under roofline.cache_resident).

With N > 1 every rank (one process per GPU) owns its own 1e6-point shard (weak scaling).  The
solve needs no data-path collective: shards are independent and their results stay on the GPU
that produced them, so the timed region holds none.  What a single consumer of all shards would
add -- an RCCL gather of one batch's solutions and exit flags to rank 0 over xGMI (every rank
sends on its own link) -- is timed behind the timed region and reported as config.exchange
(--gather final puts it inside, --gather step all-gathers after every step, overlapped).
By default three batches are kept in flight on three HIP streams (each with its own solver
handle): the streaming pass of one batch overlaps the latency-bound iterating pass of another.

Rank 0 prints a COMPACT JSON line (< 4 KB; compact_line(); the driver keeps only a tail of stdout) -- once as soon as
the headline is measured, and again, with the per-config summaries, as the LAST line of stdout; the full report
(every histogram, verification block and note) goes to bench_full.json and to stderr:
  value / ms_per_step   whole-job throughput of the timed region (cold HBM, batches in flight)
  roofline              ONE call at a time on one stream, cold HBM: algorithmic bytes of a call /
                        the HIP-event duration of that call; pipelined_frac = the same bytes / ms_per_step
  cpu_baseline          the C oracle on the host cores (kind "port": libdaqp is not available)
  configs               the other single-GPU BASELINE.json configurations, measured in the same process, one summary
                        each {value, verified[, unit unless solves/s][, frac[, bound unless valu]][, gram_value, gram_frac, gram_frac_executed], cpu_1core
                        [, first_run_value]}: pendulum_hard, mass_spring (the reference's example), mass_spring_3in
                        (config 3) and its feasible-dominated companion, game_avi (is_avi), hybrid_f32 (config 5),
                        pendulum_N50..125 (the reference's benchmark class), three closed loops, region_discovery (config 4)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 1_000_000
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}   # dense vector peaks (MI355X_MICROARCH.md)
L3_BYTES = 256 * 1024 * 1024     # Infinity Cache


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
        if hard == "feasible":   # feasible-dominated companion of SURVEY 8(d)'s prescribed +-4 sample
            return np.ascontiguousarray(rng.uniform(-1.5, 1.5, (n, 12)))
        return np.ascontiguousarray(rng.uniform(-4, 4, (n, 12)))
    if name == "satellite20":   # hybrid MPC, theta = [x(3); r(3)] (reference mpc_examples.jl:533-546, runtests.jl:820-834)
        return np.ascontiguousarray(np.hstack([rng.uniform(-0.3, 0.3, (n, 1)), rng.uniform(-0.5, 0.5, (n, 2)),
                                               rng.uniform(-0.5, 0.5, (n, 1)), np.zeros((n, 2))]))
    if name.startswith("pendulum_N"):   # the reference's benchmark class: points a closed loop visits, perturbed
        g = make_problem(name)
        base = g["theta"][:int(g["n_closed_loop"])]
        return np.ascontiguousarray(base[rng.integers(0, len(base), n)] +
                                    rng.normal(size=(n, 7)) * [0.02, 0.05, 0.005, 0.05, 0.02, 0.0, 0.05])
    if name == "soft_doc":      # docs example with soft output bounds (reference docs/src/manual/simple.md:60-83)
        return np.ascontiguousarray(np.hstack([rng.uniform(-1, 2, (n, 2)), rng.uniform(0, 1, (n, 2)),
                                               rng.uniform(-3, 3, (n, 1))]))
    raise ValueError(name)


def algorithmic_bytes(nth, nout, real_bytes=8):
    # read theta (8*nth) + write x (8*nout) + write exit flag (4); SURVEY.md section 8(d)
    return real_bytes * nth + real_bytes * nout + 4


def rotation_depth(n_local, bytes_per):
    """Distinct batches to rotate through so that > 1.25 x 256 MiB passes between two uses of a line."""
    per_batch = max(1, n_local * bytes_per)
    return int(min(64, max(2, math.ceil(1.25 * L3_BYTES / per_batch) + 1)))


_T0 = time.perf_counter()
_NATIVE_FLAGS = None


def _phase(msg):
    """Progress line on stderr (the default run takes a few minutes: GPU legs, then CPU baselines per config)."""
    print(f"[bench {time.perf_counter() - _T0:6.1f} s] {msg}", file=sys.stderr, flush=True)


def _effective_cpus():
    """Cores this process may really use: the affinity mask capped by the cgroup's CPU quota (a GPU box hands a
    job a share of its host -- 16 of 256 logical CPUs for one GPU -- through the quota, not through the mask)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as fh:
                q, per = parse(fh.read())
            if per is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    per = fh.read().strip()
            if q not in ("max", "-1") and float(per) > 0:
                n = max(1, min(n, int(math.ceil(float(q) / float(per)))))
            break
        except (OSError, ValueError):
            continue
    return n


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def cpu_baseline(g, theta, nout, min_seconds=10.0, f32=False, all_cores_seconds=5.0, marginal=False):
    """The CPU oracle (the restated DAQP algorithm; `kind: port` -- libdaqp itself is not available
    here or on the GPU box) on a bounded sample of the same batch: one thread, then every core this
    process may use.  Built with -O3 -march=native on the machine that runs it.  A reported baseline,
    not the product path."""
    from oracle import ldp as oldp
    global _NATIVE_FLAGS
    if _NATIVE_FLAGS is None:
        _NATIVE_FLAGS = oldp.use_native()          # compiled once per process
    flags = _NATIVE_FLAGS
    dt_ = np.float32 if f32 else np.float64
    L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
    probe = min(256, theta.shape[0])
    t0 = time.perf_counter()
    oldp.solve_batch(L, theta[:probe], dtype=dt_)
    per = (time.perf_counter() - t0) / probe
    # sample: the leading rows of the batch, sized so that one pass takes ~1 s at most
    ns = int(min(theta.shape[0], max(probe, 1.0 / max(per, 1e-9))))
    run1 = oldp.runner(L, theta[:ns], dtype=dt_)
    t0 = time.perf_counter()
    passes = 0
    while True:
        run1()
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or passes >= 100000:
            break
    mrep = oldp.marginal_report(L, theta[:min(theta.shape[0], 1000000)]) if (marginal and not f32) else None
    out = {"value": passes * ns / dt, "unit": "solves/s", "cores": 1, "kind": "port",
           "sample": f"{passes} passes over the first {ns} points of the same batch, 1 thread of "
                     f"{os.cpu_count()} ({_cpu_model()}), oracle/daqp_ldp_oracle.c "
                     f"({'binary32' if f32 else 'binary64'} build) {flags}"}
    # the same oracle on all host cores this process may use: the sample cut
    # into one contiguous slice per thread; every thread makes `reps` bare C calls over its slice (the C
    # call releases the GIL), all threads start together and the slowest one ends the measurement
    ncores = _effective_cpus()
    if ncores > 1 and all_cores_seconds > 0:
        from concurrent.futures import ThreadPoolExecutor
        rate1 = out["value"]
        # bounded: at most the points all cores get through in about all_cores_seconds
        nall = int(min(theta.shape[0], max(ncores * 32, rate1 * 0.6 * ncores * all_cores_seconds)))
        theta = theta[:nall]
        parts = [p_ for p_ in np.array_split(theta, ncores) if len(p_)]
        runs = [oldp.runner(L, p_, dtype=dt_) for p_ in parts]
        reps = int(max(1, all_cores_seconds * rate1 * 0.6 / max(len(parts[0]), 1)))   # (all cores busy: lower clocks)
        with ThreadPoolExecutor(len(parts)) as pool:
            list(pool.map(lambda r_: r_(1), runs))              # spin the threads up, touch the outputs
            t0 = time.perf_counter()
            list(pool.map(lambda r_: r_(reps), runs))
            dta = time.perf_counter() - t0
        out["all_cores"] = {"value": reps * theta.shape[0] / dta, "unit": "solves/s", "cores": len(parts),
                            "sample": f"{reps} passes over the first {theta.shape[0]} points, one contiguous slice "
                                      f"per thread, {len(parts)} threads = the CPUs this job may use "
                                      f"(affinity mask capped by the cgroup quota; the host has {os.cpu_count()})"}
    if mrep is not None:
        out["marginal_cases"] = mrep
    return out


class Workload:
    """One benchmark configuration resident on one GPU: solver handles (one per stream), a rotation of
    distinct theta batches and output buffers."""

    def __init__(self, torch, lmpc, workload, n_local, dev, local_rank, rank, nstreams, f32=False, rotate=True,
                 options=None):
        self.torch = torch
        self.workload = workload
        self.name = ("pendulum" if workload in ("pendulum", "pendulum_hard") else
                     ("satellite20" if workload == "hybrid" else workload.replace("_feasible", "")))
        self.hard = "feasible" if workload.endswith("_feasible") else workload == "pendulum_hard"
        self.f32 = f32
        self.g = make_problem(self.name)
        self.nout = int(self.g["nu"])
        self.n_local = n_local
        self.dev = dev
        self.tdt = torch.float32 if f32 else torch.float64
        self.nstreams = nstreams
        g = self.g
        self.qps = [lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"],
                                             g["senses"], nout=self.nout, device=local_rank,
                                             settings=lmpc.default_settings_f32() if f32 else None)
                    for _ in range(nstreams)]
        for q_ in self.qps:
            self.options = dict(options or {})
            for k_, v_ in (options or {}).items():
                q_.set_option(k_, v_)
        self.qp = self.qps[0]
        for q_ in self.qps:
            q_.reserve(n_local)             # (setup: scratch for this batch size allocated before anything is timed)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        self.stream_handles = [s_.cuda_stream for s_ in self.streams]
        self.bytes_per = algorithmic_bytes(self.qp.nth, self.nout, 4 if f32 else 8)
        self.nrot = rotation_depth(n_local, self.bytes_per) if rotate else 1
        self.theta_h = make_theta(self.name, n_local, 1234 + rank, self.hard)
        self.thetas = [torch.from_numpy(self.theta_h).to(dev).to(self.tdt)]
        for r_ in range(1, self.nrot):
            self.thetas.append(torch.from_numpy(make_theta(self.name, n_local, 7919 * r_ + 1234 + rank, self.hard))
                               .to(dev).to(self.tdt))
        self._bound = {}
        self.nbuf = max(2, nstreams, self.nrot)
        self.xbuf = [torch.empty((n_local, self.nout), dtype=self.tdt, device=dev) for _ in range(self.nbuf)]
        self.fbuf = [torch.empty(n_local, dtype=torch.int32, device=dev) for _ in range(self.nbuf)]
        if not f32:          # every (handle, batch, buffer) combination a run can meet, bound before any timing
            for k in range(math.lcm(nstreams, self.nrot, self.nbuf)):
                key = (k % nstreams, k % self.nrot, k % self.nbuf)
                self._bound[key] = self.qps[key[0]].bind_device_call(self.thetas[key[1]], self.xbuf[key[2]], self.fbuf[key[2]],
                                                                     self.stream_handles[key[0]])

    @property
    def kernel(self):
        return "wave" if self.f32 else self.qp.kernel_name

    def launch(self, k, sidx=None, resident=False):
        """Enqueue step k on its stream (raw stream handle: no torch context switch).  The batches, buffers and
        streams are a fixed set, so each (handle, batch, buffer) call is validated and marshalled once."""
        sidx = k % self.nstreams if sidx is None else sidx
        r_ = 0 if resident else k % self.nrot
        b = (k % max(2, self.nstreams)) if resident else (k % self.nbuf)
        key = (sidx, r_, b)
        call = self._bound.get(key)
        if call is None:
            if self.f32:
                call = (lambda q_=self.qps[sidx], th=self.thetas[r_], xb=self.xbuf[b], fb=self.fbuf[b], st=self.stream_handles[sidx]:
                        q_.solve_device(th, x=xb, exitflag=fb, stream=st))
            else:
                call = self.qps[sidx].bind_device_call(self.thetas[r_], self.xbuf[b], self.fbuf[b], self.stream_handles[sidx])
            self._bound[key] = call
        call()
        return b

    def timed(self, steps, warmup, resident=False, nstreams=None):
        """`steps` launches over the first `nstreams` streams, bracketed by device synchronisation; returns
        wall seconds.  (The headline's own timed region lives in main(): it adds the barrier and the gather.)"""
        torch = self.torch
        ns = self.nstreams if nstreams is None else nstreams
        for k in range(warmup):
            self.launch(k, k % ns, resident)
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        for k in range(steps):
            self.launch(k, k % ns, resident)
        torch.cuda.synchronize(self.dev)
        return time.perf_counter() - t0

    def single_launch(self, ncalls, resident=False):
        """One call at a time on ONE stream with HIP events around the call and between its two kernels
        (recorded by the library on the launch stream): the per-launch durations rocprofv3's kernel trace
        reports.  Returns (calls, call ms, screening-kernel ms, iterating-kernel ms)."""
        torch = self.torch
        # the one-launch kernel's shape options set for several batches in flight do not suit a call issued alone:
        # this section runs the library's own defaults and puts the others back afterwards
        shaped = {k_: v_ for k_, v_ in self.options.items() if k_ in ("fast_nstr", "fast_tiles", "in_flight")}
        for k_ in shaped:
            self.qp.set_option(k_, 1 if k_ == "in_flight" else 0)
        for k in range(4):
            self.launch(k, 0, resident)
        torch.cuda.synchronize(self.dev)
        self.qp.profile(True)
        for k in range(ncalls):
            self.launch(k, 0, resident)
        torch.cuda.synchronize(self.dev)
        solo = self.qp.profile_read()
        self.qp.profile(False)
        for k_, v_ in shaped.items():
            self.qp.set_option(k_, v_)
        return solo

    def several_batches_section(self, nb=8, ncalls=24):
        """lmpc_solve_batches_device on ONE stream: calls of `nb` batches each, back to back, over a rotation of nb distinct
        cold batches and nb output buffers; event-timed as a run (torch events on the launch stream).  On the headline's
        handle one call is ONE kernel launch (fast_kernel_multi) in which the solving tail of a batch runs under the
        stream of the next.  Returns the per-call / per-batch times and the oracle check of what the last call wrote."""
        torch = self.torch
        if self.f32 or not hasattr(self.qp, "bind_device_batches"):
            return None
        thetas = list(self.thetas)
        while len(thetas) < nb:
            thetas.append(torch.from_numpy(make_theta(self.name, self.n_local, 104729 * len(thetas) + 77, self.hard)).to(self.dev).to(self.tdt))
        xs = [torch.empty((self.n_local, self.nout), dtype=self.tdt, device=self.dev) for _ in range(nb)]
        fs = [torch.empty(self.n_local, dtype=torch.int32, device=self.dev) for _ in range(nb)]
        st = self.streams[0]
        call = self.qp.bind_device_batches(thetas[:nb], xs, fs, st.cuda_stream)
        shaped = {k_: v_ for k_, v_ in self.options.items() if k_ in ("fast_nstr", "fast_tiles", "in_flight")}
        for k_ in shaped:
            self.qp.set_option(k_, 1 if k_ == "in_flight" else 0)
        for _ in range(3):
            call()
        torch.cuda.synchronize(self.dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        for _ in range(ncalls):
            call()
        b.record(st)
        torch.cuda.synchronize(self.dev)
        call_ms = a.elapsed_time(b) / ncalls
        for k_, v_ in shaped.items():
            self.qp.set_option(k_, v_)
        # the last call's outputs against the oracle (a sample of every batch)
        from oracle import ldp as oldp
        pk = self.qp.ldp()
        L = oldp.LDP(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["M"], pk["du"], pk["dl"], pk["Dth"],
                     pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"]))
        rng = np.random.default_rng(5)
        nflag, worst, npts = 0, 0.0, 0
        for q in range(nb):
            idx = torch.from_numpy(np.sort(rng.choice(self.n_local, min(1024, self.n_local), replace=False))).to(self.dev)
            xo, efo, _, _ = oldp.solve_batch(L, thetas[q][idx].cpu().numpy())
            ef = fs[q][idx].cpu().numpy()
            nflag += int((ef != efo).sum())
            ok = efo >= 1
            if ok.any():
                worst = max(worst, float(np.abs(xs[q][idx].cpu().numpy()[ok] - xo[ok]).max()))
            npts += len(idx)
        by = self.bytes_per * self.n_local
        return {"batches_per_call": nb, "calls_timed": ncalls, "call_ms": call_ms, "ms_per_batch": call_ms / nb,
                "value": nb * self.n_local / (call_ms * 1e-3), "achieved": nb * by / (call_ms * 1e-3) / 1e9,
                "frac": nb * by / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s",
                "verified": bool(nflag == 0 and worst <= 1e-10), "points_checked": npts,
                "how": f"lmpc_solve_batches_device, {nb} cold batches per call = ONE launch of fast_kernel_multi (gridDim.y = {nb}); "
                       "calls back to back on ONE stream, event-timed as a run; algorithmic bytes of a call / its duration"}

    def verify_steps(self, ks, per_step=4096, seed=99):
        """Output check behind a timed region: for each step k in ks, a seeded sample of the buffers that step wrote
        against the CPU oracle on the same parameter points (the oracle is the checker here, nothing timed).  Returns
        a dict with `verified` (all sampled points bit-identical in exit flag, <= 1e-10 in x; binary32: the binary32
        oracle, 1e-6)."""
        from oracle import ldp as oldp
        torch = self.torch
        pk = self.qp.ldp()
        L = oldp.LDP(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["M"], pk["du"], pk["dl"], pk["Dth"],
                     pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"]))
        so = oldp.default_settings_f32() if self.f32 else oldp.default_settings()
        so.mode = 1 if self.options.get("gram_scan") else 0
        rng = np.random.default_rng(seed)
        worst, nflag, npts = 0.0, 0, 0
        for k in ks:
            b = k % self.nbuf
            r_ = k % self.nrot
            idx = torch.from_numpy(np.sort(rng.choice(self.n_local, min(per_step, self.n_local), replace=False))).to(self.dev)
            th = self.thetas[r_][idx].cpu().numpy()
            x = self.xbuf[b][idx].cpu().numpy()
            ef = self.fbuf[b][idx].cpu().numpy()
            xo, efo, _, _ = oldp.solve_batch(L, th, so, dtype=np.float32 if self.f32 else np.float64)
            nflag += int((ef != efo).sum())
            ok = efo >= 1
            if ok.any():
                worst = max(worst, float(np.abs(x[ok] - xo[ok]).max()))
            npts += len(idx)
        tol = 1e-6 if self.f32 else 1e-10
        return {"verified": bool(nflag == 0 and worst <= tol), "points": npts, "steps_checked": [int(k) for k in ks],
                "exit_flag_mismatches": nflag, "max_abs_dx": worst, "tolerance": tol,
                "against": "oracle/daqp_ldp_oracle.c (mode %d) on the same parameter points, after the timed region" % so.mode}

    def work_distribution(self):
        """Iteration and active-set-size histograms of batch 0 (untimed extra solve) and the algorithmic
        flop estimate of SURVEY.md section 8d: per iteration the scan 2mn, the primal step 2|W|n and the
        triangular solves 2|W|^2 (final |W| as a stand-in), plus the two affine maps."""
        torch = self.torch
        qp = self.qp
        it_d = torch.empty(self.n_local, dtype=torch.int32, device=self.dev)
        ac_d = torch.zeros((self.n_local, qp.words), dtype=torch.int64, device=self.dev)
        qp.solve_device(self.thetas[0], x=self.xbuf[0], exitflag=self.fbuf[0], iters=it_d, active=ac_d)
        torch.cuda.synchronize(self.dev)
        flags = self.fbuf[0].cpu().numpy()
        it = it_d.cpu().numpy().astype(np.int64)
        bits = ac_d.cpu().numpy().view(np.uint64)
        nact = np.zeros(self.n_local, np.int64)
        for w_ in range(bits.shape[1]):
            v_ = bits[:, w_].copy()
            while v_.any():
                nact += (v_ & np.uint64(1)).astype(np.int64)
                v_ >>= np.uint64(1)
        flop = float(np.mean(it * (2.0 * qp.m * qp.n) + it * (2.0 * nact * qp.n + 2.0 * nact * nact))
                     + 2.0 * qp.m * qp.nth + 2.0 * self.nout * qp.nth)
        # what the Gram-scan form EXECUTES for the same solves: per iteration the scan over Gram columns 2 m |W| and
        # the sweeps 2 |W|^2, the primal step 2 |W| n once at the end, plus the affine maps
        self.flop_gram_executed = float(np.mean(it * (2.0 * qp.m * nact + 2.0 * nact * nact) + 2.0 * nact * qp.n)
                                        + 2.0 * qp.m * qp.nth + 2.0 * self.nout * qp.nth)
        return {"solved_fraction": float((flags >= 1).mean()),
                "iterations_hist": np.bincount(np.minimum(it, 31), minlength=2).tolist() if it.max() < 4000 else None,
                "mean_iterations": float(it.mean()), "max_iterations": int(it.max()),
                "active_set_size_hist": np.bincount(nact, minlength=1).tolist()}, flop

    def close(self):
        self._bound = {}
        for q_ in self.qps:
            q_.close()
        self.thetas = self.xbuf = self.fbuf = None


def describe(w):
    qp = w.qp
    if w.name == "pendulum":
        body = ("inverted pendulum on cart, 4 states / 1 input, Np=50 Nc=5 (n=5 vars, 5 two-sided input bounds, "
                "theta=[x;r;u_prev] nth=7)" + (", parameters from the example's +-20 range" if w.hard else ""))
    elif w.name == "mass_spring":
        body = ("the reference's mass_spring example (mpc_examples.jl:241-286): nm=6, 1 input, Np=Nc=10 "
                f"(n={qp.n}, m={qp.m}, nth={qp.nth}), x ~ U(-4,4)^12")
    elif w.name == "mass_spring_3in":
        body = (f"oscillating masses, 12 states / 3 inputs (synthetic B), Nc=10 (n={qp.n}, m={qp.m}, nth={qp.nth}), "
                + ("x ~ U(-1.5,1.5)^12 (feasible-dominated companion sample)" if w.hard == "feasible" else "x ~ U(-4,4)^12 (SURVEY 8d)"))
    elif w.name.startswith("pendulum_N"):
        body = (f"the reference's published benchmark class (docs/src/manual/benchmark.md:4-16): inverted pendulum, "
                f"Np = Nc = {qp.n}, input bounds + soft output bounds on every step (n={qp.n}, m={qp.m}, nth={qp.nth}); "
                "parameter points = the closed loops of the example's two scenarios, perturbed")
    elif w.name == "satellite20":
        body = f"hybrid MPC, satellite Np=Nc=20, 40 binary rows, branch and bound (n={qp.n}, m={qp.m}, nth={qp.nth})"
    else:
        body = f"{w.name} (n={qp.n}, m={qp.m}, nth={qp.nth})"
    return f"{w.workload}: {body}, {w.n_local} parameter points per GPU, cold start, first move u0 returned"


def multi_abi_check(torch, lmpc, g, nout, n_per_dev, seed):
    """The C ABI's one-process multi-device entry point (lmpc_solve_batch_multi_device: resident shards, RCCL gather to
    the first device over xGMI), executed across every GPU this process sees: the gathered solutions and exit flags
    must equal the per-device shards bit for bit, and a sample of every shard the oracle.  Untimed part of the run;
    reported as config.multi_abi.  With one visible device there is nothing to gather: says so."""
    nd = torch.cuda.device_count()
    out = {"n_devices": nd}
    if nd < 2 and os.environ.get("LMPC_MULTI_TRANSPORT") == "rccl_self":
        # one GPU: the RCCL calls of the gather with the device itself as the peer (csrc/lmpc_multi.hip, test hook)
        try:
            mq = lmpc.MultiQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout, devices=[0])
            qp1 = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
            td = torch.from_numpy(make_theta("pendulum", n_per_dev, seed)).to("cuda:0")
            xs, fs, xr, fr = mq.solve_device([td], gather=True)
            x1, f1 = qp1.solve_device(td)
            torch.cuda.synchronize()
            out["rccl_self_identical"] = bool(torch.equal(xr, x1) and torch.equal(fr, f1))
        except Exception as e:
            out["error"] = f"{type(e).__name__}: {e}"[:300]
        return out
    if nd < 2:
        out["skipped"] = "one visible GPU: the nd > 1 branch (ncclCommInitAll, ncclSend/ncclRecv) cannot run here"
        return out
    try:
        from oracle import ldp as oldp
        mq = lmpc.MultiQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
        hosts = [make_theta("pendulum", n_per_dev, seed + 31 * d) for d in range(nd)]
        shards = [torch.from_numpy(hosts[d]).to(f"cuda:{d}") for d in range(nd)]
        mq.solve_device(shards, gather=True)                      # first gather: librccl load + ncclCommInitAll
        for d in range(nd):
            torch.cuda.synchronize(d)
        t0 = time.perf_counter()
        xs, fs, xr, fr = mq.solve_device(shards, gather=True)
        for d in range(nd):
            torch.cuda.synchronize(d)
        out["ms"] = 1e3 * (time.perf_counter() - t0)
        off = np.cumsum([0] + [n_per_dev] * nd)
        xr_h, fr_h = xr.cpu().numpy(), fr.cpu().numpy()
        same = all(np.array_equal(xr_h[off[d]:off[d + 1]], xs[d].cpu().numpy()) and
                   np.array_equal(fr_h[off[d]:off[d + 1]], fs[d].cpu().numpy()) for d in range(nd))
        pk = mq.parts[0].ldp()
        L = oldp.LDP(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["M"], pk["du"], pk["dl"], pk["Dth"],
                     pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"]))
        ok_oracle = True
        for d in range(nd):
            xo, efo, _, _ = oldp.solve_batch(L, hosts[d][:2048])
            ok_oracle = ok_oracle and np.array_equal(xr_h[off[d]:off[d] + 2048], xo) and np.array_equal(fr_h[off[d]:off[d] + 2048], efo)
        out.update({"identical": bool(same), "oracle_sample_identical": bool(ok_oracle), "points_per_device": n_per_dev,
                    "solves_per_s_incl_gather": nd * n_per_dev / (out["ms"] * 1e-3),
                    "note": "ONE call of lmpc_solve_batch_multi_device from one process: every device solves its resident "
                            "shard, then an RCCL gather (ncclSend/ncclRecv in one group) of x and exit flags to device 0; "
                            "wall time of the call incl. the gather and the host's launch loop"})
        mq.close()
    except Exception as e:                                        # reported, never fatal for the bench line
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def region_discovery_config(torch, lmpc, dev, local_rank, nsamples, want_cpu, reps=20):
    """BASELINE config 4 on one GPU as a measured workload: sampling-based discovery of the pendulum's critical regions
    over the example's +-20 ParameterRange (/root/reference/src/mpc_examples.jl:128-134; caller side of
    /root/reference/src/explicit.jl:23-48).  A step = draw nothing new: the resident sample is solved with the
    active-set masks kept on the device (lmpc_solve_batch_device), reduced there to the distinct masks
    (lmpc_distinct_active_sets_device) and the distinct sets are read back; samples / wall second, host in the loop."""
    from linearmpc_jl_amd import explicit
    g = make_problem("pendulum")
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], device=local_rank)
    lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0])
    ub = np.array([20.0] * 4 + [20.0, 0.0] + [2.0])
    first = explicit.discover_regions_device(qp, lb, ub, nsamples, seed=4)
    theta = first["theta"]
    # a step = ONE enqueue (solve with masks -> distinct masks -> sets published into mapped host memory,
    # lmpc_discover_regions_device) and ONE synchronisation; buffers allocated once
    sampler = explicit.DeviceRegionSampler(qp, nsamples, capacity=1024)
    sampler.run(theta)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        masks_, counts_, first_, solved_ = sampler.run(theta)
    dt = (time.perf_counter() - t0) / reps
    out = {"masks": masks_, "counts": counts_, "first_index": first_, "n_solved": solved_}
    same_as_first = bool(np.array_equal(masks_, first["masks"]) and np.array_equal(counts_, first["counts"])
                         and np.array_equal(first_, first["first_index"]))
    # the solve alone (masks written), device-timed by wall clock around a synchronised call
    act = torch.empty((nsamples, qp.words), dtype=torch.int64, device=dev)
    qp.solve_device(theta, active=act); torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(reps):
        qp.solve_device(theta, active=act)
    torch.cuda.synchronize(dev)
    solve_ms = 1e3 * (time.perf_counter() - t1) / reps
    res = {"value": nsamples / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt, "samples": nsamples,
           "distinct_active_sets": int(len(out["masks"])), "n_solved": int(out["n_solved"]),
           "solve_with_masks_ms": solve_ms, "kernel": qp.kernel_name, "dtype": "f64",
           "bytes_read_back": int(len(out["masks"]) * (qp.words + 2) * 8),
           "workload": "pendulum Nc=5 (BASELINE config 4, one GPU): sample of the +-20 parameter range resident on the "
                       "device -> batched solve with active-set masks -> distinct masks on the device -> sets published "
                       "into mapped host memory; one enqueue and one synchronisation per step",
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "achieved": nsamples * (8 * qp.nth + 8 * qp.nout + 4 + 8 * qp.words + 8 * qp.words + 4) / dt / 1e9,
                        "frac": nsamples * (8 * qp.nth + 8 * qp.nout + 4 + 8 * qp.words + 8 * qp.words + 4) / dt / 1e9 / HBM_PEAK_GBS,
                        "traffic": None,
                        "note": "algorithmic bytes per sample: theta in, x / flag / mask out, mask + flag in again for the "
                                "reduction; divided by the WALL time of a step, which includes the host's enqueue and its "
                                "one synchronisation"}}
    # cross-check against the host path on a slice
    th_h = theta[:100000].cpu().numpy()
    ref = explicit.discover_regions(qp.solve, th_h)
    dv = explicit.discover_regions_device(qp, None, None, 0, theta=theta[:100000].contiguous())
    keyf = lambda d: sorted((tuple(int(w) for w in m), int(c)) for m, c in zip(d["masks"], d["counts"]))
    res["verified"] = bool(keyf(ref) == keyf(dv)) and same_as_first
    if want_cpu:
        from oracle import ldp as oldp
        global _NATIVE_FLAGS
        if _NATIVE_FLAGS is None:
            _NATIVE_FLAGS = oldp.use_native()
        L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=qp.nout)
        ns = 200000
        th_c = theta[:ns].cpu().numpy()
        t2 = time.perf_counter()
        npass = 0
        while time.perf_counter() - t2 < 3.0:
            _, efc, _, actc = oldp.solve_batch(L, th_c)
            np.unique(actc[efc >= 1], axis=0, return_counts=True)
            npass += 1
        dtc = time.perf_counter() - t2
        res["cpu_baseline"] = {"value": npass * ns / dtc, "unit": "samples/s", "cores": 1, "kind": "port",
                               "sample": f"{npass} passes over the first {ns} samples: oracle/daqp_ldp_oracle.c solve with "
                                         f"active sets + numpy.unique over the masks, 1 thread ({_cpu_model()}) {_NATIVE_FLAGS}"}
    qp.close()
    return res


def closed_loop_config(torch, lmpc, name, nscen, T, dev, local_rank, want_cpu, gram=0, reps=2, ncheck=512):
    """SURVEY 8(f-1) as a measured workload: `nscen` closed loops of `T` steps on the device (lmpc_simulate_device,
    warm; theta = [x; r; uprev] -> solve -> x <- F x + G u, /root/reference/src/simulation.jl:93-113), scenario-steps
    per second.  name "pendulum": the headline problem (lane kernels, scenario-asynchronous loop); "pendulum_N50": the
    benchmark class's N = 50 problem (soft state rows; wavefront kernel, step-synchronous loop that keeps every
    scenario's factorisation between steps).  Verified: the first `ncheck` scenarios against the CPU checker's closed
    loop bit for bit (scenarios are independent of each other)."""
    import ctypes
    from linearmpc_jl_amd._cabi import lib as _lib, check as _check
    g = make_problem(name)
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1,
                                  device=local_rank)
    if qp.kernel_name.endswith("wave"):
        qp.set_option("gram_scan", gram)
    rng = np.random.default_rng(0)
    if "n_closed_loop" in g:
        base = g["theta"][:int(g["n_closed_loop"])]
        pick = base[rng.integers(0, len(base), nscen)] + rng.normal(size=(nscen, 7)) * [0.02, 0.05, 0.005, 0.05, 0.02, 0.0, 0.0]
        x0, r0 = np.ascontiguousarray(pick[:, :4]), np.ascontiguousarray(pick[:, 4:6])
        F, G = g["F"], g["G"]
    else:
        from oracle import mpc2mpqp as omm
        prob = omm.pendulum()
        F, G = prob.F, prob.G
        x0 = rng.uniform([-5, -5, -.3, -2], [5, 5, .3, 2], (nscen, 4))
        r0 = np.stack([rng.uniform(-2, 2, nscen), np.zeros(nscen)], 1)
    nx, nu = F.shape[0], 1
    Fc = np.ascontiguousarray(F, np.float64); Gc = np.ascontiguousarray(np.asarray(G, np.float64).reshape(nx, nu))
    vp = lambda a: ctypes.c_void_p(a)
    xd0 = torch.from_numpy(x0).to(dev); rd = torch.from_numpy(r0).to(dev)
    fm = torch.empty(nscen, dtype=torch.int32, device=dev)
    L_ = _lib()
    def run():
        xd = xd0.clone(); up = torch.zeros((nscen, 1), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        _check(L_.lmpc_simulate_device(qp._h, nscen, T, nx, 2, 1, vp(Fc.ctypes.data), vp(Gc.ctypes.data), vp(xd.data_ptr()),
                                       vp(rd.data_ptr()), vp(up.data_ptr()), None, None, vp(fm.data_ptr()), 1, None), qp._h)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, xd, up
    first = run()[0]                   # (a fresh handle: no working-set statistics yet, scratch being allocated)
    best, xd, up = min((run() for _ in range(reps)), key=lambda t: t[0])
    res = {"value": nscen * T / best, "unit": "scenario-steps/s", "ms_per_step": 1e3 * best / T, "scenarios": nscen, "steps": T,
           "first_run_value": nscen * T / first,
           "kernel": qp.kernel_name, "dtype": "f64", "warm": True, "min_flag": int(fm.min().item()),
           "options": ({"gram_scan": gram} if qp.kernel_name.endswith("wave") else {}),
           "workload": f"{name}: closed loop, states and references resident on the device, no trajectories recorded"}
    # verification against the checker's closed loop on the first scenarios
    from oracle import ldp as oldp
    global _NATIVE_FLAGS
    if _NATIVE_FLAGS is None:
        _NATIVE_FLAGS = oldp.use_native()
    pk = qp.ldp()                      # the handle's own LDP (what the tests compare on): same pack, same bits
    Lq = oldp.LDP(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["M"], pk["du"], pk["dl"], pk["Dth"],
                  pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"]))
    so = oldp.default_settings(); so.mode = 1 if (gram and qp.kernel_name.endswith("wave")) else 0
    owarm = 2 if qp.kernel_name.endswith("wave") else True
    t1 = time.perf_counter()
    ref = oldp.simulate(Lq, x0[:ncheck], T, F, G, r=r0[:ncheck], uprev=np.zeros((ncheck, 1)), settings=so, warm=owarm)
    dtc = time.perf_counter() - t1
    xg = xd[:ncheck].cpu().numpy(); ug = up[:ncheck].cpu().numpy()
    if os.environ.get("LMPC_BENCH_DEBUG"): print("closed loop check:", np.abs(xg - ref["x"]).max(), np.abs(ug - ref["uprev"]).max(), fm[:ncheck].cpu().numpy(), ref["flag_min"], file=sys.stderr)
    res["verified"] = bool(np.array_equal(xg, ref["x"]) and np.array_equal(ug, ref["uprev"])
                           and np.array_equal(fm[:ncheck].cpu().numpy(), ref["flag_min"]))
    res["verification"] = {"scenarios": ncheck, "against": f"oracle_simulate warm={owarm!r} mode={so.mode}",
                           "final_states_and_inputs_bit_identical": res["verified"]}
    if want_cpu:
        res["cpu_baseline"] = {"value": ncheck * T / dtc, "unit": "scenario-steps/s", "cores": 1, "kind": "port",
                               "sample": f"the same {ncheck} scenarios x {T} steps: oracle/daqp_ldp_oracle.c oracle_simulate, "
                                         f"1 thread ({_cpu_model()}) {_NATIVE_FLAGS}"}
    qp.close()
    return res


def avi_closed_loop_config(torch, lmpc, dev, local_rank, nscen, T, want_cpu, ncheck=512, reps=2):
    """The reference's game-theoretic test as it runs there -- a closed loop (test/runtests.jl:1345-1354:
    Simulation(mpc; x0 = 10*ones(2), r = [10, 0], N = 500)) -- batched: `nscen` scenarios around that start, `T` steps on
    the device (lmpc_simulate_device, cold starts: every step runs the register-resident chain), scenario-steps per
    second.  Verified: the first `ncheck` scenarios against the CPU checker's closed loop bit for bit, and every
    checked scenario on its way to the reference's end values."""
    import ctypes
    from linearmpc_jl_amd._cabi import lib as _lib, check as _check
    from oracle import avi as oavi
    g = make_problem("game_kat")
    nu = int(g["nu"])
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nu,
                                  device=local_rank)
    F, G = np.ascontiguousarray(g["F"], np.float64), np.ascontiguousarray(g["G"], np.float64)
    nx = F.shape[0]
    rng = np.random.default_rng(0)
    x0 = np.array([10.0, 10.0]) + rng.uniform(-5, 5, (nscen, 2)); x0[0] = [10.0, 10.0]
    r0 = np.tile([10.0, 0.0], (nscen, 1))
    vp = lambda a: ctypes.c_void_p(a)
    xd0 = torch.from_numpy(x0).to(dev); rd = torch.from_numpy(r0).to(dev)
    fm = torch.empty(nscen, dtype=torch.int32, device=dev)
    L_ = _lib()
    def run():
        xd = xd0.clone(); up = torch.zeros((nscen, nu), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        _check(L_.lmpc_simulate_device(qp._h, nscen, T, nx, 2, nu, vp(F.ctypes.data), vp(G.ctypes.data), vp(xd.data_ptr()),
                                       vp(rd.data_ptr()), vp(up.data_ptr()), None, None, vp(fm.data_ptr()), 0, None), qp._h)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, xd, up
    first = run()[0]
    best, xd, up = min((run() for _ in range(reps)), key=lambda t: t[0])
    res = {"value": nscen * T / best, "unit": "scenario-steps/s", "ms_per_step": 1e3 * best / T, "scenarios": nscen, "steps": T,
           "first_run_value": nscen * T / first, "kernel": qp.kernel_name, "dtype": "f64", "warm": False,
           "min_flag": int(fm.min().item()),
           "workload": "game_kat: closed loop of the two-player game (non-symmetric H, is_avi), states and references resident "
                       "on the device, cold start every step, no trajectories recorded"}
    pk = qp.avi_pack()
    P = oavi.AVI(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["ML"], pk["MR"], pk["G"], pk["du"], pk["dl"],
                 pk["Dth"], pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"])).contiguous()
    t1 = time.perf_counter()
    ref = oavi.simulate(P, x0[:ncheck], T, F, G, r=r0[:ncheck], uprev=np.zeros((ncheck, nu)), warm=False)
    dtc = time.perf_counter() - t1
    xg = xd[:ncheck].cpu().numpy(); ug = up[:ncheck].cpu().numpy()
    res["verified"] = bool(np.array_equal(xg, ref["x"]) and np.array_equal(ug, ref["uprev"])
                           and np.array_equal(fm[:ncheck].cpu().numpy(), ref["flag_min"]) and int(fm.min().item()) >= 1)
    res["verification"] = {"scenarios": ncheck, "against": "oracle_avi_simulate (cold)",
                           "final_states_and_inputs_bit_identical": res["verified"],
                           "max_distance_to_reference_end_values": float(np.abs(xg - [10.0, 0.0]).max())}
    if want_cpu:
        res["cpu_baseline"] = {"value": ncheck * T / dtc, "unit": "scenario-steps/s", "cores": 1, "kind": "port",
                               "sample": f"the same {ncheck} scenarios x {T} steps: oracle/daqp_avi_oracle.c oracle_avi_simulate, "
                                         f"1 thread ({_cpu_model()})"}
    qp.close()
    return res


def avi_config(torch, lmpc, dev, local_rank, batch, steps, warmup, want_cpu, cpu_seconds=3.0):
    """The reference's game-theoretic MPC (test/runtests.jl:1337-1358: two players, non-symmetric H, DAQP's is_avi
    mode, /root/reference/src/setup.jl:11-13) as a measured workload: `batch` parameter points (the fixture's
    sampling: 0 .. 6 active bounds) resident on the device, lmpc_solve_batch_device on the AVI kernel, first move of
    both players returned.  Verified against the oracle's AVI solver on a sample, bit for bit."""
    from oracle import avi as oavi, ldp as oldp
    g = make_problem("game_kat")
    nout = int(g["nu"])
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout,
                                  device=local_rank)
    for kv in os.environ.get("LMPC_BENCH_AVI_OPTIONS", "").split(","):      # (tools/: "avi_tiers=0,avi_tiers_first=3")
        if "=" in kv:
            qp.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    rng = np.random.default_rng(1234)
    th_h = np.ascontiguousarray(np.hstack([rng.uniform(-30, 30, (batch, 4)), rng.uniform(-1, 1, (batch, 2))]))
    bytes_per = algorithmic_bytes(qp.nth, nout)
    nrot = rotation_depth(batch, bytes_per)
    thetas = [torch.from_numpy(th_h).to(dev)] + [torch.from_numpy(np.ascontiguousarray(np.roll(th_h, 1 + r_, axis=0))).to(dev)
                                                  for r_ in range(nrot - 1)]
    xb = torch.empty((batch, nout), dtype=torch.float64, device=dev)
    fb = torch.empty(batch, dtype=torch.int32, device=dev)
    itb = torch.empty(batch, dtype=torch.int32, device=dev)
    qp.reserve(batch)
    torch.cuda.synchronize(dev)
    t_first = time.perf_counter()            # the very first call on the fresh handle, by itself
    qp.solve_device(thetas[0], x=xb, exitflag=fb)
    torch.cuda.synchronize(dev)
    first_call_s = time.perf_counter() - t_first
    for k in range(warmup):
        qp.solve_device(thetas[k % nrot], x=xb, exitflag=fb)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        qp.solve_device(thetas[k % nrot], x=xb, exitflag=fb)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    qp.profile(True)
    for k in range(max(3, steps)):
        qp.solve_device(thetas[k % nrot], x=xb, exitflag=fb)
    torch.cuda.synchronize(dev)
    solo = qp.profile_read()
    qp.profile(False)
    qp.solve_device(thetas[0], x=xb, exitflag=fb, iters=itb)
    torch.cuda.synchronize(dev)
    pk = qp.avi_pack()
    P = oavi.AVI(pk["n"], pk["m"], pk["ms"], pk["nth"], pk["nout"], pk["ML"], pk["MR"], pk["G"], pk["du"], pk["dl"],
                 pk["Dth"], pk["Rout"], pk["x0"], pk["Xth"], pk["sense"], np.ones(pk["m"])).contiguous()
    idx = np.sort(np.random.default_rng(99).choice(batch, min(4096, batch), replace=False))
    xo, efo, ito, _ = oavi.solve_batch(P, th_h[idx])
    ti = torch.from_numpy(idx).to(dev)
    ok = bool(np.array_equal(xb[ti].cpu().numpy(), xo) and np.array_equal(fb[ti].cpu().numpy(), efo)
              and np.array_equal(itb[ti].cpu().numpy(), ito))
    call_ms = solo[1]
    ach = bytes_per * batch / (call_ms * 1e-3) / 1e9 if call_ms > 0 else 0.0
    it_h = itb.cpu().numpy()
    res = {"value": batch * steps / el, "unit": "solves/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warmup,
           "dtype": "f64", "batch": batch, "kernel": qp.kernel_name, "rotating_batches": nrot, "verified": ok,
           "workload": f"game_kat: two-player game-theoretic MPC (non-symmetric H, is_avi; n={qp.n}, m={qp.m}, nth={qp.nth}), "
                       f"{batch} parameter points, cold start, first move of both players returned",
           "solved_fraction": float((fb.cpu().numpy() >= 1).mean()), "mean_iterations": float(it_h.mean()),
           "max_iterations": int(it_h.max()),
           "first_run_value": batch / first_call_s, "first_call_ms": 1e3 * first_call_s,
           "parity": "unpinned iteration: libdaqp's AVI mode is not restated (source unavailable); GPU == this build's own "
                     "principal-pivoting oracle bit for bit; reference-held: the closed loop's end values [10, 0] "
                     "(runtests.jl:1337-1358) and KKT certificates of the unique equilibrium",
           "verification": {"points": int(len(idx)), "against": "oracle/daqp_avi_oracle.c on the handle's pack: x, exit flag "
                                                                 "and iteration count bit-identical"},
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": None, "duration_used_ms": call_ms, "algorithmic_bytes_per_solve": bytes_per,
                        "note": "algorithmic bytes (theta in, u0 and flag out) over the HIP-event duration of one call (a chain of "
                                "three launches: register-resident tiers, lane kernel on its list, generic kernel on that "
                                "one's); the chain is VALU-issue bound (f64 chains, selects), HBM traffic 1.26x these bytes "
                                "(profiles/r04_avi_chain.md)"}}
    if want_cpu:
        ns = min(batch, 200000)
        t1 = time.perf_counter()
        npass = 0
        while time.perf_counter() - t1 < cpu_seconds:
            oavi.solve_batch(P, th_h[:ns])
            npass += 1
        dtc = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": npass * ns / dtc, "unit": "solves/s", "cores": 1, "kind": "port",
                               "sample": f"{npass} passes over the first {ns} points, 1 thread ({_cpu_model()}), "
                                         "oracle/daqp_avi_oracle.c, portable -O3 build"}
    qp.close()
    return res


def multi_abi_isolated(torch, n_per_dev, timeout_s=180):
    """multi_abi_check in a CHILD process (`bench.py --multi-abi-only N`): the nd > 1 branch of the library has never
    run on hardware, so whatever it does the first time -- raise, hang, crash inside librccl -- must not take the
    bench line with it.  One visible GPU: answered here, no child."""
    import subprocess
    nd = torch.cuda.device_count()
    if nd < 2:
        # one GPU: everything of the n_devices > 1 path except RCCL itself -- two shards on device 0, gather by
        # event-ordered peer copies (LMPC_MULTI_TRANSPORT=copy, include/lmpc_hip.h lmpc_multi_set_option)
        out = {"n_devices": nd, "transport": "copy",
               "note": "one visible GPU: RCCL (ncclCommInitAll, ncclSend/ncclRecv) cannot run here; the multi-device "
                       "control flow ran as two shards on device 0 with the gather as peer copies"}
        try:
            import linearmpc_jl_amd as lmpc
            g = make_problem("pendulum")
            os.environ["LMPC_MULTI_TRANSPORT"] = "copy"
            try:
                mq = lmpc.MultiQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"],
                                            nout=1, devices=[0, 0])
            finally:
                del os.environ["LMPC_MULTI_TRANSPORT"]
            n2 = int(min(n_per_dev, 250_000))
            hosts = [make_theta("pendulum", n2, 4242 + 31 * d) for d in range(2)]
            shards = [torch.from_numpy(h_).to("cuda:0") for h_ in hosts]
            xs, fs, xr, fr = mq.solve_device(shards, gather=True)
            qp1 = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=1)
            x1, f1 = qp1.solve_device(torch.cat(shards))
            torch.cuda.synchronize()
            out["loopback_identical"] = bool(torch.equal(xr, x1) and torch.equal(fr, f1) and torch.equal(xs[1], x1[n2:]))
            out["points_per_shard"] = n2
            mq.close(); qp1.close()
        except Exception as e:
            out["error"] = f"{type(e).__name__}: {e}"[:300]
        # ... and the RCCL calls themselves (library load, ncclCommInitAll, a grouped send / receive pair of the gather's
        # data types) with the device as its own peer, in a child process with a time limit
        try:
            env = dict(os.environ, LMPC_MULTI_TRANSPORT="rccl_self", HSA_ENABLE_IPC_MODE_LEGACY="0")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--multi-abi-only", "100000"],
                               capture_output=True, text=True, timeout=90, env=env)
            last = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
            out["rccl_self_identical"] = bool(last and json.loads(last[-1]).get("rccl_self_identical"))
        except Exception as e:
            out["rccl_self_identical"] = False
            out["rccl_self_error"] = f"{type(e).__name__}: {e}"[:200]
        return out
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--multi-abi-only", str(int(n_per_dev))],
                           capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                res = json.loads(line)
                res["isolated"] = f"child process, exit code {r.returncode}"
                return res
        return {"n_devices": nd, "error": f"child process exit code {r.returncode}, no result line; stderr tail: {r.stderr[-400:]}"}
    except subprocess.TimeoutExpired:
        return {"n_devices": nd, "error": f"child process still running after {timeout_s} s (killed)"}
    except Exception as e:
        return {"n_devices": nd, "error": f"{type(e).__name__}: {e}"}


def _forms_agreement(xa, fa, xb, fb):
    ok = fa >= 1
    both = ok & (fb >= 1)
    return {"solved_fraction": float(ok.mean()), "solved_status_flips": int((ok != (fb >= 1)).sum()),
            "exit_flags_differ_among_failed": int(((fa != fb) & ~ok).sum()),
            "max_abs_dx_on_solved": float(np.abs(xa[both] - xb[both]).max()) if both.any() else 0.0,
            "solved_points_with_dx_above_1e-6": int((np.abs(xa[both] - xb[both]).max(axis=1) > 1e-6).sum()) if both.any() else 0}


def side_config(torch, lmpc, workload, batch, dev, local_rank, steps, warmup, f32, want_cpu, cpu_seconds=4.0):
    """One of the other single-GPU BASELINE configurations, measured in the same process: one batch in
    flight on one stream (these kernels fill the chip by themselves), rotation past the Infinity Cache,
    roofline from the live HIP-event duration of a call.  Wavefront-kernel workloads are measured in both of the
    kernel's forms: the n-chain form (library default; `value`) and the Gram-scan form (`gram_scan`)."""
    _phase(f"config {workload}: setup")
    w = Workload(torch, lmpc, workload, batch, dev, local_rank, 0, 1, f32=f32)
    # the very FIRST call on the fresh handle, timed by itself (allocations, and on the wavefront path the probe that
    # decides the launch shape -- lmpc_set_option "wave_probe"): what a one-shot caller gets
    torch.cuda.synchronize(dev)
    t_first = time.perf_counter()
    w.launch(0)
    torch.cuda.synchronize(dev)
    first_call_s = time.perf_counter() - t_first
    dist_info, flop = w.work_distribution()
    x_chain, f_chain = w.xbuf[0].cpu().numpy().copy(), w.fbuf[0].cpu().numpy().copy()
    _phase(f"config {workload}: timed region")
    el = w.timed(steps, warmup)
    verification = w.verify_steps([steps - 1], per_step=2048)
    solo = w.single_launch(max(3, min(steps, 20)))
    value = batch * steps / el
    wave = w.kernel.endswith("wave")
    dtype = "f32" if f32 else "f64"
    call_ms = solo[1]
    if wave:     # VALU-bound kernel: flop roofline against the dense vector peak
        ach = flop * batch / (call_ms * 1e-3) / 1e12 if call_ms > 0 else 0.0
        roof = {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                "frac": ach / VALU_PEAK_TFLOPS[dtype], "traffic": None,
                "flop_per_solve_est": flop, "duration_used_ms": call_ms,
                "duration": "HIP-event duration of one lmpc_solve_batch_device call, one at a time on one stream",
                "hbm_algorithmic_GBs": w.bytes_per * batch / (call_ms * 1e-3) / 1e9 if call_ms > 0 else 0.0}
    else:
        ach = w.bytes_per * batch / (call_ms * 1e-3) / 1e9 if call_ms > 0 else 0.0
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": None, "duration_used_ms": call_ms, "screen_kernel_ms": solo[2], "iterate_kernel_ms": solo[3],
                "duration": "HIP-event duration of one lmpc_solve_batch_device call (screening + iterating kernel), "
                            "one at a time on one stream, cold HBM (rotating batches)",
                "fp64_flop_per_solve_est": flop}
    out = {"value": value, "unit": "solves/s", "ms_per_step": 1e3 * el / steps, "steps": steps, "warmup": warmup,
           "dtype": dtype, "batch": batch, "kernel": w.kernel, "rotating_batches": w.nrot,
           "workload": describe(w), **dist_info, "roofline": roof, "verified": verification["verified"],
           "verification": verification, "options": dict(w.options),
           "first_run_value": batch / first_call_s, "first_call_ms": 1e3 * first_call_s}
    if w.kernel.endswith("wave") and hasattr(w.qp, "wave_stats"):
        out["wave_stats"] = w.qp.wave_stats()       # working-set sizes seen; first_pass_rows > 0: two passes
    if want_cpu:
        _phase(f"config {workload}: cpu baseline")
        out["cpu_baseline"] = cpu_baseline(w.g, w.theta_h, w.nout, min_seconds=cpu_seconds, f32=f32,
                                           all_cores_seconds=cpu_seconds / 2)
        if not f32:     # marginal points of a bounded sample (oracle + dense KKT per distinct active set)
            from oracle import ldp as oldp
            Lm = oldp.qp2ldp(w.g["H"], w.g["f"], w.g["f_theta"], w.g["A"], w.g["bu"], w.g["bl"], w.g["W"], w.g["senses"], nout=w.nout)
            out["marginal_cases"] = oldp.marginal_report(Lm, w.theta_h[:4000])
    theta_h, gg, nout_ = w.theta_h, w.g, w.nout
    w.close()
    if wave:
        _phase(f"config {workload}: Gram-scan form")
        wg = Workload(torch, lmpc, workload, batch, dev, local_rank, 0, 1, f32=f32, options={"gram_scan": 1})
        dist_g, _ = wg.work_distribution()
        x_gram, f_gram = wg.xbuf[0].cpu().numpy().copy(), wg.fbuf[0].cpu().numpy().copy()
        elg = wg.timed(steps, warmup)
        ver_g = wg.verify_steps([steps - 1], per_step=2048)
        solog = wg.single_launch(max(3, min(steps, 20)))
        achg = flop * batch / (solog[1] * 1e-3) / 1e12 if solog[1] > 0 else 0.0
        out["gram_scan"] = {
            "value": batch * steps / elg, "unit": "solves/s", "ms_per_step": 1e3 * elg / steps,
            "speedup_over_n_chain_form": (batch * steps / elg) / value,
            "options": {"gram_scan": 1}, "verified": ver_g["verified"], "verification": ver_g,
            "mean_iterations": dist_g["mean_iterations"], "solved_fraction": dist_g["solved_fraction"],
            "roofline": {"bound": "valu", "achieved": achg, "peak": VALU_PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                         "frac": achg / VALU_PEAK_TFLOPS[dtype], "traffic": None, "duration_used_ms": solog[1],
                         "flop_per_solve_est": flop,
                         "flop_per_solve_executed_est": wg.flop_gram_executed,
                         "frac_executed": (wg.flop_gram_executed * batch / (solog[1] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS[dtype])
                                          if solog[1] > 0 else 0.0,
                         "note": "frac: same unit of work as the n-chain form -- SURVEY 8(d)'s flop count of the reference "
                                 "algorithm per solve (2mn + 2|W|n + 2|W|^2 per iteration) at the n-chain form's "
                                 "iteration counts, over this form's time (a work-equivalent figure, NOT executed flops); "
                                 "frac_executed: what this form itself executes (2m|W| + 2|W|^2 per iteration, 2|W|n once) "
                                 "at its own iteration counts over the same time"},
            "agreement_with_n_chain_form": _forms_agreement(x_chain, f_chain, x_gram, f_gram),
            "note": "lmpc_set_option('gram_scan', 1): row values from |W| Gram columns instead of n columns of M', "
                    "dual objective from the factorisation, pairwise lane trees for the append's dot products; "
                    "bit-identical to the oracle's mode 1; against the n-chain form: same exit flags, iteration counts "
                    "and active sets on solvable points (tests/test_gpu_gram.py), x to rounding (amplified by 1/rho_soft "
                    "on SOFT rows); on infeasible, nearly dependent problems the two forms may stop with different "
                    "failure flags (-1 / -2)"}
        wg.close()
    return out


PRIME_CALLS = 1      # untimed calls before a side configuration's timed region (plain warm-up; a fresh handle shapes its
                     # launches from a probe of its first batch, not from earlier calls)
LINE_LIMIT = 8192    # the bench line must stay far below what the driver keeps of stdout


def single_call_config(lmpc, g, nout, cpu=True):
    """BASELINE config 1 in this framework's terms: ONE parameter point per call -- lmpc_solve_one (theta in, x and flag out
    through one record of mapped host memory) and the generated controller's mpc_compute_control with one state (host
    arrays) -- wall time per call with the host in the loop, beside the CPU port's time for one solve of the same problem
    (the reference's single compute_control on Julia + DAQP cannot run here)."""
    import ctypes
    from linearmpc_jl_amd._cabi import lib
    vp = ctypes.c_void_p
    qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
    th = np.ascontiguousarray(g["theta"][0]); x = np.zeros(nout)
    for _ in range(50):
        lib().lmpc_solve_one(qp._h, vp(th.ctypes.data), vp(x.ctypes.data))
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        lib().lmpc_solve_one(qp._h, vp(th.ctypes.data), vp(x.ctypes.data))
    solve_one_us = 1e6 * (time.perf_counter() - t0) / n
    out = {"solve_one_us": solve_one_us, "kernel": qp.kernel_name,
           "note": "wall time per call, host in the loop, one GPU; the reference's counterpart is one compute_control on its "
                   "CPU path (BASELINE config 1), quoted from its plots as ~11 us at N = 50 (a larger problem)"}
    qp.close()
    try:
        q = lmpc.MPQP(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"])
        mpc = lmpc.MPC(q, nx=4, nu=1, nr=2, nuprev=1)
        gc = lmpc.GeneratedController(mpc)
        control = np.zeros((1, 1)); state = np.array([[0.5, 0.1, 0.05, 0.0]]); ref = np.array([[1.0, 0.0]])
        for _ in range(50):
            gc.mpc_compute_control(control, state, ref)
        t0 = time.perf_counter()
        for _ in range(n):
            gc.mpc_compute_control(control, state, ref)
        out["compute_control_one_state_us"] = 1e6 * (time.perf_counter() - t0) / n
    except Exception as e:                                    # never fatal for the line
        out["compute_control_error"] = f"{type(e).__name__}: {e}"[:200]
    if cpu:
        from oracle import ldp as oldp
        global _NATIVE_FLAGS
        if _NATIVE_FLAGS is None:
            _NATIVE_FLAGS = oldp.use_native()
        L = oldp.qp2ldp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], nout=nout)
        th1 = np.ascontiguousarray(g["theta"][:1])
        oldp.solve_batch(L, th1)
        t0 = time.perf_counter()
        for _ in range(2000):
            oldp.solve_batch(L, th1)
        out["cpu_port_single_solve_us"] = 1e6 * (time.perf_counter() - t0) / 2000     # (ctypes call overhead included)
        th1k = np.ascontiguousarray(np.repeat(th1, 4096, 0))
        t0 = time.perf_counter()
        oldp.solve_batch(L, th1k)
        out["cpu_port_solve_in_a_loop_us"] = 1e6 * (time.perf_counter() - t0) / 4096  # (the same point, C loop)
    return out


def traffic_stamp(workload):
    """roofline.traffic for the headline: the PMC measurement committed under profiles/ -- with the kernel and the commit it
    was taken on, and null when the kernel's source has changed since (the file would be stale)."""
    pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    if not os.path.exists(pmc):
        return None, {"note": "no PMC measurement committed for this workload"}
    try:
        d = json.load(open(pmc))
    except (OSError, ValueError):
        return None, {"note": "profiles/pmc_traffic file unreadable"}
    info = {"source": d.get("source"), "kernels": d.get("kernels"), "measured_on_commit": d.get("measured_on_commit"),
            "kernel_source_sha16": d.get("kernel_source_sha16")}
    src = os.path.join(ROOT, "linearmpc.jl_amd", "csrc", d.get("kernel_source", "lmpc_fast_kernel.hpp"))
    try:
        import hashlib
        sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    except OSError:
        sha = None
    info["kernel_source_sha16_now"] = sha
    if d.get("kernel_source_sha16") and sha and sha != d["kernel_source_sha16"]:
        info["note"] = "STALE: the kernel's source has changed since the counters were collected; traffic withheld"
        return None, info
    return d.get("hbm_bytes_per_launch"), info


def multi_rank_configs(torch, lmpc, dist, dev, local_rank, rank, world, batch):
    """world > 1 (one process per GPU): BASELINE configs 3 and 4 as they shard -- every rank its own batch / its own
    sample shard.  Config 3: independent shards, no exchange inside the solve; config 4: per-rank distinct-set tables
    merged by ONE all_gather_into_tensor on the device tables (explicit.merge_region_tables).  Times are the maximum over
    ranks between two barriers; values are whole-job aggregates.  Every rank calls this; rank 0 gets the report."""
    from linearmpc_jl_amd import explicit
    rep = {}

    def allmax(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allvals(v):
        t = torch.tensor([float(v)], dtype=torch.float64, device=dev)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        return [float(o.item()) for o in outl]

    # ---- config 3: oscillating masses, 3 inputs; one 10^6-point batch per rank
    try:
        steps = 3
        w = Workload(torch, lmpc, "mass_spring_3in", batch, dev, local_rank, rank, 1)
        w.launch(0); torch.cuda.synchronize(dev)
        dist.barrier()
        el = w.timed(steps, 1)
        el_max = allmax(el)
        ver = w.verify_steps([steps - 1], per_step=1024)
        ok = allvals(1.0 if ver["verified"] else 0.0)
        per_rank = allvals(batch * steps / el)
        rep["mass_spring_3in"] = {"value": world * batch * steps / el_max, "unit": "solves/s", "ms_per_step": 1e3 * el_max / steps,
                                  "per_rank_value": per_rank, "batch_per_gpu": batch, "kernel": w.kernel, "verified": all(v > 0 for v in ok),
                                  "scaling": "weak", "exchange": "none: independent shards, results stay on their GPU"}
        w.close()
    except Exception as e:
        rep["mass_spring_3in"] = {"error": f"{type(e).__name__}: {e}"[:300], "verified": False}
    # ---- config 4: region discovery, the sample sharded over the ranks, tables exchanged as tensors
    try:
        g = make_problem("pendulum")
        qp = lmpc.BatchedQP.from_mpqp(g["H"], g["f"], g["f_theta"], g["A"], g["bu"], g["bl"], g["W"], g["senses"], device=local_rank)
        lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0]); ub = -lb; ub[5] = 0.0
        first = explicit.discover_regions_device(qp, lb, ub, batch, seed=4 + rank)
        theta = first["theta"]
        sampler = explicit.DeviceRegionSampler(qp, batch, capacity=1024)
        sampler.run(theta)
        reps = 10
        dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); t_ex = 0.0
        for _ in range(reps):
            m_, c_, f_, s_ = sampler.run(theta)
            t1 = time.perf_counter()
            gm, gc, gf, gs = explicit.merge_region_tables(sampler.masks[:len(c_)], sampler.counts[:len(c_)], sampler.first[:len(c_)],
                                                          s_, dist.group.WORLD, capacity=1024, index_offset=rank * batch)
            t_ex += time.perf_counter() - t1
        dt = allmax((time.perf_counter() - t0) / reps)
        ex_ms = allmax(1e3 * t_ex / reps)
        # every rank must hold the same merged table, and its counts must add up to the solved samples of all ranks
        chk = float(np.bitwise_xor.reduce(gm.view(np.uint64).reshape(-1)) % (1 << 52)) if len(gc) else 0.0
        same = len(set(allvals(chk))) == 1 and len(set(allvals(float(len(gc))))) == 1
        rep["region_discovery"] = {"value": world * batch / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt, "samples_per_gpu": batch,
                                   "distinct_active_sets": int(len(gc)), "distinct_on_this_rank": int(len(c_)),
                                   "n_solved": int(gs), "verified": bool(same and int(gc.sum()) == int(gs)),
                                   "exchange": {"ms": ex_ms, "bytes_per_rank": 1025 * (qp.words + 2) * 8,
                                                "how": "one all_gather_into_tensor of the per-rank tables of distinct masks "
                                                       "(1024 rows of [mask, count, first index] + header) on the device buffers "
                                                       "the pipeline leaves them in, merged on the device (torch.unique)"}}
        qp.close()
    except Exception as e:
        rep["region_discovery"] = {"error": f"{type(e).__name__}: {e}"[:300], "verified": False}
    return rep


def _r(v, sig=6):
    """Round floats to `sig` significant digits (line size), leave the rest alone."""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}") if math.isfinite(v) else None
    return v


def compact_line(out):
    """The ONE bench line of the driver contract, cut down to what the contract names (a few KB): headline fields,
    roofline, cpu_baseline, and per-config summaries {value, verified[, unit if not solves/s][, frac[, bound if not valu]]
    [, gram_value, gram_frac]}.  The full
    report (every histogram, verification block and note) goes to bench_full.json and to stderr."""
    cfg = out.get("config", {})
    roof = out.get("roofline", {})
    line = {k: _r(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                          "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value_one_call_at_a_time"] = _r(out.get("value_one_call_at_a_time"))
    line["verified"] = out.get("verified")
    line["config"] = {"workload": str(cfg.get("workload", ""))[:300], "batch_per_gpu": cfg.get("batch_per_gpu"),
                      "kernel": cfg.get("kernel"), "batches_in_flight": cfg.get("batches_in_flight"),
                      "options": {k: v for k, v in (cfg.get("options") or {}).items() if not k.startswith("_")}}
    if "exchange" in cfg:
        line["config"]["exchange_ms"] = _r(cfg["exchange"].get("ms"))
    if isinstance(cfg.get("multi_abi"), dict):
        ma = cfg["multi_abi"]
        line["config"]["multi_abi"] = {k: _r(ma[k]) for k in ("n_devices", "identical", "oracle_sample_identical",
                                                               "loopback_identical", "rccl_self_identical", "transport") if k in ma}
    line["roofline"] = {k: _r(roof.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                                                       "algorithmic_bytes_per_solve")}
    if isinstance(roof.get("pipelined"), dict):
        line["roofline"]["pipelined_frac"] = _r(roof["pipelined"].get("frac"))
    sb = roof.get("several_batches_per_call")
    if isinstance(sb, dict) and "frac" in sb:
        line["roofline"]["calls_of_%d_batches" % sb.get("batches_per_call", 0)] = {"frac": _r(sb.get("frac"), 4), "value": _r(sb.get("value")),
                                                                               "ms_per_batch": _r(sb.get("ms_per_batch"), 4),
                                                                               "verified": sb.get("verified")}
    ti = roof.get("traffic_info")
    if isinstance(ti, dict):
        line["roofline"]["traffic_on"] = {"commit": ti.get("measured_on_commit"), "kernel": (ti.get("kernels") or [None])[0],
                                          **({"note": ti["note"]} if "note" in ti else {})}
    sc = out.get("single_call")
    if isinstance(sc, dict):
        line["single_call"] = {k: _r(sc[k], 4) for k in ("solve_one_us", "compute_control_one_state_us", "cpu_port_single_solve_us",
                                                          "cpu_port_solve_in_a_loop_us") if k in sc}
        if "error" in sc:
            line["single_call"]["error"] = sc["error"][:120]
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"),
                                "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:140]}
        if "all_cores" in cb:
            line["cpu_baseline"]["all_cores"] = {"value": _r(cb["all_cores"].get("value")), "cores": cb["all_cores"].get("cores")}
    if "configs" in out:
        cs = {}
        for name, c in out["configs"].items():
            e = {"value": _r(c.get("value")), "verified": c.get("verified")}
            if c.get("unit") != "solves/s":
                e["unit"] = c.get("unit")          # (solves/s unless stated)
            r_ = c.get("roofline") or {}
            if "frac" in r_:
                e["frac"] = _r(r_.get("frac"), 4)
                if r_.get("bound") != "valu":
                    e["bound"] = r_.get("bound")   # (valu unless stated)
            if "error" in c:
                e["error"] = str(c["error"])[:120]
            if "parity" in c:
                e["parity"] = str(c["parity"])[:28]
            gs = c.get("gram_scan")
            if isinstance(gs, dict):
                e["gram_value"] = _r(gs.get("value"))
                e["gram_verified"] = gs.get("verified")
                gr = gs.get("roofline") or {}
                if "frac" in gr:
                    e["gram_frac"] = _r(gr.get("frac"), 4)
                if "frac_executed" in gr:
                    e["gram_frac_executed"] = _r(gr.get("frac_executed"), 4)
            if isinstance(c.get("cpu_baseline"), dict):
                e["cpu_1core"] = _r(c["cpu_baseline"].get("value"), 4)
            if "first_run_value" in c:
                e["first_run_value"] = _r(c.get("first_run_value"), 4)
            if "per_rank_value" in c:
                e["per_rank_value"] = [_r(v, 4) for v in c["per_rank_value"]]
            if "exchange" in c:
                e["exchange"] = ({"ms": _r(c["exchange"].get("ms"), 4), "bytes_per_rank": c["exchange"].get("bytes_per_rank")}
                                 if isinstance(c["exchange"], dict) else str(c["exchange"])[:80])
            cs[name] = e
        line["configs"] = cs
    line["full_report"] = "bench_full.json"
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) > LINE_LIMIT:          # never let the line outgrow the driver's window: drop the optional parts
        line.pop("configs", None)
        line["truncated"] = "configs dropped (line limit); see bench_full.json"
        txt = json.dumps(line, separators=(",", ":"))
    return txt


def emit(out, final):
    """Print the compact line on stdout (flush); on the final call also write the full report to bench_full.json (repo
    root and gpurun_out/ when present) and to stderr."""
    print(compact_line(out), flush=True)
    if final:
        full = json.dumps(out)
        for d_ in (ROOT, os.path.join(ROOT, "gpurun_out")):
            if os.path.isdir(d_):
                try:
                    with open(os.path.join(d_, "bench_full.json"), "w") as fh:
                        fh.write(full + "\n")
                except OSError:
                    pass
        print("[bench full report] " + full, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="pendulum", choices=["pendulum", "pendulum_hard", "mass_spring", "mass_spring_3in", "mass_spring_3in_feasible", "soft_doc", "hybrid",
                             "pendulum_N50", "pendulum_N75", "pendulum_N100", "pendulum_N125"])
    ap.add_argument("--f32", action="store_true",
                    help="binary32 path (lmpc_solve_batch_f32_device; wavefront kernel; reference codegen float_type=float)")
    ap.add_argument("--wave", action="store_true", help="force the wavefront-per-QP kernel (diagnostic)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the other BASELINE configurations (profiling runs: keeps the kernel trace to one workload)")
    ap.add_argument("--gather", default="after", choices=["after", "final", "step", "none"],
                    help="N > 1: the shards are independent, the timed region holds no collective; 'after' (default) "
                         "times ONE RCCL gather of a batch's x and exit flags to rank 0 behind the timed region and "
                         "reports it next to the headline (config.exchange); 'final' puts that gather inside the timed "
                         "region, 'step' all-gathers after every step (overlapped with the next solve), 'none' skips it")
    ap.add_argument("--no-single-launch", action="store_true",
                    help="skip the one-call-at-a-time sections after the timed region (profiling runs: keeps the "
                         "kernel trace to the launches of the timed region)")
    ap.add_argument("--no-rotate", action="store_true", help="reuse ONE theta buffer every step (cache-resident; diagnostic)")
    ap.add_argument("--no-screen", action="store_true", help="iterating kernel only (diagnostic)")
    ap.add_argument("--lane-per", type=int, default=0, help="work-list workgroups per shard (tuning)")
    ap.add_argument("--lane-tier", type=int, default=-1, help="lane kernel: first-tier capacity on (1) / off (0) (tuning)")
    ap.add_argument("--lane-block", type=int, default=0, help="lane-kernel workgroup size (tuning)")
    ap.add_argument("--wave-level", type=int, default=-1, help="wave kernel: LDS staging level 0..3 (tuning)")
    ap.add_argument("--wave-nwv", type=int, default=0, help="wave kernel: wavefronts per workgroup (tuning)")
    ap.add_argument("--wave-cap", type=int, default=0, help="wave kernel: wavefronts per CU of the grid (tuning; option wave_waves)")
    ap.add_argument("--multi-abi-only", type=int, default=0, metavar="N_PER_DEVICE",
                    help="run only the one-process multi-device entry-point check (lmpc_solve_batch_multi_device over every "
                         "visible GPU) and print its result as one JSON line; what the bench line's config.multi_abi runs "
                         "in a child process")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="lmpc_set_option(NAME, VALUE) on every handle of the headline workload (recorded in config.options)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches kept in flight per GPU (each has its own handle and HIP stream)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import linearmpc_jl_amd as lmpc

    if args.multi_abi_only > 0:
        print(json.dumps(multi_abi_check(torch, lmpc, make_problem("pendulum"), 1, args.multi_abi_only, 4242)), flush=True)
        return
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

    nstreams = max(1, min(8, args.streams))
    opts = {}
    if args.no_screen:
        opts["screen"] = 0
    if args.wave:
        opts["wave"] = 1
    if args.lane_per:
        opts["lane_per"] = args.lane_per
    if args.wave_cap:
        opts["wave_waves"] = args.wave_cap
    if args.wave_level >= 0:
        opts["wave_level"] = args.wave_level
    if args.wave_nwv:
        opts["wave_nwv"] = args.wave_nwv
    # several batches in flight: the library is TOLD so (lmpc_set_option "in_flight", include/lmpc_hip.h) and picks
    # its own workgroup shapes for a shared chip; everything else runs on library defaults.  Every option set on the
    # handles of the timed region is recorded in config.options.
    if nstreams > 1 and not args.f32 and not args.wave and args.workload in ("pendulum", "pendulum_hard"):
        opts["in_flight"] = nstreams
    if args.lane_block:
        opts["lane_block"] = args.lane_block
    if args.lane_tier >= 0:
        opts["lane_tier"] = args.lane_tier
    for kv in args.opt:
        k_, _, v_ = kv.partition("=")
        opts[k_.strip()] = int(v_)
    n_local = args.batch
    W = Workload(torch, lmpc, args.workload, n_local, dev, local_rank, rank, nstreams, f32=args.f32,
                 rotate=not args.no_rotate, options=opts)
    qp, qps, streams, nout, tdt = W.qp, W.qps, W.streams, W.nout, W.tdt
    xbuf, fbuf, nbuf = W.xbuf, W.fbuf, W.nbuf

    do_gather = world > 1 and args.gather == "step"
    final_gather = world > 1 and args.gather == "final"
    after_gather = world > 1 and args.gather == "after"
    if final_gather or after_gather:
        # gather to rank 0: every rank sends its shard straight to the root over its own xGMI link
        # (ncclSend/ncclRecv pairs), N-1 links in parallel -- a ring all-gather would move N-1 shards
        # through every link and nobody but the root reads them
        xfin = torch.empty((world * n_local, nout), dtype=tdt, device=dev) if rank == 0 else None
        ffin = torch.empty(world * n_local, dtype=torch.int32, device=dev) if rank == 0 else None

    gather_impl = {"mode": "gather"}

    def gather_to_root(xs, fs):
        nonlocal xfin, ffin
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
    last_b = [0]

    def step(k):
        # step k runs on stream k % nstreams with that stream's own handle (work list, counters) on batch
        # k % nrot: consecutive steps are independent batches, so the streaming pass of one overlaps the
        # latency-bound iterating pass of the other
        if not do_gather:                        # raw stream handle, no context switch
            last_b[0] = W.launch(k)
            return
        b = k % nbuf
        sidx = k % nstreams
        with torch.cuda.stream(streams[sidx]):
            if pending[b] is not None:           # buffer b is free once its gather has finished
                for w in pending[b]:
                    w.wait()
                pending[b] = None
            qps[sidx].solve_device(W.thetas[k % W.nrot], x=xbuf[b], exitflag=fbuf[b])
            pending[b] = (dist.all_gather_into_tensor(xall[b], xbuf[b], async_op=True),
                          dist.all_gather_into_tensor(fall[b], fbuf[b], async_op=True))
        last_b[0] = b

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

    _phase("headline: buffers ready, warm-up")
    for k in range(args.warmup):
        step(k)
    drain()
    if final_gather or after_gather:
        # untimed: the first collective of this shape sets up RCCL's channels and buffers
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
    # ---- the timed region: EXACTLY args.steps steps between two barrier + synchronize fences
    # (nothing but the steps inside: the per-stream event spans behind `stream_call_ms` are taken in an untimed
    # repeat below -- six event records cost a 20-step run 5 % of its wall time)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    enqueue_s = time.perf_counter() - t0          # host time to issue all steps (diagnostic)
    drain()
    if final_gather and args.steps:
        # the one exchange step of the sharded job: rank 0 receives every shard's solutions
        torch.cuda.synchronize(dev)
        gather_to_root(xbuf[last_b[0]], fbuf[last_b[0]])
    fence()
    elapsed = time.perf_counter() - t0
    exchange_ms = None
    if after_gather and args.steps:
        # the exchange a single consumer of all shards would add: one batch's solutions and flags to rank 0
        tg = time.perf_counter()
        gather_to_root(xbuf[last_b[0]], fbuf[last_b[0]])
        fence()
        tgl = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=dev)
        dist.all_reduce(tgl, op=dist.ReduceOp.MAX)
        exchange_ms = 1e3 * float(tgl.item())
    # ---- what the timed region wrote, checked against the oracle (untimed; behind the exchange measurement so that
    # rank 0's check does not sit inside the other ranks' gather time): the last steps of every stream
    verification = None
    if rank == 0 and args.steps and not do_gather:
        verification = W.verify_steps(list(range(max(0, args.steps - max(nstreams, 2)), args.steps)))
    # average device time of one launch on its stream = event span of the stream / its launches (diagnostic,
    # untimed repeat of the same steps, at most 200 of them)
    nrep = min(args.steps, 200)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in streams]
    for e_, s_ in zip(ev0, streams):
        e_.record(s_)
    for k in range(nrep):
        step(k)
    for e_, s_ in zip(ev1, streams):
        e_.record(s_)
    drain()
    fence()
    calls_on = [len(range(i_, nrep, nstreams)) for i_ in range(nstreams)]
    span_ms = [ev0[i_].elapsed_time(ev1[i_]) if calls_on[i_] else 0.0 for i_ in range(nstreams)]
    stream_call_ms = sum(span_ms) / max(sum(calls_on), 1)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # world > 1: BASELINE configs 3 and 4 sharded over the ranks (every rank takes part; reported by rank 0 below)
    rank_cfgs = None
    if world > 1 and args.workload == "pendulum" and not args.no_configs and not args.f32 and not args.wave:
        _phase("multi-rank configs: mass_spring_3in and region discovery, one shard per rank")
        rank_cfgs = multi_rank_configs(torch, lmpc, dist, dev, local_rank, rank, world, n_local)

    if rank == 0:
        total = world * n_local * args.steps
        value = total / elapsed if elapsed > 0 else 0.0
        step_ms = 1e3 * elapsed / max(args.steps, 1)
        bytes_call = W.bytes_per * n_local
        gbs = lambda ms: bytes_call / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        dist_info, flop_est = W.work_distribution()
        wave = W.kernel.endswith("wave")
        dtype = "f32" if args.f32 else "f64"
        roof = {"bound": "hbm", "achieved": gbs(step_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs(step_ms) / HBM_PEAK_GBS, "traffic": None,
                "duration_used_ms": step_ms,
                "duration": "no single-launch section was run: wall time per step of the timed region "
                            f"({nstreams} batches in flight)"}
        # ---- one call at a time, cold HBM: THE roofline figure
        if args.steps and not args.no_single_launch:
            nsolo = int(min(200, max(100, args.steps)))      # (23 us each: a hundred calls cost nothing and settle the clocks)
            cold = W.single_launch(nsolo, resident=False)
            if cold[0] > 0 and cold[1] > 0:
                roof.update({"achieved": gbs(cold[1]), "frac": gbs(cold[1]) / HBM_PEAK_GBS, "duration_used_ms": cold[1],
                             "duration": "HIP-event duration of ONE lmpc_solve_batch_device call (all its kernels; "
                                         "timing events without the system-scope fence, hipEventDisableSystemFence, "
                                         "recorded by the library on the launch stream), calls "
                                         f"issued one at a time on one stream over {W.nrot} rotating batches (cold HBM: "
                                         f"{W.nrot * bytes_call / 2**20:.0f} MiB pass between two uses of a line)",
                             "launches_timed": cold[0], "kernel_ms": cold[1],
                             "screen_kernel_ms": cold[2], "iterate_kernel_ms": cold[3],
                             "screen_kernel_frac": gbs(cold[2]) / HBM_PEAK_GBS if cold[2] > 0 else None})
            if args.workload == "pendulum" and not args.f32 and not args.wave:
                try:
                    roof["several_batches_per_call"] = W.several_batches_section(8, 24)
                except Exception as e:                       # never fatal for the line
                    roof["several_batches_per_call"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            if not args.no_rotate:
                # cache-resident counterparts (ONE theta buffer reused): what round 1 reported
                res = W.single_launch(nsolo, resident=True)
                el_res = W.timed(max(args.steps, 60), 6, resident=True)
                res_step_ms = 1e3 * el_res / max(args.steps, 60)
                roof["cache_resident"] = {
                    "note": "one 56 MB theta buffer reused every step: inputs served from the 256 MiB Infinity Cache",
                    "single_launch": {"kernel_ms": res[1], "screen_kernel_ms": res[2], "iterate_kernel_ms": res[3],
                                      "achieved": gbs(res[1]), "frac": gbs(res[1]) / HBM_PEAK_GBS},
                    "pipelined": {"ms_per_step": res_step_ms, "value": n_local / (res_step_ms * 1e-3),
                                  "achieved": gbs(res_step_ms), "frac": gbs(res_step_ms) / HBM_PEAK_GBS,
                                  "batches_in_flight": nstreams}}
        roof["pipelined"] = {"ms_per_step": step_ms, "achieved": gbs(step_ms), "frac": gbs(step_ms) / HBM_PEAK_GBS,
                             "batches_in_flight": nstreams, "stream_call_ms": stream_call_ms,
                             "note": "algorithmic bytes of a step / wall time per step of the timed region: a pipeline "
                                     "throughput figure (launches of several batches overlap), not a kernel duration"}
        roof.update({"host_enqueue_ms_per_step": 1e3 * enqueue_s / max(args.steps, 1),
                     "algorithmic_bytes_per_solve": W.bytes_per,
                     "fp64_flop_per_solve_est": flop_est, "fp64_tflops_est": flop_est * value / 1e12,
                     "solves_per_s_per_cu": value / world / 256.0})
        if wave:       # diagnostics on the wavefront kernel: VALU bound, not HBM
            ms_ = roof["duration_used_ms"]
            ach = flop_est * n_local / (ms_ * 1e-3) / 1e12 if ms_ > 0 else 0.0
            roof.update({"bound": "valu", "achieved": ach, "peak": VALU_PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                         "frac": ach / VALU_PEAK_TFLOPS[dtype]})
        roof["traffic"], roof["traffic_info"] = traffic_stamp(args.workload)
        out = {
            "metric": "condensed-MPC QP solves/sec (batch 1e6 params), pendulum Nc=5"
                      if args.workload == "pendulum" else f"condensed-MPC QP solves/sec ({args.workload})",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            # `value` = the timed region: `batches_in_flight` independent batches kept in flight (a pipeline figure);
            # next to it the same workload ONE call at a time on one stream (cold HBM), which is what
            # roofline.achieved / frac are computed from
            "value_one_call_at_a_time": (world * n_local / (roof["kernel_ms"] * 1e-3)) if roof.get("kernel_ms") else None,
            "verified": (verification or {}).get("verified"),
            "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": describe(W), "batch_per_gpu": n_local, "kernel": W.kernel,
                       "batches_in_flight": nstreams, "rotating_batches": W.nrot,
                       "options": {**W.options, "_note": "every lmpc_set_option of the timed region's handles; all else "
                                                          "library defaults.  in_flight = k is the documented hint "
                                                          "'k batches in flight on this GPU' (include/lmpc_hip.h)"},
                       "verification": verification,
                       "inputs": ("cold HBM: every step reads a batch that has been evicted from the Infinity Cache"
                                  if W.nrot > 1 else "cache-resident: one theta buffer reused"),
                       "gather": ("all_gather(x, exitflag) over RCCL after every step, overlapped" if do_gather
                                  else (gather_impl["mode"] + "(x, exitflag) to rank 0 over RCCL once, after the last step") if final_gather
                                  else ("none inside the timed region (independent shards, results stay on their GPU); "
                                        + gather_impl["mode"] + "(x, exitflag) of one batch to rank 0 timed behind it") if after_gather
                                  else "none"),
                       **({"exchange": {"ms": exchange_ms, "bytes_per_rank": n_local * (nout * (4 if args.f32 else 8) + 4),
                                        "note": "one RCCL gather of a batch's solutions and exit flags to rank 0, outside "
                                                "the timed region: at this solve rate a GPU produces results faster than "
                                                "xGMI could collect them in one place, so a sharded job consumes them where "
                                                "they are"}} if exchange_ms is not None else {}),
                       **dist_info},
            "roofline": roof,
        }
        if args.workload == "pendulum" and not args.f32 and not args.wave:
            if world == 1:
                _phase("headline: multi-device ABI check")
                out["config"]["multi_abi"] = multi_abi_isolated(torch, min(n_local, 1_000_000))
            else:
                out["config"]["multi_abi"] = {"skipped": "torch.distributed run: the one-process multi-device entry point is "
                                                         "exercised in the N = 1 invocation when it sees several GPUs"}
        _phase("headline: timed region and single-launch sections done")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W.g, W.theta_h, nout, f32=args.f32, marginal=True)
            _phase("headline: cpu baseline done")
            if "marginal_cases" in out["cpu_baseline"]:
                out["config"]["marginal_cases"] = out["cpu_baseline"].pop("marginal_cases")
        # the headline line goes out NOW (and again, with the per-config summaries, as the last line): a failure or a
        # timeout in a side configuration cannot lose it
        run_configs = world == 1 and args.workload == "pendulum" and not args.no_configs and not args.f32 and not args.wave
        if rank_cfgs is not None:
            out["configs"] = rank_cfgs
        if args.workload == "pendulum" and not args.f32 and not args.wave and world == 1 and not args.no_configs:
            _phase("single call: lmpc_solve_one / mpc_compute_control with one state")
            try:
                out["single_call"] = single_call_config(lmpc, W.g, nout, cpu=not args.no_cpu_baseline)
            except Exception as e:
                out["single_call"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        emit(out, final=not run_configs)
        # ---- the other single-GPU BASELINE configurations, same process (driver-timed as part of this run)
        if run_configs:
            W.close()
            want_cpu = not args.no_cpu_baseline
            cfgs = {}

            def cfg(name, fn, *a, **kw):
                try:
                    cfgs[name] = fn(*a, **kw)
                except Exception as e:                        # reported in the line, never fatal for it
                    cfgs[name] = {"error": f"{type(e).__name__}: {e}"[:300], "verified": False}
                    _phase(f"config {name}: FAILED {cfgs[name]['error']}")
                return cfgs[name]
            cfg("pendulum_hard", side_config, torch, lmpc, "pendulum_hard", BATCH, dev, local_rank, 40, 5, False, want_cpu, 3.0)
            cfg("mass_spring", side_config, torch, lmpc, "mass_spring", BATCH, dev, local_rank, 20, 3, False, want_cpu, 3.0)
            cfg("mass_spring_3in", side_config, torch, lmpc, "mass_spring_3in", BATCH, dev, local_rank, 4, 1, False, want_cpu, 4.0)
            cfg("mass_spring_3in_feasible", side_config, torch, lmpc, "mass_spring_3in_feasible", BATCH, dev, local_rank, 4, 1, False, want_cpu, 3.0)
            _phase("config game_avi")
            cfg("game_avi", avi_config, torch, lmpc, dev, local_rank, BATCH, 10, 2, want_cpu)
            cfg("hybrid_f32", side_config, torch, lmpc, "hybrid", 100_000, dev, local_rank, 4, 1, True, want_cpu, 4.0)
            cfg("hybrid_f64", side_config, torch, lmpc, "hybrid", 100_000, dev, local_rank, 3, 1, False, False, 0.0)   # (the same searches in binary64)
            # the reference's only published numbers (plots, unstated hardware, generated C, one solve at a time
            # in closed loop): quoted beside the batched rate as context, not as a baseline
            ref_us = {50: 11.0, 75: 16.0, 100: 22.0, 125: 31.0}
            for n_ in (50, 75, 100, 125):
                c_ = cfg(f"pendulum_N{n_}", side_config, torch, lmpc, f"pendulum_N{n_}", 200_000, dev, local_rank, 3, PRIME_CALLS, False, want_cpu, 3.0)
                c_["reference_context"] = {"median_solve_time_us": ref_us[n_], "solves_per_s_one_thread": 1e6 / ref_us[n_],
                                           "source": "docs/src/assets/benchmark_scaling_time.png (benchmark.md:23), read off the "
                                                     "plot +-10 %, hardware unstated, state constraints of the benchmark script "
                                                     "unpublished (this fixture uses its own, see DESIGN.md)"}
            _phase("config closed_loop")
            cfg("closed_loop_pendulum", closed_loop_config, torch, lmpc, "pendulum", BATCH, 100, dev, local_rank, want_cpu)
            c_ = cfg("closed_loop_pendulum_N50", closed_loop_config, torch, lmpc, "pendulum_N50", 200_000, 100, dev, local_rank, want_cpu)
            try:
                c_["gram_scan"] = closed_loop_config(torch, lmpc, "pendulum_N50", 200_000, 100, dev, local_rank, False, gram=1)
            except Exception as e:
                c_["gram_scan"] = {"error": f"{type(e).__name__}: {e}"[:300], "verified": False}
            _phase("config closed_loop_game_avi")
            cfg("closed_loop_game_avi", avi_closed_loop_config, torch, lmpc, dev, local_rank, 200_000, 100, want_cpu)
            _phase("config region_discovery")
            cfg("region_discovery", region_discovery_config, torch, lmpc, dev, local_rank, BATCH, want_cpu)
            out["configs"] = cfgs
            emit(out, final=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
