"""Sampling-based discovery of explicit-MPC critical regions on top of the batched solve.

The reference builds explicit controllers with ParametricDAQP's exact region enumeration
(/root/reference/src/explicit.jl:23-48, a third-party solver that never calls `solve`).  What the
batched backend can contribute is the data-parallel part: solve the condensed QP on a large sample
of the parameter range (`ParameterRange`, /root/reference/src/types.jl:184-224 ->
`range2region` utils.jl:285-289), collect the distinct optimal active sets (one per critical
region that the sample hit) and derive each region's affine control law from the KKT system of
that active set.  This is a candidate generator, NOT a replacement for exact enumeration: regions
the sample misses are not found, and no region boundaries are certified.

`discover_regions` / `certify_sampled` are host-side bookkeeping around `BatchedQP.solve`; `discover_regions_device`
keeps the whole per-sample part on the GPU (sample, solve, reduction to the distinct masks:
`lmpc_distinct_active_sets_device`, csrc/lmpc_regions.hip) and brings back one row per region.
"""
from __future__ import annotations

import numpy as np


def sample_range(lb, ub, n, seed=0):
    """Uniform sample of the box [lb, ub] (zero-width coordinates stay fixed)."""
    lb = np.asarray(lb, float).reshape(-1)
    ub = np.asarray(ub, float).reshape(-1)
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray(lb + (ub - lb) * rng.random((int(n), lb.size)))


def unique_active_sets(active, exitflag=None):
    """Distinct active-set masks among the solved problems.

    Returns (masks (R x words uint64), counts (R), first_index (R)) sorted by decreasing count."""
    active = np.ascontiguousarray(np.asarray(active, np.uint64).reshape(len(active), -1))
    idx = np.arange(active.shape[0])
    if exitflag is not None:
        idx = idx[np.asarray(exitflag) >= 1]
    if idx.size == 0:
        return np.zeros((0, active.shape[1]), np.uint64), np.zeros(0, int), np.zeros(0, int)
    masks, first, counts = np.unique(active[idx], axis=0, return_index=True, return_counts=True)
    order = np.argsort(-counts, kind="stable")
    return masks[order], counts[order], idx[first[order]]


def mask_to_sets(mask, m):
    """(upper-active rows, lower-active rows) of one mask (bit j upper, bit m+j lower)."""
    mask = np.asarray(mask, np.uint64).reshape(-1)
    bit = lambda b: (int(mask[b >> 6]) >> (b & 63)) & 1
    return [j for j in range(m) if bit(j)], [j for j in range(m) if bit(m + j)]


def affine_law(H, f, f_theta, A, bu, bl, W, mask, nout=None):
    """U*(theta) = Fz theta + gz on the critical region whose active set is `mask`.

    KKT of  min 1/2 U'HU + (f + f_theta th)'U  s.t.  E U = e + Wa th  (active rows at their bound):
        [H E'; E 0] [U; lam] = [-(f + f_theta th); e + Wa th]."""
    H = np.asarray(H, float)
    n = H.shape[0]
    f_theta = np.asarray(f_theta, float).reshape(n, -1)
    nth = f_theta.shape[1]
    bu = np.asarray(bu, float).reshape(-1)
    bl = np.asarray(bl, float).reshape(-1)
    m = bu.size
    A = np.asarray(A, float).reshape(-1, n)
    ms = m - A.shape[0]
    Afull = np.vstack([np.eye(n)[:ms], A])
    W = np.asarray(W, float).reshape(m, nth)
    up, lo = mask_to_sets(mask, m)
    rows = up + lo
    E = Afull[rows]
    e = np.concatenate([bu[up], bl[lo]])
    Wa = W[rows]
    k = len(rows)
    KKT = np.block([[H, E.T], [E, np.zeros((k, k))]])
    rhs_th = np.vstack([-f_theta, Wa])
    rhs_0 = np.concatenate([-np.asarray(f, float).reshape(n), e])
    sol_th = np.linalg.solve(KKT, rhs_th)
    sol_0 = np.linalg.solve(KKT, rhs_0)
    nout = n if nout is None else nout
    return sol_th[:nout], sol_0[:nout]


def merge_region_tables(masks, counts, first, n_solved, group, capacity=None, index_offset=0):
    """Merge of the per-rank tables of distinct active sets across a torch.distributed group, as TENSORS: every rank
    contributes one fixed-size table -- `capacity` rows of [mask words, count, first index] plus a header row [sets,
    problems solved] -- in ONE all_gather_into_tensor on the device the tables live on (RCCL for the tables the device
    pipeline leaves in HBM, gloo for CPU tensors), and the gathered rows are reduced on that device: equal masks are
    found by torch.unique, their counts added, the smallest (global) first index kept.  What crosses the fabric is
    (capacity + 1) x (words + 2) x 8 bytes per rank -- 24 kB for the pendulum's 45 sets at capacity 1024 -- against
    the pickled object lists of round 4.

    masks (R x words, int64 / uint64 tensor or array), counts (R), first (R) or None, n_solved (int); `index_offset` is
    added to this rank's first indices (its shard's offset in the global sample).  Returns on every rank
    (masks uint64 array, counts, first_index, n_solved), sets sorted by decreasing count, ties by first index.
    A rank with more than `capacity` sets makes every rank repeat the exchange with four times the room."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t_masks = torch.as_tensor(np.ascontiguousarray(np.asarray(masks).view(np.int64))) if not torch.is_tensor(masks) else masks
    dev = t_masks.device
    t_masks = t_masks.reshape(-1, t_masks.shape[-1] if t_masks.dim() > 1 else 1).to(torch.int64)
    R, words = int(t_masks.shape[0]), int(t_masks.shape[1])
    t_counts = torch.as_tensor(np.asarray(counts, np.int64)).to(dev) if not torch.is_tensor(counts) else counts.to(torch.int64)
    if first is None:
        t_first = torch.full((R,), -1, dtype=torch.int64, device=dev)
    else:
        t_first = (torch.as_tensor(np.asarray(first, np.int64)).to(dev) if not torch.is_tensor(first) else first.to(torch.int64)) + int(index_offset)
    # room: agreed on by all ranks (the largest table, rounded up to a power of two, at least 64 rows)
    need = torch.tensor([R], dtype=torch.int64, device=dev)
    dist.all_reduce(need, op=dist.ReduceOp.MAX, group=group)
    cap = int(capacity) if capacity else 64
    while cap < int(need.item()):
        cap *= 4
    table = torch.zeros((cap + 1, words + 2), dtype=torch.int64, device=dev)
    table[0, 0], table[0, 1] = R, int(n_solved)
    if R:
        table[1:R + 1, :words] = t_masks
        table[1:R + 1, words] = t_counts[:R]
        table[1:R + 1, words + 1] = t_first[:R]
    gathered = torch.empty((world * (cap + 1), words + 2), dtype=torch.int64, device=dev)
    try:
        dist.all_gather_into_tensor(gathered, table, group=group)
    except (RuntimeError, NotImplementedError):          # (a backend without the flat form: the list form, same bytes)
        parts = [torch.empty_like(table) for _ in range(world)]
        dist.all_gather(parts, table, group=group)
        gathered = torch.cat(parts, 0)
    gathered = gathered.reshape(world, cap + 1, words + 2)
    nsets = gathered[:, 0, 0]
    solved = int(gathered[:, 0, 1].sum().item())
    rows = gathered[:, 1:, :]
    valid = torch.arange(cap, device=dev)[None, :] < nsets[:, None]
    rows = rows[valid]
    if rows.shape[0] == 0:
        return np.zeros((0, words), np.uint64), np.zeros(0, np.int64), np.zeros(0, np.int64), solved
    um, inv = torch.unique(rows[:, :words], dim=0, return_inverse=True)
    uc = torch.zeros(um.shape[0], dtype=torch.int64, device=dev).index_add_(0, inv, rows[:, words])
    big = torch.iinfo(torch.int64).max
    fi = rows[:, words + 1]
    uf = torch.full((um.shape[0],), big, dtype=torch.int64, device=dev).scatter_reduce_(0, inv, torch.where(fi < 0, torch.full_like(fi, big), fi), reduce="amin")
    um_h, uc_h, uf_h = um.cpu().numpy().view(np.uint64), uc.cpu().numpy(), uf.cpu().numpy()
    uf_h = np.where(uf_h == big, -1, uf_h)
    order = np.lexsort((uf_h, -uc_h))
    return um_h[order], uc_h[order], uf_h[order], solved


def discover_regions(solve_fn, theta, group=None, index_offset=0):
    """Distinct optimal active sets over a parameter sample, optionally merged across ranks.

    solve_fn(theta) -> (x, exitflag, iters, active) is `BatchedQP.solve` of a handle on this rank's
    GPU; with `group` given (torch.distributed) every rank passes ITS shard of the sample, the
    per-rank tables of distinct masks are exchanged as tensors (`merge_region_tables`: one all_gather_into_tensor) and
    merged on every rank -- the sample grid shards across the GPUs of a node exactly like any other batch.
    `index_offset`: this rank's offset in the global sample (first indices come back global)."""
    x, ef, it, act = solve_fn(theta)
    masks, counts, first = unique_active_sets(act, ef)
    solved = int(np.sum(np.asarray(ef) >= 1))
    if group is not None:
        masks, counts, first, solved = merge_region_tables(masks, counts, first, solved, group, index_offset=index_offset)
    return {"masks": masks, "counts": counts, "first_index": first, "n_solved": solved}


class DeviceRegionSampler:
    """One step of the sampling-based region discovery as ONE enqueue (`lmpc_discover_regions_device`): solve the
    resident sample with the masks kept on the device, reduce them to the distinct sets there, publish the sets into a
    block of mapped host memory; the host synchronises once and reads the block.  All device buffers are allocated
    here, once, for `nsamples` samples; `enqueue` / `result` can be used apart (enqueue the next sample's draw in
    between), `run` is both.  A step that finds more distinct sets than there is room for is repeated with four times
    the room."""

    def __init__(self, qp, nsamples, capacity=4096, device=None):
        import torch
        self.qp, self.N, self.capacity = qp, int(nsamples), int(capacity)
        self.dev = torch.device("cuda", qp.device if getattr(qp, "device", None) is not None else torch.cuda.current_device()) \
            if device is None else device
        d = self.dev
        self.x = torch.empty((self.N, qp.nout), dtype=torch.float64, device=d)
        self.ef = torch.empty(self.N, dtype=torch.int32, device=d)
        self.act = torch.empty((self.N, qp.words), dtype=torch.int64, device=d)
        self._alloc_sets()
        self._res = None
        self._theta = None

    def _alloc_sets(self):
        import torch
        d, c = self.dev, self.capacity
        self.masks = torch.empty((c, self.qp.words), dtype=torch.int64, device=d)
        self.counts = torch.empty(c, dtype=torch.int64, device=d)
        self.first = torch.empty(c, dtype=torch.int64, device=d)
        self.nset = torch.zeros(1, dtype=torch.int32, device=d)
        torch.cuda.synchronize(d)

    def enqueue(self, theta, stream=None):
        import ctypes
        from ._cabi import lib, check
        if not (theta.is_cuda and theta.is_contiguous() and theta.shape == (self.N, self.qp.nth)):
            raise ValueError("theta must be a contiguous (nsamples, nth) float64 CUDA tensor")
        vp = ctypes.c_void_p
        res = ctypes.c_void_p()
        L = lib()
        L.lmpc_discover_regions_device.restype = ctypes.c_int
        check(L.lmpc_discover_regions_device(
            self.qp._h, ctypes.c_int64(self.N),
            vp(theta.data_ptr()), vp(self.x.data_ptr()), vp(self.ef.data_ptr()),
            vp(self.act.data_ptr()), ctypes.c_int32(self.capacity), vp(self.masks.data_ptr()), vp(self.counts.data_ptr()),
            vp(self.first.data_ptr()), vp(self.nset.data_ptr()), ctypes.byref(res),
            vp(int(stream)) if stream else None), self.qp._h)
        self._res, self._theta, self._stream = res.value, theta, stream

    def result(self):
        """-> (masks (R x words uint64), counts, first_index, n_solved), sets sorted by decreasing count, ties by first
        index (the order `unique_active_sets` gives)."""
        import ctypes
        import torch
        if self._stream:
            torch.cuda.ExternalStream(int(self._stream), device=self.dev).synchronize()
        else:
            torch.cuda.synchronize(self.dev)
        w = self.qp.words
        head = np.ctypeslib.as_array((ctypes.c_longlong * 4).from_address(self._res))
        nfound, over = int(head[0]), int(head[1])
        if over == 2:
            from ._cabi import LmpcError
            raise LmpcError(-102, "lmpc_discover_regions_device: table slot never published")
        if nfound > self.capacity or over != 0:
            self.capacity *= 4
            self._alloc_sets()
            self.enqueue(self._theta, self._stream)
            return self.result()
        solved = int(head[2])
        rows = np.ctypeslib.as_array((ctypes.c_longlong * (nfound * (w + 2))).from_address(self._res + 32)).reshape(nfound, w + 2).copy() \
            if nfound else np.zeros((0, w + 2), np.int64)
        m = np.ascontiguousarray(rows[:, :w]).view(np.uint64)
        c, f = rows[:, w].copy(), rows[:, w + 1].copy()
        order = np.lexsort((f, -c))
        return m[order], c[order], f[order], solved

    def run(self, theta, stream=None):
        self.enqueue(theta, stream)
        return self.result()


def discover_regions_device(qp, lb, ub, nsamples, seed=0, group=None, capacity=65536, theta=None, index_offset=0):
    """`discover_regions` with everything per-sample on the GPU: the sample of the box [lb, ub] is drawn on the device
    (torch generator, seeded), solved by `lmpc_solve_batch_device` with the active-set masks kept there, and reduced to
    its distinct masks by `lmpc_distinct_active_sets_device`; what crosses PCIe is one row per region that was hit.
    `theta`: a ready-made (N, nth) float64 CUDA tensor instead of the drawn sample.  With a torch.distributed `group`
    every rank passes its own seed / shard; the per-rank tables of distinct sets stay ON THE DEVICE and are exchanged
    by one all_gather_into_tensor (RCCL over xGMI) and merged there (`merge_region_tables`)."""
    import torch
    dev = torch.device("cuda", qp.device if hasattr(qp, "device") and qp.device is not None else torch.cuda.current_device())
    if theta is None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed))
        lo = torch.as_tensor(np.asarray(lb, float).reshape(-1), dtype=torch.float64, device=dev)
        hi = torch.as_tensor(np.asarray(ub, float).reshape(-1), dtype=torch.float64, device=dev)
        theta = lo + (hi - lo) * torch.rand((int(nsamples), lo.numel()), dtype=torch.float64, device=dev, generator=gen)
    sampler = DeviceRegionSampler(qp, int(theta.shape[0]), capacity=min(int(capacity), 4096))
    masks, counts, first, solved = sampler.run(theta)
    out = {"masks": masks, "counts": counts, "first_index": first, "n_solved": solved, "theta": theta}
    if group is not None:
        # the device tables as the pipeline left them (sampler.masks / counts / first, `nfound` rows)
        nfound = len(counts)
        gm, gc, gf, gs = merge_region_tables(sampler.masks[:nfound], sampler.counts[:nfound], sampler.first[:nfound], solved,
                                             group, index_offset=index_offset)
        out.update({"masks": gm, "counts": gc, "first_index": gf, "n_solved": gs})
    return out


def certify_sampled(solve_fn, theta, group=None):
    """Empirical iteration-complexity certificate over a parameter sample.

    The reference's `certify` (/root/reference/src/certify.jl:18-30) hands the mpQP to ASCertain, which
    partitions the parameter range into regions of equal solver behaviour and returns the exact worst-case
    iteration count (`max_iterations`, `partition`).  The batched backend's share of that job is the
    data-parallel one: solve the sample, report the largest iteration count that occurred (a LOWER bound
    of the certified maximum -- points the sample misses are not covered), where it occurred, the
    iteration histogram, and the distinct (final active set, iteration count) pairs that were seen
    (each is a union of ASCertain's partition cells, which are finer -- equal working-set SEQUENCE;
    the reference's test finds more than 100 of them for the pendulum, runtests.jl:199-204, where the
    sample sees about 50 pairs).

    solve_fn / group as in `discover_regions`: with a torch.distributed group every rank passes its
    shard of the sample and all ranks return the merged result."""
    theta = np.asarray(theta, float)
    x, ef, it, act = solve_fn(theta)
    ef = np.asarray(ef)
    it = np.asarray(it).astype(np.int64)
    act = np.ascontiguousarray(np.asarray(act, np.uint64).reshape(len(ef), -1))
    ok = ef >= 1
    hist = np.bincount(it[ok], minlength=1)
    if ok.any():
        worst = int(np.flatnonzero(ok)[np.argmax(it[ok])])
        max_it, arg = int(it[worst]), theta[worst].copy()
    else:
        max_it, arg = 0, None
    cells = np.unique(np.hstack([act[ok], it[ok, None].astype(np.uint64)]), axis=0) if ok.any() \
        else np.zeros((0, act.shape[1] + 1), np.uint64)
    flags = {int(k): int(v) for k, v in zip(*np.unique(ef, return_counts=True))}
    if group is not None:
        import torch.distributed as dist
        gathered = [None] * dist.get_world_size(group)
        dist.all_gather_object(gathered, (max_it, arg, hist, cells, flags), group=group)
        best = max(range(len(gathered)), key=lambda r_: (gathered[r_][0], -r_))
        max_it, arg = gathered[best][0], gathered[best][1]
        L = max(len(g_[2]) for g_ in gathered)
        hist = sum(np.pad(g_[2], (0, L - len(g_[2]))) for g_ in gathered)
        nonempty = [g_[3] for g_ in gathered if len(g_[3])]
        cells = np.unique(np.concatenate(nonempty, 0), axis=0) if nonempty else cells
        flags = {}
        for g_ in gathered:
            for k, v in g_[4].items():
                flags[k] = flags.get(k, 0) + v
    return {"max_iterations": max_it, "argmax_theta": arg, "iterations_hist": np.asarray(hist),
            "cells": cells, "n_cells": int(len(cells)), "exitflags": flags}
