"""world_size-2 gloo tests of the sharding / gather layer (the N>1 path of bench.py).

The per-shard solutions are produced by the CPU oracle here (no GPU in this container); what is
under test is the partition + exchange: contiguous shards, ragged remainder, all-gather and
gather-to-root, and that the gathered result equals the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, oracle_ldp_from


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, out_dir):
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import linearmpc_jl_amd as lmpc
    from oracle import ldp as oldp
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    theta = g["theta"][:n_total]
    lo, hi = lmpc.shard_bounds(n_total, world, rank)
    X, ef, _, _ = oldp.solve_batch(L, theta[lo:hi])
    Xl, efl = torch.from_numpy(X), torch.from_numpy(ef)
    Xall = lmpc.gather_shards(Xl, n_total)                     # all-gather
    efall = lmpc.gather_shards(efl, n_total)
    Xroot = lmpc.gather_shards(Xl, n_total, dst=0)             # gather to rank 0
    assert (Xroot is None) == (rank != 0)
    np.save(os.path.join(out_dir, f"X{rank}.npy"), Xall.numpy())
    np.save(os.path.join(out_dir, f"ef{rank}.npy"), efall.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "Xroot.npy"), Xroot.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [512, 333, 1])
def test_sharded_gather_equals_single_process(tmp_path, n_total):
    from oracle import ldp as oldp
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    X, ef, _, _ = oldp.solve_batch(oracle_ldp_from(pk), g["theta"][:n_total])
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"X{r}.npy"), X)
        assert np.array_equal(np.load(tmp_path / f"ef{r}.npy"), ef)
    assert np.array_equal(np.load(tmp_path / "Xroot.npy"), X)


def _region_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import linearmpc_jl_amd as lmpc
    from oracle import ldp as oldp
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0]); ub = -lb; ub[5] = 0.0
    theta = lmpc.explicit.sample_range(lb, ub, 20000, seed=3)
    lo, hi = lmpc.shard_bounds(len(theta), world, rank)
    out = lmpc.explicit.discover_regions(lambda th: oldp.solve_batch(L, th), theta[lo:hi], group=dist.group.WORLD, index_offset=lo)
    np.save(os.path.join(out_dir, f"masks{rank}.npy"), out["masks"])
    np.save(os.path.join(out_dir, f"counts{rank}.npy"), out["counts"])
    np.save(os.path.join(out_dir, f"first{rank}.npy"), out["first_index"])
    np.save(os.path.join(out_dir, f"solved{rank}.npy"), np.array([out["n_solved"]]))
    # the exchange itself, on tensors as the device pipeline leaves them (int64 masks, unsorted, a small capacity that
    # has to grow; an EMPTY table on rank 1)
    import torch
    x, ef, it, act = oldp.solve_batch(L, theta[lo:hi])
    m, c, f = lmpc.explicit.unique_active_sets(act, ef)
    perm = np.random.default_rng(rank).permutation(len(c))
    tm = torch.from_numpy(np.ascontiguousarray(m[perm]).view(np.int64))
    gm, gc, gf, gs = lmpc.explicit.merge_region_tables(tm, torch.from_numpy(c[perm].astype(np.int64)), torch.from_numpy(f[perm].astype(np.int64)),
                                                       int((ef >= 1).sum()), dist.group.WORLD, capacity=4, index_offset=lo)
    assert np.array_equal(gm, out["masks"]) and np.array_equal(gc, out["counts"]) and np.array_equal(gf, out["first_index"]) and gs == out["n_solved"]
    keep = slice(0, 0) if rank == 1 else slice(None)
    em, ec, ef_, es = lmpc.explicit.merge_region_tables(tm[keep], torch.from_numpy(c[perm].astype(np.int64))[keep],
                                                        torch.from_numpy(f[perm].astype(np.int64))[keep], 0 if rank == 1 else 7,
                                                        dist.group.WORLD, index_offset=lo)
    np.save(os.path.join(out_dir, f"half{rank}.npy"), np.array([len(ec), int(ec.sum()), es]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_region_discovery_equals_single_process(tmp_path):
    import linearmpc_jl_amd as lmpc
    from oracle import ldp as oldp
    world = 2
    mp.spawn(_region_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0]); ub = -lb; ub[5] = 0.0
    theta = lmpc.explicit.sample_range(lb, ub, 20000, seed=3)
    ref = lmpc.explicit.discover_regions(lambda th: oldp.solve_batch(L, th), theta)
    key = lambda M, c: sorted((tuple(r), int(k)) for r, k in zip(M.tolist(), c.tolist()))
    for r in range(world):
        assert key(np.load(tmp_path / f"masks{r}.npy"), np.load(tmp_path / f"counts{r}.npy")) == \
            key(ref["masks"], ref["counts"])
        # tensor exchange: same order on every rank (decreasing count, ties by first index), GLOBAL first indices, the
        # solved counts added up
        o = np.lexsort((ref["first_index"], -ref["counts"]))
        assert np.array_equal(np.load(tmp_path / f"masks{r}.npy"), ref["masks"][o])
        assert np.array_equal(np.load(tmp_path / f"first{r}.npy"), ref["first_index"][o])
        assert int(np.load(tmp_path / f"solved{r}.npy")[0]) == ref["n_solved"]
    # an empty table on one rank: what comes back is rank 0's table
    lo0, hi0 = lmpc.shard_bounds(len(theta), world, 0)
    x0, ef0, _, act0 = oldp.solve_batch(L, theta[lo0:hi0])
    m0, c0, _ = lmpc.explicit.unique_active_sets(act0, ef0)
    for r in range(world):
        assert np.load(tmp_path / f"half{r}.npy").tolist() == [len(c0), int(c0.sum()), 7]



def _certify_worker(rank, world, port, out_dir):
    import pickle
    import sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import linearmpc_jl_amd as lmpc
    from oracle import ldp as oldp
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0]); ub = -lb; ub[5] = 0.0
    theta = lmpc.explicit.sample_range(lb, ub, 20001, seed=5)
    lo, hi = lmpc.shard_bounds(len(theta), world, rank)
    out = lmpc.explicit.certify_sampled(lambda th: oldp.solve_batch(L, th), theta[lo:hi], group=dist.group.WORLD)
    with open(os.path.join(out_dir, f"cert{rank}.pkl"), "wb") as fh:
        pickle.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_sampled_certificate_equals_single_process(tmp_path):
    # the caller side of /root/reference/src/certify.jl on the batched path, sharded like any batch
    import pickle
    import linearmpc_jl_amd as lmpc
    from oracle import ldp as oldp
    world = 2
    mp.spawn(_certify_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("pendulum")
    pk = dict(g); pk["sense"] = g["senses"]
    L = oracle_ldp_from(pk)
    lb = np.array([-20.0] * 4 + [-20.0, 0.0] + [-2.0]); ub = -lb; ub[5] = 0.0
    theta = lmpc.explicit.sample_range(lb, ub, 20001, seed=5)
    ref = lmpc.explicit.certify_sampled(lambda th: oldp.solve_batch(L, th), theta)
    # (the reference's exact partition, by working-set SEQUENCE, has > 100 regions: runtests.jl:199-204; the
    # sample's cells -- final active set x iteration count -- are unions of those: ~40-50 on this range)
    assert ref["max_iterations"] >= 6 and ref["n_cells"] >= 30
    x, ef, it, _ = oldp.solve_batch(L, ref["argmax_theta"][None])
    assert it[0] == ref["max_iterations"]
    for r in range(world):
        out = pickle.load(open(tmp_path / f"cert{r}.pkl", "rb"))
        assert out["max_iterations"] == ref["max_iterations"] and out["n_cells"] == ref["n_cells"]
        assert np.array_equal(out["cells"], ref["cells"]) and np.array_equal(out["iterations_hist"], ref["iterations_hist"])
        assert out["exitflags"] == ref["exitflags"]
        x, ef, it, _ = oldp.solve_batch(L, out["argmax_theta"][None])
        assert it[0] == ref["max_iterations"]
