"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): scatter the batch from rank 0, every rank solves ITS shard,
gather the solutions — must equal the single-process result.  On the CPU the per-rank solver is the oracle (this is a test
of the sharding plumbing; on GPUs each rank drives its own lexls handle, see bench.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lexls_amd import problems as P, sharding
    from oracle import oracle_ctypes as oc
    n, dims = 12, [4, 4, 5]
    cap = sum(dims)
    lod = P.lse_batch(4242, batch, n, dims) if rank == 0 else None
    mine = sharding.scatter_problems(lod, batch, n, cap)
    lo, hi = sharding.shard_range(batch, rank, world)
    assert mine.shape[0] == hi - lo
    x = torch.from_numpy(oc.lse_run(mine.numpy(), dims, n)["x"]) if hi > lo else torch.zeros((0, n), dtype=torch.float64)
    full = sharding.gather_solutions(x, batch, n)
    dist.barrier()
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


def _run(batch, tmp_path):
    out = str(tmp_path / f"x_{batch}.npy")
    mp.spawn(_worker, args=(2, _free_port(), batch, out), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from lexls_amd import problems as P
    from oracle import oracle_ctypes as oc
    ref = oc.lse_run(P.lse_batch(4242, batch, 12, [4, 4, 5]), [4, 4, 5], 12)["x"]
    np.testing.assert_array_equal(np.load(out), ref)


def test_scatter_solve_gather_even(tmp_path):
    _run(10, tmp_path)


def test_scatter_solve_gather_ragged(tmp_path):
    _run(7, tmp_path)


# ---- LexLSI batches (BASELINE configs[4]): contiguous instance blocks per rank, no exchange between iterations ------------------
def _lsi_worker(rank, world, port, batch, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lexls_amd import lexlsi, problems as P, sharding
    from oracle import oracle_ctypes as oc
    n, dims = 10, [4, 3, 3]
    probs = [P.lsi_problem(777 + b, n, dims) for b in range(batch)]
    dims_a, types_a = lexlsi.flatten(n, probs[0])[:2]
    pk_root = lexlsi.pack_batch(n, probs) if rank == 0 else None
    mine = sharding.scatter_lsi_batch(pk_root, n, dims_a, types_a, batch)
    lo, hi = sharding.shard_range(batch, rank, world)
    assert mine.batch == hi - lo
    if rank == 0:  # the scattered block is the split of the root's batch
        own = sharding.split_packed_batch(pk_root, 0, world)
        assert np.array_equal(own.data, mine.data) and np.array_equal(own.var_index, mine.var_index)
    total = int(dims_a.sum())
    x, info, act, v = np.zeros((mine.batch, n)), [], np.zeros((mine.batch, total), np.uint8), np.zeros((mine.batch, total))
    for i in range(mine.batch):  # on a GPU: ONE lexlsi.LsiBatch.run over the block; here the CPU oracle, instance by instance
        objs = lexlsi.unflatten(n, mine.dims, mine.types, mine.data[i], None if mine.var_index is None else mine.var_index[i])
        r = oc.lsi_run(n, objs)
        x[i], act[i], v[i] = r["x"], np.concatenate(r["active"]), np.concatenate(r["v"])
        info.append(r["info"])
    full = sharding.gather_lsi_results(dict(x=x, info=info, active=act, v=v), batch)
    dist.barrier()
    if rank == 0:
        np.savez(out_path, **full)
    dist.destroy_process_group()


def _run_lsi(batch, tmp_path):
    out = str(tmp_path / f"lsi_{batch}.npz")
    mp.spawn(_lsi_worker, args=(2, _free_port(), batch, out), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from lexls_amd import problems as P
    from oracle import oracle_ctypes as oc
    got = np.load(out)
    for b in range(batch):
        r = oc.lsi_run(10, P.lsi_problem(777 + b, 10, [4, 3, 3]))
        np.testing.assert_array_equal(got["x"][b], r["x"])
        np.testing.assert_array_equal(got["active"][b], np.concatenate(r["active"]))
        np.testing.assert_array_equal(got["v"][b], np.concatenate(r["v"]))
        assert got["info"][b].tolist() == [r["info"][k] for k in ("status", "iterations", "activations", "deactivations", "factorizations", "total_rank")]


def test_lsi_scatter_solve_gather_even(tmp_path):
    _run_lsi(6, tmp_path)


def test_lsi_scatter_solve_gather_ragged(tmp_path):
    _run_lsi(5, tmp_path)
