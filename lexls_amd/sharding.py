"""Batch sharding across the GPUs of one node: problems are independent, so rank r owns a contiguous block of
problem indices and factorizes it with its own handle — there is NO collective between factorizations.

`scatter_problems` / `gather_solutions` serve callers whose whole batch starts on rank 0 (north_star: "RCCL over xGMI
used only to scatter problem blocks and gather solutions"); they use torch.distributed (backend "nccl" = RCCL on the
GPUs, "gloo" in the CPU tests) and must stay OUTSIDE any timed region: the root's 7 xGMI links (~1.07 TB/s) are slower
than one GPU's HBM."""
from __future__ import annotations

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one (earlier ranks get the extra item)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int) -> list[int]:
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def scatter_problems(lod_root, batch: int, nvar: int, cap: int, device=None, src: int = 0):
    """Rank `src` holds `lod_root` (batch, nVar+1, cap) float64 (numpy or torch); every rank gets its shard as a torch tensor."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = shard_sizes(batch, world)
    mine = torch.empty((sizes[rank], nvar + 1, cap), dtype=torch.float64, device=device)
    chunks = None
    if rank == src:
        t = torch.as_tensor(np.ascontiguousarray(lod_root) if isinstance(lod_root, np.ndarray) else lod_root).to(device)
        if tuple(t.shape) != (batch, nvar + 1, cap):
            raise ValueError("lod_root has the wrong shape")
        offs = np.cumsum([0] + sizes)
        chunks = [t[offs[r]:offs[r + 1]].contiguous() for r in range(world)]
    if len(set(sizes)) == 1:
        dist.scatter(mine, chunks, src=src)
    else:  # ragged shards: point-to-point (scatter needs equal sizes)
        if rank == src:
            for r in range(world):
                if r == src:
                    mine.copy_(chunks[r])
                else:
                    dist.send(chunks[r], dst=r)
        else:
            dist.recv(mine, src=src)
    return mine


def gather_solutions(x_local, batch: int, nvar: int, dst: int = 0):
    """Inverse of scatter_problems for the solutions: rank `dst` returns the (batch, nVar) tensor, others None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = shard_sizes(batch, world)
    x_local = x_local.contiguous()
    if len(set(sizes)) == 1:
        out = [torch.empty_like(x_local) for _ in range(world)] if rank == dst else None
        dist.gather(x_local, out, dst=dst)
        return torch.cat(out) if rank == dst else None
    if rank == dst:
        parts = []
        for r in range(world):
            if r == dst:
                parts.append(x_local)
            else:
                buf = torch.empty((sizes[r], nvar), dtype=x_local.dtype, device=x_local.device)
                dist.recv(buf, src=r)
                parts.append(buf)
        return torch.cat(parts)
    dist.send(x_local, dst=dst)
    return None


# ---- LexLSI batches (BASELINE configs[4]) -------------------------------------------------------------------------
# Instances are independent: rank r owns the contiguous block shard_range(batch, r, world) of the instances and runs ONE lock-step
# batch object (lexls_lsi_batch_*) on its GPU.  The constraint data is scattered once; an active-set iteration moves nothing
# between ranks.  Results (x, info, working sets, residuals) are gathered on the root.

def split_packed_batch(pk, rank: int, world: int):
    """this rank's contiguous instance block of a lexlsi.PackedBatch (views of the root's arrays, no copy)"""
    from .lexlsi import PackedBatch
    lo, hi = shard_range(pk.batch, rank, world)
    vi = None if pk.var_index is None else np.ascontiguousarray(pk.var_index[lo:hi])
    return PackedBatch(pk.nvar, pk.dims, pk.types, np.ascontiguousarray(pk.data[lo:hi]), vi)


def _scatter_rows(root_array, batch: int, row_shape, dtype, device=None, src: int = 0):
    """rows [lo, hi) of a (batch, *row_shape) array held by rank `src` -> every rank (equal shards: scatter, ragged: send/recv)"""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = shard_sizes(batch, world)
    mine = torch.empty((sizes[rank],) + tuple(row_shape), dtype=dtype, device=device)
    chunks = None
    if rank == src:
        t = torch.as_tensor(np.ascontiguousarray(root_array)).to(device)
        offs = np.cumsum([0] + sizes)
        chunks = [t[offs[r]:offs[r + 1]].contiguous() for r in range(world)]
    if len(set(sizes)) == 1:
        dist.scatter(mine, chunks, src=src)
    elif rank == src:
        for r in range(world):
            if r == src:
                mine.copy_(chunks[r])
            else:
                dist.send(chunks[r], dst=r)
    else:
        dist.recv(mine, src=src)
    return mine


def _gather_rows(local, batch: int, dst: int = 0):
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    sizes = shard_sizes(batch, world)
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, out, dst=dst)
        return torch.cat(out) if rank == dst else None
    if rank == dst:
        parts = []
        for r in range(world):
            if r == dst:
                parts.append(local)
            else:
                buf = torch.empty((sizes[r],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
                dist.recv(buf, src=r)
                parts.append(buf)
        return torch.cat(parts)
    dist.send(local, dst=dst)
    return None


def scatter_lsi_batch(pk_root, nvar: int, dims, types, batch: int, device=None, src: int = 0):
    """Rank `src` holds the whole PackedBatch (others pass None); every rank gets the PackedBatch of its instance block.
    `dims` / `types` (the structure, known everywhere) size the receive buffers."""
    import torch
    from .lexlsi import PackedBatch
    dims = np.ascontiguousarray(dims, np.uint32)
    types = np.ascontiguousarray(types, np.int32)
    per = sum(int(d) * (2 if t == 1 else nvar + 2) for d, t in zip(dims, types))  # doubles per instance (lexls_hip.h: flat layout)
    data = _scatter_rows(None if pk_root is None else pk_root.data, batch, (per,), torch.float64, device, src)
    nsimple = int(dims[0]) if len(types) and types[0] == 1 else 0
    vi = None
    if nsimple:
        vi = _scatter_rows(None if pk_root is None else pk_root.var_index.astype(np.int32), batch, (nsimple,), torch.int32, device, src)
        vi = np.ascontiguousarray(vi.cpu().numpy().astype(np.uint32))
    return PackedBatch(nvar, dims, types, np.ascontiguousarray(data.cpu().numpy()), vi)


def gather_lsi_results(res_local: dict, batch: int, device=None, dst: int = 0):
    """inverse of scatter_lsi_batch for the results of LsiBatch.run: rank `dst` returns dict(x, info, active, v) over all instances"""
    import torch
    info = np.array([[i[k] for k in ("status", "iterations", "activations", "deactivations", "factorizations", "total_rank")] for i in res_local["info"]],
                    np.int32).reshape(-1, 6)
    out = {}
    for name, arr, dt in (("x", res_local["x"], torch.float64), ("info", info, torch.int32), ("active", res_local["active"], torch.uint8),
                          ("v", res_local["v"], torch.float64)):
        t = torch.as_tensor(np.ascontiguousarray(arr), dtype=dt)
        if device is not None:
            t = t.to(device)
        g = _gather_rows(t, batch, dst)
        out[name] = None if g is None else g.cpu().numpy()
    import torch.distributed as dist
    return out if dist.get_rank() == dst else None
