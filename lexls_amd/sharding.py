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
