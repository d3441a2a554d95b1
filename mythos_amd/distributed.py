"""Replica sharding over the GPUs of one node: one process per GPU (torchrun), replica r runs on
rank r mod world, observables are combined with RCCL collectives over xGMI
(``torch.distributed`` backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).

Replaces the reference's Ray task fan-out / object-store gather
(mythos/optimization/optimization.py:151-169, 225-247).  The MD data path itself has no
collective: replicas are independent.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise the default process group from the torchrun environment -> (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_replicas(n_replicas: int, rank: int, world: int) -> list[int]:
    """Replica ids owned by ``rank``: r with r mod world == rank (balanced to within one)."""
    return list(range(rank, n_replicas, world))


def all_gather_observables(local: torch.Tensor, group=None) -> torch.Tensor:
    """(R_local, ...) per rank -> (R_total, ...) ordered by replica id (ranks may own different
    counts: rows are padded to the maximum and trimmed after the gather)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    total = sum(counts)
    out = torch.empty((total, *local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):  # replica id = rank + k * world
        ids = torch.arange(r, total, world, device=local.device)[: counts[r]]
        out[ids] = bufs[r][: counts[r]]
    del rank
    return out


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
