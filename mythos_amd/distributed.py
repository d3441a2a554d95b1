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


def all_gather_observables(local: torch.Tensor, group=None, n_total: int | None = None, force: bool = False) -> torch.Tensor:
    """(R_local, ...) per rank -> (R_total, ...) ordered by replica id (replica r lives on rank r mod world).

    ``n_total`` given (the caller knows how many replicas the job has - always true for ``shard_replicas``): every rank's
    count follows from it, so the gather is ONE collective (``all_gather_into_tensor`` of rows padded to ceil(n_total /
    world)) and nothing crosses to the host.  Without it the counts are gathered first (a second collective and a host
    read-back), which is fine off the timed path.  ``force``: issue the collective in a group of one as well (the
    single-GPU test of the RCCL path, tests/test_gpu_difftre.py)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    if n_total is not None:
        counts = [len(range(r, n_total, world)) for r in range(world)]
        if local.shape[0] != counts[dist.get_rank(group)]:
            raise ValueError(f"rank {dist.get_rank(group)} holds {local.shape[0]} rows, {counts[dist.get_rank(group)]} expected of {n_total}")
    else:
        n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        gathered = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(gathered, n_local, group=group)
        counts = [int(c.item()) for c in gathered]
    mx, total = max(counts), sum(counts)
    if local.shape[0] == mx:
        pad = local.contiguous()
    else:
        pad = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    buf = torch.empty((world * mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    buf = buf.reshape(world, mx, *local.shape[1:])
    if total == world * mx:  # equal shares: replica k * world + r is row k of rank r - a transpose, no index lists
        return buf.transpose(0, 1).reshape(total, *local.shape[1:])
    out = torch.empty((total, *local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):  # replica id = rank + k * world
        out[r:total:world] = buf[r, : counts[r]]
    return out


def all_reduce_sum(t: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t
