"""Multi-GPU layer of the path: clips shard by batch, one process per GPU, and the only
collective is an all-gather of per-clip logits (RCCL over xGMI when the process group's
backend is "nccl"; gloo on CPU for the tests).  No weights or activations ever cross GPUs."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_batch(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of `batch` clips over `world` ranks: rank r owns [lo, hi).  The batch
    must divide evenly so every rank contributes the same number of logit rows."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if batch % world:
        raise ValueError(f"batch {batch} does not divide over {world} ranks")
    per = batch // world
    return rank * per, (rank + 1) * per


def gather_logits(local_logits: torch.Tensor, group=None) -> torch.Tensor:
    """[B_local, C] on every rank -> [B_local * world, C] on every rank, rank order preserved."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_logits
    world = dist.get_world_size(group)
    local_logits = local_logits.contiguous()
    out = torch.empty((world * local_logits.shape[0],) + tuple(local_logits.shape[1:]),
                      dtype=local_logits.dtype, device=local_logits.device)
    dist.all_gather_into_tensor(out, local_logits, group=group)
    return out
