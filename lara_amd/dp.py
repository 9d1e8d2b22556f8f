"""Per-scene data parallelism around the raster path.

LaRa trains with single-node DDP (train_lightning.py:68-81): each rank renders its own scenes; the
only exchange step is the bucketed gradient all-reduce of the encoder parameters.  The raster is
per view and is never sharded.  These helpers are what ``bench.py --gpus N`` uses; they are
backend-agnostic (``nccl`` = RCCL over xGMI on MI355X, ``gloo`` in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

DDP_BUCKET_BYTES = 25 * 1024 * 1024  # torch DDP's default bucket size, what Lightning's DDPStrategy uses


def scene_seeds(rank: int, scenes_per_rank: int) -> list:
    """Distinct synthetic scenes per rank (weak scaling: per-GPU work is fixed)."""
    return [1000 * rank + i for i in range(scenes_per_rank)]


def bucketed_all_reduce(flat: torch.Tensor, bucket_bytes: int = DDP_BUCKET_BYTES, average: bool = True,
                        group=None) -> int:
    """All-reduce a flat gradient buffer bucket by bucket (DDP-style); returns the bucket count."""
    n = max(1, bucket_bytes // flat.element_size())
    world = dist.get_world_size(group)
    count = 0
    for o in range(0, flat.numel(), n):
        chunk = flat[o:o + n]
        dist.all_reduce(chunk, group=group)
        if average:
            chunk.div_(world)
        count += 1
    return count


def max_over_ranks(seconds: float, device, group=None) -> float:
    """The step time the driver scores is the slowest rank's."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
