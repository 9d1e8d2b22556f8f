"""Per-scene data parallelism around the raster path.

LaRa trains with single-node DDP (train_lightning.py:68-81): each rank renders its own scenes; the
only exchange step is the bucketed gradient all-reduce of the encoder parameters, which torch's own
DistributedDataParallel issues (bench.py wraps the trainable VolTransformer in it).  The raster is
per view and is never sharded.  These helpers are what ``bench.py --gpus N`` uses; they are
backend-agnostic (``nccl`` = RCCL over xGMI on MI355X, ``gloo`` in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

DDP_BUCKET_MB = 25  # torch DDP's default bucket size, what Lightning's DDPStrategy uses (train_lightning.py:68-81)
# find_unused_parameters: train_lightning.py:72.  gradient_as_bucket_view: a parameter's .grad IS its slice of the bucket, so
# autograd writes the HIP backward's gradients straight into the buffer RCCL reduces -- no copy into the bucket before the
# all-reduce, none back after it (158 MB each way per step at LaRa's 39.5 M parameters).
DDP_KW = dict(find_unused_parameters=True, bucket_cap_mb=DDP_BUCKET_MB, gradient_as_bucket_view=True)


def scene_seeds(rank: int, scenes_per_rank: int) -> list:
    """Distinct synthetic scenes per rank (weak scaling: per-GPU work is fixed)."""
    return [1000 * rank + i for i in range(scenes_per_rank)]


def max_over_ranks(seconds: float, device, group=None) -> float:
    """The step time the driver scores is the slowest rank's."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
