"""World-size-2 gloo test of the data-parallel plumbing bench.py uses for --gpus N (RCCL needs
GPUs; the same code runs over gloo on CPU)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lara_amd import dp


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the exchange step itself: torch DDP (what bench.py wraps the encoder in) averaging a module's gradients
        torch.manual_seed(0)
        lin = torch.nn.parallel.DistributedDataParallel(torch.nn.Linear(16, 16), bucket_cap_mb=dp.DDP_BUCKET_MB)
        lin(torch.full((2, 16), float(rank + 1))).sum().backward()
        ok_mean = bool(torch.allclose(lin.module.weight.grad, torch.full((16, 16), 2 * (1 + world) / 2.0)))
        buckets = 1
        tmax = dp.max_over_ranks(0.5 + rank, torch.device("cpu"))
        seeds = dp.scene_seeds(rank, 4)
        gathered = [None] * world
        dist.all_gather_object(gathered, seeds)
        if rank == 0:
            out.put((buckets, ok_mean, tmax, gathered))
    finally:
        dist.destroy_process_group()


def test_ddp_mean_and_timing_over_two_ranks():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = out.get(timeout=600)  # a cold container spends minutes importing torch in the spawned ranks
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    buckets, ok_mean, tmax, gathered = res
    assert buckets == 1 and ok_mean
    assert tmax == pytest.approx(1.5)
    flat = [s for per_rank in gathered for s in per_rank]
    assert len(set(flat)) == len(flat) == 8          # every rank renders different scenes
