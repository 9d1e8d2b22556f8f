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
        n = 3 * (dp.DDP_BUCKET_BYTES // 4) // 8 + 17          # several small buckets + a ragged tail
        flat = torch.full((n,), float(rank + 1))
        buckets = dp.bucketed_all_reduce(flat, bucket_bytes=dp.DDP_BUCKET_BYTES // 8)
        ok_mean = bool(torch.allclose(flat, torch.full((n,), (1 + world) / 2.0)))
        tmax = dp.max_over_ranks(0.5 + rank, torch.device("cpu"))
        seeds = dp.scene_seeds(rank, 4)
        gathered = [None] * world
        dist.all_gather_object(gathered, seeds)
        if rank == 0:
            out.put((buckets, ok_mean, tmax, gathered))
    finally:
        dist.destroy_process_group()


def test_bucketed_all_reduce_and_timing_over_two_ranks():
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
    assert buckets == 4 and ok_mean
    assert tmax == pytest.approx(1.5)
    flat = [s for per_rank in gathered for s in per_rank]
    assert len(set(flat)) == len(flat) == 8          # every rank renders different scenes
