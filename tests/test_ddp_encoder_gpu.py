"""The trainable volume transformer behind torch's own DistributedDataParallel, two ranks (RCCL wants one GPU per
rank; the test box has one, so both ranks share it and talk over gloo -- DDP's hooks, buckets and averaging are the
same code either way): after backward both ranks hold the same gradients, equal to the mean of the two ranks' local
gradients, i.e. the HIP backward feeds DDP like any torch module (train_lightning.py:72)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _model(dev="cuda:0"):
    from lara_amd.encoder_train import VolTransformer
    torch.manual_seed(3)
    return VolTransformer(embed_dim=256, image_feat_dim=800, n_groups=[2], vol_low_res=4, vol_high_res=8, out_dim=80,
                          num_layers=2, num_heads=16).to(dev)


def _inputs(rank, dev="cuda:0"):
    g = torch.Generator().manual_seed(100 + rank)
    return (torch.randn(1, 4, 800, 2, 2, 2, generator=g).to(dev), torch.randn(1, 8, 8, 8, 80, generator=g).to(dev))


def _worker(rank, world, port, out, backend="gloo"):
    """`backend="gloo"`: both ranks share cuda:0; `backend="nccl"` (= RCCL): rank r owns cuda:r."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        from lara_amd import encoder_train
        # small buckets, so that the two blocks' parameters land in different ones
        ddp = torch.nn.parallel.DistributedDataParallel(_model(dev), device_ids=[dev.index] if backend == "nccl" else None,
                                                        find_unused_parameters=True, bucket_cap_mb=1)
        log = encoder_train._block_bwd_log = []

        def hook(state, bucket):      # called by the reducer the moment a bucket is full: this is where its all-reduce is enqueued
            log.append(("allreduce_enqueued", bucket.index()))
            return default_hooks.allreduce_hook(state, bucket)
        ddp.register_comm_hook(None, hook)
        feats, dout = _inputs(rank, dev)
        (ddp(feats) * dout).sum().backward()
        torch.cuda.synchronize()
        out.put((rank, {n: p.grad.cpu().numpy() for n, p in ddp.module.named_parameters()}, list(log)))   # (by value: this process exits)
    finally:
        dist.destroy_process_group()


def test_hip_backward_feeds_torch_ddp(hip_lib):
    _two_rank_ddp("gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two RCCL ranks want two GPUs (the round's box has one; the driver's 8-GPU node runs this)")
def test_hip_backward_feeds_torch_ddp_over_two_rccl_ranks(hip_lib):
    """The same assertions with backend nccl (= RCCL over xGMI), one rank per GPU: identical post-all-reduce gradients on
    both ranks, equal to the mean of the local ones, and the first bucket's all-reduce enqueued before the last block's
    backward.  Skipped on a one-GPU box: no two-rank RCCL run has happened in this environment."""
    _two_rank_ddp("nccl")


def _two_rank_ddp(backend):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (11 if backend == "nccl" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=600) for _ in range(2)]
    logs = {r: log for r, _, log in res}
    got = {r: {n: torch.from_numpy(a) for n, a in d.items()} for r, d, _ in res}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    local = []
    for rank in range(2):
        m = _model()
        feats, dout = _inputs(rank)
        (m(feats) * dout).sum().backward()
        local.append({n: p.grad.cpu() for n, p in m.named_parameters()})
    for n in got[0]:
        assert torch.equal(got[0][n], got[1][n]), n
        mean = (local[0][n] + local[1][n]) / 2
        assert float((got[0][n] - mean).abs().max()) <= 1e-6 * float(mean.abs().max()) + 1e-9, n
    # Overlap (train_lightning.py:68-81: DDP all-reduces bucket by bucket WHILE the backward runs): the encoder's backward
    # is one autograd node per block, so the first bucket's all-reduce is enqueued before the last block's backward
    # (layer 0) is even launched.  (As one node for the whole transformer -- round 2 -- every bucket came after it.)
    for rank in range(2):
        log = logs[rank]
        first_allreduce = next(i for i, e in enumerate(log) if e[0] == "allreduce_enqueued")
        last_block = next(i for i, e in enumerate(log) if e == ("block_backward_launch", 0))
        assert [e[1] for e in log if e[0] == "block_backward_launch"] == [1, 0]
        assert first_allreduce < last_block, log
        assert sum(e[0] == "allreduce_enqueued" for e in log) >= 3


def _rccl_worker(port, out):
    """One rank, backend nccl (= RCCL on ROCm): the communicator initialises and DDP's bucketed all-reduce runs through it
    around the whole pipeline step.  (One GPU here, so the ring has one member; what this pins is that the RCCL path --
    init, streams, the reducer's hooks on the per-block encoder nodes and the multi-stream raster -- works end to end.)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        from lara_amd.pipeline import lara_loss
        from tests.test_pipeline import _small_problem
        t = torch.ones(1024, device="cuda:0")
        dist.all_reduce(t)
        pipe, batch, feat_vol = _small_problem(torch.device("cuda:0"))
        pipe.fine_mask = "plain"

        def grads(model):
            for p in pipe.parameters():
                p.grad = None
            loss, _ = lara_loss(batch, model(batch, feat_vol, with_fine=True), 2000, ms_ssim=False)
            loss.backward()
            pipe.join_streams()
            torch.cuda.synchronize()
            return {n: p.grad.detach().cpu().numpy() for n, p in pipe.named_parameters()}
        plain = grads(pipe)
        ddp = torch.nn.parallel.DistributedDataParallel(pipe, device_ids=[0], find_unused_parameters=True, bucket_cap_mb=1)
        reduced = grads(ddp)
        out.put((float(t.sum()), plain, reduced))
    finally:
        dist.destroy_process_group()


def test_rccl_initialises_and_ddp_reduces_through_it(hip_lib):
    import numpy as np
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(29500 + (os.getpid() % 2000) + 7, out))
    p.start()
    total, plain, reduced = out.get(timeout=900)
    p.join(timeout=120)
    assert p.exitcode == 0 and total == 1024.0
    for n in plain:      # world size 1: the mean over ranks is the rank's own gradient (bf16 backward: same kernels, same bits
        a, b = plain[n], reduced[n]     # up to the order autograd accumulates the views' terms in)
        assert np.isfinite(b).all() and np.abs(a - b).max() <= 1e-2 * np.abs(a).max() + 1e-12, n
