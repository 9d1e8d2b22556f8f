"""`python bench.py --gpus 2` on the GPU box: the launcher starts two ranks itself; each runs the REAL step -- the
whole pipeline step (trainable HIP VolTransformer -> decoder -> coarse views -> sampler -> forward_fine -> fine views -> loss
-> backward) under torch's DistributedDataParallel (bucketed all-reduce of the encoder + decoder gradients).  The box has one GPU and RCCL wants one device per rank,
so both ranks share cuda:0 and the process group is gloo (LARA_BENCH_BACKEND=gloo): the number means nothing, the
code path is the one `--gpus N` takes with nccl on an N-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_real_step_over_gloo(hip_lib):
    env = dict(os.environ, LARA_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LARA_BENCH_PLUMBING"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scenes", "1",
           "--views", "4", "--grid", "16", "--res", "128", "--encoder-layers", "2", "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["data"] == "synthetic"
    assert out["config"]["frames_per_step"] == 2 * 1 * 4 * 2           # ranks x scenes x views x (coarse + fine)
    assert out["config"]["step"] == "pipeline"
    ar = out["config"]["grad_allreduce"]
    assert ar["backend"] == "gloo" and ar["buckets"] and sum(ar["buckets"]) == ar["bytes_per_step"]
    assert ar["bytes_per_step"] > 4 * 5_000_000                        # two GroupAttBlocks + pos_embed + tail + decoder, fp32
    assert ar["gradient_as_bucket_view"] is True and ar["every_n_steps"] == 1
    assert out["value"] > 0


def test_bench_gpus_2_with_the_references_accumulation_cadence(hip_lib):
    """`--accumulate 2` = train_lightning.py:73: the loss halved, DDP's no_sync() on the first micro-batch of a pair (no
    all-reduce, no update), all-reduce + clip + AdamW on the second.  Two ranks over gloo; the line says what ran."""
    env = dict(os.environ, LARA_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LARA_BENCH_PLUMBING"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--scenes", "1",
           "--views", "4", "--grid", "16", "--res", "128", "--encoder-layers", "2", "--accumulate", "2", "--no-cpu-baseline",
           "--no-roofline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["steps"] == 4
    assert out["config"]["grad_allreduce"]["every_n_steps"] == 2
    assert out["config"]["optimizer"]["accumulate_grad_batches"] == 2
    assert out["value"] > 0


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="two RCCL ranks want two GPUs (the round's box has one)")
def test_bench_gpus_2_real_step_over_rccl(hip_lib):
    """`python bench.py --gpus 2` as the driver launches it on a multi-GPU node: backend nccl (= RCCL), one rank per device."""
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LARA_BENCH_PLUMBING", "LARA_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scenes", "1",
           "--views", "4", "--grid", "16", "--res", "128", "--encoder-layers", "2", "--no-cpu-baseline", "--no-roofline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["grad_allreduce"]["backend"] == "nccl"
    assert out["config"]["frames_per_step"] == 2 * 1 * 4 * 2 and out["value"] > 0
