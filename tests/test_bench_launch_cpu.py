"""`python bench.py --gpus 2` starts its own two ranks (no torchrun around it), wraps the encoder stand-in in torch's
DistributedDataParallel, times with the barrier / max-over-ranks protocol and prints ONE JSON line from rank 0 that
says n_gpus = the ranks that actually joined and carries the all-reduce's real bucket sizes.

LARA_BENCH_PLUMBING=1 replaces the HIP work by a few CPU linear layers (there is no GPU in this container; RCCL needs
GPUs, so the process group is gloo): this is a test of the launcher and of the DDP / timing / reporting plumbing, not
a measurement.  The same code path with the real HIP step runs in tests/test_bench_launch_gpu.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra):
    env = dict(os.environ, LARA_BENCH_PLUMBING="1", OMP_NUM_THREADS="2", **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]        # rank 0 only
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks_and_reports_real_buckets():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {})
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["unit"] == "frames/s"
    assert out["config"]["parallelism"].startswith("dp2")
    assert out["config"]["frames_per_step"] == 2 * 4 * 8 * 2          # 2 ranks x 4 scenes x (8 coarse + 8 fine)
    ar = out["config"]["grad_allreduce"]
    n_par = 64 * 256 + 256 + 256 * 64 + 64
    assert ar["bytes_per_step"] == 4 * n_par and ar["backend"] in ("nccl", "gloo")
    assert ar["buckets"] and sum(ar["buckets"]) == 4 * n_par           # the reducer's own record
    assert "PLUMBING" in out["data"]
    assert out["value"] > 0 and out["ms_per_step"] > 0


def test_bench_single_rank_plumbing_has_no_collective():
    out = _run(["--steps", "2", "--warmup", "1"], {})
    assert out["n_gpus"] == 1 and out["config"]["grad_allreduce"] is None
    assert out["config"]["frames_per_step"] == 4 * 8 * 2


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, LARA_BENCH_PLUMBING="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_reference_optimizer_groups_follow_system_configure_optimizers():
    """`bench.reference_optimizer` builds what lightning/system.py:78-106 builds: every LayerNorm parameter and every bias in the
    group without weight decay, every other parameter (once) in the decayed one; AdamW with configs/base.yaml's betas."""
    import torch
    import bench
    m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.LayerNorm(8), torch.nn.Conv3d(2, 2, 3), torch.nn.Linear(8, 4, bias=False))
    m[0].weight.requires_grad_(True)
    opt = bench.reference_optimizer(m, 4e-4)
    decay, no_decay = opt.param_groups
    assert decay["weight_decay"] == 0.05 and no_decay["weight_decay"] == 0.0
    ids = lambda g: {id(p) for p in g["params"]}
    assert ids(decay) == {id(m[0].weight), id(m[2].weight), id(m[3].weight)}
    assert ids(no_decay) == {id(m[0].bias), id(m[1].weight), id(m[1].bias), id(m[2].bias)}
    assert len(no_decay["params"]) == 4 and opt.defaults["betas"] == (0.9, 0.95) and decay["lr"] == 4e-4
    m[0](torch.randn(3, 8)).sum().backward()
    opt.step()          # (the CPU fallback of the fused update)
