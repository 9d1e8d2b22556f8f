#!/usr/bin/env python
"""Trainable VolTransformer (lara_amd.encoder_train) at LaRa's size: forward and forward+backward wall time,
and the backward's kernel groups (HIP events).  Run on the GPU box."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import rasterizer
from lara_amd.encoder_train import VolTransformer

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=4); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--layers", type=int, default=12)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
vt = VolTransformer(256, 800, [16], 32, 64, 80, a.layers, 16).to(dev)   # the reference's own initialisation
B, M = a.scenes, a.scenes * 32 ** 3
feats = torch.randn(B, 4, 800, 16, 16, 16, device=dev, requires_grad=True)
dout = torch.randn(B, 64, 64, 64, 80, device=dev)


def step(backward):
    out = vt(feats)
    if backward:
        out.backward(dout)
        vt.zero_grad(set_to_none=True); feats.grad = None


def wall(backward):
    step(backward); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.reps): step(backward)
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / a.reps


w_f, w_fb = wall(False), wall(True)
rasterizer.profile_enable(True)
for _ in range(a.reps): step(True)
torch.cuda.synchronize()
rec = rasterizer.profile_collect(); rasterizer.profile_enable(False)
agg = {}
for k, ms in rec:
    t = agg.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += ms
G = M // 8
gemm = {"q": 2 * M * 256 * 256, "kv": 2 * G * 4 * 800 * 512, "o": 2 * M * 256 * 256, "mlp": 2 * 2 * M * 256 * 512,
        "conv": 2 * M * 27 * 256 * 256}
fl = {"gbb_dw_conv": gemm["conv"], "gbb_dx_conv": gemm["conv"], "gbb_dx_mlp": gemm["mlp"], "gbb_dw_linear": gemm["mlp"] + gemm["q"] + gemm["kv"] + gemm["o"],
      "gbb_dx_attn": gemm["q"] + gemm["kv"] + gemm["o"], 
      "gbb_recompute": gemm["q"] + gemm["kv"] + gemm["o"] + gemm["mlp"]}
tot = 0.0
for k, (n, t) in sorted(agg.items()):
    us = 1e3 * t / n; per = n / a.reps; tot += us * per
    print(f"{k:14s} {us:9.1f} us x{per:4.0f}/step  {fl.get(k, 0) / us / 1e6:8.1f} TFLOP/s")
fwd_fl = (sum(gemm.values()) * a.layers + 2 * M * 256 * 640)
print(f"trainable VolTransformer, {B} scenes, {a.layers} layers: forward {w_f:.2f} ms, forward+backward {w_fb:.2f} ms "
      f"(timed kernel groups {tot / 1e3:.2f} ms); {3 * fwd_fl / (w_fb * 1e-3) / 1e12:.0f} TFLOP/s of model work (3x forward) "
      f"over the step; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
