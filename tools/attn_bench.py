#!/usr/bin/env python
"""Group cross-attention (forward) kernel timings and MFMA throughput: one transformer layer's
attention step for B scenes (G = 4096 * B groups).  Run on the GPU box."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
from lara_amd import rasterizer
from lara_amd.attention import GroupCrossAttention

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=4); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
mod = GroupCrossAttention.from_modules(nn.LayerNorm(256), nn.MultiheadAttention(256, 16, kdim=800, vdim=800, bias=False, batch_first=True)).to(dev)
G = 4096 * a.scenes
x = torch.randn(G, 8, 256, device=dev); cond = torch.randn(G, 4, 800, device=dev)
lib = rasterizer.load_library()
ref = None
for mode in ["K|V projection + one fused kernel"]:
    with torch.no_grad():
        for _ in range(3): y = mod(x, cond)
        torch.cuda.synchronize()
        rasterizer.profile_enable(True)
        for _ in range(a.reps): mod(x, cond)
        torch.cuda.synchronize()
    rec = rasterizer.profile_collect(); rasterizer.profile_enable(False)
    if ref is None:
        ref = y.clone()
    agg = {}
    for k, ms in rec:
        t = agg.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += ms
    fl = {"ga_gemm_q": 2 * G * 8 * 256 * 256, "ga_gemm_kv": 2 * G * 4 * 800 * 512, "ga_gemm_o": 2 * G * 8 * 256 * 256,
          "ga_attn": 2 * 2 * (8 * 4 * 16) * 16 * G, "ga_ln_cast": 0,
          "ga_fused": 2 * 2 * G * 8 * 256 * 256 + 2 * 2 * (8 * 4 * 16) * 16 * G}
    by = {"ga_ln_cast": G * 8 * 256 * 6, "ga_attn": G * (8 * 256 * 2 * 2 + 4 * 512 * 2), "ga_fused": G * (8 * 256 * 4 * 2 + 4 * 512 * 2)}
    tot_t = tot_f = 0
    print(f"--- {mode}")
    for k, (n, t) in agg.items():
        us = 1e3 * t / n; tot_t += us; tot_f += fl.get(k, 0)
        extra = f" {by[k] / us / 1e3:7.1f} GB/s" if k in by else ""
        print(f"{k:12s} {us:8.1f} us  {fl.get(k, 0) / us / 1e6:8.1f} TFLOP/s{extra}")
    print(f"layer attention step, {a.scenes} scenes, mode {mode}: {tot_t:.1f} us, {tot_f / tot_t / 1e6:.1f} TFLOP/s overall "
          f"({tot_f / tot_t / 1e6 / 2500 * 100:.1f} % of the 2.5 PF dense bf16 MFMA peak)")
