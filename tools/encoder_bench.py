#!/usr/bin/env python
"""VolTransformer (forward) kernel timings and MFMA throughput at LaRa's size: 12 GroupAttBlocks on a
32^3 x 256 volume + the x2 deconvolution, B scenes.  Run on the GPU box."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import rasterizer
from lara_amd.encoder import VolTransformer

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=4); ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--layers", type=int, default=12)
a = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
vt = VolTransformer(256, 800, [16], 32, 64, 80, a.layers, 16).to(dev)
with torch.no_grad():  # random-init weights of the reference's shapes (no checkpoint in this environment)
    for n, p in list(vt.named_parameters()) + list(vt.named_buffers()):
        if p.dtype == torch.bfloat16: p.copy_((torch.randn(p.shape, device=dev) * (p.shape[-1] ** -0.5)).to(torch.bfloat16))
        elif n.endswith("_w"): p.fill_(1.0)
    vt.pos_embed.normal_(0, 1 / 16)
B, M = a.scenes, a.scenes * 32 ** 3
feats = torch.randn(B, 4, 800, 16, 16, 16, device=dev)
with torch.no_grad():
    vt(feats, use_graph=False); torch.cuda.synchronize()
    rasterizer.profile_enable(True)   # per-kernel HIP events need plain launches
    for _ in range(a.reps): vt(feats, use_graph=False)
    torch.cuda.synchronize()
    rec = rasterizer.profile_collect(); rasterizer.profile_enable(False)
    walls = {}
    for mode in (False, True):        # wall time per forward: plain launches vs one HIP graph replay
        vt(feats, use_graph=mode); torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(a.reps): vt(feats, use_graph=mode)
        t1.record(); torch.cuda.synchronize()
        walls[mode] = t0.elapsed_time(t1) / a.reps
agg = {}
for k, ms in rec:
    t = agg.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += ms
G = M // 8
fl = {"ga_gemm_q": 2 * M * 256 * 256, "ga_gemm_kv": 2 * G * 4 * 800 * 512, "ga_gemm_o": 2 * M * 256 * 256,
      "ga_attn": 2 * 2 * (8 * 4 * 16) * 16 * G, "gb_mlp1": 2 * M * 256 * 512, "gb_mlp2": 2 * M * 512 * 256,
      "gb_conv3d": 2 * M * 27 * 256 * 256, "vt_deconv": 2 * M * 256 * 640}
tot_t = tot_f = 0.0
for k, (n, t) in agg.items():
    us = 1e3 * t / n; per_fwd = n / a.reps
    tot_t += us * per_fwd; tot_f += fl.get(k, 0) * per_fwd
    print(f"{k:12s} {us:9.1f} us x{per_fwd:4.0f}/fwd  {fl.get(k, 0) / us / 1e6:8.1f} TFLOP/s")
print(f"VolTransformer forward, {B} scenes, {a.layers} layers: kernels {tot_t / 1e3:.2f} ms, wall {walls[False]:.2f} ms "
      f"(launches) / {walls[True]:.2f} ms (HIP graph), "
      f"{tot_f / tot_t / 1e6:.1f} TFLOP/s over kernel time ({tot_f / tot_t / 1e6 / 2500 * 100:.1f} % of the 2.5 PF dense bf16 MFMA peak)")
