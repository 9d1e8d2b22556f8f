#!/usr/bin/env python
"""Kernel times of the fine-stage point sampler (HIP events): random vs grid-ordered points.  Run on the GPU box."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, rasterizer, synthetic
from lara_amd.fine import sample_point_feats
dev = "cuda:0"
g = torch.Generator().manual_seed(3)
V, h, w = 4, 512, 512
w2c = torch.linalg.inv(cameras.turntable_c2w(V).double()).float().to(dev)
focal = 0.5 * w / math.tan(0.5 * 0.75)
ixt = torch.tensor([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1.0]], dtype=torch.float32).expand(V, 3, 3).contiguous().to(dev)
img_ref = torch.rand(V, 3, h, w, generator=g).to(dev)
maps = [torch.rand(V, h, w, 3, generator=g).to(dev), torch.rand(V, h, w, generator=g).to(dev), (1.5 + torch.rand(V, h, w, 1, generator=g)).to(dev)]
grid_pts = synthetic.make_scene(grid=64, K=2, seed=0, device=dev)["centers"][::2].contiguous()      # LaRa's order: a jittered 64^3 grid
for name, pts in (("random", ((torch.rand(262144, 3, generator=g) * 2 - 1) * 0.5).to(dev)), ("grid-ordered", grid_pts)):
    n = pts.shape[0]
    gout = torch.randn(V, 8, n, generator=g).to(dev)
    for rep in range(2):
        if rep == 1: rasterizer.profile_enable(True)
        p = pts.clone().requires_grad_(True); m = [t.clone().requires_grad_(True) for t in maps]
        sample_point_feats(p, w2c, ixt, img_ref, *m).backward(gout)
        torch.cuda.synchronize()
    rec = rasterizer.profile_collect(); rasterizer.profile_enable(False)
    print(name, n, {k: round(ms * 1e3, 1) for k, ms in rec})
