#!/usr/bin/env python
"""HIP-event times of the fused render post-processing kernels at 512 x 512.  Run on the GPU box."""
import sys, math, torch, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lara_amd import rasterizer
from lara_amd.renderer import surface_maps
dev="cuda:0"; H=W=512
g=torch.Generator().manual_seed(0)
allmap=torch.rand(7,H,W,generator=g).to(dev).requires_grad_(True); color=torch.rand(3,H,W,generator=g).to(dev).requires_grad_(True)
rays=torch.cat([torch.zeros(H,W,3),F.normalize(torch.randn(H,W,3,generator=g),dim=-1)],-1).to(dev); rot=torch.eye(3,device=dev)
for rep in range(3):
    if rep==1: rasterizer.profile_enable(True)
    outs=surface_maps(color,allmap,rays,rot,0.0); sum(o.sum() for o in outs).backward(); torch.cuda.synchronize()
rec=rasterizer.profile_collect(); print({k:round(ms*1e3,1) for k,ms in rec})
