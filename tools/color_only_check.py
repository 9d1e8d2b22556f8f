#!/usr/bin/env python
"""How far the colour-only backward (dL_dallmap = NULL) is from the full kernel fed seven planes of zeros: worst |difference| per
gradient tensor relative to max|gradient|, for the parity suite's small scenes and for one full-size view (P = 524 288, 512 x 512).
Run on the GPU box:  python tools/color_only_check.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, GaussianRasterizer, GaussianRasterizationSettings

dev = torch.device("cuda:0")


def grads(act, rs, dc, zeros):
    inp = {k: v.detach().clone().requires_grad_(True) for k, v in act.items()}
    m2 = torch.zeros_like(inp["means3D"], requires_grad=True)
    color, _, allmap = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=m2, shs=inp["shs"], opacities=inp["opacities"],
                                              scales=inp["scales"], rotations=inp["rotations"])
    loss = (color * dc).sum()
    if zeros:
        loss = loss + (allmap * torch.zeros_like(allmap)).sum()
    loss.backward()
    torch.cuda.synchronize()
    out = {k: v.grad for k, v in inp.items()}
    out["means2D"] = m2.grad
    return out


def one(name, grid, res, regime, seed, scale_boost=None, opacity_boost=0.0):
    sc = synthetic.make_scene(grid=grid, K=2, regime=regime, seed=seed, device=dev)
    if scale_boost is not None:
        sc["scales"] = sc["scales"] + math.log(scale_boost)
    sc["opacity"] = sc["opacity"] + opacity_boost
    act = synthetic.activate(sc)
    near, far = (1.106, 2.706) if res == 512 else (0.5, 2.5)
    cam = cameras.make_cameras(cameras.turntable_c2w(4), res, res, 0.75, 0.75, near, far, device=dev)[1]
    rs = GaussianRasterizationSettings(res, res, math.tan(0.375), math.tan(0.375), torch.tensor([0.2, 0.5, 1.0], device=dev), 1.0,
                                       cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(), 1,
                                       cam.camera_center, False, False)
    dc = torch.randn(3, res, res, generator=torch.Generator().manual_seed(2)).to(dev)
    a, b = grads(act, rs, dc, True), grads(act, rs, dc, False)
    row = {k: float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30)) for k in a}
    same = {k: float((a[k] == b[k]).float().mean()) for k in a}
    print(f"{name:34s} worst |full(zeros) - colour-only| / max|grad|: " + "  ".join(f"{k} {v:.1e}" for k, v in row.items()))
    print(f"{'':34s} share of entries identical:                  " + "  ".join(f"{k} {v:.4f}" for k, v in same.items()))


one("init 16^3 x 2 @128", 16, 128, "init", 0)
one("trained-like 16^3 x 2 @128", 16, 128, "trained", 3)
one("deep lists 24^3 x 2 @64", 24, 64, "init", 8, 3.0, -1.0)
one("sub-pixel splats (low-pass) @64", 12, 64, "init", 6, 0.05, 3.0)
one("full size 64^3 x 2 @512, init", 64, 512, "init", 0)
one("full size 64^3 x 2 @512, trained-like", 64, 512, "trained", 0)
