#!/usr/bin/env python
"""SHA-256 of the rasteriser's outputs and input gradients on the benchmark scene (two views, both synthetic regimes): two
builds of the library that print the same lines compute the same bits.  LARA2DGS_LIB selects the build.  Run on the GPU box."""
import hashlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, GaussianRasterizer, GaussianRasterizationSettings

dev = torch.device("cuda:0")
res = int(os.environ.get("RES", "512"))
cams = cameras.make_cameras(cameras.turntable_c2w(8), res, res, 0.75, 0.75, 1.106, 2.706, device=dev)
for regime in ("init", "trained"):
    sc = synthetic.make_scene(grid=int(os.environ.get("GRID", "64")), K=2, regime=regime, seed=0, device=dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    gc = (torch.randn(3, res, res, generator=g) / res ** 2).to(dev)
    ga = (torch.randn(7, res, res, generator=g) / res ** 2 * 0.1).to(dev)
    for ci in (0, 5):
        act = {k: v.detach().clone().requires_grad_(True) for k, v in synthetic.activate(sc).items()}
        cam = cams[ci]
        rs = GaussianRasterizationSettings(res, res, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0,
                                           cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(), 1,
                                           cam.camera_center, False, False)
        color, radii, allmap = GaussianRasterizer(rs)(means3D=act["means3D"], means2D=torch.zeros_like(act["means3D"]), shs=act["shs"],
                                                      opacities=act["opacities"], scales=act["scales"], rotations=act["rotations"])
        torch.autograd.backward([color, allmap], [gc, ga])
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in (color, allmap, radii, *[act[k].grad for k in ("means3D", "opacities", "scales", "rotations", "shs")]):
            h.update(t.detach().cpu().numpy().tobytes())
        print(f"{regime} view {ci}: {h.hexdigest()[:32]}  |grad means3D| max {float(act['means3D'].grad.abs().max()):.6e}")
