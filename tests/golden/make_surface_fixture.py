"""Generates tests/golden/surface_ref.npz by running the REFERENCE's own `Renderer.render_img`
(/root/reference/lightning/renderer_2dgs.py:167-268, with its `depth_to_normal` :78-89) on CPU, fp32, around a
stand-in rasteriser that returns seeded (color, radii, allmap) leaf tensors: everything after the rasteriser
call -- the code lara_surface_maps_forward / _backward replaces -- is the reference's, including its autograd.
Run in the build container only:
    python tests/golden/make_surface_fixture.py

Two cases (depth_ratio 0 and 0.3); the maps contain empty pixels (alpha = 0, where the reference's gradients are
NaN/inf: stored as they come), colours outside [0, 1] and exact 0 / 1 values."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)           # the shim package `diff_surfel_rasterization` (import only; never called here)
sys.path.insert(0, "/root/reference")
import lightning.renderer_2dgs as ref  # noqa: E402

H, W, SEED = 24, 40, 99
g = torch.Generator().manual_seed(SEED)
yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
alpha = (torch.rand(H, W, generator=g) * 0.9 + 0.05) * ((xx ** 2 + yy ** 2) < 0.8)       # empty corners
depth_true = 1.8 + 0.3 * xx - 0.2 * yy * xx + 0.05 * torch.rand(H, W, generator=g)
allmap = torch.zeros(7, H, W)
allmap[0] = depth_true * alpha
allmap[1] = alpha
allmap[2:5] = torch.randn(3, H, W, generator=g) * alpha
allmap[5] = (depth_true + 0.02 * torch.randn(H, W, generator=g)) * (alpha > 0)
allmap[6] = torch.rand(H, W, generator=g) * alpha
color = torch.rand(3, H, W, generator=g) * 1.4 - 0.2
color[0, 0, :5] = torch.tensor([0.0, 1.0, 0.5, -0.1, 1.1])
dirs = torch.nn.functional.normalize(torch.stack([xx * 0.4, yy * 0.3, torch.ones_like(xx)], -1), dim=-1)
rays = torch.cat([torch.tensor([0.1, -0.2, 0.3]).expand(H, W, 3), dirs], -1).contiguous()
q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
wvt = torch.eye(4)
wvt[:3, :3] = q
cam = types.SimpleNamespace(world_view_transform=wvt, FoVx=0.75, FoVy=0.75, image_height=H, image_width=W)
P = 5
params = dict(centers=torch.randn(P, 3, generator=g), shs=torch.randn(P, 4, 3, generator=g), opacity=torch.randn(P, 1, generator=g),
              scales=torch.randn(P, 2, generator=g), rotations=torch.randn(P, 4, generator=g))
store = {"H": H, "W": W, "color": color.numpy(), "allmap": allmap.numpy(), "rays": rays.numpy(), "rot": wvt[:3, :3].T.contiguous().numpy()}
keys = ("image", "depth", "acc_map", "rend_normal", "depth_normal", "rend_dist")
for case, ratio in enumerate((0.0, 0.3)):
    c, a = color.clone().requires_grad_(True), allmap.clone().requires_grad_(True)
    r = ref.Renderer(sh_degree=1, white_background=True)
    r.set_rasterizer = lambda cam_, device="cpu", c=c, a=a: (lambda **kw: (c, torch.ones(P, dtype=torch.int32), a))
    out = r.render_img(cam, rays, device="cpu", depth_ratio=ratio, **params)
    gg = torch.Generator().manual_seed(SEED + 1 + case)
    loss = 0
    for k in keys:
        go = torch.randn(out[k].shape, generator=gg)
        store[f"g{case}/{k}"] = go.numpy()
        store[f"out{case}/{k}"] = out[k].detach().numpy()
        loss = loss + (out[k] * go).sum()
    loss.backward()
    store[f"ratio{case}"] = ratio
    store[f"d_color{case}"], store[f"d_allmap{case}"] = c.grad.numpy(), a.grad.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "surface_ref.npz"), **store)
print("wrote", {k: store[k].shape for k in store if k.startswith("out0")}, "non-finite grads:",
      int((~np.isfinite(store["d_allmap0"])).sum()), "of", store["d_allmap0"].size)
