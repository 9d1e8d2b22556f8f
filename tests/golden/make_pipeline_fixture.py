"""Generates tests/golden/pipeline_ref.npz by running the REFERENCE's own glue between the operators of the LaRa step
(/root/reference/lightning/network.py) on CPU, fp32:
  * `Decoder.forward_coarse` (:259-278) + `Network.get_offseted_pt` (:425-429) + the opacity mask (:463-465),
  * `Network._check_mask` (:381-388) in its three regimes, train and eval, under fixed seeds,
  * `Losses.forward` (lightning/loss.py:17-60) with `pytorch_msssim.MS_SSIM` stubbed to return 1 (the package is not
    installed; the MS-SSIM term then contributes 0 and everything else is the reference's arithmetic).
Run in the build container only:  python tests/golden/make_pipeline_fixture.py"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pl = types.ModuleType("pytorch_lightning")


class _LM(nn.Module):
    @property
    def device(self):
        return torch.device("cpu")


pl.LightningModule = _LM
sys.modules["pytorch_lightning"] = pl
sys.modules["timm"] = types.ModuleType("timm")
tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tvt.Normalize = lambda *a, **k: None
tv.transforms = tvt
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tvt
ms = types.ModuleType("pytorch_msssim")


class _One(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, a, b):
        return torch.ones((), dtype=a.dtype)


ms.MS_SSIM = _One
sys.modules["pytorch_msssim"] = ms
sys.path.insert(0, "/root/reference")
import lightning.network as net  # noqa: E402
from lightning.loss import Losses  # noqa: E402

out = {}
torch.manual_seed(21)
K, R = 2, 4                                      # grid_reso 4 -> 8^3 voxels x K Gaussians
dec = net.Decoder(80, 12, 2, 4, 1, K=K)
with torch.no_grad():
    for layer in dec.mlp_coarse:
        if isinstance(layer, nn.Linear):
            layer.bias.add_(torch.randn_like(layer.bias) * 0.1)
vol = torch.randn(2, 2 * R, 2 * R, 2 * R, 80)
opacity_shift, voxel_size = -2.1792, 2.0 / (R * 2)
scaling_shift = float(np.log(0.5 * voxel_size / 3.0))
offset, sh, scaling, rotation, opacity = dec.forward_coarse(vol, opacity_shift, scaling_shift)
me = types.SimpleNamespace(device=torch.device("cpu"), scene_size=0.5, n_offset_groups=32)
me.group_centers = net.Network.build_dense_grid(me, R * 2).reshape(1, -1, 3)
centers = net.Network.get_offseted_pt(me, offset, K)
masks = torch.sigmoid(opacity).squeeze(-1) > 0.005
for k, p in dec.mlp_coarse.named_parameters():
    out["dec.mlp_coarse." + k] = p.detach().numpy()
out.update(vol=vol.numpy(), centers=centers.detach().numpy(), sh=sh.detach().numpy(), scaling=scaling.detach().numpy(),
           rotation=rotation.detach().numpy(), opacity=opacity.detach().numpy(), masks=masks.numpy(),
           opacity_shift=np.float64(opacity_shift), scaling_shift=np.float64(scaling_shift))

# _check_mask: sparse (< 0.1 %), dense (> 50 %), in between; training and eval
g = torch.Generator().manual_seed(5)
base = {"sparse": torch.zeros(4096, dtype=torch.bool), "dense": torch.rand(4096, generator=g) < 0.8,
        "middle": torch.rand(4096, generator=g) < 0.3}
base["sparse"][7] = True
for name, m in base.items():
    for training in (True, False):
        me.training = training
        torch.manual_seed(100)
        res = net.Network._check_mask(me, m.clone())
        out[f"mask.{name}.in"] = m.numpy()
        out[f"mask.{name}.{'train' if training else 'eval'}"] = res.numpy().astype(bool)

# the loss on a small output dictionary
g = torch.Generator().manual_seed(9)
B, V, H, W = 2, 3, 8, 8
batch = {"tar_rgb": torch.rand(B, V, H, W, 3, generator=g)}
outp = {}
for prex in ("", "_fine"):
    outp[f"image{prex}"] = torch.rand(B, H, V * W, 3, generator=g)
    outp[f"acc_map{prex}"] = torch.rand(B, H, V * W, generator=g)
    outp[f"rend_dist{prex}"] = torch.rand(B, H, V * W, generator=g) * 1e-3
    outp[f"rend_normal{prex}"] = torch.randn(B, H, V * W, 3, generator=g)
    outp[f"depth_normal{prex}"] = torch.randn(B, H, V * W, 3, generator=g)
L = Losses()
for it in (500, 2000):
    for with_fine in (True, False):
        o = {k: v.clone().requires_grad_(True) for k, v in outp.items() if with_fine or not k.endswith("_fine")}
        loss, _ = L(batch, o, it)
        loss.backward()
        tag = f"loss.{it}.{'fine' if with_fine else 'coarse'}"
        out[tag] = loss.detach().numpy()
        for k, v in o.items():
            out[f"{tag}.d_{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
for k, v in outp.items():
    out["loss.in." + k] = v.numpy()
out["loss.in.tar_rgb"] = batch["tar_rgb"].numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pipeline_ref.npz"), **out)
print("wrote", len(out), "arrays; masks kept", float(masks.float().mean()))
