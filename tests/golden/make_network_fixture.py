"""Generates tests/golden/network_ref.npz by running the REFERENCE's unmodified `Network.forward`
(/root/reference/lightning/network.py:431-532, eval mode, with_fine=True) on CPU -- its own `VolTransformer`, `Decoder`,
`get_offseted_pt`, `MiniCam`, `Renderer.render_img` (renderer_2dgs.py:167-268, `depth_to_normal`), `get_point_feats` and
`forward_fine`, composed by its own loop -- with two stand-ins for what cannot run here:
  * `diff_surfel_rasterization`: the CUDA rasteriser is absent (empty submodule); a module with the same two names whose
    `GaussianRasterizer.forward` calls the CPU oracle (oracle/surfel_oracle.c) takes its place;
  * `DinoWrapper` (timm + pretrained weights, no network here): a seeded patch-embedding with the same output shape.
What is stored: the batch, the decoder's parameters, the volume features the encoder handed to the decoder
(`volume_feat_up`), and the output dictionary.  tests/test_pipeline.py feeds those volume features and parameters to
`lara_amd.pipeline.LaRaPipeline` on the GPU: everything from the coarse decoder to the concatenated fine renders is then
the HIP path against the reference's own composition (fp32 on both sides).
Run in the build container only:  python tests/golden/make_network_fixture.py"""
import math
import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from lara_amd.rasterizer import GaussianRasterizationSettings  # noqa: E402  (the 12-field record, renderer_2dgs.py:124-137)

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


class _OracleRasterizer(nn.Module):
    """`GaussianRasterizer(raster_settings)(means3D=..., ...)` -> (color, radii, allmap), computed by the CPU oracle."""

    def __init__(self, raster_settings):
        super().__init__()
        self.rs = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.rs
        f = lambda t: t.detach().cpu().float().numpy()
        view = oracle.View(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), f(rs.bg), float(rs.scale_modifier),
                           f(rs.viewmatrix), f(rs.projmatrix), int(rs.sh_degree), f(rs.campos))
        r = oracle.forward(view, f(means3D), f(opacities), shs=f(shs), scales=f(scales), rotations=f(rotations))
        return torch.from_numpy(r.color.copy()), torch.from_numpy(r.radii.copy()), torch.from_numpy(r.allmap.copy())


dsr = types.ModuleType("diff_surfel_rasterization")
dsr.GaussianRasterizationSettings = GaussianRasterizationSettings
dsr.GaussianRasterizer = _OracleRasterizer
sys.modules["diff_surfel_rasterization"] = dsr
sys.path.insert(0, "/root/reference")
import lightning.network as net  # noqa: E402
_pkg = types.ModuleType("dataLoader")          # bare package: only dataLoader.utils is executed (the __init__ imports h5py, PIL, ...)
_pkg.__path__ = ["/root/reference/dataLoader"]
sys.modules["dataLoader"] = _pkg
from dataLoader.utils import build_rays  # noqa: E402


class _PatchEncoder(nn.Module):        # stands in for DinoWrapper: [N,3,H,W] -> [N, (H/16)(W/16), 768]
    def __init__(self, model_name, is_train=False):
        super().__init__()
        self.model = types.SimpleNamespace(num_features=768)
        self.proj = nn.Conv2d(3, 768, kernel_size=16, stride=16)

    def forward(self, image):
        return self.proj(image).flatten(2).transpose(1, 2)


net.DinoWrapper = _PatchEncoder
oracle.build()
torch.manual_seed(17)
ns = types.SimpleNamespace
cfg = ns(n_views=4, train=ns(use_rand_views=False),
         model=ns(encoder_backbone="stub", n_groups=[2], n_offset_groups=4, K=2, sh_degree=1, num_layers=2, num_heads=16, view_embed_dim=32,
                  embedding_dim=256, vol_feat_reso=2, vol_embedding_reso=4, vol_embedding_out_dim=80))
model = net.Network(cfg).eval()
with torch.no_grad():   # spread opacities around the 0.005 threshold and give the fine decoder non-trivial norms / biases
    model.decoder.mlp_coarse[4].weight.mul_(6.0)
    for p in (model.decoder.norm.weight, model.decoder.norm.bias, model.decoder.mlp_fine[0].bias, model.decoder.mlp_fine[2].bias):
        p.add_(torch.randn_like(p) * 0.2)
model.opacity_shift = -5.0

B, V, H, W, fov, radius = 2, 6, 48, 48, 0.75, 1.906
g = torch.Generator().manual_seed(5)
c2w_all, w2c_all, ixt_all, rays_all, raysd_all, bg_all = [], [], [], [], [], []
for b in range(B):
    az = torch.rand(V, generator=g) * 2 * math.pi
    el = (torch.rand(V, generator=g) - 0.5) * 1.2
    pos = radius * torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], -1)
    fwd = torch.nn.functional.normalize(-pos, dim=-1)
    right = torch.nn.functional.normalize(torch.cross(fwd, torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd), dim=-1), dim=-1)
    down = torch.cross(fwd, right, dim=-1)
    c2w = torch.eye(4).repeat(V, 1, 1)
    c2w[:, :3, 0], c2w[:, :3, 1], c2w[:, :3, 2], c2w[:, :3, 3] = right, down, fwd, pos
    focal = 0.5 * W / math.tan(0.5 * fov)
    ixt = torch.tensor([[focal, 0, W / 2], [0, focal, H / 2], [0, 0, 1.0]]).repeat(V, 1, 1)
    c2w_all.append(c2w)
    w2c_all.append(torch.linalg.inv(c2w))
    ixt_all.append(ixt)
    rays_all.append(torch.from_numpy(build_rays(c2w.numpy(), ixt.numpy().copy(), H, W, 1.0)))
    raysd_all.append(torch.from_numpy(build_rays(c2w.numpy(), ixt.numpy().copy(), H, W, 1.0 / 16)))
    lv = torch.tensor([0.0, 0.5, 1.0])[torch.randint(0, 3, (V,), generator=g)]
    lv[:4] = 1.0
    bg_all.append(lv[:, None].expand(-1, 3).contiguous())
batch = {"tar_rgb": torch.rand(B, V, H, W, 3, generator=g), "tar_c2w": torch.stack(c2w_all), "tar_w2c": torch.stack(w2c_all),
         "tar_ixt": torch.stack(ixt_all), "tar_rays": torch.stack(rays_all), "tar_rays_down": torch.stack(raysd_all),
         "bg_color": torch.stack(bg_all), "near_far": torch.tensor([[radius - 0.8, radius + 0.8]] * B),
         "fovx": torch.full((B,), fov), "fovy": torch.full((B,), fov),
         "meta": {"tar_h": torch.full((B,), H), "tar_w": torch.full((B,), W)}}

seen = {}
model.vol_decoder.register_forward_hook(lambda m, i, o: seen.update(feat_vol=i[0].detach().clone(), volume_feat_up=o.detach().clone()))
with torch.no_grad():
    out = model(batch, with_fine=True)
kept = float((torch.sigmoid(model.decoder.forward_coarse(seen["volume_feat_up"], model.opacity_shift, model.scaling_shift)[4]) > 0.005).float().mean())
res = {"B": np.array(B), "V": np.array(V), "H": np.array(H), "W": np.array(W), "opacity_shift": np.array(model.opacity_shift),
       "kept_fraction": np.array(kept), "volume_feat_up": seen["volume_feat_up"].numpy()}
for k, v in batch.items():
    if k not in ("meta", "tar_rays_down"):      # (the down-sampled rays only feed the image-feature volume, upstream of what is compared)
        res["batch." + k] = v.numpy()
for k, v in model.decoder.state_dict().items():
    res["decoder." + k] = v.numpy()
for k, v in out.items():
    res["out." + k] = v.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "network_ref.npz"), **res)
print("wrote", len(res), "arrays; kept fraction", kept, "; image mean", float(out["image"].mean()), "acc mean", float(out["acc_map"].mean()),
      {k: tuple(v.shape) for k, v in out.items()})
