"""Generates tests/golden/pointfeat_ref.npz by running the REFERENCE's own `Network.get_point_feats`
(/root/reference/lightning/network.py:390-411, with `projection` :182-187) on CPU, fp32, including its autograd
gradients w.r.t. the points and the coarse renderings.  Run in the build container only:
    python tests/golden/make_pointfeat_fixture.py
The method reads only `self.device` from its module, so it is called unbound on a stand-in object."""
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
        return next(self.parameters()).device


pl.LightningModule = _LM
sys.modules["pytorch_lightning"] = pl
sys.modules["timm"] = types.ModuleType("timm")
tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tvt.Normalize = lambda *a, **k: None
tv.transforms = tvt
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tvt
sys.path.insert(0, "/root/reference")
import lightning.network as net  # noqa: E402

from lara_amd import cameras  # noqa: E402  (camera poses only)

SEED, V, h, w, N = 7, 4, 36, 52, 600
g = torch.Generator().manual_seed(SEED)
c2w = cameras.turntable_c2w(V)
w2c = torch.linalg.inv(c2w.double()).float()
focal = 0.5 * w / np.tan(0.5 * 0.75)
ixt = torch.tensor([[focal, 0, w / 2], [0, focal * 0.9, h / 2], [0, 0, 1]], dtype=torch.float32).expand(V, 3, 3).contiguous()
batch = {"tar_ixt": ixt[None], "tar_w2c": w2c[None]}
points_all = (torch.rand(N, 3, generator=g) * 2 - 1) * 0.75        # some project outside the images
mask = torch.rand(N, generator=g) > 0.3
img_ref = torch.rand(V, 3, h, w, generator=g)
ren = {"image": torch.rand(V, h, w, 3, generator=g), "acc_map": torch.rand(V, h, w, generator=g),
       "depth": 1.5 + torch.rand(V, h, w, 1, generator=g)}
pts = points_all.clone().requires_grad_(True)
rr = {k: v.clone().requires_grad_(True) for k, v in ren.items()}
fake_self = types.SimpleNamespace(device="cpu")
feats, m2 = net.Network.get_point_feats(fake_self, 0, img_ref, rr, V, batch, pts, mask)
gout = torch.randn(feats.shape, generator=torch.Generator().manual_seed(SEED + 1))
(feats * gout).sum().backward()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pointfeat_ref.npz"), points=points_all.numpy(), mask=mask.numpy(),
                    w2c=w2c.numpy(), ixt=ixt.numpy(), img_ref=img_ref.numpy(), image=ren["image"].numpy(), acc_map=ren["acc_map"].numpy(),
                    depth=ren["depth"].numpy(), feats=feats.detach().numpy(), gout=gout.numpy(), d_points=pts.grad.numpy(),
                    d_image=rr["image"].grad.numpy(), d_acc_map=rr["acc_map"].grad.numpy(), d_depth=rr["depth"].grad.numpy())
print("wrote", tuple(feats.shape), "points inside some image:", int((feats[:, 6].abs().sum(0) > 0).sum()), "of", int(mask.sum()))
