"""Generates tests/golden/finedec_ref.npz by running the REFERENCE's own `Decoder.forward_fine`
(/root/reference/lightning/network.py:280-284; modules :234-240) on CPU, fp32, including its autograd gradients w.r.t.
both inputs and every parameter it touches.  Run in the build container only:
    python tests/golden/make_finedec_fixture.py"""
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

SEED, N = 11, 700
torch.manual_seed(SEED)
# Decoder(in_dim=80, sh_dim=12, scaling_dim=2, rotation_dim=4, opacity_dim=1, K=2): network.py:322-330 at configs/base.yaml
dec = net.Decoder(80, 12, 2, 4, 1, K=2)
with torch.no_grad():   # the reference zero-initialises the biases and LayerNorm is (1, 0): make every parameter matter
    for p in (dec.norm.weight, dec.norm.bias, dec.mlp_fine[0].bias, dec.mlp_fine[2].bias):
        p.add_(torch.randn_like(p) * 0.2)
g = torch.Generator().manual_seed(SEED + 1)
vol = (torch.randn(N, 80, generator=g) * 1.5 + 0.3).requires_grad_(True)
# the sampler's output layout [4,8,n] passed through the reference's einsum (network.py:515)
pf_planar = (torch.randn(4, 8, N, generator=g) * 1.2).requires_grad_(True)
pf = torch.einsum('lcb->blc', pf_planar)
sh = dec.forward_fine(vol, pf)
gout = torch.randn(sh.shape, generator=g)
(sh * gout).sum().backward()
names = ["norm.weight", "norm.bias", "cross_att.q_proj_weight", "cross_att.k_proj_weight", "cross_att.v_proj_weight",
         "cross_att.out_proj.weight", "mlp_fine.0.weight", "mlp_fine.0.bias", "mlp_fine.2.weight", "mlp_fine.2.bias"]
params = dict(dec.named_parameters())
out = {"vol": vol.detach().numpy(), "pf_planar": pf_planar.detach().numpy(), "sh": sh.detach().numpy(), "gout": gout.numpy(),
       "d_vol": vol.grad.numpy(), "d_pf_planar": pf_planar.grad.numpy()}
for k in names:
    out["p." + k] = params[k].detach().numpy()
    out["g." + k] = params[k].grad.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "finedec_ref.npz"), **out)
print("wrote sh", tuple(sh.shape), "|sh| mean", float(sh.abs().mean()))
