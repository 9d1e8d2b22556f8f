"""Generates tests/golden/coarsedec_ref.npz by running the REFERENCE's own `Decoder.forward_coarse`
(/root/reference/lightning/network.py:259-278; modules :229-233) on CPU, twice: in fp32, and under
`torch.autocast("cpu", dtype=torch.bfloat16)` -- the arithmetic of the bf16-mixed precision the reference trains with
(train_lightning.py:74) -- each with its autograd gradients w.r.t. the input features and the six parameters.
Run in the build container only:  python tests/golden/make_coarsedec_fixture.py"""
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

SEED, B, V = 23, 2, 333          # 666 voxel rows: not a multiple of the kernel's 128-row trips
OPACITY_SHIFT, SCALING_SHIFT = -2.1792, -5.257
torch.manual_seed(SEED)
dec = net.Decoder(80, 12, 2, 4, 1, K=2)     # network.py:322-330 at configs/base.yaml
with torch.no_grad():   # the reference zero-initialises the biases: make them matter
    for i in (0, 2, 4):
        dec.mlp_coarse[i].bias.add_(torch.randn_like(dec.mlp_coarse[i].bias) * 0.3)
g = torch.Generator().manual_seed(SEED + 1)
feats = torch.randn(B, V, 80, generator=g) * 1.2 + 0.1
gouts = [torch.randn(B, V * 2, *s, generator=g) for s in ((3,), (4, 3), (2,), (4,), (1,))]
names = ["offset", "sh", "scaling", "rotation", "opacity"]
out = {"feats": feats.numpy(), "opacity_shift": np.array(OPACITY_SHIFT), "scaling_shift": np.array(SCALING_SHIFT)}
for n, go in zip(names, gouts):
    out["gout." + n] = go.numpy()
for k, p in dec.mlp_coarse.named_parameters():
    out["p." + k] = p.detach().numpy()
for tag, ctx in (("fp32", torch.autocast("cpu", enabled=False)), ("bf16", torch.autocast("cpu", dtype=torch.bfloat16))):
    x = feats.clone().requires_grad_(True)
    for p in dec.parameters():
        p.grad = None
    with ctx:
        res = dec.forward_coarse(x, OPACITY_SHIFT, SCALING_SHIFT)
    sum((r * go).sum() for r, go in zip(res, gouts)).backward()
    for n, r in zip(names, res):
        out[f"{tag}.{n}"] = r.detach().numpy()
    out[f"{tag}.d_feats"] = x.grad.numpy()
    for k, p in dec.mlp_coarse.named_parameters():
        out[f"{tag}.g.{k}"] = p.grad.numpy()
    print(tag, {n: tuple(r.shape) for n, r in zip(names, res)}, "|d_feats| max", float(x.grad.abs().max()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "coarsedec_ref.npz"), **out)
