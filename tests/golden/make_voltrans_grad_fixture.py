"""Generates tests/golden/voltrans_grad_ref.npz: gradients of the REFERENCE's own VolTransformer
(/root/reference/lightning/network.py:105-164) from torch autograd on CPU, fp32, for the seeded model and
inputs of make_voltrans_fixture.py and the loss  L = sum(out * dOut)  with a seeded dOut.
Run in the build container only:
    python tests/golden/make_voltrans_grad_fixture.py

Stored: the module's state_dict keys (the trainable drop-in must expose the same), d(image_feats) and
d(pos_embed) in full, every parameter gradient either in full (<= 20k elements) or as a strided sample
(every `stride`-th element of the flattened tensor) plus its sum and sum of squares."""
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
import lightning.network as net  # noqa: E402  (the reference's package)

SEED = 4321
STRIDE = 127
CFG = dict(embed_dim=256, image_feat_dim=800, n_groups=[2], vol_low_res=4, vol_high_res=8, out_dim=80,
           num_layers=2, num_heads=16)
B, V = 2, 4
torch.manual_seed(SEED)
vt = net.VolTransformer(**CFG)
g = torch.Generator().manual_seed(SEED + 1)
feats = torch.randn(B, V, CFG["image_feat_dim"], 2, 2, 2, generator=g).requires_grad_(True)
out = vt(feats)
dout = torch.randn(out.shape, generator=torch.Generator().manual_seed(SEED + 2))
(out * dout).sum().backward()

store = {"seed": SEED, "B": B, "stride": STRIDE, "keys": np.array(list(vt.state_dict().keys())),
         "d_feats": feats.grad.numpy()}
for name, p in vt.named_parameters():
    gflat = p.grad.reshape(-1)
    store["sum/" + name] = float(gflat.double().sum())
    store["sq/" + name] = float((gflat.double() ** 2).sum())
    store["g/" + name] = (gflat if gflat.numel() <= 20000 else gflat[::STRIDE]).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "voltrans_grad_ref.npz"), **store)
print("wrote", len(store), "entries;", sum(v.nbytes for v in store.values() if hasattr(v, "nbytes")) // 1024, "KB raw")
