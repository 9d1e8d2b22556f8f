"""Generates tests/golden/voltrans_ref.npz by running the REFERENCE's own VolTransformer
(/root/reference/lightning/network.py:105-164, built from its GroupAttBlock :57-102) on CPU, fp32.
Run in the build container only (the reference is not available on the GPU box):
    python tests/golden/make_voltrans_fixture.py

A reduced configuration keeps the fixture small (2 scenes, 4^3 -> 8^3 volume, 2 layers; channel
widths, head count, cond width and the x2 deconvolution are LaRa's).  The fixture stores the output for
seeded inputs plus checksums of the seeded weights, so the test rebuilds identical weights with plain
torch modules constructed in the reference's order."""
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
CFG = dict(embed_dim=256, image_feat_dim=800, n_groups=[2], vol_low_res=4, vol_high_res=8, out_dim=80,
           num_layers=2, num_heads=16)
B, V = 2, 4
torch.manual_seed(SEED)
vt = net.VolTransformer(**CFG)
g = torch.Generator().manual_seed(SEED + 1)
feats = torch.randn(B, V, CFG["image_feat_dim"], 2, 2, 2, generator=g)
with torch.no_grad():
    out = vt(feats)
    blk = vt.layers[0]
    x0 = vt.pos_embed.repeat(B, 1, 1, 1, 1)
    cond = feats.permute(0, 3, 4, 5, 1, 2).reshape(B * 8, V, CFG["image_feat_dim"])
    layer0 = blk(x0, cond, 2, 2)
wsum = [float(p.double().sum()) for p in (vt.pos_embed, blk.cross_attn.q_proj_weight, blk.cnn.weight, blk.mlp[0].weight,
                                            vt.layers[1].mlp[3].bias, vt.deconv.weight, vt.deconv.bias)]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "voltrans_ref.npz"), seed=SEED, B=B,
                    out=out.numpy(), layer0=layer0.numpy(), wsum=np.array(wsum),
                    eps_block=blk.norm1.eps, eps_final=vt.norm.eps)
print("wrote fixture", tuple(out.shape), float(out.abs().mean()), tuple(layer0.shape))
