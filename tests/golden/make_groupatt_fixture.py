"""Generates tests/golden/groupatt_ref.npz by running the REFERENCE's own GroupAttBlock
(/root/reference/lightning/network.py:57-102) on CPU.  Run in the build container only (the
reference is not available on the GPU box):  python tests/golden/make_groupatt_fixture.py

The reference module needs four absent packages only for unrelated code paths; they are stubbed.
The fixture stores the attention step's output for seeded inputs plus checksums of the seeded
weights, so the test can rebuild identical weights with plain torch modules."""
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

SEED, G = 1234, 32
torch.manual_seed(SEED)
block = net.GroupAttBlock(inner_dim=256, cond_dim=800, num_heads=16, eps=1e-6)
g = torch.Generator().manual_seed(SEED + 1)
x = torch.randn(G, 8, 256, generator=g)
cond = torch.randn(G, 4, 800, generator=g)
with torch.no_grad():
    out = x + block.cross_attn(block.norm1(x), cond, cond, need_weights=False)[0]
mha = block.cross_attn
np.savez_compressed(
    os.path.join(ROOT, "tests", "golden", "groupatt_ref.npz"),
    seed=SEED, G=G, out=out.numpy(),
    wsum=np.array([float(mha.q_proj_weight.double().sum()), float(mha.k_proj_weight.double().sum()),
                   float(mha.v_proj_weight.double().sum()), float(mha.out_proj.weight.double().sum())]),
    ln_eps=block.norm1.eps)
print("wrote fixture", out.shape, float(out.abs().mean()))
