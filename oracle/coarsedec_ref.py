"""TEST INFRASTRUCTURE (oracle): CPU restatement of LaRa's coarse decoder `Decoder.forward_coarse`
(/root/reference/lightning/network.py:259-278, modules :229-233).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package.

* `forward_coarse(x, params, K, sh_dim, ...)`   the three Linear layers with the arithmetic of the bf16-mixed autocast the
      reference trains under (train_lightning.py:74: operands and each layer's result rounded to bf16, fp32 accumulation)
      or in plain fp32 (`bf16=False`), then the split and the three activations.  Differentiable (torch autograd; the bf16
      roundings are straight-through for x and the activations, exactly as autocast's backward treats them up to the
      rounding of the gradients themselves).  Pinned to the reference's own `Decoder.forward_coarse` -- run under
      `torch.autocast("cpu", dtype=torch.bfloat16)` and in fp32 -- by tests/golden/coarsedec_ref.npz
      (tests/golden/make_coarsedec_fixture.py).
"""
import torch


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g


def forward_coarse(x, w1, b1, w2, b2, w3, b3, K, sh_dim, opacity_shift, scaling_shift, bf16=True):
    """x [M,80] -> (offset [M*K,3], sh [M*K,sh_dim], scaling [M*K,2], rotation [M*K,4], opacity [M*K,1])."""
    rnd = _RoundBf16.apply if bf16 else (lambda t: t)
    h = rnd(x.float())
    for i, (w, b) in enumerate(((w1, b1), (w2, b2), (w3, b3))):                       # network.py:260
        h = rnd(h @ rnd(w.float()).t() + rnd(b.float()))
        if i < 2:
            h = torch.relu(h)
    par = h.view(h.shape[0] * K, -1)                                                 # network.py:262-263
    offset, sh, opacity, scaling, rotation = torch.split(par, [3, sh_dim, 1, 2, 4], dim=-1)   # network.py:264-270
    return torch.sigmoid(offset) * 2 - 1.0, sh, scaling + scaling_shift, rotation, opacity + opacity_shift   # :271-278
