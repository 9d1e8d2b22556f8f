"""Group cross-attention: (CPU) our torch restatement of the op equals the REFERENCE's
GroupAttBlock output stored in tests/golden/groupatt_ref.npz; (GPU) the MFMA kernels match the
restatement.  Tolerances: the kernels compute in bf16 with fp32 accumulation (what the reference
does under bf16-mixed autocast): vs the fp32 restatement max |diff| <= 3e-2 on outputs of magnitude
~1 (bf16 has 8 mantissa bits; K = 800 dot products), mean |diff| <= 4e-3; vs a torch bf16-autocast
run of the same op the mean |diff| must be of the same size as autocast's own error."""
import os

import numpy as np
import pytest
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))


def build_modules(seed):
    """Same construction order as GroupAttBlock.__init__ (network.py:64-67): norm1, then cross_attn."""
    torch.manual_seed(seed)
    norm1 = nn.LayerNorm(256)
    mha = nn.MultiheadAttention(embed_dim=256, num_heads=16, kdim=800, vdim=800, dropout=0.0,
                                bias=False, batch_first=True)
    return norm1, mha


def restated(norm1, mha, x, cond):
    """Plain-torch fp32 restatement of network.py:93 with explicit projections."""
    G = x.shape[0]
    xn = torch.nn.functional.layer_norm(x, (256,), norm1.weight, norm1.bias, norm1.eps)
    q = (xn @ mha.q_proj_weight.t()).view(G, 8, 16, 16).transpose(1, 2)
    k = (cond @ mha.k_proj_weight.t()).view(G, 4, 16, 16).transpose(1, 2)
    v = (cond @ mha.v_proj_weight.t()).view(G, 4, 16, 16).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2) / 4.0, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(G, 8, 256)
    return x + o @ mha.out_proj.weight.t()


def fixture():
    f = np.load(os.path.join(HERE, "golden", "groupatt_ref.npz"))
    seed, G = int(f["seed"]), int(f["G"])
    norm1, mha = build_modules(seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(G, 8, 256, generator=g)
    cond = torch.randn(G, 4, 800, generator=g)
    return f, norm1, mha, x, cond


def test_restatement_matches_reference_groupattblock():
    f, norm1, mha, x, cond = fixture()
    ws = [float(w.double().sum()) for w in (mha.q_proj_weight, mha.k_proj_weight, mha.v_proj_weight,
                                            mha.out_proj.weight)]
    np.testing.assert_allclose(ws, f["wsum"], rtol=1e-9)      # same seeded weights as the reference block
    assert norm1.eps == float(f["ln_eps"])
    with torch.no_grad():
        out = restated(norm1, mha, x, cond)
    np.testing.assert_allclose(out.numpy(), f["out"], atol=2e-5, rtol=1e-5)


@pytest.mark.gpu
def test_mfma_kernels_match_restatement_and_reference(hip_lib):
    from lara_amd.attention import GroupCrossAttention
    f, norm1, mha, x, cond = fixture()
    dev = "cuda:0"
    mod = GroupCrossAttention.from_modules(norm1, mha).to(dev)
    with torch.no_grad():
        y = mod(x.to(dev), cond.to(dev)).cpu()
        ref = restated(norm1, mha, x, cond)
    d = (y - ref).abs()
    assert float(d.max()) <= 3e-2 and float(d.mean()) <= 4e-3, (float(d.max()), float(d.mean()))
    assert float((y - torch.from_numpy(f["out"])).abs().max()) <= 3e-2        # the reference's own output
    # error budget check against torch's own bf16 autocast of the same op (on the GPU)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        auto = (x.to(dev) + mha.to(dev)(norm1.to(dev)(x.to(dev)), cond.to(dev), cond.to(dev),
                                        need_weights=False)[0]).float().cpu()
    assert float(d.mean()) <= 2.0 * float((auto - ref).abs().mean()) + 1e-4


@pytest.mark.gpu
def test_ragged_group_count_and_larger_batch(hip_lib):
    from lara_amd.attention import GroupCrossAttention
    norm1, mha = build_modules(7)
    g = torch.Generator().manual_seed(8)
    for G in (1, 7, 130, 4096):
        x = torch.randn(G, 8, 256, generator=g)
        cond = torch.randn(G, 4, 800, generator=g)
        mod = GroupCrossAttention.from_modules(norm1, mha).to("cuda:0")
        with torch.no_grad():
            y = mod(x.cuda(), cond.cuda()).cpu()
            ref = restated(norm1, mha, x, cond)
        d = (y - ref).abs()
        assert float(d.max()) <= 4e-2 and float(d.mean()) <= 4e-3, (G, float(d.max()), float(d.mean()))
