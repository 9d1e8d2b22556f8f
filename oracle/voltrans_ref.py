"""Plain-torch fp32 restatement of the reference's volume transformer and coarse decoder -- TEST INFRASTRUCTURE ONLY
(the checker of tests/test_voltrans*.py and the `cpu_baseline.encoder` leg of bench.py; nothing under lara_amd/
imports it).

Follows lightning/network.py:81-102 (`GroupAttBlock.forward`), :138-164 (`VolTransformer.forward`) and :259-278
(`Decoder.forward_coarse`).  Pinned: tests/golden/voltrans_ref.npz holds the output of the REFERENCE's own modules
(tests/golden/make_voltrans_fixture.py imports them from /root/reference); tests/test_voltrans.py checks this
restatement against it to 3e-5, and its autograd against the reference's own gradients (voltrans_grad_ref.npz).
"""
import torch
import torch.nn.functional as F
from torch import nn

E, COND, HEADS, OUT = 256, 800, 16, 80


def build_modules(seed, R, n_layers):
    """Same construction (and RNG consumption) order as the reference: VolTransformer.__init__
    (network.py:126-136) draws pos_embed, then each GroupAttBlock builds norm1, cross_attn, cnn, norm2,
    norm3, mlp (network.py:64-79), then the final norm and the deconvolution."""
    torch.manual_seed(seed)
    m = {"pos": torch.randn(1, E, R, R, R) * (1.0 / E) ** 0.5, "layers": []}
    for _ in range(n_layers):
        blk = {"norm1": nn.LayerNorm(E),
               "mha": nn.MultiheadAttention(embed_dim=E, num_heads=HEADS, kdim=COND, vdim=COND, dropout=0.0,
                                            bias=False, batch_first=True),
               "cnn": nn.Conv3d(E, E, kernel_size=3, padding=1, bias=False),
               "norm2": nn.LayerNorm(E), "norm3": nn.LayerNorm(E),
               "mlp": nn.Sequential(nn.Linear(E, 2 * E), nn.GELU(), nn.Dropout(0.0), nn.Linear(2 * E, E), nn.Dropout(0.0))}
        m["layers"].append(blk)
    m["norm"] = nn.LayerNorm(E, eps=1e-6)
    m["deconv"] = nn.ConvTranspose3d(E, OUT, kernel_size=2, stride=2, padding=0)
    return m


def restated_block(blk, x, cond):
    """x [B, E, R, R, R] fp32, cond [B * (R/2)^3, 4, COND]; plain-torch restatement of network.py:81-102."""
    B, _, R = x.shape[:3]
    g = R // 2
    G = B * g ** 3
    p = x.view(B, E, g, 2, g, 2, g, 2).permute(0, 2, 4, 6, 3, 5, 7, 1).reshape(G, 8, E)
    mha = blk["mha"]
    xn = F.layer_norm(p, (E,), blk["norm1"].weight, blk["norm1"].bias, blk["norm1"].eps)
    q = (xn @ mha.q_proj_weight.t()).view(G, 8, HEADS, 16).transpose(1, 2)
    k = (cond @ mha.k_proj_weight.t()).view(G, 4, HEADS, 16).transpose(1, 2)
    v = (cond @ mha.v_proj_weight.t()).view(G, 4, HEADS, 16).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / 4.0, dim=-1)
    p = p + (a @ v).transpose(1, 2).reshape(G, 8, E) @ mha.out_proj.weight.t()
    h = F.layer_norm(p, (E,), blk["norm2"].weight, blk["norm2"].bias, blk["norm2"].eps)
    p = p + F.linear(F.gelu(F.linear(h, blk["mlp"][0].weight, blk["mlp"][0].bias)), blk["mlp"][3].weight, blk["mlp"][3].bias)
    pn = F.layer_norm(p, (E,), blk["norm3"].weight, blk["norm3"].bias, blk["norm3"].eps)
    vol = pn.view(B, g, g, g, 2, 2, 2, E).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, E, R, R, R)
    return vol + F.conv3d(vol, blk["cnn"].weight, padding=1)


def restated_cond(feats):
    B, V, C, D = feats.shape[:4]
    return feats.permute(0, 3, 4, 5, 1, 2).reshape(B * D ** 3, V, C)


def restated_voltrans(m, feats):
    B = feats.shape[0]
    cond = restated_cond(feats)
    x = m["pos"].expand(B, -1, -1, -1, -1).contiguous()
    for blk in m["layers"]:
        x = restated_block(blk, x, cond)
    xn = F.layer_norm(x.permute(0, 2, 3, 4, 1), (E,), m["norm"].weight, m["norm"].bias, m["norm"].eps)
    up = F.conv_transpose3d(xn.permute(0, 4, 1, 2, 3), m["deconv"].weight, m["deconv"].bias, stride=2)
    return up.permute(0, 2, 3, 4, 1).contiguous()


def build_decoder_coarse(seed, in_dim=OUT, K=2, sh_dim=12, scaling_dim=2, rotation_dim=4, opacity_dim=1):
    """`Decoder.mlp_coarse` as constructed at network.py:224-229 (Linear-ReLU-Linear-ReLU-Linear, xavier / zero bias,
    network.py:243-257)."""
    torch.manual_seed(seed)
    out_dim = 3 + sh_dim + opacity_dim + scaling_dim + rotation_dim
    mlp = nn.Sequential(nn.Linear(in_dim, in_dim), nn.ReLU(), nn.Linear(in_dim, in_dim), nn.ReLU(), nn.Linear(in_dim, out_dim * K))
    for layer in mlp:
        if isinstance(layer, nn.Linear):
            nn.init.xavier_uniform_(layer.weight.data)
            nn.init.zeros_(layer.bias.data)
    return mlp


def restated_decoder_coarse(mlp, feats, opacity_shift, scaling_shift, K=2, sh_dim=12, scaling_dim=2, rotation_dim=4, opacity_dim=1):
    """network.py:259-278: feats [B, 64, 64, 64, 80] -> offset [B,P,3], sh [B,P,4,3], scaling [B,P,2], rotation [B,P,4],
    opacity [B,P,1] with P = 64^3 K."""
    p = mlp(feats).float()
    p = p.view(*p.shape[:-1], K, -1)
    offset, sh, opacity, scaling, rotation = torch.split(p, [3, sh_dim, opacity_dim, scaling_dim, rotation_dim], dim=-1)
    B = opacity.shape[0]
    return (torch.sigmoid(offset).view(B, -1, 3) * 2 - 1.0, sh.reshape(B, -1, sh_dim // 3, 3),
            (scaling + scaling_shift).reshape(B, -1, scaling_dim), rotation.reshape(B, -1, rotation_dim),
            (opacity + opacity_shift).reshape(B, -1, opacity_dim))
