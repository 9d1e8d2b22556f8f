"""Group cross-attention of LaRa's volume transformer on MI355X matrix cores (forward).

Mirrors the attention step of ``GroupAttBlock.forward`` (lightning/network.py:88-93):

    patches = patches + self.cross_attn(self.norm1(patches), cond, cond, need_weights=False)[0]

with ``cross_attn = nn.MultiheadAttention(256, 16, kdim=800, vdim=800, bias=False,
batch_first=True)`` (network.py:65-67).  ``GroupCrossAttention.from_reference(block)`` takes its
weights from a reference ``GroupAttBlock``; ``forward(patches, cond)`` returns the updated patches.
Compute is bf16 MFMA with fp32 accumulation (the reference runs this under bf16-mixed autocast),
LayerNorm / softmax / residual in fp32.  Forward only in this round: calling it on tensors that
require grad raises (no silent fallback to torch).
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from .rasterizer import _check, load_library

_configured = False


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        lib.lara_groupattn_workspace_bytes.restype = ctypes.c_int64
        lib.lara_groupattn_workspace_bytes.argtypes = [i32]
        lib.lara_groupattn_forward.restype = ctypes.c_int
        lib.lara_groupattn_forward.argtypes = [i32, i32, vp, vp, vp, vp, ctypes.c_float, vp, vp, vp, vp, vp, vp]
        _configured = True
    return lib


class GroupCrossAttention(nn.Module):
    """LN -> cross-MHA(256, 16 heads, kdim = vdim = cond_dim, no bias) -> residual, per group."""

    def __init__(self, embed_dim: int = 256, cond_dim: int = 800, num_heads: int = 16, eps: float = 1e-5):
        super().__init__()
        if embed_dim != 256 or num_heads != 16:
            raise ValueError("kernels are specialised for LaRa's 256-dim, 16-head blocks (configs/base.yaml:17-20)")
        self.embed_dim, self.cond_dim, self.num_heads, self.eps = embed_dim, cond_dim, num_heads, eps
        self.ln_weight = nn.Parameter(torch.ones(embed_dim))
        self.ln_bias = nn.Parameter(torch.zeros(embed_dim))
        self.register_buffer("wq", torch.zeros(embed_dim, embed_dim, dtype=torch.bfloat16))
        self.register_buffer("wkv", torch.zeros(2 * embed_dim, cond_dim, dtype=torch.bfloat16))
        self.register_buffer("wo", torch.zeros(embed_dim, embed_dim, dtype=torch.bfloat16))
        self._ws = None

    @classmethod
    def from_modules(cls, norm1: nn.LayerNorm, mha: nn.MultiheadAttention) -> "GroupCrossAttention":
        m = cls(mha.embed_dim, mha.kdim, mha.num_heads, norm1.eps)
        with torch.no_grad():
            m.ln_weight.copy_(norm1.weight)
            m.ln_bias.copy_(norm1.bias)
            m.wq.copy_(mha.q_proj_weight.to(torch.bfloat16))
            m.wkv.copy_(torch.cat([mha.k_proj_weight, mha.v_proj_weight], 0).to(torch.bfloat16))
            m.wo.copy_(mha.out_proj.weight.to(torch.bfloat16))
        return m

    @classmethod
    def from_reference(cls, block) -> "GroupCrossAttention":
        """``block``: a reference ``GroupAttBlock`` (lightning/network.py:57-79)."""
        return cls.from_modules(block.norm1, block.cross_attn)

    def forward(self, patches: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        if patches.requires_grad or cond.requires_grad:
            raise RuntimeError("lara_amd.GroupCrossAttention is forward-only in this round")
        if not patches.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        G = patches.shape[0]
        if patches.shape[1:] != (8, self.embed_dim) or cond.shape != (G, 4, self.cond_dim):
            raise RuntimeError("expected patches [G,8,256] and cond [G,4,cond_dim]")
        lib = _lib()
        x = patches.float().contiguous()
        cond_bf16 = cond.to(torch.bfloat16).contiguous()
        y = torch.empty_like(x)
        need = lib.lara_groupattn_workspace_bytes(G)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.lara_groupattn_forward(
                G, self.cond_dim, x.data_ptr(), cond_bf16.data_ptr(), self.ln_weight.data_ptr(),
                self.ln_bias.data_ptr(), float(self.eps), self.wq.data_ptr(), self.wkv.data_ptr(),
                self.wo.data_ptr(), y.data_ptr(), self._ws.data_ptr(),
                torch.cuda.current_stream(x.device).cuda_stream)
        _check(rc, "lara_groupattn_forward")
        return y
