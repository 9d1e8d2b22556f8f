"""LaRa's volume transformer on MI355X matrix cores (forward): ``GroupAttBlock`` and
``VolTransformer`` with the reference's constructor arguments and call signatures
(lightning/network.py:57-102 and :105-164), running on the HIP kernels of ``liblara2dgs.so``.

Differences in *how*, not *what*:

* activations stay fp32 token rows ``[B * R^3, 256]`` in the attention's group-major order for all
  layers (the 3x3x3 convolution gathers its neighbours through that order), so the two
  volume <-> patches permutes per layer of the reference (network.py:82-86, 96-98) disappear;
  ``GroupAttBlock.forward(x, cond, group_axis, block_size)`` still accepts / returns the
  reference's ``[B, C, D, H, W]`` volume for drop-in use and for the parity tests;
* matmuls / the convolution run in bf16 with fp32 accumulation (what ``precision="bf16-mixed"``
  gives the reference, train_lightning.py:74); LayerNorm, softmax, GELU, residuals in fp32.

Inference classes (bf16 weight buffers, no autograd): tensors that require grad raise -- training goes
through ``lara_amd.encoder_train`` (same kernels + the HIP backward, fp32 master parameters).
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from .rasterizer import _check, load_library

_configured = False


class _BlockWeights(ctypes.Structure):  # struct lara_groupblock_weights, include/lara_groupattn.h
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("ln1_w", "ln1_b", "wq", "wkv", "wo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2",
                 "ln3_w", "ln3_b", "wconv")] + [("eps", ctypes.c_float)]


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
        lib.lara_groupblock_workspace_bytes.restype = ctypes.c_int64
        lib.lara_groupblock_workspace_bytes.argtypes = [i32, i32]
        lib.lara_groupblock_forward.restype = ctypes.c_int
        lib.lara_groupblock_forward.argtypes = [i32, i32, i32, vp, vp, ctypes.POINTER(_BlockWeights), vp, vp]
        lib.lara_voltrans_head_forward.restype = ctypes.c_int
        lib.lara_voltrans_head_forward.argtypes = [i32, i32, vp, vp, vp, f32, vp, vp, i32, vp, vp, vp]
        for fn in (lib.lara_tokens_from_volume, lib.lara_volume_from_tokens):
            fn.restype = ctypes.c_int
            fn.argtypes = [i32, i32, i32, vp, vp, vp]
        _configured = True
    return lib


def _require_device(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
    if t.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("lara_amd.encoder holds the inference classes; use lara_amd.encoder_train.VolTransformer to train")


def volume_to_tokens(volume: torch.Tensor) -> torch.Tensor:
    """[B, C, R, R, R] fp32 -> group-major token rows [B * R^3, C]."""
    _require_device(volume)
    B, C, R = volume.shape[0], volume.shape[1], volume.shape[2]
    v = volume.float().contiguous()
    out = torch.empty(B * R ** 3, C, dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _check(_lib().lara_tokens_from_volume(B, R, C, v.data_ptr(), out.data_ptr(),
                                              torch.cuda.current_stream(v.device).cuda_stream), "lara_tokens_from_volume")
    return out


def tokens_to_volume(tokens: torch.Tensor, B: int, R: int) -> torch.Tensor:
    _require_device(tokens)
    C = tokens.shape[1]
    t = tokens.float().contiguous()
    out = torch.empty(B, C, R, R, R, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _check(_lib().lara_volume_from_tokens(B, R, C, t.data_ptr(), out.data_ptr(),
                                              torch.cuda.current_stream(t.device).cuda_stream), "lara_volume_from_tokens")
    return out


def cond_tokens(image_feats: torch.Tensor, n_group: int) -> torch.Tensor:
    """network.py:145-150 for block_size 1: [B, V, C, D, D, D] -> bf16 [B * D^3, V, C], one token per view."""
    B, V, C, D = image_feats.shape[:4]
    if D != n_group or V != 4:
        raise RuntimeError("kernels are specialised for one image-feature voxel per group and 4 input views "
                           "(configs/base.yaml: n_groups [16], 16^3 feature volume, n_views 4)")
    return image_feats.permute(0, 3, 4, 5, 1, 2).reshape(B * D ** 3, V, C).to(torch.bfloat16).contiguous()


class GroupAttBlock(nn.Module):
    """Mirror of the reference ``GroupAttBlock`` (network.py:57-102)."""

    def __init__(self, inner_dim: int, cond_dim: int, num_heads: int, eps: float = 1e-5,
                 attn_drop: float = 0., attn_bias: bool = False, mlp_ratio: float = 2., mlp_drop: float = 0.):
        super().__init__()
        if inner_dim != 256 or num_heads != 16 or attn_bias or mlp_ratio != 2.:
            raise ValueError("kernels are specialised for LaRa's 256-dim, 16-head, bias-free blocks "
                             "(configs/base.yaml:17-20)")
        self.inner_dim, self.cond_dim = inner_dim, cond_dim
        self.eps = 1e-5  # the reference builds its three norms with nn.LayerNorm's default (network.py:64,71,72)
        f = lambda *s: nn.Parameter(torch.zeros(*s))
        h = lambda *s: torch.zeros(*s, dtype=torch.bfloat16)
        self.ln1_w, self.ln1_b, self.ln2_w, self.ln2_b = f(256), f(256), f(256), f(256)
        self.ln3_w, self.ln3_b, self.b1, self.b2 = f(256), f(256), f(512), f(256)
        self.register_buffer("wq", h(256, 256))
        self.register_buffer("wkv", h(512, cond_dim))
        self.register_buffer("wo", h(256, 256))
        self.register_buffer("w1", h(512, 256))
        self.register_buffer("w2", h(256, 512))
        self.register_buffer("wconv", h(256, 27 * 256))
        self._ws = None

    @classmethod
    def from_reference(cls, block) -> "GroupAttBlock":
        """``block``: a reference ``GroupAttBlock`` (or any module with the same attribute names)."""
        mha = block.cross_attn
        m = cls(mha.embed_dim, mha.kdim, mha.num_heads)
        m.eps = float(block.norm1.eps)
        bf = torch.bfloat16
        with torch.no_grad():
            for dst, src in ((m.ln1_w, block.norm1.weight), (m.ln1_b, block.norm1.bias),
                             (m.ln2_w, block.norm2.weight), (m.ln2_b, block.norm2.bias),
                             (m.ln3_w, block.norm3.weight), (m.ln3_b, block.norm3.bias),
                             (m.b1, block.mlp[0].bias), (m.b2, block.mlp[3].bias)):
                dst.copy_(src)
            m.wq.copy_(mha.q_proj_weight.to(bf))
            m.wkv.copy_(torch.cat([mha.k_proj_weight, mha.v_proj_weight], 0).to(bf))
            m.wo.copy_(mha.out_proj.weight.to(bf))
            m.w1.copy_(block.mlp[0].weight.to(bf))
            m.w2.copy_(block.mlp[3].weight.to(bf))
            # cnn.weight [out, in, kd, kh, kw] -> [out][kd][kh][kw][in]: K-contiguous per filter tap
            m.wconv.copy_(block.cnn.weight.permute(0, 2, 3, 4, 1).reshape(256, 27 * 256).to(bf))
        return m

    def _weights(self) -> _BlockWeights:
        w = _BlockWeights()
        for n in ("ln1_w", "ln1_b", "wq", "wkv", "wo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2", "ln3_w", "ln3_b", "wconv"):
            setattr(w, n, getattr(self, n).data_ptr())
        w.eps = self.eps
        return w

    def forward_tokens(self, x: torch.Tensor, cond_bf16: torch.Tensor, scenes: int, R: int) -> torch.Tensor:
        """In place on fp32 token rows ``x`` [scenes * R^3, 256] (group-major); returns ``x``."""
        _require_device(x)
        if x.dtype != torch.float32 or not x.is_contiguous() or x.shape != (scenes * R ** 3, 256):
            raise RuntimeError("expected contiguous fp32 token rows [scenes * R^3, 256]")
        if cond_bf16.dtype != torch.bfloat16 or cond_bf16.shape != (scenes * (R // 2) ** 3, 4, self.cond_dim):
            raise RuntimeError("expected bf16 cond [scenes * (R/2)^3, 4, cond_dim]")
        lib = _lib()
        need = lib.lara_groupblock_workspace_bytes(scenes, R)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        w = self._weights()
        with torch.cuda.device(x.device):
            rc = lib.lara_groupblock_forward(scenes, R, self.cond_dim, x.data_ptr(), cond_bf16.contiguous().data_ptr(),
                                             ctypes.byref(w), self._ws.data_ptr(),
                                             torch.cuda.current_stream(x.device).cuda_stream)
        _check(rc, "lara_groupblock_forward")
        return x

    def forward(self, x: torch.Tensor, cond: torch.Tensor, group_axis: int, block_size: int) -> torch.Tensor:
        """The reference's signature (network.py:81): ``x`` [B, C, D, H, W], ``cond`` [B * G, 4, cond_dim]."""
        B, C, D = x.shape[0], x.shape[1], x.shape[2]
        if block_size != 2 or group_axis * block_size != D:
            raise RuntimeError("kernels are specialised for block_size 2 (configs/base.yaml: 32^3 volume, 16 groups per axis)")
        tok = volume_to_tokens(x)
        self.forward_tokens(tok, cond.to(torch.bfloat16), B, D)
        return tokens_to_volume(tok, B, D)


class VolTransformer(nn.Module):
    """Mirror of the reference ``VolTransformer`` (network.py:105-164): ``forward(image_feats)`` with
    ``image_feats`` [B, n_views, C, D, H, W] returns ``[B, 2R, 2R, 2R, out_dim]`` (channels last)."""

    def __init__(self, embed_dim: int, image_feat_dim: int, n_groups: list, vol_low_res: int, vol_high_res: int,
                 out_dim: int, num_layers: int, num_heads: int, eps: float = 1e-6):
        super().__init__()
        if len(n_groups) != 1 or vol_low_res != 2 * n_groups[0] or vol_high_res != 2 * vol_low_res or out_dim % 4:
            raise ValueError("kernels are specialised for one group size with block_size 2 and a x2 deconvolution "
                             "(configs/base.yaml: n_groups [16], vol 32 -> 64)")
        self.vol_low_res, self.vol_high_res, self.out_dim, self.n_groups = vol_low_res, vol_high_res, out_dim, list(n_groups)
        self.embed_dim, self.eps = embed_dim, eps
        self.pos_embed = nn.Parameter(torch.zeros(1, embed_dim, vol_low_res, vol_low_res, vol_low_res))
        self.layers = nn.ModuleList([GroupAttBlock(embed_dim, image_feat_dim, num_heads) for _ in range(num_layers)])
        self.norm_w, self.norm_b = nn.Parameter(torch.ones(embed_dim)), nn.Parameter(torch.zeros(embed_dim))
        self.register_buffer("wdeconv", torch.zeros(8 * out_dim, embed_dim, dtype=torch.bfloat16))
        self.deconv_b = nn.Parameter(torch.zeros(out_dim))
        self._pos_tokens = None
        self._ws = None
        self._graphs = {}

    @classmethod
    def from_reference(cls, vt) -> "VolTransformer":
        mha = vt.layers[0].cross_attn
        m = cls(vt.embed_dim, mha.kdim, vt.n_groups, vt.vol_low_res, vt.vol_high_res, vt.out_dim, len(vt.layers),
                mha.num_heads, float(vt.norm.eps))
        with torch.no_grad():
            m.pos_embed.copy_(vt.pos_embed)
            m.layers = nn.ModuleList([GroupAttBlock.from_reference(b) for b in vt.layers])
            m.norm_w.copy_(vt.norm.weight)
            m.norm_b.copy_(vt.norm.bias)
            # deconv.weight [in, out, i, j, k] -> [(i*2+j)*2+k][out][in]
            m.wdeconv.copy_(vt.deconv.weight.permute(2, 3, 4, 1, 0).reshape(8 * vt.out_dim, vt.embed_dim).to(torch.bfloat16))
            m.deconv_b.copy_(vt.deconv.bias)
        return m

    def _run(self, cond: torch.Tensor, x: torch.Tensor, out: torch.Tensor, B: int) -> None:
        """The 12 blocks + tail on the current stream: x <- pos tokens; x <- block(x, cond) ...; out <- tail(x)."""
        R, dev = self.vol_low_res, cond.device
        x.view(B, -1, self.embed_dim).copy_(self._pos_tokens)  # network.py:152: the same positional volume per scene
        for layer in self.layers:
            layer.forward_tokens(x, cond, B, R)
        with torch.cuda.device(dev):
            rc = _lib().lara_voltrans_head_forward(B, R, x.data_ptr(), self.norm_w.data_ptr(), self.norm_b.data_ptr(),
                                                   float(self.eps), self.wdeconv.data_ptr(), self.deconv_b.data_ptr(),
                                                   self.out_dim, out.data_ptr(), self._ws.data_ptr(),
                                                   torch.cuda.current_stream(dev).cuda_stream)
        _check(rc, "lara_voltrans_head_forward")

    def forward(self, image_feats: torch.Tensor, use_graph: bool = False) -> torch.Tensor:
        """``use_graph``: replay the ~125 launches of a forward as ONE HIP graph (captured on first use per
        batch size; weights and buffers must stay where they are; the result is a fresh tensor).  Off by
        default: the launches are long enough (30-450 us) that the host stays ahead of the GPU anyway --
        measured 12.6 ms (launches) vs 12.7 ms (graph) at 4 scenes, 4.35 vs 4.50 ms at one."""
        _require_device(image_feats)
        B, R = image_feats.shape[0], self.vol_low_res
        dev = image_feats.device
        cond = cond_tokens(image_feats, self.n_groups[0])
        # positional volume in token order, cached with the identity of what it was computed from: a
        # load_state_dict / copy_ into pos_embed bumps the version, and any graph captured on the old copy goes too
        pos_key = (self.pos_embed.data_ptr(), self.pos_embed._version, dev)
        if self._pos_tokens is None or getattr(self, "_pos_key", None) != pos_key:
            with torch.no_grad():
                self._pos_tokens = volume_to_tokens(self.pos_embed.detach().to(dev))
            self._pos_key = pos_key
            self._graphs = {}
        need = B * R ** 3 * 512
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._graphs = {}
        if not use_graph:
            x = torch.empty(B * R ** 3, self.embed_dim, dtype=torch.float32, device=dev)
            out = torch.empty(B, 2 * R, 2 * R, 2 * R, self.out_dim, dtype=torch.float32, device=dev)
            self._run(cond, x, out, B)
            return out
        key = (B, dev.index)
        entry = self._graphs.get(key)
        if entry is None:
            s_cond = torch.empty_like(cond)
            s_x = torch.empty(B * R ** 3, self.embed_dim, dtype=torch.float32, device=dev)
            s_out = torch.empty(B, 2 * R, 2 * R, 2 * R, self.out_dim, dtype=torch.float32, device=dev)
            s_cond.copy_(cond)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up outside capture: workspaces, function attributes, lazy init
                self._run(s_cond, s_x, s_out, B)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._run(s_cond, s_x, s_out, B)
            entry = (graph, s_cond, s_x, s_out)
            self._graphs[key] = entry
        graph, s_cond, s_x, s_out = entry
        s_cond.copy_(cond)
        graph.replay()
        return s_out.clone()
