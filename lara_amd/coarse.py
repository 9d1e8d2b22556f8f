"""LaRa's coarse Gaussian decoder on MI355X: ``Decoder.forward_coarse`` (lightning/network.py:259-278) as one HIP kernel
per direction (``lara_coarse_decoder_forward`` / ``_backward``, include/lara_coarsedec.h) instead of three bf16 GEMMs,
their casts, two ReLUs, a split and three activations -- and, backward, eleven GEMMs / reductions over the 10^6 voxel rows.

``forward_coarse(decoder, feats, opacity_shift, scaling_shift)`` takes the reference's own ``Decoder`` (its ``mlp_coarse``
parameters stay the trainable ones) and returns the reference's five tensors in the reference's order
(offset, sh, scaling, rotation, opacity); bind it with ``Decoder.forward_coarse = lara_amd.coarse.forward_coarse``.
Arithmetic: bf16 operands, fp32 accumulation, bf16 layer results -- what the three Linear layers do under the bf16-mixed
autocast the reference trains with (train_lightning.py:74).  Opt-in; no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from .rasterizer import _check, load_library

_configured = False
_F, _FA, _O = 80, 88, 48


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        lib.lara_coarse_decoder_padded_rows.restype = i64
        lib.lara_coarse_decoder_padded_rows.argtypes = [i64]
        lib.lara_coarse_decoder_forward.restype = ctypes.c_int
        lib.lara_coarse_decoder_forward.argtypes = [i32, i32, i32] + [vp] * 7 + [f32, f32] + [vp] * 6
        lib.lara_coarse_decoder_backward.restype = ctypes.c_int
        lib.lara_coarse_decoder_backward.argtypes = [i32, i32, i32] + [vp] * 20
        lib.lara_gemm_tn_workspace_bytes.restype = i64
        lib.lara_gemm_tn_workspace_bytes.argtypes = []
        lib.lara_gemm_tn_bf16.restype = ctypes.c_int
        lib.lara_gemm_tn_bf16.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp]
        _configured = True
    return lib


def _ptr(t):
    return None if t is None else t.data_ptr()


class _CoarseDecoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, K, sh_dim, opacity_shift, scaling_shift):
        if not x.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        f = lambda t: t.detach().float().contiguous()
        x, w1, b1, w2, b2, w3, b3 = map(f, (x, w1, b1, w2, b2, w3, b3))
        M, n_par = x.shape[0], K * (10 + sh_dim)
        if (x.shape != (M, _F) or w1.shape != (_F, _F) or w2.shape != (_F, _F) or w3.shape != (n_par, _F) or n_par > _O
                or b1.shape != (_F,) or b2.shape != (_F,) or b3.shape != (n_par,)):
            raise RuntimeError("expected x [M,80], Linear(80,80), Linear(80,80), Linear(80, K*(10+sh_dim) <= 48)")
        new = lambda c: torch.empty(M, K * c, dtype=torch.float32, device=x.device)
        offset, sh, scaling, rotation, opacity = new(3), new(sh_dim), new(2), new(4), new(1)
        with torch.cuda.device(x.device):
            _check(_lib().lara_coarse_decoder_forward(M, K, sh_dim, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                      b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), float(opacity_shift),
                                                      float(scaling_shift), offset.data_ptr(), sh.data_ptr(), scaling.data_ptr(),
                                                      rotation.data_ptr(), opacity.data_ptr(),
                                                      torch.cuda.current_stream(x.device).cuda_stream), "lara_coarse_decoder_forward")
        ctx.save_for_backward(x, w1, b1, w2, b2, w3, offset)
        ctx.dims = (M, K, sh_dim, n_par)
        ctx.set_materialize_grads(False)
        return offset, sh, scaling, rotation, opacity

    @staticmethod
    def backward(ctx, g_offset, g_sh, g_scaling, g_rotation, g_opacity):
        x, w1, b1, w2, b2, w3, offset = ctx.saved_tensors
        M, K, sh_dim, n_par = ctx.dims
        lib = _lib()
        gs = [None if g is None else g.float().contiguous() for g in (g_offset, g_sh, g_scaling, g_rotation, g_opacity)]
        dev = x.device
        Mp = int(lib.lara_coarse_decoder_padded_rows(M))
        dx = torch.empty_like(x)
        # the factor matrices of the parameter gradients (bf16): one allocation, carved
        cols = (_FA, _FA, _FA, _F, _F, _O)
        slab = torch.empty(Mp * sum(cols), dtype=torch.bfloat16, device=dev)
        mats, o = [], 0
        for c in cols:
            mats.append(slab[o:o + Mp * c].view(Mp, c))
            o += Mp * c
        xb, h1, h2, dz1, dz2, dz3 = mats
        grads = torch.zeros((2 * _F + _O) * _FA, dtype=torch.float32, device=dev)     # [dW | db | 0] of the three layers
        g1, g2, g3 = grads[:_F * _FA].view(_F, _FA), grads[_F * _FA:2 * _F * _FA].view(_F, _FA), grads[2 * _F * _FA:].view(_O, _FA)
        ws = torch.empty(int(lib.lara_gemm_tn_workspace_bytes()), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            s = torch.cuda.current_stream(dev).cuda_stream
            _check(lib.lara_coarse_decoder_backward(M, K, sh_dim, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                    b2.data_ptr(), w3.data_ptr(), offset.data_ptr(), *[_ptr(g) for g in gs],
                                                    dx.data_ptr(), xb.data_ptr(), h1.data_ptr(), h2.data_ptr(), dz1.data_ptr(),
                                                    dz2.data_ptr(), dz3.data_ptr(), s), "lara_coarse_decoder_backward")
            if M:
                for dz, act, g in ((dz1, xb, g1), (dz2, h1, g2), (dz3, h2, g3)):      # G += dz^T [act | 1 | 0]
                    _check(lib.lara_gemm_tn_bf16(Mp, dz.shape[1], _FA, dz.data_ptr(), act.data_ptr(), g.data_ptr(), ws.data_ptr(), s),
                           "lara_gemm_tn_bf16")
        return (dx, g1[:, :_F], g1[:, _F], g2[:, :_F], g2[:, _F], g3[:n_par, :_F], g3[:n_par, _F], None, None, None, None)


def supported(decoder) -> bool:
    """The sizes the kernels are built for: Linear(80,80) ReLU Linear(80,80) ReLU Linear(80, K (10 + sh_dim) <= 48)."""
    seq = getattr(decoder, "mlp_coarse", None)
    if any(not hasattr(decoder, a) for a in ("K", "sh_dim", "opacity_dim", "scaling_dim", "rotation_dim")):
        return False
    if seq is None or len(seq) != 5 or not all(isinstance(seq[i], nn.Linear) for i in (0, 2, 4)):
        return False
    if not all(isinstance(seq[i], nn.ReLU) for i in (1, 3)) or any(seq[i].bias is None for i in (0, 2, 4)):
        return False
    return (tuple(seq[0].weight.shape) == (_F, _F) and tuple(seq[2].weight.shape) == (_F, _F)
            and tuple(seq[4].weight.shape) == (decoder.K * (10 + decoder.sh_dim), _F) and seq[4].weight.shape[0] <= _O
            and decoder.opacity_dim == 1 and decoder.scaling_dim == 2 and decoder.rotation_dim == 4)


def forward_coarse(decoder, feats, opacity_shift, scaling_shift):
    """Same arguments and return value as ``Decoder.forward_coarse`` (network.py:259-278): feats [B,...,80] ->
    (offset [B,P,3] in (-1,1), sh [B,P,sh_dim/3,3], scaling [B,P,2], rotation [B,P,4], opacity [B,P,1]), P = voxels * K."""
    if not supported(decoder):
        raise RuntimeError("lara_amd.coarse: mlp_coarse must be Linear(80,80) ReLU Linear(80,80) ReLU Linear(80, K*(10+sh_dim) <= 48)")
    seq, K, B = decoder.mlp_coarse, decoder.K, feats.shape[0]
    offset, sh, scaling, rotation, opacity = _CoarseDecoder.apply(
        feats.reshape(-1, feats.shape[-1]), seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, seq[4].weight, seq[4].bias,
        K, decoder.sh_dim, opacity_shift, scaling_shift)
    return (offset.view(B, -1, 3), sh.view(B, -1, decoder.sh_dim // 3, 3), scaling.view(B, -1, 2), rotation.view(B, -1, 4),
            opacity.view(B, -1, 1))
