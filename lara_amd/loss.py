"""The pixel terms of LaRa's training loss on MI355X: lightning/loss.py:17-60 (without the MS-SSIM term: pytorch_msssim is
absent from this image) over the stacked outputs of ``Network.forward``, as one HIP kernel per direction
(``lara_loss_terms_forward`` / ``_backward``, include/lara_loss.h) instead of ~25 elementwise and reduction kernels forward
and ~30 backward.  ``lara_loss(batch, output, it)`` has the signature and return value of ``lara_amd.pipeline.lara_loss``
(itself the restatement of ``Losses.forward``): (loss, scalar_stats).  Opt-in; no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes

import torch

from .rasterizer import _check, load_library

_configured = False
_weights = {}


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
        lib.lara_loss_partial_floats.restype = i64
        lib.lara_loss_partial_floats.argtypes = [i64]
        lib.lara_loss_terms_forward.restype = ctypes.c_int
        lib.lara_loss_terms_forward.argtypes = [i32, i32, i32, i32] + [vp] * 10
        lib.lara_loss_terms_backward.restype = ctypes.c_int
        lib.lara_loss_terms_backward.argtypes = [i32, i32, i32, i32] + [vp] * 13
        _configured = True
    return lib


def _ptr(t):
    return None if t is None else t.data_ptr()


class _LossTerms(torch.autograd.Function):
    """(tar_rgb [B,V,H,W,3], image [B,H,V*W,3], image_fine | None, rend_dist [B,H,V*W] | None, rend_normal | None,
    depth_normal | None, acc_map | None) -> terms [4] = (mse, mse_fine, mean distortion, mean normal error)."""

    @staticmethod
    def forward(ctx, tar, image, image_fine, rend_dist, rend_normal, depth_normal, acc_map):
        if not image.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        f = lambda t: None if t is None else t.detach().float().contiguous()
        tar, image, image_fine, rend_dist, rend_normal, depth_normal, acc_map = map(
            f, (tar, image, image_fine, rend_dist, rend_normal, depth_normal, acc_map))
        B, V, H, W = tar.shape[:4]
        n = B * V * H * W
        if tar.shape != (B, V, H, W, 3) or image.shape != (B, H, V * W, 3) or (image_fine is not None and image_fine.shape != image.shape):
            raise RuntimeError("expected tar_rgb [B,V,H,W,3] and image(_fine) [B,H,V*W,3]")
        if (rend_normal is None) != (depth_normal is None) or (rend_normal is not None and acc_map is None):
            raise RuntimeError("rend_normal, depth_normal and acc_map come together")
        for t, c in ((rend_dist, 1), (rend_normal, 3), (depth_normal, 3), (acc_map, 1)):
            if t is not None and t.numel() != n * c:
                raise RuntimeError("map sizes do not match tar_rgb")
        lib = _lib()
        terms = torch.empty(4, dtype=torch.float32, device=image.device)
        partials = torch.empty(int(lib.lara_loss_partial_floats(n)), dtype=torch.float32, device=image.device)
        with torch.cuda.device(image.device):
            _check(lib.lara_loss_terms_forward(B, V, H, W, tar.data_ptr(), image.data_ptr(), _ptr(image_fine), _ptr(rend_dist),
                                               _ptr(rend_normal), _ptr(depth_normal), _ptr(acc_map), terms.data_ptr(),
                                               partials.data_ptr(), torch.cuda.current_stream(image.device).cuda_stream),
                   "lara_loss_terms_forward")
        ctx.dims = (B, V, H, W)
        ctx.have = (image_fine is not None, rend_dist is not None, rend_normal is not None)
        ctx.save_for_backward(*[t for t in (tar, image, image_fine, rend_normal, depth_normal, acc_map) if t is not None])
        ctx.dist_shape = None if rend_dist is None else rend_dist.shape
        return terms

    @staticmethod
    def backward(ctx, g):
        B, V, H, W = ctx.dims
        have_fine, have_dist, have_normal = ctx.have
        saved = list(ctx.saved_tensors)
        tar, image = saved[0], saved[1]
        image_fine = saved[2] if have_fine else None
        rend_normal, depth_normal, acc_map = (saved[-3], saved[-2], saved[-1]) if have_normal else (None, None, None)
        need = ctx.needs_input_grad
        dev = image.device
        d_image = torch.empty_like(image) if need[1] else None
        d_fine = torch.empty_like(image_fine) if have_fine and need[2] else None
        d_dist = torch.empty(ctx.dist_shape, dtype=torch.float32, device=dev) if have_dist and need[3] else None
        d_rn = torch.empty_like(rend_normal) if have_normal and need[4] else None
        d_dn = torch.empty_like(depth_normal) if have_normal and need[5] else None
        g = g.float().contiguous()
        with torch.cuda.device(dev):
            _check(_lib().lara_loss_terms_backward(B, V, H, W, tar.data_ptr(), image.data_ptr(), _ptr(image_fine), _ptr(rend_normal),
                                                   _ptr(depth_normal), _ptr(acc_map), g.data_ptr(), _ptr(d_image), _ptr(d_fine),
                                                   _ptr(d_dist), _ptr(d_rn), _ptr(d_dn), torch.cuda.current_stream(dev).cuda_stream),
                   "lara_loss_terms_backward")
        return None, d_image, d_fine, d_dist, d_rn, d_dn, None


def lara_loss(batch, output, it=10000):
    """lightning/loss.py:17-60 without MS-SSIM, fused: same arguments and return value as ``lara_amd.pipeline.lara_loss``."""
    if "image" not in output:
        return 0, {}
    fine = output.get("image_fine") if "acc_map_fine" in output else None                  # loss.py:31
    reg = "rend_dist" in output and it > 1000                                               # loss.py:47
    terms = _LossTerms.apply(batch["tar_rgb"], output["image"], fine, output["rend_dist"] if reg else None,
                             output["rend_normal"] if reg else None, output["depth_normal"] if reg else None,
                             output["acc_map"] if reg else None)
    key = (terms.device, fine is not None, reg)
    if key not in _weights:      # loss.py:34, :45 (absent), :49, :57
        _weights[key] = torch.tensor([1.0, 1.0 if fine is not None else 0.0, 1000.0 if reg else 0.0, 0.2 if reg else 0.0], device=terms.device)
    loss = torch.dot(terms, _weights[key])
    t = terms.detach()
    stats = {"mse": t[0]}
    if fine is not None:
        stats["mse_fine"] = t[1]
    if reg:
        stats["distortion"], stats["normal"] = t[2], t[3]
    return loss, stats
