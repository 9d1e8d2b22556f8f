"""LaRa's fine-stage point sampler on MI355X: ``get_point_feats`` with the
reference's signature (lightning/network.py:390-411), running the projection + 8-channel bilinear gather +
depth residual as one HIP kernel per direction (``lara_point_feats_forward`` / ``_backward``,
include/lara_pointfeat.h) instead of `projection`, a [V,8,h,w] concatenation + permute, `F.grid_sample` and
their autograd counterparts.  Differentiable w.r.t. the points (the Gaussian centres: gradients reach the
decoder's offsets) and the three render-derived maps (image, acc_map, depth: gradients reach the coarse
rasteriser pass); the input images get none, as in the reference where they are data.

Opt-in: bind it over the reference's method, e.g. ``Network.get_point_feats = lara_amd.fine.get_point_feats``.
No CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes

import torch

from .rasterizer import _check, load_library

_configured = False


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        lib.lara_point_feats_forward.restype = ctypes.c_int
        lib.lara_point_feats_forward.argtypes = [i32, i32, i32, i32] + [vp] * 10
        lib.lara_point_feats_backward.restype = ctypes.c_int
        lib.lara_point_feats_backward.argtypes = [i32, i32, i32, i32] + [vp] * 14
        lib.lara_point_feats_workspace_bytes.restype = ctypes.c_int64
        lib.lara_point_feats_workspace_bytes.argtypes = [i32, i32, i32]
        _configured = True
    return lib


def _workspace(device, V, h, w):
    n = _lib().lara_point_feats_workspace_bytes(V, h, w)
    if n < 0:
        _check(int(n), "lara_point_feats_workspace_bytes")
    return torch.empty(int(n), dtype=torch.uint8, device=device)


class _PointFeats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, w2cs, ixts, img_ref, image, acc_map, depth):
        if not points.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        f = lambda t: t.detach().float().contiguous()
        points, w2cs, ixts, img_ref, image, acc_map, depth = map(f, (points, w2cs, ixts, img_ref, image, acc_map, depth))
        n, V, h, w = points.shape[0], img_ref.shape[0], img_ref.shape[2], img_ref.shape[3]
        if (points.shape != (n, 3) or w2cs.shape != (V, 4, 4) or ixts.shape != (V, 3, 3) or img_ref.shape != (V, 3, h, w)
                or image.shape != (V, h, w, 3) or acc_map.shape != (V, h, w) or depth.shape != (V, h, w, 1)):
            raise RuntimeError("expected points [n,3], w2cs [V,4,4], ixts [V,3,3], img_ref [V,3,h,w], image [V,h,w,3], "
                               "acc_map [V,h,w], depth [V,h,w,1]")
        out = torch.empty(V, 8, n, dtype=torch.float32, device=points.device)
        ws = _workspace(points.device, V, h, w)
        with torch.cuda.device(points.device):
            _check(_lib().lara_point_feats_forward(n, V, h, w, points.data_ptr(), w2cs.data_ptr(), ixts.data_ptr(),
                                                   img_ref.data_ptr(), image.data_ptr(), acc_map.data_ptr(), depth.data_ptr(),
                                                   out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream(points.device).cuda_stream),
                   "lara_point_feats_forward")
        ctx.save_for_backward(points, w2cs, ixts, img_ref, image, acc_map, depth)
        return out

    @staticmethod
    def backward(ctx, g_out):
        points, w2cs, ixts, img_ref, image, acc_map, depth = ctx.saved_tensors
        n, V, h, w = points.shape[0], img_ref.shape[0], img_ref.shape[2], img_ref.shape[3]
        g_out = g_out.float().contiguous()
        need = ctx.needs_input_grad
        d_points = torch.empty_like(points)
        d_image = torch.zeros_like(image) if need[4] else None
        d_acc = torch.zeros_like(acc_map) if need[5] else None
        d_depth = torch.zeros_like(depth) if need[6] else None
        ptr = lambda t: None if t is None else t.data_ptr()
        ws = _workspace(points.device, V, h, w)
        with torch.cuda.device(points.device):
            _check(_lib().lara_point_feats_backward(n, V, h, w, points.data_ptr(), w2cs.data_ptr(), ixts.data_ptr(),
                                                    img_ref.data_ptr(), image.data_ptr(), acc_map.data_ptr(), depth.data_ptr(),
                                                    g_out.data_ptr(), d_points.data_ptr(), ptr(d_image), ptr(d_acc), ptr(d_depth),
                                                    ws.data_ptr(), torch.cuda.current_stream(points.device).cuda_stream),
                   "lara_point_feats_backward")
        return d_points if need[0] else None, None, None, None, d_image, d_acc, d_depth


def sample_point_feats(points, w2cs, ixts, img_ref, image, acc_map, depth):
    """points [n,3] -> [V, 8, n]: channels 0-2 the input image, 3-5 the coarse render, 6 acc_map, 7 |depth - z|."""
    return _PointFeats.apply(points, w2cs, ixts, img_ref, image, acc_map, depth)


def get_point_feats(self, idx, img_ref, renderings, n_views_sel, batch, points, mask):
    """Same arguments and return value as ``Network.get_point_feats`` (network.py:390-411); ``self`` is unused
    (the reference reads only ``self.device`` from it)."""
    points = points[mask]
    src_ixts = batch['tar_ixt'][idx, :n_views_sel].reshape(-1, 3, 3)
    src_w2cs = batch['tar_w2c'][idx, :n_views_sel].reshape(-1, 4, 4)
    feats = sample_point_feats(points, src_w2cs, src_ixts, img_ref, renderings['image'], renderings['acc_map'],
                               renderings['depth'])
    return feats, mask
