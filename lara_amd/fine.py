"""LaRa's fine-stage point sampler on MI355X: ``get_point_feats`` with the
reference's signature (lightning/network.py:390-411), running the projection + 8-channel bilinear gather +
depth residual as one HIP kernel per direction (``lara_point_feats_forward`` / ``_backward``,
include/lara_pointfeat.h) instead of `projection`, a [V,8,h,w] concatenation + permute, `F.grid_sample` and
their autograd counterparts.  Differentiable w.r.t. the points (the Gaussian centres: gradients reach the
decoder's offsets) and the three render-derived maps (image, acc_map, depth: gradients reach the coarse
rasteriser pass); the input images get none, as in the reference where they are data.

Opt-in: bind it over the reference's method, e.g. ``Network.get_point_feats = lara_amd.fine.get_point_feats``.
No CPU path: tensors must live on the GPU.

Second half of the row: ``Decoder.forward_fine`` (network.py:280-284) -- LayerNorm, a one-query / four-view
cross-attention (8 heads of 10) and a two-layer MLP on every surviving Gaussian -- as one HIP kernel per direction
(``lara_fine_decoder_forward`` / ``_backward``, include/lara_finedec.h).  ``forward_fine(decoder, volume_feat,
point_feats)`` takes the reference's own ``Decoder`` (its parameters stay the trainable ones) and the same two
arguments; bind it with ``Decoder.forward_fine = lara_amd.fine.forward_fine``.
"""
from __future__ import annotations

import ctypes
import math
import os

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
        lib.lara_point_feats_forward_concat.restype = ctypes.c_int
        lib.lara_point_feats_forward_concat.argtypes = [i32, i32, i32, i32, i32] + [vp] * 10
        lib.lara_point_feats_backward_concat.restype = ctypes.c_int
        lib.lara_point_feats_backward_concat.argtypes = [i32, i32, i32, i32, i32] + [vp] * 14
        lib.lara_point_feats_workspace_bytes.restype = ctypes.c_int64
        lib.lara_point_feats_workspace_bytes.argtypes = [i32, i32, i32]
        lib.lara_fine_decoder_forward.restype = ctypes.c_int
        lib.lara_fine_decoder_forward.argtypes = [i32] + [vp] * 9
        lib.lara_fine_decoder_backward.restype = ctypes.c_int
        lib.lara_fine_decoder_backward.argtypes = [i32] + [vp] * 15
        lib.lara_fine_wgrad_floats.restype = i32
        lib.lara_fine_wgrad_workspace_bytes.restype = ctypes.c_int64
        lib.lara_fine_wgrad_workspace_bytes.argtypes = [i32]
        lib.lara_fine_decoder_wgrad.restype = ctypes.c_int
        lib.lara_fine_decoder_wgrad.argtypes = [i32] + [vp] * 9
        lib.lara_fine_ln_blocks.restype = i32
        lib.lara_fine_ln_blocks.argtypes = [i32]
        lib.lara_fine_ln_forward.restype = ctypes.c_int
        lib.lara_fine_ln_forward.argtypes = [i32, vp, vp, vp, ctypes.c_float, vp, vp, vp]
        lib.lara_fine_ln_backward.restype = ctypes.c_int
        lib.lara_fine_ln_backward.argtypes = [i32] + [vp] * 7
        lib.lara_take_rows.restype = ctypes.c_int
        lib.lara_take_rows.argtypes = [i32, vp, i32, ctypes.POINTER(_RowsItem), i32, vp]
        lib.lara_voxel_rows.restype = ctypes.c_int
        lib.lara_voxel_rows.argtypes = [i32, i32, vp, vp, vp, i32, vp]
        _configured = True
    return lib


def _workspace(device, V, h, w):
    n = _lib().lara_point_feats_workspace_bytes(V, h, w)
    if n < 0:
        _check(int(n), "lara_point_feats_workspace_bytes")
    return torch.empty(int(n), dtype=torch.uint8, device=device)


class _PointFeats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, w2cs, ixts, img_ref, image, acc_map, depth, row_views):
        if not points.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        f = lambda t: t.detach().float().contiguous()
        points, w2cs, ixts, img_ref, image, acc_map, depth = map(f, (points, w2cs, ixts, img_ref, image, acc_map, depth))
        n, V, h, w = points.shape[0], img_ref.shape[0], img_ref.shape[2], img_ref.shape[3]
        if points.shape != (n, 3) or w2cs.shape != (V, 4, 4) or ixts.shape != (V, 3, 3) or img_ref.shape != (V, 3, h, w):
            raise RuntimeError("expected points [n,3], w2cs [V,4,4], ixts [V,3,3], img_ref [V,3,h,w]")
        if row_views:      # the maps of `row_views` >= V views side by side
            R = int(row_views)
            if R < V or image.shape != (h, R * w, 3) or acc_map.numel() != h * R * w or depth.shape != (h, R * w, 1):
                raise RuntimeError("expected points [n,3], ... and image [h,R*w,3], acc_map [h,R*w], depth [h,R*w,1] with R = row_views >= V")
        elif image.shape != (V, h, w, 3) or acc_map.shape != (V, h, w) or depth.shape != (V, h, w, 1):
            raise RuntimeError("expected points [n,3], w2cs [V,4,4], ixts [V,3,3], img_ref [V,3,h,w], image [V,h,w,3], acc_map [V,h,w], depth [V,h,w,1]")
        out = torch.empty(V, 8, n, dtype=torch.float32, device=points.device)
        ws = _workspace(points.device, V, h, w)
        with torch.cuda.device(points.device):
            _check(_lib().lara_point_feats_forward_concat(n, V, int(row_views or 0), h, w, points.data_ptr(), w2cs.data_ptr(),
                                                          ixts.data_ptr(), img_ref.data_ptr(), image.data_ptr(), acc_map.data_ptr(),
                                                          depth.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                                          torch.cuda.current_stream(points.device).cuda_stream),
                   "lara_point_feats_forward")
        ctx.save_for_backward(points, w2cs, ixts, img_ref, image, acc_map, depth)
        ctx.row_views = int(row_views or 0)
        return out

    @staticmethod
    def backward(ctx, g_out):
        points, w2cs, ixts, img_ref, image, acc_map, depth = ctx.saved_tensors
        n, V, h, w = points.shape[0], img_ref.shape[0], img_ref.shape[2], img_ref.shape[3]
        g_out = g_out.float().contiguous()
        need = ctx.needs_input_grad
        d_points = torch.empty_like(points)
        # the three map gradients as sections of ONE zero-filled buffer (one fill launch instead of three)
        maps = [(image, need[4]), (acc_map, need[5]), (depth, need[6])]
        sizes = [(t.numel() + 3) // 4 * 4 if nd else 0 for t, nd in maps]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=points.device)
        secs, o = [], 0
        for (t, nd), sz in zip(maps, sizes):
            secs.append(flat[o:o + t.numel()].view(t.shape) if nd else None)
            o += sz
        d_image, d_acc, d_depth = secs
        ptr = lambda t: None if t is None else t.data_ptr()
        ws = _workspace(points.device, V, h, w)
        with torch.cuda.device(points.device):
            _check(_lib().lara_point_feats_backward_concat(n, V, ctx.row_views, h, w, points.data_ptr(), w2cs.data_ptr(),
                                                           ixts.data_ptr(), img_ref.data_ptr(), image.data_ptr(), acc_map.data_ptr(),
                                                           depth.data_ptr(), g_out.data_ptr(), d_points.data_ptr(), ptr(d_image),
                                                           ptr(d_acc), ptr(d_depth), ws.data_ptr(),
                                                           torch.cuda.current_stream(points.device).cuda_stream),
                   "lara_point_feats_backward")
        return d_points if need[0] else None, None, None, None, d_image, d_acc, d_depth, None


class _TakeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        out = g.new_zeros((ctx.n,) + tuple(g.shape[1:]))
        out.index_copy_(0, idx, g.contiguous())
        return out, None


class _RowsItem(ctypes.Structure):        # include/lara_pointfeat.h: lara_rows_item
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("width", ctypes.c_int32)]


def _rows_call(idx, srcs, dsts, scatter):
    items = (_RowsItem * len(srcs))()
    for k, (a, b) in enumerate(zip(srcs, dsts)):
        items[k].src, items[k].dst, items[k].width = a.data_ptr(), b.data_ptr(), a[0].numel()
    dev = srcs[0].device
    with torch.cuda.device(dev):
        _check(_lib().lara_take_rows(idx.numel(), idx.data_ptr(), len(srcs), items, int(scatter), torch.cuda.current_stream(dev).cuda_stream),
               "lara_take_rows")


class _TakeRowsMulti(torch.autograd.Function):
    """(idx, x_0 .. x_{k-1}) -> (x_0[idx] .. x_{k-1}[idx]) for UNIQUE indices, one launch per direction for all k tensors
    (`lara_take_rows`, include/lara_pointfeat.h)."""

    @staticmethod
    def forward(ctx, idx, *xs):
        if not xs[0].is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        if idx.dtype != torch.int64 or idx.device != xs[0].device or idx.dim() != 1:
            raise RuntimeError(f"lara_amd: take_rows needs a 1-D int64 index tensor on {xs[0].device}, got {idx.dtype} on {idx.device}")
        idx = idx.contiguous()
        if os.environ.get("LARA_DEBUG_CHECKS") == "1" and idx.numel():
            # the backward scatters dst[idx[r]] = src[r]: duplicates would drop gradient instead of accumulating it (a mask's
            # `nonzero()` is unique by construction); synchronises, so only on request
            if int(idx.min()) < 0 or int(idx.max()) >= min(x.shape[0] for x in xs) or idx.unique().numel() != idx.numel():
                raise RuntimeError("lara_amd: take_rows indices must be unique and in range")
        xs = [x.detach().float().contiguous() for x in xs]
        n = idx.numel()
        outs = [torch.empty((n,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device) for x in xs]
        if n:
            _rows_call(idx, xs, outs, False)
        ctx.save_for_backward(idx)
        ctx.shapes = [tuple(x.shape) for x in xs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        (idx,) = ctx.saved_tensors
        dev = idx.device
        # ONE zero fill for the k gradients (sections of a flat buffer, each starting on a 16-byte boundary), not k
        sizes = [0 if g is None else (math.prod(shape) + 3) // 4 * 4 for g, shape in zip(gs, ctx.shapes)]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        grads, o = [], 0
        for g, shape, sz in zip(gs, ctx.shapes, sizes):
            grads.append(None if g is None else flat[o:o + math.prod(shape)].view(shape))
            o += sz
        live = [(g.float().contiguous(), d) for g, d in zip(gs, grads) if g is not None]
        if live and idx.numel():
            _rows_call(idx, [g for g, _ in live], [d for _, d in live], True)
        return (None,) + tuple(grads)


class _VoxelRowsScenes(torch.autograd.Function):
    """The volume-feature rows of EVERY scene's kept Gaussians (network.py:509, `x.unsqueeze(1).expand(-1, K, -1)[mask.view(-1, K)]`
    per scene) as one autograd node: vol [B, V, C], vox_i = ascending voxel indices of scene i -> rows_i = vol[i][vox_i].  One node
    instead of one per scene because of the BACKWARD: the per-scene nodes each returned a dense [V, C] gradient (a zero fill and an
    `index_add_` per scene), which the unbind in front of them stacked (a 335 MB concatenation) -- here the gradient of `vol` is
    allocated once, zero-filled once, and each scene's rows are summed into its slice by `lara_voxel_rows` (runs of <= K rows per
    voxel added in row order: no atomics, bit-reproducible).  Round 5: 4 x (16 + 74) + 137 us per step; now 64 + 4 x ~30."""

    @staticmethod
    def forward(ctx, vol, *vox):
        if not vol.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        if vol.dim() != 3 or len(vox) != vol.shape[0] or vol.shape[2] % 4:
            raise RuntimeError("lara_amd: voxel rows need vol [B, V, C] (C % 4 == 0) and one index tensor per scene")
        x = vol.detach().float().contiguous()
        dev, C = x.device, x.shape[2]
        outs = []
        with torch.cuda.device(dev):
            for i, v in enumerate(vox):
                if v.dtype != torch.int64 or v.device != dev or v.dim() != 1:
                    raise RuntimeError(f"lara_amd: voxel indices must be 1-D int64 tensors on {dev}")
                v = v.contiguous()
                out = torch.empty((v.numel(), C), dtype=torch.float32, device=dev)
                _check(_lib().lara_voxel_rows(v.numel(), C, v.data_ptr(), x[i].data_ptr(), out.data_ptr(), 0,
                                              torch.cuda.current_stream(dev).cuda_stream), "lara_voxel_rows")
                outs.append(out)
        ctx.save_for_backward(*[v.contiguous() for v in vox])
        ctx.shape = tuple(x.shape)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        vox = ctx.saved_tensors
        B, V, C = ctx.shape
        dev = vox[0].device
        d = torch.zeros(ctx.shape, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            for i, (v, g) in enumerate(zip(vox, gs)):
                if g is None or v.numel() == 0:
                    continue
                g = g.float().contiguous()
                _check(_lib().lara_voxel_rows(v.numel(), C, v.data_ptr(), g.data_ptr(), d[i].data_ptr(), 1,
                                              torch.cuda.current_stream(dev).cuda_stream), "lara_voxel_rows")
        return (d,) + (None,) * len(vox)


def voxel_rows_scenes(vol, vox_list):
    """``[vol[i][vox_list[i]] for i in range(B)]`` (see `_VoxelRowsScenes`): vol [B, V, C], ascending int64 voxel indices per scene."""
    return _VoxelRowsScenes.apply(vol, *vox_list)


def take_rows_multi(xs, idx):
    """``[x[idx] for x in xs]`` for the fine stage's subsets with ONE launch per direction (see `take_rows`; `idx` = unique
    row indices, e.g. ``mask.nonzero().squeeze(-1)``)."""
    return _TakeRowsMulti.apply(idx, *xs)


def take_rows(x, idx):
    """``x[idx]`` for the fine stage's subsets (``_centers[mask]``, ``_opacity_coarse[i][mask]`` ..., network.py:514-524)
    where ``idx`` holds UNIQUE row indices (``mask.nonzero()``): the backward is a plain row copy into zeros.  torch's
    advanced-indexing backward does not know the indices are unique and sorts them first (a dozen rocPRIM merge
    launches per tensor, 2.4 ms per training step at LaRa's sizes)."""
    return _TakeRows.apply(x, idx)


def sample_point_feats(points, w2cs, ixts, img_ref, image, acc_map, depth, row_views=0):
    """points [n,3] -> [V, 8, n]: channels 0-2 the input image, 3-5 the coarse render, 6 acc_map, 7 |depth - z|.
    `row_views` = R > 0: image / acc_map / depth are `Renderer.render_views(concat=True)`'s maps of R >= V views side by
    side ([h, R*w, c]) and the sampler reads the first V of them in place -- no `[:, :V]` slice + stack in front, no
    zero-padded slice gradient behind (network.py:499 stacks the first n_views_sel renders for exactly this call)."""
    return _PointFeats.apply(points, w2cs, ixts, img_ref, image, acc_map, depth, row_views)


def get_point_feats(self, idx, img_ref, renderings, n_views_sel, batch, points, mask):
    """Same arguments and return value as ``Network.get_point_feats`` (network.py:390-411); ``self`` is unused
    (the reference reads only ``self.device`` from it)."""
    points = points[mask]
    src_ixts = batch['tar_ixt'][idx, :n_views_sel].reshape(-1, 3, 3)
    src_w2cs = batch['tar_w2c'][idx, :n_views_sel].reshape(-1, 4, 4)
    feats = sample_point_feats(points, src_w2cs, src_ixts, img_ref, renderings['image'], renderings['acc_map'],
                               renderings['depth'])
    return feats, mask


# ---------------------------------------------------------------------------------------------
# Decoder.forward_fine (network.py:280-284)
# ---------------------------------------------------------------------------------------------
_FD, _NH, _HD, _CD, _NV, _HID, _SH = 80, 8, 10, 8, 4, 64, 12   # the only sizes the kernel is built for (configs/base.yaml)


class _FineDecoder(torch.autograd.Function):
    """(xn [n,80], pf [4,8,n], Wqk [64,80], W1ov [64,64], b1 [64], W2 [12,64], b2 [12]) -> sh [n,12]."""

    @staticmethod
    def forward(ctx, xn, pf, Wqk, W1ov, b1, W2, b2):
        if not xn.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        f = lambda t: t.detach().float().contiguous()
        xn, pf, Wqk, W1ov, b1, W2, b2 = map(f, (xn, pf, Wqk, W1ov, b1, W2, b2))
        n = xn.shape[0]
        if (xn.shape != (n, _FD) or pf.shape != (_NV, _CD, n) or Wqk.shape != (_NH * _CD, _FD) or W1ov.shape != (_HID, _NH * _CD)
                or b1.shape != (_HID,) or W2.shape != (_SH, _HID) or b2.shape != (_SH,)):
            raise RuntimeError("lara_fine_decoder_forward: the kernel is built for xn [n,80], pf [4,8,n], Wqk [64,80], "
                               "W1ov [64,64], b1 [64], W2 [12,64], b2 [12]")
        sh = torch.empty(n, _SH, dtype=torch.float32, device=xn.device)
        with torch.cuda.device(xn.device):
            _check(_lib().lara_fine_decoder_forward(n, xn.data_ptr(), pf.data_ptr(), Wqk.data_ptr(), W1ov.data_ptr(),
                                                    b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), sh.data_ptr(),
                                                    torch.cuda.current_stream(xn.device).cuda_stream),
                   "lara_fine_decoder_forward")
        ctx.save_for_backward(xn, pf, Wqk, W1ov, b1, W2, b2)
        return sh

    @staticmethod
    def backward(ctx, d_sh):
        xn, pf, Wqk, W1ov, b1, W2, b2 = ctx.saved_tensors
        n = xn.shape[0]
        d_sh = d_sh.float().contiguous()
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=xn.device)
        d_xn, d_pf = new(n, _FD), new(_NV, _CD, n)
        U, H, DH, DT = new(n, 64), new(n, _HID), new(n, _HID), new(n, 64)
        with torch.cuda.device(xn.device):
            _check(_lib().lara_fine_decoder_backward(n, xn.data_ptr(), pf.data_ptr(), Wqk.data_ptr(), W1ov.data_ptr(),
                                                     b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), d_sh.data_ptr(),
                                                     d_xn.data_ptr(), d_pf.data_ptr(), U.data_ptr(), H.data_ptr(),
                                                     DH.data_ptr(), DT.data_ptr(),
                                                     torch.cuda.current_stream(xn.device).cuda_stream),
                   "lara_fine_decoder_backward")
            # the five parameter gradients: reductions of the factor arrays over the points, one launch + an ordered sum
            # (include/lara_finedec.h: lara_fine_decoder_wgrad; rounds 2-4: batched BLAS GEMMs + torch reductions, 79 launches)
            lib = _lib()
            dw = new(lib.lara_fine_wgrad_floats())
            ws = torch.empty(max(lib.lara_fine_wgrad_workspace_bytes(n), 16), dtype=torch.uint8, device=xn.device)
            _check(lib.lara_fine_decoder_wgrad(n, xn.data_ptr(), U.data_ptr(), H.data_ptr(), DH.data_ptr(), DT.data_ptr(),
                                               d_sh.data_ptr(), dw.data_ptr(), ws.data_ptr(),
                                               torch.cuda.current_stream(xn.device).cuda_stream), "lara_fine_decoder_wgrad")
        o = 0
        out = []
        for shape in ((_NH * _CD, _FD), (_HID, _NH * _CD), (_HID,), (_SH, _HID), (_SH,)):
            k = 1
            for d in shape:
                k *= d
            out.append(dw[o:o + k].view(shape))
            o += k
        return (d_xn, d_pf, *out)


def _tn_over_points(a, b, chunk=1024):
    """a^T b for a [n,p], b [n,q] with n in the hundreds of thousands and p, q <= 80: one output tile, so a plain GEMM
    call runs on one or two workgroups (measured 1 ms for [64,524288] x [524288,80]).  Split the point axis into
    `chunk`-sized slabs -> a batched GEMM that fills the chip, then add the slabs in fp32 (fixed order: reproducible;
    bf16 operands give bf16 slab products, summed in fp32)."""
    n = a.shape[0]
    m = n // chunk * chunk
    out = None
    if m:
        out = torch.bmm(a[:m].view(-1, chunk, a.shape[1]).transpose(1, 2), b[:m].view(-1, chunk, b.shape[1])).float().sum(0)
    if m < n:
        tail = (a[m:].t() @ b[m:]).float()
        out = tail if out is None else out + tail
    return out


class _FineLayerNorm(torch.autograd.Function):
    """LayerNorm over rows of 80 features, one thread per row (include/lara_finedec.h): x [n,80], gamma, beta, eps."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        if not x.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        x, gamma, beta = x.detach().float().contiguous(), gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        n = x.shape[0]
        xn = torch.empty_like(x)
        stats = torch.empty(n, 2, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _check(_lib().lara_fine_ln_forward(n, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps), xn.data_ptr(),
                                               stats.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                   "lara_fine_ln_forward")
        ctx.save_for_backward(x, gamma, stats)
        return xn

    @staticmethod
    def backward(ctx, d_xn):
        x, gamma, stats = ctx.saved_tensors
        n = x.shape[0]
        d_xn = d_xn.float().contiguous()
        d_x = torch.empty_like(x)
        lib = _lib()
        parts = torch.empty(max(lib.lara_fine_ln_blocks(n), 1), 2 * _FD, dtype=torch.float32, device=x.device)
        if n == 0:
            parts.zero_()
        with torch.cuda.device(x.device):
            _check(lib.lara_fine_ln_backward(n, x.data_ptr(), gamma.data_ptr(), stats.data_ptr(), d_xn.data_ptr(), d_x.data_ptr(),
                                             parts.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream),
                   "lara_fine_ln_backward")
        tot = parts.sum(0)
        return d_x, tot[:_FD], tot[_FD:], None


def _fold_fine_weights(decoder):
    """The folded matrices of include/lara_finedec.h, formed WITH autograd from the reference Decoder's parameters
    (network.py:234-240), in fp32."""
    att = decoder.cross_att
    E = att.embed_dim
    if (E != _FD or att.num_heads != _NH or att.kdim != _CD or att.vdim != _CD or att.in_proj_bias is not None
            or att.out_proj.bias is not None or decoder.mlp_fine[0].out_features != _HID
            or decoder.mlp_fine[2].out_features != _SH):
        raise RuntimeError("lara_amd.fine.forward_fine is built for LaRa's sizes only: embed 80, 8 heads, kdim = vdim = 8, "
                           "no biases in the attention, hidden 64, 12 SH outputs (sh_degree 1)")
    Wq = att.q_proj_weight.float().view(_NH, _HD, E)            # [h, d, i]
    Wk = att.k_proj_weight.float().view(_NH, _HD, _CD)          # [h, d, c]
    Wv = att.v_proj_weight.float().view(_NH, _HD, _CD)
    Wo = att.out_proj.weight.float().view(E, _NH, _HD)          # [o, h, d]
    Wqk = torch.einsum("hdc,hdi->hci", Wk, Wq).reshape(_NH * _CD, E) * (_HD ** -0.5)
    Wov = torch.einsum("ohd,hdc->ohc", Wo, Wv).reshape(E, _NH * _CD)
    W1, b1 = decoder.mlp_fine[0].weight.float(), decoder.mlp_fine[0].bias.float()
    W2, b2 = decoder.mlp_fine[2].weight.float(), decoder.mlp_fine[2].bias.float()
    return Wqk, W1 @ Wov, b1, W2, b2


def fold_fine_weights(decoder):
    """The folded matrices `forward_fine` runs on (include/lara_finedec.h), formed with autograd from the decoder's
    parameters.  A caller that runs several scenes with the same parameters (a training step: network.py:473-525) folds once
    and passes the result as `folded=`: the ~12 small products of the fold, and their backward, run once per step instead
    of once per scene (the scenes' gradients accumulate on the folded tensors first)."""
    return _fold_fine_weights(decoder)


def forward_fine(decoder, volume_feat, point_feats, folded=None):
    """Same arguments and return value as ``Decoder.forward_fine`` (network.py:280-284): volume_feat [n,80],
    point_feats [n,4,8] (the reference passes the sampler's [4,8,n] output through `einsum('lcb->blc')`, a view --
    its storage is used as it is) -> sh [n,1,12] fp32.  `folded`: the result of `fold_fine_weights(decoder)`."""
    if point_feats.dim() != 3 or point_feats.shape[1:] != (_NV, _CD) or volume_feat.shape[-1] != _FD:
        raise RuntimeError("expected volume_feat [n,80] and point_feats [n,4,8]")
    xn = _FineLayerNorm.apply(volume_feat.float(), decoder.norm.weight, decoder.norm.bias, decoder.norm.eps)
    pf = point_feats.float().permute(1, 2, 0)      # [4,8,n]; contiguous() is a no-op on the sampler's own layout
    sh = _FineDecoder.apply(xn, pf, *(_fold_fine_weights(decoder) if folded is None else folded))
    return sh.unsqueeze(1)
