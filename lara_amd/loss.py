"""LaRa's training loss on MI355X (lightning/loss.py:17-60) over the stacked outputs of ``Network.forward``: the pixel terms
(colour MSE, distortion, normal consistency) as one HIP kernel per direction (``lara_loss_terms_forward`` / ``_backward``,
include/lara_loss.h) instead of ~25 elementwise and reduction kernels forward and ~30 backward, plus the reference's
``0.5 * (1 - MS_SSIM)`` term and its psnr / ssim statistics (loss.py:36-45) through ``ms_ssim`` below -- plain torch
operators (the package the reference imports, `pytorch_msssim`, is absent from this image and un-pinned in the reference;
``ms_ssim`` restates its published algorithm and is tested against an independent float64 restatement, tests/test_loss_cpu.py:
parity with the package itself is unpinned).  ``lara_loss(batch, output, it)`` has the signature and return value of
``Losses.forward``: (loss, scalar_stats).  The fused pixel terms have no CPU path: tensors must live on the GPU.

Round 5: on the GPU the MS-SSIM term runs as HIP kernels too (``ms_ssim_fused``, csrc/msssim.hip, include/lara_loss.h: the five
scales' filters, maps, means and their backward on the images where they lie -- 31 ms of torch operators per training step
before); ``ms_ssim`` stays as the torch formulation that ``pipeline.lara_loss`` and the CPU tests use, and as what the kernels
are held to.
"""
from __future__ import annotations

import ctypes

import torch

from .rasterizer import _check, load_library

_configured = False
_weights = {}


class _ImgView(ctypes.Structure):      # include/lara_loss.h: lara_image_view (element strides)
    _fields_ = [("p", ctypes.c_void_p)] + [(n, ctypes.c_int64) for n in ("sN", "sC", "sY", "sV", "sX")] + [("Wv", ctypes.c_int32)]


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
        lib.lara_ms_ssim_workspace_floats.restype = i64
        lib.lara_ms_ssim_workspace_floats.argtypes = [i32, i32, i32, i32]
        lib.lara_ms_ssim_forward.restype = ctypes.c_int
        lib.lara_ms_ssim_forward.argtypes = [i32, i32, i32, i32, ctypes.POINTER(_ImgView), ctypes.POINTER(_ImgView), vp, vp, vp, vp]
        lib.lara_ms_ssim_backward.restype = ctypes.c_int
        lib.lara_ms_ssim_backward.argtypes = [i32, i32, i32, i32, ctypes.POINTER(_ImgView), ctypes.POINTER(_ImgView), vp, vp,
                                              ctypes.POINTER(_ImgView), vp, vp]
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


MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)
_win_cache = {}


def _gauss_window(device, size=11, sigma=1.5):
    key = (str(device), size, sigma)
    if key not in _win_cache:
        c = torch.arange(size, dtype=torch.float32) - size // 2
        g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
        _win_cache[key] = (g / g.sum()).to(device)
        _win_cache[key]._lara_window = (size, sigma)
    return _win_cache[key]


_band_cache = {}


def _band(win, n_in, n_out, device, dtype):
    """[n_in, n_out] matrix B with B[j + i, j] = win[i]: x @ B = the 'valid' correlation of x's last dimension with win."""
    # keyed on what the window IS -- the (size, sigma) `_gauss_window` built it from -- not on its address: `ms_ssim` converts the
    # window to the images' dtype, and for float64 (the finite-difference tests) that is a fresh tensor per call, i.e. a fresh entry
    # per call that kept its window alive.  No device value is read on a hit either way (`float(win[0])` here cost a
    # host synchronisation per call, ~20 per step with MS-SSIM on both images).  A window from elsewhere falls back to its address.
    key = (str(device), dtype, getattr(win, "_lara_window", win.data_ptr()), win.numel(), n_in, n_out)
    if key not in _band_cache:
        k = win.numel()
        B = torch.zeros(n_in, n_out, dtype=dtype, device=device)
        j = torch.arange(n_out, device=device)
        for i in range(k):
            B[j + i, j] = win[i].to(dtype)
        _band_cache[key] = (B, win)
    return _band_cache[key][0]


def _blur(x, win, block=128):
    """Separable 'valid' Gaussian filter of every channel plane of x [N,C,H,W], along H then along W, as two matrix
    products with banded matrices (the filter along H as one [H-k+1, H] band from the left; along W in blocks of `block`
    outputs with a k-1 halo, so that the band stays 138 wide whatever W is).  Same sums as the two depthwise
    convolutions `pytorch_msssim` issues; on this ROCm build a grouped `conv2d` over [60, 512, 4096] planes takes tens of
    milliseconds, the two products a fraction of one."""
    k = win.numel()
    H, W = x.shape[-2:]
    x = torch.matmul(_band(win, H, H - k + 1, x.device, x.dtype).t(), x)            # along H
    n_out = W - k + 1
    nb = -(-n_out // block)
    xp = torch.nn.functional.pad(x, (0, nb * block + k - 1 - W))
    y = torch.matmul(xp.unfold(-1, block + k - 1, block), _band(win, block + k - 1, block, x.device, x.dtype))
    return y.flatten(-2)[..., :n_out]


def _ssim_cs(X, Y, win, data_range=1.0, K=(0.01, 0.03)):
    """Per-(image, channel) means of the SSIM map and of its contrast-structure factor."""
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    N = X.shape[0]
    f = _blur(torch.cat([X, Y, X * X, Y * Y, X * Y]), win)      # the five filters of an SSIM level as one pass
    mu1, mu2, xx, yy, xy = (f[i * N:(i + 1) * N] for i in range(5))
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    cs_map = (2 * (xy - mu12) + C2) / ((xx - mu1_sq) + (yy - mu2_sq) + C2)
    ssim_map = ((2 * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)


def ms_ssim(X, Y, data_range=1.0, win_size=11, win_sigma=1.5, weights=MS_SSIM_WEIGHTS):
    """Multi-scale SSIM of two image batches [N,C,H,W] as `pytorch_msssim.MS_SSIM(data_range=1.0, size_average=True,
    channel=3)` computes it (lightning/loss.py:15, :42): five scales (2 x 2 average pooling in between, odd sides
    padded), 11-tap sigma-1.5 Gaussian windows without padding, K = (0.01, 0.03), the contrast-structure means of the
    first four scales and the SSIM mean of the last clamped at zero, raised to the five published weights, multiplied, and
    averaged over images and channels.  fp32 (the reference calls it under ``autocast(enabled=False)``)."""
    if X.shape != Y.shape or X.dim() != 4:
        raise ValueError("ms_ssim: two image batches [N,C,H,W] of the same shape expected")
    if min(X.shape[-2:]) <= (win_size - 1) * 2 ** 4:
        raise ValueError(f"ms_ssim: the smaller image side must exceed {(win_size - 1) * 2 ** 4} (four 2x downsamplings)")
    if X.dtype != torch.float64 or Y.dtype != torch.float64:       # (float64 stays float64: the tests' finite differences)
        X, Y = X.float(), Y.float()
    base = _gauss_window(X.device, win_size, win_sigma)
    win = base.to(X.dtype)
    if win is not base:
        win._lara_window = base._lara_window      # (the band cache's key: see `_band`)
    w = torch.tensor(weights, dtype=X.dtype, device=X.device)
    vals = []
    for i in range(len(weights)):
        s, cs = _ssim_cs(X, Y, win, data_range)
        if i < len(weights) - 1:
            vals.append(torch.relu(cs))
            pad = [d % 2 for d in X.shape[2:]]
            X = torch.nn.functional.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = torch.nn.functional.avg_pool2d(Y, kernel_size=2, padding=pad)
    vals.append(torch.relu(s))
    return torch.prod(torch.stack(vals, 0) ** w.view(-1, 1, 1), dim=0).mean()


_win_host = {}


def _window_host(size=11, sigma=1.5):
    """The filter taps as a ctypes array (the same fp32 numbers `_gauss_window` puts on the device)."""
    key = (size, sigma)
    if key not in _win_host:
        w = _gauss_window("cpu", size, sigma)
        _win_host[key] = (ctypes.c_float * size)(*[float(v) for v in w])
    return _win_host[key]


class _MsSsimMeans(torch.autograd.Function):
    """(image [B,H,V*W,3] as `Network.forward` stacks it, tar_rgb [B,V,H,W,3]) -> means [5, B*3, 2]: per scale and (image,
    channel) the mean of the SSIM map and of its contrast-structure factor (include/lara_loss.h: lara_ms_ssim_forward).  Both
    images are read where they lie (the reference permutes them to [B,3,H,V*W], loss.py:24-25)."""

    @staticmethod
    def forward(ctx, image, tar):
        if not image.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        image, tar = image.detach().float().contiguous(), tar.detach().float().contiguous()
        B, V, H, W = tar.shape[:4]
        if tar.shape != (B, V, H, W, 3) or image.shape != (B, H, V * W, 3):
            raise RuntimeError("expected tar_rgb [B,V,H,W,3] and image [B,H,V*W,3]")
        lib = _lib()
        nws = int(lib.lara_ms_ssim_workspace_floats(B, 3, H, V * W))
        if nws < 0:
            raise ValueError(f"ms_ssim: the smaller image side must exceed {(11 - 1) * 2 ** 4} (four 2x downsamplings)")
        ws = torch.empty(nws, dtype=torch.float32, device=image.device)
        means = torch.empty(5, B * 3, 2, dtype=torch.float32, device=image.device)
        xv = _ImgView(image.data_ptr(), H * V * W * 3, 1, V * W * 3, W * 3, 3, W)
        yv = _ImgView(tar.data_ptr(), V * H * W * 3, 1, W * 3, H * W * 3, 3, W)
        with torch.cuda.device(image.device):
            _check(lib.lara_ms_ssim_forward(B, 3, H, V * W, ctypes.byref(xv), ctypes.byref(yv), _window_host(), means.data_ptr(),
                                            ws.data_ptr(), torch.cuda.current_stream(image.device).cuda_stream), "lara_ms_ssim_forward")
        ctx.dims = (B, V, H, W)
        ctx.save_for_backward(image, tar, ws)
        return means

    @staticmethod
    def backward(ctx, d_means):
        image, tar, ws = ctx.saved_tensors
        B, V, H, W = ctx.dims
        d_means = d_means.float().contiguous()
        d_image = torch.empty_like(image)
        xv = _ImgView(image.data_ptr(), H * V * W * 3, 1, V * W * 3, W * 3, 3, W)
        yv = _ImgView(tar.data_ptr(), V * H * W * 3, 1, W * 3, H * W * 3, 3, W)
        dv = _ImgView(d_image.data_ptr(), H * V * W * 3, 1, V * W * 3, W * 3, 3, W)
        with torch.cuda.device(image.device):
            _check(_lib().lara_ms_ssim_backward(B, 3, H, V * W, ctypes.byref(xv), ctypes.byref(yv), _window_host(), d_means.data_ptr(),
                                                ctypes.byref(dv), ws.data_ptr(), torch.cuda.current_stream(image.device).cuda_stream),
                   "lara_ms_ssim_backward")
        return d_image, None


def ms_ssim_fused(image, tar_rgb, weights=MS_SSIM_WEIGHTS):
    """MS-SSIM of the stacked render `image` [B,H,V*W,3] against `batch['tar_rgb']` [B,V,H,W,3] -- the value `ms_ssim` gives for
    the two tensors permuted to [B,3,H,V*W] (loss.py:24-25, :42) -- with the filters, maps, means and their backward as HIP
    kernels; the five means per (image, channel) are combined here (relu, published weights, product, mean: [5, 3 B] numbers)."""
    means = _MsSsimMeans.apply(image, tar_rgb)
    vals = torch.cat([means[:4, :, 1], means[4:, :, 0]], 0)        # contrast-structure means of scales 0-3, SSIM mean of the last
    key = (means.device, tuple(weights))
    if key not in _weights:
        _weights[key] = torch.tensor(weights, dtype=torch.float32, device=means.device).view(-1, 1)
    return torch.prod(torch.relu(vals) ** _weights[key], dim=0).mean()


def ms_ssim_terms(batch, output, prexes=("", "_fine"), fused=None):
    """loss.py:36-45 for the images present: ({prex: 0.5 * (1 - MS_SSIM)}, {psnr / ssim statistics}).  `fused`: the HIP kernels
    (default on the GPU) or the torch formulation."""
    B, V, H, W = batch["tar_rgb"].shape[:-1]
    tar = None
    terms, stats = {}, {}
    for prex in prexes:
        if f"image{prex}" not in output or (prex == "_fine" and "acc_map_fine" not in output):
            continue
        if output[f"image{prex}"].is_cuda if fused is None else fused:          # HIP kernels on the tensors where they lie
            with torch.autocast(device_type="cuda", enabled=False):
                val = ms_ssim_fused(output[f"image{prex}"], batch["tar_rgb"])
            terms[prex] = 0.5 * (1 - val)
            stats[f"ssim{prex}"] = val.detach()
            continue
        if tar is None:
            tar = batch["tar_rgb"].permute(0, 2, 1, 3, 4).reshape(B, H, V * W, 3).permute(0, 3, 1, 2).float()
        img = output[f"image{prex}"].permute(0, 3, 1, 2).float()
        with torch.autocast(device_type=img.device.type, enabled=False):
            val = ms_ssim(img, tar)
        terms[prex] = 0.5 * (1 - val)
        stats[f"ssim{prex}"] = val.detach()
    return terms, stats


def lara_loss(batch, output, it=10000, ms_ssim=True):
    """lightning/loss.py:17-60: same arguments and return value as ``Losses.forward``; the pixel terms fused, the MS-SSIM
    term (``ms_ssim=False`` leaves it and the ssim statistics out) through torch operators."""
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
    stats = {"mse": t[0], "psnr": -10.0 * torch.log10(t[0])}                                # loss.py:36-39
    if fine is not None:
        stats["mse_fine"], stats["psnr_fine"] = t[1], -10.0 * torch.log10(t[1])
    if reg:
        stats["distortion"], stats["normal"] = t[2], t[3]
    if ms_ssim:
        extra, st = ms_ssim_terms(batch, output, ("", "_fine") if fine is not None else ("",))
        for v in extra.values():
            loss = loss + v                                                                  # loss.py:45
        stats.update(st)
    return loss, stats
