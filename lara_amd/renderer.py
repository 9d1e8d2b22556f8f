"""Opt-in replacement for the reference's ``Renderer`` (lightning/renderer_2dgs.py:91-268): the same
``render_img(cam, rays, centers, shs, opacity, scales, rotations, device, cov3D_precomp=None, prex='',
depth_ratio=0.0)`` returning the same dictionary, with

* the rasteriser call going to the HIP path (``lara_amd.GaussianRasterizer``), and
* everything after it (renderer_2dgs.py:220-268: clamp, expected / median depth, normals to world space,
  ``depth_to_normal``, the channel-last permutes) done by ONE HIP kernel per direction
  (``lara_surface_maps_forward`` / ``_backward``, include/lara_surface.h) instead of ~15 + ~25 launch-bound
  torch kernels per view;
* the three activations (sigmoid / exp / normalize, renderer_2dgs.py:183-189) computed once per set of
  Gaussians instead of once per view: the reference's loop (network.py:487-497) calls ``render_img`` for each of
  a scene's 8 views with the same parameter tensors.

SURVEY.md section 8f row 2.  Use: ``net.gs_render = lara_amd.renderer.Renderer(sh_degree=..., white_background=...)``;
the unchanged-call-signature path (the reference's own ``Renderer`` on top of the drop-in rasteriser) keeps
working.  No CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes
import math

import torch
from torch import nn

from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, _check, load_library,
                         rasterize_gaussians_views)

_configured = False


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
        lib.lara_surface_maps_forward.restype = ctypes.c_int
        lib.lara_surface_maps_forward.argtypes = [i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]
        lib.lara_surface_maps_backward.restype = ctypes.c_int
        lib.lara_surface_maps_backward.argtypes = [i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        lib.lara_activate_gaussians_forward.restype = ctypes.c_int
        lib.lara_activate_gaussians_forward.argtypes = [ctypes.c_int64, vp, vp, vp, vp, vp, vp, vp]
        lib.lara_activate_gaussians_backward.restype = ctypes.c_int
        lib.lara_activate_gaussians_backward.argtypes = [ctypes.c_int64] + [vp] * 10
        lib.lara_surface_maps_forward_views.restype = ctypes.c_int
        lib.lara_surface_maps_forward_views.argtypes = [i32, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp]
        lib.lara_surface_maps_backward_views.restype = ctypes.c_int
        lib.lara_surface_maps_backward_views.argtypes = [i32, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        _configured = True
    return lib


class _SurfaceMaps(torch.autograd.Function):
    """(color [3,H,W], allmap [7,H,W], rays [H,W,6], rot [3,3], depth_ratio) -> image [H,W,3], depth [H,W,1],
    acc_map [H,W], rend_normal [H,W,3], depth_normal [H,W,3], rend_dist [H,W]"""

    @staticmethod
    def forward(ctx, color, allmap, rays, rot, depth_ratio):
        if not color.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        color, allmap = color.float().contiguous(), allmap.float().contiguous()
        rays, rot = rays.detach().float().contiguous(), rot.detach().float().contiguous()
        H, W = color.shape[1], color.shape[2]
        if allmap.shape != (7, H, W) or rays.shape != (H, W, 6) or rot.shape != (3, 3):
            raise RuntimeError("expected color [3,H,W], allmap [7,H,W], rays [H,W,6], rot [3,3]")
        o = dict(dtype=torch.float32, device=color.device)
        image, depth, acc = torch.empty(H, W, 3, **o), torch.empty(H, W, 1, **o), torch.empty(H, W, **o)
        rnorm, dnorm, rdist = torch.empty(H, W, 3, **o), torch.empty(H, W, 3, **o), torch.empty(H, W, **o)
        with torch.cuda.device(color.device):
            _check(_lib().lara_surface_maps_forward(H, W, color.data_ptr(), allmap.data_ptr(), rays.data_ptr(), rot.data_ptr(),
                                                    float(depth_ratio), image.data_ptr(), depth.data_ptr(), acc.data_ptr(),
                                                    rnorm.data_ptr(), dnorm.data_ptr(), rdist.data_ptr(),
                                                    torch.cuda.current_stream(color.device).cuda_stream),
                   "lara_surface_maps_forward")
        ctx.save_for_backward(color, allmap, rays, rot)
        ctx.depth_ratio = float(depth_ratio)
        ctx.set_materialize_grads(False)     # an output the loss does not read keeps a None gradient (kernel: NULL = zero)
        return image, depth, acc, rnorm, dnorm, rdist

    @staticmethod
    def backward(ctx, g_image, g_depth, g_acc, g_rnorm, g_dnorm, g_rdist):
        color, allmap, rays, rot = ctx.saved_tensors
        H, W = color.shape[1], color.shape[2]
        gs = [None if g is None else g.float().contiguous() for g in (g_image, g_depth, g_acc, g_rnorm, g_dnorm, g_rdist)]
        d_color = torch.empty_like(color)
        d_allmap = None if all(g is None for g in gs[1:]) else torch.empty_like(allmap)     # (see _SurfaceMapsViews.backward)
        with torch.cuda.device(color.device):
            _check(_lib().lara_surface_maps_backward(H, W, color.data_ptr(), allmap.data_ptr(), rays.data_ptr(), rot.data_ptr(),
                                                     ctx.depth_ratio, *[None if g is None else g.data_ptr() for g in gs],
                                                     d_color.data_ptr(), None if d_allmap is None else d_allmap.data_ptr(),
                                                     torch.cuda.current_stream(color.device).cuda_stream),
                   "lara_surface_maps_backward")
        return d_color, d_allmap, None, None, None


class _Activate(torch.autograd.Function):
    """(opacity [P,1], scales [P,2], rotations [P,4]) -> (sigmoid, exp, F.normalize) of them (renderer_2dgs.py:181-189) in one
    launch per direction instead of ~6 torch kernels forward and ~17 backward per set of Gaussians."""

    @staticmethod
    def forward(ctx, opacity, scales, rotations):
        if not opacity.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        P = opacity.shape[0]
        if opacity.shape != (P, 1) or scales.shape != (P, 2) or rotations.shape != (P, 4):
            raise RuntimeError("expected opacity [P,1], scales [P,2], rotations [P,4]")
        o, s, r = (t.detach().float().contiguous() for t in (opacity, scales, rotations))
        oa, sa, ra = torch.empty_like(o), torch.empty_like(s), torch.empty_like(r)
        with torch.cuda.device(o.device):
            _check(_lib().lara_activate_gaussians_forward(P, o.data_ptr(), s.data_ptr(), r.data_ptr(), oa.data_ptr(), sa.data_ptr(),
                                                          ra.data_ptr(), torch.cuda.current_stream(o.device).cuda_stream),
                   "lara_activate_gaussians_forward")
        ctx.save_for_backward(oa, sa, r)
        ctx.set_materialize_grads(False)
        return oa, sa, ra

    @staticmethod
    def backward(ctx, g_o, g_s, g_r):
        oa, sa, r = ctx.saved_tensors
        P = oa.shape[0]
        gs = [None if g is None else g.float().contiguous() for g in (g_o, g_s, g_r)]
        need = ctx.needs_input_grad
        d = [torch.empty_like(t) if n else None for t, n in zip((oa, sa, r), need)]
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(oa.device):
            _check(_lib().lara_activate_gaussians_backward(P, oa.data_ptr(), sa.data_ptr(), r.data_ptr(), ptr(gs[0]), ptr(gs[1]),
                                                           ptr(gs[2]), ptr(d[0]), ptr(d[1]), ptr(d[2]),
                                                           torch.cuda.current_stream(oa.device).cuda_stream),
                   "lara_activate_gaussians_backward")
        return tuple(d)


def activate_gaussians(opacity, scales, rotations):
    """sigmoid(opacity), exp(scales), F.normalize(rotations): the fused form of renderer_2dgs.py:181-189."""
    return _Activate.apply(opacity, scales, rotations)


class _SurfaceMapsViews(torch.autograd.Function):
    """(color [n,3,H,W], allmap [n,7,H,W], rays [n,H,W,6], rots [n,3,3], depth_ratio) -> the six maps of all n views side by
    side, each [H, n*W, C] -- the per-scene concatenation of lightning/network.py:527 written directly by the kernel."""

    @staticmethod
    def forward(ctx, color, allmap, rays, rots, depth_ratio, into=None):
        if not color.is_cuda:
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        color, allmap = color.float().contiguous(), allmap.float().contiguous()
        rays, rots = rays.detach().float().contiguous(), rots.detach().float().contiguous()
        n, H, W = color.shape[0], color.shape[2], color.shape[3]
        if color.shape != (n, 3, H, W) or allmap.shape != (n, 7, H, W) or rays.shape != (n, H, W, 6) or rots.shape != (n, 3, 3):
            raise RuntimeError("expected color [n,3,H,W], allmap [n,7,H,W], rays [n,H,W,6], rots [n,3,3]")
        o = dict(dtype=torch.float32, device=color.device)
        shapes = ((H, n * W, 3), (H, n * W, 1), (H, n * W), (H, n * W, 3), (H, n * W, 3), (H, n * W))
        if into is None:
            image, depth, acc, rnorm, dnorm, rdist = (torch.empty(*sh, **o) for sh in shapes)
        else:
            # `into`: six caller-owned buffers the kernel writes instead -- a scene's slices of the batch's [B, H, n*W, C] outputs
            # (lara_amd.pipeline: the stack of network.py:529 then copies nothing).  They are not inputs of the graph: the
            # outputs are fresh tensors over the same memory.
            for t, sh in zip(into, shapes):
                if tuple(t.shape) != sh or t.dtype != torch.float32 or not t.is_contiguous() or t.device != color.device or t.requires_grad:
                    raise RuntimeError("lara_amd: `into` buffers must be contiguous fp32 tensors of the maps' shapes that need no gradient")
            image, depth, acc, rnorm, dnorm, rdist = (t.detach() for t in into)
        with torch.cuda.device(color.device):
            _check(_lib().lara_surface_maps_forward_views(n, H, W, color.data_ptr(), allmap.data_ptr(), rays.data_ptr(), rots.data_ptr(),
                                                          float(depth_ratio), image.data_ptr(), depth.data_ptr(), acc.data_ptr(),
                                                          rnorm.data_ptr(), dnorm.data_ptr(), rdist.data_ptr(),
                                                          torch.cuda.current_stream(color.device).cuda_stream),
                   "lara_surface_maps_forward_views")
        ctx.save_for_backward(color, allmap, rays, rots)
        ctx.depth_ratio = float(depth_ratio)
        ctx.set_materialize_grads(False)     # an output the loss does not read keeps a None gradient (kernel: NULL = zero)
        return image, depth, acc, rnorm, dnorm, rdist

    @staticmethod
    def backward(ctx, g_image, g_depth, g_acc, g_rnorm, g_dnorm, g_rdist):
        color, allmap, rays, rots = ctx.saved_tensors
        n, H, W = color.shape[0], color.shape[2], color.shape[3]
        gs = [None if g is None else g.float().contiguous() for g in (g_image, g_depth, g_acc, g_rnorm, g_dnorm, g_rdist)]
        # no gradient on any of the five maps (LaRa's fine pass: the loss reads its image only, lightning/loss.py:35-47): the
        # seven planes of d_allmap would be zeros -- None instead, which the rasteriser's backward takes as "colour only"
        d_color = torch.empty_like(color)
        d_allmap = None if all(g is None for g in gs[1:]) else torch.empty_like(allmap)
        with torch.cuda.device(color.device):
            _check(_lib().lara_surface_maps_backward_views(n, H, W, color.data_ptr(), allmap.data_ptr(), rays.data_ptr(), rots.data_ptr(),
                                                           ctx.depth_ratio, *[None if g is None else g.data_ptr() for g in gs],
                                                           d_color.data_ptr(), None if d_allmap is None else d_allmap.data_ptr(),
                                                           torch.cuda.current_stream(color.device).cuda_stream),
                   "lara_surface_maps_backward_views")
        return d_color, d_allmap, None, None, None, None


def surface_maps_views(color, allmap, rays, rots, depth_ratio=0.0, into=None):
    """The fused post-processing of n views at once, outputs concatenated along the width (see `_SurfaceMapsViews`)."""
    return _SurfaceMapsViews.apply(color, allmap, rays, rots, depth_ratio, into)


MAP_KEYS = ("image", "depth", "acc_map", "rend_normal", "depth_normal", "rend_dist")      # the order of the six maps everywhere


class _AssembleScenes(torch.autograd.Function):
    """`torch.stack(parts)` (network.py:529) when the parts already LIVE in `buf[i]` -- each scene's post-processing wrote its
    slice of the batch buffer (`surface_maps_views(..., into=...)`): no copy forward, slices of the gradient backward."""

    @staticmethod
    def forward(ctx, buf, *parts):
        for i, p in enumerate(parts):
            if p.data_ptr() != buf[i].data_ptr() or p.shape != buf[i].shape:
                raise RuntimeError("lara_amd: a scene's map does not live in its slice of the batch buffer")
        return buf.detach()

    @staticmethod
    def backward(ctx, g):
        return (None,) + tuple(g.unbind(0))


def batch_map_buffers(B, H, VW, device, prexes=("",)):
    """The batch's output dictionary as uninitialised buffers {key+prex: [B, H, V*W, C]} for `render_views(..., into=)`."""
    o = dict(dtype=torch.float32, device=device)
    chans = {"image": (3,), "depth": (1,), "acc_map": (), "rend_normal": (3,), "depth_normal": (3,), "rend_dist": ()}
    return {k + prex: torch.empty((B, H, VW) + chans[k], **o) for prex in prexes for k in MAP_KEYS}


def surface_maps(color, allmap, rays, rot, depth_ratio=0.0):
    """The fused post-processing alone (see the module docstring); differentiable w.r.t. color and allmap."""
    return _SurfaceMaps.apply(color, allmap, rays, rot, depth_ratio)


class Renderer(nn.Module):
    """Same constructor and ``render_img`` as the reference ``Renderer`` (renderer_2dgs.py:91-268)."""

    def __init__(self, sh_degree=3, white_background=True, radius=1):
        super().__init__()
        self.sh_degree, self.white_background, self.radius = sh_degree, white_background, radius
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize
        self.bg_color = torch.tensor([1, 1, 1] if white_background else [0, 0, 0], dtype=torch.float32)
        self._act_key, self._act_val = None, None

    def set_bg_color(self, bg):
        self.bg_color = bg

    def _settings(self, viewpoint_camera, scaling_modifier=1.0, device="cuda", bg=None):
        return GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=(self.bg_color if bg is None else bg).to(device), scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
            sh_degree=self.sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)

    def set_rasterizer(self, viewpoint_camera, scaling_modifier=1.0, device="cuda"):   # renderer_2dgs.py:119-139
        return GaussianRasterizer(raster_settings=self._settings(viewpoint_camera, scaling_modifier, device))

    def get_opacity(self, _opacity):
        return self.opacity_activation(_opacity)

    def get_scaling(self, _scaling):
        return self.scaling_activation(_scaling)

    def get_rotation(self, _rotation):
        return self.rotation_activation(_rotation)

    def _zero_means2D(self, centers):
        z = getattr(self, "_zeros2d", None)
        if z is None or z.shape != centers.shape or z.device != centers.device or z.dtype != centers.dtype:
            z = torch.zeros_like(centers, requires_grad=False)
            self._zeros2d = z
        return z

    @staticmethod
    def _tensor_key(t):
        # what the tensor IS, not which Python object wraps it: the reference's loop (network.py:496, :524) passes
        # `_opacity_coarse[i]` etc., a fresh view object on every call over the same storage
        if t is None:
            return None
        # identity of the autograd variable the view was taken from (`_opacity_coarse` for `_opacity_coarse[i]`): two
        # tensors over the same bytes can belong to different graphs (e.g. `x.detach()[None]` taken twice), and a
        # cached activation must never be replayed into another graph
        base = t._base if t._is_view() else t
        return (id(base), t.data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype, t.device,
                t._version, t.requires_grad)

    def _activated(self, opacity, scales, rotations):
        """The activated tensors of the last call are reused while the caller passes tensors over the same
        memory (views of the same base tensor, same address, layout, dtype, version counter, requires_grad) in the same autograd mode: the 8
        views of a scene then share one sigmoid / exp / normalize node although `_opacity_coarse[i]` is a new
        Python object per call.  The cache holds the inputs it was computed from, so their storage cannot be
        freed and handed to another tensor while the entry lives (an address match is a real match).  Not covered:
        writes through `.data` (they do not bump the version counter) -- call `invalidate_activations()` after
        such a write.  The fine pass's `x[mask]` arguments are new storage per call and are recomputed per view,
        as in the reference; `render_views` shares them explicitly.  (Call `loss.backward()` after all views of
        those tensors, as network.py does: a backward in between frees the shared nodes' graph.)"""
        key = tuple(self._tensor_key(t) for t in (opacity, scales, rotations)) + (torch.is_grad_enabled(),)
        if key != self._act_key:
            fused = (scales is not None and rotations is not None and opacity.is_cuda and opacity.dim() == 2
                     and self.scaling_activation is torch.exp and self.opacity_activation is torch.sigmoid
                     and self.rotation_activation is torch.nn.functional.normalize)
            if fused:       # one launch per direction for the three of them
                self._act_val = activate_gaussians(opacity, scales, rotations) + ((opacity, scales, rotations),)
            else:
                self._act_val = (self.get_opacity(opacity), None if scales is None else self.get_scaling(scales),
                                 None if rotations is None else self.get_rotation(rotations), (opacity, scales, rotations))
            self._act_key = key
        return self._act_val[:3]

    def invalidate_activations(self):
        """Drop the cached activations (and the references to the tensors they were computed from)."""
        self._act_key, self._act_val = None, None

    def render_img(self, cam, rays, centers, shs, opacity, scales, rotations, device, cov3D_precomp=None, prex='',
                   depth_ratio=0.0):
        rasterizer = self.set_rasterizer(cam, device=device)
        opacity, scales, rotations = self._activated(opacity, scales, rotations)
        # the reference builds a fresh zero tensor that requires grad per view (renderer_2dgs.py:194-205) to read
        # screen-space gradients it no longer returns (:264-266 are commented out): one shared constant does here
        screenspace_points = self._zero_means2D(centers)
        rendered_image, radii, allmap = rasterizer(means3D=centers, means2D=screenspace_points, shs=shs, opacities=opacity,
                                                   scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
        if rays is None:
            return rendered_image.clamp(0, 1)
        rot = cam.world_view_transform[:3, :3].T          # renderer_2dgs.py:231
        image, depth, acc, rnorm, dnorm, rdist = surface_maps(rendered_image, allmap, rays, rot, depth_ratio)
        return {f"image{prex}": image, f"depth{prex}": depth, f"acc_map{prex}": acc, f"rend_normal{prex}": rnorm,
                f"depth_normal{prex}": dnorm, f"rend_dist{prex}": rdist}

    def render_views(self, cams, rays, centers, shs, opacity, scales, rotations, device, bg_colors=None,
                     cov3D_precomp=None, prex='', depth_ratio=0.0, concat=False, into=None, raster_out=None, subset_of=None):
        """All views of a scene in ONE rasteriser call (one autograd node, per-camera state carved from one
        allocation, gradients summed over the views inside the library): what the reference's inner loop
        (lightning/network.py:486-497 coarse, :516-525 fine) does with one ``render_img`` per view.  ``cams`` is a
        sequence of cameras, ``rays`` the matching sequence of ray maps (or a stacked tensor), ``bg_colors`` the
        per-view backgrounds the loop passes through ``set_bg_color`` (default: this renderer's colour).  Returns the
        list of per-view dictionaries ``render_img`` would have returned; with ``concat=True`` ONE dictionary whose maps
        are the views' maps side by side, [H, n*W, C] -- what network.py:527 builds with ``torch.cat(..., dim=1)`` --
        written by one post-processing launch for all views -- into the caller's buffers when ``into`` ({key+prex: [H, n*W, C]})
        is given.  ``raster_out`` (a list) receives the rasteriser's own (color, radii, allmap); ``subset_of = (color, idx)`` names
        an earlier call's colour output and this call's rows in it (`rasterize_gaussians_views`: the fine pass filters the coarse
        pass's lists instead of binning again)."""
        n = len(cams)
        if torch.is_tensor(bg_colors) and bg_colors.dim() == 2 and not (
                bg_colors.dtype == torch.float32 and bg_colors.stride() == (4, 1) and bg_colors.data_ptr() % 16 == 0):
            bg_colors = torch.nn.functional.pad(bg_colors.float(), (0, 1))[:, :3]     # rows on 16-byte boundaries (see cameras.make_cameras)
        bgs = [None] * n if bg_colors is None else list(bg_colors)
        settings = [self._settings(cam, device=device, bg=bg) for cam, bg in zip(cams, bgs)]
        opacity, scales, rotations = self._activated(opacity, scales, rotations)
        color, radii, allmap = rasterize_gaussians_views(settings, centers, self._zero_means2D(centers), opacity, shs=shs,
                                                         scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                                                         subset_of=subset_of)
        if raster_out is not None:
            raster_out.append((color, radii, allmap))
        if concat and rays is not None:
            rots = torch.stack([cam.world_view_transform[:3, :3] for cam in cams]).transpose(1, 2)       # renderer_2dgs.py:231
            rays_t = rays if torch.is_tensor(rays) else torch.stack(list(rays))
            image, depth, acc, rnorm, dnorm, rdist = surface_maps_views(
                color, allmap, rays_t, rots, depth_ratio, None if into is None else tuple(into[k + prex] for k in MAP_KEYS))
            return {f"image{prex}": image, f"depth{prex}": depth, f"acc_map{prex}": acc, f"rend_normal{prex}": rnorm,
                    f"depth_normal{prex}": dnorm, f"rend_dist{prex}": rdist}
        out = []
        for i, cam in enumerate(cams):
            if rays is None:
                out.append(color[i].clamp(0, 1))
                continue
            rot = cam.world_view_transform[:3, :3].T
            image, depth, acc, rnorm, dnorm, rdist = surface_maps(color[i], allmap[i], rays[i], rot, depth_ratio)
            out.append({f"image{prex}": image, f"depth{prex}": depth, f"acc_map{prex}": acc, f"rend_normal{prex}": rnorm,
                        f"depth_normal{prex}": dnorm, f"rend_dist{prex}": rdist})
        return out
