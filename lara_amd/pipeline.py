"""The whole data-dependent LaRa step on the hot path, opt-in (rows A1-A3, R0-R11, section 8f-2
and 8f-4 composed the way ``Network.forward`` composes them, lightning/network.py:455-532):

    image-feature volume  --VolTransformer-->  volume features [B,64,64,64,80]            (network.py:455)
      --Decoder.forward_coarse (plain torch: outside section 8a)-->  offsets, SHs, scales, rotations, opacities   (:458)
      --get_offseted_pt-->  centres; masks = sigmoid(opacity) > 0.005                      (:461-465)
      per scene:  8 coarse views (`Renderer.render_views`)                                 (:486-497)
                  -> `_check_mask` -> `get_point_feats` on the input views                 (:504-505)
                  -> `Decoder.forward_fine` -> refined SHs                                 (:509-510)
                  -> 8 fine views over the masked subset                                   (:516-525)
      -> the reference's output dictionary: per key [B, H, V*W, C]                         (:527-529)

Nothing here is a new kernel: every stage is one of the individually tested operators of this package
(``encoder_train.VolTransformer``, ``renderer.Renderer.render_views``, ``fine.sample_point_feats``,
``fine.forward_fine``, ``fine.take_rows``); tests/test_pipeline_gpu.py holds the composition to the composition of those
operators called one by one.  What the module adds is the ORDER the work reaches the device in:

* scenes are independent after the decoder, so scene i runs on HIP stream i % n_streams (the composite kernels end in a
  tail of a few heavy tiles; the other scene's kernels fill it);
* the only host synchronisations are the ones the reference has too -- the size of each scene's masked subset
  (``x[mask]``, network.py:514-524) -- and they are taken AFTER the coarse views of all scenes are enqueued, on the
  stream that holds only the encoder + decoder, so the device never waits for the host;
* the masks' random thinning (``_check_mask``, network.py:381-388) is computed without a host branch.

``lara_loss`` is lightning/loss.py:17-60 minus MS-SSIM (pytorch_msssim is not installed here): the consumer that turns
the output dictionary into one backward pass through everything above.

No CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import contextlib
import math

import os

import torch
from torch import nn

from . import cameras, coarse
from .fine import _tn_over_points, fold_fine_weights, forward_fine, sample_point_feats, take_rows, take_rows_multi, voxel_rows_scenes
from .renderer import Renderer, _AssembleScenes, batch_map_buffers


class CoarseFineDecoder(nn.Module):
    """Parameter container with the attribute names of the reference ``Decoder`` (network.py:215-251: ``mlp_coarse``,
    ``norm``, ``cross_att``, ``mlp_fine``) for benchmarks and tests -- a real deployment passes the reference's own
    ``Decoder`` instance, which has the same attributes.  Initialised as the reference does (xavier weights, zero
    biases in both MLPs)."""

    def __init__(self, in_dim=80, sh_dim=12, scaling_dim=2, rotation_dim=4, opacity_dim=1, K=2):
        super().__init__()
        self.K, self.sh_dim, self.opacity_dim, self.scaling_dim, self.rotation_dim = K, sh_dim, opacity_dim, scaling_dim, rotation_dim
        self.out_dim = 3 + sh_dim + opacity_dim + scaling_dim + rotation_dim
        self.mlp_coarse = nn.Sequential(nn.Linear(in_dim, in_dim), nn.ReLU(), nn.Linear(in_dim, in_dim), nn.ReLU(),
                                        nn.Linear(in_dim, self.out_dim * K))
        self.norm = nn.LayerNorm(in_dim)
        self.cross_att = nn.MultiheadAttention(embed_dim=in_dim, num_heads=8, kdim=8, vdim=8, dropout=0.0, bias=False,
                                               batch_first=True)
        self.mlp_fine = nn.Sequential(nn.Linear(in_dim, 64), nn.ReLU(), nn.Linear(64, sh_dim))
        for seq in (self.mlp_coarse, self.mlp_fine):
            for layer in seq:
                if isinstance(layer, nn.Linear):
                    nn.init.xavier_uniform_(layer.weight)
                    nn.init.zeros_(layer.bias)


class _LinearBf16(torch.autograd.Function):
    """`nn.Linear` as it runs under bf16 autocast (operands and result bf16, fp32 accumulation inside the GEMM), with
    ONE difference in the backward: the weight gradient dY^T X has a [out, in] <= [80, 80] result and a reduction over
    the ~10^6 voxel rows -- as one GEMM call that is a single output tile on one or two workgroups (4.7 ms per step for
    the three layers of `mlp_coarse`); here the rows are split into slabs -> one batched GEMM over the whole chip + an
    fp32 sum of the slab products."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        xb = x.reshape(-1, x.shape[-1]).to(torch.bfloat16)
        wb = weight.to(torch.bfloat16)
        y = torch.nn.functional.linear(xb, wb, None if bias is None else bias.to(torch.bfloat16))
        ctx.save_for_backward(xb, wb)
        ctx.x_shape, ctx.x_dtype, ctx.has_bias = x.shape, x.dtype, bias is not None
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        dy = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16).contiguous()
        dx = (dy @ wb).view(ctx.x_shape).to(ctx.x_dtype) if ctx.needs_input_grad[0] else None
        dw = _tn_over_points(dy, xb)
        db = dy.float().sum(0) if ctx.has_bias else None
        return dx, dw, db


def _mlp_bf16(seq, x):
    for m in seq:
        x = _LinearBf16.apply(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
    return x


def decode_coarse(decoder, feats, opacity_shift, scaling_shift, autocast=True):
    """``Decoder.forward_coarse`` (network.py:259-278) as plain torch (outside SURVEY.md section 8a): feats [B,...,80] ->
    (offset [B,P,3] in (-1,1), sh [B,P,sh_dim/3,3], scaling [B,P,2], rotation [B,P,4], opacity [B,P,1]), P = voxels * K.
    `autocast`: the reference trains under bf16-mixed (train_lightning.py:74), so its three Linear layers run in bf16
    and the result is cast back with `.float()`; `_LinearBf16` is that arithmetic with a weight gradient that fills
    the chip."""
    par = (_mlp_bf16(decoder.mlp_coarse, feats) if autocast else decoder.mlp_coarse(feats)).float()
    K = decoder.K
    par = par.view(*par.shape[:-1], K, -1)
    offset, sh, opacity, scaling, rotation = torch.split(
        par, [3, decoder.sh_dim, decoder.opacity_dim, decoder.scaling_dim, decoder.rotation_dim], dim=-1)
    B = par.shape[0]
    return ((torch.sigmoid(offset) * 2 - 1.0).reshape(B, -1, 3), sh.reshape(B, -1, decoder.sh_dim // 3, 3),
            (scaling + scaling_shift).reshape(B, -1, decoder.scaling_dim), rotation.reshape(B, -1, decoder.rotation_dim),
            (opacity + opacity_shift).reshape(B, -1, decoder.opacity_dim))


def check_mask(mask, training, generator=None):
    """``Network._check_mask`` (network.py:381-388): a mask keeping < 0.1 % of the Gaussians gains random ones
    (``mask + rand > 0.8``), one keeping > 50 % in training loses every second one at random (``mask * rand > 0.5``).
    Same arithmetic, but the case is selected on the device (`torch.where` on the ratio) instead of by a Python branch
    on a device scalar, which would stall the host once per scene."""
    m = mask.to(torch.float32)
    ratio = m.mean()
    rnd = torch.rand(mask.shape, device=mask.device, generator=generator)
    sparse = (m + rnd) > 0.8
    out = torch.where(ratio < 1e-3, sparse, mask)
    if training:
        out = torch.where(ratio > 0.5, (m * rnd) > 0.5, out)
    return out


def hand_over(obj, stream):
    """Tell the caching allocator that `stream` reads the tensor(s) in `obj` (a tensor, a Camera-like object with tensor
    attributes, or a nested list / tuple / dict of those) although another stream allocated them.

    The allocator hands a freed block back to the stream that ALLOCATED it at once, on the host's clock.  A tensor made
    on the caller's stream and read by a scene stream -- the decoder's outputs, the subset indices, the cameras -- is
    freed when its last Python / autograd reference goes, typically in the middle of the backward, while the scene
    stream's kernels that read it are still queued; the next allocation on the caller's stream may get the same bytes
    and overwrite them under the reader (that is what hung the device in round 3, with subset
    INDICES as the overwritten tensor).  `record_stream` makes the free wait for the reader."""
    if stream is None or obj is None:
        return
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            hand_over(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            hand_over(v, stream)
    elif hasattr(obj, "__dict__"):
        for v in vars(obj).values():
            if torch.is_tensor(v):
                hand_over(v, stream)


class _TakeVoxelRows(torch.autograd.Function):
    """``x.unsqueeze(1).expand(-1, K, -1)[mask.view(-1, K)]`` (network.py:509): row idx // K of x for every kept
    Gaussian.  Backward: at most K rows add into one voxel; for K <= 2 the sum of two floats does not depend on the
    order the atomics land in, so it is bit-reproducible."""

    @staticmethod
    def forward(ctx, x, vox):
        ctx.save_for_backward(vox)
        ctx.n = x.shape[0]
        return x.index_select(0, vox)

    @staticmethod
    def backward(ctx, g):
        (vox,) = ctx.saved_tensors
        out = g.new_zeros((ctx.n,) + tuple(g.shape[1:]))
        out.index_add_(0, vox, g.contiguous())
        return out, None


class LaRaPipeline(nn.Module):
    """``vol_decoder``: a ``VolTransformer`` (``lara_amd.encoder_train`` or the reference's); ``decoder``: the
    reference's ``Decoder`` or ``CoarseFineDecoder``.  Constants as ``Network.__init__`` derives them from
    configs/base.yaml (network.py:306-343): grid_reso = vol_embedding_reso, n_offset_groups, K, sh_degree."""

    def __init__(self, vol_decoder, decoder, grid_reso=32, n_offset_groups=32, sh_degree=1, white_bkgd=True, n_views=4,
                 scene_size=0.5, n_streams=2):
        super().__init__()
        self.vol_decoder, self.decoder = vol_decoder, decoder
        self.K = decoder.K
        self.n_views, self.scene_size, self.n_offset_groups = n_views, scene_size, n_offset_groups
        self.gs_render = Renderer(sh_degree=sh_degree, white_background=white_bkgd, radius=1)
        self.opacity_shift = -2.1792                                           # network.py:339-342
        self.voxel_size = 2.0 / (grid_reso * 2)
        self.scaling_shift = math.log(0.5 * self.voxel_size / 3.0)
        r = grid_reso * 2                                                      # network.py:345-349, :325-326
        a = torch.arange(r)
        grid = (torch.stack(torch.meshgrid(a, a, a, indexing="ij"), dim=-1) + 0.5) / r * 2 - 1
        self.register_buffer("group_centers", (grid * scene_size).reshape(1, -1, 3).float())
        self.n_streams = n_streams
        self._streams = []
        self.fused_coarse = True          # False: `decode_coarse` (torch operators) instead of lara_amd.coarse
        self.fine_mask = "reference"      # "reference": _check_mask as the reference applies it; "plain": opacity > 0.005 only
        self.fine_reuses_coarse_lists = True    # the fine pass's per-tile lists as a filter of the coarse pass's (rasterize_gaussians_views: subset_of)
        self.stage_events = None          # set to a list to collect (stage, start event, end event) per call
        self.opacity_bias = None          # benchmarks only: a [1,P,1] logit bias (see `trained_like_opacity_bias`)

    # -- stages --------------------------------------------------------------------------------------------------
    def gaussians(self, feat_vol):
        """network.py:455-465 -> dict of [B,P,...] tensors + the volume features flattened to [B, voxels, 80]."""
        return self.gaussians_from_volume(self.vol_decoder(feat_vol))

    def gaussians_from_volume(self, vol, autocast=True):
        if autocast and self.fused_coarse and vol.is_cuda and coarse.supported(self.decoder):
            # the three bf16 Linear layers, the split and its activations as one HIP kernel per direction (lara_coarsedec.h)
            offset, shs, scaling, rotation, opacity = coarse.forward_coarse(self.decoder, vol, self.opacity_shift, self.scaling_shift)
        else:
            offset, shs, scaling, rotation, opacity = decode_coarse(self.decoder, vol, self.opacity_shift, self.scaling_shift, autocast)
        if self.opacity_bias is not None:
            opacity = opacity + self.opacity_bias
        half_cell = 0.5 * self.scene_size / self.n_offset_groups                # network.py:425-429
        B = offset.shape[0]
        centers = self.group_centers.unsqueeze(-2).expand(B, -1, self.K, -1).reshape(offset.shape) + offset * half_cell
        masks = torch.sigmoid(opacity.detach()).squeeze(-1) > 0.005
        return {"centers": centers, "shs": shs, "scaling": scaling, "rotation": rotation, "opacity": opacity,
                "masks": masks, "vol": vol.view(B, -1, vol.shape[-1])}

    def trained_like_opacity_bias(self):
        """SURVEY.md section 8d's second regime for the WHOLE step: a logit bias that makes the random-init network emit an opaque
        thin shell (|r - 0.35| < 0.02: +4 - shift, i.e. opacity ~ 0.98) in empty space (-6 - shift elsewhere), as
        `synthetic.make_scene(regime="trained")` does for the raster-only scenes.  Assign the result to `opacity_bias`."""
        r = self.group_centers.norm(dim=-1, keepdim=True)                                   # [1, voxels, 1]
        bias = torch.where((r - 0.35).abs() < 0.02, 4.0 - self.opacity_shift, -6.0 - self.opacity_shift)
        return bias.unsqueeze(-2).expand(-1, -1, self.K, -1).reshape(1, -1, 1).contiguous()

    @staticmethod
    def host_scalars(batch):
        """near/far and the fields of view enter ``MiniCam`` / the raster settings as Python floats (network.py:476-477:
        one device->host read per scene and value in the reference); here ONE read per batch, first thing in the step."""
        packed = torch.cat([batch["near_far"].float(), batch["fovx"].float()[:, None], batch["fovy"].float()[:, None]], 1).cpu()
        return [tuple(float(x) for x in row) for row in packed]

    def scene_cameras(self, batch, i, scalars=None):
        """``MiniCam`` for every target view of scene i (network.py:477-492, lightning/utils.py:22-48), batched."""
        near, far, fx, fy = (scalars or self.host_scalars(batch))[i]
        H, W = int(batch["meta"]["tar_h"][i]), int(batch["meta"]["tar_w"][i])
        return cameras.make_cameras(batch["tar_c2w"][i], W, H, fx, fy, near, far, device=batch["tar_c2w"].device)

    def _mark(self, name):
        """Stage boundary on the current stream (meaningful with n_streams = 1, where one stream holds the whole step)."""
        if self.stage_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.stage_events.append((name, e))

    # -- the step ------------------------------------------------------------------------------------------------
    def forward(self, batch, feat_vol, with_fine=True, n_views_sel=None):
        """`n_views_sel`: the number of INPUT views, which the reference draws at random in {2, 3, 4} when
        ``cfg.train.use_rand_views`` is set (network.py:437-441).  The fused sampler / fine decoder and the encoder's K|V layout
        are built for ``self.n_views`` (4: configs/base.yaml); any other count is rejected here rather than mis-indexed --
        run such a step through ``tools/reference_style.py: network_forward``-style torch operators instead."""
        if n_views_sel is not None and int(n_views_sel) != self.n_views:
            raise NotImplementedError(f"lara_amd.pipeline: built for {self.n_views} input views (configs/base.yaml n_views); got n_views_sel="
                                      f"{n_views_sel} (cfg.train.use_rand_views): not supported by the fused fine stage")
        if feat_vol.dim() == 6 and feat_vol.shape[1] != self.n_views:
            raise NotImplementedError(f"lara_amd.pipeline: the image-feature volume holds {feat_vol.shape[1]} views, the pipeline is built for {self.n_views}")
        return self._step(batch, lambda: self.gaussians(feat_vol), feat_vol.device, feat_vol.shape[0], with_fine)

    def forward_from_volume(self, batch, volume_feat_up, with_fine=True, autocast=True):
        """The step from the encoder's OUTPUT on (network.py:458-532): what tests hold against the reference's own
        `Network.forward` run with the same volume features (tests/golden/network_ref.npz)."""
        return self._step(batch, lambda: self.gaussians_from_volume(volume_feat_up, autocast), volume_feat_up.device,
                          volume_feat_up.shape[0], with_fine)

    def _step(self, batch, make_gaussians, dev, B, with_fine):
        if dev.type != "cuda":
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        n_sel = self.n_views
        scalars = self.host_scalars(batch)
        cur = torch.cuda.current_stream(dev)
        n_streams = min(self.n_streams, B)
        while len(self._streams) < n_streams and n_streams > 1:
            self._streams.append(torch.cuda.Stream(dev))
        sides = self._streams[:n_streams] if n_streams > 1 else [None] * B
        on = lambda s: torch.cuda.stream(s) if s is not None else contextlib.nullcontext()

        sizes = [(int(batch["meta"]["tar_w"][i]), int(batch["meta"]["tar_h"][i])) for i in range(B)]
        cams_of = cameras.make_cameras_scenes(batch["tar_c2w"], sizes, scalars, device=batch["tar_c2w"].device)   # every MiniCam of the batch
        # the views' background colours with their rows on 16-byte boundaries, once per batch (render_views would pad per call)
        bg_rows = torch.nn.functional.pad(batch["bg_color"].float(), (0, 1))[..., :3]
        self._mark("start")
        g = make_gaussians()
        # the input images as [B, n_sel, 3, H, W] (network.py:437-438, :469): what the sampler reads as `img_ref`
        inps = batch["tar_rgb"][:, :n_sel].permute(0, 1, 4, 2, 3).float().contiguous()
        if with_fine:
            masks = g["masks"]
            if self.fine_mask == "reference":
                masks = torch.stack([check_mask(masks[i], self.training) for i in range(B)])
        self._mark("encoder+decoder")
        # The output dictionary's [B, H, V*W, C] tensors (network.py:527-529) are allocated up front and every scene's
        # post-processing writes its slice: the stack at the end copies nothing (round 5: 12 concatenation launches, 0.37 ms).
        # Needs scenes of one size (a batch of the reference's loader is).
        bufs = None
        if len(set(sizes)) == 1 and len({len(c) for c in cams_of}) == 1:
            bufs = batch_map_buffers(B, sizes[0][1], sizes[0][0] * len(cams_of[0]), dev, ("", "_fine") if with_fine else ("",))
        for s in sides:
            if s is not None:
                s.wait_stream(cur)
                # made on the caller's stream, read (and saved for the backward) on the scene streams -- the batch's own tensors
                # too: with a generator that frees the previous batch at the next iteration their safety would otherwise
                # rest on the caller joining the streams before it drops the batch
                hand_over((g, inps, cams_of, bufs, bg_rows, [batch[k] for k in ("tar_rays", "bg_color", "tar_w2c", "tar_ixt") if k in batch]), s)

        # per-scene tensors: one unbind per tensor (its backward is one stack; `x[i]` per use would cost a zero-filled
        # [B,P,C] buffer and an accumulation for every use)
        sc = {k: g[k].unbind(0) for k in ("centers", "shs", "opacity", "scaling", "rotation")}
        per_scene = [None] * B
        coarse_raster = [[] for _ in range(B)]       # the coarse calls' own outputs: the fine calls filter their lists (`subset_of`)
        for i in range(B):                                                      # network.py:473-497
            s = sides[i % len(sides)]
            with on(s):
                per_scene[i] = self.gs_render.render_views(
                    cams_of[i], batch["tar_rays"][i], sc["centers"][i], sc["shs"][i], sc["opacity"][i], sc["scaling"][i],
                    sc["rotation"][i], dev, bg_colors=bg_rows[i], concat=True,
                    into=None if bufs is None else {k: v[i] for k, v in bufs.items() if not k.endswith("_fine")},
                    raster_out=coarse_raster[i])
                self._mark("coarse views")
        if with_fine:
            # the sizes of the masked subsets: host reads on the stream that holds only the encoder + decoder + masks
            # (every scene's coarse views are already enqueued on the side streams)
            idx = [masks[i].nonzero().squeeze(-1) for i in range(B)]
            folded = fold_fine_weights(self.decoder)      # once per step, shared by the scenes (on the caller's stream)
            # every scene's volume-feature rows (network.py:509) through ONE autograd node, on the caller's stream: its backward
            # builds the gradient of `vol` in place instead of four dense per-scene gradients + their concatenation
            vox = [torch.div(idx[i], self.K, rounding_mode="floor") for i in range(B)]
            vol_rows = voxel_rows_scenes(g["vol"], vox)
            for s in sides:
                if s is not None:
                    s.wait_stream(cur)
                    hand_over((masks, folded, vol_rows), s)
            for i in range(B):                                                  # network.py:502-525
                s = sides[i % len(sides)]
                hand_over(idx[i], s)
                with on(s):
                    co = per_scene[i]
                    H, VW = co["acc_map"].shape
                    V, W = len(cams_of[i]), VW // len(cams_of[i])
                    # the five x[mask] of network.py:514-524 as one launch per direction
                    five = [sc["centers"][i], sc["shs"][i], sc["opacity"][i], sc["scaling"][i], sc["rotation"][i]]
                    centers_f, shs_sel, opacity_f, scaling_f, rotation_f = take_rows_multi(five, idx[i])
                    # the sampler reads the first n_sel views of the side-by-side maps in place (network.py:499 stacks them)
                    pf = sample_point_feats(centers_f, batch["tar_w2c"][i, :n_sel], batch["tar_ixt"][i, :n_sel], inps[i],
                                            co["image"], co["acc_map"], co["depth"], row_views=V)
                    sh_res = forward_fine(self.decoder, vol_rows[i], torch.einsum("lcb->blc", pf), folded)
                    shs_f = sh_res.view(-1, *g["shs"].shape[-2:]) + shs_sel
                    self._mark("sampler+forward_fine")
                    co.update(self.gs_render.render_views(
                        cams_of[i], batch["tar_rays"][i], centers_f, shs_f, opacity_f, scaling_f, rotation_f, dev,
                        bg_colors=bg_rows[i], prex="_fine", concat=True,
                        into=None if bufs is None else {k: v[i] for k, v in bufs.items() if k.endswith("_fine")},
                        subset_of=(coarse_raster[i][0][0], idx[i]) if self.fine_reuses_coarse_lists else None))
                    self._mark("fine views")
        outs = per_scene                                                        # network.py:527: already [H, V*W, C] per key
        for i, s in enumerate(sides):
            if s is not None:
                cur.wait_stream(s)
        if sides[0] is not None:
            hand_over(per_scene, cur)       # made on the scene streams, read by the stack below on the caller's
        if bufs is not None:
            out = {k: _AssembleScenes.apply(bufs[k], *[o[k] for o in outs]) for k in outs[0]}       # network.py:529, without the copy
        else:
            out = {k: torch.stack([o[k] for o in outs]) for k in outs[0]}       # network.py:529
        self._mark("outputs")
        return out

    def join_streams(self):
        """Call after ``backward()``: the next step's allocations on the caller's stream must not overtake kernels the
        backward left on the scene streams."""
        cur = torch.cuda.current_stream()
        for s in self._streams:
            cur.wait_stream(s)


def lara_loss(batch, output, it=10000, ms_ssim=True):
    """lightning/loss.py:17-60 as plain torch: colour MSE and ``0.5 * (1 - MS_SSIM)`` (``lara_amd.loss.ms_ssim``: the
    restatement of the absent `pytorch_msssim` package; ``ms_ssim=False`` leaves the term out) for the coarse and the fine
    images, and after iteration 1000 the coarse pass's distortion (x 1000) and normal-consistency (x 0.2) terms.  Returns
    (loss, scalar_stats) with the reference's keys (mse, psnr, ssim, distortion, normal; `_fine` variants)."""
    from .loss import ms_ssim_terms
    B, V, H, W = batch["tar_rgb"].shape[:-1]
    tar = batch["tar_rgb"].permute(0, 2, 1, 3, 4).reshape(B, H, V * W, 3)
    loss, stats = 0, {}
    for prex in ("", "_fine"):
        if f"image{prex}" not in output:
            continue
        if prex == "_fine" and "acc_map_fine" not in output:                    # loss.py:31
            continue
        mse = ((output[f"image{prex}"] - tar) ** 2).mean()
        loss = loss + mse
        stats[f"mse{prex}"] = mse.detach()
        stats[f"psnr{prex}"] = -10.0 * torch.log10(mse.detach())               # loss.py:36-39
        if ms_ssim:
            extra, st = ms_ssim_terms(batch, output, (prex,), fused=False)       # (this function is the torch formulation)
            loss = loss + extra[prex]                                           # loss.py:41-45
            stats.update(st)
        if f"rend_dist{prex}" in output and it > 1000 and prex != "_fine":
            dist = output[f"rend_dist{prex}"].mean()
            err = ((1 - (output[f"rend_normal{prex}"] * output[f"depth_normal{prex}"]).sum(dim=-1))
                   * output[f"acc_map{prex}"].detach()).mean()
            loss = loss + dist * 1000 + err * 0.2
            stats[f"distortion{prex}"], stats[f"normal{prex}"] = dist.detach(), err.detach()
    return loss, stats
