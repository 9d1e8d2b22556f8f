"""What an UNMODIFIED LaRa executes around the drop-in rasteriser, as the same sequence of plain torch operators -- the
yardstick the opt-in fused paths of this package are measured against, and the step `bench.py` reports as
`drop_in_step` beside its headline.

`/root/reference` does not travel to the GPU box, so the reference's own classes cannot run there; these functions issue,
operator for operator, what they issue on the device:

* `render_img`         lightning/renderer_2dgs.py:167-268 (+ `depth_to_normal` :78-89): per view the three activations, a
                       fresh zero `means2D` that requires grad, ONE `GaussianRasterizer` call (the shim), ~15 elementwise /
                       matmul / slicing kernels of post-processing;
* `point_feats`        `Network.get_point_feats` + `projection`, lightning/network.py:182-187, :390-411 (`F.grid_sample`);
* `forward_fine`       `Decoder.forward_fine`, network.py:280-284, with the decoder's own nn modules under bf16 autocast;
* `network_forward`    the loop of `Network.forward`, network.py:473-529: scene by scene, view by view, on ONE stream, boolean
                       `x[mask]` indexing, `torch.cat` / `torch.stack` of the per-view dictionaries.

Nothing here is a product path and nothing here is faster than it looks: it exists to be slow in exactly the reference's way.
Values equal the fused paths' to fp32 rounding (tests/test_pipeline.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from lara_amd.rasterizer import GaussianRasterizer


def render_img(renderer, cam, rays, centers, shs, opacity, scales, rotations, device, prex="", depth_ratio=0.0):
    """`renderer`: a `lara_amd.renderer.Renderer` (used for its `_settings` only -- the 12 raster settings of
    renderer_2dgs.py:124-137).  Returns the reference's dictionary of six maps."""
    rast = GaussianRasterizer(raster_settings=renderer._settings(cam, device=device))
    opacity = torch.sigmoid(opacity)
    scales = torch.exp(scales)
    rotations = F.normalize(rotations)
    screenspace_points = torch.zeros_like(centers, dtype=centers.dtype, requires_grad=True, device=device) + 0
    rendered_image, radii, allmap = rast(means3D=centers, means2D=screenspace_points, shs=shs, opacities=opacity, scales=scales,
                                         rotations=rotations, cov3D_precomp=None)
    rendered_image = rendered_image.clamp(0, 1)
    render_alpha = allmap[1:2]
    render_normal = allmap[2:5]
    render_normal = (render_normal.permute(1, 2, 0) @ (cam.world_view_transform[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    surf_depth = render_depth_expected * (1 - depth_ratio) + depth_ratio * render_depth_median
    # depth_to_normal: finite differences of the back-projected surface
    points = (rays[..., :3].reshape(-1, 3) + surf_depth.reshape(-1, 1) * rays[..., 3:].reshape(-1, 3)).reshape(*surf_depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    output[1:-1, 1:-1, :] = F.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    surf_normal = output.permute(2, 0, 1) * render_alpha.detach()
    return {f"image{prex}": rendered_image.permute(1, 2, 0), f"depth{prex}": surf_depth.permute(1, 2, 0),
            f"acc_map{prex}": render_alpha.permute(1, 2, 0).squeeze(-1), f"rend_normal{prex}": render_normal.permute(1, 2, 0),
            f"depth_normal{prex}": surf_normal.permute(1, 2, 0), f"rend_dist{prex}": render_dist.squeeze(0)}


def point_feats(points, w2cs, ixts, img_ref, image, acc_map, depth):
    """points [n,3]; w2cs [V,4,4]; ixts [V,3,3]; img_ref [V,3,h,w]; image [V,h,w,3]; acc_map [V,h,w]; depth [V,h,w,1]
    -> [V, 8, n] (network.py:390-411)."""
    V, n = img_ref.shape[0], points.shape[0]
    h, w = img_ref.shape[-2:]
    pc = points.reshape(1, -1, 3) @ w2cs[:, :3, :3].permute(0, 2, 1) + w2cs[:, :3, 3][:, None]
    q = pc @ ixts.permute(0, 2, 1)
    xy, z = q[..., :2] / q[..., -1:], q[..., -1:]
    grid = (xy + 0.5) / torch.tensor([w, h], device=points.device) * 2 - 1.0
    imgs = torch.cat((image, acc_map.unsqueeze(-1), depth), dim=-1)
    imgs = torch.cat((img_ref, torch.einsum("bhwc->bchw", imgs)), dim=1)
    feats = F.grid_sample(imgs, grid.unsqueeze(1), align_corners=False).view(V, -1, n).to(imgs)
    z_diff = (feats[:, -1:] - z.view(V, -1, n)).abs()
    return torch.cat((feats[:, :-1], z_diff), dim=1)


def forward_fine(decoder, volume_feat, pf, chunk=32768):
    """`Decoder.forward_fine` (network.py:280-284) through the decoder's nn modules under bf16 autocast, in chunks of rows
    (`nn.MultiheadAttention` materialises [n, heads, 1, 4] score tensors)."""
    outs = []
    with torch.autocast("cuda", dtype=torch.bfloat16):
        for a, b in zip(volume_feat.split(chunk), pf.split(chunk)):
            x = decoder.norm(a.unsqueeze(1))
            x = decoder.cross_att(x, b, b, need_weights=False)[0]
            outs.append(decoder.mlp_fine(x).float())
    return torch.cat(outs)


def network_forward(pipe, batch, feat_vol, with_fine=True):
    """`Network.forward` from the image-feature volume on (network.py:455-532) the way the reference issues it; `pipe` is a
    `LaRaPipeline` (its encoder, decoder and constants).  The encoder stays `pipe.vol_decoder` (this package's HIP
    VolTransformer or the reference's module, whichever the pipeline holds); everything behind it is the reference's sequence."""
    from lara_amd.pipeline import check_mask, decode_coarse
    vol = pipe.vol_decoder(feat_vol)
    offset, shs, scaling, rotation, opacity = decode_coarse(pipe.decoder, vol, pipe.opacity_shift, pipe.scaling_shift, autocast=True)
    if pipe.opacity_bias is not None:
        opacity = opacity + pipe.opacity_bias
    B = offset.shape[0]
    dev = feat_vol.device
    half_cell = 0.5 * pipe.scene_size / pipe.n_offset_groups
    centers = pipe.group_centers.unsqueeze(-2).expand(B, -1, pipe.K, -1).reshape(offset.shape) + offset * half_cell
    masks = torch.sigmoid(opacity.detach()).squeeze(-1) > 0.005
    volf = vol.view(B, -1, vol.shape[-1])
    n_sel = pipe.n_views
    inps = batch["tar_rgb"][:, :n_sel].permute(0, 1, 4, 2, 3).float()
    outs = []
    for i in range(B):
        cams = pipe.scene_cameras(batch, i)                     # (MiniCam per view: network.py:477-492; one host read per scene)
        views = []
        for j, cam in enumerate(cams):
            pipe.gs_render.set_bg_color(batch["bg_color"][i, j])
            views.append(render_img(pipe.gs_render, cam, batch["tar_rays"][i, j], centers[i], shs[i], opacity[i], scaling[i], rotation[i], dev))
        if with_fine:
            mask = masks[i]
            if pipe.fine_mask == "reference":
                mask = check_mask(mask, pipe.training)
            ren = {k: torch.stack([v[k] for v in views[:n_sel]]) for k in ("image", "acc_map", "depth")}
            centers_f = centers[i][mask]
            pf = point_feats(centers_f, batch["tar_w2c"][i, :n_sel], batch["tar_ixt"][i, :n_sel], inps[i], ren["image"], ren["acc_map"], ren["depth"])
            vpf = volf[i].unsqueeze(1).expand(-1, pipe.K, -1)[mask.view(-1, pipe.K)]
            sh_res = forward_fine(pipe.decoder, vpf, torch.einsum("lcb->blc", pf))
            shs_f = sh_res.view(-1, *shs.shape[-2:]) + shs[i][mask]
            for j, cam in enumerate(cams):
                pipe.gs_render.set_bg_color(batch["bg_color"][i, j])
                views[j].update(render_img(pipe.gs_render, cam, batch["tar_rays"][i, j], centers_f, shs_f, opacity[i][mask],
                                           scaling[i][mask], rotation[i][mask], dev, prex="_fine"))
        outs.append({k: torch.cat([v[k] for v in views], dim=1) for k in views[0]})
    return {k: torch.stack([o[k] for o in outs]) for k in outs[0]}
