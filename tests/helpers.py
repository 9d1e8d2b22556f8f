"""Shared scene / view builders for the tests (CPU side: numpy + the oracle; GPU side: torch)."""
import math

import numpy as np
import torch

import oracle
from lara_amd import cameras, synthetic


def oracle_view(cam, bg, sh_degree=1, scale_modifier=1.0):
    return oracle.View(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5),
                       math.tan(cam.FoVy * 0.5), np.asarray(bg, np.float32), scale_modifier,
                       cam.world_view_transform.cpu().numpy(), cam.full_proj_transform.cpu().numpy(),
                       sh_degree, cam.camera_center.cpu().numpy())


def raster_settings(cam, bg, sh_degree=1, device="cuda", scale_modifier=1.0, debug=False):
    from lara_amd import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.tensor(bg, dtype=torch.float32, device=device), scale_modifier=scale_modifier,
        viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device),
        sh_degree=sh_degree, campos=cam.camera_center.to(device), prefiltered=False, debug=debug)


def small_scene(grid=16, K=2, size=128, n_views=4, seed=0, regime="init", scale_boost=None,
                opacity_boost=0.0, sh_coeffs=4):
    """A LaRa-distributed scene scaled so that splats keep ~the same pixel footprint as the
    64^3 / 512^2 configuration: voxel size grows with 64/grid, image shrinks with size/512."""
    sc = synthetic.make_scene(grid=grid, K=K, regime=regime, seed=seed, sh_coeffs=sh_coeffs)
    if scale_boost is not None:
        sc["scales"] = sc["scales"] + math.log(scale_boost)
    sc["opacity"] = sc["opacity"] + opacity_boost
    act = synthetic.activate(sc)
    cams = cameras.make_cameras(cameras.turntable_c2w(n_views), size, size, 0.75, 0.75, 0.5, 2.5)
    return act, cams


def to_numpy(act):
    return {k: v.detach().cpu().numpy() for k, v in act.items()}


def run_oracle(view, act_np, **kw):
    return oracle.forward(view, act_np["means3D"], act_np["opacities"], shs=act_np.get("shs"),
                          scales=act_np.get("scales"), rotations=act_np.get("rotations"), **kw)


def psnr(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 999.0 if mse == 0 else 10.0 * math.log10(peak * peak / mse)
