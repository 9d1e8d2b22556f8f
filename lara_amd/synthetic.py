"""Seeded synthetic surfel scenes with the statistics LaRa's decoder emits.

Every constant is the reference's own:
  * P = (2 * vol_embedding_reso)^3 * K = 64^3 * 2 surfels     configs/base.yaml:13,23
  * grid centres ((i + 0.5) / 64 * 2 - 1) * 0.5                lightning/network.py:345-349
  * per-surfel offset  U(-1,1) * 0.5 * scene_size / 64         lightning/network.py:425-429
  * log-scale shift  ln(0.5 * (2 / 64) / 3)                    lightning/network.py:341-342
  * opacity logit shift -2.1792                                lightning/network.py:340
  * SH degree 1 (4 coefficients x 3)                           configs/base.yaml:14

Tensors are returned *pre-activation* in the layout ``Renderer.render_img`` takes
(renderer_2dgs.py:167-180): centers [P,3], shs [P,4,3], opacity logit [P,1], log-scales [P,2],
raw quaternions [P,4]; ``activate`` applies the reference's activations (renderer_2dgs.py:106-114).
"""
from __future__ import annotations

import math

import torch

SCENE_SIZE = 0.5
OPACITY_SHIFT = -2.1792


def make_scene(grid: int = 64, K: int = 2, regime: str = "init", seed: int = 0,
               sh_coeffs: int = 4, device="cpu") -> dict:
    """Synthetic decoder output.  regime: "init" (semi-transparent fog, every pixel walks the
    whole tile list) or "trained" (opaque thin shell at radius 0.35, early termination)."""
    g = torch.Generator().manual_seed(seed)
    ar = (torch.arange(grid, dtype=torch.float32) + 0.5) / grid * 2 - 1
    centers = torch.stack(torch.meshgrid(ar, ar, ar, indexing="ij"), dim=-1).reshape(-1, 3) * SCENE_SIZE
    centers = centers[:, None, :].expand(-1, K, -1).reshape(-1, 3)
    P = centers.shape[0]
    half_cell = 0.5 * SCENE_SIZE / (grid // 2)
    centers = centers + (torch.rand(P, 3, generator=g) * 2 - 1) * half_cell
    voxel = 2.0 / grid
    scale_shift = math.log(0.5 * voxel / 3.0)
    scales = torch.randn(P, 2, generator=g) * 0.3 + scale_shift
    rotations = torch.randn(P, 4, generator=g)
    opacity = torch.randn(P, 1, generator=g) + OPACITY_SHIFT
    if regime == "trained":
        r = centers.norm(dim=-1, keepdim=True)
        shell = (r - 0.35).abs() < 0.02
        opacity = torch.where(shell, opacity + 4.0 - OPACITY_SHIFT, torch.full_like(opacity, -6.0))
    elif regime != "init":
        raise ValueError(regime)
    shs = torch.randn(P, sh_coeffs, 3, generator=g) * 0.5
    shs[:, 0, :] = torch.randn(P, 3, generator=g) * 0.8  # DC: colours land inside (0,1) mostly
    out = dict(centers=centers, shs=shs, opacity=opacity, scales=scales, rotations=rotations)
    return {k: v.contiguous().to(device) for k, v in out.items()}


def activate(scene: dict) -> dict:
    """The reference's activations (renderer_2dgs.py:106-114,181-189)."""
    return dict(
        means3D=scene["centers"],
        shs=scene["shs"],
        opacities=torch.sigmoid(scene["opacity"]),
        scales=torch.exp(scene["scales"]),
        rotations=torch.nn.functional.normalize(scene["rotations"]),
    )
