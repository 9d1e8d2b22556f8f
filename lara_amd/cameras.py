"""Camera matrices in the conventions the reference hands to the rasteriser.

Mirrors (does not import) the reference's host code:
  * ``getProjectionMatrix``  -- lightning/utils.py:5-19  (P[3,2] = 1, depth in [0,1])
  * ``MiniCam``              -- lightning/utils.py:22-48 (row-vector convention:
        world_view_transform = inverse(c2w)^T, full_proj_transform = that @ P^T,
        camera_center = -c2w[:3,3]  -- sic, the reference negates the translation)
  * canonical gobjaverse turntable -- tools/gen_video_path.py:16,23-37

The reference builds one camera per view on the host with a 4x4 ``torch.inverse`` and several
tiny kernels (SURVEY.md section 7 "Python overhead per view"); here a whole batch of views is
built with batched linear algebra in one go, on whatever device ``c2w`` lives on.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float,
                      dtype=torch.float32, device=None) -> torch.Tensor:
    """Perspective matrix of lightning/utils.py:5-19 (column-vector form, NOT yet transposed)."""
    P = torch.zeros(4, 4, dtype=dtype, device=device)
    P[0, 0] = 1.0 / math.tan(fovx / 2.0)
    P[1, 1] = 1.0 / math.tan(fovy / 2.0)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class Camera:
    """Duck-types the attributes ``Renderer.set_rasterizer`` reads (renderer_2dgs.py:119-137)."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    znear: float
    zfar: float
    world_view_transform: torch.Tensor  # [4,4] = w2c^T
    projection_matrix: torch.Tensor     # [4,4] = P^T
    full_proj_transform: torch.Tensor   # [4,4] = w2c^T @ P^T
    camera_center: torch.Tensor         # [3]   = -c2w[:3,3]


def make_cameras(c2w: torch.Tensor, width: int, height: int, fovx: float, fovy: float,
                 znear: float, zfar: float, device=None) -> list:
    """Batch version of ``MiniCam.__init__`` (lightning/utils.py:22-48).  ``c2w``: [V,4,4]."""
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    if c2w.dim() == 2:
        c2w = c2w[None]
    device = device if device is not None else c2w.device
    # inv_ex: the same factorisation as torch.inverse without the host read of its status word (one stall per call)
    w2c = torch.linalg.inv_ex(c2w.double())[0].float()
    wvt = w2c.transpose(1, 2).contiguous().to(device)
    PT = projection_matrix(znear, zfar, fovx, fovy).t().contiguous().to(device)
    full = (wvt @ PT).float().contiguous()
    # (rows of 4 floats: every view's centre starts on a 16-byte boundary, which the rasteriser's camera loads want --
    # a [V,3] tensor's rows do not, and each would be cloned on every rasteriser call)
    centers = torch.nn.functional.pad(-c2w[:, :3, 3], (0, 1)).contiguous().to(device)[:, :3]
    return [Camera(int(width), int(height), float(fovx), float(fovy), float(znear), float(zfar),
                   wvt[i], PT, full[i], centers[i]) for i in range(c2w.shape[0])]


def make_cameras_scenes(c2w: torch.Tensor, sizes, scalars, device=None) -> list:
    """``make_cameras`` for every scene of a batch in ONE pass of batched linear algebra (one inverse, one product, one
    padded copy for all B x V cameras instead of one set per scene: LaRa's loop builds them scene by scene,
    network.py:476-492).  ``c2w``: [B,V,4,4]; ``sizes``: per scene (width, height); ``scalars``: per scene
    (znear, zfar, fovx, fovy) as Python floats.  Returns a list (scene) of lists (view) of `Camera`."""
    c2w = torch.as_tensor(c2w, dtype=torch.float32)
    B, V = c2w.shape[:2]
    device = device if device is not None else c2w.device
    w2c = torch.linalg.inv_ex(c2w.reshape(B * V, 4, 4).double())[0].float()
    wvt = w2c.transpose(1, 2).contiguous().to(device).view(B, V, 4, 4)
    PT = torch.stack([projection_matrix(near, far, fx, fy).t() for near, far, fx, fy in scalars]).contiguous().to(device)
    full = torch.matmul(wvt, PT[:, None]).float().contiguous()
    centers = torch.nn.functional.pad(-c2w[..., :3, 3], (0, 1)).contiguous().to(device)[..., :3]
    return [[Camera(int(sizes[b][0]), int(sizes[b][1]), float(scalars[b][2]), float(scalars[b][3]), float(scalars[b][0]),
                    float(scalars[b][1]), wvt[b, i], PT[b], full[b, i], centers[b, i]) for i in range(V)] for b in range(B)]


def turntable_c2w(n_views: int, elevation_deg: float = 0.0) -> torch.Tensor:
    """Canonical gobjaverse pose rotated about +z in steps of 2*pi/n (tools/gen_video_path.py:23-37).

    Returns [n_views,4,4] fp32 camera-to-world matrices (|t| ~ 1.906).
    """
    base = torch.eye(4, dtype=torch.float64)
    base[:3, :3] = torch.tensor([[0.0, 1.0, 0.0],
                                 [0.4515947, 0.0, -0.8922232],
                                 [-0.8922232, 0.0, -0.4515947]], dtype=torch.float64).t()
    base[:3, 3] = torch.tensor([1.70006549, 0.0, 0.8604804], dtype=torch.float64)
    if elevation_deg != 0.0:
        a = elevation_deg / 180.0 * math.pi
        ry = torch.eye(4, dtype=torch.float64)
        ry[0, 0], ry[0, 2], ry[2, 0], ry[2, 2] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        base = ry @ base
    out = []
    for i in range(n_views):
        a = 2.0 * math.pi * i / n_views
        rz = torch.eye(4, dtype=torch.float64)
        rz[0, 0], rz[0, 1], rz[1, 0], rz[1, 1] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
        out.append(rz @ base)
    return torch.stack(out).float()
