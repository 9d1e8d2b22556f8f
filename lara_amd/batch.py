"""LaRa's batch dictionary with device-side ray generation.

The reference builds one dictionary per scene on the CPU workers (dataLoader/gobjverse.py:40-96;
default collate adds the batch axis) and ships ``tar_rays`` / ``tar_rays_down`` -- 6 floats per
pixel per view, 50 MB per scene at 8 x 512^2 -- through the DataLoader.  Here the cameras are device
tensors and ``build_rays`` is a HIP kernel (``include/lara_rays.h``); ``synthetic_batch`` assembles a
dictionary with the reference's keys, shapes and dtypes from seeded synthetic cameras (no dataset in
this environment), aligned to the first view exactly as the loader does (gobjverse.py:57-64).
"""
from __future__ import annotations

import ctypes
import math

import torch

from .rasterizer import _check, load_library

_configured = False


def _lib():
    global _configured
    lib = load_library()
    if not _configured:
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        lib.lara_build_rays_out.restype = ctypes.c_int
        lib.lara_build_rays_out.argtypes = [i32, i32, i32, ctypes.c_float, vp, vp, vp, vp]
        _configured = True
    return lib


def fov_to_ixt(fov: torch.Tensor, reso) -> torch.Tensor:
    """gobjverse.py:10-15: ``fov`` [..., 2] (x, y) radians, ``reso`` (W, H) -> intrinsics [..., 3, 3]."""
    reso_t = torch.as_tensor(reso, dtype=torch.float32, device=fov.device)
    focal = 0.5 * reso_t / torch.tan(0.5 * fov)
    ixt = torch.zeros(*fov.shape[:-1], 3, 3, dtype=torch.float32, device=fov.device)
    ixt[..., 0, 0], ixt[..., 1, 1], ixt[..., 2, 2] = focal[..., 0], focal[..., 1], 1.0
    ixt[..., 0, 2], ixt[..., 1, 2] = reso_t[0] / 2, reso_t[1] / 2
    return ixt


def build_rays(c2ws: torch.Tensor, ixts: torch.Tensor, H: int, W: int, scale: float = 1.0) -> torch.Tensor:
    """``build_rays`` of dataLoader/utils.py:21-34 on the device: c2ws [V,4,4], ixts [V,3,3] ->
    rays [V, int(H*scale), int(W*scale), 6] (origin, unnormalised direction), fp32.  ``ixts`` is
    not modified (the reference scales it in place; its callers pass copies)."""
    if not c2ws.is_cuda:
        raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
    V = c2ws.shape[0]
    if c2ws.shape != (V, 4, 4) or ixts.shape != (V, 3, 3):
        raise RuntimeError("expected c2ws [V,4,4] and ixts [V,3,3]")
    c = c2ws.detach().float().contiguous()
    k = ixts.detach().float().contiguous().to(c.device)
    Hs, Ws = int(H * scale), int(W * scale)
    rays = torch.empty(V, Hs, Ws, 6, dtype=torch.float32, device=c.device)
    with torch.cuda.device(c.device):
        # the output size is computed once, here, as the reference does (double precision), and handed over
        rc = _lib().lara_build_rays_out(V, Hs, Ws, float(scale), c.data_ptr(), k.data_ptr(), rays.data_ptr(),
                                    torch.cuda.current_stream(c.device).cuda_stream)
    _check(rc, "lara_build_rays_out")
    return rays


def _canonical_c2ws(n_views: int, radius: float, g: torch.Generator) -> torch.Tensor:
    """Cameras on a sphere looking at the origin (OpenCV convention: +z forward, +y down), seeded."""
    az = torch.rand(n_views, generator=g) * 2 * math.pi
    el = (torch.rand(n_views, generator=g) - 0.5) * 1.2
    pos = radius * torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], -1)
    fwd = torch.nn.functional.normalize(-pos, dim=-1)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.nn.functional.normalize(torch.cross(fwd, up, dim=-1), dim=-1)
    down = torch.cross(fwd, right, dim=-1)
    c2w = torch.eye(4).repeat(n_views, 1, 1)
    c2w[:, :3, 0], c2w[:, :3, 1], c2w[:, :3, 2], c2w[:, :3, 3] = right, down, fwd, pos
    return c2w


def synthetic_batch(batch_size: int = 4, n_views: int = 8, H: int = 512, W: int = 512, n_input: int = 4,
                    fov: float = 0.75, radius: float = 1.906, seed: int = 0, device="cuda") -> dict:
    """A collated training batch with the keys / shapes / dtypes of gobjverse.__getitem__ + default
    collate: tar_c2w, tar_w2c [B,V,4,4]; tar_ixt [B,V,3,3]; tar_rgb [B,V,H,W,3]; tar_msk [B,V,H,W] (uint8);
    transform_mats [B,1,4,4]; bg_color [B,V,3]; near_far [B,2]; fovx, fovy [B]; tar_rays [B,V,H,W,6];
    tar_rays_down [B,V,H/16,W/16,6]; meta.  Images are the background colour (there is no dataset here);
    everything lives on ``device`` and the rays come from the HIP kernel."""
    dev = torch.device(device)
    g = torch.Generator().manual_seed(seed)
    out = {k: [] for k in ("tar_c2w", "tar_w2c", "tar_ixt", "tar_rgb", "tar_msk", "transform_mats", "bg_color",
                           "near_far", "tar_rays", "tar_rays_down")}
    for b in range(batch_size):
        c2w = _canonical_c2ws(n_views, radius, g)
        w2c = torch.linalg.inv(c2w)
        # align cameras using the first view (gobjverse.py:57-64): it ends up on the -z axis at distance r
        r = float(c2w[0, :3, 3].norm())
        ref_c2w, ref_w2c = torch.eye(4)[None].clone(), torch.eye(4)[None].clone()
        ref_c2w[:, 2, 3], ref_w2c[:, 2, 3] = -r, r
        transform = ref_c2w @ w2c[:1]
        w2c_al = w2c @ c2w[:1] @ ref_w2c
        c2w_al = transform @ c2w
        ixt = fov_to_ixt(torch.full((n_views, 2), fov), (W, H))
        # input / test views see a white background, the novel views of a training step a random grey
        # level from {0, 0.5, 1} (gobjverse.py:103-106)
        lv = torch.tensor([0.0, 0.5, 1.0])[torch.randint(0, 3, (n_views,), generator=g)]
        lv[:n_input] = 1.0
        bg = lv[:, None].expand(-1, 3).contiguous()
        c2w_d, ixt_d = c2w_al.to(dev), ixt.to(dev)
        out["tar_c2w"].append(c2w_d)
        out["tar_w2c"].append(w2c_al.to(dev))
        out["tar_ixt"].append(ixt_d)
        out["tar_rgb"].append(bg.to(dev)[:, None, None, :].expand(-1, H, W, -1).contiguous())
        out["tar_msk"].append(torch.zeros(n_views, H, W, dtype=torch.uint8, device=dev))
        out["transform_mats"].append(transform.to(dev))
        out["bg_color"].append(bg.to(dev))
        out["near_far"].append(torch.tensor([r - 0.8, r + 0.8], device=dev))
        out["tar_rays"].append(build_rays(c2w_d, ixt_d, H, W, 1.0))
        out["tar_rays_down"].append(build_rays(c2w_d, ixt_d, H, W, 1.0 / 16))
    batch = {k: torch.stack(v) for k, v in out.items()}
    batch["fovx"] = torch.full((batch_size,), fov, device=dev)
    batch["fovy"] = torch.full((batch_size,), fov, device=dev)
    batch["meta"] = {"scene": [f"synthetic_{seed}_{b}" for b in range(batch_size)],
                     "tar_view": [list(range(n_views)) for _ in range(batch_size)],
                     "frame_id": torch.zeros(batch_size, dtype=torch.long),
                     "tar_h": torch.full((batch_size,), H), "tar_w": torch.full((batch_size,), W)}
    return batch
