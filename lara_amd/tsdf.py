"""TSDF fusion of rendered views on the GPU: the `volume.integrate(...)` loop of the
reference's ``MeshExtractor.extract`` (tools/meshExtractor.py:67-110) without leaving the device -- the reference
copies depth, alpha and colour of every one of its 48 views to the host and feeds Open3D's CPU
``ScalableTSDFVolume``.  Dense grid, Open3D's voxel conventions and per-voxel update (include/lara_tsdf.h).
Mesh extraction from the fused volume (marching cubes + the cluster filter, meshExtractor.py:112-135) is not part of
the hot path and stays with the caller: ``volume()`` returns the grid as tensors.  No CPU path."""
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
        vp, i32, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
        lib.lara_tsdf_integrate.restype = ctypes.c_int
        lib.lara_tsdf_integrate.argtypes = [i32, ctypes.POINTER(f32 * 3), f32, f32, i32, i32, i32] + [vp] * 9
        _configured = True
    return lib


class TSDFVolume:
    """``TSDFVolume(origin, voxel_length, sdf_trunc, resolution)``: voxel (i,j,k) is centred at
    origin + voxel_length * (i + 0.5, j + 0.5, k + 0.5); `MeshExtractor.extract` uses voxel_length = radius / 256 and
    sdf_trunc = 2 voxels around the object's bounding box (tools/meshExtractor.py:54-58)."""

    def __init__(self, origin, voxel_length: float, sdf_trunc: float, resolution: int, device="cuda"):
        self.origin = tuple(float(o) for o in origin)
        self.voxel_length, self.sdf_trunc, self.res = float(voxel_length), float(sdf_trunc), int(resolution)
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("lara_amd: tensors must live on an MI355X (HIP) device; there is no CPU path")
        n = self.res ** 3
        self.tsdf = torch.zeros(n, dtype=torch.float32, device=dev)
        self.weight = torch.zeros(n, dtype=torch.float32, device=dev)
        self.rgb = torch.zeros(n, 3, dtype=torch.float32, device=dev)

    @torch.no_grad()
    def integrate(self, depth, color, intrinsics, extrinsics, depth_trunc):
        """depth [V,H,W] (0 = no measurement), color [V,H,W,3] in 0..255, intrinsics [V,4] = (fx, fy, cx, cy),
        extrinsics [V,4,4] world->camera, depth_trunc [V] or a float.  Views are folded in the given order."""
        dev = self.tsdf.device
        f = lambda t: torch.as_tensor(t, dtype=torch.float32, device=dev).contiguous()
        depth, color, intrinsics, extrinsics = f(depth), f(color), f(intrinsics), f(extrinsics)
        V, H, W = depth.shape
        if color.shape != (V, H, W, 3) or intrinsics.shape != (V, 4) or extrinsics.shape != (V, 4, 4):
            raise RuntimeError("expected depth [V,H,W], color [V,H,W,3], intrinsics [V,4], extrinsics [V,4,4]")
        depth_trunc = f(depth_trunc).expand(V).contiguous() if torch.as_tensor(depth_trunc).dim() == 0 else f(depth_trunc)
        origin = (ctypes.c_float * 3)(*self.origin)
        with torch.cuda.device(dev):
            _check(_lib().lara_tsdf_integrate(self.res, ctypes.byref(origin), self.voxel_length, self.sdf_trunc, V, H, W,
                                              depth.data_ptr(), color.data_ptr(), intrinsics.data_ptr(), extrinsics.data_ptr(),
                                              depth_trunc.data_ptr(), self.tsdf.data_ptr(), self.weight.data_ptr(),
                                              self.rgb.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "lara_tsdf_integrate")

    def integrate_render(self, cam, render_pkg, alpha_thres=0.08, depth_trunc=10.0):
        """One rendered view as `MeshExtractor.extract` prepares it (meshExtractor.py:76-108): pinhole intrinsics from
        the camera's field of view, depth zeroed where acc_map < alpha_thres, colour quantised to 8 bits."""
        H, W = int(cam.image_height), int(cam.image_width)
        K = torch.tensor([[W / (2 * math.tan(cam.FoVx / 2.0)), H / (2 * math.tan(cam.FoVy / 2.0)), W / 2, H / 2]])
        depth = render_pkg["depth"].detach().reshape(1, H, W)
        depth = torch.where(render_pkg["acc_map"].detach().reshape(1, H, W) < alpha_thres, torch.zeros_like(depth), depth)
        color = (render_pkg["image"].detach().reshape(1, H, W, 3) * 255).to(torch.uint8).float()
        self.integrate(depth, color, K, cam.world_view_transform.T.reshape(1, 4, 4), depth_trunc)

    def volume(self):
        """(tsdf [R,R,R], weight [R,R,R], rgb [R,R,R,3]) indexed [x][y][z]."""
        r = self.res
        return self.tsdf.view(r, r, r), self.weight.view(r, r, r), self.rgb.view(r, r, r, 3)
