"""Generates tests/golden/cameras_ref.npz with the REFERENCE's own `MiniCam` / `getProjectionMatrix`
(/root/reference/lightning/utils.py:5-48) on CPU: for seeded camera poses and the two (znear, zfar) conventions
LaRa uses (r -/+ 0.8, gobjverse.py:86; 0.5 / 2.5, google_scanned_objects.py:114), the four tensors the renderer
hands to the rasteriser (renderer_2dgs.py:119-137).  Run in the build container only:
    python tests/golden/make_cameras_fixture.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from lightning.utils import MiniCam, getProjectionMatrix  # noqa: E402  (the reference's code, unmodified)
from lara_amd.cameras import turntable_c2w  # noqa: E402  (poses only; the matrices below come from the reference)

g = torch.Generator().manual_seed(5)
poses = [m for m in turntable_c2w(8)] + [m for m in turntable_c2w(3, elevation_deg=25.0)]
# a few generic rigid poses (random rotation, translation of norm ~2)
for _ in range(4):
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    m = torch.eye(4)
    m[:3, :3] = q
    m[:3, 3] = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0) * (1.5 + torch.rand(1, generator=g))
    poses.append(m)
c2w = torch.stack(poses).float()
store = {"c2w": c2w.numpy()}
cases = [(512, 512, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8), (1024, 768, 0.6, 0.9, 0.5, 2.5)]
for ci, (W, H, fovx, fovy, zn, zf) in enumerate(cases):
    wvt, proj, full, ctr = [], [], [], []
    for m in c2w:
        # the reference passes tensors for fovx / fovy / znear / zfar (network.py:479-480,492)
        cam = MiniCam(m.clone(), W, H, torch.tensor(fovy), torch.tensor(fovx), torch.tensor(zn), torch.tensor(zf), "cpu")
        wvt.append(cam.world_view_transform); proj.append(cam.projection_matrix)
        full.append(cam.full_proj_transform); ctr.append(cam.camera_center)
    store[f"case{ci}/params"] = np.array([W, H, fovx, fovy, zn, zf], np.float64)
    store[f"case{ci}/world_view_transform"] = torch.stack(wvt).numpy()
    store[f"case{ci}/projection_matrix"] = torch.stack(proj).numpy()
    store[f"case{ci}/full_proj_transform"] = torch.stack(full).numpy()
    store[f"case{ci}/camera_center"] = torch.stack(ctr).numpy()
    store[f"case{ci}/P"] = getProjectionMatrix(torch.tensor(zn), torch.tensor(zf), torch.tensor(fovx), torch.tensor(fovy)).numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cameras_ref.npz"), **store)
print("wrote", {k: v.shape for k, v in store.items()})
