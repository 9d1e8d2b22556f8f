"""Generates tests/golden/rays_ref.npz with the REFERENCE's own build_rays and fov_to_ixt
(/root/reference/dataLoader/utils.py:21-34, dataLoader/gobjverse.py:10-15).  Run in the build container
only:  python tests/golden/make_rays_fixture.py"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for name in ("h5py", "cv2"):  # imported by the loader module for unrelated code paths
    sys.modules.setdefault(name, types.ModuleType(name))
# the package's __init__ imports every dataset (imageio, PIL, ...): register a bare package so that only
# the two modules needed here are executed
pkg = types.ModuleType("dataLoader")
pkg.__path__ = ["/root/reference/dataLoader"]
sys.modules["dataLoader"] = pkg
sys.path.insert(0, "/root/reference")
from dataLoader.utils import build_rays  # noqa: E402
from dataLoader.gobjverse import fov_to_ixt  # noqa: E402

rng = np.random.default_rng(7)
V, H, W = 3, 48, 32
c2ws = np.tile(np.eye(4, dtype=np.float32), (V, 1, 1))
for v in range(V):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    c2ws[v, :3, :3] = q.astype(np.float32)
    c2ws[v, :3, 3] = rng.normal(size=3).astype(np.float32) * 2
ixts = np.stack([fov_to_ixt(np.array([0.6 + 0.1 * v, 0.5 + 0.1 * v], np.float32), np.array([W, H])) for v in range(V)])
ixts[:, 0, 1] = rng.normal(size=V).astype(np.float32) * 0.01  # a little skew: the general 3x3 inverse
rays = build_rays(c2ws, ixts.copy(), H, W, 1.0)
rays_down = build_rays(c2ws, ixts.copy(), H, W, 1.0 / 16)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rays_ref.npz"), c2ws=c2ws, ixts=ixts, H=H, W=W,
                    rays=rays, rays_down=rays_down)
print("wrote", rays.shape, rays_down.shape, rays.dtype)
