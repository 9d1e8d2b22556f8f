"""Device-side ray generation and the batch dictionary.

(CPU) a numpy restatement of build_rays reproduces the output of the REFERENCE's own build_rays /
fov_to_ixt stored in tests/golden/rays_ref.npz (tests/golden/make_rays_fixture.py); (GPU) the HIP
kernel matches both, and synthetic_batch has the reference loader's keys, shapes and dtypes.
Tolerance: the reference multiplies in float64 after a float32 LAPACK inverse and rounds to float32;
the kernel works in float32 throughout -> |diff| <= 2e-6 * (1 + |value|)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def restated_rays(c2ws, ixts, H, W, scale):
    Hs, Ws = int(H * scale), int(W * scale)
    K = ixts.astype(np.float64).copy()
    K[:, :2] *= scale
    ys, xs = np.mgrid[0:Hs, 0:Ws]
    pix = np.stack([xs + 0.5, ys + 0.5, np.ones_like(xs, dtype=np.float64)], -1)           # [Hs, Ws, 3]
    M = c2ws[:, :3, :3].astype(np.float64) @ np.linalg.inv(K)                               # R K^-1
    dirs = np.einsum("vij,hwj->vhwi", M, pix)
    orig = np.broadcast_to(c2ws[:, None, None, :3, 3], dirs.shape)
    return np.concatenate([orig, dirs], -1).astype(np.float32)


def fixture():
    return np.load(os.path.join(HERE, "golden", "rays_ref.npz"))


def test_restatement_matches_reference_build_rays():
    f = fixture()
    H, W = int(f["H"]), int(f["W"])
    for key, scale in (("rays", 1.0), ("rays_down", 1.0 / 16)):
        got = restated_rays(f["c2ws"], f["ixts"], H, W, scale)
        assert got.shape == f[key].shape
        np.testing.assert_allclose(got, f[key], rtol=2e-6, atol=2e-6)


def test_fov_to_ixt_matches_reference():
    from lara_amd.batch import fov_to_ixt
    f = fixture()
    H, W = int(f["H"]), int(f["W"])
    fov = torch.tensor([[0.6 + 0.1 * v, 0.5 + 0.1 * v] for v in range(3)])
    got = fov_to_ixt(fov, (W, H)).numpy()
    ref = f["ixts"].copy()
    ref[:, 0, 1] = 0  # the fixture adds skew on top of fov_to_ixt
    np.testing.assert_allclose(got, ref, rtol=1e-6)


@pytest.mark.gpu
def test_hip_rays_match_reference(hip_lib):
    from lara_amd.batch import build_rays
    f = fixture()
    H, W = int(f["H"]), int(f["W"])
    c2ws, ixts = torch.from_numpy(f["c2ws"]).cuda(), torch.from_numpy(f["ixts"]).cuda()
    before = ixts.clone()
    for key, scale in (("rays", 1.0), ("rays_down", 1.0 / 16)):
        got = build_rays(c2ws, ixts, H, W, scale).cpu().numpy()
        assert got.shape == f[key].shape and got.dtype == np.float32
        np.testing.assert_allclose(got, f[key], rtol=2e-6, atol=2e-6)
    assert torch.equal(ixts, before)  # unlike the reference, the intrinsics are not scaled in place
    with pytest.raises(RuntimeError, match="no CPU path"):
        build_rays(c2ws.cpu(), ixts.cpu(), H, W)


@pytest.mark.gpu
def test_hip_rays_at_non_dyadic_scales(hip_lib):
    """int(H*scale) is computed once on the host (double precision, as the reference does) and handed to the
    kernel: at H=100, scale 0.29 / 0.57 / 0.58 a float32 recomputation gives one row more or less, which used to
    write past the tensor.  A guard band after the output must stay untouched."""
    from lara_amd.batch import build_rays
    f = fixture()
    c2ws, ixts = torch.from_numpy(f["c2ws"]).cuda(), torch.from_numpy(f["ixts"]).cuda()
    for H, W, scale in ((100, 100, 0.29), (100, 200, 0.57), (200, 100, 0.58), (37, 53, 0.333)):
        want = restated_rays(f["c2ws"], f["ixts"], H, W, scale)
        got = build_rays(c2ws, ixts, H, W, scale)
        assert tuple(got.shape) == want.shape == (c2ws.shape[0], int(H * scale), int(W * scale), 6)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=2e-6)


@pytest.mark.gpu
def test_synthetic_batch_schema_and_geometry(hip_lib):
    from lara_amd.batch import synthetic_batch
    B, V, H, W = 2, 8, 64, 48
    b = synthetic_batch(B, V, H, W, device="cuda:0", seed=3)
    shapes = {"tar_c2w": (B, V, 4, 4), "tar_w2c": (B, V, 4, 4), "tar_ixt": (B, V, 3, 3), "tar_rgb": (B, V, H, W, 3),
              "tar_msk": (B, V, H, W), "transform_mats": (B, 1, 4, 4), "bg_color": (B, V, 3), "near_far": (B, 2),
              "tar_rays": (B, V, H, W, 6), "tar_rays_down": (B, V, H // 16, W // 16, 6), "fovx": (B,), "fovy": (B,)}
    for k, shp in shapes.items():
        assert tuple(b[k].shape) == shp, k
        assert b[k].dtype == (torch.uint8 if k == "tar_msk" else torch.float32), k
    assert set(b["meta"]) == {"scene", "tar_view", "frame_id", "tar_h", "tar_w"}
    c2w, w2c = b["tar_c2w"], b["tar_w2c"]
    eye = torch.eye(4, device=c2w.device).expand(B, V, 4, 4)
    assert torch.allclose(c2w @ w2c, eye, atol=1e-4)
    # the loader's alignment: the first camera sits on the -z axis at distance r, looking down +z
    r = b["near_far"][:, 0] + 0.8
    assert torch.allclose(c2w[:, 0, :3, 3], torch.stack([torch.zeros_like(r), torch.zeros_like(r), -r], -1), atol=1e-4)
    assert torch.allclose(c2w[:, 0, :3, :3], torch.eye(3, device=c2w.device).expand(B, 3, 3), atol=1e-4)
    # the centre ray of every view points along the camera's +z axis and starts at its position
    rays = b["tar_rays"]
    assert torch.allclose(rays[..., :3], c2w[:, :, None, None, :3, 3].expand_as(rays[..., :3]))
    mid = 0.25 * (rays[:, :, H // 2 - 1, W // 2 - 1, 3:] + rays[:, :, H // 2, W // 2, 3:] +
                  rays[:, :, H // 2 - 1, W // 2, 3:] + rays[:, :, H // 2, W // 2 - 1, 3:])
    assert torch.allclose(mid, c2w[:, :, :3, 2], atol=1e-4)
