"""GPU TSDF fusion (include/lara_tsdf.h) against the numpy restatement of Open3D's per-voxel update
(oracle/tsdf_ref.py; Open3D itself is absent: parity unpinned by the reference).  Both sides run the same fp32
operation order without FMA contraction: weights exact, tsdf / colour within 1e-6 (division and sqrt are correctly
rounded on both sides; the bar leaves room for v_rcp-based sequences)."""
import math

import numpy as np
import pytest
import torch

from oracle import tsdf_ref


def _sphere_views(V, H, W, radius=0.3, dist=1.5, fov=0.75):
    """Depth maps of a sphere at the origin seen from a ring of cameras (analytic), colours by view."""
    f = 0.5 * W / math.tan(0.5 * fov)
    K = np.tile(np.array([f, f * H / W * (W / H), W / 2, H / 2], np.float32), (V, 1))
    E = np.zeros((V, 4, 4), np.float32)
    depth = np.zeros((V, H, W), np.float32)
    color = np.zeros((V, H, W, 3), np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    for v in range(V):
        a = 2 * math.pi * v / V
        cpos = np.array([dist * math.cos(a), dist * math.sin(a), 0.3], np.float64)
        fwd = -cpos / np.linalg.norm(cpos)
        right = np.cross(fwd, [0, 0, 1.0]); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])            # world -> camera rows
        E[v, :3, :3] = R
        E[v, :3, 3] = -R @ cpos
        E[v, 3, 3] = 1
        dirs = np.stack([(xs - K[v, 2]) / K[v, 0], (ys - K[v, 3]) / K[v, 1], np.ones_like(xs, float)], -1)  # camera space, z = 1
        o = E[v, :3, 3].astype(np.float64)           # sphere centre in camera space = R(0 - cpos)
        b = (dirs * o).sum(-1)
        disc = b * b - (dirs * dirs).sum(-1) * (o @ o - radius * radius)
        hit = disc > 0
        tz = np.where(hit, (b - np.sqrt(np.maximum(disc, 0))) / (dirs * dirs).sum(-1), 0)
        depth[v] = np.where(hit, tz, 0).astype(np.float32)
        color[v] = np.floor(np.array([40 + 20 * v, 200 - 15 * v, 90.0]))
    return depth, color, K, E


def test_oracle_recovers_the_sphere():
    """Known answer: after fusing a ring of views, the zero crossing of the TSDF along +x sits at the sphere's radius."""
    res, vl = 48, 1.0 / 48
    depth, color, K, E = _sphere_views(6, 64, 64)
    tsdf, weight, rgb = tsdf_ref.integrate(res, (-0.5, -0.5, -0.5), vl, 4 * vl, depth, color, K, E, np.full(6, 10.0, np.float32))
    t = tsdf.reshape(res, res, res)[:, res // 2, res // 2]
    w = weight.reshape(res, res, res)[:, res // 2, res // 2]
    xs = -0.5 + vl * (np.arange(res) + 0.5)
    i = max(j for j in range(res - 1) if w[j] > 0 and w[j + 1] > 0 and t[j] < 0 <= t[j + 1])   # inside -> outside along +x
    x0 = xs[i] + (xs[i + 1] - xs[i]) * (-t[i]) / (t[i + 1] - t[i])
    assert abs(x0 - 0.3) < 0.6 * vl, x0
    assert weight.max() <= 6 and (weight > 0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("res,V,H,W", [(40, 5, 48, 64), (33, 1, 32, 32), (64, 8, 96, 96)])
def test_hip_tsdf_matches_the_oracle(res, V, H, W):
    from lara_amd.tsdf import TSDFVolume
    vl = 1.0 / res
    depth, color, K, E = _sphere_views(V, H, W)
    trunc = np.linspace(1.35, 10.0, V).astype(np.float32)    # the first views lose their far pixels to depth_trunc
    rt, rw, rc = tsdf_ref.integrate(res, (-0.5, -0.5, -0.5), vl, 3 * vl, depth, color, K, E, trunc)
    vol = TSDFVolume((-0.5, -0.5, -0.5), vl, 3 * vl, res)
    # two calls (3 views, then the rest): the running averages continue across calls, as with Open3D's integrate()
    k = min(3, V)
    vol.integrate(depth[:k], color[:k], K[:k], E[:k], trunc[:k])
    if V > k:
        vol.integrate(depth[k:], color[k:], K[k:], E[k:], trunc[k:])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(vol.weight.cpu().numpy(), rw)
    assert np.abs(vol.tsdf.cpu().numpy() - rt).max() <= 1e-6
    assert np.abs(vol.rgb.cpu().numpy() - rc).max() <= 1e-3      # colours live in 0..255
    assert (rw > 0).sum() > 100


@pytest.mark.gpu
def test_hip_tsdf_argument_errors():
    from lara_amd.tsdf import TSDFVolume
    with pytest.raises(RuntimeError, match="no CPU path"):
        TSDFVolume((0, 0, 0), 0.1, 0.2, 8, device="cpu")
    vol = TSDFVolume((0, 0, 0), 0.1, 0.2, 8)
    with pytest.raises(RuntimeError, match="expected depth"):
        vol.integrate(torch.zeros(2, 4, 4), torch.zeros(2, 4, 4, 3), torch.zeros(1, 4), torch.zeros(2, 4, 4), 10.0)


@pytest.mark.gpu
def test_integrate_render_prepares_a_view_as_the_mesh_extractor_does():
    """`integrate_render` = tools/meshExtractor.py:76-108 for one rendered view: pinhole intrinsics from the camera's
    field of view, depth zeroed where acc_map < alpha_thres, colour quantised to 8 bits, extrinsic = world_view^T."""
    from lara_amd import cameras
    from lara_amd.tsdf import TSDFVolume
    H = W = 64
    cam = cameras.make_cameras(cameras.turntable_c2w(4)[1:2], W, H, 0.75, 0.75, 0.5, 2.5, device="cuda")[0]
    g = torch.Generator().manual_seed(3)
    depth = 1.2 + 0.8 * torch.rand(H, W, 1, generator=g)
    acc = torch.rand(H, W, generator=g)
    img = torch.rand(H, W, 3, generator=g)
    pkg = {"depth": depth.cuda(), "acc_map": acc.cuda(), "image": img.cuda()}
    res, vl = 32, 1.0 / 32
    vol = TSDFVolume((-0.5, -0.5, -0.5), vl, 3 * vl, res)
    vol.integrate_render(cam, pkg, alpha_thres=0.3, depth_trunc=10.0)
    torch.cuda.synchronize()
    f = W / (2 * math.tan(0.75 / 2))
    K = np.array([[f, f, W / 2, H / 2]], np.float32)
    E = cam.world_view_transform.T.cpu().numpy().reshape(1, 4, 4)
    d = depth.numpy().reshape(1, H, W).copy()
    d[acc.numpy().reshape(1, H, W) < 0.3] = 0
    c = np.floor(img.numpy().reshape(1, H, W, 3) * 255).astype(np.uint8).astype(np.float32)
    rt, rw, rc = tsdf_ref.integrate(res, (-0.5, -0.5, -0.5), vl, 3 * vl, d, c, K, E, np.array([10.0], np.float32))
    np.testing.assert_array_equal(vol.weight.cpu().numpy(), rw)
    assert np.abs(vol.tsdf.cpu().numpy() - rt).max() <= 1e-6
    assert np.abs(vol.rgb.cpu().numpy() - rc).max() <= 1e-3
    assert (rw > 0).sum() > 50
