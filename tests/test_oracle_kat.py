"""Known-answer tests that pin the CPU oracle to closed-form results (the
reference ships no golden vectors for the rasteriser, so these are self-made from the published
2DGS equations and the call-site contract of lightning/renderer_2dgs.py:226-242)."""
import math

import numpy as np
import pytest
import torch

import oracle
from lara_amd import cameras
from tests.helpers import oracle_view

C0 = 0.28209479177387814
C1 = 0.4886025119029199


def front_camera(W=33, H=33, fov=0.75):
    """Camera at the origin looking down +z (view space == world space)."""
    return cameras.make_cameras(torch.eye(4)[None], W, H, fov, fov, 0.5, 2.5)[0]


def sh_for_rgb(rgb):
    sh = np.zeros((1, 4, 3), np.float32)
    sh[0, 0] = (np.asarray(rgb, np.float32) - 0.5) / C0
    return sh


def quat_identity(n=1):
    q = np.zeros((n, 4), np.float32)
    q[:, 0] = 1
    return q


def test_fronto_parallel_single_splat_matches_closed_form():
    W = H = 33
    cam = front_camera(W, H)
    tan = math.tan(0.375)
    z0, s, o = 1.5, 0.2, 0.8
    rgb, bg = np.array([0.9, 0.4, 0.1]), np.array([0.2, 0.3, 1.0])
    v = oracle_view(cam, bg, sh_degree=0)
    res = oracle.forward(v, np.array([[0, 0, z0]], np.float32), np.array([o], np.float32),
                         shs=sh_for_rgb(rgb), scales=np.array([[s, s]], np.float32),
                         rotations=quat_identity())
    assert res.radii[0] > 0
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    # pixel -> ndc -> point on the plane z = z0 -> local (u, v)
    ndc_x = (2 * xs + 1) / W - 1
    ndc_y = (2 * ys + 1) / H - 1
    u = ndc_x * tan * z0 / s
    vv = ndc_y * tan * z0 / s
    alpha = np.minimum(0.99, o * np.exp(-0.5 * (u * u + vv * vv)))
    alpha = np.where(alpha < 1 / 255, 0, alpha)
    exp_color = alpha[None] * rgb[:, None, None] + (1 - alpha)[None] * bg[:, None, None]
    np.testing.assert_allclose(res.color, exp_color, atol=2e-5)
    np.testing.assert_allclose(res.allmap[1], alpha, atol=2e-6)          # accumulated alpha
    np.testing.assert_allclose(res.allmap[0], alpha * z0, atol=1e-5)     # un-normalised depth
    np.testing.assert_allclose(res.allmap[2:5], alpha[None] * np.array([0, 0, -1.0])[:, None, None], atol=2e-6)
    np.testing.assert_allclose(res.allmap[5], np.where(alpha > 0, z0, 0), atol=1e-5)  # median depth
    np.testing.assert_allclose(res.allmap[6], 0, atol=1e-7)               # one splat: no distortion
    # the count of blended (pixel, splat) pairs -- the path's useful work, bench.py's `valu_useful_frac` -- is here
    # the number of pixels the one splat reaches
    assert res.blended_pairs == int((alpha > 0).sum()) == int((res.n_contrib[0] > 0).sum())
    assert res.n_contrib[0].max() == 1


def test_two_splats_composite_front_to_back():
    cam = front_camera()
    v = oracle_view(cam, [0, 0, 0], sh_degree=0)
    means = np.array([[0, 0, 2.0], [0, 0, 1.0]], np.float32)  # deliberately back first
    o = np.array([0.5, 0.6], np.float32)
    shs = np.concatenate([sh_for_rgb([1, 0, 0]), sh_for_rgb([0, 1, 0])])
    res = oracle.forward(v, means, o, shs=shs, scales=np.full((2, 2), 0.5, np.float32),
                         rotations=quat_identity(2))
    c = 16  # centre pixel of a 33x33 image is exactly on the optical axis
    a_front, a_back = 0.6, 0.5
    np.testing.assert_allclose(res.color[:, c, c], [a_back * (1 - a_front), a_front, 0], atol=1e-5)
    np.testing.assert_allclose(res.allmap[1, c, c], 1 - (1 - a_front) * (1 - a_back), atol=1e-6)
    np.testing.assert_allclose(res.allmap[0, c, c], a_front * 1.0 + (1 - a_front) * a_back * 2.0, atol=1e-5)
    # sorted list of the centre tile starts with the nearer surfel (id 1)
    tile = (c // 16) * 3 + (c // 16)
    r0, r1 = res.ranges[tile]
    assert list(res.point_list[r0:r1]) == [1, 0]
    # distortion = w0*w1*(m0-m1)^2 with m = far/(far-near) * (1 - near/z)
    m = lambda z: 100.0 / (100.0 - 0.2) * (1 - 0.2 / z)
    w0, w1 = a_front, (1 - a_front) * a_back
    np.testing.assert_allclose(res.allmap[6, c, c], w0 * w1 * (m(1.0) - m(2.0)) ** 2, rtol=1e-4)
    # median depth: last splat composited while T > 0.5 -> the front one only (T after it = 0.4)
    assert res.allmap[5, c, c] == pytest.approx(1.0, abs=1e-5)


def test_near_plane_cull_and_behind_camera():
    cam = front_camera()
    v = oracle_view(cam, [1, 1, 1], sh_degree=0)
    means = np.array([[0, 0, 0.2], [0, 0, 0.21], [0, 0, -1.0]], np.float32)
    res = oracle.forward(v, means, np.full(3, 0.5, np.float32), shs=np.repeat(sh_for_rgb([0, 0, 0]), 3, 0),
                         scales=np.full((3, 2), 0.01, np.float32), rotations=quat_identity(3))
    assert res.radii[0] == 0 and res.radii[2] == 0 and res.radii[1] > 0
    assert list(oracle.mark_visible(means, v.viewmatrix)) == [False, True, False]


def test_transmittance_termination():
    cam = front_camera()
    v = oracle_view(cam, [0, 0, 0], sh_degree=0)
    n = 12
    means = np.stack([np.zeros(n), np.zeros(n), 1.0 + 0.05 * np.arange(n)], 1).astype(np.float32)
    res = oracle.forward(v, means, np.full(n, 0.95, np.float32), shs=np.repeat(sh_for_rgb([1, 1, 1]), n, 0),
                         scales=np.full((n, 2), 0.5, np.float32), rotations=quat_identity(n))
    # alpha = 0.95 -> T after k splats = 0.05^k; the splat that would push T below 1e-4 is not
    # composited: 0.05^3 = 1.25e-4 stays, 0.05^4 is refused  -> 3 contributors
    assert res.n_contrib[0, 16, 16] == 3
    assert res.final_T[0, 16, 16] == pytest.approx(0.05 ** 3, rel=1e-4)


def test_low_pass_branch_for_subpixel_splat():
    W = H = 33
    cam = front_camera(W, H)
    v = oracle_view(cam, [0, 0, 0], sh_degree=0)
    z0, o = 1.0, 0.9
    res = oracle.forward(v, np.array([[0, 0, z0]], np.float32), np.array([o], np.float32),
                         shs=sh_for_rgb([1, 1, 1]), scales=np.full((1, 2), 1e-5, np.float32),
                         rotations=quat_identity())
    assert res.radii[0] == 3  # ceil(3 * 0.707106)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    d2 = (xs - 16) ** 2 + (ys - 16) ** 2
    alpha = o * np.exp(-d2)          # rho2d = 2 d^2 -> G = exp(-d^2)
    alpha = np.where(alpha < 1 / 255, 0, alpha)
    np.testing.assert_allclose(res.allmap[1], alpha, atol=2e-6)
    np.testing.assert_allclose(res.allmap[0], alpha * z0, atol=2e-6)  # low-pass depth = Tw.z


def test_sh_degree1_colour_and_clamp():
    cam = front_camera()
    v = oracle_view(cam, [0, 0, 0], sh_degree=1)
    # reference hands campos = -c2w[:3,3] (lightning/utils.py:48); here the camera is at the origin
    pos = np.array([[0.3, -0.2, 1.0]], np.float32)
    rng = np.random.default_rng(0)
    sh = rng.normal(0, 0.5, (1, 4, 3)).astype(np.float32)
    sh[0, 0, 2] = -5.0  # force a clamp on the blue channel
    res = oracle.forward(v, pos, np.array([0.5], np.float32), shs=sh, scales=np.full((1, 2), 0.1, np.float32),
                         rotations=quat_identity())
    d = pos[0] / np.linalg.norm(pos[0])
    exp = C0 * sh[0, 0] - C1 * d[1] * sh[0, 1] + C1 * d[2] * sh[0, 2] - C1 * d[0] * sh[0, 3] + 0.5
    np.testing.assert_allclose(res.rgb[0], np.maximum(exp, 0), atol=1e-6)
    assert list(res.clamped[0]) == [0, 0, 1]


def test_tile_rect_and_binning_invariants():
    from tests.helpers import small_scene, to_numpy, run_oracle
    act, cams = small_scene(grid=8, size=80, seed=3, scale_boost=2.0)
    v = oracle_view(cams[2], [1, 1, 1])
    res = run_oracle(v, to_numpy(act))
    gx = gy = 5
    vis = res.radii > 0
    # rect from centre +- radius, C truncation toward zero, clamped to the grid
    p, r = res.means2D[vis], res.radii[vis][:, None].astype(np.float32)
    lo = np.clip(np.trunc((p - r) / 16), 0, gx).astype(np.uint32)
    hi = np.clip(np.trunc((p + r + 15) / 16), 0, gx).astype(np.uint32)
    np.testing.assert_array_equal(res.rect[vis][:, :2], lo)
    np.testing.assert_array_equal(res.rect[vis][:, 2:], hi)
    area = (res.rect[:, 2] - res.rect[:, 0]) * (res.rect[:, 3] - res.rect[:, 1])
    np.testing.assert_array_equal(res.tiles_touched, np.where(vis, area, 0))
    assert res.num_rendered == int(res.tiles_touched.sum()) == int(res.point_offsets[-1])
    # keys sorted; ranges partition the list by tile id; inside a tile (depth bits, id) ascending
    assert np.all(np.diff(res.keys_sorted.astype(np.uint64)) >= 0) or np.all(res.keys_sorted[1:] >= res.keys_sorted[:-1])
    tiles_of = (res.keys_sorted >> np.uint64(32)).astype(np.int64)
    for t in range(gx * gy):
        r0, r1 = res.ranges[t]
        idx = np.nonzero(tiles_of == t)[0]
        if len(idx) == 0:
            assert (r0, r1) == (0, 0)
            continue
        assert (r0, r1) == (idx[0], idx[-1] + 1)
        ids = res.point_list[r0:r1].astype(np.int64)
        comp = (res.depths[ids].view(np.uint32).astype(np.int64) << 32) | ids
        assert np.all(np.diff(comp) > 0)
