"""GPU TSDF fusion (include/lara_tsdf.h) against the numpy restatement of Open3D's per-voxel update
(oracle/tsdf_ref.py; Open3D itself is absent: parity unpinned by the reference).  Both sides run the same fp32
operation order without FMA contraction: weights exact, tsdf / colour within 1e-6 (division and sqrt are correctly
rounded on both sides; the bar leaves room for v_rcp-based sequences)."""
import math

import numpy as np
import pytest
import torch

from oracle import tsdf_ref


def _sphere_views(V, H, W, radius=0.3, dist=1.5, fov=0.75, poles=False):
    """Depth maps of a sphere at the origin seen from a ring of cameras (analytic), colours by view.  `poles`: cameras spread
    over the whole sphere of directions instead (golden-angle spiral): with enough of them every voxel next to the surface
    is observed and the fused surface closes."""
    f = 0.5 * W / math.tan(0.5 * fov)
    K = np.tile(np.array([f, f * H / W * (W / H), W / 2, H / 2], np.float32), (V, 1))
    E = np.zeros((V, 4, 4), np.float32)
    depth = np.zeros((V, H, W), np.float32)
    color = np.zeros((V, H, W, 3), np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    for v in range(V):
        a = 2 * math.pi * v / V
        cpos = np.array([dist * math.cos(a), dist * math.sin(a), 0.3], np.float64)
        up = [0, 0, 1.0]
        if poles:
            zc = 1 - 2 * (v + 0.5) / V
            cpos = dist * np.array([math.sqrt(1 - zc * zc) * math.cos(2.399963 * v), math.sqrt(1 - zc * zc) * math.sin(2.399963 * v), zc])
            up = [0, 1.0, 0] if abs(zc) > 0.9 else up
        fwd = -cpos / np.linalg.norm(cpos)
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
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
        color[v] = np.floor(np.array([40 + 5 * v, 240 - 5 * v, 90.0]))
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
    vol = TSDFVolume((-0.5, -0.5, -0.5), vl, 3 * vl, res, block_sparse=False)
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
        TSDFVolume((0, 0, 0), 0.1, 0.2, 16, device="cpu")
    vol = TSDFVolume((0, 0, 0), 0.1, 0.2, 8, block_sparse=False)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        TSDFVolume((0, 0, 0), 0.1, 0.2, 8)
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
    vol = TSDFVolume((-0.5, -0.5, -0.5), vl, 3 * vl, res, block_sparse=False)
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


# ---- block-sparse integration (ScalableTSDFVolume's semantics) and mesh extraction ----------------------------------------
def _mesh_checks(verts, tris, radius, vl):
    """Closed, consistently oriented, genus 0, on the sphere."""
    from collections import Counter
    r = np.linalg.norm(verts, axis=1)
    assert np.abs(r - radius).max() < 1.0 * vl, np.abs(r - radius).max()      # (projective TSDF on a grid: within a voxel)
    und, dirc = Counter(), Counter()
    for t in tris:
        for i in range(3):
            a, b = int(t[i]), int(t[(i + 1) % 3])
            und[(min(a, b), max(a, b))] += 1
            dirc[(a, b)] += 1
    assert set(und.values()) == {2} and set(dirc.values()) == {1}          # watertight, no orientation clash
    assert len(verts) - len(und) + len(tris) == 2                            # Euler characteristic of a sphere
    v0, v1, v2 = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
    n = np.cross(v1 - v0, v2 - v0)
    assert ((n * (v0 + v1 + v2)).sum(1) > 0).all()                           # normals point outwards (towards growing tsdf)
    area = 0.5 * np.linalg.norm(n, axis=1).sum()
    assert abs(area / (4 * math.pi * radius ** 2) - 1) < 0.03, area


def _weld(verts, keys):
    uniq, inv = np.unique(keys.reshape(-1), return_inverse=True)
    first = np.full(len(uniq), len(inv), np.int64)
    np.minimum.at(first, inv, np.arange(len(inv)))
    return verts.reshape(-1, 3)[first], inv.reshape(-1, 3)


def test_oracle_block_sparse_semantics_and_mesh_of_the_sphere():
    """The restated ScalableTSDFVolume: a view integrates only into blocks near its depth samples -- the free space in front
    of the surface, which the dense (Uniform) volume fills with tsdf = 1, stays unobserved -- and the extracted mesh of the
    fused sphere is a closed genus-0 surface at the sphere's radius."""
    res, vl, radius, org = 64, 2.0 / 64, 0.3, (-1.0, -1.0, -1.0)
    depth, color, K, E = _sphere_views(24, 64, 64, radius=radius, poles=True)
    trunc = np.full(24, 10.0, np.float32)
    args = (res, org, vl, 3 * vl, depth, color, K, E, trunc)
    touched = tsdf_ref.touched_blocks(res, org, vl, 3 * vl, depth, K, E, trunc)
    dt, dw, dc = tsdf_ref.integrate(*args)
    st, sw, sc = tsdf_ref.integrate(*args, touched=touched)
    assert 0 < (sw > 0).sum() < 0.8 * (dw > 0).sum()
    both = sw == dw                                           # where every view that saw the voxel also touched its block ...
    np.testing.assert_array_equal(st[both], dt[both])         # ... the two volumes hold the same bits
    # a voxel half-way between camera 0 and the sphere: observed by the dense volume, never allocated by the sparse one
    cam0 = -E[0, :3, :3].T.astype(np.float64) @ E[0, :3, 3].astype(np.float64)
    ix = np.floor((cam0 * 0.55 - np.array(org)) / vl).astype(int)
    i = (ix[0] * res + ix[1]) * res + ix[2]
    assert dw[i] > 0 and dt[i] == 1.0 and sw[i] == 0 and not touched[:, ix[0] // 16, ix[1] // 16, ix[2] // 16].any()
    verts, cols, keys = tsdf_ref.extract_mesh(res, org, vl, st, sw, sc)
    assert len(verts) > 500 and cols.min() >= 0 and cols.max() <= 1
    _mesh_checks(*_weld(verts, keys), radius, vl)
    # the dense volume's free-space shell does not move the surface: its zero crossings sit on the sphere as well
    dverts, _, _ = tsdf_ref.extract_mesh(res, org, vl, dt, dw, dc)
    assert np.abs(np.linalg.norm(dverts.reshape(-1, 3), axis=1) - radius).max() < 1.0 * vl


@pytest.mark.gpu
@pytest.mark.parametrize("res,V,H,W", [(48, 6, 64, 64), (32, 1, 48, 40), (64, 9, 96, 96)])
def test_hip_block_sparse_tsdf_matches_the_oracle(res, V, H, W):
    from lara_amd.tsdf import TSDFVolume
    vl, org = 2.0 / res, (-1.0, -1.0, -1.0)
    depth, color, K, E = _sphere_views(V, H, W)
    trunc = np.linspace(1.35, 10.0, V).astype(np.float32)
    touched = tsdf_ref.touched_blocks(res, org, vl, 3 * vl, depth, K, E, trunc)
    rt, rw, rc = tsdf_ref.integrate(res, org, vl, 3 * vl, depth, color, K, E, trunc, touched=touched)
    vol = TSDFVolume(org, vl, 3 * vl, res)
    k = min(4, V)
    vol.integrate(depth[:k], color[:k], K[:k], E[:k], trunc[:k])
    np.testing.assert_array_equal(vol.last_touched.cpu().numpy().astype(bool), touched[:k].reshape(k, -1))
    if V > k:
        vol.integrate(depth[k:], color[k:], K[k:], E[k:], trunc[k:])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(vol.allocated.cpu().numpy().astype(bool), touched.any(0).reshape(-1))
    np.testing.assert_array_equal(vol.weight.cpu().numpy(), rw)
    assert np.abs(vol.tsdf.cpu().numpy() - rt).max() <= 1e-6
    assert np.abs(vol.rgb.cpu().numpy() - rc).max() <= 1e-3
    assert 100 < (rw > 0).sum() < res ** 3 // 2
    if V == 1:      # one view: on the blocks it touched, the dense volume holds the same bits; elsewhere the sparse one is empty
        dense = TSDFVolume(org, vl, 3 * vl, res, block_sparse=False)
        dense.integrate(depth, color, K, E, trunc)
        nb = res // 16
        in_block = torch.from_numpy(touched[0]).cuda().repeat_interleave(16, 0).repeat_interleave(16, 1).repeat_interleave(16, 2).reshape(-1)
        assert torch.equal(vol.tsdf[in_block], dense.tsdf[in_block]) and torch.equal(vol.weight[in_block], dense.weight[in_block])
        assert torch.equal(vol.rgb[in_block], dense.rgb[in_block]) and not vol.weight[~in_block].any()
        assert int((dense.weight > 0).sum()) >= int((vol.weight > 0).sum())


@pytest.mark.gpu
def test_hip_mesh_extraction_matches_the_oracle_and_the_sphere():
    from lara_amd.tsdf import TSDFVolume
    res, radius, org = 48, 0.3, (-0.75, -0.75, -0.75)
    vl = 1.5 / res
    depth, color, K, E = _sphere_views(24, 64, 64, radius=radius, poles=True)
    vol = TSDFVolume(org, vl, 3 * vl, res)
    vol.integrate(depth, color, K, E, 10.0)
    ov, oc, ok = tsdf_ref.extract_mesh(res, org, vl, vol.tsdf.cpu().numpy(), vol.weight.cpu().numpy(), vol.rgb.cpu().numpy())
    v, t, c = vol.extract_triangle_mesh(weld=False)
    assert v.shape == (3 * len(ov), 3) and np.abs(v.cpu().numpy().reshape(-1, 3, 3) - ov).max() <= 1e-6      # same triangles, same order
    assert np.abs(c.cpu().numpy().reshape(-1, 3, 3) - oc).max() <= 1e-5
    wv, wt, wc = vol.extract_triangle_mesh()
    want_v, want_t = _weld(ov, ok)
    assert wv.shape == want_v.shape and np.array_equal(wt.cpu().numpy(), want_t) and np.abs(wv.cpu().numpy() - want_v).max() <= 1e-6
    _mesh_checks(wv.cpu().numpy(), wt.cpu().numpy(), radius, vl)
    # a finer volume, device only: still one closed genus-0 surface on the sphere
    res = 96
    vl = 1.0 / res
    depth, color, K, E = _sphere_views(40, 160, 160, radius=radius, poles=True)
    vol = TSDFVolume((-0.5, -0.5, -0.5), vl, 3 * vl, res)
    vol.integrate(depth, color, K, E, 10.0)
    v, t, c = vol.extract_triangle_mesh()
    _mesh_checks(v.cpu().numpy(), t.cpu().numpy(), radius, vl)
    empty = TSDFVolume((-0.5, -0.5, -0.5), vl, 3 * vl, 32).extract_triangle_mesh()
    assert empty[0].shape == (0, 3) and empty[1].shape == (0, 3)


def test_marching_cubes_table_is_the_generators_output_and_watertight():
    """csrc/mc_tables.h is generated (tools/gen_mc_tables.py): the committed header equals what the generator derives, every
    case has at most 5 triangles, complementary cases produce the same vertices, and on a random field with a positive shell
    every edge of the extracted surface is shared by exactly two triangles with opposite directions."""
    import subprocess
    import sys
    from collections import Counter
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    assert subprocess.run([sys.executable, __import__("os").path.join(root, "tools", "gen_mc_tables.py"), "--check"]).returncode == 0
    sys.path.insert(0, __import__("os").path.join(root, "tools"))
    import gen_mc_tables as g
    table = g.build()
    assert max(len(t) for t in table) == 5 and not table[0] and not table[255]
    for c in range(256):
        assert sorted({e for t in table[c] for e in t}) == sorted({e for t in table[255 - c] for e in t})
    rng = np.random.default_rng(3)
    R = 10
    f = rng.normal(size=(R, R, R)).astype(np.float32)
    f[0] = f[-1] = 1; f[:, 0] = f[:, -1] = 1; f[:, :, 0] = f[:, :, -1] = 1
    verts, _, keys = tsdf_ref.extract_mesh(R, (0.0, 0.0, 0.0), 1.0, f.reshape(-1), np.ones(R ** 3, np.float32), np.zeros((R ** 3, 3), np.float32))
    und, dirc = Counter(), Counter()
    for t in keys:
        for i in range(3):
            a, b = int(t[i]), int(t[(i + 1) % 3])
            und[(min(a, b), max(a, b))] += 1
            dirc[(a, b)] += 1
    assert len(keys) > 200 and set(und.values()) == {2} and set(dirc.values()) == {1}
