"""Row R11 of SURVEY.md section 8a: `lara_amd.cameras` against the REFERENCE's own `MiniCam` /
`getProjectionMatrix` (lightning/utils.py:5-48).

(1) fixture test: tests/golden/cameras_ref.npz is the output of the reference's classes
    (tests/golden/make_cameras_fixture.py); `make_cameras` must reproduce the four tensors the renderer hands to the
    rasteriser.  Tolerance: the reference inverts c2w with fp32 LAPACK, the mirror in fp64 then rounds ->
    |diff| <= 2e-6 (1 + |value|).
(2) contract test (runs only where /root/reference exists, i.e. in the build container): the reference's UNMODIFIED
    `lightning.renderer_2dgs` is imported against the shim package `diff_surfel_rasterization`, its
    `Renderer.set_rasterizer` is fed a reference `MiniCam`, and the settings record it builds is compared field by
    field -- names, order, Python types, tensor shapes / dtypes, values -- with the one `lara_amd.renderer.Renderer`
    builds from a `lara_amd.cameras.Camera`.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def fixture():
    return np.load(os.path.join(HERE, "golden", "cameras_ref.npz"))


def test_cameras_match_reference_minicam():
    from lara_amd import cameras
    f = fixture()
    c2w = torch.from_numpy(f["c2w"])
    for ci in (0, 1):
        W, H, fovx, fovy, zn, zf = f[f"case{ci}/params"]
        cams = cameras.make_cameras(c2w, int(W), int(H), float(fovx), float(fovy), float(zn), float(zf))
        assert len(cams) == c2w.shape[0]
        P = cameras.projection_matrix(float(zn), float(zf), float(fovx), float(fovy)).numpy()
        np.testing.assert_allclose(P, f[f"case{ci}/P"], rtol=1e-6, atol=1e-7)
        for i, cam in enumerate(cams):
            for name in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
                got, ref = getattr(cam, name).numpy(), f[f"case{ci}/{name}"][i]
                assert got.dtype == np.float32 and got.shape == ref.shape, name
                np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-6, err_msg=f"case {ci} view {i} {name}")
            assert (cam.image_width, cam.image_height) == (int(W), int(H))
            assert math.isclose(cam.FoVx, fovx) and math.isclose(cam.FoVy, fovy)
        # the reference's (sic) camera centre is the NEGATED translation of c2w (utils.py:48)
        np.testing.assert_array_equal(cams[0].camera_center.numpy(), -f["c2w"][0, :3, 3])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lightning")), reason="reference tree not present (GPU box)")
def test_reference_renderer_builds_the_same_settings_through_the_shim():
    root = os.path.dirname(HERE)
    saved_path, saved_mods = list(sys.path), {k: sys.modules[k] for k in list(sys.modules) if k == "lightning" or k.startswith("lightning.")}
    try:
        for k in saved_mods:
            del sys.modules[k]
        sys.path.insert(0, root)     # the shim `diff_surfel_rasterization`
        sys.path.insert(0, REF)
        import diff_surfel_rasterization as shim
        import lightning.renderer_2dgs as ref_r          # unmodified; `from diff_surfel_rasterization import ...` at :7-10
        from lightning.utils import MiniCam
        from lara_amd import cameras, rasterizer
        from lara_amd.renderer import Renderer as OurRenderer
        assert ref_r.GaussianRasterizationSettings is rasterizer.GaussianRasterizationSettings is shim.GaussianRasterizationSettings
        assert ref_r.GaussianRasterizer is rasterizer.GaussianRasterizer
        f = fixture()
        c2w = torch.from_numpy(f["c2w"][3])
        W, H, fovx, fovy, zn, zf = 512, 384, 0.75, 0.6, 1.106, 2.706
        ref_cam = MiniCam(c2w.clone(), W, H, torch.tensor(fovy), torch.tensor(fovx), torch.tensor(zn), torch.tensor(zf), "cpu")
        our_cam = cameras.make_cameras(c2w[None], W, H, fovx, fovy, zn, zf)[0]
        bg = torch.tensor([0.5, 0.5, 0.5])
        settings = []
        for R, cam in ((ref_r.Renderer, ref_cam), (OurRenderer, our_cam)):
            r = R(sh_degree=1, white_background=True)
            r.set_bg_color(bg)                                      # network.py:489-490
            rast = r.set_rasterizer(cam, device="cpu")              # renderer_2dgs.py:119-139
            assert isinstance(rast, rasterizer.GaussianRasterizer)
            settings.append(rast.raster_settings)
        a, b = settings
        assert type(a) is type(b) and a._fields == b._fields == (
            "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
            "sh_degree", "campos", "prefiltered", "debug")
        for name in a._fields:
            x, y = getattr(a, name), getattr(b, name)
            if isinstance(x, torch.Tensor):
                assert isinstance(y, torch.Tensor) and x.shape == y.shape and x.dtype == y.dtype == torch.float32, name
                np.testing.assert_allclose(y.numpy(), x.numpy(), rtol=2e-6, atol=2e-6, err_msg=name)
            else:
                assert type(x) is type(y), (name, type(x), type(y))
                assert x == y or math.isclose(x, y, rel_tol=1e-6), name
        # the operator itself has no CPU path: the reference's renderer reaches it and it must fail loudly there
        with pytest.raises(RuntimeError, match="no CPU path|HIP library not found"):
            p = 4
            ref_r.Renderer(sh_degree=1).render_img(ref_cam, None, torch.zeros(p, 3), torch.zeros(p, 4, 3), torch.zeros(p, 1),
                                                   torch.zeros(p, 2), torch.ones(p, 4), "cpu")
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == "lightning" or k.startswith("lightning.")]:
            del sys.modules[k]
        sys.modules.update(saved_mods)


def test_cameras_of_a_whole_batch_equal_the_per_scene_cameras():
    """`make_cameras_scenes` (one batched pass for the B x V cameras of a batch) against `make_cameras` per scene."""
    import torch
    from lara_amd import cameras
    c2w = torch.stack([cameras.turntable_c2w(8), cameras.turntable_c2w(8, 20.0)])
    scalars = [(1.1, 2.7, 0.75, 0.7), (0.9, 2.5, 0.6, 0.65)]
    sizes = [(512, 512), (256, 128)]
    got = cameras.make_cameras_scenes(c2w, sizes, scalars)
    for b in range(2):
        want = cameras.make_cameras(c2w[b], sizes[b][0], sizes[b][1], scalars[b][2], scalars[b][3], scalars[b][0], scalars[b][1])
        assert len(got[b]) == len(want) == 8
        for x, y in zip(got[b], want):
            for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center"):
                assert torch.equal(getattr(x, k), getattr(y, k)), k
            assert (x.image_width, x.image_height, x.FoVx, x.FoVy, x.znear, x.zfar) == \
                   (y.image_width, y.image_height, y.FoVx, y.FoVy, y.znear, y.zfar)
