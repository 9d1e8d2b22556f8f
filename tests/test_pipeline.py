"""`lara_amd.pipeline`: the glue between the operators of the LaRa step against the REFERENCE's own glue
(tests/golden/pipeline_ref.npz, produced by tests/golden/make_pipeline_fixture.py from lightning/network.py and
lightning/loss.py run on CPU): coarse decoding + centres + masks, `_check_mask`, the loss.  The GPU test holds the whole
pipeline (streams, multi-view calls, row gathers) to the same operators called one by one, single stream, per view, with
boolean-mask indexing -- the way `Network.forward` (network.py:473-527) issues them."""
import os

import numpy as np
import pytest
import torch

FX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pipeline_ref.npz")


def _decoder(fx=None):
    from lara_amd.pipeline import CoarseFineDecoder
    torch.manual_seed(4)
    dec = CoarseFineDecoder()
    if fx is not None:
        with torch.no_grad():
            for k, p in dec.mlp_coarse.named_parameters():
                p.copy_(torch.from_numpy(fx["dec.mlp_coarse." + k]))
    return dec


def test_decoder_container_has_the_reference_state_dict_keys():
    fx = np.load(FX)
    keys = {k[len("dec."):] for k in fx.files if k.startswith("dec.")}
    sd = _decoder().state_dict()
    assert keys <= set(sd) and all(tuple(sd[k].shape) == fx["dec." + k].shape for k in keys)
    assert {"norm.weight", "cross_att.q_proj_weight", "cross_att.out_proj.weight", "mlp_fine.2.bias"} <= set(sd)


def test_decode_coarse_centres_and_masks_match_the_reference():
    from lara_amd.pipeline import LaRaPipeline, decode_coarse
    fx = np.load(FX)
    dec = _decoder(fx)
    vol = torch.from_numpy(fx["vol"])
    offset, sh, scaling, rotation, opacity = decode_coarse(dec, vol, float(fx["opacity_shift"]), float(fx["scaling_shift"]),
                                                           autocast=False)
    pipe = LaRaPipeline(torch.nn.Identity(), dec, grid_reso=4)
    assert pipe.scaling_shift == pytest.approx(float(fx["scaling_shift"]), rel=1e-12) and pipe.opacity_shift == float(fx["opacity_shift"])
    g = pipe.gaussians_from_volume(vol, autocast=False)
    for k, want in (("centers", "centers"), ("shs", "sh"), ("scaling", "scaling"), ("rotation", "rotation"), ("opacity", "opacity")):
        np.testing.assert_allclose(g[k].detach().numpy(), fx[want], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_array_equal(g["masks"].numpy(), fx["masks"])
    np.testing.assert_allclose(sh.detach().numpy(), fx["sh"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["sparse", "dense", "middle"])
@pytest.mark.parametrize("training", [True, False])
def test_check_mask_matches_the_reference(name, training):
    """Same seed -> same first `torch.rand` draw as the branch the reference takes; the branch-free selection must
    then return the reference's mask bit for bit."""
    from lara_amd.pipeline import check_mask
    fx = np.load(FX)
    torch.manual_seed(100)
    got = check_mask(torch.from_numpy(fx[f"mask.{name}.in"]), training)
    np.testing.assert_array_equal(got.numpy(), fx[f"mask.{name}.{'train' if training else 'eval'}"])


@pytest.mark.parametrize("it", [500, 2000])
@pytest.mark.parametrize("with_fine", [True, False])
def test_loss_matches_the_reference_minus_ms_ssim(it, with_fine):
    from lara_amd.pipeline import lara_loss
    fx = np.load(FX)
    batch = {"tar_rgb": torch.from_numpy(fx["loss.in.tar_rgb"])}
    o = {k[len("loss.in."):]: torch.from_numpy(fx[k]).requires_grad_(True) for k in fx.files
         if k.startswith("loss.in.") and k != "loss.in.tar_rgb" and (with_fine or not k.endswith("_fine"))}
    loss, stats = lara_loss(batch, o, it, ms_ssim=False)      # (the fixture ran the reference with MS_SSIM stubbed to 1)
    loss.backward()
    tag = f"loss.{it}.{'fine' if with_fine else 'coarse'}"
    assert float(loss) == pytest.approx(float(fx[tag]), rel=1e-6)
    for k, v in o.items():
        got = v.grad.numpy() if v.grad is not None else np.zeros(v.shape, np.float32)
        np.testing.assert_allclose(got, fx[f"{tag}.d_{k}"], rtol=1e-5, atol=1e-9, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("it", [500, 2000])
@pytest.mark.parametrize("with_fine", [True, False])
def test_fused_loss_matches_the_reference_minus_ms_ssim(hip_lib, it, with_fine):
    """`lara_amd.loss.lara_loss` (one HIP kernel per direction) against the same fixture of the reference's `Losses.forward`."""
    from lara_amd.loss import lara_loss
    fx = np.load(FX)
    batch = {"tar_rgb": torch.from_numpy(fx["loss.in.tar_rgb"]).cuda()}
    o = {k[len("loss.in."):]: torch.from_numpy(fx[k]).cuda().requires_grad_(True) for k in fx.files
         if k.startswith("loss.in.") and k != "loss.in.tar_rgb" and (with_fine or not k.endswith("_fine"))}
    loss, stats = lara_loss(batch, o, it, ms_ssim=False)
    loss.backward()
    tag = f"loss.{it}.{'fine' if with_fine else 'coarse'}"
    assert float(loss) == pytest.approx(float(fx[tag]), rel=2e-6)
    assert set(stats) == {"mse", "psnr"} | ({"mse_fine", "psnr_fine"} if with_fine else set()) | ({"distortion", "normal"} if it > 1000 else set())
    for k, v in o.items():
        got = v.grad.cpu().numpy() if v.grad is not None else np.zeros(v.shape, np.float32)
        np.testing.assert_allclose(got, fx[f"{tag}.d_{k}"], rtol=1e-5, atol=1e-9, err_msg=k)


def test_voxel_row_gather_is_the_expand_and_mask_of_the_reference():
    from lara_amd.pipeline import _TakeVoxelRows
    g = torch.Generator().manual_seed(2)
    K, n = 2, 50
    x = torch.randn(n, 5, generator=g, requires_grad=True)
    mask = torch.rand(n * K, generator=g) < 0.6
    want = x.unsqueeze(1).expand(-1, K, -1)[mask.view(-1, K)]           # network.py:509
    idx = mask.nonzero().squeeze(-1)
    got = _TakeVoxelRows.apply(x, torch.div(idx, K, rounding_mode="floor"))
    assert torch.equal(got, want)
    w = torch.randn(want.shape, generator=g)
    gw, = torch.autograd.grad((want * w).sum(), x)
    gg, = torch.autograd.grad((got * w).sum(), x)
    assert torch.equal(gw, gg)


# ------------------------------------------------------------------------------------------------------------------
def _small_problem(dev, seed=0, B=2, V=6, res=64):
    from lara_amd.batch import synthetic_batch
    from lara_amd.encoder_train import VolTransformer
    from lara_amd.pipeline import CoarseFineDecoder, LaRaPipeline
    torch.manual_seed(seed)
    enc = VolTransformer(embed_dim=256, image_feat_dim=800, n_groups=[2], vol_low_res=4, vol_high_res=8, out_dim=80,
                         num_layers=2, num_heads=16).to(dev)
    dec = CoarseFineDecoder().to(dev)
    with torch.no_grad():       # spread the opacities around the 0.005 threshold so that the mask really selects
        dec.mlp_coarse[4].weight.mul_(6.0)
        for p in (dec.norm.weight, dec.norm.bias, dec.mlp_fine[0].bias, dec.mlp_fine[2].bias):
            p.add_(torch.randn_like(p) * 0.2)
    pipe = LaRaPipeline(enc, dec, grid_reso=4, n_offset_groups=4, n_views=4).to(dev)
    pipe.opacity_shift = -5.0
    batch = synthetic_batch(batch_size=B, n_views=V, H=res, W=res, n_input=4, seed=seed, device=dev)
    g = torch.Generator().manual_seed(seed + 1)
    batch["tar_rgb"] = torch.rand(batch["tar_rgb"].shape, generator=g).to(dev)
    feat_vol = torch.randn(B, 4, 800, 2, 2, 2, generator=g).to(dev).requires_grad_(True)
    return pipe, batch, feat_vol


def _one_by_one(pipe, batch, feat_vol, with_fine):
    """The operators called the way the reference's loop calls them: one stream, one `render_img` per view, `x[mask]`."""
    from lara_amd.fine import forward_fine, sample_point_feats
    from lara_amd.renderer import Renderer
    r = Renderer(sh_degree=1, white_background=True)
    g = pipe.gaussians(feat_vol)
    B, n_sel, dev = feat_vol.shape[0], pipe.n_views, feat_vol.device
    inps = batch["tar_rgb"][:, :n_sel].permute(0, 1, 4, 2, 3).float().contiguous()
    outs = []
    for i in range(B):
        cams = pipe.scene_cameras(batch, i)
        views = []
        for j, cam in enumerate(cams):
            r.set_bg_color(batch["bg_color"][i, j])
            views.append(r.render_img(cam, batch["tar_rays"][i, j], g["centers"][i], g["shs"][i], g["opacity"][i], g["scaling"][i],
                                      g["rotation"][i], dev))
        if with_fine:
            mask = g["masks"][i]
            ren = {k: torch.stack([v[k] for v in views[:n_sel]]) for k in ("image", "acc_map", "depth")}
            centers_f = g["centers"][i][mask]
            pf = sample_point_feats(centers_f, batch["tar_w2c"][i, :n_sel], batch["tar_ixt"][i, :n_sel], inps[i], ren["image"],
                                    ren["acc_map"], ren["depth"])
            vpf = g["vol"][i].unsqueeze(1).expand(-1, pipe.K, -1)[mask.view(-1, pipe.K)]
            shs_f = forward_fine(pipe.decoder, vpf, torch.einsum("lcb->blc", pf)).view(-1, 4, 3) + g["shs"][i][mask]
            for j, cam in enumerate(cams):
                r.set_bg_color(batch["bg_color"][i, j])
                views[j].update(r.render_img(cam, batch["tar_rays"][i, j], centers_f, shs_f, g["opacity"][i][mask],
                                             g["scaling"][i][mask], g["rotation"][i][mask], dev, prex="_fine"))
        outs.append({k: torch.cat([v[k] for v in views], dim=1) for k in views[0]})
    return {k: torch.stack([o[k] for o in outs]) for k in outs[0]}


@pytest.mark.gpu
@pytest.mark.parametrize("with_fine,n_streams", [(True, 2), (True, 1), (False, 2)])
def test_pipeline_equals_the_operators_called_one_by_one(hip_lib, with_fine, n_streams):
    from lara_amd.pipeline import lara_loss
    dev = torch.device("cuda:0")
    pipe, batch, feat_vol = _small_problem(dev)
    pipe.n_streams, pipe.fine_mask = n_streams, "plain"
    params = [p for p in pipe.parameters() if p.requires_grad]

    def run(fn):
        for p in params:
            p.grad = None
        feat_vol.grad = None
        out = fn()
        loss, _ = lara_loss(batch, out, 2000, ms_ssim=False)      # (64-pixel images: below MS-SSIM's 161-pixel minimum)
        # (+ a term on the depth maps, which the reference's loss does not read, so that every returned map carries a gradient)
        loss = loss + sum(out[k].mean() * 0.01 for k in out if k.startswith("depth") and not k.startswith("depth_normal"))
        loss.backward()
        pipe.join_streams()
        torch.cuda.synchronize()
        return ({k: v.detach().clone() for k, v in out.items()}, float(loss),
                {n: (p.grad.clone() if p.grad is not None else None) for n, p in pipe.named_parameters()}, feat_vol.grad.clone())

    out_a, loss_a, grads_a, gfeat_a = run(lambda: pipe(batch, feat_vol, with_fine=with_fine))
    out_b, loss_b, grads_b, gfeat_b = run(lambda: _one_by_one(pipe, batch, feat_vol, with_fine))
    assert set(out_a) == set(out_b) and (("image_fine" in out_a) == with_fine)
    kept = float(torch.sigmoid(pipe.gaussians(feat_vol)["opacity"]).gt(0.005).float().mean())
    assert 0.05 < kept < 0.95, f"the mask keeps {kept:.2f} of the Gaussians: not a test of the subset path"
    for k in out_a:     # the multi-view call returns each view's maps bit for bit (tests/test_views_gpu.py)
        assert out_a[k].shape == out_b[k].shape == (2, 64, 6 * 64) + out_a[k].shape[3:]
        assert torch.equal(out_a[k], out_b[k]), k
    assert loss_a == loss_b
    # gradients: the same terms, summed over the views inside the library instead of by autograd (other order: ~1e-6
    # relative on the Gaussians' gradients).  The fine decoder's parameters see that directly (fp32 path: 2e-4 of max);
    # the coarse MLP (bf16 autocast) and the encoder (bf16 operands in every backward product) round those inputs to
    # bf16 first, where a last-bit difference is 4e-3 relative and a few such roundings chain: 1e-2 of max, direction to 1 - 1e-5
    def close(a, b, n):
        fp32_path = n.startswith(("decoder.norm", "decoder.cross_att", "decoder.mlp_fine"))
        assert float((a - b).abs().max()) <= (2e-4 if fp32_path else 1e-2) * float(b.abs().max()) + 1e-12, n
        cos = float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-300))
        assert cos >= 1 - 1e-5, (n, cos)
    for n in grads_a:
        a, b = grads_a[n], grads_b[n]
        assert (a is None) == (b is None), n
        if a is not None:
            close(a, b, n)
    close(gfeat_a, gfeat_b, "feat_vol")
    unused = [n for n, g in grads_a.items() if g is None]
    assert (not unused) if with_fine else all(n.startswith(("decoder.norm", "decoder.cross_att", "decoder.mlp_fine")) for n in unused)


@pytest.mark.gpu
@pytest.mark.parametrize("it,with_fine", [(2000, True), (500, True), (500, False), (2000, False)])
def test_passes_whose_maps_take_no_gradient_run_the_colour_only_backward(hip_lib, it, with_fine):
    """lightning/loss.py:35-60: the fine pass contributes its image only; the coarse pass its distortion and normal maps from
    iteration 1000 on, and -- whenever there is a fine pass -- its image, depth and alpha maps through the fine decoder's
    sampler (network.py:499-504).  A pass without a gradient on its maps hands `None` down to the rasteriser (no seven planes of
    zeros written and read) and the library runs composite_bwd's colour-only form for it."""
    from lara_amd import rasterizer
    from lara_amd.pipeline import lara_loss
    dev = torch.device("cuda:0")
    pipe, batch, feat_vol = _small_problem(dev)
    out = pipe(batch, feat_vol, with_fine=with_fine)
    loss, _ = lara_loss(batch, out, it, ms_ssim=False)
    pipe.join_streams()
    torch.cuda.synchronize()
    rasterizer.profile_enable(True)
    try:
        rasterizer.profile_collect()
        loss.backward()
        pipe.join_streams()
        torch.cuda.synchronize()
        names = [n for n, _ in rasterizer.profile_collect()]
    finally:
        rasterizer.profile_enable(False)
    scenes = batch["tar_rgb"].shape[0]
    coarse_full = with_fine or it > 1000
    assert names.count("composite_bwd") == (scenes if coarse_full else 0), names
    assert names.count("composite_bwd_color") == (scenes if with_fine else 0) + (0 if coarse_full else scenes), names
    used = [p for n, p in pipe.named_parameters() if p.requires_grad and (with_fine or not n.startswith(("decoder.norm", "decoder.cross_att", "decoder.mlp_fine")))]
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in used)


@pytest.mark.gpu
def test_pipeline_reference_mask_thins_dense_masks_in_training(hip_lib):
    """`fine_mask = "reference"` applies `_check_mask` (network.py:381-388): with > 50 % of the Gaussians above the
    opacity threshold a training step renders about half of them in the fine pass, an eval step all of them."""
    dev = torch.device("cuda:0")
    pipe, batch, feat_vol = _small_problem(dev)
    pipe.opacity_shift = 0.0          # every Gaussian passes the 0.005 test
    sizes = {}
    for training in (True, False):
        pipe.train(training)
        seen = []
        orig = pipe.gs_render.render_views

        def spy(cams, rays, centers, *a, **k):
            seen.append(centers.shape[0])
            return orig(cams, rays, centers, *a, **k)
        pipe.gs_render.render_views = spy
        with torch.no_grad():
            pipe(batch, feat_vol, with_fine=True)
        pipe.gs_render.render_views = orig
        sizes[training] = seen
    P = 8 ** 3 * 2
    assert sizes[False] == [P, P, P, P]
    assert sizes[True][:2] == [P, P] and all(0.4 * P < n < 0.6 * P for n in sizes[True][2:])


@pytest.mark.gpu
@pytest.mark.parametrize("n_streams", [2, 1])
def test_pipeline_reproduces_the_references_own_network_forward(hip_lib, n_streams):
    """tests/golden/network_ref.npz: the REFERENCE's unmodified `Network.forward` (eval, with_fine) run on CPU with the CPU
    oracle standing in for its absent CUDA rasteriser (tests/golden/make_network_fixture.py).  From the encoder's output on --
    coarse decoder, centres, masks, `MiniCam`, `render_img` with its post-processing, the stacked coarse renders, the
    sampler, `forward_fine`, the fine views over `x[mask]`, the concatenated output dictionary -- the fixture is the
    reference's own composition; here the same volume features and decoder parameters go through `LaRaPipeline` on the GPU
    (fp32 on both sides: the HIP rasteriser against the oracle is a tolerance class)."""
    from lara_amd.pipeline import CoarseFineDecoder, LaRaPipeline
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "network_ref.npz"))
    dev = torch.device("cuda:0")
    dec = CoarseFineDecoder()
    dec.load_state_dict({k[len("decoder."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("decoder.")})
    pipe = LaRaPipeline(torch.nn.Identity(), dec.to(dev), grid_reso=4, n_offset_groups=4, n_views=4, n_streams=n_streams).to(dev).eval()
    pipe.opacity_shift = float(fx["opacity_shift"])
    B, H, W = int(fx["B"]), int(fx["H"]), int(fx["W"])
    batch = {k[len("batch."):]: torch.from_numpy(fx[k]).to(dev) for k in fx.files if k.startswith("batch.")}
    batch["meta"] = {"tar_h": torch.full((B,), H), "tar_w": torch.full((B,), W)}
    with torch.no_grad():
        out = pipe.forward_from_volume(batch, torch.from_numpy(fx["volume_feat_up"]).to(dev), with_fine=True, autocast=False)
        pipe.join_streams()
    torch.cuda.synchronize()
    want = {k[len("out."):]: fx[k] for k in fx.files if k.startswith("out.")}
    assert set(out) == set(want) and 0.5 < float(fx["kept_fraction"]) < 0.95
    for k, w in want.items():
        got = out[k].cpu().numpy()
        assert got.shape == w.shape, k
        err = np.abs(got - w)
        scale = max(1.0, float(np.abs(w).max()))
        # a depth normal is a normalised cross product of finite differences: where the surface depth is ~0 on both sides it
        # is ill-conditioned, so that map gets the looser bar on its worst pixels
        worst, typical = (5e-2, 2e-3) if k.startswith("depth_normal") else (5e-3, 1e-4)
        assert err.max() <= worst * scale, (k, float(err.max()))
        assert np.quantile(err, 0.999) <= typical * scale, (k, float(np.quantile(err, 0.999)))
    mse = float(((out["image_fine"].cpu().numpy() - want["image_fine"]) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-30)) >= 70.0


@pytest.mark.gpu
def test_reference_style_network_forward_matches_the_fused_pipeline(hip_lib):
    """`tools.reference_style.network_forward` -- the reference's own sequence of torch operators around the drop-in
    rasteriser (what bench.py times as `drop_in_step`) -- against the opt-in pipeline on the same parameters: the coarse maps to
    fp32 rounding (same rasteriser kernels, post-processing as torch operators instead of the fused kernel), the fine maps to the
    bf16 rounding of `Decoder.forward_fine` under autocast (the fused fine decoder runs its products in fp32)."""
    from tools import reference_style
    dev = torch.device("cuda:0")
    pipe, batch, feat_vol = _small_problem(dev)
    pipe.fine_mask = "plain"
    with torch.no_grad():
        a = pipe(batch, feat_vol, with_fine=True)
        pipe.join_streams()
        b = reference_style.network_forward(pipe, batch, feat_vol, with_fine=True)
    torch.cuda.synchronize()
    assert set(a) == set(b)
    for k in a:
        assert a[k].shape == b[k].shape, k
        err = float((a[k] - b[k]).abs().max())
        scale = max(1.0, float(b[k].abs().max()))
        if k.endswith("_fine"):
            assert err <= 5e-2 * scale and float((a[k] - b[k]).abs().mean()) <= 2e-3 * scale, (k, err)
        elif k.startswith("depth_normal"):      # normalised cross products of finite differences: ill-conditioned where depth ~ 0
            assert float((a[k] - b[k]).abs().mean()) <= 1e-4, (k, err)
        else:
            assert err <= 2e-5 * scale, (k, err)
    # and it is differentiable end to end (the bench runs its backward)
    out = reference_style.network_forward(pipe, batch, feat_vol, with_fine=True)
    from lara_amd.pipeline import lara_loss
    loss, _ = lara_loss(batch, out, 2000, ms_ssim=False)
    loss.backward()
    assert feat_vol.grad is not None and torch.isfinite(feat_vol.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in pipe.parameters() if p.requires_grad)


def test_pipeline_rejects_a_random_number_of_input_views():
    """`cfg.train.use_rand_views` (network.py:437-441) draws 2-4 input views; the fused fine stage is built for 4 and says so."""
    from lara_amd.pipeline import CoarseFineDecoder, LaRaPipeline
    pipe = LaRaPipeline(torch.nn.Identity(), CoarseFineDecoder(), grid_reso=4, n_offset_groups=4, n_views=4)
    with pytest.raises(NotImplementedError):
        pipe({}, torch.zeros(1, 4, 800, 2, 2, 2), n_views_sel=3)
    with pytest.raises(NotImplementedError):
        pipe({}, torch.zeros(1, 3, 800, 2, 2, 2))
