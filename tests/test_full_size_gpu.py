"""The configuration the headline times (BASELINE.json configs[2]: P = 524 288 surfels, 8 views, 512 x 512, 4 scenes on two
scene streams), held to the same bars as the small cases:

* the multi-view launch (`blockIdx.z` = view, 8 x 282 MB of state at a stride -- the 64-bit strides, the capacity carving
  of `ckpt` / `pair_mask` and the one-launch `preprocess_bwd_views` fold only meet their real sizes here) against the
  per-view operator: outputs and summed gradients bit for bit, one view against the CPU oracle;
* `LaRaPipeline` (lightning/network.py:473-527) with two scene streams against one stream over 50 training steps whose
  fine subsets change size every step;
* the two other BASELINE configurations at their real size (round 5): the "trained-like" regime of SURVEY section 8d (the
  one the second bench line is quoted on) through the multi-view launch, and configs[4]'s 1024 x 1024 eval views over all
  524 288 surfels (D ~ 6 M pairs) through the per-view operator -- forward surface and gradients against the oracle.
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import oracle_view, raster_settings, run_oracle, to_numpy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_multi_view_launch_at_benchmark_size_equals_the_per_view_operator_and_the_oracle(hip_lib):
    from lara_amd import GaussianRasterizer, cameras, rasterize_gaussians_views, rasterizer, synthetic
    from tests.test_raster_parity_gpu import _check_forward
    P, n, S = 524288, 8, 512
    act = synthetic.activate(synthetic.make_scene(grid=64, K=2, seed=0))
    assert act["means3D"].shape[0] == P
    cams = cameras.make_cameras(cameras.turntable_c2w(n), S, S, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8)
    bgs = [(1.0, 1.0, 1.0)] * 4 + [(0.0, 0.0, 0.0), (0.5, 0.5, 0.5), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)]   # gobjverse.py:103-106
    settings = [raster_settings(c, bg, device=DEV) for c, bg in zip(cams, bgs)]
    g = torch.Generator().manual_seed(5)
    dc = torch.randn(n, 3, S, S, generator=g).to(DEV)
    da = (torch.randn(n, 7, S, S, generator=g) * 0.1).to(DEV)
    leaves = lambda: {k: v.to(DEV).clone().requires_grad_(True) for k, v in act.items()}

    inp = leaves()
    color, radii, allmap = rasterize_gaussians_views(settings, inp["means3D"], None, inp["opacities"], shs=inp["shs"],
                                                     scales=inp["scales"], rotations=inp["rotations"])
    node = color.grad_fn        # the forward's state buffer and capacity
    state, (sb, _), cap = node.state, node.strides, node.cap
    ((color * dc).sum() + (allmap * da).sum()).backward()
    torch.cuda.synchronize()

    # the same scene through the FORWARD-ONLY instantiation of every kernel (what evaluation.py / the mesh extractor get under
    # no_grad): the multi-view call and one per-view call, bit-identical to the training-mode forward at full size
    with torch.no_grad():
        c_fo, r_fo, a_fo = rasterize_gaussians_views(settings, inp["means3D"], None, inp["opacities"], shs=inp["shs"],
                                                     scales=inp["scales"], rotations=inp["rotations"])
        c1, r1, a1 = GaussianRasterizer(settings[6])(means3D=inp["means3D"], means2D=None, shs=inp["shs"], opacities=inp["opacities"],
                                                     scales=inp["scales"], rotations=inp["rotations"])
    assert c_fo.grad_fn is None and torch.equal(c_fo, color.detach()) and torch.equal(a_fo, allmap.detach()) and torch.equal(r_fo, radii)
    assert torch.equal(c1, color[6].detach()) and torch.equal(a1, allmap[6].detach()) and torch.equal(r1, radii[6])
    del c_fo, r_fo, a_fo, c1, r1, a1

    want = None
    for i, rs in enumerate(settings):
        li = leaves()
        c, r, a = GaussianRasterizer(rs)(means3D=li["means3D"], means2D=None, shs=li["shs"], opacities=li["opacities"],
                                         scales=li["scales"], rotations=li["rotations"])
        ((c * dc[i]).sum() + (a * da[i]).sum()).backward()
        assert torch.equal(color[i].detach(), c.detach()), f"view {i}: colour"
        assert torch.equal(allmap[i].detach(), a.detach()), f"view {i}: allmap"
        assert torch.equal(radii[i], r), f"view {i}: radii"
        gi = {k: v.grad for k, v in li.items()}
        want = gi if want is None else {k: want[k] + gi[k] for k in gi}            # view order, as the library folds them
    for k in want:
        assert torch.equal(inp[k].grad, want[k]), f"grad {k}: max diff {(inp[k].grad - want[k]).abs().max().item():.3e}"

    # view 5 (grey background, a novel view) of the multi-view launch against the oracle, the integer surface included
    v = 5
    views = rasterizer.state_views(state.view(n, sb)[v], P, S, S, cap)
    r = {"views": views, "radii": radii[v], "color": color[v].detach(), "allmap": allmap[v].detach()}
    D = _check_forward(r, run_oracle(oracle_view(cams[v], bgs[v]), to_numpy(act)), S, S)
    assert 1_000_000 < D < 3_000_000


def test_trained_like_regime_at_benchmark_size_multi_view_launch_against_the_oracle(hip_lib):
    """SURVEY section 8d's second regime (opaque thin shell, everything else transparent: pixels saturate after tens of
    surfels, most backward work items end on `tile_maxc`, the work items differ widely in cost) at P = 524 288 / 8 views /
    512^2, through the multi-view launch: one view's integer surface, images and contributor records against the oracle, and
    its gradients -- the upstream gradients of the other seven views are zero, so the call's summed gradient IS that view's --
    under the full-size gradient bar."""
    from lara_amd import cameras, rasterize_gaussians_views, rasterizer, synthetic
    from tests.test_raster_parity_gpu import _check_forward, oracle_conditioned_gradient_check
    P, n, S, v = 524288, 8, 512, 6
    act = synthetic.activate(synthetic.make_scene(grid=64, K=2, seed=0, regime="trained"))
    cams = cameras.make_cameras(cameras.turntable_c2w(n), S, S, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8)
    bgs = [(1.0, 1.0, 1.0)] * 4 + [(0.0, 0.0, 0.0), (0.5, 0.5, 0.5), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)]   # gobjverse.py:103-106
    settings = [raster_settings(c, bg, device=DEV) for c, bg in zip(cams, bgs)]
    g = np.random.default_rng(3)
    dc = g.normal(size=(3, S, S)).astype(np.float32)
    da = (0.1 * g.normal(size=(7, S, S))).astype(np.float32)
    gc = torch.zeros(n, 3, S, S, device=DEV); gc[v] = torch.from_numpy(dc).to(DEV)
    ga = torch.zeros(n, 7, S, S, device=DEV); ga[v] = torch.from_numpy(da).to(DEV)
    inp = {k: t.to(DEV).clone().requires_grad_(True) for k, t in act.items()}
    color, radii, allmap = rasterize_gaussians_views(settings, inp["means3D"], None, inp["opacities"], shs=inp["shs"],
                                                     scales=inp["scales"], rotations=inp["rotations"])
    state, (sb, _), cap = color.grad_fn.state, color.grad_fn.strides, color.grad_fn.cap
    ((color * gc).sum() + (allmap * ga).sum()).backward()
    torch.cuda.synchronize()
    views = rasterizer.state_views(state.view(n, sb)[v], P, S, S, cap)
    r = {"views": views, "radii": radii[v], "color": color[v].detach(), "allmap": allmap[v].detach()}
    D = _check_forward(r, run_oracle(oracle_view(cams[v], bgs[v]), to_numpy(act)), S, S)
    assert 1_000_000 < D < 3_000_000
    # the regime is what it claims: the walk ends early almost everywhere inside the silhouette
    nc = views["n_contrib"][0].cpu().numpy()
    lens = (views["ranges"][:, 1] - views["ranges"][:, 0]).cpu().numpy().reshape(S // 16, S // 16)
    inside = allmap[v, 1].detach().cpu().numpy() > 0.99
    lens_px = np.repeat(np.repeat(lens, 16, 0), 16, 1)
    assert inside.mean() > 0.1 and (nc[inside] < lens_px[inside]).mean() > 0.95 and np.median(nc[inside]) < 0.75 * np.median(lens_px[inside])
    rep = oracle_conditioned_gradient_check(to_numpy(act), cams[v], bgs[v], dc, da, {k: t.grad.cpu().numpy() for k, t in inp.items()})
    print("trained-like, full size, view %d: D = %d; (rel-L2 HIP, rel-L2 oracle +1ulp, surfels > 1e-3 HIP, oracle): %s" % (
        v, D, {k: (round(a, 6), round(b, 6), c, d) for k, (a, b, c, d) in rep.items()}))


def test_eval_resolution_at_full_surfel_count_against_the_oracle(hip_lib):
    """BASELINE.json configs[4] at its real size: a 1024 x 1024 novel view (4096 tiles) of all 524 288 surfels, as
    `eval_all.py` / the mesh extractor render them -- D = 3.1 M (tile, surfel) pairs, twice the training frame's, ~4 k full
    backward segments.  The per-view operator (what `render_img` calls), forward surface and gradients against the oracle.
    The frame sits at the very top of the capacity a new size class starts from (4 pairs per quantised surfel -> 3 Mi pairs;
    rounds 1-4 ran such a frame -- more than half the capacity -- with unsegmented backward tiles): the first call of the class
    reads D before it returns, and repeats itself if it has to (rasterizer.py, workspace policy)."""
    from lara_amd import GaussianRasterizer, cameras, rasterizer, synthetic
    from tests.test_raster_parity_gpu import _check_forward, oracle_conditioned_gradient_check
    rasterizer.reset_capacity_history()
    P, S = 524288, 1024
    act = synthetic.activate(synthetic.make_scene(grid=64, K=2, seed=0))
    cam = cameras.make_cameras(cameras.turntable_c2w(8)[5:6], S, S, 0.75, 0.75, 0.5, 2.5)[0]     # GSO near / far (google_scanned_objects.py:114)
    bg = (1.0, 1.0, 1.0)
    g = np.random.default_rng(4)
    dc = g.normal(size=(3, S, S)).astype(np.float32)
    da = (0.1 * g.normal(size=(7, S, S))).astype(np.float32)
    inp = {k: t.to(DEV).clone().requires_grad_(True) for k, t in act.items()}
    reruns = rasterizer._reruns
    color, radii, allmap = GaussianRasterizer(raster_settings(cam, bg, device=DEV))(
        means3D=inp["means3D"], means2D=torch.zeros_like(inp["means3D"]), shs=inp["shs"], opacities=inp["opacities"],
        scales=inp["scales"], rotations=inp["rotations"])
    class run:      # the forward's state buffer and capacity (settled before the operator returned)
        state, cap = color.grad_fn.state, color.grad_fn.cap
    ((color * torch.from_numpy(dc).to(DEV)).sum() + (allmap * torch.from_numpy(da).to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    views = rasterizer.state_views(run.state, P, S, S, run.cap)
    r = {"views": views, "radii": radii, "color": color.detach(), "allmap": allmap.detach()}
    D = _check_forward(r, run_oracle(oracle_view(cam, bg), to_numpy(act)), S, S)
    assert D > 2_500_000 and run.cap >= D > run.cap // 2 and int(views["header"][3]) > 2500      # pairs; full 512-entry segments
    assert rasterizer._reruns - reruns == (1 if D > rasterizer.binning_capacity(P) else 0)
    assert np.array_equal(views["seg_cnt"].cpu().numpy(), np.maximum(ranges_len(views) - 1, 0) // 512), "every tile keeps its segments"
    rep = oracle_conditioned_gradient_check(to_numpy(act), cam, bg, dc, da, {k: t.grad.cpu().numpy() for k, t in inp.items()})
    print("1024^2, P = 524 288: D = %d, capacity %d; (rel-L2 HIP, rel-L2 oracle +1ulp, surfels > 1e-3 HIP, oracle): %s" % (
        D, run.cap, {k: (round(a, 6), round(b, 6), c, d) for k, (a, b, c, d) in rep.items()}))
    rasterizer.reset_capacity_history()


def ranges_len(views):
    rg = views["ranges"].cpu().numpy().astype(np.int64)
    return rg[:, 1] - rg[:, 0]


def _full_size_pipeline(dev, layers=2):
    from lara_amd.batch import synthetic_batch
    from lara_amd.encoder_train import VolTransformer
    from lara_amd.pipeline import CoarseFineDecoder, LaRaPipeline
    torch.manual_seed(0)
    enc = VolTransformer(256, 800, [16], 32, 64, 80, layers, 16)
    pipe = LaRaPipeline(enc, CoarseFineDecoder(), grid_reso=32, n_streams=2).to(dev)
    pipe.fine_mask = "reference"
    pipe.train()
    batch = synthetic_batch(batch_size=4, n_views=8, H=512, W=512, n_input=4, seed=7, device=dev)
    g = torch.Generator(device="cpu").manual_seed(11)
    batch["tar_rgb"] = torch.rand(batch["tar_rgb"].shape, generator=g).to(dev)
    feat_vol = torch.randn(4, 4, 800, 16, 16, 16, generator=g).to(dev).requires_grad_(True)
    return pipe, batch, feat_vol


def test_pipeline_at_benchmark_size_two_scene_streams_equal_one_over_50_steps(hip_lib):
    """4 scenes x (8 coarse + 8 fine) views at P = 524 288 / 512^2, `_check_mask` thinning the fine subsets at random so
    that their sizes (and with them every buffer size of the fine pass) change with every step; step k with two scene
    streams must give the outputs of step k on one stream bit for bit and the same gradients up to the order of the sampler's
    float atomics (bars below)."""
    from lara_amd import rasterizer
    from lara_amd.loss import lara_loss
    dev = torch.device(DEV)
    pipe, batch, feat_vol = _full_size_pipeline(dev)
    params = [p for p in pipe.parameters() if p.requires_grad]
    sizes_seen = set()
    poison = os.environ.get("LARA2DGS_POISON_BUFFERS") == "1"
    orig = pipe.gs_render.render_views

    def spy(cams, rays, centers, *a, **k):
        sizes_seen.add(int(centers.shape[0]))
        return orig(cams, rays, centers, *a, **k)
    pipe.gs_render.render_views = spy

    def run(step, n_streams):
        pipe.n_streams = n_streams
        torch.manual_seed(1000 + step)               # the same random thinning in both runs
        for p in params:
            p.grad = None
        feat_vol.grad = None
        out = pipe(batch, feat_vol, with_fine=True)
        loss, _ = lara_loss(batch, out, 2000, ms_ssim=False)
        loss.backward()
        pipe.join_streams()
        keep = {k: out[k].detach() for k in ("image", "image_fine", "acc_map", "rend_dist", "depth_fine")}
        grads = [p.grad for p in params] + [feat_vol.grad]
        if poison:      # (debug mode: release the guarded buffers of this step -- 30 GB of them -- and check their guard zones)
            assert rasterizer.check_poison_guards() == []
        return keep, float(loss.detach()), grads

    # gradient bars as in tests/test_pipeline.py (`close`): the only run-to-run noise is the order of the sampler's float atomics
    # (~1e-7 relative on the coarse maps' gradients); the fine decoder's parameters see it directly (fp32 path: 2e-4 of max),
    # the coarse MLP and the encoder round their backward operands to bf16 first, where a last-bit flip is 4e-3 relative: 1e-2
    names = [n for n, p in pipe.named_parameters() if p.requires_grad] + ["feat_vol"]
    fp32_path = lambda n: n.startswith(("decoder.norm", "decoder.cross_att", "decoder.mlp_fine"))
    worst = {"fp32": 0.0, "bf16": 0.0, "cos": 1.0}
    for step in range(50):
        o2, l2, g2 = run(step, 2)
        o1, l1, g1 = run(step, 1)
        for k in o1:
            assert torch.equal(o1[k], o2[k]), f"step {step}: output {k} differs between one and two scene streams"
        assert l1 == l2, (step, l1, l2)
        for n, a, b in zip(names, g1, g2):
            assert (a is None) == (b is None), n
            if a is not None:
                assert torch.isfinite(b).all(), (step, n)
                d = float((a - b).abs().max()) / (float(a.abs().max()) + 1e-30)
                cos = float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-300))
                key = "fp32" if fp32_path(n) else "bf16"
                worst[key] = max(worst[key], d)
                worst["cos"] = min(worst["cos"], cos)
                assert d <= (2e-4 if fp32_path(n) else 1e-2) and cos >= 1 - 1e-5, (step, n, d, cos)
    torch.cuda.synchronize()
    # the fine subsets really changed size: P for the coarse pass + a different count per (step, scene)
    assert 524288 in sizes_seen and len(sizes_seen) > 50, len(sizes_seen)
    print(f"two-stream vs one-stream over 50 full-size steps: outputs bit-identical; worst gradient difference (of max) "
          f"{worst['fp32']:.2e} on the fp32 path, {worst['bf16']:.2e} behind bf16 products, cosine >= {worst['cos']:.7f}; "
          f"{len(sizes_seen) - 1} distinct fine-subset sizes")
