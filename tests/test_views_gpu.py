"""The multi-view operator (one call = the n views of a scene) against the per-view operator the
reference's loop issues (lightning/network.py:486-497): same kernels, so every per-view output must be the same bits,
and the gradients must be the per-view gradients added in view order (the library's `sum_view_grads` kernel)."""
import numpy as np
import pytest
import torch

from tests.helpers import raster_settings, small_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(act):
    return {k: v.to(DEV).clone().requires_grad_(True) for k, v in act.items()}


@pytest.mark.parametrize("n_views,size", [(4, 128), (8, 96), (1, 64), (5, 80), (11, 64)])
def test_views_match_the_per_view_loop(n_views, size):
    from lara_amd import GaussianRasterizer, rasterize_gaussians_views
    act, cams = small_scene(grid=12, size=size, n_views=n_views, seed=3)
    bgs = [[1.0, 1.0, 1.0], [0.0, 0.0, 0.0], [0.5, 0.5, 0.5]]
    settings = [raster_settings(c, bgs[i % 3], device=DEV) for i, c in enumerate(cams)]
    g = torch.Generator().manual_seed(5)
    dcs = [torch.randn(3, size, size, generator=g).to(DEV) for _ in cams]
    das = [(torch.randn(7, size, size, generator=g) * 0.1).to(DEV) for _ in cams]

    # the reference's loop: one node per view, one backward per view so that the per-view gradients are visible
    per_view, outs = [], []
    for i, rs in enumerate(settings):
        inp = _inputs(act)
        m2 = torch.zeros_like(inp["means3D"], requires_grad=True)
        c, r, a = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=m2, shs=inp["shs"], opacities=inp["opacities"],
                                         scales=inp["scales"], rotations=inp["rotations"])
        ((c * dcs[i]).sum() + (a * das[i]).sum()).backward()
        outs.append((c.detach(), r, a.detach()))
        per_view.append({k: v.grad.clone() for k, v in inp.items()} | {"means2D": m2.grad.clone()})

    inp = _inputs(act)
    m2 = torch.zeros_like(inp["means3D"], requires_grad=True)
    color, radii, allmap = rasterize_gaussians_views(settings, inp["means3D"], m2, inp["opacities"], shs=inp["shs"],
                                                     scales=inp["scales"], rotations=inp["rotations"])
    assert color.shape == (n_views, 3, size, size) and allmap.shape == (n_views, 7, size, size)
    assert radii.shape == (n_views, act["means3D"].shape[0])
    loss = sum((color[i] * dcs[i]).sum() + (allmap[i] * das[i]).sum() for i in range(n_views))
    loss.backward()
    torch.cuda.synchronize()
    for i in range(n_views):
        assert torch.equal(color[i].detach(), outs[i][0]), f"view {i}: colour differs from the per-view call"
        assert torch.equal(radii[i], outs[i][1])
        assert torch.equal(allmap[i].detach(), outs[i][2])
    got = {k: v.grad for k, v in inp.items()} | {"means2D": m2.grad}
    for k in got:
        want = per_view[0][k].clone()
        for pv in per_view[1:]:
            want = want + pv[k]            # view order, as sum_slices_kernel adds
        assert torch.equal(got[k], want), f"grad {k}: max diff {(got[k] - want).abs().max().item():.3e}"


@pytest.mark.parametrize("deg", [0, 2, 3])
def test_views_other_sh_degrees(deg):
    """SH degrees 0, 2, 3 with 16 stored coefficients through the multi-view call (the folded backward preprocess is
    one template instance per degree; coefficients above the active degree must come back with zero gradient)."""
    from lara_amd import GaussianRasterizer, rasterize_gaussians_views
    act, cams = small_scene(grid=10, size=80, n_views=3, seed=30 + deg, sh_coeffs=16)
    settings = [raster_settings(c, [0.0, 0.5, 1.0], sh_degree=deg, device=DEV) for c in cams]
    g = torch.Generator().manual_seed(deg)
    dcs = [torch.randn(3, 80, 80, generator=g).to(DEV) for _ in cams]
    want = None
    for i, rs in enumerate(settings):
        inp = _inputs(act)
        c, r, a = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=None, shs=inp["shs"], opacities=inp["opacities"],
                                         scales=inp["scales"], rotations=inp["rotations"])
        (c * dcs[i]).sum().backward()
        gi = {k: v.grad.clone() for k, v in inp.items()}
        want = gi if want is None else {k: want[k] + gi[k] for k in gi}
    inp = _inputs(act)
    color, radii, allmap = rasterize_gaussians_views(settings, inp["means3D"], None, inp["opacities"], shs=inp["shs"],
                                                     scales=inp["scales"], rotations=inp["rotations"])
    sum((color[i] * dcs[i]).sum() for i in range(len(cams))).backward()
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(inp[k].grad, want[k]), f"deg {deg} grad {k}: max diff {(inp[k].grad - want[k]).abs().max().item():.3e}"
    assert float(inp["shs"].grad[:, (deg + 1) ** 2:].abs().max() if deg < 3 else 0.0) == 0.0
    assert float(inp["shs"].grad[:, :(deg + 1) ** 2].abs().max()) > 0.0


def test_views_with_precomputed_colour_and_transmat():
    """The other input combination of the operator (colors_precomp + cov3D_precomp = the 3x3 splat-to-pixel matrices)
    through the multi-view call:
    outputs and gradients equal the per-view operator's.  The matrices are one camera's (from the oracle's preprocess),
    used for every view: geometrically meaningless for the others, numerically as good a test as any."""
    from lara_amd import GaussianRasterizer, rasterize_gaussians_views
    from tests.helpers import oracle_view, run_oracle, to_numpy
    act, cams = small_scene(grid=10, size=80, n_views=3, seed=9)
    a = to_numpy(act)
    tm = run_oracle(oracle_view(cams[0], (1.0, 1.0, 1.0)), a).transMats
    cols = np.random.default_rng(0).uniform(0, 1, (a["means3D"].shape[0], 3)).astype(np.float32)
    settings = [raster_settings(c, [1.0, 1.0, 1.0], device=DEV) for c in cams]
    g = torch.Generator().manual_seed(6)
    dcs = [torch.randn(3, 80, 80, generator=g).to(DEV) for _ in cams]
    das = [(torch.randn(7, 80, 80, generator=g) * 0.1).to(DEV) for _ in cams]

    def leaves():
        return {k: torch.tensor(v, device=DEV, requires_grad=True) for k, v in
                dict(means3D=a["means3D"], opac=a["opacities"], cols=cols, tm=tm).items()}

    want, outs = None, []
    for i, rs in enumerate(settings):
        t = leaves()
        c, r, al = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opac"],
                                          colors_precomp=t["cols"], cov3D_precomp=t["tm"])
        ((c * dcs[i]).sum() + (al * das[i]).sum()).backward()
        outs.append((c.detach(), al.detach()))
        gi = {k: v.grad.clone() for k, v in t.items()}
        want = gi if want is None else {k: want[k] + gi[k] for k in gi}       # view order
    t = leaves()
    color, radii, allmap = rasterize_gaussians_views(settings, t["means3D"], None, t["opac"], colors_precomp=t["cols"],
                                                     cov3D_precomp=t["tm"])
    sum((color[i] * dcs[i]).sum() + (allmap[i] * das[i]).sum() for i in range(len(cams))).backward()
    torch.cuda.synchronize()
    for i in range(len(cams)):
        assert torch.equal(color[i].detach(), outs[i][0]) and torch.equal(allmap[i].detach(), outs[i][1])
    for k in want:
        assert torch.equal(t[k].grad, want[k]), f"grad {k}: max diff {(t[k].grad - want[k]).abs().max().item():.3e}"


def test_views_repeatable():
    """Same bits run to run (no floating-point atomics anywhere in the rasteriser)."""
    from lara_amd import rasterize_gaussians_views
    act, cams = small_scene(grid=12, size=96, n_views=6, seed=4)
    settings = [raster_settings(c, [1.0, 1.0, 1.0], device=DEV) for c in cams]

    def run():
        inp = _inputs(act)
        m2 = torch.zeros_like(inp["means3D"], requires_grad=True)
        c, r, a = rasterize_gaussians_views(settings, inp["means3D"], m2, inp["opacities"], shs=inp["shs"],
                                            scales=inp["scales"], rotations=inp["rotations"])
        (c.sum() + 0.1 * a.sum()).backward()
        torch.cuda.synchronize()
        return c.detach().clone(), {k: v.grad.clone() for k, v in inp.items()}

    c0, g0 = run()
    c1, g1 = run()
    assert torch.equal(c0, c1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("P", [0, 3, 257])
def test_views_tiny_and_empty_inputs(P):
    """P = 0 (background only, nothing to differentiate), a handful of surfels, one more than a workgroup's worth."""
    from lara_amd import GaussianRasterizer, rasterize_gaussians_views
    act, cams = small_scene(grid=8, size=64, n_views=3, seed=2)
    act = {k: v[:P].contiguous() for k, v in act.items()}
    settings = [raster_settings(c, [0.25, 0.5, 0.75], device=DEV) for c in cams]
    inp = _inputs(act)
    color, radii, allmap = rasterize_gaussians_views(settings, inp["means3D"], None, inp["opacities"], shs=inp["shs"],
                                                     scales=inp["scales"], rotations=inp["rotations"])
    assert color.shape == (3, 3, 64, 64) and radii.shape == (3, P) and torch.isfinite(color).all()
    if P == 0:
        assert torch.equal(color, torch.tensor([0.25, 0.5, 0.75], device=DEV).view(1, 3, 1, 1).expand(3, 3, 64, 64))
        assert float(allmap.detach().abs().max()) == 0.0
    (color.sum() + allmap.sum()).backward()
    torch.cuda.synchronize()
    want = None
    for rs in settings:
        ref = _inputs(act)
        c, r, a = GaussianRasterizer(rs)(means3D=ref["means3D"], means2D=None, shs=ref["shs"], opacities=ref["opacities"],
                                         scales=ref["scales"], rotations=ref["rotations"])
        (c.sum() + a.sum()).backward()
        gi = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in ref.items()}
        want = gi if want is None else {k: want[k] + gi[k] for k in gi}
    for k in want:
        got = inp[k].grad if inp[k].grad is not None else torch.zeros_like(inp[k])
        assert torch.equal(got, want[k]), k


def test_views_from_two_caller_streams_at_once():
    """Two scenes issued from two HIP streams (what the bench's scene streams do): the library keeps a lane pool per
    caller stream, the calls must not disturb each other -- same bits as issued one after the other."""
    from lara_amd import rasterize_gaussians_views
    scenes = [small_scene(grid=12, size=96, n_views=4, seed=s) for s in (5, 6)]
    settings = [[raster_settings(c, [1.0, 1.0, 1.0], device=DEV) for c in cams] for _, cams in scenes]

    def run(streams):
        outs, leaves = [], []
        cur = torch.cuda.current_stream()
        for i, (act, _) in enumerate(scenes):
            inp = _inputs(act)
            leaves.append(inp)
            st = streams[i] if streams else None
            if st is not None:
                st.wait_stream(cur)
            with torch.cuda.stream(st) if st is not None else torch.cuda.stream(cur):
                c, r, a = rasterize_gaussians_views(settings[i], inp["means3D"], None, inp["opacities"], shs=inp["shs"],
                                                    scales=inp["scales"], rotations=inp["rotations"])
                outs.extend((c, a))
        for st in streams or []:
            cur.wait_stream(st)
        torch.autograd.backward(outs, [torch.ones_like(o) for o in outs])
        for st in streams or []:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        return [o.detach().clone() for o in outs], [{k: v.grad.clone() for k, v in l.items()} for l in leaves]

    o1, g1 = run(None)
    o2, g2 = run([torch.cuda.Stream(), torch.cuda.Stream()])
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    for ga, gb in zip(g1, g2):
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k


def test_views_argument_errors():
    from lara_amd import rasterize_gaussians_views
    act, cams = small_scene(grid=8, size=64, n_views=2, seed=1)
    t = {k: v.to(DEV) for k, v in act.items()}
    s_ok = raster_settings(cams[0], [1.0, 1.0, 1.0], device=DEV)
    s_other = raster_settings(cams[1], [1.0, 1.0, 1.0], device=DEV)._replace(image_height=48)
    m2 = torch.zeros_like(t["means3D"])
    with pytest.raises(RuntimeError, match="must agree"):
        rasterize_gaussians_views([s_ok, s_other], t["means3D"], m2, t["opacities"], shs=t["shs"], scales=t["scales"],
                                  rotations=t["rotations"])
    s_scaled = raster_settings(cams[1], [1.0, 1.0, 1.0], device=DEV)._replace(scale_modifier=0.5)
    with pytest.raises(RuntimeError, match="must agree"):   # the batched preprocess runs all cameras with one scale_modifier
        rasterize_gaussians_views([s_ok, s_scaled], t["means3D"], m2, t["opacities"], shs=t["shs"], scales=t["scales"],
                                  rotations=t["rotations"])
    with pytest.raises(Exception, match="excatly one"):
        rasterize_gaussians_views([s_ok], t["means3D"], m2, t["opacities"], scales=t["scales"], rotations=t["rotations"])
    with pytest.raises(RuntimeError, match="at least one view"):
        rasterize_gaussians_views([], t["means3D"], m2, t["opacities"], shs=t["shs"], scales=t["scales"],
                                  rotations=t["rotations"])


def test_render_views_matches_render_img_loop():
    """`Renderer.render_views` = the reference loop's n `render_img` calls (network.py:486-497)."""
    from lara_amd import synthetic
    from lara_amd.renderer import Renderer
    from lara_amd import cameras
    import math
    sc = synthetic.make_scene(grid=12, K=2, regime="init", seed=2)
    size, n = 96, 4
    cams = cameras.make_cameras(cameras.turntable_c2w(n), size, size, 0.75, 0.75, 0.5, 2.5, device=DEV)
    raw = {k: v.to(DEV) for k, v in sc.items()}
    g = torch.Generator().manual_seed(9)
    rays = [torch.nn.functional.normalize(torch.randn(size, size, 6, generator=g), dim=-1).to(DEV) for _ in cams]
    bgs = [torch.tensor([1.0, 1.0, 1.0]), torch.tensor([0.0, 0.0, 0.0]), torch.tensor([0.5, 0.5, 0.5]), torch.tensor([1.0, 1.0, 1.0])]

    def leafs():
        return {k: v.clone().requires_grad_(True) for k, v in raw.items()}

    ren = Renderer(sh_degree=1, white_background=True)
    a = leafs()
    frames = []
    for cam, r, bg in zip(cams, rays, bgs):
        ren.set_bg_color(bg)
        frames.append(ren.render_img(cam, r, a["centers"], a["shs"], a["opacity"], a["scales"], a["rotations"], DEV))
    sum(f["image"].sum() + f["depth"].sum() + 0.1 * f["rend_normal"].sum() for f in frames).backward()
    b = leafs()
    ren2 = Renderer(sh_degree=1, white_background=True)
    frames2 = ren2.render_views(cams, rays, b["centers"], b["shs"], b["opacity"], b["scales"], b["rotations"], DEV, bg_colors=bgs)
    sum(f["image"].sum() + f["depth"].sum() + 0.1 * f["rend_normal"].sum() for f in frames2).backward()
    torch.cuda.synchronize()
    for f, f2 in zip(frames, frames2):
        for k in f:
            assert torch.equal(f[k], f2[k]), k
    for k in a:
        # same per-view gradients, added in a different order by autograd (loop) and by the library (views)
        ga, gb = a[k].grad, b[k].grad
        assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-6 * float(ga.abs().max())), k


def test_state_and_scratch_contents_do_not_matter(monkeypatch):
    """LARA2DGS_POISON_BUFFERS=1 fills every state / scratch buffer with 0xFF before the library sees it (the caching
    allocator otherwise hands back blocks that still hold a previous call's, valid-looking, contents): same bits forward
    and backward, per-view operator and multi-view call -- no kernel reads a field before it is written; the buffers then also
    sit between 0xFF guard zones, which must survive (no write outside a buffer; a read outside would see poison)."""
    from lara_amd import GaussianRasterizer, rasterize_gaussians_views
    act, cams = small_scene(grid=12, size=96, n_views=5, seed=8)
    settings = [raster_settings(c, [1.0, 0.5, 0.0], device=DEV) for c in cams]

    def run():
        inp = _inputs(act)
        c, r, a = rasterize_gaussians_views(settings, inp["means3D"], None, inp["opacities"], shs=inp["shs"],
                                            scales=inp["scales"], rotations=inp["rotations"])
        (c.sum() + 0.1 * a.sum()).backward()
        one = _inputs(act)
        c1, r1, a1 = GaussianRasterizer(settings[2])(means3D=one["means3D"], means2D=None, shs=one["shs"], opacities=one["opacities"],
                                                     scales=one["scales"], rotations=one["rotations"])
        (c1.sum() + 0.1 * a1.sum()).backward()
        torch.cuda.synchronize()
        return [c.detach().clone(), a.detach().clone(), r.clone(), c1.detach().clone(), r1.clone()] + \
               [v.grad.clone() for v in inp.values()] + [v.grad.clone() for v in one.values()]

    from lara_amd import rasterizer
    clean = run()
    monkeypatch.setenv("LARA2DGS_POISON_BUFFERS", "1")
    poisoned = run()
    assert rasterizer.check_poison_guards() == []       # ... and none writes beyond either end of a buffer
    for x, y in zip(clean, poisoned):
        assert torch.equal(x, y)


@pytest.mark.parametrize("keep,scale_boost", [(0.5, 1.0), (0.93, 3.0), (0.02, 3.0)])
def test_a_subset_call_filters_the_earlier_calls_lists_bit_for_bit(hip_lib, keep, scale_boost):
    """LaRa's fine pass (network.py:502-525) renders `x[mask]` of the coarse pass's Gaussians from the same cameras with refined SH
    coefficients.  `rasterize_gaussians_views(..., subset_of=(coarse colour, idx))` builds its per-tile lists by a stable filter
    of the coarse call's (the sort key is (depth bits, id), the subset's numbering is monotone): images, radii, `point_list`,
    `ranges`, the surfel-major pair map and every gradient must equal the full path's -- scatter + sort again -- BIT FOR BIT,
    including lists several thousand entries deep (scale_boost 3) and a nearly empty subset."""
    from lara_amd import rasterize_gaussians_views, rasterizer
    act, cams = small_scene(grid=20, size=96, seed=11, scale_boost=scale_boost, opacity_boost=-1.0 if scale_boost > 1 else 0.0)
    n, S = 4, 96
    settings = [raster_settings(c, bg, device=DEV) for c, bg in zip(cams[:n], ((1, 1, 1), (0, 0, 0), (.5, .5, .5), (1, 1, 1)))]
    P = act["means3D"].shape[0]
    g = torch.Generator().manual_seed(3)
    idx = (torch.rand(P, generator=g) < keep).nonzero().squeeze(-1).to(DEV)
    assert 0 < idx.numel() < P
    coarse = {k: v.to(DEV).clone().requires_grad_(True) for k, v in act.items()}
    c_color, c_radii, c_allmap = rasterize_gaussians_views(settings, coarse["means3D"], None, coarse["opacities"], shs=coarse["shs"],
                                                           scales=coarse["scales"], rotations=coarse["rotations"])
    dc = torch.randn(n, 3, S, S, generator=g).to(DEV)
    da = (0.1 * torch.randn(n, 7, S, S, generator=g)).to(DEV)

    def fine(subset_of):
        t = {k: v.detach()[idx].clone().requires_grad_(True) for k, v in coarse.items()}
        with torch.no_grad():
            t["shs"].add_(0.05)                 # the refined coefficients: the only thing that differs from the coarse pass
        color, radii, allmap = rasterize_gaussians_views(settings, t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"],
                                                         rotations=t["rotations"], subset_of=subset_of)
        node = color.grad_fn
        v0 = rasterizer.state_views(node.state.view(n, node.strides[0])[2], idx.numel(), S, S, node.cap)
        snap = {k: v0[k].clone() for k in ("header", "point_list", "ranges", "pair_pos", "pair_base", "geom", "tile_order", "seg_cnt")}
        torch.autograd.backward([color, allmap], [dc, da])
        return color.detach(), radii, allmap.detach(), {k: v.grad for k, v in t.items()}, snap, node.D

    full = fine(None)
    filt = fine((c_color, idx))
    D = full[5]
    assert filt[5] == D and D > 0
    for a, b, name in zip(full[:3], filt[:3], ("color", "radii", "allmap")):
        assert torch.equal(a, b), name
    for k in full[3]:
        assert torch.equal(full[3][k], filt[3][k]), f"grad {k}"
    Dv = int(full[4]["header"][0])
    for k in ("ranges", "geom", "seg_cnt", "pair_base"):
        assert torch.equal(full[4][k], filt[4][k]), k
    assert torch.equal(full[4]["point_list"][:Dv], filt[4]["point_list"][:Dv])
    assert torch.equal(full[4]["pair_pos"][:Dv], filt[4]["pair_pos"][:Dv])
    if scale_boost > 1 and keep > 0.9:
        assert int(full[4]["header"][2]) > 2048, "the case must hold lists several rounds deep"
    # the coarse call's own backward is untouched by having been read
    torch.autograd.backward([c_color, c_allmap], [dc, da])
    assert all(torch.isfinite(v.grad).all() for v in coarse.values())


def test_a_subset_call_that_outgrows_its_buffers_is_repeated_like_any_other(hip_lib, monkeypatch):
    from lara_amd import rasterize_gaussians_views, rasterizer
    act, cams = small_scene(grid=16, size=128, seed=0, scale_boost=3.0)
    settings = [raster_settings(c, (1, 1, 1), device=DEV) for c in cams[:2]]
    P = act["means3D"].shape[0]
    idx = torch.arange(0, P, 2, device=DEV)
    coarse = {k: v.to(DEV).clone().requires_grad_(True) for k, v in act.items()}
    c_color, _, _ = rasterize_gaussians_views(settings, coarse["means3D"], None, coarse["opacities"], shs=coarse["shs"],
                                              scales=coarse["scales"], rotations=coarse["rotations"])
    t = {k: v.detach()[idx].clone().requires_grad_(True) for k, v in coarse.items()}
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    want, _, _ = rasterize_gaussians_views(settings, t["means3D"], None, t["opacities"], **kw)
    monkeypatch.setattr(rasterizer, "_next_capacity", lambda bucket: 4096)      # far too small: the subset call must come back repaired
    before = rasterizer._reruns
    got, _, _ = rasterize_gaussians_views(settings, t["means3D"], None, t["opacities"], subset_of=(c_color, idx), **kw)
    assert rasterizer._reruns == before + 1 and torch.equal(got, want)
    rasterizer.reset_capacity_history()
