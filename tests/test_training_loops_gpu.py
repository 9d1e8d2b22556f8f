"""End-to-end learning checks on the GPU (no oracle involved): gradients that pass the parity tests must also
*train*.  (1) the rasteriser: recover a scene's colours / opacities / geometry from its own renders with Adam, the
way LaRa's loss drives the decoder through `Renderer.render_img`; (2) the trainable volume transformer: fit a fixed
target volume.  Both must cut their loss by a large factor in a few dozen steps."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_rasteriser_gradients_fit_a_scene(hip_lib):
    from lara_amd import cameras, synthetic
    from lara_amd.renderer import Renderer
    torch.manual_seed(0)
    gt = synthetic.make_scene(grid=12, K=2, regime="trained", seed=4, device=DEV)
    gt["scales"] = gt["scales"] + math.log(64 / 12)            # keep the splats' pixel footprint at this grid size
    cams = cameras.make_cameras(cameras.turntable_c2w(4), 96, 96, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8, device=DEV)
    rays = torch.cat([torch.zeros(96, 96, 3), F.normalize(torch.randn(96, 96, 3), dim=-1)], -1).to(DEV)
    r = Renderer(sh_degree=1, white_background=True)
    with torch.no_grad():
        targets = [r.render_img(c, rays, gt["centers"], gt["shs"], gt["opacity"], gt["scales"], gt["rotations"], DEV) for c in cams]
    # start from perturbed colours, opacities, scales and positions
    p = {k: v.clone() for k, v in gt.items()}
    p["shs"] = p["shs"] + 0.6 * torch.randn_like(p["shs"])
    p["opacity"] = p["opacity"] - 1.0
    p["scales"] = p["scales"] + 0.2 * torch.randn_like(p["scales"])
    p["centers"] = p["centers"] + 0.004 * torch.randn_like(p["centers"])
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam([{"params": [p["shs"]], "lr": 5e-2}, {"params": [p["opacity"]], "lr": 5e-2},
                            {"params": [p["scales"]], "lr": 1e-2}, {"params": [p["centers"]], "lr": 2e-4},
                            {"params": [p["rotations"]], "lr": 1e-3}])

    def loss_fn():
        tot = 0
        for c, t in zip(cams, targets):
            o = r.render_img(c, rays, p["centers"], p["shs"], p["opacity"], p["scales"], p["rotations"], DEV)
            tot = tot + F.mse_loss(o["image"], t["image"]) + 0.1 * F.mse_loss(o["acc_map"], t["acc_map"]) \
                + 0.1 * F.mse_loss(o["depth"], t["depth"])
        return tot / len(cams)

    first = None
    for it in range(60):
        opt.zero_grad()
        loss = loss_fn()
        loss.backward()
        opt.step()
        first = float(loss) if first is None else first
    last = float(loss_fn())
    assert math.isfinite(last) and last < 0.12 * first, (first, last)


def test_trainable_voltransformer_fits_a_target(hip_lib):
    from lara_amd.encoder_train import VolTransformer
    torch.manual_seed(1)
    vt = VolTransformer(embed_dim=256, image_feat_dim=800, n_groups=[2], vol_low_res=4, vol_high_res=8, out_dim=80,
                        num_layers=2, num_heads=16).to(DEV)
    feats = torch.randn(2, 4, 800, 2, 2, 2, device=DEV)
    target = 0.5 * torch.randn(2, 8, 8, 8, 80, device=DEV)
    opt = torch.optim.Adam(vt.parameters(), lr=2e-3)
    first = None
    for it in range(40):
        opt.zero_grad()
        loss = F.mse_loss(vt(feats), target)
        loss.backward()
        opt.step()
        first = float(loss) if first is None else first
    with torch.no_grad():
        last = float(F.mse_loss(vt(feats), target))
    assert math.isfinite(last) and last < 0.5 * first, (first, last)


def test_ops_are_unaffected_by_bf16_autocast(hip_lib):
    """LaRa trains under Lightning's bf16-mixed precision (train_lightning.py:74), i.e. inside torch.autocast: the
    custom ops take fp32 (or cast what they get) and must give the same bits inside and outside the context, with
    half-precision inputs accepted."""
    from lara_amd import cameras, synthetic
    from lara_amd.encoder_train import VolTransformer
    from lara_amd.renderer import Renderer
    sc = synthetic.make_scene(grid=10, K=2, seed=2, device=DEV)
    sc["scales"] = sc["scales"] + math.log(64 / 10)
    cam = cameras.make_cameras(cameras.turntable_c2w(4), 64, 64, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8, device=DEV)[1]
    rays = torch.cat([torch.zeros(64, 64, 3), F.normalize(torch.randn(64, 64, 3), dim=-1)], -1).to(DEV)
    r = Renderer(sh_degree=1, white_background=True)
    torch.manual_seed(0)
    vt = VolTransformer(256, 800, [2], 4, 8, 80, 1, 16).to(DEV)
    feats = torch.randn(1, 4, 800, 2, 2, 2, device=DEV)

    def run(autocast):
        p = {k: v.clone().requires_grad_(True) for k, v in sc.items()}
        f = feats.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            shs = p["shs"].to(torch.bfloat16).float() if autocast else p["shs"].to(torch.bfloat16).float()
            o = r.render_img(cam, rays, p["centers"], shs, p["opacity"], p["scales"], p["rotations"], DEV)
            vol = vt(f.to(torch.bfloat16) if autocast else f.to(torch.bfloat16).float())
            loss = o["image"].sum() + o["depth"].sum() + o["depth_normal"].sum() + vol.float().pow(2).mean()
        loss.backward()
        return [o["image"].detach(), o["depth"].detach(), vol.detach(), p["centers"].grad, p["shs"].grad, f.grad,
                vt.layers[0].cnn.weight.grad.clone()]

    a = run(False)
    vt.zero_grad()
    b = run(True)
    for x, y in zip(a, b):
        assert x.dtype == y.dtype and torch.equal(x, y)
