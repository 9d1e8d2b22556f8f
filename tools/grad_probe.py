#!/usr/bin/env python
"""Where do the full-size gradient differences between the HIP path and the fp32 CPU oracle sit, and are they
conditioning?  View 0 of the bench scene (P = 524 288, 512 x 512), the gradient mix of the parity tests.  Prints,
per parameter: max / L2 error, the share of surfels above 1e-3 of max, and the oracle's OWN sensitivity -- its
gradient after every input was moved by one part in 2^23 (one fp32 ulp) -- on the same surfels.  Run on the GPU box."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from lara_amd import cameras, synthetic, GaussianRasterizationSettings, GaussianRasterizer

res, regime = 512, (sys.argv[1] if len(sys.argv) > 1 else "init")
sc = synthetic.make_scene(grid=64, K=2, regime=regime, seed=0)
act = {k: v.numpy() for k, v in synthetic.activate(sc).items()}
cam = cameras.make_cameras(cameras.turntable_c2w(8), res, res, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8)[0]
oracle.build()
g = np.random.default_rng(0)
dc = g.normal(size=(3, res, res)).astype(np.float32)
da = (0.1 * g.normal(size=(7, res, res))).astype(np.float32)


def run_oracle(a):
    view = oracle.View(res, res, math.tan(0.375), math.tan(0.375), np.ones(3, np.float32), 1.0, cam.world_view_transform.numpy(),
                       cam.full_proj_transform.numpy(), 1, cam.camera_center.numpy())
    r = oracle.forward(view, a["means3D"], a["opacities"], shs=a["shs"], scales=a["scales"], rotations=a["rotations"])
    return r, oracle.backward(r, dc, da)


r0, g0 = run_oracle(act)
rng = np.random.default_rng(1)
pert = {k: (v * (1 + (rng.integers(0, 2, v.shape) * 2 - 1) * 2.0 ** -23)).astype(np.float32) for k, v in act.items()}
r1, g1 = run_oracle(pert)
dev = torch.device("cuda:0")
rs = GaussianRasterizationSettings(image_height=res, image_width=res, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375),
                                   bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                   projmatrix=cam.full_proj_transform.to(dev), sh_degree=1, campos=cam.camera_center.to(dev),
                                   prefiltered=False, debug=False)
t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in act.items()}
m2d = torch.zeros_like(t["means3D"], requires_grad=True)
color, radii, allmap = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                                              scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
((color * torch.from_numpy(dc).to(dev)).sum() + (allmap * torch.from_numpy(da).to(dev)).sum()).backward()
print(f"regime {regime}: D = {r0.num_rendered}, colour PSNR HIP vs oracle {10 * math.log10(1 / max(float(((color.detach().cpu().numpy() - r0.color) ** 2).mean()), 1e-30)):.1f} dB, "
      f"oracle vs 1-ulp-perturbed oracle {10 * math.log10(1 / max(float(((r1.color - r0.color) ** 2).mean()), 1e-30)):.1f} dB")
for k in ("means3D", "opacities", "scales", "rotations", "shs"):
    ref = g0[k].astype(np.float64)
    P = ref.shape[0]
    hip = t[k].grad.cpu().numpy().reshape(ref.shape).astype(np.float64)
    per = g1[k].astype(np.float64)
    mx = np.abs(ref).max()
    e_hip = np.abs(hip - ref).reshape(P, -1).max(1) / mx
    e_per = np.abs(per - ref).reshape(P, -1).max(1) / mx
    worst = np.argsort(-e_hip)[:5]
    print(f"{k:10s} HIP-oracle: max {e_hip.max():.2e}  L2 {np.sqrt(((hip - ref) ** 2).sum() / (ref ** 2).sum()):.2e}  "
          f"surfels > 1e-3: {(e_hip > 1e-3).sum()} of {P};   oracle(1 ulp)-oracle: max {e_per.max():.2e}  "
          f"L2 {np.sqrt(((per - ref) ** 2).sum() / (ref ** 2).sum()):.2e}  > 1e-3: {(e_per > 1e-3).sum()};  "
          f"on HIP's worst 5 surfels: HIP {np.array2string(e_hip[worst], precision=4)} vs oracle's own {np.array2string(e_per[worst], precision=4)}")
