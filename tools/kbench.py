#!/usr/bin/env python
"""Per-kernel HIP-event timings of one scene x 8 views (fwd+bwd) for both synthetic regimes.
Run on the GPU box:  python tools/kbench.py [--reps 3]"""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, rasterizer, GaussianRasterizer, GaussianRasterizationSettings

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--grid", type=int, default=64)
ap.add_argument("--regimes", default="init,trained")
ap.add_argument("--color-only", action="store_true", help="no gradient on the seven maps (LaRa's fine pass): composite_bwd's colour-only form")
args = ap.parse_args()
dev = torch.device("cuda:0")
rasterizer.load_library()
cams = cameras.make_cameras(cameras.turntable_c2w(8), args.res, args.res, 0.75, 0.75, 1.106, 2.706, device=dev)
for regime in args.regimes.split(","):
    sc = synthetic.make_scene(grid=args.grid, K=2, regime=regime, seed=0, device=dev)
    act = {k: v.requires_grad_(True) for k, v in synthetic.activate(sc).items()}
    gc = torch.randn(3, args.res, args.res, device=dev) / args.res ** 2
    ga = torch.randn(7, args.res, args.res, device=dev) / args.res ** 2 * 0.1
    agg = {}
    for rep in range(args.reps + 1):
        if rep == 1:
            rasterizer.profile_enable(True)
        outs, grads = [], []
        for cam in cams:
            rs = GaussianRasterizationSettings(args.res, args.res, math.tan(0.375), math.tan(0.375),
                                               torch.ones(3, device=dev), 1.0, cam.world_view_transform.contiguous(),
                                               cam.full_proj_transform.contiguous(), 1, cam.camera_center, False, False)
            color, radii, allmap = GaussianRasterizer(rs)(means3D=act["means3D"], means2D=torch.zeros_like(act["means3D"]),
                                                          shs=act["shs"], opacities=act["opacities"],
                                                          scales=act["scales"], rotations=act["rotations"])
            outs += [color] if args.color_only else [color, allmap]
            grads += [gc] if args.color_only else [gc, ga]
        torch.autograd.backward(outs, grads)
        torch.cuda.synchronize()
    for name, ms in rasterizer.profile_collect():
        a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += ms
    rasterizer.profile_enable(False)
    r = rasterizer.forward_with_state(rs, act["means3D"].detach(), act["opacities"].detach(), shs=act["shs"].detach(),
                                      scales=act["scales"].detach(), rotations=act["rotations"].detach())
    torch.cuda.synchronize()
    hdr = r["views"]["header"].cpu()
    nc = r["views"]["n_contrib"][0].float()
    tot = sum(t / n for n, t in agg.values())
    print(f"[{regime}] D={int(hdr[0])} max_list={int(hdr[2])} mean n_contrib={float(nc.mean()):.1f} "
          f"sum_kernels={tot*1e3:.0f} us/frame -> {1e3/tot:.0f} frames/s (kernel time only)")
    for k, (n, t) in agg.items():
        print(f"   {k:18s} {1e3*t/n:9.1f} us  x{n}")
