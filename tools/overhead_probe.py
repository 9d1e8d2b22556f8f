"""Host-side cost of one operator call (forward + backward) on a scene small enough that the kernels are idle time:
what the Python wrapper + ctypes + allocator + autograd add per view on the drop-in path.  GPU box."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, GaussianRasterizer, GaussianRasterizationSettings, rasterize_gaussians_views
dev = torch.device("cuda:0")
sc = synthetic.make_scene(grid=8, K=2, regime="init", seed=0, device=dev)
act = {k: v.requires_grad_(True) for k, v in synthetic.activate(sc).items()}
cams = cameras.make_cameras(cameras.turntable_c2w(8), 64, 64, 0.75, 0.75, 1.106, 2.706, device=dev)
rss = [GaussianRasterizationSettings(64, 64, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0,
                                     c.world_view_transform.contiguous(), c.full_proj_transform.contiguous(), 1, c.camera_center, False, False) for c in cams]
m2 = torch.zeros_like(act["means3D"])
def loop():
    outs = []
    for rs in rss:
        c, r, a = GaussianRasterizer(rs)(means3D=act["means3D"], means2D=m2, shs=act["shs"], opacities=act["opacities"], scales=act["scales"], rotations=act["rotations"])
        outs += [c.sum(), a.sum()]
    torch.autograd.backward(outs)
def views():
    c, r, a = rasterize_gaussians_views(rss, act["means3D"], m2, act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"])
    (c.sum() + a.sum()).backward()
for name, fn in (("one call per view", loop), ("one call per scene", views)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50 / 8
    print(f"{name}: {dt * 1e6:.0f} us per frame (fwd+bwd, 1024 surfels, 64x64: wall time is host time)")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50): loop()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
