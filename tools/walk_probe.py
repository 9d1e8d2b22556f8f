import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, rasterizer, GaussianRasterizationSettings
dev=torch.device('cuda:0')
cams = cameras.make_cameras(cameras.turntable_c2w(8), 512, 512, 0.75, 0.75, 1.106, 2.706, device=dev)
sc = synthetic.make_scene(grid=64, K=2, regime='init', seed=0, device=dev)
act = synthetic.activate(sc)
for name, op in (("all culled (opacity 0.001)", torch.full_like(act["opacities"], 0.001)),
                 ("all opaque (0.98)", torch.full_like(act["opacities"], 0.98)),
                 ("init", act["opacities"])):
    rasterizer.profile_enable(True)
    for rep in range(3):
        for cam in cams[:4]:
            rs = GaussianRasterizationSettings(512,512, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0, cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(), 1, cam.camera_center, False, False)
            r = rasterizer.forward_with_state(rs, act["means3D"], op, shs=act["shs"], scales=act["scales"], rotations=act["rotations"])
    torch.cuda.synchronize()
    agg={}
    for k,ms in rasterizer.profile_collect():
        a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=ms
    rasterizer.profile_enable(False)
    print(name, {k: round(1e3*t/n,1) for k,(n,t) in agg.items() if k in ("composite_fwd","tile_sort_large","tile_sort_small")}, 'mean n_contrib', float(r["views"]["n_contrib"][0].float().mean()))
