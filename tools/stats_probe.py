"""Walk statistics of composite_fwd (LARA2DGS_DEBUG_FLAGS=4): quad candidates, valid (pixel, entry) pairs,
wave iterations, tile list lengths.  Run on the GPU box:  LARA2DGS_DEBUG_FLAGS=4 python tools/stats_probe.py"""
import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, rasterizer, GaussianRasterizationSettings
dev=torch.device('cuda:0')
cams = cameras.make_cameras(cameras.turntable_c2w(8), 512, 512, 0.75, 0.75, 1.106, 2.706, device=dev)
for regime in ('init','trained'):
    sc = synthetic.make_scene(grid=64, K=2, regime=regime, seed=0, device=dev)
    act = synthetic.activate(sc)
    cam=cams[3]
    rs = GaussianRasterizationSettings(512,512, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0, cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(), 1, cam.camera_center, False, False)
    r = rasterizer.forward_with_state(rs, act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"])
    torch.cuda.synchronize()
    h = r["views"]["header"].cpu().numpy().astype('int64') & 0xffffffff
    D=h[0]
    print(regime, 'D',D,'entries*4 quadrants',4*D,'candidates',h[4], 'frac', h[4]/(4*D), 'valid pairs',h[5],'lane eff', h[5]/(64*h[4]), 'live pairs', h[6], h[6]/(64*h[4]))
    ranges = r["views"]["ranges"].cpu().numpy()
    n = ranges[:,1]-ranges[:,0]
    import numpy as np
    print('  list len: mean', n.mean(), 'nonzero tiles', (n>0).sum(), 'p50', np.percentile(n[n>0],50), 'p90', np.percentile(n[n>0],90), 'max', n.max())
