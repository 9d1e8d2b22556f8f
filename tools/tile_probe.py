"""Per-wave cycle counts of composite_fwd (LARA2DGS_DEBUG_FLAGS=32): heaviest wave vs aggregate work.
Run on the GPU box:  LARA2DGS_DEBUG_FLAGS=32 python tools/tile_probe.py"""
import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from lara_amd import cameras, synthetic, rasterizer, GaussianRasterizationSettings
dev=torch.device('cuda:0')
cams = cameras.make_cameras(cameras.turntable_c2w(8), 512, 512, 0.75, 0.75, 1.106, 2.706, device=dev)
for regime in ('trained','init'):
    sc = synthetic.make_scene(grid=64, K=2, regime=regime, seed=0, device=dev)
    act = synthetic.activate(sc)
    cam=cams[3]
    rs = GaussianRasterizationSettings(512,512, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0, cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(), 1, cam.camera_center, False, False)
    for _ in range(2):
        r = rasterizer.forward_with_state(rs, act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"])
    torch.cuda.synchronize()
    am = r["allmap"].cpu().numpy()
    # lane 0 of wave w of tile (tx,ty) is pixel (tx*16 + (w&1)*8, ty*16 + (w>>1)*8)
    cyc = am[6][::8, ::8]; rounds = am[5][::8, ::8]
    ranges = r["views"]["ranges"].cpu().numpy(); n=(ranges[:,1]-ranges[:,0]).reshape(32,32)
    nc = r["views"]["n_contrib"][0].cpu().numpy()
    print(regime, 'wave cycles: mean %.0f p50 %.0f p90 %.0f max %.0f' % (cyc.mean(), np.percentile(cyc,50), np.percentile(cyc,90), cyc.max()))
    idx = np.argsort(-cyc.ravel())[:8]
    for i in idx:
        wy, wx = divmod(i, 64); ty, tx = wy//2, wx//2
        blk = nc[ty*16:(ty+1)*16, tx*16:(tx+1)*16]
        print('   wave(%d,%d) tile(%d,%d) cycles %.0f rounds %.0f list %d  tile n_contrib max %d mean %.0f' % (wx,wy,tx,ty,cyc.ravel()[i], rounds.ravel()[i], n[ty,tx], blk.max(), blk.mean()))
