import sys, math, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, rasterizer, GaussianRasterizationSettings
dev=torch.device('cuda:0')
cams = cameras.make_cameras(cameras.turntable_c2w(8), 512, 512, 0.75, 0.75, 1.106, 2.706, device=dev)
for regime in ('init',):
    sc = synthetic.make_scene(grid=64, K=2, regime=regime, seed=0, device=dev)
    act = synthetic.activate(sc)
    cam=cams[3]
    rs = GaussianRasterizationSettings(512,512, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0, cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(), 1, cam.camera_center, False, False)
    r = rasterizer.forward_with_state(rs, act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"])
    torch.cuda.synchronize()
    v=r["views"]; D=int(v["header"][0])
    ranges=v["ranges"].long(); plist=v["point_list"][:D].long(); cb=v["cullbox"]
    n=(ranges[:,1]-ranges[:,0])
    tile_of=torch.repeat_interleave(torch.arange(1024,device=dev), n)
    X0=(tile_of%32*16).float(); Y0=(tile_of//32*16).float()
    b=cb[plist]  # minx,maxx,miny,maxy
    for G in (8,4,2):
        nb=16//G
        tot=torch.zeros(1024, nb*nb, device=dev)
        for by in range(nb):
            for bx in range(nb):
                x0=X0+bx*G; y0=Y0+by*G
                ov=(b[:,0]<=x0+G-1)&(b[:,1]>=x0)&(b[:,2]<=y0+G-1)&(b[:,3]>=y0)
                tot[:,by*nb+bx].index_add_(0, tile_of, ov.float())
        cand=tot.sum().item()
        # group blocks into waves of 64 pixels: (64/(G*G)) blocks per wave -> wave iterations = max over its blocks
        per_wave=64//(G*G)
        # choose blocks forming 8x8 quadrant: indices
        t=tot.view(1024, nb, nb)
        q=8//G
        waves=t.view(1024, nb//q, q, nb//q, q).permute(0,1,3,2,4).reshape(1024, -1, q*q)
        it=waves.max(dim=2).values.sum().item()
        print(f"G={G}: candidates (entry,block) {cand/1e6:.2f}M lane-evals {cand*G*G/1e6:.1f}M ; wave iterations with row-SIMT {it/1e6:.2f}M (lane slots {it*64/1e6:.1f}M)")
