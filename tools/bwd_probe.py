"""Work-group residency statistics of composite_bwd (LARA2DGS_DEBUG_FLAGS=32): is the kernel bound
by its heaviest tile or by aggregate work?  Run on the GPU box."""
import sys, math, os
os.environ["LARA2DGS_DEBUG_FLAGS"] = str(32 | 64 | int(os.environ.get("EXTRA_FLAGS", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, rasterizer, GaussianRasterizer, GaussianRasterizationSettings
dev = torch.device('cuda:0')
cams = cameras.make_cameras(cameras.turntable_c2w(8), 512, 512, 0.75, 0.75, 1.106, 2.706, device=dev)
for regime in ('init', 'trained', 'fine'):  # 'fine': the opacity > 0.005 subset of the trained scene (network.py:465)
    sc = synthetic.make_scene(grid=64, K=2, regime='trained' if regime == 'fine' else regime, seed=0, device=dev)
    act = synthetic.activate(sc)
    if regime == 'fine':
        keep = act["opacities"][:, 0] > 0.005
        act = {k: v[keep].contiguous() for k, v in act.items()}
    act = {k: v.requires_grad_(True) for k, v in act.items()}
    gc = torch.randn(3, 512, 512, device=dev) / 512 ** 2
    ga = torch.randn(7, 512, 512, device=dev) / 512 ** 2 * 0.1
    for ci in (0, 3):
        cam = cams[ci]
        rs = GaussianRasterizationSettings(512, 512, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0,
                                           cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(),
                                           1, cam.camera_center, False, False)
        for rep in range(2):
            color, radii, allmap = GaussianRasterizer(rs)(means3D=act["means3D"], means2D=torch.zeros_like(act["means3D"]),
                                                          shs=act["shs"], opacities=act["opacities"],
                                                          scales=act["scales"], rotations=act["rotations"])
            state, cap = color.grad_fn.state, color.grad_fn.cap
            torch.autograd.backward([color, allmap], [gc, ga])
            torch.cuda.synchronize()
        P = act["means3D"].shape[0]
        h = rasterizer.state_views(state, P, 512, 512, cap)["header"].cpu().numpy().astype('uint32')
        span = (int(h[11]) - (~int(h[10]) & 0xffffffff)) & 0xffffffff
        v = rasterizer.state_views(state, P, 512, 512, cap)
        blk = int(h[12]) & 0xffff; rounds = int(h[12]) >> 16; nfull = int(h[3])
        if blk < nfull:
            tile, seg = [int(x) for x in v["bwd_items"][blk].cpu()]
        else:
            tile = int(v["bwd_order"][blk - nfull]); seg = int(v["seg_cnt"][tile])
        rg = v["ranges"][tile].cpu().numpy(); ids = v["point_list"][int(rg[0]) + seg * 512: min(int(rg[0]) + seg * 512 + 512, int(rg[1]))].long()
        cb = v["cullbox"][ids].cpu().numpy(); nonempty = int(((cb[:, 1] >= cb[:, 0]) & (cb[:, 3] >= cb[:, 2])).sum())
        ty, tx = divmod(tile, 32)
        ncb = v["n_contrib"][0][ty*16:ty*16+16, tx*16:tx*16+16]
        print(f"   slowest WG: {int(h[13])/100:.0f} us, block {blk} tile {tile} seg {seg} list {int(rg[1]-rg[0])} rounds {rounds} "
              f"entries {len(ids)} non-empty cull boxes {nonempty}, mean box {float((cb[:,1]-cb[:,0]).clip(0).mean()):.1f} x {float((cb[:,3]-cb[:,2]).clip(0).mean()):.1f}, tile n_contrib max {int(ncb.max())}")
        ph = [int(x) * 64 / 2.4e3 for x in h[16:21]]  # us at 2.4 GHz, summed over workgroups
        print("   phase time summed over WGs (us): prologue %.0f stage %.0f setup %.0f P %.0f S2 %.0f" % tuple(ph))
        if int(os.environ.get("EXTRA_FLAGS", "0")) & 512:      # walk statistics of the launch (composite.hip, flag 512)
            items, windows, rounds_, trips, entries, slots, rows = [int(x) for x in h[24:31]]
            print(f"   walk: {items} work items, {windows} windows, {rounds_} rounds ({rounds_ / max(windows, 1):.2f} per window), "
                  f"{entries / max(rounds_, 1):.1f} entries and {slots / max(rounds_, 1):.0f} slots per round ({slots / max(entries, 1):.2f} per entry), "
                  f"{trips} wave-trips ({trips / max(4 * rounds_, 1):.2f} per wave and round), {rows} gradient rows")
        print(f"{regime} view {ci}: pairs {h[0]} max_list {h[2]}  WG max {h[8]/100:.0f} us  sum {h[9]/100:.0f} us "
              f"(/512 = {h[9]/100/512:.0f} us)  span {span/100:.0f} us")
