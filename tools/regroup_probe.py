#!/usr/bin/env python
"""How unevenly a backward window's candidates fall on the quads of a wave -- from the forward's stored candidate masks
(`pair_mask`: per list position, which 2x2 blocks of the tile may see the entry).  A wave of composite_bwd walks for as
many trips as its busiest quad has candidates in the round; this prints the wave-trips of (a) the fixed quadrant mapping
and (b) the same blocks sorted by candidate count and dealt to the four waves in that order, per 128-entry window.
Run on the GPU box."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lara_amd import cameras, synthetic, rasterizer, GaussianRasterizationSettings

dev = torch.device("cuda:0")
res = 512
for regime in ("init", "trained"):
    sc = synthetic.make_scene(grid=64, K=2, regime=regime, seed=0, device=dev)
    act = synthetic.activate(sc)
    cam = cameras.make_cameras(cameras.turntable_c2w(8), res, res, 0.75, 0.75, 1.106, 2.706, device=dev)[0]
    rs = GaussianRasterizationSettings(res, res, math.tan(0.375), math.tan(0.375), torch.ones(3, device=dev), 1.0,
                                       cam.world_view_transform.contiguous(), cam.full_proj_transform.contiguous(), 1,
                                       cam.camera_center, False, False)
    r = rasterizer.forward_with_state(rs, act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"])
    torch.cuda.synchronize()
    v = r["views"]
    ranges = v["ranges"].cpu().long()
    masks = v["pair_mask"]
    last = v["n_contrib"][0].view(torch.int32).long().view(res // 16, 16, res // 16, 16).permute(0, 2, 1, 3).reshape(-1, 256).max(1).values.cpu()
    bits = torch.arange(64, device=dev)
    quadrant = ((bits // 8) // 4) * 2 + (bits % 8) // 4          # wave of block (gy*8+gx) in the fixed mapping
    tot_now = tot_sorted = tot_ideal = tot_seg = 0
    tot_w = {256: 0.0, 512: 0.0}
    for t in range(ranges.shape[0]):
        s, e = int(ranges[t, 0]), int(ranges[t, 1])
        n = min(e - s, int(last[t]))          # the backward only visits entries below the tile's last contributor
        if n <= 0:
            continue
        m = masks[s:s + n]
        cand = ((m[:, None] >> bits[None, :]) & 1).float()                       # [n, 64]
        pad = (-n) % 128
        if pad:
            cand = torch.cat([cand, cand.new_zeros(pad, 64)])
        c = cand.view(-1, 128, 64).sum(1)                                        # [windows, 64] candidates per block
        now = torch.stack([c[:, quadrant == w].max(1).values for w in range(4)], 1).sum(1)
        srt = c.sort(1, descending=True).values.view(-1, 4, 16).max(2).values.sum(1)
        tot_now += float(now.sum()); tot_sorted += float(srt.sum()); tot_ideal += float(c.sum()) / 16
        # (c) STATIC deal per 512-entry segment (the backward's work item): blocks ranked once by their candidate count over
        # the whole segment, ranks 0-15 -> one wave, 16-31 -> the next ...; trips still counted per 128-entry window
        nw = c.shape[0]
        padw = (-nw) % 4
        c4 = (torch.cat([c, c.new_zeros(padw, 64)]) if padw else c).view(-1, 4, 64)     # [segments, 4 windows, 64]
        order = c4.sum(1).argsort(1, descending=True)                                       # [segments, 64]
        dealt = torch.gather(c4, 2, order[:, None, :].expand(-1, 4, -1)).view(-1, 4, 4, 16)
        tot_seg += float(dealt.max(3).values.sum())
        for W in tot_w:       # the fixed mapping with longer rounds (what a bigger slot pool would buy)
            k = W // 128
            padw = (-c.shape[0]) % k
            cw = torch.cat([c, c.new_zeros(padw, 64)]) if padw else c
            cw = cw.view(-1, k, 64).sum(1)
            tot_w[W] += float(torch.stack([cw[:, quadrant == w].max(1).values for w in range(4)], 1).sum())
    print(f"[{regime}] wave-trips per frame: fixed quadrants {tot_now:.3e}, blocks sorted by count {tot_sorted:.3e} "
          f"({tot_sorted / tot_now:.3f}x), blocks dealt ONCE per 512-entry segment by their count over the segment {tot_seg:.3e} "
          f"({tot_seg / tot_now:.3f}x), perfectly even {tot_ideal:.3e} ({tot_ideal / tot_now:.3f}x); fixed quadrants with "
          f"256-entry rounds {tot_w[256] / tot_now:.3f}x, 512-entry rounds {tot_w[512] / tot_now:.3f}x")
