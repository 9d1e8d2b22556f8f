#!/usr/bin/env python
"""Who is closer to the truth at the benchmark size -- the HIP backward or the fp32 CPU oracle?

At P = 524 288, 512 x 512 the HIP gradients sit up to 2-3e-2 of max from the fp32 oracle's on a handful of surfels
(DESIGN.md section 5), about 2x the oracle's own one-ulp sensitivity.  That says the entries are ill-conditioned, not
which side is right.  This tool arbitrates with fp64:
  1. full frame, HIP vs fp32 oracle: the `n_worst` surfels per gradient tensor with the largest disagreement;
  2. the tiles whose lists hold those surfels; upstream gradients masked to those tiles (a surfel's gradient is the sum
     over the tiles it touches, so for the chosen surfels nothing is lost);
  3. the same masked backward three ways: HIP, fp32 oracle, and `oracle/autograd_ref.py` in fp64 (torch autograd over
     the same sorted lists, published low-pass quirk included) on those tiles;
  4. per tensor: |HIP - fp64| and |fp32 oracle - fp64| (max over entries relative to max|fp64|, and relative L2), over
     all surfels the tiles touch and on the chosen surfels alone.
usage (GPU box): python tools/grad_arbiter.py [--worst 3] [--regime init] [--json out.json]"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

KEYS = ("means3D", "opacities", "scales", "rotations", "shs")


def arbitrate(n_worst=3, regime="init", res=512, grid=64, max_tiles=40, verbose=False):
    import oracle
    from oracle import autograd_ref
    from lara_amd import cameras, synthetic, GaussianRasterizationSettings, GaussianRasterizer
    t_start = time.perf_counter()
    sc = synthetic.make_scene(grid=grid, K=2, regime=regime, seed=0)
    act = {k: v.numpy() for k, v in synthetic.activate(sc).items()}
    cam = cameras.make_cameras(cameras.turntable_c2w(8), res, res, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8)[0]
    oracle.build()
    rng = np.random.default_rng(0)
    dc = rng.normal(size=(3, res, res)).astype(np.float32)
    da = (0.1 * rng.normal(size=(7, res, res))).astype(np.float32)
    view = oracle.View(res, res, math.tan(0.375), math.tan(0.375), np.ones(3, np.float32), 1.0, cam.world_view_transform.numpy(),
                       cam.full_proj_transform.numpy(), 1, cam.camera_center.numpy())
    r = oracle.forward(view, act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"], rotations=act["rotations"])
    dev = torch.device("cuda:0")
    rs = GaussianRasterizationSettings(image_height=res, image_width=res, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375),
                                       bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev),
                                       projmatrix=cam.full_proj_transform.to(dev), sh_degree=1, campos=cam.camera_center.to(dev),
                                       prefiltered=False, debug=False)

    def hip(dc_, da_):
        t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in act.items()}
        color, radii, allmap = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), shs=t["shs"],
                                                      opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                                      cov3D_precomp=None)
        ((color * torch.from_numpy(dc_).to(dev)).sum() + (allmap * torch.from_numpy(da_).to(dev)).sum()).backward()
        g = {k: t[k].grad.cpu().numpy().astype(np.float64) for k in KEYS}
        g["_color"], g["_allmap"] = color.detach().cpu().numpy(), allmap.detach().cpu().numpy()
        return g

    g_or = oracle.backward(r, dc, da)
    g_hip = hip(dc, da)
    chosen = []
    for k in KEYS:
        ref = g_or[k].astype(np.float64)
        e = np.abs(g_hip[k].reshape(ref.shape) - ref).reshape(ref.shape[0], -1).max(1)
        chosen.extend(int(i) for i in np.argsort(-e)[:n_worst])
    chosen = sorted(set(chosen))
    # tiles whose sorted lists hold a chosen surfel
    pl = r.point_list[:r.num_rendered]
    pos = np.nonzero(np.isin(pl, np.array(chosen, dtype=pl.dtype)))[0]
    starts = r.ranges[:, 0].astype(np.int64)
    nonempty = np.nonzero(r.ranges[:, 1] > r.ranges[:, 0])[0]
    tiles = sorted({int(nonempty[np.searchsorted(starts[nonempty], p, side="right") - 1]) for p in pos})
    if len(tiles) > max_tiles:       # bound the fp64 walk: keep the chosen surfels whose tiles fit
        keep, kept_tiles = [], set()
        for s_ in chosen:
            ts = {int(nonempty[np.searchsorted(starts[nonempty], p, side="right") - 1]) for p in pos if pl[p] == s_}
            if len(kept_tiles | ts) <= max_tiles:
                keep.append(s_)
                kept_tiles |= ts
        chosen, tiles = keep, sorted(kept_tiles)
    gx = (res + 15) // 16
    mask = np.zeros((res, res), dtype=bool)
    for t_ in tiles:
        mask[(t_ // gx) * 16:(t_ // gx) * 16 + 16, (t_ % gx) * 16:(t_ % gx) * 16 + 16] = True
    dc_m, da_m = dc * mask, da * mask
    entries = int(sum(int(r.ranges[t_, 1]) - int(r.ranges[t_, 0]) for t_ in tiles))
    if verbose:
        print(f"{len(chosen)} surfels, {len(tiles)} tiles, {entries} list entries to walk in fp64", flush=True)
    g_or_m = {k: v.astype(np.float64) for k, v in oracle.backward(r, dc_m, da_m).items() if k in KEYS}
    g_hip_m = hip(dc_m, da_m)
    inp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in act.items()}
    t0 = time.perf_counter()
    c64, a64 = autograd_ref.render(view, inp["means3D"], inp["opacities"], inp["shs"], inp["scales"], inp["rotations"], r.ranges,
                                   r.point_list, tiles=tiles, lowpass_depth_quirk=True)
    ((c64 * torch.from_numpy(dc_m).double()).sum() + (a64 * torch.from_numpy(da_m).double()).sum()).backward()
    t64 = time.perf_counter() - t0
    out = {"what": "masked backward over the tiles of the surfels on which HIP and the fp32 oracle disagree most (full frame), three ways; "
                   "errors relative to max|fp64| of the tensor over the surfels those tiles touch",
           "regime": regime, "surfels_chosen": len(chosen), "tiles": len(tiles), "list_entries_walked_in_fp64": entries,
           "fp64_seconds": round(t64, 1), "forward_max_abs_diff_fp32_oracle_vs_fp64_on_tiles": float(
               max(np.abs(c64.detach().numpy()[:, mask] - r.color[:, mask]).max(), np.abs(a64.detach().numpy()[:, mask] - r.allmap[:, mask]).max())),
           "per_tensor": {}}
    # the HIP forward against the fp64 walk on the same tiles: a (pixel, entry) pair decided the other way at the alpha = 1/255
    # cut shows up here as a step of about alpha T = 2e-3 in one pixel, against 1e-5 of rounding everywhere else
    dcol = np.abs(c64.detach().numpy() - g_hip_m["_color"])[:, mask]
    out["forward_max_abs_diff_hip_vs_fp64_on_tiles"] = float(max(dcol.max(), np.abs(a64.detach().numpy()[1] - g_hip_m["_allmap"][1])[mask].max()))
    out["forward_pixels_beyond_2e-4_hip_vs_fp64"] = int((dcol.max(0) > 2e-4).sum())
    out["forward_pixels_beyond_2e-4_fp32_oracle_vs_fp64"] = int((np.abs(c64.detach().numpy() - r.color)[:, mask].max(0) > 2e-4).sum())
    for k in KEYS:
        g64 = inp[k].grad.numpy().reshape(g_or_m[k].shape)
        P = g64.shape[0]
        mx = np.abs(g64).max() + 1e-300
        l2 = np.sqrt((g64 ** 2).sum()) + 1e-300
        row = {}
        for name, g in (("hip", g_hip_m[k].reshape(g64.shape)), ("fp32_oracle", g_or_m[k])):
            d = g - g64
            per = np.abs(d).reshape(P, -1).max(1) / mx
            row[name] = {"max_err_rel_to_max": float(f"{per.max():.3e}"), "rel_l2": float(f"{np.sqrt((d ** 2).sum()) / l2:.3e}"),
                         "max_err_on_chosen_surfels": float(f"{per[chosen].max():.3e}")}
        row["closer_to_fp64"] = "hip" if row["hip"]["max_err_rel_to_max"] <= row["fp32_oracle"]["max_err_rel_to_max"] else "fp32_oracle"
        row["full_frame_hip_vs_fp32_oracle_max"] = float(f"{(np.abs(g_hip[k].reshape(g_or[k].shape) - g_or[k]).max() / (np.abs(g_or[k]).max() + 1e-300)):.3e}")
        out["per_tensor"][k] = row
    out["seconds"] = round(time.perf_counter() - t_start, 1)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--worst", type=int, default=3)
    ap.add_argument("--regime", default="init")
    ap.add_argument("--max-tiles", type=int, default=40)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = arbitrate(a.worst, a.regime, max_tiles=a.max_tiles, verbose=True)
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
