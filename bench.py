#!/usr/bin/env python
"""bench.py -- novel-view frames/sec @512x512 (4-view in, 2DGS fwd+bwd) on MI355X.

One *step* = one pass of the raster hot path over one training batch as LaRa's step issues it
(lightning/network.py:473-497): B = 4 scenes x 8 target views (4 input + 4 novel,
dataLoader/gobjverse.py:46-47), each view = one GaussianRasterizer forward over the scene's
P = 524 288 surfels at 512x512 (configs/base.yaml:13,23,34) + its backward.  A *frame* is one such
forward + backward.  `value` = frames of all ranks / max-over-ranks wall time of the timed steps,
with scenes, cameras and incoming gradients already resident in HBM.  Data is synthetic (no
dataset / checkpoint in this environment): SURVEY.md section 8d, lara_amd/synthetic.py.

Multi-GPU: per-scene data parallel, one process per GPU (torch.distributed, backend nccl = RCCL);
the raster itself is per view and is NOT sharded (BASELINE.json north_star) -> weak scaling, every
rank renders its own B scenes.  The only exchange step of LaRa's training step is DDP's gradient
all-reduce of the encoder parameters (train_lightning.py:72); the encoder is outside this path, so
ranks all-reduce a stand-in fp32 buffer of the encoder's size (126.3 M parameters, SURVEY.md
section 2 #12) in 25 MB buckets on a side stream, overlapped with the raster backward.

The scenes of a step are independent, so they are spread over `--streams` HIP streams (default 2: the
composite kernels end in a tail of a few heavy tiles and the binning has single-workgroup steps; a
second stream fills those holes).  "single_stream" repeats the measurement with every call on the
current stream, i.e. exactly as the reference's Python loop would issue it.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed, algorithmic bytes
from DESIGN.md / SURVEY.md section 8d), "cpu_baseline" (the CPU oracle on a bounded sample),
"single_stream", "attention", "encoder" and "encoder_train" (MFMA legs, reported beside the raster).
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
ENCODER_PARAMS = 126_300_000  # VolTransformer 39.45 M + Decoder + ViT-B/16 ~86 M


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scenes", type=int, default=4, help="scenes per rank per step (config 3: 4)")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--grid", type=int, default=64, help="surfels = grid^3 * 2 (64 -> 524288)")
    ap.add_argument("--regime", default="init", choices=["init", "trained"])
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the independent scenes of a step are spread over (1 = the reference's sequential loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample-res", type=int, default=512)
    return ap.parse_args()


def build_batch(args, device, rank):
    from lara_amd import cameras, synthetic, GaussianRasterizationSettings
    from lara_amd import dp
    scenes = []
    for seed in dp.scene_seeds(rank, args.scenes):
        sc = synthetic.make_scene(grid=args.grid, K=2, regime=args.regime, seed=seed, device=device)
        scenes.append({k: v.requires_grad_(True) for k, v in sc.items()})
    cams = cameras.make_cameras(cameras.turntable_c2w(args.views), args.res, args.res, 0.75, 0.75,
                                1.906 - 0.8, 1.906 + 0.8, device=device)
    # background: 1 for the input views, {0, 0.5, 1} for the novel ones (gobjverse.py:103-106)
    bgs = [1.0] * (args.views // 2) + [(0.0, 0.5, 1.0)[j % 3] for j in range(args.views - args.views // 2)]
    settings = []
    for cam, b in zip(cams, bgs):
        settings.append(GaussianRasterizationSettings(
            image_height=args.res, image_width=args.res, tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.full((3,), b, device=device), scale_modifier=1.0,
            viewmatrix=cam.world_view_transform.contiguous(), projmatrix=cam.full_proj_transform.contiguous(),
            sh_degree=1, campos=cam.camera_center.contiguous(), prefiltered=False, debug=False))
    g = torch.Generator(device="cpu").manual_seed(7 + rank)
    gc = (torch.randn(3, args.res, args.res, generator=g) / (args.res * args.res)).to(device)
    ga = (torch.randn(7, args.res, args.res, generator=g) / (args.res * args.res) * 0.1).to(device)
    return scenes, settings, gc, ga


_streams = []


def step(scenes, settings, gc, ga, n_streams=1):
    """All forwards of the batch, then one backward through every view (as loss.backward() does).
    Scenes are independent (the unit the north star data-parallelises over), so scene i is enqueued on
    HIP stream i % n_streams; autograd replays each view's backward on its forward's stream.  The
    composite kernels end with a tail of a few heavy tiles and the binning has single-workgroup
    steps: a second stream fills those holes with the next scene's work."""
    from lara_amd import GaussianRasterizer
    outs, grads = [], []
    cur = torch.cuda.current_stream()
    while len(_streams) < n_streams and n_streams > 1:
        _streams.append(torch.cuda.Stream())
    for i, sc in enumerate(scenes):
        side = _streams[i % n_streams] if n_streams > 1 else None
        if side is not None:
            side.wait_stream(cur)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
          for rs in settings:
            # the reference's activations, applied per view (renderer_2dgs.py:181-189)
            opac = torch.sigmoid(sc["opacity"])
            scales = torch.exp(sc["scales"])
            rots = torch.nn.functional.normalize(sc["rotations"])
            means2D = torch.zeros_like(sc["centers"], requires_grad=True)
            color, radii, allmap = GaussianRasterizer(rs)(
                means3D=sc["centers"], means2D=means2D, shs=sc["shs"], opacities=opac,
                scales=scales, rotations=rots, cov3D_precomp=None)
            outs += [color, allmap]
            grads += [gc, ga]
    for side in _streams[:n_streams if n_streams > 1 else 0]:
        cur.wait_stream(side)
    torch.autograd.backward(outs, grads)
    for side in _streams[:n_streams if n_streams > 1 else 0]:
        cur.wait_stream(side)
    for sc in scenes:
        for v in sc.values():
            v.grad = None


def forward_only_leg(scenes, settings, args):
    """BASELINE.json configs[1] (inference): forward renders only, same scenes / views / streams."""
    from lara_amd import GaussianRasterizer
    cur = torch.cuda.current_stream()

    def fwd():
        with torch.no_grad():
            for i, sc in enumerate(scenes):
                side = _streams[i % args.streams] if args.streams > 1 and _streams else None
                if side is not None:
                    side.wait_stream(cur)
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    opac, scales = torch.sigmoid(sc["opacity"]), torch.exp(sc["scales"])
                    rots = torch.nn.functional.normalize(sc["rotations"])
                    for rs in settings:
                        GaussianRasterizer(rs)(means3D=sc["centers"], means2D=None, shs=sc["shs"], opacities=opac,
                                               scales=scales, rotations=rots, cov3D_precomp=None)
            for side in _streams[:args.streams if args.streams > 1 else 0]:
                cur.wait_stream(side)

    fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fwd()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = len(scenes) * len(settings) * args.steps
    return {"value": round(frames / dt, 1), "unit": "frames/s",
            "workload": "forward renders only (configs[1]: inference), same scenes and views"}


def measure_roofline(scenes, settings, gc, ga, args):
    """Per-kernel HIP-event times over one step; returns (roofline dict, per-kernel table, D)."""
    from lara_amd import rasterizer
    rasterizer.profile_enable(True)
    step(scenes, settings, gc, ga)
    torch.cuda.synchronize()
    rec = rasterizer.profile_collect()
    rasterizer.profile_enable(False)
    agg = {}
    for name, ms in rec:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += ms
    # D (pairs per frame) of one representative view
    from lara_amd import synthetic
    act = synthetic.activate({k: v.detach() for k, v in scenes[0].items()})
    r = rasterizer.forward_with_state(settings[0], act["means3D"], act["opacities"], shs=act["shs"],
                                      scales=act["scales"], rotations=act["rotations"])
    torch.cuda.synchronize()
    D = int(r["views"]["header"][0].item())
    P = scenes[0]["centers"].shape[0]
    HW = args.res * args.res
    # ALGORITHMIC bytes per launch (DESIGN.md section "kernels and roofs")
    alg = {
        "preprocess_fwd": 88 * P + 92 * P,
        "tile_scan": 12 * (HW // 256),
        "scatter": 12 * P + 8 * D,
        "tile_sort_small": 12 * D,
        "tile_sort_large": 0,
        "composite_fwd": 84 * D + 60 * HW,
        "composite_bwd": 84 * D + 100 * HW + 144 * D,
        "preprocess_bwd": (88 + 80 + 80) * P + 88 * P,
    }
    table = {k: {"launches": n, "avg_us": 1e3 * t / n, "alg_bytes": alg.get(k, 0),
                 "alg_GBs": (alg.get(k, 0) / (1e-3 * t / n) / 1e9) if t > 0 else 0.0}
             for k, (n, t) in agg.items()}
    dom = max(table, key=lambda k: table[k]["avg_us"] * table[k]["launches"])
    t = table[dom]
    # HBM bytes per launch from the PMC counters (FETCH_SIZE + WRITE_SIZE): they need rocprofv3 passes of
    # their own, so they are collected by tools/gpu_traffic.sh and committed under profiles/
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01i.json")))
        if args.regime == "init" and args.grid == 64 and args.res == 512:
            traffic = tj["bytes_per_launch"][dom]["total"]
    except Exception:
        traffic = None
    # what a plain device-to-device copy reaches on this box (read + write bytes / time): the achievable
    # HBM rate to hold beside the vendor peak (SURVEY.md section 8d asks for both)
    buf = torch.empty(2, 1 << 28, dtype=torch.uint8, device=scenes[0]["centers"].device)
    buf[1].copy_(buf[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        buf[1].copy_(buf[0])
    e1.record()
    torch.cuda.synchronize()
    copy_GBs = 5 * 2 * (1 << 28) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del buf
    # the whole frame against the byte model of SURVEY.md section 8d (175 P + 120 D + 60 HW forward,
    # 248 P + 220 D + 100 HW backward)
    frame_bytes = (175 + 248) * P + (120 + 220) * D + 160 * HW
    frame_us = sum(v["avg_us"] for v in table.values())
    roof = {"kernel": dom, "bound": "hbm", "achieved": round(t["alg_GBs"], 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(t["alg_GBs"] / HBM_PEAK_GBS, 5), "traffic": traffic,
            "avg_launch_us": round(t["avg_us"], 2), "alg_bytes_per_launch": t["alg_bytes"],
            "pairs_per_frame_D": D, "measured_copy_GBs": round(copy_GBs, 1),
            "whole_frame": {"alg_bytes": frame_bytes, "kernel_us": round(frame_us, 1),
                            "achieved": round(frame_bytes / frame_us / 1e3, 1),
                            "frac": round(frame_bytes / frame_us / 1e3 / HBM_PEAK_GBS, 5)}}
    return roof, table, D


def attention_leg(device, scenes):
    """Group cross-attention step of one transformer layer for this rank's scenes (forward, bf16
    MFMA): time and FLOP rate against the 2.5 PFLOP/s dense bf16 peak.  Reported beside the raster;
    not part of `value`."""
    from torch import nn
    from lara_amd import rasterizer
    from lara_amd.attention import GroupCrossAttention
    torch.manual_seed(0)
    mod = GroupCrossAttention.from_modules(
        nn.LayerNorm(256), nn.MultiheadAttention(256, 16, kdim=800, vdim=800, bias=False, batch_first=True)).to(device)
    G = 4096 * scenes
    x = torch.randn(G, 8, 256, device=device)
    cond = torch.randn(G, 4, 800, device=device)
    with torch.no_grad():
        for _ in range(2):
            mod(x, cond)
        torch.cuda.synchronize()
        rasterizer.profile_enable(True)
        for _ in range(5):
            mod(x, cond)
        torch.cuda.synchronize()
    rec = rasterizer.profile_collect()
    rasterizer.profile_enable(False)
    us = 1e3 * sum(ms for _, ms in rec) / 5
    flops = G * (2 * 8 * 256 * 256 * 2 + 2 * 4 * 800 * 512 + 2 * 2 * 8 * 4 * 16 * 16)
    return {"workload": f"GroupAttBlock attention step (LN, q/k/v/out projections, QK^T, softmax, AV), "
                        f"{scenes} scenes = {G} groups, forward, bf16 MFMA / fp32 accumulate",
            "us_per_layer": round(us, 1), "achieved": round(flops / us / 1e6, 1), "peak": 2500.0,
            "unit": "TFLOP/s", "frac": round(flops / us / 1e6 / 2500.0, 4), "bound": "mfma"}


def encoder_leg(device, scenes):
    """The whole volume transformer (12 GroupAttBlocks on a 32^3 x 256 volume + the x2 deconvolution,
    network.py:105-164) for this rank's scenes, forward, random-init weights of the reference's shapes:
    time and FLOP rate against the 2.5 PFLOP/s dense bf16 peak.  Reported beside the raster."""
    from lara_amd import rasterizer
    from lara_amd.encoder import VolTransformer
    torch.manual_seed(0)
    vt = VolTransformer(256, 800, [16], 32, 64, 80, 12, 16).to(device)
    with torch.no_grad():
        for n, p in list(vt.named_parameters()) + list(vt.named_buffers()):
            if p.dtype == torch.bfloat16:
                p.copy_((torch.randn(p.shape, device=device) * (p.shape[-1] ** -0.5)).to(torch.bfloat16))
            elif n.endswith("_w"):
                p.fill_(1.0)
        vt.pos_embed.normal_(0, 1 / 16)
        feats = torch.randn(scenes, 4, 800, 16, 16, 16, device=device)
        vt(feats)
        torch.cuda.synchronize()
        rasterizer.profile_enable(True)  # per-kernel HIP events need plain launches
        for _ in range(2):
            vt(feats)
        torch.cuda.synchronize()
        rec = rasterizer.profile_collect()
        rasterizer.profile_enable(False)
        vt(feats, use_graph=True)  # captures the HIP graph
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            vt(feats, use_graph=True)
        e1.record()
        torch.cuda.synchronize()
        wall_graph = e0.elapsed_time(e1) / 3
    M, G = scenes * 32 ** 3, scenes * 4096
    ms = sum(t for _, t in rec) / 2
    conv_us = 1e3 * sum(t for k, t in rec if k == "gb_conv3d") / 24
    per_layer = 2 * M * 256 * 256 * 2 + 2 * G * 4 * 800 * 512 + 2 * 2 * 8 * 4 * 16 * 16 * G + 2 * M * 256 * 512 * 2 + 2 * M * 27 * 256 * 256
    flops = 12 * per_layer + 2 * M * 256 * 640
    return {"workload": f"VolTransformer forward (12 x [attention, MLP, LayerNorms, Conv3d 3x3x3] + deconv), {scenes} scenes "
                        f"x 32^3 voxels, bf16 MFMA / fp32 accumulate, random-init weights",
            "ms_per_forward": round(ms, 2), "ms_wall_hip_graph": round(wall_graph, 2),
            "achieved": round(flops / ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(flops / ms / 1e9 / 2500.0, 4), "bound": "mfma",
            "conv3d_us": round(conv_us, 1), "conv3d_TFLOPs": round(2 * M * 27 * 256 * 256 / conv_us / 1e6, 1)}


def encoder_train_leg(device, scenes):
    """The trainable drop-in (lara_amd.encoder_train.VolTransformer: fp32 master parameters with the
    reference's state_dict, HIP forward AND backward) on this rank's scenes: one forward + backward per step,
    gradients for every parameter and for the image features.  Reported beside the raster."""
    from lara_amd.encoder_train import VolTransformer
    torch.manual_seed(0)
    vt = VolTransformer(256, 800, [16], 32, 64, 80, 12, 16).to(device)  # the reference's own initialisation
    feats = torch.randn(scenes, 4, 800, 16, 16, 16, device=device, requires_grad=True)
    dout = torch.randn(scenes, 64, 64, 64, 80, device=device)

    def one():
        vt(feats).backward(dout)
        vt.zero_grad(set_to_none=True)
        feats.grad = None

    torch.cuda.reset_peak_memory_stats(device)
    one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        one()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    M, G = scenes * 32 ** 3, scenes * 4096
    per_layer = 2 * M * 256 * 256 * 2 + 2 * G * 4 * 800 * 512 + 2 * 2 * 8 * 4 * 16 * 16 * G + 2 * M * 256 * 512 * 2 + 2 * M * 27 * 256 * 256
    flops = 3 * (12 * per_layer + 2 * M * 256 * 640)   # backward = 2 x forward (dX and dW for every product)
    return {"workload": f"VolTransformer forward + backward, {scenes} scenes x 32^3 voxels, 12 layers, bf16 MFMA / fp32 "
                        f"accumulate, fp32 master parameters, all parameter and image-feature gradients",
            "ms_per_step": round(ms, 2), "achieved": round(flops / ms / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
            "frac": round(flops / ms / 1e9 / 2500.0, 4), "bound": "mfma",
            "peak_memory_GiB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1)}


def render_img_leg(device, args):
    """One scene's 8 views through the whole of `Renderer.render_img` (renderer_2dgs.py:167-268), forward +
    backward with a gradient on every returned map: (a) the reference's sequence on the drop-in rasteriser --
    activations per view and ~15 torch kernels of post-processing per view (restated in torch here, as the
    reference's class does it); (b) `lara_amd.renderer.Renderer` -- activations once per scene, one fused HIP
    kernel per direction.  Single stream, as the reference's Python loop issues it."""
    import torch.nn.functional as F
    from lara_amd import batch, cameras, synthetic
    from lara_amd.renderer import Renderer
    sc = synthetic.make_scene(grid=args.grid, K=2, regime=args.regime, seed=123, device=device)
    c2w = cameras.turntable_c2w(args.views).to(device)
    cams = cameras.make_cameras(c2w, args.res, args.res, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8, device=device)
    ixt = batch.fov_to_ixt(torch.tensor([0.75, 0.75], device=device), (args.res, args.res))
    rays = batch.build_rays(c2w, ixt.reshape(1, 3, 3).expand(args.views, 3, 3).contiguous(), args.res, args.res)
    keys = ("image", "depth", "acc_map", "rend_normal", "depth_normal", "rend_dist")
    r = Renderer(sh_degree=1, white_background=True)

    def ref_style(cam, ray, p):
        rast = r.set_rasterizer(cam, device=device)
        sp = torch.zeros_like(p["centers"], requires_grad=True) + 0
        col, _, allmap = rast(means3D=p["centers"], means2D=sp, shs=p["shs"], opacities=torch.sigmoid(p["opacity"]),
                              scales=torch.exp(p["scales"]), rotations=F.normalize(p["rotations"]), cov3D_precomp=None)
        image = col.clamp(0, 1)
        alpha = allmap[1:2]
        normal = (allmap[2:5].permute(1, 2, 0) @ (cam.world_view_transform[:3, :3].T)).permute(2, 0, 1)
        median = torch.nan_to_num(allmap[5:6], 0, 0)
        expected = torch.nan_to_num(allmap[0:1] / alpha, 0, 0)
        surf = expected * (1 - 0.0) + 0.0 * median
        pts = (ray[..., :3].reshape(-1, 3) + surf.reshape(-1, 1) * ray[..., 3:].reshape(-1, 3)).reshape(*surf.shape[1:], 3)
        out = torch.zeros_like(pts)
        dx = pts[2:, 1:-1] - pts[:-2, 1:-1]
        dy = pts[1:-1, 2:] - pts[1:-1, :-2]
        out[1:-1, 1:-1, :] = F.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
        dn = out.permute(2, 0, 1) * alpha.detach()
        return {"image": image.permute(1, 2, 0), "depth": surf.permute(1, 2, 0), "acc_map": alpha.squeeze(0),
                "rend_normal": normal.permute(1, 2, 0), "depth_normal": dn.permute(1, 2, 0), "rend_dist": allmap[6]}

    def one(mode):
        p = {k: v.detach().requires_grad_(True) for k, v in sc.items()}
        loss = 0
        for cam, ray in zip(cams, rays):
            o = (r.render_img(cam, ray, p["centers"], p["shs"], p["opacity"], p["scales"], p["rotations"], device)
                 if mode == "fused" else ref_style(cam, ray, p))
            for k in keys:
                loss = loss + o[k].sum() * 1e-6
        loss.backward()

    res = {}
    for mode in ("reference_style", "fused"):
        one(mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            one(mode)
        torch.cuda.synchronize()
        res[mode] = round(2 * args.views / (time.perf_counter() - t0), 1)
    return {"workload": f"Renderer.render_img fwd+bwd, 1 scene x {args.views} views @{args.res}x{args.res}, gradients on all six "
                        f"returned maps, one stream", "unit": "frames/s", "reference_style_torch_postprocessing": res["reference_style"],
            "fused_renderer": res["fused"]}


def point_feats_leg(device, args):
    """The fine stage's point sampler (network.py:390-411): 262 144 points (half of a scene's Gaussians pass the
    opacity mask) into the 4 input views @res, forward + backward: the reference's torch sequence (projection,
    cat + permute, grid_sample, and their autograd) against the fused HIP kernels."""
    import torch.nn.functional as F
    from lara_amd import cameras
    from lara_amd.fine import sample_point_feats
    g = torch.Generator().manual_seed(3)
    V, h, w, n = 4, args.res, args.res, 262144
    w2c = torch.linalg.inv(cameras.turntable_c2w(V).double()).float().to(device)
    focal = 0.5 * w / math.tan(0.5 * 0.75)
    ixt = torch.tensor([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1.0]], dtype=torch.float32).expand(V, 3, 3).contiguous().to(device)
    pts0 = ((torch.rand(n, 3, generator=g) * 2 - 1) * 0.5).to(device)
    img_ref = torch.rand(V, 3, h, w, generator=g).to(device)
    maps0 = [torch.rand(V, h, w, 3, generator=g).to(device), torch.rand(V, h, w, generator=g).to(device),
             (1.5 + torch.rand(V, h, w, 1, generator=g)).to(device)]
    gout = torch.randn(V, 8, n, generator=g).to(device)

    def torch_path(points, image, acc, depth):
        pc = points.reshape(1, -1, 3) @ w2c[:, :3, :3].permute(0, 2, 1) + w2c[:, :3, 3][:, None]
        q = pc @ ixt.permute(0, 2, 1)
        xy, z = q[..., :2] / q[..., -1:], q[..., -1:]
        grid = (xy + 0.5) / torch.tensor([w, h], device=device) * 2 - 1.0
        stack = torch.cat((img_ref, torch.einsum('bhwc->bchw', torch.cat((image, acc.unsqueeze(-1), depth), dim=-1))), dim=1)
        feats = F.grid_sample(stack, grid.unsqueeze(1), align_corners=False).view(V, -1, n)
        return torch.cat((feats[:, :-1], (feats[:, -1:] - z.view(V, -1, n)).abs()), dim=1)

    def one(fused):
        p = pts0.clone().requires_grad_(True)
        m = [t.clone().requires_grad_(True) for t in maps0]
        out = sample_point_feats(p, w2c, ixt, img_ref, *m) if fused else torch_path(p, *m)
        out.backward(gout)

    res = {}
    for fused in (False, True):
        one(fused)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            one(fused)
        e1.record()
        torch.cuda.synchronize()
        res[fused] = e0.elapsed_time(e1) / 5 * 1e3
    return {"workload": f"get_point_feats fwd+bwd, {n} points x {V} views @{h}x{w} (incl. cloning the inputs)", "unit": "us",
            "torch_sequence_us": round(res[False], 1), "fused_us": round(res[True], 1)}


def rays_leg(device, scenes, views, res):
    """Device-side generation of the step's tar_rays + tar_rays_down (dataLoader/utils.py:21-34):
    a pure store stream, priced against the HBM peak."""
    from lara_amd import cameras
    from lara_amd.batch import build_rays, fov_to_ixt
    c2w = cameras.turntable_c2w(views).float().to(device)
    ixt = fov_to_ixt(torch.full((views, 2), 0.75), (res, res)).to(device)
    from lara_amd import rasterizer
    for _ in range(2):
        build_rays(c2w, ixt, res, res)
    torch.cuda.synchronize()
    rasterizer.profile_enable(True)
    for _ in range(scenes * 5):
        build_rays(c2w, ixt, res, res)
        build_rays(c2w, ixt, res, res, 1.0 / 16)
    torch.cuda.synchronize()
    rec = rasterizer.profile_collect()
    rasterizer.profile_enable(False)
    us = 1e3 * sum(t for k, t in rec if k == "build_rays") / 5  # kernel time (HIP events on the launch stream)
    nbytes = scenes * views * 24 * (res * res + (res // 16) ** 2)
    return {"workload": f"tar_rays + tar_rays_down of {scenes} scenes x {views} views @{res}x{res}, written on the device",
            "us_per_step": round(us, 1), "achieved": round(nbytes / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4), "bound": "hbm"}


def cpu_baseline(args):
    """The CPU oracle (fp32 restatement, OpenMP over tiles) on a bounded sample of the same
    workload: the 8 views of scene 0, forward + backward each (about 10-30 s of CPU work)."""
    import numpy as np
    import oracle
    from lara_amd import cameras, synthetic
    sc = synthetic.make_scene(grid=args.grid, K=2, regime=args.regime, seed=0)
    act = {k: v.numpy() for k, v in synthetic.activate(sc).items()}
    res = args.cpu_sample_res
    cams = cameras.make_cameras(cameras.turntable_c2w(args.views), res, res, 0.75, 0.75,
                                1.906 - 0.8, 1.906 + 0.8)
    oracle.build()
    g = np.random.default_rng(0)
    dc = g.normal(size=(3, res, res)).astype(np.float32)
    da = (0.1 * g.normal(size=(7, res, res))).astype(np.float32)   # (the gradient mix of tests/test_raster_parity_gpu.py)
    t0 = time.perf_counter()
    D = 0
    first = None
    for cam in cams:
        view = oracle.View(res, res, math.tan(0.375), math.tan(0.375), np.ones(3, np.float32), 1.0,
                           cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), 1,
                           cam.camera_center.numpy())
        r = oracle.forward(view, act["means3D"], act["opacities"], shs=act["shs"], scales=act["scales"],
                           rotations=act["rotations"])
        gr = oracle.backward(r, dc, da)
        D = r.num_rendered
        if first is None:
            first = (r, gr)
    dt = time.perf_counter() - t0
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    out = {"value": round(len(cams) / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{len(cams)} frames (fwd+bwd, the {len(cams)} views of scene 0, {res}x{res}, "
                     f"P={act['means3D'].shape[0]}, D~{D}) with the OpenMP fp32 oracle; {dt:.2f} s"}
    # the checker's other job: the HIP frame of view 0 against the oracle's, at full size (BASELINE's "PSNR vs ref";
    # the oracle stands in for the absent reference rasteriser)
    from lara_amd import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda", torch.cuda.current_device())
    cam = cams[0]
    rs = GaussianRasterizationSettings(
        image_height=res, image_width=res, tanfovx=math.tan(0.375), tanfovy=math.tan(0.375), bg=torch.ones(3, device=dev),
        scale_modifier=1.0, viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev),
        sh_degree=1, campos=cam.camera_center.to(dev), prefiltered=False, debug=False)
    t = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in act.items()}
    m2d = torch.zeros_like(t["means3D"], requires_grad=True)
    color, radii, allmap = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                                                  scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    ((color * torch.from_numpy(dc).to(dev)).sum() + (allmap * torch.from_numpy(da).to(dev)).sum()).backward()
    r, gr = first
    mse = float(((color.detach().cpu().numpy() - r.color) ** 2).mean())
    gerr, l2err = {}, {}
    for k in ("means3D", "opacities", "scales", "rotations", "shs"):
        d = t[k].grad.cpu().numpy().reshape(gr[k].shape).astype(np.float64) - gr[k]
        gerr[k] = float(np.abs(d).max() / (np.abs(gr[k]).max() + 1e-20))
        l2err[k] = float(np.sqrt((d ** 2).sum() / ((gr[k].astype(np.float64) ** 2).sum() + 1e-300)))
    # yardstick: the fp32 oracle's own gradient after every input moved by one ulp (tools/grad_probe.py)
    rng = np.random.default_rng(1)
    pert = {k: (v * (1 + (rng.integers(0, 2, v.shape) * 2 - 1) * 2.0 ** -23)).astype(np.float32) for k, v in act.items()}
    view0 = oracle.View(res, res, math.tan(0.375), math.tan(0.375), np.ones(3, np.float32), 1.0, cam.world_view_transform.numpy(),
                        cam.full_proj_transform.numpy(), 1, cam.camera_center.numpy())
    rp = oracle.forward(view0, pert["means3D"], pert["opacities"], shs=pert["shs"], scales=pert["scales"], rotations=pert["rotations"])
    gp = oracle.backward(rp, dc, da)
    self_l2 = {k: float(np.sqrt(((gp[k].astype(np.float64) - gr[k]) ** 2).sum() / ((gr[k].astype(np.float64) ** 2).sum() + 1e-300)))
               for k in l2err}
    out["parity_vs_oracle"] = {"view": 0, "psnr_color_dB": round(10 * math.log10(1.0 / max(mse, 1e-30)), 1),
                               "oracle_1ulp_self": {"psnr_color_dB": round(10 * math.log10(1.0 / max(float(((rp.color - r.color) ** 2).mean()), 1e-30)), 1),
                                                    "grad_rel_l2": {k: float(f"{v:.2e}") for k, v in self_l2.items()}},
                               "radii_identical": bool(np.array_equal(radii.cpu().numpy(), r.radii)),
                               "max_grad_err_rel_to_max": {k: float(f"{v:.2e}") for k, v in gerr.items()},
                               "grad_err_rel_l2": {k: float(f"{v:.2e}") for k, v in l2err.items()}}
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    local = local % torch.cuda.device_count()  # (several ranks on one GPU only in the gloo self-test)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LARA_BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI; gloo for plumbing tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from lara_amd import rasterizer
    rasterizer.load_library()
    scenes, settings, gc, ga = build_batch(args, device, rank)

    from lara_amd import dp
    comm_stream = torch.cuda.Stream(device) if world > 1 else None
    grad_buf = torch.zeros(ENCODER_PARAMS, device=device) if world > 1 else None

    def full_step():
        if world > 1:  # DDP-style bucketed all-reduce, overlapped with the raster work on a side stream
            comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm_stream):
                dp.bucketed_all_reduce(grad_buf)
        step(scenes, settings, gc, ga, args.streams)
        if world > 1:
            torch.cuda.current_stream().wait_stream(comm_stream)

    for _ in range(args.warmup):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rasterizer.check_pending(block=True)
    if world > 1:
        dt = dp.max_over_ranks(dt, device)

    frames_per_step = args.scenes * args.views * world
    out = {
        "metric": "novel-view frames/sec @512x512 (4-view in, 2DGS fwd+bwd)",
        "value": round(frames_per_step * args.steps / dt, 3),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"configs[2]: training-step raster fwd+bwd, {args.scenes} scenes/GPU x {args.views} "
                        f"views @{args.res}x{args.res}, P={scenes[0]['centers'].shape[0]} surfels/scene, "
                        f"SH degree 1, regime={args.regime}",
            "frames_per_step": frames_per_step,
            "parallelism": f"dp{world} (per-scene; raster not sharded)",
            "hip_streams": args.streams,
            "grad_allreduce": (f"{ENCODER_PARAMS * 4 / 1e6:.0f} MB fp32 stand-in for the encoder's DDP "
                               "gradient, 25 MB buckets, RCCL, overlapped") if world > 1 else None,
        },
    }
    if rank == 0 and world == 1 and args.streams != 1 and not args.no_roofline:
        # the same step as the reference's loop would issue it: every scene on the current stream
        for _ in range(2):
            step(scenes, settings, gc, ga, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(scenes, settings, gc, ga, 1)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        out["single_stream"] = {"value": round(frames_per_step * args.steps / dt1, 3), "unit": "frames/s",
                                "ms_per_step": round(1e3 * dt1 / args.steps, 3)}
    if rank == 0 and world == 1 and not args.no_roofline:
        # opt-in, not in the reference: surfels with opacity < 1/255 (never drawn) culled in the preprocess; same
        # images to an ulp, same gradients (tests/test_raster_parity_gpu.py); nothing to cull at LaRa's initialisation,
        # most of the volume in a trained-like scene
        from lara_amd import rasterizer as _rz
        prev = _rz.set_cull_transparent(True)
        try:
            for _ in range(2):
                step(scenes, settings, gc, ga, args.streams)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step(scenes, settings, gc, ga, args.streams)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
        finally:
            _rz.set_cull_transparent(prev)
        out["cull_transparent_opt_in"] = {"value": round(frames_per_step * args.steps / dt1, 3), "unit": "frames/s",
                                          "ms_per_step": round(1e3 * dt1 / args.steps, 3)}
        out["forward_only"] = forward_only_leg(scenes, settings, args)
    if rank == 0 and not args.no_roofline:
        roof, table, D = measure_roofline(scenes, settings, gc, ga, args)
        out["roofline"] = roof
        out["kernels"] = {k: {"avg_us": round(v["avg_us"], 2), "launches": v["launches"],
                              "alg_GBs": round(v["alg_GBs"], 1)} for k, v in table.items()}
    if rank == 0 and world == 1 and not args.no_roofline:   # the side legs run at N = 1 only (the other ranks would wait)
        out["attention"] = attention_leg(device, args.scenes)
        out["encoder"] = encoder_leg(device, args.scenes)
        out["encoder_train"] = encoder_train_leg(device, args.scenes)
        out["rays"] = rays_leg(device, args.scenes, args.views, args.res)
        out["render_img"] = render_img_leg(device, args)
        out["point_feats"] = point_feats_leg(device, args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
