"""Forward-only renders in isolation: the per-view loop and the multi-view call under `no_grad`, with the
caching allocator's counters before / after, at the capacity a fresh process starts from and at the capacity a training run
has grown the size class to.  Usage: python tools/fwdonly_probe.py [--steps 20] [--out gpurun_out/fwdonly_probe.json]"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--scenes", type=int, default=4)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--regime", default="init")
    ap.add_argument("--grown", type=int, default=7_500_000, help="pair count a training run was seen to reach (D per view)")
    ap.add_argument("--only", default="", help="per_view | views_api: run just that loop (counter passes)")
    ap.add_argument("--out", default="gpurun_out/fwdonly_probe.json")
    args = ap.parse_args()
    from lara_amd import cameras, synthetic, GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_views
    from lara_amd import rasterizer as rz
    dev = torch.device("cuda:0")
    scenes = [synthetic.make_scene(grid=args.grid, K=2, regime=args.regime, seed=s, device=dev) for s in range(args.scenes)]
    cams = cameras.make_cameras(cameras.turntable_c2w(args.views), args.res, args.res, 0.75, 0.75, 1.906 - 0.8, 1.906 + 0.8, device=dev)
    settings = [GaussianRasterizationSettings(
        image_height=args.res, image_width=args.res, tanfovx=math.tan(c.FoVx * 0.5), tanfovy=math.tan(c.FoVy * 0.5),
        bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=c.world_view_transform.contiguous(),
        projmatrix=c.full_proj_transform.contiguous(), sh_degree=1, campos=c.camera_center.contiguous(), prefiltered=False,
        debug=False) for c in cams]
    act = []
    with torch.no_grad():
        for sc in scenes:
            act.append((sc["centers"], sc["shs"], torch.sigmoid(sc["opacity"]), torch.exp(sc["scales"]),
                        torch.nn.functional.normalize(sc["rotations"])))

    def per_view():
        with torch.no_grad():
            for c, sh, o, s, r in act:
                for rs in settings:
                    GaussianRasterizer(rs)(means3D=c, means2D=None, shs=sh, opacities=o, scales=s, rotations=r)

    def views_api():
        with torch.no_grad():
            for c, sh, o, s, r in act:
                rasterize_gaussians_views(settings, c, None, o, shs=sh, scales=s, rotations=r)

    def stats():
        st = torch.cuda.memory_stats()
        return {"alloc_retries": st.get("num_alloc_retries", 0), "reserved_MB": round(st.get("reserved_bytes.all.current", 0) / 2**20),
                "reserved_peak_MB": round(st.get("reserved_bytes.all.peak", 0) / 2**20),
                "allocated_MB": round(st.get("allocated_bytes.all.current", 0) / 2**20),
                "allocated_peak_MB": round(st.get("allocated_bytes.all.peak", 0) / 2**20),
                "device_mallocs": st.get("segment.all.allocated", 0)}

    def rate(fn):
        fn()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        s0 = stats()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s1 = stats()
        return {"frames_per_s": round(args.scenes * args.views * args.steps / dt, 1), "host_enqueue_frac": round(t_host / dt, 3),
                "before": s0, "after": s1}

    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    out = {"config": vars(args)}

    def phase(name):
        out[name] = {}
        for what, fn in (("per_view", per_view), ("views_api", views_api)):
            if args.only and what != args.only:
                continue
            try:
                out[name][what] = rate(fn)
            except torch.OutOfMemoryError as e:      # (round 5: forwards whose pair counts were not read yet own their state)
                out[name][what] = {"out_of_memory": str(e)[:300], "at": stats()}
                torch.cuda.synchronize()
            with open(args.out, "w") as f:
                json.dump(out, f, indent=1)

    phase("fresh")
    if args.only:
        print(json.dumps(out))
        return
    # what `step_with_reference_lr` leaves behind: the size class has seen D = --grown pairs per view
    b = rz._bucket(dev, scenes[0]["centers"].shape[0], args.res, args.res)
    if hasattr(rz, "note_pair_count"):
        for _ in range(4):
            rz.note_pair_count(b, args.grown)
    else:
        rz._hwm[b] = args.grown
    out["grown_capacity"] = rz.binning_capacity(scenes[0]["centers"].shape[0], args.res, args.res, dev)
    phase("grown")
    # ... and many calls later (a history that decays gives the capacity back)
    try:
        for _ in range(6):
            views_api()
            per_view()
    except torch.OutOfMemoryError as e:
        out["later_warmup"] = {"out_of_memory": str(e)[:300], "at": stats()}
    torch.cuda.synchronize()
    out["later_capacity"] = rz.binning_capacity(scenes[0]["centers"].shape[0], args.res, args.res, dev)
    phase("later")
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
