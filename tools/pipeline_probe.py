#!/usr/bin/env python
"""Host + device time of every operator call inside one pipeline step (synchronised around each call: no overlap,
what each call costs on its own).  usage: python tools/pipeline_probe.py [--scenes 4]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from lara_amd import fine, pipeline, rasterizer, renderer

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=4)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
args = argparse.Namespace(gpus=1, steps=3, warmup=1, scenes=a.scenes, views=8, res=512, grid=64, regime="init", step="pipeline",
                          no_fine=False, encoder_layers=12, raster_api="views", streams=1, no_cpu_baseline=True, no_roofline=True,
                          no_side_legs=True, fine_mask="reference")
dev = torch.device("cuda:0")
full_step, info = bench.make_pipeline_step(args, dev, 0, 1, False)
for _ in range(2):
    full_step()
torch.cuda.synchronize()
acc = {}


def wrap(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or name

    def timed(*x, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*x, **k)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        e = acc.setdefault(label, [0, 0.0, 0.0])
        e[0] += 1; e[1] += t1 - t0; e[2] += t2 - t0
        return r
    setattr(mod, name, timed)


wrap(pipeline, "take_rows")
wrap(pipeline, "sample_point_feats")
wrap(pipeline, "forward_fine")
wrap(pipeline, "decode_coarse")
wrap(pipeline, "check_mask")
wrap(renderer, "rasterize_gaussians_views")
wrap(renderer, "surface_maps_views")
pipe = info["pipeline"][0]
wrap(pipe.vol_decoder, "forward", "encoder forward")
for _ in range(a.steps):
    t0 = time.perf_counter()
    full_step()
    torch.cuda.synchronize()
    print(f"step (with per-call syncs): {1e3 * (time.perf_counter() - t0):.1f} ms")
print(f"{'call':32s} {'n/step':>7s} {'host ms/step':>13s} {'host+device ms/step':>20s}")
for k, (n, h, t) in sorted(acc.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:32s} {n / a.steps:7.1f} {1e3 * h / a.steps:13.2f} {1e3 * t / a.steps:20.2f}")
