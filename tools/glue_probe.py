#!/usr/bin/env python
"""Which line of host code launches the step's torch kernels?

Runs the headline pipeline step (bench.make_pipeline_step, one stream) under torch.profiler with Python stacks and
attributes every device kernel that is NOT one of the library's own (lara_amd/csrc) to
  * the innermost frame of its launching operator's stack that lies in lara_amd/ or bench.py (forward, and the
    backward of the custom autograd Functions, which run Python), or
  * the autograd node that launched it (backward of torch's own operators: no Python stack there).
Prints count and device microseconds per step for each (site, kernel) pair, largest first.
usage (GPU box): python tools/glue_probe.py [--steps 2] [--top 70]"""
import argparse
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--top", type=int, default=70)
a = ap.parse_args()
args = argparse.Namespace(gpus=1, steps=3, warmup=1, scenes=4, views=8, res=512, grid=64, regime="init", step="pipeline",
                          no_fine=False, encoder_layers=12, raster_api="views", streams=1, no_cpu_baseline=True, no_roofline=True,
                          no_side_legs=True, fine_mask="reference", lr=0.0, ms_ssim=False, no_optimizer=False, accumulate=1)
dev = torch.device("cuda:0")
full_step, info = bench.make_pipeline_step(args, dev, 0, 1, False)
for _ in range(3):
    full_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(a.steps):
        full_step()
    torch.cuda.synchronize()

OURS = re.compile(r"\(anonymous namespace\)::(?!.*at::native)")


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)(?:<[^>]*>)?.*?(?:at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+))?", name)
    if name.startswith("Cijk"):
        return "GEMM " + name[:24] + " " + "".join(re.findall(r"MT\d+x\d+x\d+", name)[:1])
    if m:
        inner = re.findall(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", name)
        return "/".join(dict.fromkeys(inner[:3]))
    return name[:60]


def site(evt):
    e = evt
    while e is not None:
        for fr in (e.stack or []):
            if "/lara_amd/" in fr or "bench.py" in fr or "/tools/" in fr:
                m = re.match(r"(.*?)\((\d+)\): (\S+)", fr)
                if m:
                    return f"{os.path.relpath(m.group(1), ROOT)}:{m.group(2)} {m.group(3)}"
                return fr[:80]
        e = e.cpu_parent
    e, top = evt, evt
    while e is not None:
        top = e
        e = e.cpu_parent
    return "[autograd] " + top.name[:70]


acc = collections.defaultdict(lambda: [0, 0.0])
total = 0.0
for evt in prof.events():
    for k in (evt.kernels or []):
        if OURS.search(k.name):
            continue
        # count a kernel at the innermost operator that owns it
        if any(k in (c.kernels or []) for c in (evt.cpu_children or [])):
            continue
        key = (site(evt), short(k.name))
        acc[key][0] += 1
        acc[key][1] += k.duration
        total += k.duration
print(f"torch kernels: {total / a.steps / 1e3:.2f} ms per step over {sum(v[0] for v in acc.values()) / a.steps:.0f} launches")
by_site = collections.defaultdict(lambda: [0, 0.0])
for (s, _), v in acc.items():
    by_site[s][0] += v[0]
    by_site[s][1] += v[1]
print("---- by site")
for s, v in sorted(by_site.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{v[1] / a.steps:9.1f} us {v[0] / a.steps:7.1f}  {s}")
print("---- by (site, kernel)")
for (s, k), v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:a.top]:
    print(f"{v[1] / a.steps:9.1f} us {v[0] / a.steps:7.1f}  {s}  <- {k}")
