#!/usr/bin/env python
"""How long does the HOST need to enqueue one pipeline step, against how long the device needs to run it?  (If the two are
close the step is launch-bound and fewer / larger launches pay directly; if the host is well ahead they do not.)
usage: python tools/host_probe.py [--streams 2]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=2)
a = ap.parse_args()
args = argparse.Namespace(gpus=1, steps=5, warmup=2, scenes=4, views=8, res=512, grid=64, regime="init", step="pipeline", no_fine=False,
                          encoder_layers=12, raster_api="views", streams=a.streams, no_cpu_baseline=True, no_roofline=True,
                          no_side_legs=True, fine_mask="reference")
full_step, info = bench.make_pipeline_step(args, torch.device("cuda:0"), 0, 1, False)
for _ in range(3):
    full_step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(6):
    t0 = time.perf_counter()
    full_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(1e3 * (t1 - t0))
    total.append(1e3 * (t2 - t0))
print("host enqueue ms per step (includes the step's own host reads of the subset sizes):", [round(x, 1) for x in host])
print("step ms with a sync after each:", [round(x, 1) for x in total])
