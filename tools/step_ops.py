#!/usr/bin/env python
"""Which torch operators does the headline step launch, how often, and from which line of this package?  One step of
`bench.py`'s pipeline under torch.profiler (with_stack), device kernels grouped by operator (an autograd Function's
time INCLUDES the torch operators it calls, which are also listed on their own: the rows do not add up to the step).  Run on the GPU box:  python tools/step_ops.py [--top 40]"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--top", type=int, default=40); a = ap.parse_args()
sys.argv = [sys.argv[0], "--no-side-legs", "--no-roofline", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
step, info = bench.make_pipeline_step(args, dev, 0, 1, False)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    kt = sum(k.duration for k in (getattr(ev, "kernels", None) or []))
    if ev.device_type == torch.autograd.DeviceType.CPU and kt > 0:
        where = "?"
        for fr in (ev.stack or []):
            if ("lara_amd/" in fr or "bench.py" in fr) and "site-packages" not in fr:
                where = fr.replace(root + "/", "").split(" ")[0] if root in fr else fr.split(",")[0]
                break
        k = (ev.name, where)
        agg[k][0] += 1
        agg[k][1] += kt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]
tot = sum(v[1] for v in agg.values())
print(f"leaf aten ops with device time: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.2f} ms of device time in one step")
for (name, where), (n, t) in rows:
    print(f"{t / 1e3:8.3f} ms {n:5d} x  {name:34s} {where}")
