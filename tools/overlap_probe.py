#!/usr/bin/env python
"""How well do the encoder (MFMA / HBM bound) and the raster (vector-ALU bound) overlap when they run on different streams?
Times K iterations of (a) the trainable VolTransformer forward + backward alone, (b) the raster forward + backward of 4 scenes x
(8 coarse + 8 fine) views alone (its own two scene streams), (c) both enqueued together on separate streams.  If (c) is close to
max(a, b), splitting the batch so that one scene group's encoder runs beside the other's raster would pay; if it is close to
a + b, the kernels only take turns.  Run on the GPU box."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=8); a = ap.parse_args()
sys.argv = [sys.argv[0], "--no-side-legs", "--no-roofline", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
from lara_amd import rasterizer
from lara_amd.encoder_train import VolTransformer
rasterizer.load_library()
scenes, settings, gc, ga = bench.build_batch(args, dev, 0)
fine_idx = bench.fine_subsets(scenes)
torch.manual_seed(0)
enc = VolTransformer(256, 800, [16], 32, 64, 80, 12, 16).to(dev)
feats = torch.randn(4, 4, 800, 16, 16, 16, device=dev)
dout = torch.randn(4, 64, 64, 64, 80, device=dev) * 1e-3
s_enc = torch.cuda.Stream()

def enc_step():
    with torch.cuda.stream(s_enc):
        out = enc(feats)
        out.backward(dout)
        for p in enc.parameters():
            p.grad = None

def ras_step():
    bench.step(scenes, settings, gc, ga, 2, fine_idx, None, api="views")

def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.iters

t_enc = timed(enc_step)
t_ras = timed(ras_step)
t_both = timed(lambda: (enc_step(), ras_step()))
t_both2 = timed(lambda: (ras_step(), enc_step()))
print(f"encoder fwd+bwd alone {t_enc:.2f} ms; raster (4 scenes x 16 views, fwd+bwd) alone {t_ras:.2f} ms; together {t_both:.2f} ms "
      f"(raster enqueued first: {t_both2:.2f}); sum {t_enc + t_ras:.2f}, max {max(t_enc, t_ras):.2f} -> overlap hides "
      f"{100 * (t_enc + t_ras - min(t_both, t_both2)) / min(t_enc, t_ras):.0f} % of the shorter one")
