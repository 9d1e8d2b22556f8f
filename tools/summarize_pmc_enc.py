#!/usr/bin/env python
"""Mean per launch of the counters tools/gpu_pmc_enc.sh collected for the encoder forward kernels -> profiles/<tag>_encoder_pmc.txt
usage: tools/summarize_pmc_enc.py <tag>"""
import collections, csv, glob, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmce_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
dur, dn = collections.defaultdict(float), collections.Counter()
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmce_{tag}_1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        dn[k] += 1
lines = ["# encoder forward kernels (tools/encoder_bench.py --layers 2, 4 scenes): rocprofv3 --pmc, separate passes, mean per launch",
         "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES (matrix-pipe busy cycles, summed over the chip's 1024 SIMDs) / (1024 x launch duration x 2.4 GHz);",
         "# the duration is the launch's wall time under the counter pass (kernel trace of the same pass); under MFMA load the part runs well",
         "# below 2.4 GHz (second line: the clock estimated from SQ_BUSY_CYCLES, and the busy fraction at that clock)"]
for k, d in sorted(agg.items()):
    if "gemm" not in k and "attn" not in k and "conv" not in k and "ln_" not in k:
        continue
    n = {c: d[c] / cnt[(k, c)] for c in d}
    lines.append(k)
    for c in sorted(n):
        lines.append(f"   {c:34s} {n[c]:16.0f}")
    if n.get("SQ_VALU_MFMA_BUSY_CYCLES") and dn[k]:
        us = dur[k] / dn[k] / 1e3
        lines.append(f"   -> launch {us:8.1f} us, MFMA busy {n['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * us * 2400.0):.3f} of SIMD cycles @2.4 GHz")
        if n.get("SQ_BUSY_CYCLES"):   # summed over the 32 shader engines; a launch that fills the chip keeps every one busy throughout
            ghz = n["SQ_BUSY_CYCLES"] / 32.0 / (us * 1e3)
            lines.append(f"      clock under the counter pass ~ SQ_BUSY_CYCLES / 32 / duration = {ghz:.2f} GHz -> MFMA busy "
                         f"{n['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * us * 1e3 * ghz):.3f} of the SIMD cycles that elapsed")
open(os.path.join(root, "profiles", f"{tag}_encoder_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
