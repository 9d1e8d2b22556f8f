#!/usr/bin/env python
"""Turn the raw rocprofv3 output of tools/gpu_profile.sh / gpu_traffic.sh (under gpurun_out/) into the
small summaries committed under profiles/.   usage: tools/summarize_profiles.py <tag>"""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def short(name):
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    return m.group(1).replace("_kernel", "") if m else name.split("(")[0][:40]


def pmc_table(pattern):
    agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    return agg, cnt


sts = glob.glob(os.path.join(out, f"prof_{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
st = sts[0] if sts else ""
if st:
    shutil.copy(st, os.path.join(prof, f"{tag}_kernel_stats.csv"))
sts = glob.glob(os.path.join(out, f"prof_{tag}_stats_train", "**", "*kernel_stats.csv"), recursive=True)
if sts:
    shutil.copy(sts[0], os.path.join(prof, f"{tag}_kernel_stats_train_step.csv"))
sts = glob.glob(os.path.join(out, f"prof_{tag}_stats_pipe1", "**", "*kernel_stats.csv"), recursive=True)
if sts:
    shutil.copy(sts[0], os.path.join(prof, f"{tag}_kernel_stats_pipeline_one_stream.csv"))
agg, cnt = pmc_table(os.path.join(out, f"prof_{tag}_pmc*", "**", "*counter_collection.csv"))
if agg:
    cols = sorted({c for d in agg.values() for c in d})
    with open(os.path.join(prof, f"{tag}_pmc_summary.csv"), "w") as f:
        f.write("kernel,launches," + ",".join(cols) + "\n")
        for k, d in agg.items():
            n = max(cnt[(k, c)] for c in cols)
            f.write(f"{k},{n}," + ",".join(str(int(d.get(c, 0) / max(cnt[(k, c)], 1))) for c in cols) + "\n")
agg, cnt = pmc_table(os.path.join(out, f"prof_{tag}_vpmc", "**", "*counter_collection.csv"))      # the multi-view launches
if agg:
    cols = sorted({c for d in agg.values() for c in d})
    with open(os.path.join(prof, f"{tag}_views_pmc_summary.csv"), "w") as f:
        f.write("kernel,launches," + ",".join(cols) + "\n")
        for k, d in agg.items():
            n = max(cnt[(k, c)] for c in cols)
            f.write(f"{k},{n}," + ",".join(str(int(d.get(c, 0) / max(cnt[(k, c)], 1))) for c in cols) + "\n")
traffic = {}
for c, key in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    agg, cnt = pmc_table(os.path.join(out, f"traffic_{tag}_{c}", "**", "*counter_collection.csv"))
    for k, d in agg.items():
        if "lara" in k or k.endswith("_fwd") or k.endswith("_bwd") or "tile_" in k or "scatter" in k or "<" in k:
            name = re.sub(r"<.*", "", k)
            traffic.setdefault(name, {})[key] = int(d[c] / cnt[(k, c)] * 1024)  # counter unit: KB
if traffic:
    for v in traffic.values():
        v["total"] = v.get("fetch", 0) + v.get("write", 0)
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                         "tools/gpu_traffic.sh), bench.py --scenes 1 --regime init, averaged over the launches of "
                         "each kernel; counter unit KB; raw values (see DESIGN.md section 5 for the calibration)",
               "bytes_per_launch": traffic}, open(os.path.join(prof, f"traffic_{tag}.json"), "w"), indent=1)
print("wrote", [f for f in os.listdir(prof) if f.startswith(tag) or f == f"traffic_{tag}.json"])
