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
# ---- HBM traffic, calibrated (tools/ubench/traffic_calib.hip -> profiles/<tag>_traffic_calibration.json) ----------------------
# gfx950's FETCH_SIZE reports HALF of the bytes read (128-byte requests tallied at 64 bytes) for every read pattern this library
# uses -- 4 B and 16 B per lane streams, 80-byte records read in order, 80-byte and 128-byte records gathered through an index
# list, HBM-resident and Infinity-Cache-resident sizes alike; WRITE_SIZE is 1:1.  So: bytes = 2 x FETCH_SIZE + WRITE_SIZE.
calib = {}
known_f = os.path.join(out, f"{tag}_traffic_calib_known.json")
if os.path.exists(known_f):
    known = json.load(open(known_f))["known_bytes"]
    meas = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for f in glob.glob(os.path.join(out, f"calib_{tag}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                agg[k] += float(r["Counter_Value"]) * 1024; cnt[k] += 1
        for k in agg:
            meas.setdefault(k, {})[c] = agg[k] / cnt[k]
    for k, kb in known.items():
        if k in meas:
            rd = kb.get("read", kb.get("read_referenced"))
            calib[k] = {"known_read": rd, "known_write": kb["write"], "FETCH_SIZE": int(meas[k].get("FETCH_SIZE", 0)),
                        "WRITE_SIZE": int(meas[k].get("WRITE_SIZE", 0)),
                        "fetch_over_known_read": round(meas[k].get("FETCH_SIZE", 0) / rd, 3),
                        "write_over_known": round(meas[k].get("WRITE_SIZE", 0) / kb["write"], 3), **{a: b for a, b in kb.items() if a not in ("read", "write")}}
    if calib:
        json.dump({"source": "tools/ubench/traffic_calib.hip under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); counter "
                             "unit KB; every buffer touched exactly once per launch",
                   "finding": "FETCH_SIZE = 0.5 x bytes read for every pattern (streams of 4 B and 16 B per lane, 80-byte records in order and "
                              "gathered, 128-byte records gathered; HBM-resident and cache-resident sizes); a gathered 80-byte record costs 1.5 "
                              "lines of 128 bytes (it straddles a line boundary 4 times in 8); WRITE_SIZE = 1.0 x bytes written",
                   "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE", "patterns": calib},
                  open(os.path.join(prof, f"{tag}_traffic_calibration.json"), "w"), indent=1)
traffic = {}
for c, key in (("FETCH_SIZE", "fetch_counter"), ("WRITE_SIZE", "write_counter")):
    # (the second directory: per-view calls under no_grad -- the forward-only instantiation of the library, tools/gpu_traffic.sh)
    for sub, suffix in ((f"traffic_{tag}_{c}", ""), (f"traffic_{tag}_{c}_fwdonly", "@forward_only_call")):
        agg, cnt = pmc_table(os.path.join(out, sub, "**", "*counter_collection.csv"))
        for k, d in agg.items():
            if "lara" in k or k.endswith("_fwd") or k.endswith("_bwd") or "tile_" in k or "scatter" in k or "<" in k:
                name = re.sub(r"<.*", "", k) + suffix
                traffic.setdefault(name, {})[key] = int(d[c] / cnt[(k, c)] * 1024)  # counter unit: KB
if traffic:
    for v in traffic.values():
        v["read"] = 2 * v.get("fetch_counter", 0)         # calibrated (see above)
        v["write"] = v.get("write_counter", 0)
        v["total"] = v["read"] + v["write"]
    cal = [f for f in sorted(os.listdir(prof)) if f.endswith("_traffic_calibration.json")]
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_traffic.sh), bench.py "
                         "--scenes 1, averaged over the launches of each kernel; counter unit KB.  `read` = 2 x FETCH_SIZE, `write` = "
                         f"WRITE_SIZE, `total` = their sum: the calibration of profiles/{cal[-1] if cal else '<tag>_traffic_calibration.json'} "
                         "(gfx950's FETCH_SIZE tallies 128-byte requests at 64 bytes for every access pattern of this library).  Files up "
                         "to round 3 (traffic_r01*-r03*) hold the RAW counter sums: double their `fetch`.",
               "bytes_per_launch": traffic}, open(os.path.join(prof, f"traffic_{tag}.json"), "w"), indent=1)
print("wrote", [f for f in os.listdir(prof) if f.startswith(tag) or f == f"traffic_{tag}.json"])
