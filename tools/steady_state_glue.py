#!/usr/bin/env python
"""Per-step torch / runtime ("glue") kernels of the pipeline step from a rocprofv3 kernel TRACE, steady state only.

`--stats` sums over the whole process -- scene construction, parameter initialisation, the first step's one-off work included --
so dividing its `at::native` + `rocclr` rows by the number of steps (4.6 ms / 693 launches) overstates what a
step costs; the torch profiler (tools/glue_probe.py: 2.4 ms / 300 launches) sees only the steps it brackets.  This reads the
trace itself, cuts it at the optimiser's kernels (one group per step) and reports the LAST step.
usage: tools/steady_state_glue.py <kernel_trace.csv> [--top N]"""
import collections, csv, re, sys

path = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step ends with AdamW's fused multi-tensor kernels: the last dispatch of each run of them is a step boundary
opt = [i for i, (_, _, n) in enumerate(rows) if "FusedOptimizerTensorListMetadata" in n]
ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] - i > 50]      # (the update is a few runs of them, a few dispatches apart)
if len(ends) < 2:
    raise SystemExit("fewer than two optimiser steps in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1


def is_glue(n):
    return n.startswith("void at::") or "at::native" in n or "rocclr" in n or n.startswith("Cijk") or "rocprim" in n or "hipcub" in n


def short(n):
    m = re.search(r"(\w+Functor\w*<[^,>]*|\w+_kernel\w*|indexFunc\w+|CatArray\w+|__amd_rocclr_\w+|Cijk_\w{0,12})", n)
    f = re.search(r"(FillFunctor<\w+>|CUDAFunctor_add<\w+>|MulFunctor|direct_copy|bfloat16_copy|where_kernel|compare_scalar|sum_functor|MeanOps|"
                  r"LpNormFunctor|FusedOptimizer|uniform_and_transform|flip_kernel|div_floor|arange|gather)", n)
    return (m.group(1) if m else n[:40]) + (" " + f.group(1) if f else "")


step = rows[lo:hi]
glue = [(e - s, n) for s, e, n in step if is_glue(n)]
lib = [(e - s, n) for s, e, n in step if not is_glue(n)]
print(f"last step of {len(ends)}: {len(step)} dispatches, {sum(e - s for s, e, _ in step) / 1e6:.2f} ms of kernels over "
      f"{(step[-1][1] - step[0][0]) / 1e6:.2f} ms; library {len(lib)} launches {sum(t for t, _ in lib) / 1e6:.2f} ms; "
      f"torch / runtime {len(glue)} launches {sum(t for t, _ in glue) / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for t, n in glue:
    a = agg[short(n)]
    a[0] += t; a[1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{t / 1e3:9.1f} us {c:5d}  {k}")
