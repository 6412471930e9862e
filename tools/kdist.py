#!/usr/bin/env python3
"""Per-kernel duration distribution (p10 / p50 / p90 / max, us) of a rocprofv3 --kernel-trace run.
    python tools/kdist.py <dir with */*kernel_trace.csv> [skip_first_n_dispatches_per_kernel]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = sorted(glob.glob(d + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0].replace("ratsdf::", "").replace("void ", "")
    dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(dur.items()):
    if not n.startswith("k_"):
        continue
    first = v[0]
    v = sorted(v[skip:]) or [0.0]
    q = lambda p: v[min(len(v) - 1, int(p * len(v)))]
    print(f"{n:24s} n={len(v):5d} first={first:8.1f} p10={q(.1):7.1f} p50={q(.5):7.1f} p90={q(.9):7.1f} max={v[-1]:8.1f} mean={sum(v)/len(v):7.1f}")
