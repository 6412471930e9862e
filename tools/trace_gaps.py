#!/usr/bin/env python3
"""Where the time between frames goes, from a rocprofv3 --kernel-trace CSV of a bench.py run:
steady-state period (k_front start to k_front start), what a batch boundary costs, idle gaps.

    python tools/trace_gaps.py <dir with *kernel_trace.csv>
"""
import collections
import csv
import glob
import os
import statistics
import sys


def main():
    fs = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    f = max(fs, key=os.path.getmtime)
    rows = []
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("ratsdf::", "").replace("void ", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    fr = [i for i, (a, b, n) in enumerate(rows) if n == "k_front"]
    gaps = [(rows[fr[i + 1]][0] - rows[fr[i]][0]) / 1e3 for i in range(len(fr) - 1)]
    # a batch boundary = a gap that contains something else than the frame's own launches
    own = {"k_front", "k_alloc_rank", "k_integrate<2>"}  # (k_alloc_rank: only with RATSDF_FUSED_SERIAL=0)
    inner, boundary = [], []
    for i in range(len(fr) - 1):
        names = {rows[j][2] for j in range(fr[i], fr[i + 1])}
        (inner if names <= own else boundary).append(gaps[i])
    s = sorted(inner)
    print(f"frames {len(gaps)}: inner {len(inner)} median {statistics.median(inner):.2f} mean "
          f"{sum(inner) / len(inner):.2f} p90 {s[int(.9 * len(s))]:.2f} p99 {s[int(.99 * len(s))]:.2f} max {s[-1]:.2f} us")
    if boundary:
        print(f"batch boundaries {len(boundary)}: mean {sum(boundary) / len(boundary):.1f} us "
              f"(= {sum(boundary) / len(boundary) - statistics.median(inner):.1f} us over a steady frame)")
    dur = collections.defaultdict(list)
    idle = []
    for i in range(1, len(rows)):
        dur[rows[i][2]].append((rows[i][1] - rows[i][0]) / 1e3)
        if rows[i][2] in own and rows[i - 1][2] in own:
            idle.append((rows[i][0] - rows[i - 1][1]) / 1e3)
    print("kernel means:", {k: round(sum(v) / len(v), 2) for k, v in dur.items() if k in own or k in ("k_cand", "k_settle")})
    print(f"idle between a frame's launches: median {statistics.median(idle):.2f} mean {sum(idle) / len(idle):.2f} us")
    total = (rows[fr[-1]][0] - rows[fr[0]][0]) / 1e3
    print(f"whole trace: {total / (len(fr) - 1):.2f} us per frame")


if __name__ == "__main__":
    main()
