#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv -> one line per kernel (name, calls, average / min / max us): tools/kstats_print.py <csv>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "").replace("ratsdf::", "")
    print(f'{n:28s} {r["Calls"]:>6s}  avg {float(r["AverageNs"]) / 1e3:8.2f}  min {float(r["MinNs"]) / 1e3:8.2f}  max {float(r["MaxNs"]) / 1e3:8.2f} us')
