#!/usr/bin/env python3
"""Vector-ALU issue fraction of the frame's kernels from a tools/sq.sh pass:

    issue = SQ_INSTS_VALU x 4 cycles / (1 024 SIMDs x kernel duration x 2.4 GHz)

(a wave instruction occupies its SIMD's 16 lanes for 4 cycles; 256 CUs x 4 SIMDs; MI355X_MICROARCH.md chip table:
2.4 GHz engine clock).  Durations: the rocprofv3 --kernel-trace --stats summaries of the same build (profiles/
<tag>_kernel_stats_<cfg>.csv); `frame` adds up every kernel of a frame against the frame period of the bench line.

    tools/issue_json.py <tag>_sq_counters.txt <dir with <tag>_kernel_stats_*.csv and <tag>_bench_line*.json> <tag>
      -> profiles/issue_latest.json {vga5mm, hd2mm}, profiles/issue_group4.json {group}"""
import csv
import json
import re
import sys
from pathlib import Path

SIMDS, CLOCK_GHZ = 1024, 2.4
sq, pdir, tag = Path(sys.argv[1]), Path(sys.argv[2]), sys.argv[3]
insts = {}   # workload -> kernel -> SQ_INSTS_VALU per launch
for line in sq.read_text().splitlines():
    m = re.match(r"(\S+) (k_\S+?)(?:<.*?>)? (.*)", line)
    if not m or "SQ_INSTS_VALU=" not in line:
        continue
    wl, k = m.group(1), m.group(2)
    insts.setdefault(wl, {})[k] = float(re.search(r"SQ_INSTS_VALU=(\S+)", line).group(1))


def durations(cfg):
    f = pdir / f"{tag}_kernel_stats_{cfg}.csv"
    out = {}
    if not f.exists():
        return out
    for r in csv.DictReader(f.open()):
        name = r["Name"].split("(")[0].replace("void ", "").replace("ratsdf::", "")
        name = re.sub(r"<.*", "", name)
        calls, avg = int(r["Calls"]), float(r["AverageNs"])
        # (the graph-replayed k_*_g rows and the sampled k_* rows run the same body: the row with more calls counts)
        if name not in out or calls > out[name][0]:
            out[name] = (calls, avg)
    return {k: v[1] for k, v in out.items()}


def block(wl, cfg, k_int, k_front, period_us):
    d = durations(cfg)
    iv = insts.get(wl, {})
    ki = next((k for k in (k_int, k_int + "_g") if k in iv), None)
    kd = next((k for k in (k_int + "_g", k_int) if k in d), None)
    if ki is None or kd is None:
        return None
    us = d[kd] / 1e3
    frac = iv[ki] * 4 / (SIMDS * us * 1e3 * CLOCK_GHZ)
    rec = dict(kernel=k_int, valu_instructions_per_launch=round(iv[ki]), avg_launch_us=round(us, 2),
               valu_issue_frac=round(frac, 4),
               formula="SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz)")
    tot = sum(v for k, v in iv.items() if k.startswith(("k_integrate", "k_front")))
    if period_us:
        rec["frame"] = dict(valu_instructions=round(tot), period_us=round(period_us, 2),
                            valu_issue_frac=round(tot * 4 / (SIMDS * period_us * 1e3 * CLOCK_GHZ), 4))
    return rec


def period(cfg, key=None):
    f = pdir / (f"{tag}_bench_line.json" if cfg == "vga5mm" else f"{tag}_bench_line_{cfg}.json")
    if not f.exists():
        return None
    d = json.loads(f.read_text())
    if key:
        return d[key]["us_per_frame_step"]
    return d["ms_per_step"] * 1e3 / d["config"]["frames_per_step"]


latest = {}
for wl, cfg in (("vga", "vga5mm"), ("hd2mm", "hd2mm"), ("bigmap", "bigmap")):
    b = block(wl, cfg, "k_integrate", "k_front", period(cfg))
    if b:
        latest[cfg] = b
out = Path(__file__).resolve().parent.parent / "profiles"
(out / "issue_latest.json").write_text(json.dumps(latest, indent=1) + "\n")
g = None
if "groupS4" in insts:
    d = durations("vga5mm")
    iv = insts["groupS4"]
    # the S = 4 launches of the vga5mm collection (bench.py --streams 4) are the k_integrate_g rows with FEWER calls
    f = pdir / f"{tag}_kernel_stats_group4.csv"
    if f.exists():
        for r in csv.DictReader(f.open()):
            if "k_integrate_g" in r["Name"]:
                us = float(r["AverageNs"]) / 1e3
                g = dict(kernel="k_integrate_g", streams=4, valu_instructions_per_launch=round(iv.get("k_integrate_g", 0)),
                         avg_launch_us=round(us, 2),
                         valu_issue_frac=round(iv.get("k_integrate_g", 0) * 4 / (SIMDS * us * 1e3 * CLOCK_GHZ), 4),
                         formula="SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz)")
                break
if g:
    (out / "issue_group4.json").write_text(json.dumps({"group": g}, indent=1) + "\n")
print(json.dumps(latest, indent=1))
print(json.dumps(g))
