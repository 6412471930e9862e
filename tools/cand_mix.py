#!/usr/bin/env python3
"""Instruction mix of the stand-alone candidate pass (k_cand) per basic block, with the source lines behind each:
  make -C ra-slam_amd/csrc listing ; python tools/cand_mix.py [symbol-prefix]
(`tools/hotpath.py` covers only the update loop.)"""
import collections
import re
import sys
from pathlib import Path

A = Path(__file__).resolve().parent.parent / "ra-slam_amd/csrc/build/engine_g.s"
sym = sys.argv[1] if len(sys.argv) > 1 else "_ZN6ratsdf6k_candE"
lines = A.read_text().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1].replace("kernels_", "k_").replace(".h", "")
s = [i for i, l in enumerate(lines) if l.startswith(sym) and ": " in l][0]
e = next(i for i in range(s, len(lines)) if ".amdhsa_next_free_sgpr" in lines[i])
new = lambda name: {"name": name, "v": 0, "s": 0, "m": 0, "lines": collections.Counter(), "br": []}
blocks, cur, loc = [], new("entry"), None
for l in lines[s:e]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = new(m.group(1))
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    t = l.strip()
    if t.startswith("v_"):
        cur["v"] += 1
        cur["lines"][loc] += 1
    elif t.startswith("s_cbranch") or t.startswith("s_branch"):
        cur["br"].append(t.split()[0][2:] + ">" + t.split()[-1])
        cur["s"] += 1
    elif t.startswith("s_"):
        cur["s"] += 1
    elif re.match(r"(ds_|global_|buffer_|flat_)", t):
        cur["m"] += 1
blocks.append(cur)
print(f"{sym}: {sum(b['v'] for b in blocks)} vector, {sum(b['s'] for b in blocks)} scalar, "
      f"{sum(b['m'] for b in blocks)} memory / LDS instructions (static)")
for b in blocks:
    if b["v"] + b["m"] < 3:
        continue
    top = " ".join(f"{fn}:{ln}x{c}" for (fn, ln), c in b["lines"].most_common(6) if c)
    print(f"{b['name']:10s} V{b['v']:4d} S{b['s']:4d} M{b['m']:3d} {' '.join(b['br'])[:44]:44s} {top}")
