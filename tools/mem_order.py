#!/usr/bin/env python3
"""Program-order listing of the memory instructions, waits and barriers of one kernel with the source line behind
each (make the listing with -g1: see tools/cand_mix.py): where a role's dependent round trips really are.
  python tools/mem_order.py <mangled-prefix> [source-file-filter ...]"""
import re
import sys
from pathlib import Path
A = Path(__file__).resolve().parent.parent / "ra-slam_amd/csrc/build/engine_g.s"
sym = sys.argv[1]
filt = sys.argv[2:]
lines = A.read_text().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1].replace("kernels_", "k_").replace(".h", "")
s = [i for i, l in enumerate(lines) if l.startswith(sym) and ": " in l][0]
e = next(i for i in range(s, len(lines)) if ".amdhsa_next_free_sgpr" in lines[i])
loc, n = None, 0
for l in lines[s:e]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        loc = (files.get(int(m.group(1))), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    n += 1
    if re.match(r"(global_|flat_|buffer_|s_barrier|s_sleep|s_endpgm|s_load|ds_.*rtn)", t) or "vmcnt" in t:
        if not filt or (loc and any(f in str(loc[0]) for f in filt)):
            print(n, loc, t[:100])
