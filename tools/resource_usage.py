#!/usr/bin/env python3
"""Prints VGPR / SGPR / occupancy / LDS / scratch per kernel of the gfx950 engine (hipcc remarks)."""
import re
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "ra-slam_amd" / "csrc"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
       "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null",
       "ratsdf_engine.hip"]
err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True,
                             text=True).stdout.split("(")[0].strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w /\[\]]+): (\w+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
print(f"{'kernel':44s} {'VGPR':>5s} {'SGPR':>5s} {'occ':>4s} {'LDS':>7s} {'scratch':>8s}")
for k, v in rows.items():
    print(f"{k:44s} {v.get('VGPRs', '?'):>5s} {v.get('TotalSGPRs', '?'):>5s} "
          f"{v.get('Occupancy [waves/SIMD]', '?'):>4s} {v.get('LDS Size [bytes/block]', '?'):>7s} "
          f"{v.get('ScratchSize [bytes/lane]', '?'):>8s}")
