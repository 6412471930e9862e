#!/usr/bin/env python3
"""The ray cast on a map whose block filter is saturated (tens of thousands of blocks: nearly every bit of the 16 KiB
Bloom filter set, so the march falls back on directory probes) against the CPU oracle: 1280x720 / 2 mm, 40 frames of the
'room' sweep.  tools/raycast_bigmap_check.py   (GPU box)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import ratsdf
from ratsdf import synthetic
from ratsdf._abi import Engine
from oracle_binding import load_oracle

vs, md = 0.002, 4.0
gpu = ratsdf.TSDFGrid(vs, 6 * vs)
cpu = Engine(load_oracle(), vs, 6 * vs, threads=16)
frames = [synthetic.frame("room", i % 20, cam="l515_720p", noise=True, holes=True) for i in range(40)]
for f in frames:
    for e in (gpu, cpu):
        e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
print("map:", gpu.num_active_blocks(), "blocks (oracle", cpu.num_active_blocks(), ")")
for name, f in (("view 5", frames[5]), ("view 19", frames[19])):
    H, W = f["depth"].shape
    t0 = time.perf_counter()
    ga, gn = gpu.raycast(f["intrinsics"], H, W, f["pose"], 2 * md)
    t1 = time.perf_counter()
    ca, cn = cpu.raycast(f["intrinsics"], H, W, f["pose"], 2 * md)
    d = max(int(np.abs(ga.astype(np.int16) - ca.astype(np.int16)).max()), int(np.abs(gn.astype(np.int16) - cn.astype(np.int16)).max()))
    frac = float(((ga != ca) | (gn != cn)).mean())
    print(f"{name}: HIP {1e3 * (t1 - t0):.2f} ms, hit {float((ga[..., 3] == 255).mean()):.3f}; max byte difference {d}, bytes that differ {frac:.2e}")
    assert d <= 1 and frac < 1e-3
print("OK")
