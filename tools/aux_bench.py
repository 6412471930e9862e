#!/usr/bin/env python3
"""Timings of the query-side entry points on the bench map, HIP engine vs
the CPU oracle (16 threads): tools/aux_bench.py  (GPU box)"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
torch.cuda.init()
import ratsdf
from ratsdf import synthetic
from ratsdf._abi import Engine
from oracle_binding import load_oracle

vs, md = 0.005, 4.0
gpu = ratsdf.TSDFGrid(vs, 6 * vs)
cpu = Engine(load_oracle(), vs, 6 * vs, threads=16)
frames = [synthetic.frame("room", i, noise=True, holes=True) for i in range(45)]
for f in frames:
    for e in (gpu, cpu):
        e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
print("map:", gpu.num_active_blocks(), "blocks")

def timed(fn, reps):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t) / reps, r

f = frames[20]
H, W = f["depth"].shape
rows = []
for name, g, c, reps in [
    ("raycast 640x480 (host images out)", lambda: gpu.raycast(f["intrinsics"], H, W, f["pose"], 2 * md),
     lambda: cpu.raycast(f["intrinsics"], H, W, f["pose"], 2 * md), 20),
    ("gather_valid (16-B records)", gpu.gather_valid, cpu.gather_valid, 5),
    ("gather_valid_semantic (20-B records)", gpu.gather_valid_semantic, cpu.gather_valid_semantic, 5),
    ("query, 1 m cube", lambda: gpu.query((-0.5, 0.5, -0.5, 0.5, 1.0, 2.0)), lambda: cpu.query((-0.5, 0.5, -0.5, 0.5, 1.0, 2.0)), 10),
    ("gather_valid_mesh (marching cubes)", gpu.gather_valid_mesh, cpu.gather_valid_mesh, 3),
]:
    tg, rg = timed(g, reps)
    tc, rc = timed(c, max(1, reps // 3))
    n = len(rg[0]) if isinstance(rg, tuple) else len(rg)
    rows.append((name, tg * 1e3, tc * 1e3, n))
    print(f"{name:40s} HIP {tg * 1e3:8.2f} ms   CPU-16T oracle {tc * 1e3:9.2f} ms   ({n} items)")
