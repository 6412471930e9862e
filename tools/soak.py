#!/usr/bin/env python3
"""Long parity soak (GPU box): batched frames through the pipelined engine vs the frame-at-a-time CPU
oracle, with a small directory so that chained buckets, slow deletes and pool reuse all occur.
tools/soak.py [frames] [bucket_bits] [voxel_size]   (bucket_bits 9 + voxel 0.01: ~1 600 blocks in 512 buckets, chains everywhere)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd")); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
torch.cuda.init()
import ratsdf
from ratsdf import synthetic
from ratsdf._abi import Engine
from oracle_binding import load_oracle
from parity import assert_maps_equal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
bb = int(sys.argv[2]) if len(sys.argv) > 2 else 13
kw = dict(bucket_bits=bb, block_bits=14)
vs, md = (float(sys.argv[3]) if len(sys.argv) > 3 else 0.02), 4.0
gpu = ratsdf.TSDFGrid(vs, 6 * vs, **kw)
cpu = Engine(load_oracle(), vs, 6 * vs, threads=8, **kw)
dev = torch.device("cuda", 0)
base = synthetic.stream("room", 60, scale=0.25, noise=True, holes=True) + synthetic.stream("sphere", 40, scale=0.25, noise=True)
seq = base + base[::-1]
d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in seq]
h, w = seq[0]["depth"].shape
rng = np.random.default_rng(0)
at = 0
slow = 0
while at < n:
    c = int(rng.integers(1, 17))
    idx = [(at + j) % len(seq) for j in range(c)]
    b = gpu.make_batch([d[i]["rgb"].data_ptr() for i in idx], [d[i]["depth"].data_ptr() for i in idx],
                       [d[i]["ht"].data_ptr() for i in idx], [d[i]["lt"].data_ptr() for i in idx], h, w, md,
                       [seq[i]["intrinsics"] for i in idx], [seq[i]["pose"] for i in idx])
    gpu.integrate_device_batch(b)
    for i in idx:
        f = seq[i]
        cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
        slow += cpu.last_frame_stats()["slow_requests"]
    at += c
    if (at // 100) != ((at - c) // 100):
        worst = assert_maps_equal(gpu, cpu)
        print(at, "frames ok", worst, "active", cpu.num_active_blocks(), "slow requests so far", slow, flush=True)
assert gpu.totals() == cpu.totals()
print("SOAK OK", at, "frames")
