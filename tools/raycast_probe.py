#!/usr/bin/env python3
"""Which waves of k_raycast are slow, and why (GPU box, diagnostic build): per wave, the wall time, the groups and
samples of its slowest lane and how many lanes hit.
RATSDF_LIB=.../libratsdf_stamps.so tools/raycast_probe.py"""
import sys, time, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np
import ratsdf
from ratsdf import synthetic

vs, md = 0.005, 4.0
gpu = ratsdf.TSDFGrid(vs, 6 * vs)
frames = [synthetic.frame("room", i, noise=True, holes=True) for i in range(45)]
for f in frames:
    gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
print("map:", gpu.num_active_blocks(), "blocks")
dll = gpu.lib.dll
for name, f in (("view 20", frames[20]), ("view 44", frames[44])):
    H, W = f["depth"].shape
    gpu.raycast(f["intrinsics"], H, W, f["pose"], 2 * md)
    dll.ratsdf_debug_wave_stamps(gpu._h, 1)
    # ws[0] is an atomicMin target: start from all-ones
    rgba, _n = gpu.raycast(f["intrinsics"], H, W, f["pose"], 2 * md)
    buf = (C.c_ulonglong * (16384 * 8))()
    st = dll.ratsdf_debug_wave_records(gpu._h, buf, C.c_size_t(16384 * 8))
    a = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 8)
    used = a[:, 2] > 0
    a = a[used]
    t_end = a[:, 2].astype(np.int64)
    t0 = t_end.min()
    life_end = (t_end - t0) * 0.01
    order = np.argsort(-life_end)
    print(f"{name}: {used.sum()} waves; end times (us after the first wave ended): p50 {np.median(life_end):.1f} p90 {np.percentile(life_end, 90):.1f} "
          f"p99 {np.percentile(life_end, 99):.1f} max {life_end.max():.1f}")
    print("   slowest waves: end us | groups (slowest lane) | samples (slowest lane) | lanes that hit | a full-length lane's cycles: positions + judging, probes, voxel loads")
    for w in order[:8]:
        print(f"   {life_end[w]:8.1f} | {int(a[w, 3]):5d} | {int(a[w, 4]):5d} | {int(a[w, 5]):3d} | {int(a[w, 1])} {int(a[w, 6])} {int(a[w, 7])}")
    g = a[:, 3].astype(np.float64)
    print(f"   groups of the slowest lane per wave: mean {g.mean():.1f} p90 {np.percentile(g, 90):.0f} max {g.max():.0f}; "
          f"correlation of a wave's end time with its groups: {np.corrcoef(g, life_end)[0, 1]:.3f}")
