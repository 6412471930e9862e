#!/usr/bin/env python3
"""GPU box: ratsdf_integrate_batch(pinned) alone -- frames/s and link rate of the page-locked host path, 32 frames
per call from one arena.  Run under `rocprofv3 --memory-copy-trace --kernel-trace --output-format csv` and feed the
trace to tools/copy_gaps.py to see where the link idles.   usage: tools/pinned_probe.py [frames] [frames per call]"""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np
import ratsdf
from ratsdf import synthetic
total = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
C = int(sys.argv[2]) if len(sys.argv) > 2 else 32
half = [synthetic.frame("room", i, noise=True, holes=True) for i in range(C)]
frames = half + half[::-1]
eng = ratsdf.TSDFGrid(0.005, 0.03)
npx = frames[0]["depth"].size
arena = eng.host_alloc((len(frames) * npx * 16,), np.uint8)
pin = []
for i, f in enumerate(frames):
    blk = arena[i * npx * 16:(i + 1) * npx * 16]
    g = dict(f)
    g["depth"] = blk[:npx * 4].view(np.float32).reshape(f["depth"].shape)
    g["ht"] = blk[npx * 4:npx * 8].view(np.float32).reshape(f["ht"].shape)
    g["lt"] = blk[npx * 8:npx * 12].view(np.float32).reshape(f["lt"].shape)
    g["rgb"] = blk[npx * 12:npx * 15].reshape(f["rgb"].shape)
    for k in ("rgb", "depth", "ht", "lt"):
        g[k][...] = f[k]
    pin.append(g)
chunks = [pin[c0:c0 + C] for c0 in range(0, len(pin), C)]
calls = [eng.make_host_batch(ch, 4.0, pinned=True) for ch in chunks]
for call in calls:
    eng.integrate_host_batch(call)
n, t_calls = 0, []
t0 = time.perf_counter()
while n < total:
    for ch, call in zip(chunks, calls):
        t1 = time.perf_counter()
        eng.integrate_host_batch(call)
        t_calls.append(time.perf_counter() - t1)
        n += len(ch)
eng.synchronize()
dt = time.perf_counter() - t0
print(f"pinned path: {n / dt:.1f} frames/s, {n * npx * 15 / dt / 1e9:.1f} GB/s; in calls {sum(t_calls):.4f} s of {dt:.4f} s "
      f"(median call {sorted(t_calls)[len(t_calls) // 2] * 1e3:.3f} ms for {C} frames)")
eng.host_free(arena)
