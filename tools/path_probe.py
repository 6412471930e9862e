#!/usr/bin/env python3
"""GPU box: which form of the serial role the frames of the benchmark streams take (ratsdf_pipeline_counters),
frames/s of the batched stream, and -- with the diagnostic build (RATSDF_LIB=.../libratsdf_stamps.so) -- the
timeline of k_front's tail.   usage: tools/path_probe.py [vga|hd] [half-sweep frames]"""
import ctypes
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
hd = len(sys.argv) > 1 and sys.argv[1] == "hd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (45 if not hd else 10)
cam, vs = ("l515_720p", 0.002) if hd else ("scannet", 0.005)
dev = torch.device("cuda", 0)
half = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(n)]
frames = half + half[::-1]
H, W = frames[0]["depth"].shape
d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
eng = ratsdf.TSDFGrid(vs, 6 * vs)
batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d],
                       [x["ht"].data_ptr() for x in d], [x["lt"].data_ptr() for x in d], H, W, 4.0,
                       [f["intrinsics"] for f in frames], [f["pose"] for f in frames])
for rep in range(3):
    eng.integrate_device_batch(batch)
    eng.synchronize()
    print("pass", rep, eng.pipeline_counters(reset=True), eng.totals(reset=True), flush=True)
stamps = "stamps" in os.environ.get("RATSDF_LIB", "")
if stamps:
    ts = eng.lib.dll.ratsdf_debug_tail_stamps
    ts.argtypes = [ctypes.c_void_p]
    ts(eng._h)   # (reads and resets)
reps = 20
t0 = time.perf_counter()
for rep in range(reps):
    eng.integrate_device_batch(batch)
eng.synchronize()
dt = time.perf_counter() - t0
print("frames/s", round(reps * len(frames) / dt, 1), "us/frame", round(dt / (reps * len(frames)) * 1e6, 2),
      eng.pipeline_counters(), flush=True)
if stamps:
    ts(eng._h)
