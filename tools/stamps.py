#!/usr/bin/env python3
"""Runs a short stream on the diagnostic (stamps) build and prints per-phase cycles (GPU box)."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["RATSDF_LIB"] = str(ROOT / "ra-slam_amd/csrc/build/libratsdf_stamps.so")
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
dev = torch.device("cuda", 0)
frames = [synthetic.frame("room", i, noise=True, holes=True) for i in range(30)]
frames = frames + frames[::-1]
H, W = frames[0]["depth"].shape
dd = [(torch.from_numpy(f["rgb"]).to(dev), torch.from_numpy(f["depth"]).to(dev),
       torch.from_numpy(f["ht"]).to(dev), torch.from_numpy(f["lt"]).to(dev)) for f in frames]
eng = ratsdf.TSDFGrid(0.005, 0.03)
for rep in range(4):
    for f, d in zip(frames, dd):
        eng.integrate_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), H, W,
                             4.0, f["intrinsics"], f["pose"])
eng.synchronize()
ws = eng.lib.dll.ratsdf_debug_wave_stamps
ws.argtypes = [ctypes.c_void_p, ctypes.c_int]
import os
ws(eng._h, 1)
f, d = frames[10], dd[10]
eng.integrate_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), H, W, 4.0, f['intrinsics'], f['pose'])
eng.synchronize()
ws(eng._h, 0)
print(eng.last_frame_stats())
fn = eng.lib.dll.ratsdf_debug_stamps
fn.argtypes = [ctypes.c_void_p]
fn(eng._h)
