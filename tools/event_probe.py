#!/usr/bin/env python3
"""GPU box: per-frame k_integrate durations from HIP events attached to the dispatches (ratsdf_profile_enable(2)),
by position in the batch -- to compare with rocprofv3's kernel trace of the same launches."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
dev = torch.device("cuda", 0)
half = [synthetic.frame("room", i, noise=True, holes=True) for i in range(45)]
frames = half + half[::-1]
H, W = frames[0]["depth"].shape
d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
eng = ratsdf.TSDFGrid(0.005, 0.03)
batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d],
                       [x["ht"].data_ptr() for x in d], [x["lt"].data_ptr() for x in d], H, W, 4.0,
                       [f["intrinsics"] for f in frames], [f["pose"] for f in frames])
for _ in range(3):
    eng.integrate_device_batch(batch)
eng.synchronize()
for mode, every in (("every frame", True), ("every 4th frame of every 4th batch", False)):
    eng.profile_enable(True, every_frame=every)
    for _ in range(8):
        eng.integrate_device_batch(batch)
    eng.synchronize()
    if every:
        k, p = eng.profile_read_frames()
        k = k.reshape(-1, len(frames))
        print(mode, "mean", round(float(k.mean()), 2), "median", round(float(np.median(k)), 2),
              "first of batch", round(float(k[:, 0].mean()), 2), "last of batch", round(float(k[:, -1].mean()), 2),
              "positions 0,4,8.. mean", round(float(k[:, ::4].mean()), 2), "period median", round(float(np.median(p)), 2))
    else:
        ms, n = eng.profile_read()
        print(mode, "mean", round(ms / max(n, 1) * 1e3, 2), "launches", n)
    eng.profile_enable(False)
