#!/usr/bin/env python3
"""k_raycast wall time (HIP events around ratsdf_raycast_device, images stay on the device) on two maps: the bench
'room' (a closed room: every ray ends on a wall) and 'sphere' (an object in empty space: most rays hit nothing and run
their full length).  RATSDF_LIB selects the library.  tools/raycast_ab.py"""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np, torch
import ratsdf
from ratsdf import synthetic

vs, md = 0.005, 4.0
out = {"lib": str(ratsdf.library().path)}
for scene in ("room", "sphere"):
    gpu = ratsdf.TSDFGrid(vs, 6 * vs)
    frames = [synthetic.frame(scene, i, noise=True, holes=True) for i in range(45)]
    for f in frames:
        gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
    H, W = frames[0]["depth"].shape
    rgba = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    normal = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    ext = torch.cuda.ExternalStream(gpu.stream())
    res = {}
    for name, f in (("view 20", frames[20]), ("view 44", frames[44])):
        K, T = ratsdf.Intrinsics(*f["intrinsics"]), ratsdf.Pose(*f["pose"])
        times = []
        for rep in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(ext)
            gpu.raycast_device(K, H, W, T, 2 * md, rgba.data_ptr(), normal.data_ptr())
            b.record(ext)
            gpu.synchronize()
            times.append(a.elapsed_time(b) * 1e3)
        times.sort()
        res[name] = {"us_p50": round(times[len(times) // 2], 1), "hit": round(float((rgba[..., 3] == 255).float().mean().item()), 3),
                     "checksum": int(rgba.to(torch.int64).sum().item()) ^ int(normal.to(torch.int64).sum().item())}
    out[scene] = {"blocks": gpu.num_active_blocks(), **res}
    gpu.close()
print(json.dumps(out))
