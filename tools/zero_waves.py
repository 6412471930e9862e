#!/usr/bin/env python3
"""How many of k_integrate's update waves change no voxel at all?  (GPU box, diagnostic build, RATSDF_DEBUG=30)

VERDICT r4 item 5a: a third (5 mm) / a fifth (2 mm) of the voxel slots of visible blocks are not updated, yet every
slot is loaded, projected and written back at line granularity.  A wave = 128 consecutive voxels = two z-slices of
a block; one that ends with no update could have been skipped IF a cheap test knew beforehand.  This counts them on
the bench streams (steady state: the ping-pong sweep, second pass on)."""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["RATSDF_LIB"] = str(ROOT / "ra-slam_amd/csrc/build/libratsdf_stamps.so")
os.environ["RATSDF_DEBUG"] = "30"
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ratsdf  # noqa: E402
from ratsdf import synthetic  # noqa: E402

dev = torch.device("cuda", 0)
for name, cam, vs, half in (("640x480 / 5 mm", "scannet", 0.005, 45), ("1280x720 / 2 mm", "l515_720p", 0.002, 10),
                            ("1280x720 / 2 mm, 120-frame sweep", "l515_720p", 0.002, 120)):
    fr = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(half)]
    frames = fr + fr[::-1]
    H, W = frames[0]["depth"].shape
    dd = [tuple(torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")) for f in fr]
    dd = dd + dd[::-1]
    eng = ratsdf.TSDFGrid(vs, 6 * vs)
    cnt = eng.lib.dll.ratsdf_debug_counters
    cnt.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    out = (ctypes.c_ulonglong * 8)()
    batch = eng.make_batch([d[0].data_ptr() for d in dd], [d[1].data_ptr() for d in dd], [d[2].data_ptr() for d in dd],
                           [d[3].data_ptr() for d in dd], H, W, 4.0, [f["intrinsics"] for f in frames],
                           [f["pose"] for f in frames])
    eng.integrate_device_batch(batch)      # first pass: the map is built
    eng.synchronize()
    cnt(eng._h, out)
    eng.totals(reset=True)
    eng.integrate_device_batch(batch)      # steady state
    eng.synchronize()
    cnt(eng._h, out)
    t = eng.totals()
    zw, w, zb, b = out[0], out[1], out[2], out[3]
    print(f"{name}: {len(frames)} frames, {b / len(frames):.0f} blocks / frame, updated voxels / slot "
          f"{t['updated_voxels'] / max(b * 512, 1):.3f}; waves without an update {zw} of {w} = {zw / max(w, 1):.3f}; "
          f"blocks without an update {zb} of {b} = {zb / max(b, 1):.3f}")
    eng.close()
    del dd
    torch.cuda.empty_cache()
