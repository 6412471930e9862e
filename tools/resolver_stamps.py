#!/usr/bin/env python3
"""GPU box, diagnostic build: phases of the chained-bucket resolver over the 1280x720 / 2 mm non-repeating
pass (where a map past 100 k blocks files 100-250 chained requests per frame)."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["RATSDF_LIB"] = str(ROOT / "ra-slam_amd/csrc/build/libratsdf_stamps.so")
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 360
dev = torch.device("cuda", 0)
eng = ratsdf.TSDFGrid(0.002, 0.012)
for i in range(n):
    f = synthetic.frame("room", i, cam="l515_720p", noise=True, holes=True)
    d = [torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")]
    H, W = f["depth"].shape
    eng.integrate_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), H, W, 4.0, f["intrinsics"], f["pose"])
    eng.synchronize()
fn = eng.lib.dll.ratsdf_debug_stamps
fn.argtypes = [ctypes.c_void_p]
fn(eng._h)
