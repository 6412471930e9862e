#!/usr/bin/env python3
"""GPU box, diagnostic build: timeline of k_integrate inside a BATCH (look-ahead candidate pass hosted), from
in-kernel wall-clock stamps: when the serial role published, when the first passes / the waves' last passes
ended.  The buffer keeps the last two frames apart (frame parity): the last frame of a batch hosts no
look-ahead, the one before it does.   usage: tools/timeline_batch.py [vga|hd]"""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["RATSDF_LIB"] = str(ROOT / "ra-slam_amd/csrc/build/libratsdf_stamps.so")
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
hd = len(sys.argv) > 1 and sys.argv[1] == "hd"
cam, vs = ("l515_720p", 0.002) if hd else ("scannet", 0.005)
dev = torch.device("cuda", 0)
frames = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(20)]
frames = frames + frames[::-1]
H, W = frames[0]["depth"].shape
d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
eng = ratsdf.TSDFGrid(vs, 6 * vs)
def batch(lo, hi):
    idx = list(range(lo, hi))
    return eng.make_batch([d[i]["rgb"].data_ptr() for i in idx], [d[i]["depth"].data_ptr() for i in idx],
                          [d[i]["ht"].data_ptr() for i in idx], [d[i]["lt"].data_ptr() for i in idx], H, W, 4.0,
                          [frames[i]["intrinsics"] for i in idx], [frames[i]["pose"] for i in idx])
for rep in range(3):
    eng.integrate_device_batch(batch(0, 40))
eng.synchronize()
ws = eng.lib.dll.ratsdf_debug_wave_stamps
ws.argtypes = [ctypes.c_void_p, ctypes.c_int]
ws(eng._h, 1)
eng.integrate_device_batch(batch(12, 14))  # two frames: the first hosts the look-ahead of the second
eng.synchronize()
for par in (0, 1):
    print(f"--- frame parity {par}", flush=True)
    ws(eng._h, -1 - par)
