#!/usr/bin/env python3
"""GPU box, diagnostic (stamps) build: phase cycles of the candidate pass's workgroups.
  python tools/cand_probe.py [single|batch] [hd]
single: one frame per call (k_cand as its own launch: the pass alone on the GPU); batch: the default frame (20 % of
the pass in k_front, 80 % in k_integrate beside the voxel update)."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("RATSDF_LIB", str(ROOT / "ra-slam_amd/csrc/build/libratsdf_stamps.so"))
os.environ.setdefault("RATSDF_GRAPH", "0")
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
mode = sys.argv[1] if len(sys.argv) > 1 else "single"
hd = len(sys.argv) > 2 and sys.argv[2] == "hd"
dev = torch.device("cuda", 0)
if hd:
    half = [synthetic.frame("room", i, cam="l515_720p", noise=True, holes=True) for i in range(10)]
    eng = ratsdf.TSDFGrid(0.002, 0.012)
else:
    half = [synthetic.frame("room", i, noise=True, holes=True) for i in range(30)]
    eng = ratsdf.TSDFGrid(0.005, 0.03)
frames = half + half[::-1]
H, W = frames[0]["depth"].shape
d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d],
                       [x["ht"].data_ptr() for x in d], [x["lt"].data_ptr() for x in d], H, W, 4.0,
                       [f["intrinsics"] for f in frames], [f["pose"] for f in frames])
ts = eng.lib.dll.ratsdf_debug_tail_stamps
ts.argtypes = [ctypes.c_void_p]
def run():
    if mode == "single":
        for f, x in zip(frames, d):
            eng.integrate_device(x["rgb"].data_ptr(), x["depth"].data_ptr(), x["ht"].data_ptr(), x["lt"].data_ptr(),
                                 H, W, 4.0, f["intrinsics"], f["pose"])
    else:
        eng.integrate_device_batch(batch)
    eng.synchronize()
run(); run()
ts(eng._h)   # (clears)
import io
print("--- measured ---", mode, "hd" if hd else "vga", file=sys.stderr)
# the workgroup counters (Ctl::stamps 14..18) accumulate from the start: read the difference over the runs below
fn = eng.lib.dll.ratsdf_debug_tail_stamps
for _ in range(3):
    run()
ts(eng._h)
