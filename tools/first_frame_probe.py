#!/usr/bin/env python3
"""GPU box, diagnostic (stamps) build: the first frame of a view (every block new -> the serial role's
general path) at 1280x720 / 2 mm; prints the role's phase stamps and the frame's wall time.
    python tools/first_frame_probe.py [vga]"""
import ctypes, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("RATSDF_LIB", str(ROOT / "ra-slam_amd/csrc/build/libratsdf_stamps.so"))
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
vga = len(sys.argv) > 1 and sys.argv[1] == "vga"
cam, vs = ("scannet", 0.005) if vga else ("l515_720p", 0.002)
dev = torch.device("cuda", 0)
frames = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in (0, 1, 2, 40)]
H, W = frames[0]["depth"].shape
dd = [[torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")] for f in frames]
for rep in range(2):
    eng = ratsdf.TSDFGrid(vs, 6 * vs)
    eng.synchronize()
    stamps = "stamps" in os.environ["RATSDF_LIB"]
    if stamps:
        ws = eng.lib.dll.ratsdf_debug_wave_stamps
        ws.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for f, d in zip(frames, dd):
        if stamps and f is frames[0]:
            ws(eng._h, 1)
        t0 = time.perf_counter()
        eng.integrate_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), H, W, 4.0,
                             f["intrinsics"], f["pose"])
        eng.synchronize()
        dt = time.perf_counter() - t0
        if stamps and f is frames[0]:
            ws(eng._h, 0)   # prints the wave stamps of the launch just finished
            fn0 = eng.lib.dll.ratsdf_debug_stamps
            fn0.argtypes = [ctypes.c_void_p]
            fn0(eng._h)     # ... and the serial role's phases of this frame alone
        print(f"rep {rep}: frame wall {dt * 1e6:8.1f} us  {eng.last_frame_stats()}", flush=True)
    if "stamps" in os.environ["RATSDF_LIB"]:
        fn = eng.lib.dll.ratsdf_debug_stamps
        fn.argtypes = [ctypes.c_void_p]
        fn(eng._h)
    eng.close()
