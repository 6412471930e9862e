#!/usr/bin/env python3
"""GPU box: the headline loop (90-frame ping-pong at 640x480 / 5 mm, ratsdf_integrate_device_batch, inputs resident in
HBM) in a process that does NOT load PyTorch: device memory through the system ROCm runtime (ctypes on libamdhip64).
A process that imports torch runs libratsdf.so on the HIP runtime bundled with the PyTorch wheel; this is the same loop
on the runtime a C / C++ caller links (bench.py itself does so at N = 1 since round 5: ratsdf.devmem).  Best of 5
repetitions, no event sampling: a ceiling, not comparable with bench.py's median.
   usage: tools/notorch_probe.py [vga|hd]"""
import ctypes as C
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np
import ratsdf
from ratsdf import synthetic
hd = len(sys.argv) > 1 and sys.argv[1] == "hd"
nosem = "nosem" in sys.argv[1:]   # TSDF-only frames (ht / lt NULL): BASELINE configs[1]'s shape
cam, vs, n = ("l515_720p", 0.002, 15) if hd else ("scannet", 0.005, 45)
half = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(n)]
frames = half + half[::-1]
H, W = frames[0]["depth"].shape
eng = ratsdf.TSDFGrid(vs, 6 * vs)          # loads libratsdf.so -> the system libamdhip64
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]


def up(a):
    a = np.ascontiguousarray(a)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
    assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
    return p.value


d = [{k: up(f[k]) for k in ("rgb", "depth", "ht", "lt")} for f in half]
d = d + d[::-1]
batch = eng.make_batch([x["rgb"] for x in d], [x["depth"] for x in d], None if nosem else [x["ht"] for x in d],
                       None if nosem else [x["lt"] for x in d],
                       H, W, 4.0, [f["intrinsics"] for f in frames], [f["pose"] for f in frames])
for _ in range(3):
    eng.integrate_device_batch(batch)
eng.synchronize()
best = 0.0
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(40):
        eng.integrate_device_batch(batch)
    eng.synchronize()
    best = max(best, 40 * len(frames) / (time.perf_counter() - t0))
print(f"no-torch process, {W}x{H} / {vs * 1e3:g} mm{', TSDF-only' if nosem else ''}: {best:.1f} frames/s (best of 5 x 40 batches of {len(frames)})")
