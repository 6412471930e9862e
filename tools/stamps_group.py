#!/usr/bin/env python3
"""Diagnostic build, GPU box: per-wave phase stamps of ONE k_integrate launch in three settings --
(a) 640x480 single stream, (b) 1280x720 / 2 mm single stream, (c) member 0 of an S = 4 group of 640x480
streams (k_integrate_g) -- to see where a wave's lifetime differs between them."""
import ctypes, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
os.environ["RATSDF_LIB"] = str(ROOT / "ra-slam_amd/csrc/build/libratsdf_stamps.so")
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
dev = torch.device("cuda", 0)

def stream(cam, s, n=20):
    fr = [synthetic.frame("room", 45 * s + i, cam=cam, noise=True, holes=True) for i in range(n // 2)]
    fr = fr + fr[::-1]
    t = {k: [torch.from_numpy(f[k]).to(dev) for f in fr] for k in ("rgb", "depth", "ht", "lt")}
    return fr, t

def ws_fn(e):
    f = e.lib.dll.ratsdf_debug_wave_stamps
    f.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return f

def single(cam, vs, tag):
    fr, t = stream(cam, 0)
    H, W = fr[0]["depth"].shape
    e = ratsdf.TSDFGrid(vs, 6 * vs)
    b = e.make_batch([x.data_ptr() for x in t["rgb"]], [x.data_ptr() for x in t["depth"]], [x.data_ptr() for x in t["ht"]],
                     [x.data_ptr() for x in t["lt"]], H, W, 4.0, [f["intrinsics"] for f in fr], [f["pose"] for f in fr])
    for _ in range(3):
        e.integrate_device_batch(b)
    e.synchronize()
    print("==", tag, flush=True)
    ws_fn(e)(e._h, 1)
    e.integrate_device_batch(b)      # stamps of the LAST launch of the batch survive
    e.synchronize()
    ws_fn(e)(e._h, 0)
    print(e.last_frame_stats(), flush=True)
    e.close()

def group(S):
    cam, vs = "scannet", 0.005
    ss = [stream(cam, s) for s in range(S)]
    H, W = ss[0][0][0]["depth"].shape
    engs = [ratsdf.TSDFGrid(vs, 6 * vs) for _ in range(S)]
    g = ratsdf.Group(engs)
    n = len(ss[0][0])
    rows = lambda key: [[ss[s][1][key][f].data_ptr() for s in range(S)] for f in range(n)]
    gb = g.make_batch(rows("rgb"), rows("depth"), rows("ht"), rows("lt"), H, W, 4.0,
                      [[ss[s][0][f]["intrinsics"] for s in range(S)] for f in range(n)],
                      [[ss[s][0][f]["pose"] for s in range(S)] for f in range(n)])
    for _ in range(3):
        g.integrate_device_batch(gb)
    g.synchronize()
    print(f"== group S={S}, member 0", flush=True)
    ws_fn(engs[0])(engs[0]._h, 1)
    g.integrate_device_batch(gb)
    g.synchronize()
    ws_fn(engs[0])(engs[0]._h, 0)
    print(engs[0].last_frame_stats(), flush=True)
    g.close()
    for e in engs:
        e.close()

single("scannet", 0.005, "640x480 / 5 mm single")
single("l515_720p", 0.002, "1280x720 / 2 mm single")
group(4)
