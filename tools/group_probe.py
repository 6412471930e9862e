#!/usr/bin/env python3
"""GPU box: frames/s of S streams through one launch pair per frame step (ratsdf_group_*), no parity legs.
usage: tools/group_probe.py [S]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
streams = []
for s in range(S):
    fr = [synthetic.frame("room", 45 * s + i, noise=True, holes=True) for i in range(30)]
    streams.append(fr + fr[::-1])
H, W = streams[0][0]["depth"].shape
dt_ = [[{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in fr] for fr in streams]
engines = [ratsdf.TSDFGrid(0.005, 0.03) for _ in range(S)]
grp = ratsdf.Group(engines)
n = len(streams[0])
rows = lambda key: [[dt_[s][f][key].data_ptr() for s in range(S)] for f in range(n)]
gb = grp.make_batch(rows("rgb"), rows("depth"), rows("ht"), rows("lt"), H, W, 4.0,
                    [[streams[s][f]["intrinsics"] for s in range(S)] for f in range(n)],
                    [[streams[s][f]["pose"] for s in range(S)] for f in range(n)])
for _ in range(3):
    grp.integrate_device_batch(gb)
grp.synchronize()
for e in engines:
    e.pipeline_counters(reset=True)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    grp.integrate_device_batch(gb)
grp.synchronize()
dt = time.perf_counter() - t0
print("S", S, "frames/s", round(S * reps * n / dt, 1), "us/frame-step", round(dt / (reps * n) * 1e6, 2), engines[0].pipeline_counters())
