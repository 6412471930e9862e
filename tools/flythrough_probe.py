#!/usr/bin/env python3
"""GPU box: the non-repeating 1280x720 / 2 mm pass frame by frame (a sync and a statistics read after
every frame): k_integrate time (HIP events attached to the dispatch), visible / allocated / deleted
blocks, chained-bucket requests per frame; then the same views once more on the populated map, and a
least-squares fit of the kernel time against those counts.
usage: tools/flythrough_probe.py [frames] [passes]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np
import torch, ratsdf
from ratsdf import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 360
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cam, vs = ("l515_720p", 0.002)
dev = torch.device("cuda", 0)
eng = ratsdf.TSDFGrid(vs, 6 * vs)
eng.profile_enable(True, every_frame=True)
rows = []
for p in range(passes):
    for i in range(n):
        f = synthetic.frame("room", i, cam=cam, noise=True, holes=True)
        d = [torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")]
        H, W = f["depth"].shape
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.integrate_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), H, W, 4.0, f["intrinsics"], f["pose"])
        eng.synchronize()
        dt = (time.perf_counter() - t0) * 1e6
        s = eng.last_frame_stats()
        rows.append([p, i, dt, s["visible_blocks"], s["allocated_blocks"], s["deleted_blocks"], s["active_blocks"], s["slow_requests"]])
k_us, _ = eng.profile_read_frames()
assert len(k_us) == len(rows), (len(k_us), len(rows))
for r, k in zip(rows, k_us):
    r.append(float(k))
    if r[1] % 20 == 0 or k > 600:
        print("pass %d frame %3d  wall %7.1f us  k_integrate %7.1f us  visible %6d alloc %5d del %5d active %6d slow %5d"
              % (r[0], r[1], r[2], r[8], r[3], r[4], r[5], r[6], r[7]), flush=True)
a = np.array(rows, dtype=np.float64)
for p in range(passes):
    m = (a[:, 0] == p) & (a[:, 1] > 0)
    X = np.stack([np.ones(m.sum()), a[m, 3] / 1e3, a[m, 4] / 1e3, a[m, 5] / 1e3, a[m, 7] / 1e2], axis=1)
    coef, *_ = np.linalg.lstsq(X, a[m, 8], rcond=None)
    res = a[m, 8] - X @ coef
    print("pass %d: k_integrate us ~= %.1f + %.2f /1000 visible + %.1f /1000 allocated + %.1f /1000 deleted + %.1f /100 chained requests"
          "   (rms residual %.1f us; mean %.1f us, mean visible %.0f)" % (p, *coef, np.sqrt((res ** 2).mean()), a[m, 8].mean(), a[m, 3].mean()))
