#!/usr/bin/env python3
"""GPU box: the non-repeating 1280x720 / 2 mm pass frame by frame (a sync and a statistics read after
every frame): wall time, visible / allocated / deleted blocks, chained-bucket requests per frame."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 360
cam, vs = ("l515_720p", 0.002)
dev = torch.device("cuda", 0)
eng = ratsdf.TSDFGrid(vs, 6 * vs)
rows = []
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
    rows.append((i, dt, s["visible_blocks"], s["allocated_blocks"], s["deleted_blocks"], s["active_blocks"], s["slow_requests"]))
    if i % 20 == 0 or dt > 1000:
        print("frame %3d  %8.1f us  visible %6d alloc %5d del %5d active %6d slow %5d" % rows[-1], flush=True)
