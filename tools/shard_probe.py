#!/usr/bin/env python3
"""GPU box: one block-ownership shard of a stream, alone on the GPU (BASELINE config 4's split, one rank at a time):
every rank integrates the SAME frames and keeps the blocks owner(block) = floormod(block.x >> 2, N) == r.
Prints the shard's frame period; under rocprofv3 --kernel-trace the per-kernel times come from the trace
(tools/shard_table.sh).   usage: tools/shard_probe.py vga|hd N r"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import torch, ratsdf
from ratsdf import synthetic
cfg, N, r = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cam, vs, half = ("l515_720p", 0.002, 10) if cfg == "hd" else ("scannet", 0.005, 30)
dev = torch.device("cuda", 0)
fr = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(half)]
frames = fr + fr[::-1]
H, W = frames[0]["depth"].shape
d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
kw = dict(shard_rank=r, shard_count=N, shard_slab_bits=2) if N > 1 else {}
eng = ratsdf.TSDFGrid(vs, 6 * vs, **kw)
batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d],
                       [x["ht"].data_ptr() for x in d], [x["lt"].data_ptr() for x in d], H, W, 4.0,
                       [f["intrinsics"] for f in frames], [f["pose"] for f in frames])
for _ in range(3):
    eng.integrate_device_batch(batch)
eng.synchronize()
eng.totals(reset=True)
reps = 6 if cfg == "hd" else 10
t0 = time.perf_counter()
for _ in range(reps):
    eng.integrate_device_batch(batch)
eng.synchronize()
dt = time.perf_counter() - t0
t = eng.totals()
print(f"SHARD {cfg} N={N} r={r} us_per_frame={dt / (reps * len(frames)) * 1e6:.2f} "
      f"visible={t['visible_blocks'] / t['frames']:.0f} updated={t['updated_voxels'] / t['frames']:.0f} "
      f"allocated={t['allocated_blocks'] / t['frames']:.1f} active={eng.last_frame_stats()['active_blocks']}", flush=True)
