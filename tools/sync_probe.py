#!/usr/bin/env python3
"""What a synchronous caller pays (GPU box): ratsdf_synchronize on an idle engine, and ratsdf_integrate_device +
ratsdf_synchronize per frame (TSDFGrid::Integrate's convention, voxel_tsdf.cu:376-452) on the bench stream's saturated
map.  No PyTorch in the process (ratsdf.devmem).  RATSDF_LIB selects the library (same-box A/B).
tools/sync_probe.py [frames]"""
import sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np
import ratsdf
from ratsdf import devmem, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 360
T = devmem.TorchLike()
frames = synthetic.stream("room", 45, cam="scannet", noise=True, holes=True)
frames = frames + frames[::-1]
H, W = frames[0]["depth"].shape
d = [{k: T.from_numpy(f[k]).to(0) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
intr = [ratsdf.Intrinsics(*f["intrinsics"]) for f in frames]
pose = [ratsdf.Pose(*f["pose"]) for f in frames]
eng = ratsdf.TSDFGrid(0.005, 0.03)
batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d], [x["ht"].data_ptr() for x in d],
                       [x["lt"].data_ptr() for x in d], H, W, 4.0, intr, pose)
for _ in range(3):
    eng.integrate_device_batch(batch)
eng.synchronize()
idle = []
for _ in range(200):
    t0 = time.perf_counter(); eng.synchronize(); idle.append(time.perf_counter() - t0)
lat = []
t_all = time.perf_counter()
for j in range(n):
    i = j % len(frames)
    t0 = time.perf_counter()
    eng.integrate_device(d[i]["rgb"].data_ptr(), d[i]["depth"].data_ptr(), d[i]["ht"].data_ptr(), d[i]["lt"].data_ptr(), H, W, 4.0,
                         intr[i], pose[i])
    t1 = time.perf_counter()
    eng.synchronize()
    lat.append((time.perf_counter() - t0, t1 - t0))
t_all = time.perf_counter() - t_all
# the same through the batch entry point with batches of ONE frame (one HIP-graph replay per call)
ones = [eng.make_batch([d[i]["rgb"].data_ptr()], [d[i]["depth"].data_ptr()], [d[i]["ht"].data_ptr()], [d[i]["lt"].data_ptr()], H, W, 4.0,
                       [intr[i]], [pose[i]]) for i in range(len(frames))]
for b in ones[:4]:
    eng.integrate_device_batch(b)
eng.synchronize()
lat1 = []
for j in range(n):
    t0 = time.perf_counter()
    eng.integrate_device_batch(ones[j % len(frames)])
    t1 = time.perf_counter()
    eng.synchronize()
    lat1.append((time.perf_counter() - t0, t1 - t0))
tot1 = sorted(x[0] for x in lat1); enq1 = sorted(x[1] for x in lat1)
tot = sorted(x[0] for x in lat); enq = sorted(x[1] for x in lat); idle.sort()
print(json.dumps({"lib": str(ratsdf.library().path),
                  "idle_synchronize_us_p50": round(idle[len(idle) // 2] * 1e6, 1),
                  "frame_sync_us": {"p50": round(tot[len(tot) // 2] * 1e6, 1), "p99": round(tot[int(len(tot) * .99)] * 1e6, 1)},
                  "enqueue_us_p50": round(enq[len(enq) // 2] * 1e6, 1), "frames_per_s": round(n / t_all, 1),
                  "batch_of_one_sync_us": {"p50": round(tot1[len(tot1) // 2] * 1e6, 1), "p99": round(tot1[int(len(tot1) * .99)] * 1e6, 1),
                                           "enqueue_p50": round(enq1[len(enq1) // 2] * 1e6, 1)}}))
