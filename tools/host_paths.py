#!/usr/bin/env python3
"""The host-image entry points, PCIe included, in a process WITHOUT PyTorch (the system ROCm runtime a C / C++ caller
of libratsdf.so links -- bench.py runs this as a child, as it runs the C++ TSDFSystem leg).  Prints one JSON line:
  host_image_path   ratsdf_integrate, one frame of pageable images per call; ratsdf_integrate_batch, 8 pageable frames per call
  pinned_h2d_path   ratsdf_integrate_batch(pinned=1), 32 frames per call from one page-locked arena
(the reference's calling convention hands over cv::Mat images in host memory: modules/tsdf_module.cc:22-37,88-115;
examples/tsdf/offline.cc:169).  Never the headline value."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import numpy as np  # noqa: E402
import ratsdf  # noqa: E402
from ratsdf import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cam", default="scannet")
ap.add_argument("--scene", default="room")
ap.add_argument("--voxel", type=float, default=0.005)
ap.add_argument("--max-depth", type=float, default=4.0)
ap.add_argument("--frames", type=int, default=90, help="frames of the ping-pong stream")
ap.add_argument("--host-frames", type=int, default=60)
ap.add_argument("--device", type=int, default=0)
a = ap.parse_args()
vs = a.voxel
half = [synthetic.frame(a.scene, i, cam=a.cam, noise=True, holes=True) for i in range((a.frames + 1) // 2)]
frames = (half + half[::-1])[:a.frames]

hp = ratsdf.TSDFGrid(vs, 6 * vs, device=a.device)
nh = min(a.host_frames, len(frames))
for f in frames[:4]:
    hp.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], a.max_depth, f["intrinsics"], f["pose"])
hp.synchronize()
nh_total = 0
th = time.perf_counter()
for _ in range(5):   # (the calls do not wait for their frames: the timed region ends with a synchronisation)
    for f in frames[:nh]:
        hp.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], a.max_depth, f["intrinsics"], f["pose"])
    nh_total += nh
hp.synchronize()
th = time.perf_counter() - th
nh = nh_total
hp.integrate_batch(frames[:8], a.max_depth)
hp.synchronize()
nb = 0
tb = time.perf_counter()
for _ in range(4):
    for c0 in range(0, len(frames) - 7, 8):
        hp.integrate_batch(frames[c0:c0 + 8], a.max_depth)
        nb += 8
hp.synchronize()   # (the calls return when the images are staged, not when the frames are integrated)
tb = time.perf_counter() - tb
bytes_per_frame = sum(frames[0][k].nbytes for k in ("rgb", "depth", "ht", "lt"))
host_path = dict(frames_per_s=round(nh / th, 1), frames=nh, batched_frames_per_s=round(nb / tb, 1),
                 batched_frames=nb, batched_h2d_gbps=round(nb * bytes_per_frame / tb / 1e9, 1),
                 note="ratsdf_integrate with pageable host images, one call per frame (the calling "
                      "convention of examples/tsdf/offline.cc:169): staging copy into the engine's "
                      "page-locked ring (4.6 MB/frame, 4 threads), H2D on a copy stream, frame enqueued; "
                      "one synchronisation at the end of the timed region; batched = "
                      "ratsdf_integrate_batch, 8 frames per call from pageable memory")
# page-locked copies of the stream's frames: ONE arena, a block of 16 bytes per pixel per frame, the
# frame's images side by side in it as depth | ht | lt | rgb -- the order and stride of the engine's staging
# ring, so a frame goes up as one copy and neighbouring frames up to four per copy (include/ratsdf.h,
# ratsdf_integrate_batch; ratsdf::TSDFSystem's queue lays its frames out the same way)
pin = []
npx = frames[0]["depth"].size
arena = hp.host_alloc((len(frames) * npx * 16,), np.uint8)
for i, f in enumerate(frames):
    blk = arena[i * npx * 16:(i + 1) * npx * 16]
    g = dict(f)
    g["depth"] = blk[:npx * 4].view(np.float32).reshape(f["depth"].shape)
    g["ht"] = blk[npx * 4:npx * 8].view(np.float32).reshape(f["ht"].shape)
    g["lt"] = blk[npx * 8:npx * 12].view(np.float32).reshape(f["lt"].shape)
    g["rgb"] = blk[npx * 12:npx * 15].reshape(f["rgb"].shape)
    for k in ("rgb", "depth", "ht", "lt"):
        g[k][...] = f[k]
    pin.append(g)
C = 32
chunks = [pin[c0:c0 + C] for c0 in range(0, len(pin) - C + 1, C)] or [pin]
calls = [hp.make_host_batch(ch, a.max_depth, pinned=True) for ch in chunks]   # pointer tables built once
hp.integrate_host_batch(calls[0])
hp.synchronize()
npin = 0
tp = time.perf_counter()
while npin < 2000:
    for ch, call in zip(chunks, calls):
        hp.integrate_host_batch(call)
        npin += len(ch)
hp.synchronize()   # (a call returns when its images have been uploaded)
tp = time.perf_counter() - tp
pinned_path = dict(frames_per_s=round(npin / tp, 1), frames=npin,
                   h2d_gbps=round(npin * bytes_per_frame / tp / 1e9, 1), link_gbps_spec=63.0,
                   note=f"ratsdf_integrate_batch(pinned=1), {len(chunks[0])} frames per call from one "
                        "ratsdf_host_alloc arena (a 16 B/pixel block per frame, depth | ht | lt | rgb): up to "
                        "4 neighbouring frames per copy, on the engine's two copy streams, up to 15 frames "
                        "ahead of the integration; a call returns when its uploads are done, one "
                        "synchronisation at the end of the timed region")
hp.host_free(arena)
hp.close()


print(json.dumps(dict(host_image_path=host_path, pinned_h2d_path=pinned_path)))
