#!/usr/bin/env python3
"""GPU box: what the config-4 data path costs a rank BESIDES the link -- the same stream integrated (a) as bench.py's
single-GPU loop does (one ratsdf_integrate_device_batch per 90-frame step) and (b) through ratsdf.framecast on the
device path with a ONE-rank RCCL group (chunks of C frames, ring of R buffers, a broadcast per chunk on the side stream,
events both ways, headers through page-locked memory, one HIP-graph replay per chunk).  One rank: no subvolumes, no
link -- the difference is the chunking and the event traffic.   usage: tools/framecast_probe.py [vga|hd] [chunk] [ring]"""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
torch.cuda.init()
import ratsdf
from ratsdf import framecast, synthetic
hd = len(sys.argv) > 1 and sys.argv[1] == "hd"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 15
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cam, vs, half = ("l515_720p", 0.002, 15) if hd else ("scannet", 0.005, 45)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
fr = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(half)]
frames = fr + fr[::-1]
B = len(frames)
H, W = frames[0]["depth"].shape
md = 4.0
steps = 20 if not hd else 12


def direct():
    eng = ratsdf.TSDFGrid(vs, 6 * vs)
    d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
    batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d],
                           [x["ht"].data_ptr() for x in d], [x["lt"].data_ptr() for x in d], H, W, md,
                           [f["intrinsics"] for f in frames], [f["pose"] for f in frames])
    for _ in range(3):
        eng.integrate_device_batch(batch)
    eng.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.integrate_device_batch(batch)
        eng.synchronize()
        best = max(best, steps * B / (time.perf_counter() - t0))
    eng.close()
    return best


def cast():
    eng = ratsdf.TSDFGrid(vs, 6 * vs)
    ext = torch.cuda.ExternalStream(eng.stream(), device=dev)
    n_chunks = B // C
    packed = [torch.from_numpy(framecast.pack_chunk(frames[c * C:(c + 1) * C], md, H, W, C, first_frame_no=c * C)).to(dev)
              for c in range(n_chunks)]
    fc = framecast.FrameCaster(H, W, C, ring=R, src=0, device=dev)

    def run(nsteps):
        total, posted = nsteps * n_chunks, 0
        for k in range(total):
            while posted < total and fc.can_post():
                fc.post(packed[posted % n_chunks])
                posted += 1
            ch = fc.take(ext)
            framecast.integrate_chunk(eng, ch)
            fc.done(ch, ext)
    run(3)
    eng.synchronize()
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        run(steps)
        eng.synchronize()
        torch.cuda.synchronize()
        best = max(best, steps * n_chunks * C / (time.perf_counter() - t0))
    eng.close()
    return best


a, b = direct(), cast()
print(f"{W}x{H} / {vs * 1e3:g} mm, one rank: direct {a:.1f} frames/s; through framecast (chunks of {C}, ring {R}, RCCL broadcast "
      f"of {framecast.chunk_bytes(H, W, C) / 1e6:.1f} MB per chunk in a group of one) {b:.1f} frames/s = {b / a:.3f} x")
dist.destroy_process_group()
