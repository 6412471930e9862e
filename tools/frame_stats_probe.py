#!/usr/bin/env python3
"""GPU box: per-frame statistics of the bench stream in its steady state (which frames leave the serial
role's fast path: more than 2048 requests or deletes, or any chained-bucket request).

    python tools/frame_stats_probe.py [--config vga5mm|hd2mm|bigmap]
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ratsdf  # noqa: E402
from bench import make_stream  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="vga5mm")
    a = ap.parse_args()
    cam, vs, B = {"vga5mm": ("scannet", 0.005, 90), "hd2mm": ("l515_720p", 0.002, 30),
                  "bigmap": ("l515_720p", 0.002, 240)}[a.config]
    frames = make_stream("room", cam, (B + 1) // 2, phase=0)[:B]
    dev = torch.device("cuda", 0)
    t = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
    H, W = frames[0]["depth"].shape
    eng = ratsdf.TSDFGrid(vs, 6 * vs)
    rows = []
    for rep in range(3):
        for f, d in zip(frames, t):
            eng.integrate_device(d["rgb"].data_ptr(), d["depth"].data_ptr(), d["ht"].data_ptr(),
                                 d["lt"].data_ptr(), H, W, 4.0, f["intrinsics"], f["pose"])
            s = eng.last_frame_stats()
            if rep == 2:
                rows.append((s["allocated_blocks"], s["deleted_blocks"], s["slow_requests"], s["visible_blocks"]))
    r = np.array(rows)
    q = lambda c: [int(x) for x in np.percentile(r[:, c], [0, 50, 90, 100])]
    print(json.dumps(dict(config=a.config, frames=len(rows), allocated_min_med_p90_max=q(0),
                          deleted_min_med_p90_max=q(1), slow_frames=int((r[:, 2] > 0).sum()),
                          slow_max=int(r[:, 2].max()), visible_med=q(3)[1])))
    eng.close()


if __name__ == "__main__":
    main()
