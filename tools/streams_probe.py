#!/usr/bin/env python3
"""GPU box: (a) host cost of enqueueing frames on an idle queue, (b) S independent engines on S HIP
streams driven by S host threads (the cheap multi-stream experiment of VERDICT r1 item 3).

    python tools/streams_probe.py [--streams 1,2,4,8] [--steps 20] [--config vga5mm|hd2mm]
"""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))

import torch  # noqa: E402

import ratsdf  # noqa: E402
from ratsdf import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--config", default="vga5mm")
    ap.add_argument("--only-group", action="store_true")
    a = ap.parse_args()
    cam, vs = ("scannet", 0.005) if a.config == "vga5mm" else ("l515_720p", 0.002)
    dev = torch.device("cuda", 0)
    torch.cuda.init()
    smax = max(int(s) for s in a.streams.split(","))
    half = a.frames // 2

    def make(s):
        fr = [synthetic.frame("room", 45 * s + i, cam=cam, noise=True, holes=True) for i in range(half)]
        fr = fr + fr[::-1]
        H, W = fr[0]["depth"].shape
        t = {k: [torch.from_numpy(f[k]).to(dev) for f in fr] for k in ("rgb", "depth", "ht", "lt")}
        eng = ratsdf.TSDFGrid(vs, 6 * vs)
        batch = eng.make_batch([x.data_ptr() for x in t["rgb"]], [x.data_ptr() for x in t["depth"]],
                               [x.data_ptr() for x in t["ht"]], [x.data_ptr() for x in t["lt"]], H, W,
                               4.0, [f["intrinsics"] for f in fr], [f["pose"] for f in fr])
        return eng, batch, t, (H, W)

    engs = [make(s) for s in range(smax)]
    torch.cuda.synchronize()
    for e, b, _, _ in engs:      # build the maps
        for _ in range(3):
            e.integrate_device_batch(b)
        e.synchronize()

    # (a) enqueue cost on an idle queue: short bursts, nothing queued before
    e, b, t, (H, W) = engs[0]
    for burst in (() if a.only_group else (4, 16, 60)):
        small = e.make_batch([x.data_ptr() for x in t["rgb"][:burst]], [x.data_ptr() for x in t["depth"][:burst]],
                             [x.data_ptr() for x in t["ht"][:burst]], [x.data_ptr() for x in t["lt"][:burst]],
                             H, W, 4.0, b[8][:burst], b[9][:burst])
        best = 1e9
        tot = 0.0
        for _ in range(5):
            e.synchronize()
            t0 = time.perf_counter()
            e.integrate_device_batch(small)
            t1 = time.perf_counter()
            e.synchronize()
            t2 = time.perf_counter()
            best = min(best, (t1 - t0) / burst)
            tot = (t2 - t0) / burst
        print(json.dumps(dict(probe="enqueue", burst=burst, host_us_per_frame=round(best * 1e6, 2),
                              total_us_per_frame=round(tot * 1e6, 2))), flush=True)

    # (b) S engines, S threads
    for S in ([] if a.only_group else [int(s) for s in a.streams.split(",")]):
        sel = engs[:S]
        for e, _, _, _ in sel:
            e.synchronize()
            e.totals(reset=True)
        bar = threading.Barrier(S + 1)

        def work(e, b):
            bar.wait()
            for _ in range(a.steps):
                e.integrate_device_batch(b)
            e.synchronize()

        th = [threading.Thread(target=work, args=(e, b)) for e, b, _, _ in sel]
        for x in th:
            x.start()
        bar.wait()
        t0 = time.perf_counter()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        nfr = S * a.steps * a.frames
        alg = 0.0
        for e, _, _, (H, W) in sel:
            tt = e.totals()
            alg += 15.0 * W * H * tt["frames"] + 12.0 * tt["visible_blocks"] + 24.0 * tt["updated_voxels"]
        print(json.dumps(dict(probe="streams", S=S, frames_per_s=round(nfr / dt, 1),
                              alg_gbps=round(alg / dt / 1e9, 1), frac_of_8TBs=round(alg / dt / 8e12, 4))),
              flush=True)
    # (c) the same S streams through ONE launch triple per frame step (ratsdf_group_*)
    for S in [int(s) for s in a.streams.split(",")]:
        sel = engs[:S]
        grp = ratsdf.Group([e for e, _, _, _ in sel])
        H, W = sel[0][3]
        fr_all = [[synthetic.frame("room", 45 * s + i, cam=cam, noise=True, holes=True) for i in range(half)]
                  for s in range(S)]
        fr_all = [fr + fr[::-1] for fr in fr_all]
        rows = lambda key: [[sel[s][2][key][f].data_ptr() for s in range(S)] for f in range(a.frames)]
        gb = grp.make_batch(rows("rgb"), rows("depth"), rows("ht"), rows("lt"), H, W, 4.0,
                            [[fr_all[s][f]["intrinsics"] for s in range(S)] for f in range(a.frames)],
                            [[fr_all[s][f]["pose"] for s in range(S)] for f in range(a.frames)])
        grp.integrate_device_batch(gb)
        grp.synchronize()
        for e, _, _, _ in sel:
            e.totals(reset=True)
        grp.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            grp.integrate_device_batch(gb)
        t1 = time.perf_counter()
        grp.synchronize()
        dt = time.perf_counter() - t0
        kms, kn = grp.profile_read()
        grp.profile_enable(False)
        alg = 0.0
        for e, _, _, (H, W) in sel:
            tt = e.totals()
            alg += 15.0 * W * H * tt["frames"] + 12.0 * tt["visible_blocks"] + 24.0 * tt["updated_voxels"]
        nfr = S * a.steps * a.frames
        k_us = kms / max(kn, 1) * 1e3
        print(json.dumps(dict(probe="group", S=S, frames_per_s=round(nfr / dt, 1),
                              us_per_step=round(dt / (a.steps * a.frames) * 1e6, 2),
                              host_enqueue_frac=round((t1 - t0) / dt, 3),
                              alg_gbps=round(alg / dt / 1e9, 1), frac_of_8TBs=round(alg / dt / 8e12, 4),
                              k_integrate_us=round(k_us, 2),
                              k_integrate_frac=round(alg / nfr * S / (k_us * 1e-6) / 8e12, 4) if kn else None)),
              flush=True)
        grp.close()
    for e, _, _, _ in engs:
        e.close()


if __name__ == "__main__":
    main()
