#!/usr/bin/env python3
"""bench.py -- depth+semantic frames/s integrated by the MI355X TSDF engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one batch of `--frames-per-step` synthetic 640x480 frames (depth + rgb + ht/lt, 5 mm voxels,
truncation 30 mm, max depth 4 m, ScanNet intrinsics; the "room" stream of ratsdf.synthetic)
integrated through the C ABI (ratsdf_integrate_device_batch) with every input already resident in
HBM; `--reps` timed repetitions of K steps each, `value` = the median.  For N > 1 the driver launches
one process per GPU with torch.distributed.run (started plainly as `python bench.py --gpus N`, the script
starts those N ranks itself, as child processes, before it touches the GPU); each rank integrates its own stream into its own map
(frame-batched config of BASELINE.json) and the ranks all-gather the deltas of their block directories
over RCCL once per step (ratsdf.multi.DirectoryDeltaExchange).  Rank 0 prints ONE JSON line.

The line also carries
  roofline      HBM roofline of the dominant kernel (k_integrate): algorithmic bytes per launch
                (15 W H + 12 V + 24 U) / its average duration measured with HIP events
                attached to the dispatch on the engine's stream inside the timed region; `traffic` from
                the committed PMC passes (`traffic_source`); `launch_also_hosts` = what else rides in
                that launch; `frame_frac` = the whole frame against the same peak
  cpu_baseline  the CPU oracle (multithreaded port; the reference has no CPU path) timed on this
                box's host cores on a bounded prefix of the same stream (rank 0, N = 1 only), after
                asserting that it and the HIP engine produce the same map on that prefix
  secondary     1280x720 / 2 mm / L515 intrinsics (north_star's second stream size), bounded, with its
                own roofline block and parity check
  multi_stream  S streams of this GPU through one launch pair per frame step (ratsdf_group_*);
                `value` stays the single-stream number
  host_image_path / pinned_h2d_path   the PCIe-inclusive entry points (never `value`)
`--config hd2mm | bigmap` run the 1280x720 / 2 mm workload (on a 416 MB map for bigmap) as the main line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=90)
    ap.add_argument("--cam", default="scannet")
    ap.add_argument("--scene", default="room")
    ap.add_argument("--voxel", type=float, default=0.005)
    ap.add_argument("--max-depth", type=float, default=4.0)
    ap.add_argument("--cpu-frames", type=int, default=1080,
                    help="frames of the stream timed on the CPU oracle (0 = skip)")
    ap.add_argument("--config", default="vga5mm", choices=["vga5mm", "hd2mm", "bigmap", "flythrough"],
                    help="vga5mm: BASELINE metric config (640x480, 5 mm); hd2mm: 1280x720, 2 mm "
                         "(BASELINE configs[3] workload, single GPU unless --shard); bigmap: hd2mm on a "
                         "120-degree sweep whose map (> 256 MiB of voxel data, revisited only after a "
                         "whole sweep) cannot stay in the Infinity Cache: the HBM figure of the repo")
    ap.add_argument("--shard", action="store_true",
                    help="N > 1: every rank integrates the SAME stream and owns the blocks "
                         "owner(block) == rank (spatial subvolumes, strong scaling) instead of "
                         "one independent stream per rank")
    ap.add_argument("--bcast-chunk", type=int, default=15,
                    help="--shard: frames per broadcast chunk (one collective and one HIP-graph replay each); the "
                         "largest divisor of --frames-per-step not above this is used")
    ap.add_argument("--bcast-ring", type=int, default=3,
                    help="--shard: chunk buffers per rank; the broadcasts run (ring - 1) chunks ahead of the integration")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event timing of k_integrate")
    ap.add_argument("--with-torch", action="store_true",
                    help="N = 1, diagnostic: device memory through PyTorch as for N > 1 (the process then runs on the "
                         "HIP runtime bundled with the PyTorch wheel) -- for same-box comparisons of the two runtimes")
    ap.add_argument("--host-frames", type=int, default=60,
                    help="frames timed through the host-image entry point, PCIe included (0 = skip)")
    ap.add_argument("--sync-every", type=int, default=0,
                    help="diagnostic: host-synchronise every N frames (0 = only at the end)")
    ap.add_argument("--reps", type=int, default=5,
                    help="timed repetitions of the K steps; value = the median repetition")
    ap.add_argument("--streams", type=int, default=4,
                    help="N = 1: also time S concurrent streams on this GPU through one launch "
                         "triple per frame step (ratsdf_group_*); 0 / 1 = skip")
    ap.add_argument("--no-secondary", action="store_true",
                    help="vga5mm, N = 1: skip the bounded 1280x720 / 2 mm legs (`secondary`: hd2mm, bigmap), the "
                         "non-repeating pass (`flythrough`) and the C++ TSDFSystem leg (`tsdf_system_path`)")
    ap.add_argument("--flythrough-frames", type=int, default=180,
                    help="frames of the non-repeating pass (`flythrough`; --config flythrough: 360 at 1280x720)")
    return ap.parse_args()


def make_stream(scene, cam, nframes, phase):
    """Ping-pong camera sweep so the stream can repeat forever: frames 0..n-1 then n-1..0."""
    from ratsdf import synthetic
    half = [synthetic.frame(scene, phase + i, cam=cam, noise=True, holes=True) for i in
            range(nframes)]
    return half + half[::-1]


def alg_bytes(W, H, tot):
    """15 W H + 12 V + 24 U per frame, summed over the frames counted in `tot`."""
    return 15.0 * W * H * tot["frames"] + 12.0 * tot["visible_blocks"] + 24.0 * tot["updated_voxels"]


def traffic_file(config):
    name = {"vga5mm": "traffic_latest.json", "hd2mm": "traffic_hd2mm.json"}.get(config, f"traffic_{config}.json")
    return ROOT / "profiles" / name


def roofline_block(b_alg_per_launch, k_ms, k_n, config, kernel="k_integrate", engines=1, whole_frame_gbps=None):
    """`frac` prices the dominant launch alone; `frame_frac` the whole frame (all launches and gaps)."""
    if not k_n:
        return None
    k_avg_s = k_ms / k_n / 1e3
    achieved = b_alg_per_launch / k_avg_s / 1e9
    traffic, src = None, None
    tpath = traffic_file(config) if engines == 1 else ROOT / "profiles" / f"traffic_group{engines}.json"
    if tpath.exists():  # PMC passes made on this very workload (tools/traffic.sh, tools/traffic_group.sh)
        try:
            traffic = json.loads(tpath.read_text()).get("k_integrate_bytes_per_launch")
            src = f"profiles/{tpath.name} (static: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
        except Exception:
            traffic = None
    # what binds the launch at this size is vector-ALU issue, not bytes: SQ_INSTS_VALU x 4 cycles /
    # (1 024 SIMDs x duration x clock), from the committed SQ-counter pass of the same build and workload
    issue = None
    ipath = ROOT / "profiles" / ("issue_latest.json" if engines == 1 else f"issue_group{engines}.json")
    if ipath.exists():
        try:
            rec = json.loads(ipath.read_text()).get(config if engines == 1 else "group")
            if rec:
                issue = dict(rec, source=f"profiles/{ipath.name} (static: rocprofv3 --pmc SQ_INSTS_VALU pass + kernel trace, "
                                         "tools/sq.sh + tools/issue_json.py)")
        except Exception:
            issue = None
    # (the counters' bytes over the same launch duration: what the memory system moved, as a rate -- `achieved` / `frac`
    # stay the contract's algorithmic-bytes figures)
    moved = (traffic / k_avg_s / 1e9) if traffic else None
    return dict(bound="hbm", kernel=kernel, achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=src,
                traffic_gbps=(round(moved, 1) if moved else None),
                traffic_frac=(round(moved / HBM_PEAK_GBPS, 4) if moved else None), issue=issue,
                timed_launches=("every 4th frame of every 4th batch of the timed region is launched by itself with "
                                "HIP events attached to the dispatch (k_integrate<2, false>); the other batches are HIP-graph "
                                "replays of the same frames (k_integrate_g<2, false>, one member: the same body)"
                                if engines == 1 else "every 4th launch of the timed region carries HIP events"),
                launch_also_hosts=("the frame's serial allocation-order role (workgroup 0; the commit of the "
                                   "frame's new blocks waits for it inside the launch)" +
                                   ("; 90 % of the NEXT frame's candidate pass (its workgroups come first in "
                                    "the grid; engine default at 640x480)" if config == "vga5mm" and
                                    engines == 1 else "")),
                frame_frac=(round(whole_frame_gbps / HBM_PEAK_GBPS, 4) if whole_frame_gbps else None),
                alg_bytes_per_launch=round(b_alg_per_launch), avg_launch_us=round(k_avg_s * 1e6, 2),
                launches=k_n)


def bench_streams(ratsdf, torch, dev, dev_index, S, scene, cam, vs, md, nfr, steps, reps, cpu_threads=16):
    """S concurrent streams of this GPU through ONE launch pair per frame step (ratsdf_group_*:
    frame-batched integration, BASELINE configs[4] on one device).  Every member's map is exactly
    what the member alone would produce: asserted here against the CPU oracle on the first pass over
    the streams, before anything is timed (and in tests/test_gpu_group.py)."""
    from ratsdf import synthetic
    from oracle_binding import load_oracle
    from parity import assert_maps_equal
    from ratsdf._abi import Engine
    half = nfr // 2
    streams = []
    for s in range(S):
        fr = [synthetic.frame(scene, 45 * s + i, cam=cam, noise=True, holes=True) for i in range(half)]
        streams.append(fr + fr[::-1])
    H, W = streams[0][0]["depth"].shape
    dt_ = [[{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in fr]
           for fr in streams]
    engines = [ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index) for _ in range(S)]
    grp = ratsdf.Group(engines)
    n = len(streams[0])
    rows = lambda key: [[dt_[s][f][key].data_ptr() for s in range(S)] for f in range(n)]
    gb = grp.make_batch(rows("rgb"), rows("depth"), rows("ht"), rows("lt"), H, W, md,
                        [[streams[s][f]["intrinsics"] for s in range(S)] for f in range(n)],
                        [[streams[s][f]["pose"] for s in range(S)] for f in range(n)])
    # first pass: parity of every member against the oracle fed with that member's stream
    grp.integrate_device_batch(gb)
    grp.synchronize()
    worst = dict(tsdf=0.0, prob=0.0)
    for s_ in range(S):
        cpu = Engine(load_oracle(), vs, 6 * vs, threads=cpu_threads)
        for f in streams[s_]:
            cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
        w = assert_maps_equal(engines[s_], cpu)
        worst = {k: max(worst[k], w[k]) for k in worst}
        cpu.close()
    for _ in range(2):
        grp.integrate_device_batch(gb)
    grp.synchronize()
    for e in engines:
        e.totals(reset=True)
    grp.profile_enable(True)
    rep_dt = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            grp.integrate_device_batch(gb)
        grp.synchronize()
        rep_dt.append(time.perf_counter() - t0)
    k_ms, k_n = grp.profile_read()
    grp.profile_enable(False)
    tot = [e.totals() for e in engines]
    b_alg = sum(alg_bytes(W, H, t) for t in tot)
    frames_total = sum(t["frames"] for t in tot)
    dt = sorted(rep_dt)[len(rep_dt) // 2]
    fps = S * steps * n / dt
    out = dict(streams=S, frames_per_s=round(fps, 1), frames_per_s_min_max=[round(S * steps * n / max(rep_dt), 1),
                                                                         round(S * steps * n / min(rep_dt), 1)],
               us_per_frame_step=round(dt / (steps * n) * 1e6, 2), frames_per_step=n, steps=steps, reps=reps,
               alg_gbps_whole_frame=round(b_alg / frames_total * fps / 1e9, 1),
               roofline=roofline_block(b_alg / frames_total * S, k_ms, k_n, "vga5mm", kernel="k_integrate_g",
                                       whole_frame_gbps=b_alg / frames_total * fps / 1e9,
                                       engines=S),
               parity=dict(members=S, frames_each=n, max_abs_tsdf=worst["tsdf"], max_abs_prob=worst["prob"],
                           directory="bit-exact"),
               note="one k_front_g / k_integrate_g launch pair per frame step serves all "
                    "streams (one grid slice per stream); a k_integrate_g launch = S frames")
    grp.close()
    for e in engines:
        e.close()
    return out


def bench_tsdf_only(ratsdf, frames, d_rgb, d_depth, intr, pose, vs, md, dev_index, cpu_threads, steps=20, reps=3):
    """BASELINE configs[1]'s shape: the same stream without semantics (ht / lt NULL = the all-ones images of
    modules/tsdf_module.cc:27-31), inputs resident in HBM.  A map that has never seen ht / lt holds probability 0.5
    everywhere and a TSDF-only frame leaves it there, so the engine neither loads nor stores it for existing blocks
    (FrameParams::segm_live): SURVEY 8d's TSDF-only algorithmic bytes, 7 W H + 12 V + 16 U."""
    from oracle_binding import load_oracle
    from parity import assert_maps_equal
    from ratsdf._abi import Engine
    H, W = frames[0]["depth"].shape
    n = len(frames)
    cpu = Engine(load_oracle(), vs, 6 * vs, threads=cpu_threads)
    chk = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
    n_par = 2 * n
    for j in range(n_par):
        f = frames[j % n]
        cpu.integrate(f["rgb"], f["depth"], None, None, md, f["intrinsics"], f["pose"])
        chk.integrate_device(d_rgb[j % n].data_ptr(), d_depth[j % n].data_ptr(), 0, 0, H, W, md, intr[j % n], pose[j % n])
    worst = assert_maps_equal(chk, cpu)
    chk.close()
    cpu.close()
    eng = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
    batch = eng.make_batch([t.data_ptr() for t in d_rgb], [t.data_ptr() for t in d_depth], None, None, H, W, md, intr, pose)
    for _ in range(3):
        eng.integrate_device_batch(batch)
    eng.synchronize()
    eng.totals(reset=True)
    eng.profile_enable(True)
    rep_dt = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.integrate_device_batch(batch)
        eng.synchronize()
        rep_dt.append(time.perf_counter() - t0)
    k_ms, k_n = eng.profile_read()
    eng.profile_enable(False)
    tot = eng.totals()
    eng.close()
    dt = sorted(rep_dt)[len(rep_dt) // 2]
    fps = steps * n / dt
    fr = max(tot["frames"], 1)
    b_alg = 7.0 * W * H + 12.0 * tot["visible_blocks"] / fr + 16.0 * tot["updated_voxels"] / fr
    k_us = k_ms / max(k_n, 1) * 1e3
    return dict(workload=f"the same stream without ht / lt (TSDF-only), {W}x{H}, voxel {vs * 1e3:g} mm", value=round(fps, 1),
                unit="frames/s", steps=steps, reps=reps,
                roofline=dict(bound="hbm", kernel="k_integrate", alg_bytes_per_launch=round(b_alg),
                              alg_bytes="7 W H + 12 V + 16 U (no probability traffic: SURVEY 8d's TSDF-only figure)",
                              avg_launch_us=round(k_us, 2), achieved=round(b_alg / (k_us * 1e-6) / 1e9, 1) if k_n else None,
                              peak=HBM_PEAK_GBPS, unit="GB/s",
                              frac=round(b_alg / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if k_n else None, launches=k_n),
                parity=dict(frames=n_par, max_abs_tsdf=worst["tsdf"], max_abs_prob=worst["prob"], directory="bit-exact",
                            note="probability exactly 0.5 in both"))


def bench_secondary(ratsdf, torch, dev, dev_index, md, cpu_threads, config="hd2mm"):
    """Bounded 1280x720 / 2 mm / L515 legs (BASELINE configs[3] workload on one GPU; north_star asks
    for both stream sizes): throughput, k_integrate roofline and parity against the CPU oracle.
      hd2mm   20-frame ping-pong: the 100 MB map lives in the 256 MiB Infinity Cache
      bigmap  240-frame sweep (120 degrees there and back): 416 MB of voxel data, every block revisited
              only after the rest of the map has gone through the caches -> the HBM-resident number"""
    from ratsdf import synthetic
    from oracle_binding import load_oracle
    from parity import assert_maps_equal
    from ratsdf._abi import Engine
    vs, cam = 0.002, "l515_720p"
    half, steps, reps, n_par = (10, 8, 3, 20) if config == "hd2mm" else (120, 2, 2, 12)
    fr = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(half)]
    frames = fr + fr[::-1]
    H, W = frames[0]["depth"].shape
    d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in fr]
    d = d + d[::-1]
    intr = [ratsdf.Intrinsics(*f["intrinsics"]) for f in frames]
    pose = [ratsdf.Pose(*f["pose"]) for f in frames]
    # parity on the first frames of the stream
    cpu = Engine(load_oracle(), vs, 6 * vs, threads=cpu_threads)
    chk = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
    t0 = time.perf_counter()
    for f in frames[:n_par]:
        cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
    t_cpu = time.perf_counter() - t0
    for i in range(n_par):
        chk.integrate_device(d[i]["rgb"].data_ptr(), d[i]["depth"].data_ptr(), d[i]["ht"].data_ptr(),
                             d[i]["lt"].data_ptr(), H, W, md, intr[i], pose[i])
    chk.synchronize()
    worst = assert_maps_equal(chk, cpu)
    chk.close()
    cpu.close()
    eng = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
    batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d],
                           [x["ht"].data_ptr() for x in d], [x["lt"].data_ptr() for x in d], H, W, md,
                           intr, pose)
    for _ in range(3 if config == "hd2mm" else 1):
        eng.integrate_device_batch(batch)
    eng.synchronize()
    eng.totals(reset=True)
    eng.profile_enable(True)
    rep_dt = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.integrate_device_batch(batch)
        eng.synchronize()
        rep_dt.append(time.perf_counter() - t0)
    k_ms, k_n = eng.profile_read()
    eng.profile_enable(False)
    tot = eng.totals()
    stats = eng.last_frame_stats()
    dt = sorted(rep_dt)[len(rep_dt) // 2]
    fps = steps * len(frames) / dt
    b_alg = alg_bytes(W, H, tot) / max(tot["frames"], 1)
    out = dict(config=config, workload=f"synthetic 'room' stream, {cam} intrinsics {W}x{H}, voxel 2 mm, "
                                       f"truncation 12 mm, max depth {md:g} m, {len(frames)}-frame ping-pong",
               value=round(fps, 1), unit="frames/s", steps=steps, reps=reps, frames_per_step=len(frames),
               frame=dict(avg_visible_blocks=round(tot["visible_blocks"] / tot["frames"], 1),
                          avg_updated_voxels=round(tot["updated_voxels"] / tot["frames"], 1),
                          alg_bytes=round(b_alg), active_blocks=stats["active_blocks"],
                          map_voxel_bytes=stats["active_blocks"] * 6144,
                          hbm_resident=bool(stats["active_blocks"] * 6144 > 256 * 2 ** 20)),
               roofline=roofline_block(b_alg, k_ms, k_n, config, whole_frame_gbps=b_alg * fps / 1e9),
               cpu_frames_per_s=round(n_par / t_cpu, 1), cpu_threads=cpu_threads,
               parity=dict(frames=n_par, max_abs_tsdf=worst["tsdf"], max_abs_prob=worst["prob"],
                           directory="bit-exact"))
    eng.close()
    return out


def bench_flythrough(ratsdf, torch, dev, dev_index, cam, vs, md, nframes, cpu_threads, n_par=12):
    """A non-repeating stream: one pass of the room camera (1 degree / frame, no frame seen twice), from
    an empty map.  Every frame allocates; the first one allocates a whole view (the serial role's long
    path).  Every frame's k_integrate carries events (ratsdf_profile_enable(2)): per-frame kernel time
    and start-to-start period -> p50 / p99 / max, which a repeating sweep never shows."""
    from ratsdf import synthetic
    from oracle_binding import load_oracle
    from parity import assert_maps_equal
    from ratsdf._abi import Engine
    frames = [synthetic.frame("room", i, cam=cam, noise=True, holes=True) for i in range(nframes)]
    H, W = frames[0]["depth"].shape
    d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]
    intr = [ratsdf.Intrinsics(*f["intrinsics"]) for f in frames]
    pose = [ratsdf.Pose(*f["pose"]) for f in frames]
    cpu = Engine(load_oracle(), vs, 6 * vs, threads=cpu_threads)
    chk = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
    chained = 0   # chained-bucket requests the checked frames sent through the resolver (GPU count)
    for i, f in enumerate(frames[:n_par]):
        cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
        chk.integrate_device(d[i]["rgb"].data_ptr(), d[i]["depth"].data_ptr(), d[i]["ht"].data_ptr(),
                             d[i]["lt"].data_ptr(), H, W, md, intr[i], pose[i])
        if n_par > 32:   # a long check: count them (one sync per frame; this engine is not the timed one)
            chk.synchronize()
            chained += chk.last_frame_stats()["slow_requests"]
    chk.synchronize()
    worst = assert_maps_equal(chk, cpu)
    chk.close()
    cpu.close()
    def one_pass(every_frame):
        eng = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
        batch = eng.make_batch([x["rgb"].data_ptr() for x in d], [x["depth"].data_ptr() for x in d],
                               [x["ht"].data_ptr() for x in d], [x["lt"].data_ptr() for x in d], H, W, md,
                               intr, pose)
        if not every_frame:   # scratch + the pass's HIP graph ahead of time: a long-lived engine has them already
            eng.prepare_device_batch(len(d), H, W)
        eng.synchronize()
        eng.totals(reset=True)
        if every_frame:
            eng.profile_enable(True, every_frame=True)
        t0 = time.perf_counter()
        eng.integrate_device_batch(batch)
        eng.synchronize()
        dt = time.perf_counter() - t0
        k_us = period_us = None
        if every_frame:
            k_us, period_us = eng.profile_read_frames()
            eng.profile_enable(False)
        tot = eng.totals()
        stats = eng.last_frame_stats()
        eng.close()
        return dt, k_us, period_us, tot, stats
    # (1) the product's launch mode: profiling off, the whole pass is ONE replay of a HIP graph (what a caller of
    #     ratsdf_integrate_device_batch gets) -- three passes, each from an empty map; the median is `frames_per_s`.
    #     (scratch and graph are built before the clock starts: ratsdf_prepare_device_batch)
    dts = sorted(one_pass(False)[0] for _ in range(3))
    dt_graph = dts[1]
    # (2) the same pass launched frame by frame with HIP events on every dispatch (ratsdf_profile_enable(2): no
    #     graphs): per-frame k_integrate time and start-to-start period -> the percentiles; its own frames/s is
    #     reported beside them as `frames_per_s_event_pass`
    dt, k_us, period_us, tot, stats = sorted((one_pass(True) for _ in range(3)), key=lambda r: r[0])[1]
    per = np.sort(period_us[:-1]) if len(period_us) > 1 else np.zeros(1)
    ks = np.sort(k_us)
    q = lambda a, p: float(a[min(len(a) - 1, int(p * len(a)))])
    return dict(workload=f"one pass of the 'room' camera, {cam} intrinsics {W}x{H}, voxel {vs * 1e3:g} mm, "
                         f"{nframes} frames, 1 deg / frame, from an empty map (no frame seen twice)",
                frames=nframes, frames_per_s=round(nframes / dt_graph, 1),
                frames_per_s_launch_mode="profiling off: the pass is one HIP-graph replay (median of 3 passes, each from an empty map)",
                frames_per_s_event_pass=round(nframes / dt, 1),
                percentiles_from="a separate pass launched frame by frame with HIP events on every dispatch (no graphs)",
                frame_period_us=dict(p50=round(q(per, .5), 1), p90=round(q(per, .9), 1), p99=round(q(per, .99), 1),
                                     max=round(float(per[-1]), 1)),
                k_integrate_us=dict(first_frame=round(float(k_us[0]), 1), p50=round(q(ks, .5), 1),
                                    p99=round(q(ks, .99), 1), max=round(float(ks[-1]), 1)),
                allocated_blocks_per_frame=round(tot["allocated_blocks"] / max(tot["frames"], 1), 1),
                deleted_blocks_per_frame=round(tot["deleted_blocks"] / max(tot["frames"], 1), 1),
                # (what the update has to do per frame -- compare `frame.avg_visible_blocks` of the repeating sweep:
                # a map that has been seen once keeps every block of the truncation band, a map that has been swept
                # many times has had the empty ones carved away)
                visible_blocks_per_frame=round(tot["visible_blocks"] / max(tot["frames"], 1), 1),
                updated_voxels_per_frame=round(tot["updated_voxels"] / max(tot["frames"], 1), 1),
                allocated_blocks_first_frame=None, active_blocks=stats["active_blocks"],
                map_voxel_bytes=stats["active_blocks"] * 6144,
                parity=dict(frames=n_par, max_abs_tsdf=worst["tsdf"], max_abs_prob=worst["prob"],
                            directory="bit-exact", chained_bucket_requests=chained if n_par > 32 else None),
                note="period = start of a frame's k_integrate to the start of the next frame's (HIP events "
                     "attached to the dispatches); the first frame allocates a whole view")


def bench_tsdf_system(frames, md, vs, nframes=3000):
    """The reference's calling convention end to end, in C++: TSDFSystem::Integrate (deep copy of pageable
    host images into the queue, modules/tsdf_module.cc:22-37) -> worker thread -> IntegrateBatch ->
    Flush (ra-slam_amd/host/src/system_bench.cc).  PCIe included; never the headline value."""
    import subprocess
    import tempfile
    exe = ROOT / "ra-slam_amd" / "host" / "build" / "ratsdf_system_bench"
    lib = Path(os.environ.get("RATSDF_LIB", ROOT / "ra-slam_amd" / "csrc" / "build" / "libratsdf.so"))
    if not exe.exists():
        return dict(error="ratsdf_system_bench not built (make -C ra-slam_amd/host)")
    H, W = frames[0]["depth"].shape
    use = frames[:24]
    with tempfile.NamedTemporaryFile(suffix=".bin", dir="/tmp") as fh:
        fh.write(np.array([H, W, len(use)], dtype=np.int32).tobytes())
        for f in use:
            fh.write(np.array(f["pose"], dtype=np.float32).tobytes())
            fh.write(np.array(f["intrinsics"], dtype=np.float32).tobytes())
            for k in ("depth", "ht", "lt", "rgb"):
                fh.write(np.ascontiguousarray(f[k]).tobytes())
        fh.flush()
        out = {}
        for tag, extra in (("with_semantics", []), ("tsdf_only", ["--no-sem"])):
            r = subprocess.run([str(exe), fh.name, "--lib", str(lib), "--frames", str(nframes), "--voxel", str(vs),
                                "--max-depth", str(md)] + extra, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return dict(error=(r.stdout + r.stderr)[-400:])
            out[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    out["note"] = ("ratsdf::TSDFSystem (the reference's class shape) fed from pageable std::vector images by one "
                   "producer thread; the queue's deep copy is split over 4 threads, the worker hands up to 32 queued "
                   "frames to one ratsdf_integrate_batch(pinned) call")
    return out


def bench_sharded_stream(a, rank, world, dev, dev_index, backend, torch, dist, ratsdf):
    """BASELINE configs[3]: ONE camera stream, N spatial subvolumes (block ownership), one rank per GPU.
    Rank 0 owns the stream (packed wire chunks resident in its HBM); every chunk reaches the other ranks by one
    broadcast on a side stream, (ring - 1) chunks ahead of the integration (ratsdf.framecast.FrameCaster) -- the
    broadcasts ARE inside the timed region.  Every rank integrates what it received into its own subvolume; the
    block-directory deltas are all-gathered once per step.  `value` = frames of the one stream per second
    ("scaling": "strong")."""
    from ratsdf import framecast, multi, synthetic
    vs, md, B = a.voxel, a.max_depth, a.frames_per_step
    W, H, _ = synthetic.camera(a.cam)
    C = max(c for c in range(1, min(a.bcast_chunk, B) + 1) if B % c == 0)
    n_chunks = B // C
    slab_bits = 2
    eng = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index, shard_rank=rank, shard_count=world, shard_slab_bits=slab_bits)
    ext = torch.cuda.ExternalStream(eng.stream(), device=dev)
    packed = None
    if rank == 0:   # the camera's rank: the only one that ever sees the stream
        frames = make_stream(a.scene, a.cam, (B + 1) // 2, phase=0)[:B]
        packed = [torch.from_numpy(framecast.pack_chunk(frames[c * C:(c + 1) * C], md, H, W, C, first_frame_no=c * C)).to(dev)
                  for c in range(n_chunks)]
        del frames
    torch.cuda.synchronize()
    fc = framecast.FrameCaster(H, W, C, ring=a.bcast_ring, src=0, device=dev)

    # ---- parity on the first chunk, from the bytes this rank RECEIVED: transport (byte sums from the camera's
    # rank), then HIP shard == the CPU oracle's shard fed with the same received images
    parity = None
    if a.cpu_frames > 0:
        from oracle_binding import load_oracle
        from parity import assert_maps_equal
        from ratsdf._abi import Engine
        fc.post(packed[0] if rank == 0 else None)
        chk = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index, shard_rank=rank, shard_count=world, shard_slab_bits=slab_bits)
        cext = torch.cuda.ExternalStream(chk.stream(), device=dev)
        ch = fc.take(cext, verify=True)
        framecast.integrate_chunk(chk, ch)
        chk.synchronize()
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = max(1, min(avail, int(os.environ.get("RATSDF_CPU_THREADS", "16"))) // world)
        cpu = Engine(load_oracle(), vs, 6 * vs, threads=cores, shard_rank=rank, shard_count=world, shard_slab_bits=slab_bits)
        npix = H * W
        for i in range(ch.n):
            img = fc.chunk_tensor(ch)[i * fc.stride:(i + 1) * fc.stride].cpu().numpy()
            cpu.integrate(img[npix * 12:npix * 15].reshape(H, W, 3), img[:npix * 4].view(np.float32).reshape(H, W),
                          img[npix * 4:npix * 8].view(np.float32).reshape(H, W),
                          img[npix * 8:npix * 12].view(np.float32).reshape(H, W), ch.max_depth, ch.intrinsics[i], ch.poses[i])
        worst = assert_maps_equal(chk, cpu)
        fc.done(ch, cext)
        parity = dict(frames=ch.n, max_abs_tsdf=worst["tsdf"], max_abs_prob=worst["prob"], directory="bit-exact",
                      transport="per-frame byte sums of the received chunk == the camera rank's", oracle_threads=cores)
        chk.close()
        cpu.close()

    ex = multi.DirectoryDeltaExchange(engine=eng, device=dev) if backend == "nccl" else multi.DirectoryDeltaExchange(engine=eng)

    def exchange():
        if backend == "nccl":
            ex.fill_from_engine(eng)
        else:
            ex.fill_from_numpy(eng.dump_directory()[1])
        try:
            ex.all_gather()
        except OverflowError:   # (every rank raises together; the exchange has restarted with whole directories)
            pass

    def run(nsteps):
        total, posted = nsteps * n_chunks, 0
        for k in range(total):
            while posted < total and fc.can_post():
                fc.post(packed[posted % n_chunks] if rank == 0 else None)
                posted += 1
            ch = fc.take(ext)
            framecast.integrate_chunk(eng, ch)
            fc.done(ch, ext)
            if (k + 1) % n_chunks == 0:
                exchange()

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    run(a.warmup)
    fence()
    eng.totals(reset=True)
    if not a.no_profile:
        eng.profile_enable(True)
    sent0 = fc.bytes_sent
    rep_dt = []
    for _ in range(max(a.reps, 1)):
        t0 = time.perf_counter()
        run(a.steps)
        fence()
        d = time.perf_counter() - t0
        t = torch.tensor([d], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rep_dt.append(float(t.item()))
    dt = sorted(rep_dt)[len(rep_dt) // 2]
    k_ms, k_n = (0.0, 0)
    if not a.no_profile:
        k_ms, k_n = eng.profile_read()
        eng.profile_enable(False)
    tot = eng.totals()
    stats = eng.last_frame_stats()
    shares = [None] * world
    dist.all_gather_object(shares, dict(visible_blocks=round(tot["visible_blocks"] / max(tot["frames"], 1), 1),
                                        updated_voxels=round(tot["updated_voxels"] / max(tot["frames"], 1), 1),
                                        active_blocks=stats["active_blocks"]))
    per_rank = ex.result()
    out = None
    if rank == 0:
        union = multi.check_sharded_directories(per_rank, slab_bits=slab_bits)   # every block on its owner, on no other rank
        nframes = a.steps * B
        fps = nframes / dt
        V = tot["visible_blocks"] / max(tot["frames"], 1)
        U = tot["updated_voxels"] / max(tot["frames"], 1)
        b_alg = 15.0 * W * H + 12.0 * V + 24.0 * U     # this rank's launch: the whole image, its share of the blocks
        roof = roofline_block(b_alg, k_ms, k_n, a.config, whole_frame_gbps=b_alg * fps / 1e9)
        bytes_timed = (fc.bytes_sent - sent0) / max(len(rep_dt), 1)
        out = {
            "metric": ("depth+semantic frames/sec integrated @640x480, 5mm voxels" if a.config == "vga5mm"
                       else f"depth+semantic frames/sec integrated @{W}x{H}, {vs * 1e3:g}mm voxels"),
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "reps": len(rep_dt),
            "value_min_max": [round(nframes / max(rep_dt), 1), round(nframes / min(rep_dt), 1)],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"ONE synthetic '{a.scene}' RGB-D+ht/lt stream, {a.cam} intrinsics {W}x{H}, voxel "
                            f"{vs * 1e3:g} mm, truncation {6 * vs * 1e3:g} mm, max depth {md:g} m, 1 deg/frame ping-pong "
                            f"sweep, 1 mm depth noise, 1% holes; {world} spatial subvolumes, one per GPU",
                "frames_per_step": B, "streams": 1,
                "sharding": f"block ownership: floormod(block.x >> {slab_bits}, N)",
                "directory_allgather_every_frames": B,
            },
            "frame_broadcast": {
                "camera_rank": 0, "chunk_frames": C, "ring": fc.ring, "frames_ahead": fc.frames_ahead,
                "bytes_per_frame": fc.stride + framecast.HEADER_BYTES, "in_timed_region": True,
                "broadcast_gbps": round(bytes_timed / dt / 1e9, 2), "backend": backend,
                "note": "one dist.broadcast per chunk on a side stream into a ring of device buffers, event-ordered "
                        "against the engine's stream both ways; rate = payload delivered to every rank / wall time "
                        "of the timed region (the broadcasts overlap the integration)"},
            "directory_blocks_all_ranks": int(sum(len(x) for x in per_rank)),
            "directory_union_blocks": int(union),
            "directory_delta_entries_last_step_rank0": list(ex.last_sent),
            "shards": shares,
            "frame": {"avg_visible_blocks": round(V, 1), "avg_updated_voxels": round(U, 1), "alg_bytes": round(b_alg),
                      "active_blocks": stats["active_blocks"], "note": "rank 0's subvolume"},
            "roofline": roof, "cpu_baseline": None, "parity": parity,
        }
    eng.close()
    return out


def self_launch(a):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as CHILD processes through
    torch.distributed.run (one rank per GPU over RCCL, rendezvous on 127.0.0.1), forward what they print and
    exit with their status.  Called before this process has imported torch or touched the GPU: nothing here
    initialises HIP, and nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as sock:   # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    # This pool's host driver supports dmabuf IPC only: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's intra-node
    # set-up fails in hipIpcGetMemHandle ("invalid argument").  The image exports it already (so does the GPU
    # box); setdefault only covers a caller that scrubbed the environment, and never overrides a value.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            self_launch(a)   # does not return
        a.gpus = world
    # RATSDF_BENCH_DEVICE / RATSDF_BENCH_BACKEND exist only to rehearse the N > 1 control flow on a
    # one-GPU box (all ranks on device 0, gloo instead of RCCL); the driver never sets them.
    dev_index = int(os.environ.get("RATSDF_BENCH_DEVICE", local_rank))
    backend = os.environ.get("RATSDF_BENCH_BACKEND", "nccl")
    # RATSDF_BENCH_ONE_RANK_GROUP=1 (rehearsal only, like the two above): take the N > 1 code path with a process group
    # of ONE rank -- RCCL refuses two ranks on one device, so this is how the very calls an 8-GPU run makes
    # (init_process_group("nccl"), the directory all-gather on device tensors, barrier, all_reduce, the frame broadcast
    # of --shard) are exercised over RCCL on a one-GPU box (tests/test_bench_launch.py)
    grouped = world > 1 or os.environ.get("RATSDF_BENCH_ONE_RANK_GROUP") == "1"
    dist = None
    if not grouped and not a.with_torch:
        # N = 1 needs device memory and nothing else of PyTorch: hipMalloc / hipMemcpy through the runtime libratsdf.so
        # links (ratsdf.devmem).  A process has ONE HIP runtime: with torch imported (first, or it finds no GPU) that is
        # the one bundled with the PyTorch wheel; without, the system ROCm runtime a C / C++ caller of the library links.
        # Same-box A/B of the two (--with-torch, profiles/r05_runtime_ab.txt): frames/s equal within 1 %, host cost of
        # enqueueing a frame 1.0 vs 1.9 us -- and no `import torch` (a minute or two on a fresh box) in front of the run.
        # N > 1 needs torch.distributed and runs on torch's runtime.
        from ratsdf import devmem
        if devmem.runtime_library_present():
            torch = devmem.TorchLike()
            dev = dev_index
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs an MI355X: the TSDF engine has no CPU path")
        else:   # (no HIP runtime library to bind by name: PyTorch's allocator then -- decided before anything is loaded)
            a.with_torch = True
    if grouped or a.with_torch:
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the TSDF engine has no CPU path")
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # (the one-rank rehearsal, started without a launcher)
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import ratsdf
    if a.config == "hd2mm":
        a.cam, a.voxel = "l515_720p", 0.002
        if a.frames_per_step == 90:
            a.frames_per_step = 30
        if a.cpu_frames == 1080:
            a.cpu_frames = 60
    if a.config == "flythrough":   # the non-repeating pass as the whole run (1280x720 / 2 mm, a full circle)
        if not torch.cuda.is_available():
            raise SystemExit("needs a GPU")
        n = a.flythrough_frames if a.flythrough_frames != 180 else 360
        out = bench_flythrough(ratsdf, torch, dev, dev_index, "l515_720p", 0.002, a.max_depth, n,
                               min(len(os.sched_getaffinity(0)), 16), n_par=n)  # parity over the WHOLE pass
        out = {"metric": "depth+semantic frames/sec integrated @1280x720, 2mm voxels (non-repeating pass)",
               "value": out["frames_per_s"], "unit": "frames/s", "n_gpus": 1, "higher_is_better": True,
               "dtype": "f32", "data": "synthetic", "config": {"workload": out["workload"]}, "flythrough": out}
        print(json.dumps(out))
        return
    if a.config == "bigmap":
        a.cam, a.voxel = "l515_720p", 0.002
        if a.frames_per_step == 90:
            a.frames_per_step = 240      # 120 frames of a 1 degree/frame sweep, there and back
        if a.steps == 60:
            a.steps, a.warmup, a.reps = 2, 1, 3
        if a.cpu_frames == 1080:
            a.cpu_frames = 8             # parity on the first frames only (the CPU runs ~30 frames/s)
        a.host_frames = 0
        a.streams = 0
    if a.shard and grouped:   # BASELINE configs[3]: one stream, N subvolumes, frames broadcast from rank 0
        out = bench_sharded_stream(a, rank, world, dev, dev_index, backend, torch, dist, ratsdf)
        if rank == 0:
            print(json.dumps(out))
        dist.destroy_process_group()
        return
    vs = a.voxel
    B = a.frames_per_step
    half = (B + 1) // 2
    frames = make_stream(a.scene, a.cam, half, phase=45 * rank)
    frames = frames[:B] if len(frames) >= B else frames
    H, W = frames[0]["depth"].shape
    # resident inputs
    d_rgb = [torch.from_numpy(f["rgb"]).to(dev) for f in frames]
    d_depth = [torch.from_numpy(f["depth"]).to(dev) for f in frames]
    d_ht = [torch.from_numpy(f["ht"]).to(dev) for f in frames]
    d_lt = [torch.from_numpy(f["lt"]).to(dev) for f in frames]
    intr = [ratsdf.Intrinsics(*f["intrinsics"]) for f in frames]
    pose = [ratsdf.Pose(*f["pose"]) for f in frames]
    torch.cuda.synchronize()

    shard_kw = dict(shard_rank=rank, shard_count=world, shard_slab_bits=2) if (a.shard and grouped) else {}
    eng = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index, **shard_kw)

    # ---- parity + CPU baseline on a bounded prefix (rank 0, N = 1) ---------------------------
    cpu_baseline = None
    parity = None
    if rank == 0 and not grouped and a.cpu_frames > 0:
        from oracle_binding import load_oracle
        from parity import assert_maps_equal
        from ratsdf._abi import Engine
        ncpu = a.cpu_frames  # the ping-pong stream repeats, so any length is a valid prefix
        # the 1-GPU box exposes every host CPU but grants a 16-core share; use that many threads
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(avail, int(os.environ.get("RATSDF_CPU_THREADS", "16")))
        cpu = Engine(load_oracle(), vs, 6 * vs, threads=cores)
        chk = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
        # timed on the regime the GPU is timed on: the first two sweeps of the stream build and saturate the map
        # (untimed), the rest revisits it (the whole prefix from an empty map was timed before)
        n_build = 2 * len(frames) if ncpu >= 4 * len(frames) else 0
        t0 = time.perf_counter()
        for j in range(ncpu):
            if j == n_build:
                t0 = time.perf_counter()
            f = frames[j % len(frames)]
            cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], a.max_depth, f["intrinsics"],
                          f["pose"])
        t_cpu = time.perf_counter() - t0
        for j in range(ncpu):
            i = j % len(frames)
            chk.integrate_device(d_rgb[i].data_ptr(), d_depth[i].data_ptr(), d_ht[i].data_ptr(),
                                 d_lt[i].data_ptr(), H, W, a.max_depth, intr[i], pose[i])
        chk.synchronize()
        worst = assert_maps_equal(chk, cpu)
        parity = dict(frames=ncpu, max_abs_tsdf=worst["tsdf"], max_abs_prob=worst["prob"],
                      directory="bit-exact")
        cpu_baseline = dict(value=(ncpu - n_build) / t_cpu, unit="frames/s", cores=cores, kind="port",
                            seconds=round(t_cpu, 2),
                            sample=(f"frames {n_build} .. {ncpu} of the same stream: the map has been built and "
                                    f"saturated by the {n_build} untimed frames before them, as in the GPU's timed "
                                    f"region" if n_build else f"first {ncpu} frames of the same stream from an empty map") +
                                   f" (oracle/ratsdf_oracle.cpp, {cores} threads)")
        chk.close()
        cpu.close()

    # ---- directory all-gather (N > 1): one exchange per step (ratsdf/multi.py) -------------------
    ex = None
    if grouped:
        from ratsdf import multi
        if backend == "nccl":
            # deltas of the block directories: what a rank added / deleted since the previous
            # step; the exchange buffers hold a whole block pool (2^block_bits entries), so no directory
            # or delta can overflow them; every rank keeps replicas of all directories
            ex = multi.DirectoryDeltaExchange(engine=eng, device=dev)   # capacity = the engine's block pool
        else:  # rehearsal backend (gloo): stage through the host
            ex = multi.DirectoryDeltaExchange(engine=eng)

    def exchange():
        if backend == "nccl":
            # engine stream -> (event) -> torch stream -> RCCL all-gather -> (event) -> engine stream
            # (first exchange: the whole directory; then the engine's own delta log: dirty entries + deleted positions)
            ex.fill_from_engine(eng)
        else:
            ex.fill_from_numpy(eng.dump_directory()[1])
        try:
            ex.all_gather()
        except OverflowError:   # a delta larger than the payload: every rank raises together and the exchange has
            pass                # restarted (whole directories next time): nothing to do here

    batch = eng.make_batch([t.data_ptr() for t in d_rgb], [t.data_ptr() for t in d_depth],
                           [t.data_ptr() for t in d_ht], [t.data_ptr() for t in d_lt], H, W,
                           a.max_depth, intr, pose)

    def step():
        if a.sync_every:
            for i in range(len(frames)):
                eng.integrate_device(d_rgb[i].data_ptr(), d_depth[i].data_ptr(), d_ht[i].data_ptr(),
                                     d_lt[i].data_ptr(), H, W, a.max_depth, intr[i], pose[i])
                if (i + 1) % a.sync_every == 0:
                    eng.synchronize()
        else:
            eng.integrate_device_batch(batch)  # one C call enqueues the step's frames in order
        if grouped:
            exchange()

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    eng.totals(reset=True)
    # K steps, timed `reps` times back to back (each bracketed by barrier + synchronize); the value
    # is the median repetition.  The dominant kernel is timed inside the same region: every 4th
    # k_integrate launch carries a start and a stop event attached to its dispatch.
    if not a.no_profile:
        eng.profile_enable(True)
    rep_dt = []
    for _ in range(max(a.reps, 1)):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        d = time.perf_counter() - t0
        if grouped:
            t = torch.tensor([d], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            d = float(t.item())
        rep_dt.append(d)
    dt = sorted(rep_dt)[len(rep_dt) // 2]
    k_ms, k_n = (0.0, 0)
    if not a.no_profile:
        k_ms, k_n = eng.profile_read()
        eng.profile_enable(False)
    tot = eng.totals()
    stats = eng.last_frame_stats()

    # host cost of enqueueing a frame, measured on an IDLE queue (short bursts, synchronised in
    # between): while the GPU is the bottleneck a long enqueue loop only measures the back-pressure
    # of the full hardware queue, not the host
    enq_us = None
    if not a.sync_every:
        burst = min(16, len(frames))
        small = eng.make_batch([t.data_ptr() for t in d_rgb[:burst]], [t.data_ptr() for t in d_depth[:burst]],
                               [t.data_ptr() for t in d_ht[:burst]], [t.data_ptr() for t in d_lt[:burst]],
                               H, W, a.max_depth, intr[:burst], pose[:burst])
        best = None
        for _ in range(5):
            eng.synchronize()
            t0 = time.perf_counter()
            eng.integrate_device_batch(small)
            d = (time.perf_counter() - t0) / burst
            best = d if best is None else min(best, d)
        eng.synchronize()
        enq_us = best * 1e6

    # One frame per call, the caller waiting for each (TSDFGrid::Integrate's own convention, voxel_tsdf.cu:376-452: it
    # returns when the frame is in the map; here with the images already on the device): no frame is known ahead, so no
    # look-ahead candidate pass and no graph -- three launches and a stream synchronisation per frame.  What a tracking
    # loop that needs the map between frames gets; reported separately, never the headline value.
    sync_path = None
    if rank == 0 and not grouped and not a.sync_every and a.host_frames > 0:   # (--host-frames 0, what the profiling
        lat = []                                                                # scripts pass: none of the caller-path legs)
        n_sync = min(4 * len(frames), 360)
        eng.synchronize()
        t_all = time.perf_counter()
        for j in range(n_sync):
            i = j % len(frames)
            t0 = time.perf_counter()
            eng.integrate_device(d_rgb[i].data_ptr(), d_depth[i].data_ptr(), d_ht[i].data_ptr(), d_lt[i].data_ptr(),
                                 H, W, a.max_depth, intr[i], pose[i])
            eng.synchronize()
            lat.append(time.perf_counter() - t0)
        t_all = time.perf_counter() - t_all
        lat.sort()
        sync_path = {"frames": n_sync, "frames_per_s": round(n_sync / t_all, 1),
                     "latency_us": {"p50": round(lat[len(lat) // 2] * 1e6, 1), "p99": round(lat[int(len(lat) * 0.99)] * 1e6, 1)},
                     "note": "ratsdf_integrate_device + ratsdf_synchronize per frame (call to frame-in-map, host clock); "
                             "the same saturated map as the timed region"}

    # Host-image entry points, PCIe included (the reference's calling convention: TSDFSystem::Integrate
    # hands over cv::Mat images in host memory, modules/tsdf_module.cc:22-37,88-115).  Reported
    # separately; never the headline value.
    #   per frame   ratsdf_integrate: staging copy + H2D + a stream sync per frame
    #   batched     ratsdf_integrate_batch, 8 frames per call from pageable memory (what
    #               ratsdf::TSDFSystem's worker does with its queue); uploads on a copy stream
    #   pinned      the same call on page-locked buffers (ratsdf_host_alloc), 32 frames per call:
    #               no staging copy, bound by the PCIe link (63 GB/s spec)
    # (Measured in a CHILD process that does not load PyTorch -- tools/host_paths.py -- like the C++ TSDFSystem leg: the
    # system ROCm runtime a C / C++ caller links, whatever this process runs on.)
    host_path = None
    pinned_path = None
    if rank == 0 and not grouped and a.host_frames > 0:
        import subprocess
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "host_paths.py"), "--cam", a.cam, "--scene", a.scene,
                            "--voxel", str(vs), "--max-depth", str(a.max_depth), "--frames", str(len(frames)),
                            "--host-frames", str(a.host_frames), "--device", str(dev_index)],
                           capture_output=True, text=True, timeout=600)
        if r.returncode == 0 and r.stdout.strip():
            hp_out = json.loads(r.stdout.strip().splitlines()[-1])
            host_path, pinned_path = hp_out["host_image_path"], hp_out["pinned_h2d_path"]
        else:
            host_path = dict(error=(r.stdout + r.stderr)[-400:])

    # ---- S streams on this GPU through one launch triple per frame step ------------------------
    multi = None
    if rank == 0 and not grouped and a.streams > 1 and not a.shard:
        multi = bench_streams(ratsdf, torch, dev, dev_index, a.streams, a.scene, a.cam, vs, a.max_depth,
                              min(60, len(frames)), max(a.steps // 2, 4), 3,
                              cpu_threads=min(len(os.sched_getaffinity(0)), 16))

    # ---- the other stream size north_star names, bounded ----------------------------------------
    secondary = None
    flythrough = None
    system_path = None
    tsdf_only = None
    if rank == 0 and not grouped and a.config == "vga5mm" and not a.no_secondary and a.cpu_frames > 0:
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        nthr = min(avail, int(os.environ.get("RATSDF_CPU_THREADS", "16")))
        secondary = [bench_secondary(ratsdf, torch, dev, dev_index, a.max_depth, nthr, "hd2mm"),
                     bench_secondary(ratsdf, torch, dev, dev_index, a.max_depth, nthr, "bigmap")]
        flythrough = bench_flythrough(ratsdf, torch, dev, dev_index, a.cam, vs, a.max_depth, a.flythrough_frames, nthr,
                                      n_par=a.flythrough_frames)  # checked against the oracle over the whole pass
        system_path = bench_tsdf_system(frames, a.max_depth, vs)
        tsdf_only = bench_tsdf_only(ratsdf, frames, d_rgb, d_depth, intr, pose, vs, a.max_depth, dev_index, nthr)

    nframes = a.steps * len(frames)
    if rank == 0:
        fps = (1 if a.shard else world) * nframes / dt
        V = tot["visible_blocks"] / max(tot["frames"], 1)
        U = tot["updated_voxels"] / max(tot["frames"], 1)
        b_alg = 15.0 * W * H + 12.0 * V + 24.0 * U
        roof = roofline_block(b_alg, k_ms, k_n, a.config, whole_frame_gbps=b_alg * fps / world / 1e9)
        out = {
            "metric": ("depth+semantic frames/sec integrated @640x480, 5mm voxels" if a.config == "vga5mm"
                       else f"depth+semantic frames/sec integrated @{W}x{H}, {vs * 1e3:g}mm voxels"),
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "reps": len(rep_dt),
            "value_min_max": [round((1 if a.shard else world) * nframes / max(rep_dt), 1),
                              round((1 if a.shard else world) * nframes / min(rep_dt), 1)],
            "higher_is_better": True,
            "scaling": "strong" if (a.shard and grouped) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "hip_runtime": ("the system ROCm runtime libratsdf.so links (no PyTorch in this process: device memory "
                            "through ratsdf.devmem)" if (not grouped and not a.with_torch) else
                            "the HIP runtime bundled with the PyTorch wheel (torch.distributed / RCCL need torch in the process)"),
            "config": {
                "workload": f"synthetic '{a.scene}' RGB-D+ht/lt stream, {a.cam} intrinsics {W}x{H}, "
                            f"voxel {vs * 1e3:g} mm, truncation {6 * vs * 1e3:g} mm, max depth "
                            f"{a.max_depth:g} m, 1 deg/frame ping-pong sweep, 1 mm depth noise, 1% holes",
                "frames_per_step": len(frames),
                "streams": 1 if a.shard else world,
                "sharding": ("block ownership: floormod(block.x >> 2, N)" if (a.shard and grouped) else None),
                "directory_allgather_every_frames": len(frames) if grouped else None,
            },
            "directory_blocks_all_ranks": (int(sum(len(x) for x in ex.result())) if grouped else None),
            "directory_delta_entries_last_step_rank0": (list(ex.last_sent) if grouped else None),
            "frame": {"avg_visible_blocks": round(V, 1), "avg_updated_voxels": round(U, 1),
                      "alg_bytes": round(b_alg), "alg_gbps_whole_frame": round(b_alg * fps / world / 1e9, 1),
                      "active_blocks": stats["active_blocks"],
                      "avg_allocated_blocks": round(tot["allocated_blocks"] / max(tot["frames"], 1), 2),
                      "avg_deleted_blocks": round(tot["deleted_blocks"] / max(tot["frames"], 1), 2),
                      # voxel data of the map (3 pools x 512 voxels x 4 B per block): what has to
                      # exceed the 256 MiB Infinity Cache before traffic figures mean HBM
                      "map_voxel_bytes": stats["active_blocks"] * 6144},
            # host cost of a frame's three launches on an idle queue / GPU time per frame
            "host_enqueue_us_per_frame": round(enq_us, 2) if enq_us else None,
            "host_enqueue_frac": round(enq_us * 1e-6 / (dt / nframes), 3) if enq_us else None,
            "roofline": roof,
            "cpu_baseline": cpu_baseline,
            "synchronous_frame_path": sync_path,
            "host_image_path": host_path,
            "pinned_h2d_path": pinned_path,
            "tsdf_system_path": system_path,
            "tsdf_only": tsdf_only,
            "multi_stream": multi,
            "secondary": secondary,
            "flythrough": flythrough,
            "parity": parity,
        }
        print(json.dumps(out))
    eng.close()
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
