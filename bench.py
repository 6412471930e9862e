#!/usr/bin/env python3
"""bench.py -- depth+semantic frames/s integrated by the MI355X TSDF engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A step = one batch of `--frames-per-step` synthetic 640x480 frames (depth + rgb + ht/lt, 5 mm voxels,
truncation 30 mm, max depth 4 m, ScanNet intrinsics; the "room" stream of ratsdf.synthetic)
integrated through the C ABI (ratsdf_integrate_device) with every input already resident in HBM.
For N > 1 the driver launches one process per GPU with torch.distributed.run; each rank integrates
its own stream into its own map (frame-batched config of BASELINE.json) and the ranks all-gather
their block directories over RCCL once per step.  Rank 0 prints ONE JSON line.

The line also carries
  roofline      HBM roofline of the dominant kernel (k_integrate): algorithmic bytes per launch
                (15 W H + 12 V + 24 U) / its average duration measured with HIP events
                on the engine's stream inside the timed region
  cpu_baseline  the CPU oracle (multithreaded port; the reference has no CPU path) timed on this
                box's host cores on a bounded prefix of the same stream (rank 0, N = 1 only), after
                asserting that it and the HIP engine produce the same map on that prefix
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
    ap.add_argument("--config", default="vga5mm", choices=["vga5mm", "hd2mm"],
                    help="vga5mm: BASELINE metric config (640x480, 5 mm); hd2mm: 1280x720, 2 mm "
                         "(BASELINE configs[3] workload, single GPU unless --shard)")
    ap.add_argument("--shard", action="store_true",
                    help="N > 1: every rank integrates the SAME stream and owns the blocks "
                         "owner(block) == rank (spatial subvolumes, strong scaling) instead of "
                         "one independent stream per rank")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event timing of k_integrate")
    ap.add_argument("--host-frames", type=int, default=60,
                    help="frames timed through the host-image entry point, PCIe included (0 = skip)")
    ap.add_argument("--sync-every", type=int, default=0,
                    help="diagnostic: host-synchronise every N frames (0 = only at the end)")
    return ap.parse_args()


def make_stream(scene, cam, nframes, phase):
    """Ping-pong camera sweep so the stream can repeat forever: frames 0..n-1 then n-1..0."""
    from ratsdf import synthetic
    half = [synthetic.frame(scene, phase + i, cam=cam, noise=True, holes=True) for i in
            range(nframes)]
    return half + half[::-1]


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N > 1 with: python -m torch.distributed.run --nproc-per-node N "
                             "--master-addr 127.0.0.1 bench.py --gpus N ...")
        a.gpus = world
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the TSDF engine has no CPU path")
    # RATSDF_BENCH_DEVICE / RATSDF_BENCH_BACKEND exist only to rehearse the N > 1 control flow on a
    # one-GPU box (all ranks on device 0, gloo instead of RCCL); the driver never sets them.
    dev_index = int(os.environ.get("RATSDF_BENCH_DEVICE", local_rank))
    backend = os.environ.get("RATSDF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    vs = a.voxel
    B = a.frames_per_step
    half = (B + 1) // 2
    frames = make_stream(a.scene, a.cam, half, phase=0 if a.shard else 45 * rank)
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

    shard_kw = dict(shard_rank=rank, shard_count=world, shard_slab_bits=2) if (a.shard and world > 1) else {}
    eng = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index, **shard_kw)
    ext = torch.cuda.ExternalStream(eng.stream(), device=dev)

    # ---- parity + CPU baseline on a bounded prefix (rank 0, N = 1) ---------------------------
    cpu_baseline = None
    parity = None
    if rank == 0 and world == 1 and a.cpu_frames > 0:
        from oracle_binding import load_oracle
        from parity import assert_maps_equal
        from ratsdf._abi import Engine
        ncpu = a.cpu_frames  # the ping-pong stream repeats, so any length is a valid prefix
        # the 1-GPU box exposes every host CPU but grants a 16-core share; use that many threads
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(avail, int(os.environ.get("RATSDF_CPU_THREADS", "16")))
        cpu = Engine(load_oracle(), vs, 6 * vs, threads=cores)
        chk = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
        t0 = time.perf_counter()
        for j in range(ncpu):
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
        cpu_baseline = dict(value=ncpu / t_cpu, unit="frames/s", cores=cores, kind="port",
                            seconds=round(t_cpu, 2),
                            sample=f"first {ncpu} frames of the same stream from an empty map "
                                   f"(oracle/ratsdf_oracle.cpp, {cores} threads)")
        chk.close()
        cpu.close()

    # ---- directory all-gather (N > 1): one fixed-capacity exchange per step -------------------
    cap = 1 << 16
    if world > 1:
        dir_local = torch.zeros(cap * 3, dtype=torch.int32, device=dev)   # 12-byte entries
        dir_count = torch.zeros(1, dtype=torch.int32, device=dev)
        dir_all = torch.zeros(world * cap * 3, dtype=torch.int32, device=dev)
        cnt_all = torch.zeros(world, dtype=torch.int32, device=dev)

    def exchange():
        # engine stream -> (event) -> torch stream -> RCCL all-gather -> (event) -> engine stream
        eng.export_directory_device(dir_local.data_ptr(), cap, dir_count.data_ptr())
        ev = torch.cuda.Event()
        ev.record(ext)
        torch.cuda.current_stream().wait_event(ev)
        if backend == "nccl":
            dist.all_gather_into_tensor(dir_all, dir_local)
            dist.all_gather_into_tensor(cnt_all, dir_count)
        else:  # rehearsal backend: stage through the host
            a_cpu, c_cpu = torch.zeros(world * cap * 3, dtype=torch.int32), torch.zeros(
                world, dtype=torch.int32)
            dist.all_gather_into_tensor(a_cpu, dir_local.cpu())
            dist.all_gather_into_tensor(c_cpu, dir_count.cpu())
            dir_all.copy_(a_cpu)
            cnt_all.copy_(c_cpu)
        ev2 = torch.cuda.Event()
        ev2.record(torch.cuda.current_stream())
        ext.wait_event(ev2)

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
        if world > 1:
            exchange()

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    eng.totals(reset=True)
    if not a.no_profile:
        eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enqueue = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    k_ms, k_n = (0.0, 0)
    if not a.no_profile:
        k_ms, k_n = eng.profile_read()
        eng.profile_enable(False)
    tot = eng.totals()
    stats = eng.last_frame_stats()

    # host-image entry point (ratsdf_integrate: pinned staging + PCIe copy + sync per frame, like the
    # reference's Integrate).  Reported separately; never the headline value.
    host_path = None
    if rank == 0 and world == 1 and a.host_frames > 0:
        hp = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
        nh = min(a.host_frames, len(frames))
        for f in frames[:4]:
            hp.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], a.max_depth, f["intrinsics"], f["pose"])
        th = time.perf_counter()
        for f in frames[:nh]:
            hp.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], a.max_depth, f["intrinsics"], f["pose"])
        th = time.perf_counter() - th
        # the same frames through ratsdf_integrate_batch (8 at a time, as TSDFSystem's worker does)
        tb = time.perf_counter()
        for c0 in range(0, nh, 8):
            hp.integrate_batch(frames[c0:min(c0 + 8, nh)], a.max_depth)
        tb = time.perf_counter() - tb
        host_path = dict(frames_per_s=round(nh / th, 1), frames=nh, batched_frames_per_s=round(nh / tb, 1),
                         note="ratsdf_integrate with host images: H2D copy (4.6 MB/frame) and a "
                              "stream sync per frame included; batched = ratsdf_integrate_batch, "
                              "8 frames per call")
        hp.close()

    # frames in PINNED host memory, uploaded chunk by chunk on a copy stream while the engine
    # integrates the previous chunk ("separately reported, including H2D from pinned host
    # memory").  PCIe-bound; reported separately, never the headline value.
    pinned_path = None
    if rank == 0 and world == 1 and a.host_frames > 0:
        pp = ratsdf.TSDFGrid(vs, 6 * vs, device=dev_index)
        pext = torch.cuda.ExternalStream(pp.stream(), device=dev)
        copy_stream = torch.cuda.Stream(device=dev)
        C = 6                                            # frames per chunk, two chunks in flight
        keys = ("rgb", "depth", "ht", "lt")
        h_pin = [{k: torch.from_numpy(f[k]).pin_memory() for k in keys} for f in frames]
        ring = [[{k: torch.empty_like(h_pin[0][k], device=dev) for k in keys} for _ in range(C)]
                for _ in range(2)]
        batches = {}
        def chunk_batch(slot, idx):
            key = (slot, tuple(idx))
            if key not in batches:
                r = ring[slot]
                batches[key] = pp.make_batch([r[j]["rgb"].data_ptr() for j in range(len(idx))],
                                             [r[j]["depth"].data_ptr() for j in range(len(idx))],
                                             [r[j]["ht"].data_ptr() for j in range(len(idx))],
                                             [r[j]["lt"].data_ptr() for j in range(len(idx))], H, W,
                                             a.max_depth, [intr[i] for i in idx], [pose[i] for i in idx])
            return batches[key]
        def run_pinned(n_frames):
            free_ev = [None, None]                      # ring slot may be overwritten after this event
            order = [i % len(frames) for i in range(n_frames)]
            for c0 in range(0, n_frames, C):
                idx = order[c0:c0 + C]
                slot = (c0 // C) & 1
                with torch.cuda.stream(copy_stream):
                    if free_ev[slot] is not None:
                        copy_stream.wait_event(free_ev[slot])
                    for j, i in enumerate(idx):
                        for k in keys:
                            ring[slot][j][k].copy_(h_pin[i][k], non_blocking=True)
                    up = torch.cuda.Event()
                    up.record(copy_stream)
                pext.wait_event(up)
                pp.integrate_device_batch(chunk_batch(slot, idx))
                done = torch.cuda.Event()
                done.record(pext)
                free_ev[slot] = done
            pp.synchronize()
            torch.cuda.synchronize()
        run_pinned(4 * C)
        npin = max(a.host_frames, 20 * C) // C * C
        tp = time.perf_counter()
        run_pinned(npin)
        tp = time.perf_counter() - tp
        bytes_per_frame = sum(h_pin[0][k].numel() * h_pin[0][k].element_size() for k in keys)
        pinned_path = dict(frames_per_s=round(npin / tp, 1), frames=npin,
                           h2d_gbps=round(npin * bytes_per_frame / tp / 1e9, 1),
                           note=f"frames in pinned host memory, {C}-frame chunks uploaded on a copy "
                                "stream while the previous chunk is integrated "
                                "(ratsdf_integrate_device_batch)")
        pp.close()

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    nframes = a.steps * len(frames)
    if rank == 0:
        fps = (1 if a.shard else world) * nframes / dt
        V = tot["visible_blocks"] / max(tot["frames"], 1)
        U = tot["updated_voxels"] / max(tot["frames"], 1)
        b_alg = 15.0 * W * H + 12.0 * V + 24.0 * U
        roof = None
        if k_n:
            k_avg_s = k_ms / k_n / 1e3
            achieved = b_alg / k_avg_s / 1e9
            traffic = None
            tpath = ROOT / "profiles" / ("traffic_latest.json" if a.config == "vga5mm" else "traffic_hd2mm.json")
            if tpath.exists():  # PMC passes made on this very workload (tools/traffic.sh)
                try:
                    traffic = json.loads(tpath.read_text()).get("k_integrate_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = dict(bound="hbm", kernel="k_integrate", achieved=round(achieved, 1),
                        peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBPS, 4),
                        traffic=traffic, alg_bytes_per_launch=round(b_alg),
                        avg_launch_us=round(k_avg_s * 1e6, 2), launches=k_n)
        out = {
            "metric": ("depth+semantic frames/sec integrated @640x480, 5mm voxels" if a.config == "vga5mm"
                       else f"depth+semantic frames/sec integrated @{W}x{H}, {vs * 1e3:g}mm voxels"),
            "value": round(fps, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if (a.shard and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic '{a.scene}' RGB-D+ht/lt stream, {a.cam} intrinsics {W}x{H}, "
                            f"voxel {vs * 1e3:g} mm, truncation {6 * vs * 1e3:g} mm, max depth "
                            f"{a.max_depth:g} m, 1 deg/frame ping-pong sweep, 1 mm depth noise, 1% holes",
                "frames_per_step": len(frames),
                "streams": 1 if a.shard else world,
                "sharding": ("block ownership: floormod(block.x >> 2, N)" if (a.shard and world > 1) else None),
                "directory_allgather_every_frames": len(frames) if world > 1 else None,
            },
            "directory_blocks_all_ranks": (int(cnt_all.sum().item()) if world > 1 else None),
            "frame": {"avg_visible_blocks": round(V, 1), "avg_updated_voxels": round(U, 1),
                      "alg_bytes": round(b_alg), "alg_gbps_whole_frame": round(b_alg * fps / world / 1e9, 1),
                      "active_blocks": stats["active_blocks"]},
            "host_enqueue_frac": round(t_enqueue / dt, 3),
            "roofline": roof,
            "cpu_baseline": cpu_baseline,
            "host_image_path": host_path,
            "pinned_h2d_path": pinned_path,
            "parity": parity,
        }
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
