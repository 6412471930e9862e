"""The frame forms that measurements retired live in the DIAGNOSTIC build only
(make -C ra-slam_amd/csrc stamps, -DRATSDF_STAMPS): the serial role at the tail of k_front (RATSDF_FRONT_TAIL=1), the
serial role as a launch of its own (RATSDF_FUSED_SERIAL=0), 4 / 8 voxels per lane (RATSDF_VPL).  They stay bit-exact
against the oracle here.  Not collected by the test run directly: tests/test_gpu_diagnostic_build.py runs this file in
ONE child process with RATSDF_LIB pointing at libratsdf_stamps.so (the library is chosen at import)."""
import numpy as np
import pytest
import torch

from parity import assert_maps_equal, assert_stats_equal
from ratsdf import synthetic
from test_gpu_pipeline import check_totals, device_frames, make_batch, oracle_run
from test_gpu_group import _group_batch, _upload

pytestmark = pytest.mark.gpu


def test_this_is_the_diagnostic_build():
    import os
    import ratsdf
    assert "stamps" in os.environ.get("RATSDF_LIB", ""), "run through tests/test_gpu_diagnostic_build.py"
    assert hasattr(ratsdf.library().dll, "ratsdf_debug_counters")


@pytest.mark.parametrize("front_tail", ["1", "0"])
def test_where_the_serial_role_runs(front_tail, monkeypatch, make_engine, make_oracle):
    """Ordinary frames (no chained buckets, at most 2 048 requests / 768 winners) have their allocation-order
    pass done by the last directory workgroup of k_front (front_tail_role) and their new blocks handed to
    k_integrate as work-list items; RATSDF_FRONT_TAIL=0 keeps the pass inside k_integrate (the form every
    other frame takes).  Same map either way, and the engine says which form the frames took."""
    monkeypatch.setenv("RATSDF_FRONT_TAIL", front_tail)
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    monkeypatch.delenv("RATSDF_FRONT_TAIL")
    frames = synthetic.stream("room", 14, scale=0.25, noise=True, holes=True)
    frames = frames + frames[::-1][:6]
    dev = device_frames(frames)
    lo = 0
    for n in (1, 2, 9, 8):
        gpu.integrate_device_batch(make_batch(gpu, frames, dev, lo, lo + n, md))
        oracle_run(cpu, frames[lo:lo + n], md)
        lo += n
        assert_maps_equal(gpu, cpu)
        assert_stats_equal(gpu, cpu)
    check_totals(gpu, cpu)
    c = gpu.pipeline_counters()
    assert sum(c.values()) == len(frames), c
    if front_tail == "1":
        assert c["front_tail"] >= len(frames) - 1, c   # (the first frame of the view may be too large)
    else:
        assert c["front_tail"] == 0 and c["in_launch"] == len(frames), c


def test_front_tail_at_full_size_with_fallbacks(monkeypatch, make_engine, make_oracle):
    """640x480 / 5 mm, the benchmark's stream: the first frame of the view files thousands of requests (the
    role runs inside k_integrate), the following ones are ordinary (tail of k_front); a 60-degree jump in the
    middle of the batch sends one frame back to the in-launch form.  Parity over the whole sequence."""
    vs, md = 0.005, 4.0
    monkeypatch.setenv("RATSDF_FRONT_TAIL", "1")
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    monkeypatch.delenv("RATSDF_FRONT_TAIL")
    idx = [0, 1, 2, 3, 4, 64, 65, 66, 5, 6]
    frames = [synthetic.frame("room", i, noise=True, holes=True) for i in idx]
    dev = device_frames(frames)
    gpu.integrate_device_batch(make_batch(gpu, frames, dev, 0, len(frames), md))
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    assert_stats_equal(gpu, cpu)
    check_totals(gpu, cpu)
    c = gpu.pipeline_counters()
    assert sum(c.values()) == len(frames) and c["front_tail"] >= 5 and c["in_launch"] + c["in_launch_general"] >= 2, c


@pytest.mark.parametrize("kw", [dict(), dict(bucket_bits=9, block_bits=13)])
def test_serial_role_as_a_launch_of_its_own(kw, monkeypatch, make_engine, make_oracle):
    """RATSDF_FUSED_SERIAL=0 (read when the engine is created): the frame's allocation-order role runs
    as k_alloc_rank between k_front and k_integrate instead of inside k_integrate -- the layout the
    stand-alone test hooks use, kept for A/B measurements.  Same map either way."""
    monkeypatch.setenv("RATSDF_FUSED_SERIAL", "0")
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, **kw)
    monkeypatch.delenv("RATSDF_FUSED_SERIAL")
    frames = synthetic.stream("room", 12, scale=0.25, noise=True, holes=True)
    dev = device_frames(frames)
    lo = 0
    for n in (1, 5, 6):
        gpu.integrate_device_batch(make_batch(gpu, frames, dev, lo, lo + n, md))
        oracle_run(cpu, frames[lo:lo + n], md)
        lo += n
        assert_maps_equal(gpu, cpu)
        assert_stats_equal(gpu, cpu)
    check_totals(gpu, cpu)



@pytest.mark.parametrize("vpl", ["4", "8"])
def test_more_voxels_per_lane(vpl, monkeypatch, make_engine, make_oracle):
    """RATSDF_VPL=4 / 8: half / a quarter of the update waves (measured slower, profiles/r04_front_tail.txt)."""
    monkeypatch.setenv("RATSDF_VPL", vpl)
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    monkeypatch.delenv("RATSDF_VPL")
    frames = synthetic.stream("room", 10, scale=0.25, noise=True, holes=True)
    dev = device_frames(frames)
    lo = 0
    for n in (1, 4, 5):
        gpu.integrate_device_batch(make_batch(gpu, frames, dev, lo, lo + n, md))
        oracle_run(cpu, frames[lo:lo + n], md)
        lo += n
        assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)


def test_group_with_the_serial_role_at_the_tail_of_k_front(monkeypatch, make_engine, make_oracle):
    """every member's serial role runs at the tail of its slice of k_front_g (RATSDF_FRONT_TAIL=1)"""
    import ratsdf
    members, front_tail = 3, "1"
    monkeypatch.setenv("RATSDF_FRONT_TAIL", front_tail)
    vs = 0.02
    n = 7
    scenes = ["room", "sphere", "wall"][:members]
    streams = [synthetic.stream(sc, n, scale=0.25, noise=True, holes=True) for sc in scenes]
    dev_streams = [_upload(fr) for fr in streams]
    engines = [make_engine(vs, 6 * vs) for _ in range(members)]
    oracles = [make_oracle(vs, 6 * vs) for _ in range(members)]
    group = ratsdf.Group(engines)
    # batches of different lengths, a query and a single-engine frame in between
    group.integrate_device_batch(_group_batch(group, streams, dev_streams, 0, 3))
    group.synchronize()
    for s in range(members):
        for f in streams[s][0:3]:
            oracles[s].integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(engines[s], oracles[s])
        assert_maps_equal(engines[s], oracles[s])
    # one member advances alone (frame 3), the others through a group of their own later
    f = streams[0][3]
    d = dev_streams[0][3]
    h, w = f["depth"].shape
    engines[0].integrate_device(d["rgb"].data_ptr(), d["depth"].data_ptr(), d["ht"].data_ptr(),
                                d["lt"].data_ptr(), h, w, 4.0, f["intrinsics"], f["pose"])
    oracles[0].integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    if members > 1:
        rest = ratsdf.Group(engines[1:])
        rest.integrate_device_batch(_group_batch(rest, streams[1:], dev_streams[1:], 3, 4))
        rest.close()
        for s in range(1, members):
            f = streams[s][3]
            oracles[s].integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    # members now have different frame parities inside the group
    group.integrate_device_batch(_group_batch(group, streams, dev_streams, 4, 5))
    group.integrate_device_batch(_group_batch(group, streams, dev_streams, 5, 7))  # back to back
    for s in range(members):
        for f in streams[s][4:7]:
            oracles[s].integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(engines[s], oracles[s])   # settles the member on its own stream
        assert_maps_equal(engines[s], oracles[s])
        c = engines[s].pipeline_counters()
        assert sum(c.values()) == n and (c["front_tail"] >= n - 1 if front_tail == "1" else c["front_tail"] == 0), c
    group.close()
