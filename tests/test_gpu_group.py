"""Groups (ratsdf_group_*): several engines of one GPU stepped by shared launches.  Every member's map
must be exactly what the member alone -- and the CPU oracle -- produce from its own stream."""
import numpy as np
import pytest

from parity import assert_maps_equal, assert_stats_equal
from ratsdf import synthetic

pytestmark = pytest.mark.gpu


def _upload(frames):
    import torch
    dev = torch.device("cuda", 0)
    return [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in frames]


def _group_batch(group, streams, dev_streams, lo, hi, md=4.0):
    """frames lo..hi-1 of every member stream as one group batch"""
    rows = lambda key: [[dev_streams[s][f][key].data_ptr() for s in range(len(streams))]
                        for f in range(lo, hi)]
    h, w = streams[0][0]["depth"].shape
    return group.make_batch(rows("rgb"), rows("depth"), rows("ht"), rows("lt"), h, w, md,
                            [[streams[s][f]["intrinsics"] for s in range(len(streams))] for f in range(lo, hi)],
                            [[streams[s][f]["pose"] for s in range(len(streams))] for f in range(lo, hi)])


@pytest.mark.parametrize("members", [1, 3])
def test_group_matches_members_alone_and_oracle(members, make_engine, make_oracle):
    import ratsdf
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
        assert sum(c.values()) == n and c["front_tail"] == 0, c
    group.close()


def test_group_rejects_mismatched_members(make_engine):
    import ratsdf
    a, b = make_engine(0.02, 0.12), make_engine(0.01, 0.06)
    with pytest.raises(ratsdf.RatsdfError):
        ratsdf.Group([a, b])
    with pytest.raises(ratsdf.RatsdfError):
        ratsdf.Group([a, a])


def test_group_full_resolution_two_streams(make_engine, make_oracle):
    """BASELINE configs[4] geometry on one device: two 640x480 / 5 mm streams, three frames each."""
    import ratsdf
    vs = 0.005
    streams = [[synthetic.frame("room", 45 * s + i, noise=True, holes=True) for i in range(3)]
               for s in range(2)]
    dev_streams = [_upload(fr) for fr in streams]
    engines = [make_engine(vs, 6 * vs) for _ in range(2)]
    group = ratsdf.Group(engines)
    group.integrate_device_batch(_group_batch(group, streams, dev_streams, 0, 3))
    group.synchronize()
    for s in range(2):
        cpu = make_oracle(vs, 6 * vs, threads=16)
        for f in streams[s]:
            cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(engines[s], cpu)
        assert_maps_equal(engines[s], cpu)
        cpu.close()
    group.close()


def test_group_of_four_full_resolution_streams(make_engine, make_oracle):
    """What bench.py's `multi_stream` leg times (S = 4 members, 640x480 / 5 mm, BASELINE configs[4] on one
    device): every member against the oracle after one batch of four frames."""
    import ratsdf
    vs, S, n = 0.005, 4, 4
    streams = [[synthetic.frame("room", 45 * s + i, noise=True, holes=True) for i in range(n)]
               for s in range(S)]
    dev_streams = [_upload(fr) for fr in streams]
    engines = [make_engine(vs, 6 * vs) for _ in range(S)]
    group = ratsdf.Group(engines)
    group.integrate_device_batch(_group_batch(group, streams, dev_streams, 0, n))
    group.synchronize()
    for s in range(S):
        cpu = make_oracle(vs, 6 * vs, threads=16)
        for f in streams[s]:
            cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(engines[s], cpu)
        assert_maps_equal(engines[s], cpu)
        cpu.close()
    group.close()


def test_group_of_eight_streams(make_engine, make_oracle):
    """BASELINE configs[4]'s stream count (8) in one group: small frames, mixed scenes, two batches."""
    import ratsdf
    vs, S, n = 0.02, 8, 6
    scenes = ["room", "sphere", "wall", "room", "sphere", "room", "wall", "room"]
    streams = [[synthetic.frame(scenes[s], 11 * s + i, scale=0.25, noise=(s % 2 == 0), holes=(s % 3 == 0))
                for i in range(n)] for s in range(S)]
    dev_streams = [_upload(fr) for fr in streams]
    engines = [make_engine(vs, 6 * vs) for _ in range(S)]
    oracles = [make_oracle(vs, 6 * vs) for _ in range(S)]
    group = ratsdf.Group(engines)
    for lo, hi in ((0, 2), (2, 6)):
        group.integrate_device_batch(_group_batch(group, streams, dev_streams, lo, hi))
        for s in range(S):
            for f in streams[s][lo:hi]:
                oracles[s].integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
            assert_stats_equal(engines[s], oracles[s])
            assert_maps_equal(engines[s], oracles[s])
    group.close()
