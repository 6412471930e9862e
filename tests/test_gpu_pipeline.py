"""Frames issued back to back (no query in between) take the pipelined path of the HIP engine: the
carve pass of frame f is finished inside frame f+1's launches and, for batches, the candidate pass of
frame f+1 rides in frame f's launches.  Results must be those of the frame-at-a-time oracle:
directory, free list and pool indices bit-exact, voxels within tolerance, totals equal."""
import numpy as np
import pytest
import torch

from parity import assert_maps_equal, assert_stats_equal
from ratsdf import synthetic

pytestmark = pytest.mark.gpu


def device_frames(frames, semantic=True):
    dev = torch.device("cuda", 0)
    out = []
    for f in frames:
        t = dict(rgb=torch.from_numpy(f["rgb"]).to(dev), depth=torch.from_numpy(f["depth"]).to(dev))
        if semantic and f["ht"] is not None:
            t["ht"] = torch.from_numpy(f["ht"]).to(dev)
            t["lt"] = torch.from_numpy(f["lt"]).to(dev)
        out.append(t)
    torch.cuda.synchronize()
    return out


def make_batch(gpu, frames, dev, lo, hi, md):
    h, w = frames[0]["depth"].shape
    sem = "ht" in dev[0]
    ptr = lambda key: [dev[i][key].data_ptr() for i in range(lo, hi)]
    return gpu.make_batch(ptr("rgb"), ptr("depth"), ptr("ht") if sem else None,
                          ptr("lt") if sem else None, h, w, md,
                          [frames[i]["intrinsics"] for i in range(lo, hi)],
                          [frames[i]["pose"] for i in range(lo, hi)])


def oracle_run(cpu, frames, md):
    for f in frames:
        cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])


def check_totals(gpu, cpu):
    tg, tc = gpu.totals(), cpu.totals()
    assert tg == tc, (tg, tc)


@pytest.mark.parametrize("graph", ["1", "0"])
@pytest.mark.parametrize("chunks", [(12,), (1, 2, 3, 6), (5, 7), (4, 4, 4)])
def test_batches_match_oracle(chunks, graph, monkeypatch, make_engine, make_oracle):
    """(graph = "1", the default: a batch of n >= 2 frames is ONE replay of a HIP graph captured for (image size,
    n) -- (4, 4, 4) replays the same graph three times with different frames; "0": every frame launched by itself)"""
    vs, md = 0.02, 4.0
    monkeypatch.setenv("RATSDF_GRAPH", graph)
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    monkeypatch.delenv("RATSDF_GRAPH")
    frames = synthetic.stream("room", sum(chunks), scale=0.25, noise=True, holes=True)
    dev = device_frames(frames)
    lo = 0
    for n in chunks:  # a query between batches, none inside
        gpu.integrate_device_batch(make_batch(gpu, frames, dev, lo, lo + n, md))
        oracle_run(cpu, frames[lo:lo + n], md)
        lo += n
        assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)
    assert_stats_equal(gpu, cpu)


def test_resolver_and_shared_claim_pass_in_one_frame(make_engine, make_oracle):
    """A frame with chained-bucket requests AND more than 4 096 ordinary ones: the resolver (workgroup 0 of
    k_integrate) resets claims of buckets it locks, and the pass over the requests is shared with the seven
    workgroups beside it, one per XCD -- they must see those resets (write-through stores).  A
    32 768-bucket directory under full-resolution views 90 degrees apart: the second view allocates 5 036
    blocks, some of them in buckets the first view filled."""
    vs, md = 0.004, 4.0
    kw = dict(bucket_bits=15)
    gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, threads=16, **kw)
    seen = []
    for i in (0, 90):
        f = synthetic.frame("room", i, noise=True, holes=True)
        d = device_frames([f])[0]
        h, w = f["depth"].shape
        gpu.integrate_device(d["rgb"].data_ptr(), d["depth"].data_ptr(), d["ht"].data_ptr(), d["lt"].data_ptr(),
                             h, w, md, f["intrinsics"], f["pose"])
        cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
        assert_maps_equal(gpu, cpu)
        s = cpu.last_frame_stats()
        seen.append((s["slow_requests"], s["allocated_blocks"]))
    # (allocated blocks are a lower bound of the frame's ordinary requests)
    assert any(slow > 0 and alloc > 4096 for slow, alloc in seen), seen


def test_single_frame_calls_without_queries(make_engine, make_oracle):
    """ratsdf_integrate_device frame after frame: no look-ahead, but the deferred carve tail."""
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    frames = synthetic.stream("sphere", 10, scale=0.25)
    dev = device_frames(frames)
    h, w = frames[0]["depth"].shape
    for f, d in zip(frames, dev):
        gpu.integrate_device(d["rgb"].data_ptr(), d["depth"].data_ptr(), d["ht"].data_ptr(),
                             d["lt"].data_ptr(), h, w, md, f["intrinsics"], f["pose"])
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)


@pytest.mark.parametrize("bucket_bits", [9, 11])
def test_chained_buckets_in_the_pipeline(bucket_bits, make_engine, make_oracle):
    """A tiny directory: most buckets are full or chained, so allocation goes through the resolver
    and the carve pass through head / chain deletes (resolved by the next frame's first launch)."""
    vs, md = 0.02, 4.0
    kw = dict(bucket_bits=bucket_bits, block_bits=13)
    gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, **kw)
    frames = synthetic.stream("room", 16, scale=0.25, noise=True)
    frames = frames + frames[::-1][:8]
    dev = device_frames(frames)
    slow_seen = 0
    lo = 0
    for n in (8, 8, 8):
        gpu.integrate_device_batch(make_batch(gpu, frames, dev, lo, lo + n, md))
        oracle_run(cpu, frames[lo:lo + n], md)
        lo += n
        assert_maps_equal(gpu, cpu)
        assert_stats_equal(gpu, cpu)
        slow_seen += cpu.last_frame_stats()["slow_requests"]
    check_totals(gpu, cpu)
    assert slow_seen > 0  # the configuration really exercises the chained-bucket paths


def test_more_chained_requests_than_the_lds_resolver_takes(make_engine, make_oracle):
    """A 2 048-bucket directory that fills up while the camera jumps 60 degrees a frame: every new block asks
    from several candidate workgroups and most buckets are full or chained, so frames file 200 - 1 300
    chained-bucket requests.  Up to 512: keys, lock set and stepped replay in the serial workgroup's LDS,
    order by counting; up to 1 024: the same with a bitonic sort (resolve_slow_requests<true>); beyond, keys
    and locks in device memory and one thread replays (<false>).  All against the oracle, frame by frame."""
    import torch
    vs, md = 0.01, 4.0
    kw = dict(bucket_bits=11, block_bits=13)
    gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, threads=8, **kw)
    dev = torch.device("cuda", 0)
    seen = []
    for i in (0, 60, 120, 180, 240, 300, 30, 90, 150, 210):
        f = synthetic.frame("room", i, scale=0.5, noise=True)
        d = [torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")]
        h, w = f["depth"].shape
        gpu.integrate_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), h, w, md,
                             f["intrinsics"], f["pose"])
        cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
        assert_maps_equal(gpu, cpu)
        seen.append(gpu.last_frame_stats()["slow_requests"])
    check_totals(gpu, cpu)
    assert any(0 < n <= 512 for n in seen) and any(512 < n <= 1024 for n in seen) and max(seen) > 1024, seen


def test_large_map_in_the_default_directory(make_engine, make_oracle):
    """1.25 mm voxels at 640x480: ~22 k blocks in the DEFAULT 2^21-bucket
    directory.  The first frame has far more than 2048 allocation requests (the serial role's general
    paths inside k_integrate, scratch in device memory); from then on a map of this size has full
    buckets, so later frames file chained-bucket requests (resolver with LDS keys, then the ordinary
    frame; voxel update running beside it) -- the regime of bench.py --config bigmap."""
    vs, md = 0.00125, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    frames = synthetic.stream("room", 6, noise=True, holes=True)
    dev = device_frames(frames)
    slow_seen = 0
    lo = 0
    for n in (1, 5):
        gpu.integrate_device_batch(make_batch(gpu, frames, dev, lo, lo + n, md))
        for f in frames[lo:lo + n]:
            cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
            slow_seen += cpu.last_frame_stats()["slow_requests"]
        lo += n
        assert_maps_equal(gpu, cpu)
        assert_stats_equal(gpu, cpu)
    check_totals(gpu, cpu)
    assert cpu.num_active_blocks() > 20000 and slow_seen > 100, (cpu.num_active_blocks(), slow_seen)


def test_no_semantics_batch(make_engine, make_oracle):
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    frames = synthetic.stream("wall", 6, scale=0.25, semantic=False)
    dev = device_frames(frames, semantic=False)
    gpu.integrate_device_batch(make_batch(gpu, frames, dev, 0, 6, md))
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)


def test_full_size_batch_640x480(make_engine, make_oracle):
    """BASELINE-sized frames through the batched path (the path bench.py times)."""
    vs, md = 0.005, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    frames = synthetic.stream("room", 6, scale=1.0, noise=True, holes=True)
    dev = device_frames(frames)
    gpu.integrate_device_batch(make_batch(gpu, frames, dev, 0, 6, md))
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)


@pytest.mark.parametrize("seed,bucket_bits", [(0, 0), (1, 10), (2, 0), (3, 12)])
def test_random_interleaving_of_entry_points(seed, bucket_bits, make_engine, make_oracle):
    """Frames (single and batched), queries, statistics reads and the allocation / deletion hooks in a
    random order: whatever the engine still owes from the last frame (deferred carve tail,
    look-ahead state) must be settled correctly by whichever entry point comes next."""
    rng = np.random.default_rng(100 + seed)
    vs, md = 0.02, 4.0
    kw = dict(bucket_bits=bucket_bits, block_bits=13) if bucket_bits else {}
    gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, **kw)
    frames = synthetic.stream("room", 24, scale=0.25, noise=True, holes=True)
    dev = device_frames(frames)
    h, w = frames[0]["depth"].shape
    at = 0
    for step in range(14):
        op = rng.integers(0, 6)
        if op <= 1 and at < len(frames):                     # a batch of 1..5 frames
            n = int(min(rng.integers(1, 6), len(frames) - at))
            gpu.integrate_device_batch(make_batch(gpu, frames, dev, at, at + n, md))
            oracle_run(cpu, frames[at:at + n], md)
            at += n
        elif op == 2 and at < len(frames):                   # a single device frame
            f, d = frames[at], dev[at]
            gpu.integrate_device(d["rgb"].data_ptr(), d["depth"].data_ptr(), d["ht"].data_ptr(),
                                 d["lt"].data_ptr(), h, w, md, f["intrinsics"], f["pose"])
            oracle_run(cpu, [f], md)
            at += 1
        elif op == 3:                                        # allocation hook in between
            pos = rng.integers(-40, 40, size=(int(rng.integers(1, 30)), 3)).astype(np.int16)
            gpu.test_allocate(pos)
            cpu.test_allocate(pos)
        elif op == 4:                                        # deletion hook: some existing blocks
            ei, bl = cpu.dump_directory()
            if len(bl):
                pick = rng.choice(len(bl), size=min(len(bl), int(rng.integers(1, 20))), replace=False)
                pos = np.stack([bl["x"][pick], bl["y"][pick], bl["z"][pick]], axis=1).astype(np.int16)
                gpu.test_delete(pos)
                cpu.test_delete(pos)
        else:                                                # reads
            assert gpu.num_active_blocks() == cpu.num_active_blocks()
            assert_stats_equal(gpu, cpu)
        if step % 3 == 2:
            assert_maps_equal(gpu, cpu)
    assert_maps_equal(gpu, cpu)


@pytest.mark.parametrize("n,semantic", [(3, True), (11, True), (10, False)])
def test_host_image_batch(n, semantic, make_engine, make_oracle):
    """ratsdf_integrate_batch: n frames from host memory in one call (more frames than staging slots
    in the second case) = n blocking ratsdf_integrate calls."""
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    frames = synthetic.stream("room", n, scale=0.25, noise=True, holes=True, semantic=semantic)
    gpu.integrate_batch(frames, md)
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)
    # and the CPU restatement of the same entry point
    cpu2 = make_oracle(vs, 6 * vs)
    cpu2.integrate_batch(frames, md)
    assert_maps_equal(cpu2, cpu)


def test_pool_exhaustion_inside_a_batch(make_engine, make_oracle):
    """A pool that is too small for the scene: the batch reports POOL_EXHAUSTED (sticky) like the
    frame-at-a-time oracle, and both keep the same directory (the insertions that found no block
    simply did not happen, voxel_mem.cu:39)."""
    from ratsdf import RatsdfError
    vs, md = 0.02, 4.0
    kw = dict(block_bits=5, bucket_bits=12)  # 32 blocks; the first frame alone asks for ~70
    gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, **kw)
    frames = synthetic.stream("room", 4, scale=0.25)
    dev = device_frames(frames)
    gpu.integrate_device_batch(make_batch(gpu, frames, dev, 0, 4, md))
    with pytest.raises(RatsdfError) as ei:
        gpu.synchronize()
    assert ei.value.status == 3
    status = []
    for f in frames:
        try:
            cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
        except RatsdfError as err:
            status.append(err.status)
    assert status and set(status) == {3}
    assert gpu.num_active_blocks() == cpu.num_active_blocks() <= 32
    from parity import assert_directory_equal, assert_heap_equal
    assert_directory_equal(gpu, cpu)
    assert_heap_equal(gpu, cpu)


def test_large_image_batch_1080p(make_engine, make_oracle):
    """Three 1920x1080 frames (the reference's maximum image) through the batched host-image entry
    point: 8 160 candidate tiles per frame, several consumer workgroups per candidate list, the
    8 192-workgroup grid of k_integrate."""
    vs, md = 0.01, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    h, w = 1080, 1920
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    rng = np.random.default_rng(31)
    frames = []
    for i in range(3):
        d = (1.2 + 0.3 * np.sin((xx + 40 * i) / 211.0) * np.cos(yy / 173.0)).astype(np.float32)
        frames.append(dict(rgb=rng.integers(0, 256, (h, w, 3), dtype=np.uint8), depth=d,
                           ht=np.clip(0.5 + 0.4 * np.sin(xx / 37) * np.cos(yy / 29), 0.01, 0.99).astype(np.float32),
                           lt=None, intrinsics=(1400.0, 1400.0, 959.5, 539.5),
                           pose=(0.0, 0.0, 0.0, 1.0, 0.01 * i, 0.0, 0.0)))
    for f in frames:
        f["lt"] = (1 - f["ht"]).astype(np.float32)
    gpu.integrate_batch(frames, md)
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)
    assert cpu.last_frame_stats()["visible_blocks"] > 500


def test_many_deletes_take_the_bitmap_path(make_engine, make_oracle):
    """More than 2 048 deletes in one pass (entry-indexed bitmap + popcount prefix instead of the LDS
    list): through the deletion hook, and through frames whose surface moves away so that thousands of
    blocks are carved at once -- once followed by a query (k_settle), once by another frame."""
    from parity import assert_directory_equal, assert_heap_equal
    rng = np.random.default_rng(11)
    gpu, cpu = make_engine(0.01, 0.06), make_oracle(0.01, 0.06)
    pos = np.unique(rng.integers(-300, 300, size=(6000, 3)).astype(np.int16), axis=0)
    for e in (gpu, cpu):
        for _ in range(3):  # one insertion per bucket and pass
            e.test_allocate(pos)
    assert gpu.num_active_blocks() == cpu.num_active_blocks() > 5000
    for e in (gpu, cpu):
        e.test_delete(pos[:4000])
    assert_directory_equal(gpu, cpu)
    assert_heap_equal(gpu, cpu)

    # frames: thousands of blocks sit in the view frustum with their initial values (allocation hook);
    # a frame without any valid depth updates none of them, so all of them are carved at once
    vs, md = 0.005, 4.0
    h, w = 240, 320
    intr = (285.8, 285.8, 159.5, 119.5)
    grid = np.array([(x, y, z) for z in range(12, 38) for y in range(-6, 7) for x in range(-8, 9)],
                    dtype=np.int16)
    def blind(seed):
        r = np.random.default_rng(seed)
        return dict(rgb=r.integers(0, 256, (h, w, 3), dtype=np.uint8), depth=np.zeros((h, w), dtype=np.float32),
                    ht=np.full((h, w), 0.7, dtype=np.float32), lt=np.full((h, w), 0.3, dtype=np.float32),
                    intrinsics=intr, pose=(0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0))
    for follow in ("query", "frame"):
        gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
        for e in (gpu, cpu):
            for _ in range(3):
                e.test_allocate(grid)
        n0 = cpu.num_active_blocks()
        assert n0 > 5000 and gpu.num_active_blocks() == n0
        seq = [blind(1), blind(2)] if follow == "frame" else [blind(1)]
        dev = device_frames(seq)
        gpu.integrate_device_batch(make_batch(gpu, seq, dev, 0, len(seq), md))
        deleted = []
        for f in seq:
            oracle_run(cpu, [f], md)
            deleted.append(cpu.last_frame_stats()["deleted_blocks"])
        assert deleted[0] > 2048, deleted
        assert_maps_equal(gpu, cpu)
        check_totals(gpu, cpu)


def test_bounded_soak_on_a_small_directory(make_engine, make_oracle):
    """tools/soak.py in the suite, bounded: ~200 frames in random batch sizes through the pipelined engine
    (look-ahead candidate pass, serial role and carve tail inside the launches) on a 512-bucket directory
    -- chained buckets, head / chain deletes, the resolver and pool reuse in almost every frame -- against
    the frame-at-a-time oracle, compared every ~50 frames.  This is also the standing stress of the
    fence-free hand-offs."""
    import torch
    kw = dict(bucket_bits=9, block_bits=14)
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, threads=8, **kw)
    dev = torch.device("cuda", 0)
    base = synthetic.stream("room", 40, scale=0.25, noise=True, holes=True) + \
        synthetic.stream("sphere", 25, scale=0.25, noise=True)
    seq = base + base[::-1]
    d = [{k: torch.from_numpy(f[k]).to(dev) for k in ("rgb", "depth", "ht", "lt")} for f in seq]
    h, w = seq[0]["depth"].shape
    rng = np.random.default_rng(5)
    at, slow, checked = 0, 0, 0
    while at < 200:
        c = int(rng.integers(1, 17))
        idx = [(at + j) % len(seq) for j in range(c)]
        b = gpu.make_batch([d[i]["rgb"].data_ptr() for i in idx], [d[i]["depth"].data_ptr() for i in idx],
                           [d[i]["ht"].data_ptr() for i in idx], [d[i]["lt"].data_ptr() for i in idx], h, w, md,
                           [seq[i]["intrinsics"] for i in idx], [seq[i]["pose"] for i in idx])
        gpu.integrate_device_batch(b)
        for i in idx:
            f = seq[i]
            cpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
            slow += cpu.last_frame_stats()["slow_requests"]
        at += c
        if at // 50 != (at - c) // 50:
            assert_maps_equal(gpu, cpu)
            checked += 1
    assert_maps_equal(gpu, cpu)
    assert gpu.totals() == cpu.totals()
    assert slow > 5000 and checked >= 3, (slow, checked)   # the chained-bucket paths really ran


def test_directory_delta_exchange_on_the_device(make_engine):
    """multi.DirectoryDeltaExchange as bench.py --gpus N builds it (engine-sized buffers on the GPU, export
    on the engine's stream, fixed-shape delta on torch's stream, no host read on the exchange path), with a
    single rank: the replica built from deltas equals the engine's directory at every step."""
    from ratsdf import multi
    dev = torch.device("cuda", 0)
    vs = 0.02
    eng = make_engine(vs, 6 * vs)
    frames = synthetic.stream("room", 12, scale=0.25, noise=True, holes=True)
    ex = multi.DirectoryDeltaExchange(engine=eng, device=dev, delta_capacity=4096)
    sent = []
    for i, f in enumerate(frames):
        eng.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        if i % 3 == 2:
            ex.fill_from_engine_device(eng)
            ex.all_gather()
            _, want = eng.dump_directory()
            got = ex.result()[0]
            key = lambda b: np.lexsort((b["z"], b["y"], b["x"]))
            assert len(got) == len(want)
            assert np.array_equal(got[key(got)], want[key(want)])
            sent.append((sum(ex.last_sent), len(want)))
    assert sent[0][0] == sent[0][1] and all(0 < d < n for d, n in sent[1:]), sent
    eng.synchronize()   # the export never touches the engine's sticky error


def test_host_frames_of_alternating_sizes_without_waiting(make_engine, make_oracle):
    """ratsdf_integrate does not wait for its frame (the images sit in a slot of the page-locked staging ring when
    it returns).  The slots' stride belongs to the ring, not to the call: a smaller image after a larger one lands
    in the slot its events guard, not inside the memory of other slots that earlier, still running frames read
    (ADVICE r4).  40 calls alternating between two image sizes (the ring has 16 slots), one query at the end."""
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    big = synthetic.stream("room", 20, scale=0.5, noise=True, holes=True)
    small = synthetic.stream("room", 20, scale=0.25, noise=True, holes=True)
    assert big[0]["depth"].shape != small[0]["depth"].shape
    order = [f for pair in zip(big, small) for f in pair]
    for f in order:
        gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
    oracle_run(cpu, order, md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)
    # ... and the batch entry point right behind frames of the other size that are still in flight
    for f in small[:3]:
        gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
    gpu.integrate_batch(big[:5], md)
    oracle_run(cpu, small[:3] + big[:5], md)
    assert_maps_equal(gpu, cpu)


def test_pinned_batches_upload_runs_of_side_by_side_frames(make_engine, make_oracle):
    """ratsdf_integrate_batch(pinned): frames whose page-locked blocks lie side by side at the staging ring's stride
    (16 bytes per pixel) go up several per copy; blocks elsewhere, frames without semantics and a mixture of both
    take the per-frame copies.  40 frames per call: the 16-slot ring wraps twice.  Same map as the oracle's."""
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    frames = synthetic.stream("room", 40, scale=0.25, noise=True, holes=True)
    h, w = frames[0]["depth"].shape
    npx = h * w
    arena = gpu.host_alloc((len(frames) * npx * 16,), np.uint8)

    def views(blk, f):
        g = dict(f)
        g["depth"] = blk[:npx * 4].view(np.float32).reshape(h, w)
        g["ht"] = blk[npx * 4:npx * 8].view(np.float32).reshape(h, w)
        g["lt"] = blk[npx * 8:npx * 12].view(np.float32).reshape(h, w)
        g["rgb"] = blk[npx * 12:npx * 15].reshape(h, w, 3)
        for k in ("rgb", "depth", "ht", "lt"):
            g[k][...] = f[k]
        return g
    side_by_side = [views(arena[i * npx * 16:(i + 1) * npx * 16], f) for i, f in enumerate(frames)]
    gpu.integrate_batch(side_by_side, md, pinned=True)
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    # every third block out of place (reversed pairs break the runs), then a call without semantics
    order = list(range(len(frames)))
    for i in range(0, len(order) - 1, 3):
        order[i], order[i + 1] = order[i + 1], order[i]
    shuffled = [views(arena[j * npx * 16:(j + 1) * npx * 16], frames[i]) for i, j in enumerate(order)]
    gpu.integrate_batch(shuffled, md, pinned=True)
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    nosem = [dict(g, ht=None, lt=None) for g in side_by_side[:20]]
    for i, g in enumerate(nosem):   # (the views were overwritten by `shuffled`: fill them again)
        for k in ("rgb", "depth"):
            g[k][...] = frames[i][k]
    gpu.integrate_batch(nosem, md, pinned=True)
    oracle_run(cpu, [dict(f, ht=None, lt=None) for f in frames[:20]], md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)
    gpu.host_free(arena)


def test_visible_list_over_more_pool_slots_than_one_round_takes(make_engine, make_oracle):
    """The frame's visible list is a scan of the pool slots that have ever been in use (Table::active, round 5):
    128 workgroups x 256 lanes x 2 slots = 65 536 per round.  Here 80 k blocks far from the camera exist before the
    frames come, so the frames' own blocks live in pool slots that only the SECOND round reaches; a fifth of the far
    blocks is deleted again (holes in the slot array, their pool indices re-used by the frames' allocations).  The
    frames' visible-block counts, directory, free list and a sample of the voxels must be the oracle's."""
    from parity import assert_directory_equal, assert_heap_equal, assert_voxels_close
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=8)
    rng = np.random.default_rng(5)
    pos = np.unique(rng.integers(1000, 4000, size=(90000, 3)).astype(np.int16), axis=0)   # 160 m and more away
    for lo in range(0, len(pos), 16384):
        for _ in range(2):                # (one insertion per bucket per pass)
            gpu.test_allocate(pos[lo:lo + 16384])
            cpu.test_allocate(pos[lo:lo + 16384])
    n_far = gpu.num_active_blocks()
    assert n_far == cpu.num_active_blocks() and n_far > 80000
    frames = synthetic.stream("room", 5, scale=0.25, noise=True, holes=True)
    dev = device_frames(frames)
    gpu.integrate_device_batch(make_batch(gpu, frames, dev, 0, 2, md))
    oracle_run(cpu, frames[:2], md)
    assert_stats_equal(gpu, cpu)
    gpu.test_delete(pos[::5])
    cpu.test_delete(pos[::5])
    gpu.integrate_device_batch(make_batch(gpu, frames, dev, 2, 5, md))
    oracle_run(cpu, frames[2:], md)
    assert_stats_equal(gpu, cpu)
    assert gpu.last_frame_stats()["visible_blocks"] > 50   # (16 cm blocks: the whole view is ~80 of them)
    _, blocks = assert_directory_equal(gpu, cpu)
    assert_heap_equal(gpu, cpu)
    near = blocks["idx"][np.abs(blocks["x"].astype(np.int32)) < 500]
    assert len(near) > 50
    assert_voxels_close(gpu, cpu, near)
    assert_voxels_close(gpu, cpu, blocks["idx"][::97])


def test_tsdf_only_frames_skip_the_probability_only_while_it_is_untouched(make_engine, make_oracle):
    """A map that has never seen ht / lt holds 0.5 in every voxel and a TSDF-only frame leaves it there, so the update
    neither loads nor stores the probability of existing blocks (FrameParams::segm_live, SURVEY 8d's TSDF-only bytes).
    The moment a frame with semantics arrives -- or blocks are imported with their probabilities -- every later frame,
    TSDF-only ones included, takes the full update again.  Against the oracle over TSDF-only -> semantic -> TSDF-only
    batches (the switch happens INSIDE a batch), with the probability exactly 0.5 before the switch."""
    vs, md = 0.02, 4.0
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    frames = synthetic.stream("room", 12, scale=0.25, noise=True, holes=True)
    nosem = [dict(f, ht=None, lt=None) for f in frames]
    seq = nosem[:4] + frames[4:8] + nosem[8:12]
    dev_all = device_frames(frames)
    for lo, hi in ((0, 3), (3, 6), (6, 12)):    # batch 2 = one TSDF-only frame, then two with semantics
        chunk = seq[lo:hi]
        for i, f in enumerate(chunk):          # (a batch has one semantics setting: frame by frame where it is mixed)
            d = dev_all[lo + i]
            sem = f["ht"] is not None
            gpu.integrate_device(d["rgb"].data_ptr(), d["depth"].data_ptr(), d["ht"].data_ptr() if sem else 0,
                                 d["lt"].data_ptr() if sem else 0, *f["depth"].shape, md, f["intrinsics"], f["pose"])
        oracle_run(cpu, chunk, md)
        assert_maps_equal(gpu, cpu)
        if hi <= 3:
            _, blocks = gpu.dump_directory()
            _, _, p = gpu.dump_voxels(blocks["idx"])
            assert p.size and np.all(p == np.float32(0.5))
    check_totals(gpu, cpu)
    # batches of TSDF-only frames on a fresh map, then imported blocks switch the full update on
    gpu2, cpu2 = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    dev = device_frames(frames, semantic=False)
    gpu2.integrate_device_batch(make_batch(gpu2, nosem, dev, 0, 6, md))
    oracle_run(cpu2, nosem[:6], md)
    assert_maps_equal(gpu2, cpu2)
    far = np.array([[300, 300, 300], [301, 300, 300]], dtype=np.int16)
    t = np.full((2, 512), 0.25, np.float32)
    pr = np.full((2, 512), 0.8, np.float32)
    c = np.zeros((2, 512), dtype=[("r", "u1"), ("g", "u1"), ("b", "u1"), ("weight", "u1")])
    c["weight"] = 3
    for e in (gpu2, cpu2):
        e.import_blocks(far, t, c, pr)
    gpu2.integrate_device_batch(make_batch(gpu2, nosem, dev, 6, 12, md))
    oracle_run(cpu2, nosem[6:12], md)
    assert_maps_equal(gpu2, cpu2)


@pytest.mark.parametrize("mode", ["single frames", "one batch"])
def test_more_candidates_than_a_workgroups_lds_set_holds(mode, make_engine, make_oracle):
    """A 16x16-pixel patch that asks for more than 512 distinct blocks: the candidates that find the workgroup's LDS set
    full leave by the overflow paths -- straight into the frame's global list in the look-ahead form (cand_append), a
    request filed on the spot in the form that runs inside k_front (cand_inline_role: single frames, the first frame of
    a batch).  A coarse image on 2 mm voxels: neighbouring pixels' rays are 17 voxels apart, so every pixel brings its own
    four or five blocks."""
    vs, trunc, md = 0.002, 0.03, 4.0
    kw = dict(block_bits=16)
    gpu, cpu = make_engine(vs, trunc, **kw), make_oracle(vs, trunc, threads=8, **kw)
    frames = synthetic.stream("room", 3, scale=0.1)
    dev = device_frames(frames)
    h, w = frames[0]["depth"].shape
    if mode == "single frames":
        for f, d in zip(frames, dev):
            gpu.integrate_device(d["rgb"].data_ptr(), d["depth"].data_ptr(), d["ht"].data_ptr(), d["lt"].data_ptr(), h, w,
                                 md, f["intrinsics"], f["pose"])
    else:
        gpu.integrate_device_batch(make_batch(gpu, frames, dev, 0, len(frames), md))
    oracle_run(cpu, frames, md)
    assert_maps_equal(gpu, cpu)
    check_totals(gpu, cpu)
    # the premise: far more blocks per 16x16 patch than the 512 slots of a workgroup's set
    patches = ((h + 15) // 16) * ((w + 15) // 16)
    assert cpu.num_active_blocks() > 512 * patches, (cpu.num_active_blocks(), patches)
