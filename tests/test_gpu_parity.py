"""HIP engine vs CPU oracle on identical seeded frames, through the C ABI (needs a GPU)."""
import numpy as np
import pytest

from parity import assert_maps_equal, assert_stats_equal
from ratsdf import synthetic

pytestmark = pytest.mark.gpu


def run_both(gpu, cpu, frames, md=4.0, check_every=1):
    worst = dict(tsdf=0.0, prob=0.0)
    for i, f in enumerate(frames):
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], md, f["intrinsics"], f["pose"])
        assert_stats_equal(gpu, cpu)
        if (i + 1) % check_every == 0 or i == len(frames) - 1:
            w = assert_maps_equal(gpu, cpu)
            worst = {k: max(worst[k], w[k]) for k in worst}
    return worst


@pytest.mark.parametrize("scene", ["wall", "room", "sphere"])
def test_small_sequences(scene, make_engine, make_oracle):
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    frames = synthetic.stream(scene, 6, scale=0.25)
    worst = run_both(gpu, cpu, frames)
    print(scene, worst)


def test_no_semantics_and_holes(make_engine, make_oracle):
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    frames = synthetic.stream("room", 4, scale=0.25, semantic=False, holes=True, noise=True)
    run_both(gpu, cpu, frames)
    p = gpu.gather_valid_semantic()["prob"]
    assert np.all(p == np.float32(0.5))  # ht = lt = 1 keeps the probability at exactly .5


def test_tum_camera_1cm(make_engine, make_oracle):
    vs = 0.01
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=8)
    frames = synthetic.stream("sphere", 3, cam="tum", scale=0.5)
    run_both(gpu, cpu, frames)


def test_full_resolution_5mm_frame(make_engine, make_oracle):
    """BASELINE configs[1]-sized frame: 640x480, 5 mm voxels, 2 frames of the room scene."""
    vs = 0.005
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    frames = synthetic.stream("room", 2, scale=1.0)
    run_both(gpu, cpu, frames)


def test_query_and_gather_match(make_engine, make_oracle):
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    for f in synthetic.stream("room", 3, scale=0.25):
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    for bounds in [(-1, 1, -1, 1, 0, 2), (-5, 5, -5, 5, -5, 5), (0.1, 0.2, 0.1, 0.2, 0.1, 0.2)]:
        qa, qb = gpu.query(bounds), cpu.query(bounds)
        assert len(qa) == len(qb)
        for k in ("x", "y", "z"):
            assert np.array_equal(qa[k], qb[k])
        assert np.max(np.abs(qa["tsdf"] - qb["tsdf"]), initial=0) <= 1e-4
    ga, gb = gpu.gather_valid(), cpu.gather_valid()
    assert len(ga) == len(gb) and np.array_equal(ga["x"], gb["x"])
    sa, sb = gpu.gather_valid_semantic(), cpu.gather_valid_semantic()
    assert len(sa) == len(sb)
    assert np.max(np.abs(sa["prob"] - sb["prob"]), initial=0) <= 1e-4
    assert np.max(np.abs(sa["tsdf"] - sb["tsdf"]), initial=0) <= 1e-4


@pytest.mark.parametrize("world,slab", [(2, 1), (4, 1)])
def test_sharded_engines_match_sharded_oracle(world, slab, make_engine, make_oracle):
    """Block-ownership sharding (BASELINE config 4): every shard's map is bit-exact vs the oracle
    with the same shard parameters, and the shards partition the blocks."""
    from ratsdf import multi
    vs = 0.02
    frames = synthetic.stream("room", 4, scale=0.25)
    per_rank = []
    for r in range(world):
        kw = dict(shard_rank=r, shard_count=world, shard_slab_bits=slab)
        gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, **kw)
        run_both(gpu, cpu, frames)
        per_rank.append(gpu.dump_directory()[1])
    assert multi.check_sharded_directories(per_rank, slab) == sum(len(b) for b in per_rank)
    assert all(len(b) > 0 for b in per_rank)


def test_export_directory_device(make_engine):
    """ratsdf_export_directory_device writes the same compact directory the host dump returns."""
    import torch
    from ratsdf._abi import BLOCK_DTYPE
    vs = 0.02
    gpu = make_engine(vs, 6 * vs)
    for f in synthetic.stream("room", 2, scale=0.25):
        gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    cap = 4096
    buf = torch.zeros(cap * 3, dtype=torch.int32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    gpu.export_directory_device(buf.data_ptr(), cap, cnt.data_ptr())
    gpu.synchronize()
    _, blocks = gpu.dump_directory()
    n = int(cnt.item())
    assert n == len(blocks)
    got = buf.cpu().numpy()[:n * 3].view(BLOCK_DTYPE)
    assert np.array_equal(got, blocks)


def test_directory_delta_exchange_on_the_device(make_engine):
    """multi.DirectoryDeltaExchange fed by the engine on the device (the path bench.py --gpus N takes
    with RCCL; here one rank): the replica built from deltas is the engine's directory at every step,
    and after the first exchange only what changed is sent."""
    import torch
    from ratsdf import multi
    vs = 0.02
    gpu = make_engine(vs, 6 * vs)
    dx = multi.DirectoryDeltaExchange(capacity=4096, device=torch.device("cuda", 0))
    frames = synthetic.stream("room", 8, scale=0.25, noise=True)
    by_pos = lambda b: b[np.lexsort((b["z"], b["y"], b["x"]))]
    sent = []
    for step in range(4):
        for f in frames[2 * step:2 * step + 2]:
            gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        dx.fill_from_engine_device(gpu)
        dx.all_gather()
        _, blocks = gpu.dump_directory()
        assert np.array_equal(by_pos(dx.result()[0]), by_pos(blocks)), step
        sent.append((sum(dx.last_sent), len(blocks)))
    assert sent[0][0] == sent[0][1] and all(0 < d < 0.7 * n for d, n in sent[1:]), sent


@pytest.mark.parametrize("kw", [dict(), dict(bucket_bits=9, block_bits=14)])
def test_directory_delta_from_the_engines_log(kw, make_engine):
    """The delta the ENGINE keeps (a dirty bit per directory entry + a log of deleted positions,
    ratsdf_export_directory_delta_device) instead of a diff of two sorted whole directories on the caller's side:
    the replica built from those deltas is the engine's directory -- every field, chain links included -- at every
    step, over batched frames (blocks allocated and carved within one interval, positions deleted and inserted
    again) and, in the second case, on a 512-bucket directory where most insertions link chains and most deletes
    unlink them."""
    import torch
    from ratsdf import multi
    vs = 0.02
    gpu = make_engine(vs, 6 * vs, **kw)
    dx = multi.DirectoryDeltaExchange(engine=gpu, device=torch.device("cuda", 0), delta_capacity=8192)
    frames = synthetic.stream("room", 16, scale=0.25, noise=True, holes=True)
    frames = frames + frames[::-1][:8]
    by_pos = lambda b: b[np.lexsort((b["z"], b["y"], b["x"]))]
    sent = []
    at = 0
    for n in (2, 1, 5, 3, 4, 6, 3):
        gpu.integrate_batch(frames[at:at + n], 4.0)
        at += n
        dx.fill_from_engine(gpu)
        dx.all_gather()
        _, blocks = gpu.dump_directory()
        got = dx.result()[0]
        assert len(got) == len(blocks), (at, len(got), len(blocks))
        assert np.array_equal(by_pos(got), by_pos(blocks)), at
        sent.append((sum(dx.last_sent), len(blocks)))
    assert sent[0][0] == sent[0][1] and all(0 < d for d, _ in sent[1:]), sent
    # the hooks that edit the directory outside a frame are seen too
    pos = np.array([[40, 41, 42], [-7, 3, 9], [40, 41, 43]], dtype=np.int16)
    gpu.test_allocate(pos)
    gpu.test_delete(pos[:1])
    dx.fill_from_engine(gpu)
    dx.all_gather()
    _, blocks = gpu.dump_directory()
    assert np.array_equal(by_pos(dx.result()[0]), by_pos(blocks))
    gpu.synchronize()


def test_raycast_matches_oracle(make_engine, make_oracle):
    """TSDFGrid::RayCast / TSDFSystem::Render (voxel_tsdf.cu:278-374): rgba + normal images of a
    virtual view.  Voxel weights must reach 10 before a surface is rendered, so integrate enough
    frames first."""
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=8)
    frames = [synthetic.frame("room", 0, scale=0.25) for _ in range(6)]
    frames += synthetic.stream("room", 4, scale=0.25)
    for f in frames:
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    f = frames[-1]
    h, w = f["depth"].shape
    for pose, k, hh, ww in [(f["pose"], f["intrinsics"], h, w),
                            (synthetic.frame("room", 2, scale=0.25)["pose"], f["intrinsics"], h, w),
                            (f["pose"], tuple(v * 2 for v in f["intrinsics"]), 2 * h, 2 * w)]:
        ga, gn = gpu.raycast(k, hh, ww, pose, 8.0)
        ca, cn = cpu.raycast(k, hh, ww, pose, 8.0)
        assert (ca[..., 3] == 255).mean() > 0.3, "oracle rendered almost nothing"
        for a, b, name in ((ga, ca, "rgba"), (gn, cn, "normal")):
            d = np.abs(a.astype(np.int16) - b.astype(np.int16))
            # identical arithmetic except probability (expf/logf, <= 1e-6) feeding alpha: allow a
            # one-step difference on a handful of pixels
            assert d.max() <= 1, f"{name}: max byte difference {d.max()}"
            assert (d > 0).mean() < 1e-3, f"{name}: {(d > 0).sum()} bytes differ"
        # ratsdf_raycast_rows: any row range is that part of the full rendering, byte for byte
        for r0, r1 in ((0, hh), (5, 22), (hh - 17, hh), (9, 9)):
            ra, rn = gpu.raycast_rows(k, hh, ww, pose, 8.0, r0, r1)
            assert np.array_equal(ra, ga[r0:r1]) and np.array_equal(rn, gn[r0:r1]), (r0, r1)
        # ratsdf_raycast_device: the same images left in device memory (what a renderer that displays from the GPU takes)
        import torch
        d_a = torch.zeros((hh, ww, 4), dtype=torch.uint8, device="cuda")
        d_n = torch.zeros((hh, ww, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        gpu.raycast_device(k, hh, ww, pose, 8.0, d_a.data_ptr(), d_n.data_ptr())
        gpu.synchronize()
        assert np.array_equal(d_a.cpu().numpy(), ga) and np.array_equal(d_n.cpu().numpy(), gn)
    # a far longer ray than the map is deep (most of its samples in empty space) and a map the occupancy filter of the
    # march has to be rebuilt for: delete nothing, integrate more, render again -- still the oracle's images
    for fr in synthetic.stream("sphere", 3, scale=0.25):
        for e in (gpu, cpu):
            e.integrate(fr["rgb"], fr["depth"], fr["ht"], fr["lt"], 4.0, fr["intrinsics"], fr["pose"])
    ga, gn = gpu.raycast(f["intrinsics"], h, w, f["pose"], 40.0)
    ca, cn = cpu.raycast(f["intrinsics"], h, w, f["pose"], 40.0)
    for a, b in ((ga, ca), (gn, cn)):
        d = np.abs(a.astype(np.int16) - b.astype(np.int16))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3


def _frame_like(depth, seed=0, pose=None, intr=(300.0, 300.0, 0.0, 0.0)):
    h, w = depth.shape
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = intr
    if cx == 0.0:
        intr = (fx, fy, (w - 1) / 2.0, (h - 1) / 2.0)
    return dict(rgb=rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), depth=depth.astype(np.float32),
                ht=np.clip(rng.random((h, w)), 0.02, 0.98).astype(np.float32),
                lt=np.clip(rng.random((h, w)), 0.02, 0.98).astype(np.float32),
                pose=pose or (0, 0, 0, 1, 0, 0, 0), intrinsics=intr)


def test_edge_case_inputs(make_engine, make_oracle):
    """Ragged / degenerate frames the reference would accept: odd image sizes, empty and all-invalid
    depth, depth exactly at max_depth, NaN depth, ht = 0 (log -> -inf), a 1x1 image."""
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    rng = np.random.default_rng(3)
    frames = []
    d = np.full((61, 83), 1.5, np.float32)                      # odd, non-multiple-of-64 sizes
    frames.append(_frame_like(d, 1))
    frames.append(_frame_like(np.zeros((61, 83), np.float32), 2))          # nothing valid
    frames.append(_frame_like(np.full((61, 83), 9.0, np.float32), 3))      # everything > max_depth
    d2 = d.copy()
    d2[::2, ::3] = 4.0                                          # exactly max_depth: w_new = 0
    d2[5, 7] = np.nan
    d2[6, 7] = np.inf
    d2[7, 7] = -1.0                                             # negative depth passes the reference test
    frames.append(_frame_like(d2, 4))
    f = _frame_like(d, 5)
    f["ht"][10:20, 10:30] = 0.0                                 # logf(0) = -inf -> pos = 0
    f["lt"][10:20, 10:30] = 1.0
    frames.append(f)
    frames.append(_frame_like(d + rng.normal(0, 0.01, d.shape).astype(np.float32), 6))
    for i, f in enumerate(frames):
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(gpu, cpu)
        ei, bl = gpu.dump_directory()
        eo, bo = cpu.dump_directory()
        assert np.array_equal(ei, eo) and np.array_equal(bl, bo), f"frame {i}"
        tg, cg, pg = gpu.dump_voxels(bl["idx"])
        to, co, po = cpu.dump_voxels(bo["idx"])
        assert np.array_equal(cg, co), f"frame {i}: rgbw"
        assert np.array_equal(np.isnan(tg), np.isnan(to)) and np.array_equal(np.isnan(pg), np.isnan(po))
        assert np.nanmax(np.abs(tg - to), initial=0) <= 1e-4 and np.nanmax(np.abs(pg - po), initial=0) <= 1e-4
    # a different image size on the same engine, then a 1x1 image
    for shape in [(48, 64), (1, 1)]:
        f = _frame_like(np.full(shape, 1.2, np.float32), 9)
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], None, None, 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(gpu, cpu)
    assert_maps_equal(gpu, cpu)


def test_probability_edge_cases(make_engine, make_oracle):
    """The probability update is the one place where the kernel evaluates a different expression of the same
    function (log-odds on v_log / v_exp / v_rcp instead of two exponentials of two logs, voxel_tsdf.cu:241-248):
    its special values must come out as the reference's do.  Regions of the image, over several frames:
      lt = 0            -> neg = 0, p = 1, and p stays 1 under ordinary observations (log(1 - p) = -inf)
      ht = 0            -> p = 0, stays 0
      p = 1, then ht = 0  -> 0 / 0 = NaN, which then propagates
      ht = lt = 0       -> NaN at once
      d = max_depth (w_new = 0) with ht = 0: 0 * -inf = NaN; with ordinary ht / lt: p (almost) unchanged
    NaN in exactly the oracle's voxels, everything else within 1e-4, weights / colours / tsdf as ever."""
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    h, w = 96, 128
    base = np.full((h, w), 1.5, np.float32)

    def frame(seed, edits, depth=None):
        f = _frame_like(base if depth is None else depth, seed)
        for (r0, r1, c0, c1, ht, lt) in edits:
            if ht is not None:
                f["ht"][r0:r1, c0:c1] = ht
            if lt is not None:
                f["lt"][r0:r1, c0:c1] = lt
        return f

    A = (0, 24, 0, 64)        # lt = 0 for three frames, then ordinary
    B = (24, 48, 0, 64)       # ht = 0 for two frames, then ordinary
    C = (48, 72, 0, 64)       # lt = 0 (p -> 1), then ht = 0 (NaN), then ordinary (stays NaN)
    D = (72, 96, 0, 64)       # ht = lt = 0
    E = (0, 48, 64, 128)      # depth = max_depth with ht = 0
    F = (48, 96, 64, 128)     # depth = max_depth with ordinary ht / lt, after two ordinary frames
    d_far = base.copy()
    d_far[E[0]:E[1], E[2]:E[3]] = 4.0
    d_far[F[0]:F[1], F[2]:F[3]] = 4.0
    frames = [
        frame(1, [A + (None, 0.0), B + (0.0, None), C + (None, 0.0), D + (0.0, 0.0)]),
        frame(2, [A + (None, 0.0), B + (0.0, None), C + (0.0, None), D + (0.0, 0.0)]),
        frame(3, [A + (None, 0.0), E + (0.0, None)], depth=d_far),
        frame(4, []),
        frame(5, [E + (0.0, 0.0)], depth=d_far),
        frame(6, []),
    ]
    saw_one = saw_zero = saw_nan = False
    for i, f in enumerate(frames):
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(gpu, cpu)
        ei, bl = gpu.dump_directory()
        eo, bo = cpu.dump_directory()
        assert np.array_equal(ei, eo) and np.array_equal(bl, bo), f"frame {i}"
        tg, cg, pg = gpu.dump_voxels(bl["idx"])
        to, co, po = cpu.dump_voxels(bo["idx"])
        assert np.array_equal(cg, co), f"frame {i}: rgbw"
        assert np.array_equal(tg, to), f"frame {i}: tsdf"
        assert np.array_equal(np.isnan(pg), np.isnan(po)), f"frame {i}: NaN probabilities in different voxels"
        assert np.nanmax(np.abs(pg - po), initial=0) <= 1e-4, f"frame {i}"
        # saturated values are exact, not merely close
        assert np.array_equal(pg == 1.0, po == 1.0) and np.array_equal(pg == 0.0, po == 0.0), f"frame {i}"
        saw_one |= bool((po == 1.0).any())
        saw_zero |= bool((po == 0.0).any())
        saw_nan |= bool(np.isnan(po).any())
    assert saw_one and saw_zero and saw_nan   # the regions really produced the special values
    from parity import assert_heap_equal
    assert_heap_equal(gpu, cpu)


def test_rotated_poses_and_long_motion(make_engine, make_oracle):
    """Poses whose rotation matrix takes the trace <= 0 branches of the matrix->quaternion conversion
    (180 degree turns about each axis) and a 40-frame sphere sequence with carving churn."""
    from ratsdf import pose_from_matrix
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=8)
    d = np.full((60, 80), 1.0, np.float32)
    for k, diag in enumerate([(1, -1, -1), (-1, 1, -1), (-1, -1, 1), (1, 1, 1)]):
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[1, 1], m[2, 2] = diag
        m[:3, 3] = (0.01 * k, -0.02 * k, 0.03)
        f = _frame_like(d, 20 + k, pose=pose_from_matrix(m))
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(gpu, cpu)
    assert_maps_equal(gpu, cpu)
    deleted = 0
    for f in synthetic.stream("sphere", 40, scale=0.25, noise=True, holes=True):
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(gpu, cpu)
        deleted += gpu.last_frame_stats()["deleted_blocks"]
    assert deleted > 0
    assert_maps_equal(gpu, cpu)


def test_max_image_size_1080p(make_engine, make_oracle):
    """The reference's maximum image (MAX_IMG 1920x1080, voxel_tsdf.cu:11-13) at 1 cm voxels."""
    vs = 0.01
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    h, w = 1080, 1920
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    d = (1.2 + 0.3 * np.sin(xx / 211.0) * np.cos(yy / 173.0)).astype(np.float32)
    f = _frame_like(d, 31, intr=(1400.0, 1400.0, 959.5, 539.5))
    for _ in range(2):
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        assert_stats_equal(gpu, cpu)
    assert gpu.last_frame_stats()["visible_blocks"] > 500
    assert_maps_equal(gpu, cpu)


def test_hd_1280x720_2mm_l515(make_engine, make_oracle):
    """BASELINE configs[3] workload on one GPU: 1280x720, 2 mm voxels, L515 intrinsics
    (2x configs/zed_native_l515.yaml:30-33), 3 frames of the room stream with noise and holes."""
    vs = 0.002
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    frames = synthetic.stream("room", 3, cam="l515_720p", noise=True, holes=True)
    assert frames[0]["depth"].shape == (720, 1280)
    worst = run_both(gpu, cpu, frames, check_every=3)
    assert gpu.last_frame_stats()["visible_blocks"] > 5000
    print("hd2mm", worst)


def test_hd_1280x720_2mm_four_subvolumes(make_engine, make_oracle):
    """BASELINE configs[3] split: the same 1280x720 / 2 mm frames through 4 block-ownership shards
    (spatial subvolumes), run one after another on the single GPU; every shard is bit-exact against
    the oracle with the same shard parameters, the shards partition the blocks, and their union is
    the unsharded engine's block set."""
    from ratsdf import multi
    vs = 0.002
    frames = synthetic.stream("room", 2, cam="l515_720p", noise=True, holes=True)
    world, slab = 4, 3
    per_rank = []
    for r in range(world):
        kw = dict(shard_rank=r, shard_count=world, shard_slab_bits=slab)
        gpu, cpu = make_engine(vs, 6 * vs, **kw), make_oracle(vs, 6 * vs, threads=16, **kw)
        run_both(gpu, cpu, frames, check_every=2)
        per_rank.append(gpu.dump_directory()[1])
        gpu.close()
        cpu.close()
    total = multi.check_sharded_directories(per_rank, slab)
    assert all(len(b) > 0 for b in per_rank)
    whole = make_engine(vs, 6 * vs)
    for f in frames:
        whole.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    blocks = whole.dump_directory()[1]
    have = set()
    for b in per_rank:
        have |= set(zip(b["x"].tolist(), b["y"].tolist(), b["z"].tolist()))
    # an unsharded pass can lose an insertion to a bucket-lock collision between blocks of different
    # owners (voxel_hash.cu:67-78: one insertion per bucket per pass); the sharded maps cannot
    want = set(zip(blocks["x"].tolist(), blocks["y"].tolist(), blocks["z"].tolist()))
    assert want <= have and len(have) == total
    # ... and every block only the shards hold shares its home bucket with another requested block
    from kat_cases import ref_hash
    buckets = {}
    for p in have:
        buckets.setdefault(ref_hash(p), []).append(p)
    for p in have - want:
        assert len(buckets[ref_hash(p)]) > 1, f"{p} missing from the unsharded map without a collision"
    assert len(have - want) <= len(want) // 100


def test_tum_640x480_5mm_pair(make_engine, make_oracle):
    """BASELINE configs[2] geometry: TUM intrinsics (configs/TUM_RGBD_rgbd_1.yaml:11-14) at full
    640x480 resolution, 5 mm voxels, semantic labels fused; two consecutive frames."""
    vs = 0.005
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    frames = synthetic.stream("room", 2, cam="tum", noise=True, holes=True)
    assert frames[0]["depth"].shape == (480, 640)
    worst = run_both(gpu, cpu, frames)
    print("tum 5mm", worst)


def test_voxels_in_the_camera_plane(make_engine, make_oracle):
    """pc.z == 0 exactly: x / 0 = +-inf falls outside the image, 0 / 0 = NaN picks pixel (0, 0) because
    CUDA's float -> int conversion of NaN is 0 (voxel_tsdf.cu:196-205).  The kernel's
    shared-reciprocal division does not cover z == 0 and its short pixel pick does not convert NaN to 0:
    this is the frame that takes the plain-division branch."""
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs)
    f0 = synthetic.frame("wall", 0, scale=0.25)          # fronto-parallel wall at z = 2 m, identity rotation
    assert tuple(float(v) for v in f0["pose"][:4]) == (0.0, 0.0, 0.0, 1.0)
    run_both(gpu, cpu, [f0])
    # second frame: the camera moved INTO the wall's blocks, its plane on the voxel plane gz = 100
    # (world z = fl(100 * vs)): with the identity rotation pc.z = fl(gz * vs) + t.z is exactly 0 there
    f1 = dict(f0)
    tz = -float(np.float32(100) * np.float32(vs))
    f1["pose"] = (0.0, 0.0, 0.0, 1.0, 0.0, 0.0, tz)
    f1["depth"] = np.full_like(f0["depth"], 0.05)         # valid everywhere, pixel (0, 0) included
    before = gpu.last_frame_stats()["active_blocks"]
    run_both(gpu, cpu, [f1])
    s = gpu.last_frame_stats()
    # (8 blocks straddle the camera plane; the voxel on the optical axis -- 0 / 0 -- is among the updated)
    assert s["visible_blocks"] > 0 and s["updated_voxels"] > 0 and before > 0


def test_non_finite_camera_parameters_are_refused(make_engine):
    """ratsdf_integrate*: NaN / inf in pose, intrinsics or max depth -> RATSDF_ERR_BAD_ARGUMENT, map untouched"""
    import ratsdf
    vs = 0.02
    gpu = make_engine(vs, 6 * vs)
    f = synthetic.frame("wall", 0, scale=0.25)
    gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    n0 = gpu.num_active_blocks()
    for bad_pose in [(0, 0, 0, 1, float("nan"), 0, 0), (0, 0, float("inf"), 1, 0, 0, 0)]:
        with pytest.raises(ratsdf.RatsdfError) as ei:
            gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], bad_pose)
        assert ei.value.status == 1   # RATSDF_ERR_BAD_ARGUMENT
    with pytest.raises(ratsdf.RatsdfError):
        gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, (float("nan"), 500.0, 80.0, 60.0), f["pose"])
    with pytest.raises(ratsdf.RatsdfError):
        gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], float("inf"), f["intrinsics"], f["pose"])
    assert gpu.num_active_blocks() == n0
    gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])  # still usable


def test_full_resolution_tsdf_only_pair(make_engine, make_oracle):
    """BASELINE configs[1]: 640x480, 5 mm, ScanNet intrinsics, TSDF-only (ht / lt NULL -> all ones,
    modules/tsdf_module.cc:27-31): two frames; the probability of every voxel stays exactly 0.5."""
    vs = 0.005
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=16)
    frames = synthetic.stream("room", 2, scale=1.0, semantic=False, noise=True, holes=True)
    assert frames[0]["ht"] is None and frames[0]["depth"].shape == (480, 640)
    run_both(gpu, cpu, frames)
    p = gpu.gather_valid_semantic()["prob"]
    assert len(p) > 1000 * 512 and np.all(p == np.float32(0.5))
