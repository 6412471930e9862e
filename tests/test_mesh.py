"""Marching-cubes export (TSDFGrid::GatherValidMesh / TSDFSystem::DownloadAllMesh)."""
import re
from pathlib import Path

import numpy as np
import pytest

from ratsdf import synthetic

ROOT = Path(__file__).resolve().parent.parent
CORNER = [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)]
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def load_cases(path):
    return re.findall(r'"([0-9a-b]*)"', "".join(l for l in path.read_text().splitlines()
                                                  if not l.lstrip().startswith("//")))


@pytest.mark.parametrize("path", [ROOT / "oracle" / "mc_cases.inc",
                                  ROOT / "ra-slam_amd" / "csrc" / "mc_cases.inc"])
def test_case_table_is_a_valid_triangulation(path):
    """Every sign pattern uses exactly its sign-changing edges, in whole triangles; the oracle's and
    the engine's copies are the same data."""
    cases = load_cases(path)
    assert len(cases) == 256
    for c, s in enumerate(cases):
        assert len(s) % 3 == 0 and len(s) <= 15
        used = {int(ch, 16) for ch in s}
        crossing = {e for e, (a, b) in enumerate(EDGE) if ((c >> a) & 1) != ((c >> b) & 1)}
        assert used == crossing, f"case {c}"
    assert cases == load_cases(ROOT / "oracle" / "mc_cases.inc")
    # complementary sign patterns cut the same edges
    for c in range(256):
        assert {int(ch, 16) for ch in cases[c]} == {int(ch, 16) for ch in cases[255 - c]}


def _integrate_wall(e, n=14):
    f = synthetic.frame("wall", 0, scale=0.25)
    for _ in range(n):  # marching cubes only uses voxels with weight > 10 (voxel_tsdf.cu:600)
        e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    return f


def test_oracle_mesh_of_a_wall(make_oracle, tmp_path):
    vs = 0.02
    e = make_oracle(vs, 6 * vs, threads=4)
    _integrate_wall(e)
    v, tri, p = e.gather_valid_mesh()
    assert len(v) > 1000 and len(tri) > 1000 and len(p) == len(v)
    assert tri.min() >= 0 and tri.max() < len(v)
    assert len(np.unique(tri.reshape(-1))) == len(v)          # every exported vertex is referenced
    assert np.isfinite(v).all()
    assert np.abs(v[:, 2] - 2.0).max() < 1.5 * vs             # the wall is the plane z = 2 m
    assert ((p > 0) & (p < 1)).all()
    # file formats of DownloadAllMesh (tsdf_module.cc:66-86)
    fv, fi, fp = tmp_path / "v.bin", tmp_path / "i.bin", tmp_path / "p.bin"
    e.download_all_mesh(fv, fi, fp)
    assert np.array_equal(np.fromfile(fv, "<f4").reshape(-1, 3), v)
    assert np.array_equal(np.fromfile(fi, "<i4").reshape(-1, 3), tri)
    assert np.array_equal(np.fromfile(fp, "<f4"), p)
    # an empty map exports an empty mesh
    v0, t0, p0 = make_oracle(vs, 6 * vs).gather_valid_mesh()
    assert len(v0) == 0 and len(t0) == 0 and len(p0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["wall", "room"])
def test_engine_mesh_matches_oracle(scene, make_engine, make_oracle):
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=8)
    frames = [synthetic.frame(scene, 0, scale=0.25)] * 12 + synthetic.stream(scene, 4, scale=0.25)
    for f in frames:
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    gv, gt, gp = gpu.gather_valid_mesh()
    cv, ct, cp = cpu.gather_valid_mesh()
    assert len(cv) > 500
    assert np.array_equal(gt, ct), "triangle index buffers differ"
    assert gv.shape == cv.shape and np.max(np.abs(gv - cv), initial=0) <= 1e-6
    assert np.max(np.abs(gp - cp), initial=0) <= 1e-4
    v0, t0, p0 = make_engine(vs, 6 * vs).gather_valid_mesh()
    assert len(v0) == 0 and len(t0) == 0


def _tri_rows(v, t, p):
    a = np.concatenate([v[t].reshape(-1, 9), p[t].reshape(-1, 3)], axis=1)
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.gpu
def test_mesh_across_shard_seams_on_the_hip_engine(make_engine, make_oracle):
    """Two HIP engines hold the two subvolumes of one stream (block ownership by x-slab); each meshes a scratch
    copy of its own blocks plus the neighbour's seam blocks (ratsdf_import_blocks; a sharded engine meshes only
    what it owns): together, triangle for triangle, the mesh of the same stream held by ONE engine -- HIP and
    oracle.  (The two-rank form of the same thing over gloo: tests/test_multi_gloo.py.)"""
    from ratsdf import multi
    vs = 0.02
    kw = [dict(shard_rank=r, shard_count=2, shard_slab_bits=1) for r in range(2)]
    one, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=4)
    sh = [make_engine(vs, 6 * vs, **k) for k in kw]
    frames = [synthetic.frame("room", 0, scale=0.25) for _ in range(12)] + synthetic.stream("room", 4, scale=0.25)
    for f in frames:
        for e in [one, cpu] + sh:
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    dirs = [e.dump_directory()[1] for e in sh]
    assert multi.check_sharded_directories(dirs, slab_bits=1) == one.num_active_blocks()
    plan = multi.halo_plan(dirs)
    assert len(plan[0][1]) > 0 and len(plan[1][0]) > 0
    parts = []
    for r in range(2):
        scratch = make_engine(vs, 6 * vs, **kw[r])
        halo = multi.export_blocks(sh[1 - r], plan[1 - r][r])
        parts.append(_tri_rows(*multi.mesh_with_halo(sh[r], scratch, halo)))
        assert sh[r].num_active_blocks() == len(dirs[r])          # the map itself is untouched
    got = np.concatenate(parts)
    got = got[np.lexsort(got.T[::-1])]
    ref = _tri_rows(*one.gather_valid_mesh())
    assert got.shape == ref.shape and np.array_equal(got, ref), (got.shape, ref.shape)
    ref_cpu = _tri_rows(*cpu.gather_valid_mesh())
    assert ref.shape == ref_cpu.shape and np.abs(ref - ref_cpu).max() <= 1e-4
    assert sum(len(e.gather_valid_mesh()[1]) for e in sh) < len(ref)   # the seam cells exist


@pytest.mark.gpu
def test_import_blocks_round_trip(make_engine, make_oracle):
    """ratsdf_import_blocks: blocks exported from one map arrive in another with every voxel intact, whatever the
    receiving engine's shard filter says, new and already-present blocks alike; HIP engine and oracle agree."""
    from ratsdf import multi
    vs = 0.02
    src = make_engine(vs, 6 * vs)
    for f in synthetic.stream("room", 3, scale=0.25):
        src.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    _, b = src.dump_directory()
    pos = np.stack([b["x"], b["y"], b["z"]], axis=1)
    data = multi.export_blocks(src, pos)
    for dst in (make_engine(vs, 6 * vs, shard_rank=1, shard_count=4, shard_slab_bits=1),
                make_oracle(vs, 6 * vs, shard_rank=1, shard_count=4, shard_slab_bits=1)):
        dst.import_blocks(*data)
        dst.import_blocks(data[0][:5], data[1][:5], data[2][:5], data[3][:5])   # present already: overwritten
        assert dst.num_active_blocks() == len(pos)
        back = multi.export_blocks(dst, pos)
        assert np.array_equal(back[1], data[1]) and np.array_equal(back[2], data[2]) and np.array_equal(back[3], data[3])


@pytest.mark.gpu
def test_raycast_across_shards_on_the_hip_engine(make_engine, make_oracle):
    """Two HIP engines hold the two subvolumes of one stream; the image is partitioned into strips and every strip is
    rendered on a scratch HIP engine that holds only the blocks the plan names for it (multi.raycast_plan /
    raycast_strip; the two-rank form over gloo: tests/test_multi_gloo.py): assembled, bit for bit the rendering of
    the WHOLE sharded map by one engine (every shard's blocks imported into one scratch engine).
    Not compared with the rendering of the same stream integrated by one unsharded engine: the maps differ in a
    few blocks by construction -- in one engine two absent blocks of different owners that hash to the same bucket
    collide (voxel_hash.cu:46-108: the later request finds the bucket locked and its block is inserted a frame
    later, missing an update), in the shards they never meet.  That is a property of BASELINE config 4 under the
    reference's allocation rule, in the oracle as well; each shard equals the oracle with the same ownership
    filter (test_gpu_parity.py)."""
    from ratsdf import multi
    vs = 0.01
    kw = [dict(shard_rank=r, shard_count=2, shard_slab_bits=1) for r in range(2)]
    one = make_engine(vs, 6 * vs)
    sh = [make_engine(vs, 6 * vs, **k) for k in kw]
    frames = [synthetic.frame("room", 0, scale=0.25) for _ in range(12)] + synthetic.stream("room", 6, scale=0.25)
    for f in frames:
        for e in [one] + sh:
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    dirs = [e.dump_directory()[1] for e in sh]
    assert multi.check_sharded_directories(dirs, slab_bits=1) == one.num_active_blocks()
    view = frames[-1]
    H, W = view["depth"].shape
    K, T = view["intrinsics"], view["pose"]
    whole = make_engine(vs, 6 * vs)
    for e, d in zip(sh, dirs):
        multi._import_chunks(whole, multi.export_blocks(e, np.stack([d["x"], d["y"], d["z"]], axis=1)))
    assert whole.num_active_blocks() == one.num_active_blocks()
    ref_rgba, ref_normal = whole.raycast(K, H, W, T, 4.0)
    assert (ref_rgba[..., 3] == 255).mean() > 0.5
    # (the unsharded stream's rendering is close, not equal: see above)
    one_rgba, _ = one.raycast(K, H, W, T, 4.0)
    assert (np.abs(one_rgba.astype(int) - ref_rgba.astype(int)).max(axis=2) > 2).mean() < 0.05
    plan = multi.raycast_plan(dirs, K, H, W, T, 4.0, vs)
    strips = multi.strip_rows(H, 2)
    rgba, normal = [], []
    for r in range(2):
        assert all(len(plan[q][r]) > 0 for q in range(2))          # the strip reads blocks of both subvolumes
        assert sum(len(plan[q][r]) for q in range(2)) < one.num_active_blocks()
        scratch = make_engine(vs, 6 * vs)
        a, b = multi.raycast_strip(scratch, [multi.export_blocks(sh[q], plan[q][r]) for q in range(2)], K, H, W, T,
                                   4.0, strips[r])
        rgba.append(a)
        normal.append(b)
        assert sh[r].num_active_blocks() == len(dirs[r])           # the maps themselves are untouched
    assert np.array_equal(np.concatenate(rgba), ref_rgba) and np.array_equal(np.concatenate(normal), ref_normal)
    own, _ = sh[0].raycast(K, H, W, T, 4.0)
    assert not np.array_equal(own, ref_rgba)                       # own blocks alone render something else


@pytest.mark.gpu
def test_block_exchange_in_device_memory(make_engine):
    """ratsdf_export_blocks_device / ratsdf_import_blocks_device: the records an engine writes into a device buffer are
    the voxels dump_voxels() reports (tsdf | rgbw | prob per block), a listed block the map lacks is counted and left
    zero, and a scratch engine that imports the records holds the same blocks -- no host copy of voxel data."""
    import torch
    from ratsdf import multi
    vs = 0.02
    src = make_engine(vs, 6 * vs)
    for f in synthetic.stream("room", 3, scale=0.25):
        src.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    _, b = src.dump_directory()
    pos = list(zip(b["x"].tolist(), b["y"].tolist(), b["z"].tolist()))
    host = multi.export_blocks(src, pos)
    rec = multi.export_blocks_device(src, pos, len(pos) + 3, "cuda")
    got = rec.cpu().numpy()
    assert np.array_equal(got[:len(pos), 0:512], host[1].view(np.int32))
    assert np.array_equal(got[:len(pos), 512:1024], np.ascontiguousarray(host[2]).view(np.int32).reshape(-1, 512))
    assert np.array_equal(got[:len(pos), 1024:1536], host[3].view(np.int32))
    assert not got[len(pos):].any()
    # a position the map does not hold: counted, record of zeros
    absent = [(30000, 30000, 30000)]
    with pytest.raises(RuntimeError, match="1 of 2 listed blocks"):
        multi.export_blocks_device(src, [pos[0]] + absent, 2, "cuda")
    p_t = multi._pos_tensor([pos[0]] + absent, "cuda")
    out = torch.full((2, 1536), 7, dtype=torch.int32, device="cuda")
    missing = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    src.export_blocks_device(2, p_t.data_ptr(), out.data_ptr(), missing.data_ptr())
    src.synchronize()
    assert int(missing.item()) == 1 and not out[1].any().item() and np.array_equal(out[0].cpu().numpy(), got[0])
    # into a scratch engine whose shard filter would refuse half of them; twice (present already: overwritten)
    dst = make_engine(vs, 6 * vs, shard_rank=1, shard_count=4, shard_slab_bits=1)
    multi.import_blocks_device(dst, pos, rec[:len(pos)].contiguous(), chunk=100)
    multi.copy_blocks_device(src, dst, pos[:5], "cuda")
    assert dst.num_active_blocks() == len(pos)
    back = multi.export_blocks(dst, pos)
    assert np.array_equal(back[1], host[1]) and np.array_equal(back[2], host[2]) and np.array_equal(back[3], host[3])
    assert src.num_active_blocks() == len(pos)


@pytest.mark.gpu
def test_across_shard_exports_with_voxel_data_on_the_device(make_engine):
    """The device form of multi.mesh_across_shards / raycast_across_shards (what they do under RCCL: device_exchange)
    on two HIP shard engines in one process -- the all-gather is stood in for by stacking the two send buffers: same
    triangles and same image as the host-memory form."""
    import torch
    from ratsdf import multi
    vs = 0.02
    kw = [dict(shard_rank=r, shard_count=2, shard_slab_bits=1) for r in range(2)]
    sh = [make_engine(vs, 6 * vs, **k) for k in kw]
    frames = [synthetic.frame("room", 0, scale=0.25) for _ in range(12)] + synthetic.stream("room", 4, scale=0.25)
    for f in frames:
        for e in sh:
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    dirs = [e.dump_directory()[1] for e in sh]
    assert multi.device_exchange(sh[0], "cuda") and not multi.device_exchange(sh[0], None)
    # -- mesh
    plan = multi.halo_plan(dirs)
    out_lists = [sorted(set(p for r in range(2) for p in plan[q][r])) for q in range(2)]
    nmax = max(1, max(len(l) for l in out_lists))
    assert all(len(l) > 0 for l in out_lists)
    recv = torch.stack([multi.export_blocks_device(sh[q], out_lists[q], nmax, "cuda") for q in range(2)])
    for r in range(2):
        dev = _tri_rows(*multi.mesh_rank_device(sh[r], make_engine(vs, 6 * vs, **kw[r]), recv, out_lists, plan, r, "cuda"))
        host = _tri_rows(*multi.mesh_with_halo(sh[r], make_engine(vs, 6 * vs, **kw[r]),
                                               multi.export_blocks(sh[1 - r], plan[1 - r][r])))
        assert dev.shape == host.shape and len(dev) > 0 and np.array_equal(dev, host)
        assert sh[r].num_active_blocks() == len(dirs[r])
    # -- ray cast
    view = frames[-1]
    H, W = view["depth"].shape
    K, T = view["intrinsics"], view["pose"]
    plan = multi.raycast_plan(dirs, K, H, W, T, 4.0, vs)
    strips = multi.strip_rows(H, 2)
    out_lists = [sorted(set(p for r in range(2) if r != q for p in plan[q][r])) for q in range(2)]
    nmax = max(1, max(len(l) for l in out_lists))
    recv = torch.stack([multi.export_blocks_device(sh[q], out_lists[q], nmax, "cuda") for q in range(2)])
    for r in range(2):
        dev = multi.raycast_rank_device(sh[r], make_engine(vs, 6 * vs), recv, out_lists, plan, r, "cuda", K, H, W, T, 4.0,
                                        strips[r])
        host = multi.raycast_strip(make_engine(vs, 6 * vs), [multi.export_blocks(sh[q], plan[q][r]) for q in range(2)],
                                   K, H, W, T, 4.0, strips[r])
        assert np.array_equal(dev[0], host[0]) and np.array_equal(dev[1], host[1]) and (dev[0][..., 3] == 255).any()
    # -- the entry points themselves, one rank (no process group: world 1)
    v, tri, vp = multi.mesh_across_shards(sh[0], lambda: make_engine(vs, 6 * vs, **kw[0]), [dirs[0]], device="cuda")
    v2, tri2, vp2 = multi.mesh_across_shards(sh[0], lambda: make_engine(vs, 6 * vs, **kw[0]), [dirs[0]])
    assert np.array_equal(_tri_rows(v, tri, vp), _tri_rows(v2, tri2, vp2))
    a = multi.raycast_across_shards(sh[0], lambda: make_engine(vs, 6 * vs), [dirs[0]], K, H, W, T, 4.0, vs, device="cuda")
    b = multi.raycast_across_shards(sh[0], lambda: make_engine(vs, 6 * vs), [dirs[0]], K, H, W, T, 4.0, vs)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
