"""Known-answer cases restated from the reference's own gtests, written once and run against both
the CPU oracle (tests/test_oracle_kat.py, no GPU) and the HIP engine (tests/test_gpu_kat.py).

Sources (reference tree): utils/tests/voxel_hash_test.cu:56-180, utils/tests/voxel_mem_test.cu:38-90.
Each function takes a factory returning a fresh engine with reference-default table sizes.
"""
import numpy as np

from ratsdf._abi import RGBW_DTYPE

NUM_BUCKET = 1 << 21
NUM_BLOCK = 1 << 18
BLOCK_LEN = 8


def ref_hash(p, bucket_bits=21):
    """Hash(), utils/tsdf/voxel_hash.cu:19-23, in Python integers (uint32 wraparound)."""
    m = 0xFFFFFFFF
    x, y, z = (int(v) & m for v in p)
    return (((x * 73856093) & m) ^ ((y * 19349669) & m) ^ ((z * 83492791) & m)) & (
        (1 << bucket_bits) - 1)


def rgbw(vals):
    a = np.zeros(len(vals), dtype=RGBW_DTYPE)
    for i, v in enumerate(vals):
        a[i] = (v, v, v, v)
    return a


def case_hash_known_answers():
    # voxel_hash_test.cu:130-135: all three hash to NUM_BUCKET - 1
    for p in [(33, 180, 42), (61, 16, 170), (63, 171, 45)]:
        assert ref_hash(p) == NUM_BUCKET - 1
    assert ref_hash((0, 0, 0)) == 0
    assert ref_hash((1, 1, 1)) == 1592143
    assert ref_hash((-1, -1, -1)) == 505009


def case_single(make):
    """TEST_F(VoxelHashTest, Single), voxel_hash_test.cu:56-92."""
    e = make()
    e.test_allocate([[1, 1, 1]])
    r, t, p, b = e.test_retrieve([[8, 8, 8]])
    assert e.num_active_blocks() == 1
    assert (b["x"][0], b["y"][0], b["z"][0]) == (1, 1, 1)
    assert b["idx"][0] == NUM_BLOCK - 1  # first acquire = heap[NUM_BLOCK-1], voxel_mem.cu:24,38-41
    # freshly acquired block: weight 1, tsdf -1, probability .5 (voxel_mem.cu:43-51)
    assert r["weight"][0] == 1 and t[0] == -1.0 and p[0] == 0.5
    # retrieve from an unallocated block -> default voxel, weight 0 (:70), tsdf -10, prob 0
    r, t, p, b = e.test_retrieve([[0, 0, 0]])
    assert r["weight"][0] == 0 and t[0] == -10.0 and p[0] == 0.0
    assert b["idx"][0] == -1 and b["offset"][0] == -1
    # assignment along z of block (0,0,0) (:72-91)
    e.test_allocate([[0, 0, 0]])
    pts = [[0, 0, i] for i in range(BLOCK_LEN)]
    e.test_assign_rgbw(pts, rgbw(range(BLOCK_LEN)))
    assert e.num_active_blocks() == 2
    r, _, _, _ = e.test_retrieve(pts)
    for i in range(BLOCK_LEN):
        assert (r["r"][i], r["g"][i], r["b"][i], r["weight"][i]) == (i, i, i, i)


def case_multiple(make):
    """TEST_F(VoxelHashTest, Multiple), voxel_hash_test.cu:94-126: 128 blocks in ONE pass."""
    e = make()
    n = 128
    pos = [[i, i, i] for i in range(n)]
    e.test_allocate(pos)
    assert e.num_active_blocks() == n
    pts = [[i * BLOCK_LEN] * 3 for i in range(n)]
    e.test_assign_rgbw(pts, rgbw(range(n)))
    r, _, _, b = e.test_retrieve(pts)
    for i in range(n):
        assert (r["r"][i], r["g"][i], r["b"][i], r["weight"][i]) == (i, i, i, i)
        assert (b["x"][i], b["y"][i], b["z"][i]) == (i, i, i)
    # raster order of the single pass fixes the pool indices: descending from NUM_BLOCK-1
    assert list(b["idx"]) == [NUM_BLOCK - 1 - i for i in range(n)]


def case_collision(make):
    """TEST_F(VoxelHashTest, Collision), voxel_hash_test.cu:128-180: one insertion per bucket per
    pass (2 -> 3 -> 4 active blocks), list head at entry 1, wrap-around chaining to entry 2."""
    e = make()
    pos = [[33, 180, 42], [61, 16, 170], [63, 171, 45], [0, 0, 0]]
    for expect in (2, 3, 4):
        e.test_allocate(pos)
        assert e.num_active_blocks() == expect
    pts = [[c * BLOCK_LEN for c in p] for p in pos]
    e.test_assign_rgbw(pts, rgbw(range(4)))
    r, _, _, _ = e.test_retrieve(pts)
    for i in range(4):
        assert (r["r"][i], r["g"][i], r["b"][i], r["weight"][i]) == (i, i, i, i)
    # implied by voxel_hash.cu:67-104 under the raster-order linearisation
    ei, bl = e.dump_directory()
    got = {int(k): (int(b["x"]), int(b["y"]), int(b["z"]), int(b["offset"]), int(b["idx"]))
           for k, b in zip(ei, bl)}
    last = 2 * NUM_BUCKET
    assert got == {
        0: (0, 0, 0, 0, NUM_BLOCK - 2),
        2: (63, 171, 45, 0, NUM_BLOCK - 4),        # third collider: probe wraps to entry 2
        last - 2: (33, 180, 42, 0, NUM_BLOCK - 1),
        last - 1: (61, 16, 170, 3, NUM_BLOCK - 3),  # list head, offset = 2 + 2^22 - (2^22-1) = 3
    }


def case_pool(make):
    """TEST_F(VoxelMemTest, Test1), voxel_mem_test.cu:38-90, through the hash-level hooks: distinct
    blocks, release does not clobber voxel memory, re-acquire resets weight (only) to 1."""
    e = make()
    n = 8
    pos = [[i, 0, 0] for i in range(n)]
    e.test_allocate(pos)
    _, _, _, b = e.test_retrieve([[i * BLOCK_LEN, 0, 0] for i in range(n)])
    idx = [int(v) for v in b["idx"]]
    assert len(set(idx)) == n and min(idx) >= 0
    # write weight = i (and rgb = i) into every voxel of block i
    for i in range(n):
        pts = [[i * BLOCK_LEN + x, y, z] for z in range(8) for y in range(8) for x in range(8)]
        e.test_assign_rgbw(pts, rgbw([i] * 512))
    _, w, _ = e.dump_voxels(idx)
    for i in range(n):
        assert (w["weight"][i] == i).all()
    nf0, _ = e.dump_heap()
    ei, bl = e.dump_directory()
    e.test_delete(pos)
    assert e.num_active_blocks() == 0
    nf1, heap = e.dump_heap()
    assert nf1 == nf0 + n and nf1 == NUM_BLOCK
    # released in hash-entry order (the carve pass order): heap[free++] = idx (voxel_mem.cu:56-60)
    released = [int(v) for v in bl["idx"]]          # dump_directory is in entry order
    assert [int(v) for v in heap[nf0:nf1]] == released
    _, w, _ = e.dump_voxels(idx)
    for i in range(n):
        assert (w["weight"][i] == i).all()  # release keeps the data (:68-78)
    e.test_allocate(pos)
    _, _, _, b2 = e.test_retrieve([[i * BLOCK_LEN, 0, 0] for i in range(n)])
    # LIFO free list: the most recently released block is handed out first
    assert [int(v) for v in b2["idx"]] == released[::-1]
    t, w, p = e.dump_voxels(idx)
    for i in range(n):
        assert (w["weight"][i] == 1).all()  # re-acquire resets weight (:79-89) ...
        assert (w["r"][i] == i).all()       # ... but not rgb (voxel_mem.cu:43-51)
        assert (t[i] == -1.0).all() and (p[i] == 0.5).all()


def chain_positions(bucket, count, bucket_bits=21, start=0):
    """`count` distinct small block positions hashing to `bucket` (brute force, deterministic)."""
    out = []
    rng = np.random.default_rng(12345 + bucket + start)
    while len(out) < count:
        cand = rng.integers(-2000, 2000, size=(200000, 3)).astype(np.int64)
        m = 0xFFFFFFFF
        h = (((cand[:, 0] & m) * 73856093 & m) ^ ((cand[:, 1] & m) * 19349669 & m) ^
             ((cand[:, 2] & m) * 83492791 & m)) & ((1 << bucket_bits) - 1)
        for p in cand[h == bucket]:
            t = tuple(int(v) for v in p)
            if t not in out:
                out.append(t)
            if len(out) == count:
                break
    return out


ALL_ENGINE_CASES = [case_single, case_multiple, case_collision, case_pool]
