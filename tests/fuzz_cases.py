"""Randomised allocate / delete passes on a deliberately tiny hash table (512 buckets) so that full
buckets, chains, wrap-around probing and lock interplay happen all the time.  Shared by the oracle
self-consistency test (CPU) and the HIP-vs-oracle parity test (GPU)."""
import numpy as np

SMALL = dict(block_bits=12, bucket_bits=9)


def candidate_positions(seed, n=1500, span=40):
    rng = np.random.default_rng(seed)
    pos = rng.integers(-span, span, size=(n * 2, 3)).astype(np.int16)
    pos = np.unique(pos, axis=0)
    rng.shuffle(pos)
    return pos[:n]


def passes(seed, n_pass=60, max_active=520):
    """Yields ('alloc'|'delete', positions) with the active set kept below the table's capacity."""
    rng = np.random.default_rng(seed + 1000)
    cand = candidate_positions(seed)
    active_guess = 0
    for i in range(n_pass):
        if active_guess > max_active or (i % 3 == 2):
            k = int(rng.integers(20, 200))
            yield "delete", cand[rng.choice(len(cand), size=k, replace=False)]
            active_guess = max(0, active_guess - k // 3)
        else:
            k = int(rng.integers(20, 160))
            # repeats inside one pass are legal (many pixels request the same block)
            idx = rng.choice(len(cand), size=k, replace=True)
            yield "alloc", cand[idx]
            active_guess += k // 2


def check_invariants(e, block_bits=12):
    """Structural invariants of the directory + free list."""
    ei, bl = e.dump_directory()
    nf, heap = e.dump_heap()
    nb = 1 << block_bits
    assert len(ei) == nb - nf == e.num_active_blocks()
    used = set(int(v) for v in bl["idx"])
    assert len(used) == len(bl), "pool index handed out twice"
    free = set(int(v) for v in heap[:nf])
    assert len(free) == nf and not (free & used) and (free | used) == set(range(nb))
    # every block can be found again through the hash function / chain walk
    if len(bl):
        pts = np.stack([bl["x"], bl["y"], bl["z"]], axis=1).astype(np.int32) * 8
        _, _, _, found = e.test_retrieve(pts.astype(np.int16))
        assert np.array_equal(found["idx"], bl["idx"])
    pos = set(zip(bl["x"].tolist(), bl["y"].tolist(), bl["z"].tolist()))
    assert len(pos) == len(bl), "block inserted twice"
