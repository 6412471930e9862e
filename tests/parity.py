"""Map comparison used by the parity tests: HIP engine vs CPU oracle on the same inputs.

Bar (BASELINE.json north_star): hash indices / block IDs / free list bit-exact; tsdf and
probability within 1e-4 absolute; weight and rgb exact (they are integers derived from the same
fp32 operations in the same order; the only differing functions are expf/logf, which do not feed
them).
"""
import numpy as np

TOL = 1e-4


def assert_stats_equal(a, b, keys=("visible_blocks", "updated_voxels", "allocated_blocks",
                                   "deleted_blocks", "active_blocks")):
    sa, sb = a.last_frame_stats(), b.last_frame_stats()
    for k in keys:
        assert sa[k] == sb[k], f"frame stat {k}: {sa[k]} != {sb[k]} ({sa} vs {sb})"


def assert_directory_equal(a, b):
    ea, ba = a.dump_directory()
    eb, bb = b.dump_directory()
    assert len(ea) == len(eb), f"active entries {len(ea)} != {len(eb)}"
    assert np.array_equal(ea, eb), "hash entry indices differ"
    for f in ("x", "y", "z", "offset", "idx"):
        assert np.array_equal(ba[f], bb[f]), f"directory field {f} differs"
    return ea, ba


def assert_heap_equal(a, b):
    fa, ha = a.dump_heap()
    fb, hb = b.dump_heap()
    assert fa == fb, f"num_free {fa} != {fb}"
    assert np.array_equal(ha[:fa], hb[:fb]), "free list contents differ"


def assert_voxels_close(a, b, pool_idx, tol=TOL, chunk=4096):
    worst = dict(tsdf=0.0, prob=0.0)
    for lo in range(0, len(pool_idx), chunk):
        idx = pool_idx[lo:lo + chunk]
        ta, ca, pa = a.dump_voxels(idx)
        tb, cb, pb = b.dump_voxels(idx)
        assert np.array_equal(ca["weight"], cb["weight"]), "voxel weights differ"
        for ch in ("r", "g", "b"):
            assert np.array_equal(ca[ch], cb[ch]), f"voxel colour {ch} differs"
        dt = float(np.max(np.abs(ta - tb))) if ta.size else 0.0
        dp = float(np.max(np.abs(pa - pb))) if pa.size else 0.0
        worst["tsdf"] = max(worst["tsdf"], dt)
        worst["prob"] = max(worst["prob"], dp)
        assert dt <= tol, f"tsdf differs by {dt}"
        assert dp <= tol, f"probability differs by {dp}"
    return worst


def assert_maps_equal(a, b, tol=TOL):
    _, blocks = assert_directory_equal(a, b)
    assert_heap_equal(a, b)
    assert a.num_active_blocks() == b.num_active_blocks()
    return assert_voxels_close(a, b, blocks["idx"], tol)
