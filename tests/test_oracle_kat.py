"""Pins the CPU oracle against the reference's own known answers (no GPU)."""
import pytest

import kat_cases


def test_hash_known_answers():
    kat_cases.case_hash_known_answers()


@pytest.mark.parametrize("case", kat_cases.ALL_ENGINE_CASES, ids=lambda c: c.__name__)
def test_oracle_matches_reference_gtests(case, make_oracle):
    case(make_oracle)


def test_result_buffers_are_handed_over_without_a_copy(make_oracle):
    """The arrays GatherValid / Query return wrap the library's own buffer (freed with ratsdf_free_buffer when the array
    is collected): they stay valid after the engine is gone, views keep the buffer alive, and they are writable."""
    import gc
    import numpy as np
    from ratsdf import synthetic
    e = make_oracle(0.02, 0.12)
    for f in synthetic.stream("room", 2, scale=0.25):
        e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    a = e.gather_valid()
    s = e.gather_valid_semantic()
    assert len(a) == len(s) > 0 and not a.flags.owndata and a.base is not None
    ref = a.copy()
    view = a[100:200]
    e.close()
    del a
    gc.collect()
    assert np.array_equal(view, ref[100:200])      # the view kept the buffer alive
    s["prob"][:10] = 0.25                          # writable
    assert np.all(s["prob"][:10] == 0.25) and np.array_equal(s["tsdf"], ref["tsdf"])
