"""Oracle self-consistency under heavy collisions (no GPU)."""
import pytest

import fuzz_cases


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_directory_invariants(seed, make_oracle):
    e = make_oracle(0.01, 0.06, **fuzz_cases.SMALL)
    chained = 0
    for kind, pos in fuzz_cases.passes(seed):
        (e.test_allocate if kind == "alloc" else e.test_delete)(pos)
        fuzz_cases.check_invariants(e)
        _, bl = e.dump_directory()
        chained = max(chained, int((bl["offset"] != 0).sum()))
    assert chained > 0, "fuzz never produced a chained bucket"
