"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle must still reproduce them.  GPU: the HIP engine must match them through the C ABI
(integers bit-exact, tsdf / probability within 1e-4)."""
import sys
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden"
sys.path.insert(0, str(GOLDEN))
import make_golden  # noqa: E402


def compare(out, ref, tol):
    assert np.array_equal(out["stats"], ref["stats"]), (out["stats"], ref["stats"])
    for k in ("entry_index", "blocks", "pool_idx", "heap_tail", "rgbw"):
        assert np.array_equal(out[k], ref[k]), f"{k} differs"
    assert int(out["num_free"]) == int(ref["num_free"])
    assert np.max(np.abs(out["tsdf"] - ref["tsdf"]), initial=0) <= tol
    assert np.max(np.abs(out["prob"] - ref["prob"]), initial=0) <= tol


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(name, oracle_lib):
    from ratsdf._abi import Engine
    out = make_golden.run_case(lambda vs, tr: Engine(oracle_lib, vs, tr), make_golden.CASES[name])
    compare(out, np.load(GOLDEN / f"{name}.npz"), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_engine_matches_golden(name):
    import ratsdf
    out = make_golden.run_case(lambda vs, tr: ratsdf.TSDFGrid(vs, tr), make_golden.CASES[name])
    compare(out, np.load(GOLDEN / f"{name}.npz"), 1e-4)
