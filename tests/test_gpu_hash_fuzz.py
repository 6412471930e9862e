"""HIP directory maintenance vs the oracle under heavy collisions: chained buckets, wrap-around
probing, lock interplay between buckets, head / chain deletes (needs a GPU)."""
import numpy as np
import pytest

import fuzz_cases
import kat_cases
from parity import assert_directory_equal, assert_heap_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(8))
def test_tiny_table_fuzz(seed, make_engine, make_oracle):
    gpu = make_engine(0.01, 0.06, **fuzz_cases.SMALL)
    cpu = make_oracle(0.01, 0.06, **fuzz_cases.SMALL)
    for step, (kind, pos) in enumerate(fuzz_cases.passes(seed, n_pass=80)):
        for e in (gpu, cpu):
            (e.test_allocate if kind == "alloc" else e.test_delete)(pos)
        try:
            assert_directory_equal(gpu, cpu)
            assert_heap_equal(gpu, cpu)
        except AssertionError as err:
            raise AssertionError(f"seed {seed} pass {step} ({kind}, {len(pos)} requests): {err}")
    fuzz_cases.check_invariants(gpu)


def test_long_chain_default_table(make_engine, make_oracle):
    """Six blocks in one bucket of the reference-sized table + traffic in the neighbouring buckets
    the chain spills into, requested in shuffled order over several passes, then deleted."""
    gpu, cpu = make_engine(), make_oracle()
    b = 12345
    pos = (kat_cases.chain_positions(b, 6) + kat_cases.chain_positions(b + 1, 4) +
           kat_cases.chain_positions(b + 2, 3) + kat_cases.chain_positions(b + 3, 2))
    rng = np.random.default_rng(5)
    for _ in range(8):
        order = rng.permutation(len(pos))
        req = np.array([pos[i] for i in order], dtype=np.int16)
        for e in (gpu, cpu):
            e.test_allocate(req)
        assert_directory_equal(gpu, cpu)
        assert_heap_equal(gpu, cpu)
    assert gpu.num_active_blocks() == len(pos)
    for _ in range(6):
        k = rng.choice(len(pos), size=5, replace=False)
        req = np.array([pos[i] for i in k], dtype=np.int16)
        for e in (gpu, cpu):
            e.test_delete(req)
        assert_directory_equal(gpu, cpu)
        assert_heap_equal(gpu, cpu)
        for e in (gpu, cpu):
            e.test_allocate(req[::-1].copy())
        assert_directory_equal(gpu, cpu)
        assert_heap_equal(gpu, cpu)


def test_pool_exhaustion_is_reported(make_engine, make_oracle):
    from ratsdf import RatsdfError
    small = dict(block_bits=4, bucket_bits=9)  # 16 blocks
    pos = fuzz_cases.candidate_positions(3, n=40)
    for make in (make_engine, make_oracle):
        e = make(0.01, 0.06, **small)
        with pytest.raises(RatsdfError) as ei:
            e.test_allocate(pos)
        assert ei.value.status == 3
        assert e.num_active_blocks() <= 16
