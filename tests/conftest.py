"""pytest wiring: the `gpu` marker, import paths, and the oracle / engine fixtures."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # PyTorch bundles its own HIP runtime.  If libratsdf.so (system ROCm runtime) initialises HIP
    # first, torch later reports "No HIP GPUs are available"; initialising torch first makes both
    # share one runtime.  Only matters for tests that also use torch tensors on the GPU.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure).  Built on demand with the committed Makefile."""
    from oracle_binding import load_oracle
    return load_oracle()


@pytest.fixture()
def make_oracle(oracle_lib):
    from ratsdf._abi import Engine
    made = []

    def _make(voxel_size=0.01, truncation=0.06, **kw):
        e = Engine(oracle_lib, voxel_size, truncation, **kw)
        made.append(e)
        return e

    yield _make
    for e in made:
        e.close()


@pytest.fixture()
def make_engine():
    """HIP engine factory (GPU tests only); goes through the C ABI of libratsdf.so."""
    import ratsdf
    made = []

    def _make(voxel_size=0.01, truncation=0.06, **kw):
        e = ratsdf.TSDFGrid(voxel_size, truncation, **kw)
        made.append(e)
        return e

    yield _make
    for e in made:
        e.close()
