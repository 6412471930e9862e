"""The retired frame forms and every tuning switch exist in the diagnostic build only: the shipped
library reads three environment variables (RATSDF_GRAPH, RATSDF_SYNC_INTEGRATE, RATSDF_COPY_STREAMS).  The forms are
kept bit-exact: tests/diagnostic_build_cases.py runs here, in ONE child process, on libratsdf_stamps.so."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "ra-slam_amd" / "csrc" / "build"


def test_shipped_library_reads_few_environment_variables():
    lib = BUILD / "libratsdf.so"
    if not lib.exists():
        import __graft_entry__
        __graft_entry__.build()
    names = set(re.findall(rb"RATSDF_[A-Z_]{3,}", lib.read_bytes()))
    names = {n.decode() for n in names if not n.startswith(b"RATSDF_ERR") and n != b"RATSDF_DEBUG"}
    # (RATSDF_DEBUG only appears inside a diagnostic message of ratsdf_debug_wave_stamps)
    assert names <= {"RATSDF_GRAPH", "RATSDF_SYNC_INTEGRATE", "RATSDF_COPY_STREAMS"}, names


@pytest.mark.gpu
def test_retired_frame_forms_stay_bit_exact_in_the_diagnostic_build():
    lib = BUILD / "libratsdf_stamps.so"
    assert lib.exists(), "diagnostic build missing: make -C ra-slam_amd/csrc stamps (build() does it)"
    env = dict(os.environ, RATSDF_LIB=str(lib))
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "diagnostic_build_cases.py"), "-q", "-x",
                        "-p", "no:cacheprovider", "-m", "gpu"], capture_output=True, text=True, timeout=900, env=env,
                       cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 9, r.stdout[-2000:]
