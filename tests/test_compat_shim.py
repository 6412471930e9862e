"""The reference-side binding (ra-slam_amd/host/include/ratsdf/compat/*.h: TSDFSystem / TSDFGrid with
the reference's cv::Mat / SE3 / GLImage8UC4 signatures) compiles against stand-in declarations of
OpenCV, Eigen and the reference's own headers (tests/cpp/compat_stubs: none of the real ones exists
here, so this is a syntax / type check), and the calls the reference's callers make run through it.
On CPU the C ABI underneath is the oracle library; with a GPU (-m gpu) it is the HIP engine."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "ra-slam_amd" / "host"
EXE = HOST / "build" / "test_compat_shim"


def build(prefix="ratsdf_", exe=EXE):
    src = ROOT / "tests" / "cpp" / "test_compat_shim.cc"
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-pthread", f'-DRATSDF_ABI_PREFIX="{prefix}"',
           "-I", str(ROOT / "tests" / "cpp" / "compat_stubs"), "-I", str(HOST / "include"),
           "-o", str(exe), str(src), str(HOST / "src" / "tsdf_host.cc"), "-ldl"]
    exe.parent.mkdir(exist_ok=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "warning" not in r.stderr, r.stderr


def test_binding_compiles_and_runs_on_the_oracle(oracle_lib):
    exe = HOST / "build" / "test_compat_shim_oracle"
    build("ratsdf_oracle_", exe)   # test-only build: the host layer bound to the oracle's ABI prefix
    r = subprocess.run([str(exe), oracle_lib.path, "ratsdf_oracle_"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "TSDFSystem:" in r.stdout and "TSDFGrid:" in r.stdout


@pytest.mark.gpu
def test_binding_runs_on_the_hip_engine():
    import ratsdf
    build()
    r = subprocess.run([str(EXE), str(ratsdf.LIB_PATH), "ratsdf_"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "TSDFSystem:" in r.stdout and "TSDFGrid:" in r.stdout
