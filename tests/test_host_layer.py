"""C++ host layer (ra-slam_amd/host: TSDFGrid / TSDFSystem with the reference's class shapes).

The same test program runs against the CPU oracle (here, no GPU) and against the HIP engine (-m gpu):
queue order, deep copies, pause, idempotent terminate, ones-fill for missing ht/lt, Query locking
path, DownloadAll record format, extrinsics composition."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "ra-slam_amd" / "host"
EXE = HOST / "build" / "test_tsdf_system"


def build_test_program():
    subprocess.run(["make", "-C", str(HOST)], check=True, capture_output=True)
    src = ROOT / "tests" / "cpp" / "test_tsdf_system.cc"
    deps = [src, HOST / "src" / "tsdf_host.cc"] + list((HOST / "include" / "ratsdf").glob("*.hpp"))
    if not EXE.exists() or EXE.stat().st_mtime < max(p.stat().st_mtime for p in deps):
        subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", f"-I{HOST / 'include'}", str(src),
                        str(HOST / "src" / "tsdf_host.cc"), "-ldl", "-o", str(EXE)], check=True)
    return EXE


def run(lib, prefix):
    exe = build_test_program()
    r = subprocess.run([str(exe), str(lib), prefix], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    return r.stdout


def test_host_layer_on_oracle(oracle_lib):
    out = run(oracle_lib.path, "ratsdf_oracle_")
    assert "cpu-oracle" in out


@pytest.mark.gpu
def test_host_layer_on_hip_engine():
    import ratsdf
    out = run(ratsdf.LIB_PATH, "ratsdf_")
    assert "hip-gfx950" in out
