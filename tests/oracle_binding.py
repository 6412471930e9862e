"""Loads the CPU oracle for tests (never imported by the product package)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
ORACLE_LIB = ORACLE_DIR / "build" / "libratsdf_oracle.so"
sys.path.insert(0, str(ROOT / "ra-slam_amd"))

_lib = None


def build_oracle():
    src = ORACLE_DIR / "ratsdf_oracle.cpp"
    hdr = ROOT / "include" / "ratsdf.h"
    if ORACLE_LIB.exists() and ORACLE_LIB.stat().st_mtime >= max(src.stat().st_mtime,
                                                                 hdr.stat().st_mtime):
        return ORACLE_LIB
    subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)
    return ORACLE_LIB


def load_oracle():
    global _lib
    if _lib is None:
        from ratsdf._abi import Library
        _lib = Library(build_oracle(), "ratsdf_oracle_")
    return _lib
