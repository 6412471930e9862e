"""The C-ABI library loads without a GPU and exports every symbol include/ratsdf.h declares."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "ratsdf.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ratsdf_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("ratsdf_create", "ratsdf_integrate", "ratsdf_integrate_device", "ratsdf_query",
                 "ratsdf_gather_valid_semantic", "ratsdf_destroy"):
        assert must in syms


def test_hip_library_exports_every_declared_symbol():
    import ratsdf
    if not ratsdf.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
    lib = ratsdf.library()
    for s in declared_symbols():
        assert hasattr(lib.dll, s), f"libratsdf.so does not export {s}"
    assert lib.backend() == "hip-gfx950"
    # the binding's symbol table covers the header too
    assert sorted("ratsdf_" + s for s in ratsdf._abi.SYMBOLS) == declared_symbols()


def test_oracle_exports_the_same_abi(oracle_lib):
    for s in declared_symbols():
        assert hasattr(oracle_lib.dll, s.replace("ratsdf_", "ratsdf_oracle_", 1))


def test_no_device_is_reported_not_faked():
    """Without a GPU the engine must refuse to create a map (there is no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import ratsdf
    with pytest.raises(ratsdf.RatsdfError) as ei:
        ratsdf.TSDFGrid(0.01, 0.06)
    assert ei.value.status in (5, 2)


def test_product_package_does_not_touch_the_oracle():
    """Nothing shipped under ra-slam_amd/ may load, link or import the CPU oracle."""
    pkg = ROOT / "ra-slam_amd"
    banned = ("libratsdf_oracle", "oracle/build", "oracle_binding", "ratsdf_oracle.cpp",
              "import oracle", "from oracle")
    for p in list(pkg.rglob("*.py")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.hip")) + \
            list(pkg.rglob("Makefile")):
        txt = p.read_text()
        for b in banned:
            assert b not in txt, f"{p} references the oracle ({b})"
