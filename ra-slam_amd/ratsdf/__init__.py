"""ratsdf -- Python host binding of the MI355X-native TSDF engine (libratsdf.so, HIP / gfx950).

The binding loads the in-tree C-ABI library built from ``ra-slam_amd/csrc`` and nothing else: there
is no CPU fallback.  If the HIP library has not been built the import of an engine fails loudly.
"""
import os
from pathlib import Path

from . import _abi
from ._abi import (BLOCK_DTYPE, RGBW_DTYPE, VOXEL_SEGM_DTYPE, VOXEL_TSDF_DTYPE, Bounds, Engine, Group,
                   Intrinsics, Library, Pose, RatsdfError)
from .pose import compose, identity_pose, invert, pose_from_matrix

_PKG_ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = _PKG_ROOT / "csrc" / "build" / "libratsdf.so"
_lib = None


def library():
    """The HIP engine library; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        path = Path(os.environ.get("RATSDF_LIB", LIB_PATH))
        if not path.exists():
            raise ImportError(
                f"{path} not found: build the HIP engine first "
                f"(python -c 'import __graft_entry__ as g; g.build()' or make -C ra-slam_amd/csrc)")
        _lib = Library(path, "ratsdf_")
    return _lib


class TSDFGrid(Engine):
    """``TSDFGrid(voxel_size, truncation)`` of utils/tsdf/voxel_tsdf.cuh:47 on one MI355X."""

    def __init__(self, voxel_size, truncation, device=0, **kw):
        super().__init__(library(), voxel_size, truncation, device=device, **kw)


__all__ = ["TSDFGrid", "Engine", "Group", "Library", "library", "Intrinsics", "Pose", "Bounds",
           "RatsdfError", "pose_from_matrix", "compose", "invert", "identity_pose", "BLOCK_DTYPE",
           "RGBW_DTYPE", "VOXEL_TSDF_DTYPE", "VOXEL_SEGM_DTYPE", "LIB_PATH"]
