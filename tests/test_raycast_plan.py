"""multi.frustum_blocks is a SUPERSET of the blocks a strip's rays can read: random rays through random pixels of
random strips, sampled densely out to the ray length of TSDFGrid::RayCast (voxel_tsdf.cu:296-303: max_step steps of
truncation / 2), together with the neighbourhood the renderer taps around a sample (trilinear corners and gradient
taps: +-1 voxel; two for slack) -- every block such a voxel falls into must be selected.  Pure numpy, no engine."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))

from ratsdf import multi, synthetic  # noqa: E402


def _ray_points(K, pose, px, py, ts):
    """world points at distances `ts` along the ray of pixel (px, py); pose = cam_T_world"""
    fx, fy, cx, cy = [float(v) for v in K]
    d = np.array([(px - cx) / fx, (py - cy) / fy, 1.0])
    d /= np.linalg.norm(d)
    R = multi._quat_matrix(pose[:4])
    t = np.array(pose[4:7], dtype=np.float64)
    c = -R.T @ t                                    # camera centre in the world
    return c[None, :] + ts[:, None] * (R.T @ d)[None, :]


@pytest.mark.parametrize("vs,frame", [(0.01, 0), (0.02, 17), (0.005, 33)])
def test_frustum_blocks_covers_every_voxel_a_strip_can_read(vs, frame):
    rng = np.random.default_rng(5 + frame)
    f = synthetic.frame("room", frame, scale=0.25)
    H, W = f["depth"].shape
    K, pose = f["intrinsics"], [float(v) for v in f["pose"]]
    max_depth = 4.0
    # a directory with a block everywhere the rays go: the lattice of all blocks within reach of the camera
    R = multi._quat_matrix(pose[:4])
    cam = -R.T @ np.array(pose[4:7])
    reach = int(np.ceil((max_depth + 1.0) / (8 * vs))) + 2
    c0 = np.floor(cam / (8 * vs)).astype(int)
    g = np.arange(-reach, reach + 1)
    if len(g) ** 3 > 4_000_000:
        g = g[::2]                                   # (5 mm voxels: every other block is plenty)
    bx, by, bz = np.meshgrid(g + c0[0], g + c0[1], g + c0[2], indexing="ij")
    blocks = np.zeros(bx.size, dtype=[("x", np.int16), ("y", np.int16), ("z", np.int16)])
    blocks["x"], blocks["y"], blocks["z"] = bx.ravel(), by.ravel(), bz.ravel()
    have = set(zip(blocks["x"].tolist(), blocks["y"].tolist(), blocks["z"].tolist()))
    for world in (2, 4, 5):
        for rows in multi.strip_rows(H, world):
            if rows[0] >= rows[1]:
                continue
            mask = multi.frustum_blocks(blocks, K, W, rows, pose, max_depth, vs)
            sel = set(zip(blocks["x"][mask].tolist(), blocks["y"][mask].tolist(), blocks["z"][mask].tolist()))
            assert 0 < len(sel) < len(have)          # a selection, not everything
            ts = np.arange(0.0, max_depth + 3 * vs, vs * 0.9)
            for _ in range(12):
                px = rng.integers(0, W)
                py = rng.integers(rows[0], rows[1])
                pts = _ray_points(K, pose, float(px), float(py), ts) / vs
                vox = np.rint(pts).astype(int)
                for off in ((0, 0, 0), (2, 2, 2), (-2, -2, -2), (2, -2, 2), (-2, 2, -2)):
                    b = (vox + np.array(off)) >> 3
                    need = set(map(tuple, b.tolist())) & have
                    assert need <= sel, (rows, px, py, sorted(need - sel)[:3])
