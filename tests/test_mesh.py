"""Marching-cubes export (TSDFGrid::GatherValidMesh / TSDFSystem::DownloadAllMesh)."""
import re
from pathlib import Path

import numpy as np
import pytest

from ratsdf import synthetic

ROOT = Path(__file__).resolve().parent.parent
CORNER = [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)]
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def load_cases(path):
    return re.findall(r'"([0-9a-b]*)"', "".join(l for l in path.read_text().splitlines()
                                                  if not l.lstrip().startswith("//")))


@pytest.mark.parametrize("path", [ROOT / "oracle" / "mc_cases.inc",
                                  ROOT / "ra-slam_amd" / "csrc" / "mc_cases.inc"])
def test_case_table_is_a_valid_triangulation(path):
    """Every sign pattern uses exactly its sign-changing edges, in whole triangles; the oracle's and
    the engine's copies are the same data."""
    cases = load_cases(path)
    assert len(cases) == 256
    for c, s in enumerate(cases):
        assert len(s) % 3 == 0 and len(s) <= 15
        used = {int(ch, 16) for ch in s}
        crossing = {e for e, (a, b) in enumerate(EDGE) if ((c >> a) & 1) != ((c >> b) & 1)}
        assert used == crossing, f"case {c}"
    assert cases == load_cases(ROOT / "oracle" / "mc_cases.inc")
    # complementary sign patterns cut the same edges
    for c in range(256):
        assert {int(ch, 16) for ch in cases[c]} == {int(ch, 16) for ch in cases[255 - c]}


def _integrate_wall(e, n=14):
    f = synthetic.frame("wall", 0, scale=0.25)
    for _ in range(n):  # marching cubes only uses voxels with weight > 10 (voxel_tsdf.cu:600)
        e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    return f


def test_oracle_mesh_of_a_wall(make_oracle, tmp_path):
    vs = 0.02
    e = make_oracle(vs, 6 * vs, threads=4)
    _integrate_wall(e)
    v, tri, p = e.gather_valid_mesh()
    assert len(v) > 1000 and len(tri) > 1000 and len(p) == len(v)
    assert tri.min() >= 0 and tri.max() < len(v)
    assert len(np.unique(tri.reshape(-1))) == len(v)          # every exported vertex is referenced
    assert np.isfinite(v).all()
    assert np.abs(v[:, 2] - 2.0).max() < 1.5 * vs             # the wall is the plane z = 2 m
    assert ((p > 0) & (p < 1)).all()
    # file formats of DownloadAllMesh (tsdf_module.cc:66-86)
    fv, fi, fp = tmp_path / "v.bin", tmp_path / "i.bin", tmp_path / "p.bin"
    e.download_all_mesh(fv, fi, fp)
    assert np.array_equal(np.fromfile(fv, "<f4").reshape(-1, 3), v)
    assert np.array_equal(np.fromfile(fi, "<i4").reshape(-1, 3), tri)
    assert np.array_equal(np.fromfile(fp, "<f4"), p)
    # an empty map exports an empty mesh
    v0, t0, p0 = make_oracle(vs, 6 * vs).gather_valid_mesh()
    assert len(v0) == 0 and len(t0) == 0 and len(p0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["wall", "room"])
def test_engine_mesh_matches_oracle(scene, make_engine, make_oracle):
    vs = 0.02
    gpu, cpu = make_engine(vs, 6 * vs), make_oracle(vs, 6 * vs, threads=8)
    frames = [synthetic.frame(scene, 0, scale=0.25)] * 12 + synthetic.stream(scene, 4, scale=0.25)
    for f in frames:
        for e in (gpu, cpu):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    gv, gt, gp = gpu.gather_valid_mesh()
    cv, ct, cp = cpu.gather_valid_mesh()
    assert len(cv) > 500
    assert np.array_equal(gt, ct), "triangle index buffers differ"
    assert gv.shape == cv.shape and np.max(np.abs(gv - cv), initial=0) <= 1e-6
    assert np.max(np.abs(gp - cp), initial=0) <= 1e-4
    v0, t0, p0 = make_engine(vs, 6 * vs).gather_valid_mesh()
    assert len(v0) == 0 and len(t0) == 0
