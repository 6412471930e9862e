"""The offline harness (reader -> TSDFSystem -> downloads) on the HIP engine: the map it writes must
be the one the CPU oracle builds from the Python-decoded frames (needs a GPU)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
pytestmark = pytest.mark.gpu
REC = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("tsdf", "<f4"), ("prob", "<f4")])


def test_offline_eval_on_hip_engine(tmp_path, make_oracle):
    import dataset_oracle as O
    from make_dataset import write_folder
    from ratsdf import pose as P
    from test_dataset_reader import build
    write_folder(tmp_path / "ds", n=6, scale=0.25, factor=1000.0, scene="room")
    out = tmp_path / "map.bin"
    lib = ROOT / "ra-slam_amd" / "csrc" / "build" / "libratsdf.so"
    r = subprocess.run([str(build()), str(tmp_path / "ds"), "--lib", str(lib), "--voxel", "0.02",
                        "--download-all", str(out), "--download-mesh", str(tmp_path / "mesh")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=REC)
    ds = O.read_folder(tmp_path / "ds")
    cpu = make_oracle(0.02, 0.12)
    for i in range(6):
        rgb, depth = ds["frame"](i)
        cpu.integrate(rgb, depth, None, None, 6.0, ds["intrinsics"], P.compose(ds["extrinsics"], ds["poses"][i]))
    exp = cpu.gather_valid_semantic()
    assert len(got) == len(exp) and len(got) > 1000
    key = lambda a: np.lexsort((a["z"], a["y"], a["x"]))
    g, e = got[key(got)], exp[key(exp)]
    for f in ("x", "y", "z"):
        assert np.array_equal(g[f], e[f]), f
    assert np.max(np.abs(g["tsdf"] - e["tsdf"])) <= 1e-4
    assert np.all(g["prob"] == np.float32(0.5))  # no segmentation: ht = lt = 1 keeps .5 exactly
    # mesh files of DownloadAllMesh: float3 vertices, int3 indices, one probability per vertex
    v = np.fromfile(tmp_path / "mesh_vertices.bin", dtype="<f4").reshape(-1, 3)
    t = np.fromfile(tmp_path / "mesh_indices.bin", dtype="<i4").reshape(-1, 3)
    p = np.fromfile(tmp_path / "mesh_vertices_prob.bin", dtype="<f4")
    assert len(v) > 0 and len(p) == len(v) and t.min() >= 0 and t.max() < len(v)


def test_offline_eval_on_a_sens_stream(tmp_path, make_oracle):
    """A ScanNet .sens stream (committed fixture tests/golden/tiny.sens) through the harness on the HIP
    engine == the CPU oracle fed with the frames as the REFERENCE's loader decodes them
    (tests/golden/tiny_sens_ref.npz: stb_image colour, made by make_sens_ref_golden.py) + numpy resize."""
    import make_sens as M
    import segmentation_oracle as S
    from ratsdf import pose as P
    from test_dataset_reader import build
    frames = M.synthetic_frames(3, color_hw=(97, 130))          # what tiny.sens was written from
    ref = np.load(ROOT / "tests" / "golden" / "tiny_sens_ref.npz")
    out = tmp_path / "map.bin"
    lib = ROOT / "ra-slam_amd" / "csrc" / "build" / "libratsdf.so"
    r = subprocess.run([str(build()), str(ROOT / "tests" / "golden" / "tiny.sens"), "--lib", str(lib), "--voxel",
                        "0.02", "--max-depth", "4", "--download-all", str(out)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=REC)
    cpu = make_oracle(0.02, 0.12)
    K = tuple(float(np.float32(v)) for v in (577.87, 577.87, 319.5, 239.5))
    for i, f in enumerate(frames):
        rgb = S.resize_u8_linear(ref[f"color{i}"], 480, 640)
        depth = f["depth"].astype(np.float32) * np.float32(1.0 / 1000.0)
        cpu.integrate(rgb, depth, None, None, 4.0, K, P.invert(P.pose_from_matrix(f["cam_to_world"])))
    exp = cpu.gather_valid_semantic()
    assert len(got) == len(exp) and len(got) > 1000
    key = lambda a: np.lexsort((a["z"], a["y"], a["x"]))
    g, e = got[key(got)], exp[key(exp)]
    for k in ("x", "y", "z"):
        assert np.array_equal(g[k], e[k]), k
    assert np.max(np.abs(g["tsdf"] - e["tsdf"])) <= 1e-4
