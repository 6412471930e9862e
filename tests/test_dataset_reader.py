"""Folder dataset reader of the C++ host layer (ra-slam_amd/host: folder_reader, PNG decoder,
camera_config.yaml / trajectory.txt parsing, depth scaling) against the independent Python
restatement (oracle/dataset_oracle.py) and the committed golden folder.  No GPU needed: the harness
runs in --reader-only mode, or integrates through the CPU oracle library."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "ra-slam_amd" / "host"
EXE = HOST / "build" / "ratsdf_offline_eval"
GOLD = ROOT / "tests" / "golden" / "folder_dataset"
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))


def build():
    subprocess.run(["make", "-C", str(HOST)], check=True, capture_output=True)
    return EXE


def build_oracle_bound():
    """Test-only build of the harness whose host layer binds the CPU oracle's copy of the ABI
    (RATSDF_ABI_PREFIX); the product binary has no switch for that."""
    subprocess.run(["make", "-C", str(HOST), "test-oracle-eval"], check=True, capture_output=True)
    return HOST / "build" / "test_offline_eval_oracle"


def dump(folder, tmp):
    tmp.mkdir(parents=True, exist_ok=True)
    r = subprocess.run([str(build()), str(folder), "--reader-only", "--dump-frames", str(tmp)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    head, ext = (tmp / "meta.txt").read_text().strip().splitlines()
    w, h, n = (int(v) for v in head.split()[:3])
    meta = dict(width=w, height=h, n=n, intrinsics=np.array(head.split()[3:7], dtype=np.float32),
                factor=np.float32(head.split()[7]), extrinsics=np.array(ext.split(), dtype=np.float32))
    rgb = [np.fromfile(tmp / f"{i}.rgb", dtype=np.uint8).reshape(h, w, 3) for i in range(n)]
    depth = [np.fromfile(tmp / f"{i}.depth", dtype=np.float32).reshape(h, w) for i in range(n)]
    poses = np.fromfile(tmp / "poses.bin", dtype=np.float32).reshape(n, 7)
    return meta, rgb, depth, poses


def test_golden_folder_matches_expected(tmp_path):
    meta, rgb, depth, poses = dump(GOLD, tmp_path / "d")
    exp = np.load(GOLD / "expected.npz")
    assert (meta["width"], meta["height"], meta["n"]) == (32, 24, 3)
    assert np.array_equal(meta["intrinsics"], exp["intrinsics"])
    assert np.array_equal(meta["extrinsics"], exp["extrinsics"])
    assert meta["factor"] == exp["factor"]
    assert np.array_equal(np.stack(rgb), exp["rgb"])            # bytes
    assert np.array_equal(np.stack(depth), exp["depth"])        # float32, bit-exact
    assert np.array_equal(poses, exp["poses"])                  # float32, bit-exact
    # hand-checked values: extrinsics = 90 degrees about z, translation as written in the yaml
    assert np.allclose(exp["extrinsics"], [0, 0, 2 ** -0.5, 2 ** -0.5, 0.05, -0.02, 0.1], atol=1e-7)
    assert list(exp["ids"]) == [3, 5, 7]


def test_oracle_decodes_the_golden_folder_identically():
    import dataset_oracle as O
    ds = O.read_folder(GOLD)
    exp = np.load(GOLD / "expected.npz")
    for i in range(3):
        rgb, depth = ds["frame"](i)
        assert np.array_equal(rgb, exp["rgb"][i]) and np.array_equal(depth, exp["depth"][i])
    assert np.array_equal(np.array(ds["poses"], dtype=np.float32), exp["poses"])


@pytest.mark.parametrize("factor,ext", [(5000.0, None), (1000.0, [[1, 0, 0, 0.1], [0, 0, -1, 0], [0, 1, 0, 0.3], [0, 0, 0, 1]])])
def test_generated_folder_roundtrip(factor, ext, tmp_path):
    """A fresh dataset (all five PNG filters, split IDAT, 16-bit depth) read by both implementations;
    the decoded depth must equal the quantised source exactly."""
    import dataset_oracle as O
    from make_dataset import write_folder
    frames = write_folder(tmp_path / "ds", n=3, scale=0.1, factor=factor, extrinsics=ext, scene="sphere")
    meta, rgb, depth, poses = dump(tmp_path / "ds", tmp_path / "d")
    ds = O.read_folder(tmp_path / "ds")
    alpha = np.float32(1.0 / float(np.float32(factor)))
    for i, f in enumerate(frames):
        o_rgb, o_depth = ds["frame"](i)
        assert np.array_equal(rgb[i], f["rgb"]) and np.array_equal(o_rgb, f["rgb"])
        d16 = np.clip(np.round(f["depth"].astype(np.float64) * factor), 0, 65535).astype(np.uint16)
        assert np.array_equal(depth[i], d16.astype(np.float32) * alpha)
        assert np.array_equal(depth[i], o_depth)
    assert np.array_equal(poses, np.array(ds["poses"], dtype=np.float32))
    assert np.array_equal(meta["extrinsics"], np.array(ds["extrinsics"], dtype=np.float32))


def test_color_conversions_follow_imread(tmp_path):
    """cv::imread default flag: grey replicated, alpha dropped, 16-bit samples keep the high byte."""
    import dataset_oracle as O
    from make_dataset import write_png
    rng = np.random.default_rng(5)
    cases = {
        "grey8": rng.integers(0, 256, (6, 7), dtype=np.uint8),
        "ga8": rng.integers(0, 256, (6, 7, 2), dtype=np.uint8),
        "rgba8": rng.integers(0, 256, (6, 7, 4), dtype=np.uint8),
        "rgb16": rng.integers(0, 65536, (6, 7, 3), dtype=np.uint16),
    }
    folder = tmp_path / "ds"
    folder.mkdir()
    (folder / "camera_config.yaml").write_text(
        "Camera.fx: 10\nCamera.fy: 10\nCamera.cx: 3\nCamera.cy: 2.5\ndepthmap_factor: 1000\n")
    lines = []
    for i, (name, arr) in enumerate(cases.items()):
        write_png(folder / f"{i}_rgb.png", arr)
        write_png(folder / f"{i}_depth.png", np.full((6, 7), 1000 + i, dtype=np.uint16))
        lines.append(f"{i} 1 0 0 0 0 1 0 0 0 0 1 0")
    (folder / "trajectory.txt").write_text("\n".join(lines) + "\n")
    meta, rgb, depth, poses = dump(folder, tmp_path / "d")
    for i, (name, arr) in enumerate(cases.items()):
        a = arr if arr.ndim == 3 else arr[..., None]
        if a.dtype == np.uint16:
            a = (a >> 8).astype(np.uint8)
        exp = np.repeat(a[..., :1], 3, axis=2) if a.shape[2] <= 2 else a[..., :3]
        assert np.array_equal(rgb[i], exp), name
        assert np.array_equal(O.to_rgb8(*O.read_png(folder / f"{i}_rgb.png")), exp), name
        assert np.all(depth[i] == np.float32(1000 + i) * np.float32(0.001))
    assert np.array_equal(poses, np.tile(np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float32), (4, 1)))


def test_bad_inputs_fail_loudly(tmp_path):
    folder = tmp_path / "ds"
    folder.mkdir()
    r = subprocess.run([str(build()), str(folder), "--reader-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "camera_config.yaml" in r.stderr
    r = subprocess.run([str(build()), "scene.sens"], capture_output=True, text=True)
    assert r.returncode == 1 and "could not open" in r.stderr     # .sens streams: tests/test_sens_reader.py


def test_offline_eval_on_cpu_oracle_matches_direct_integration(tmp_path, oracle_lib, make_oracle):
    """The whole harness (reader -> TSDFSystem -> DownloadAll) through the CPU oracle library equals
    integrating the Python-decoded frames directly: same valid voxels, same values."""
    import dataset_oracle as O
    from make_dataset import write_folder
    from ratsdf import pose as P
    write_folder(tmp_path / "ds", n=4, scale=0.1, factor=1000.0, scene="room")
    out = tmp_path / "map.bin"
    r = subprocess.run([str(build_oracle_bound()), str(tmp_path / "ds"), "--lib", str(oracle_lib.path),
                        "--voxel", "0.04", "--max-depth", "6", "--download-all", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("tsdf", "<f4"),
                                            ("prob", "<f4")]))
    ds = O.read_folder(tmp_path / "ds")
    cpu = make_oracle(0.04, 0.24)
    for i in range(4):
        rgb, depth = ds["frame"](i)
        # offline_eval.cc:57 hands the reader's extrinsics to TSDFSystem, which composes them again
        cpu.integrate(rgb, depth, None, None, 6.0, ds["intrinsics"], P.compose(ds["extrinsics"], ds["poses"][i]))
    exp = cpu.gather_valid_semantic()
    assert len(got) == len(exp) and len(got) > 0
    key = lambda a: np.lexsort((a["z"], a["y"], a["x"]))
    g, e = got[key(got)], exp[key(exp)]
    for f in ("x", "y", "z", "tsdf", "prob"):
        assert np.array_equal(g[f], e[f]), f
