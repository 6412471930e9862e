"""ScanNet .sens reader (ra-slam_amd/host: scannet_sens_reader, decode_jpeg, resize_rgb_linear) against
independent Python decoding of the same streams: PIL (the IJG JPEG library, whose default decoding
path jpeg.cc restates -> bit-exact), zlib, and the numpy restatement of cv::resize's 8-bit bilinear
kernel (oracle/segmentation_oracle.py).  The committed tests/golden/tiny.sens is read as well, so the
container parser is pinned by a fixture that does not depend on the PIL version installed."""
import io
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HOST = ROOT / "ra-slam_amd" / "host"
EXE = HOST / "build" / "ratsdf_offline_eval"
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))


def build():
    subprocess.run(["make", "-C", str(HOST)], check=True, capture_output=True)
    return EXE


def dump(sens, tmp):
    tmp.mkdir(parents=True, exist_ok=True)
    r = subprocess.run([str(build()), str(sens), "--reader-only", "--dump-frames", str(tmp), "--threads", "2"],
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


def expected_pose(c2w):
    """scannet_sens_reader.cc:68-75: SE3(matrix).Inverse() with the reference's (Eigen's) arithmetic"""
    from ratsdf import pose as P
    return P.invert(P.pose_from_matrix(c2w.astype(np.float32)))


def check_stream(frames, jpeg_bytes, meta, rgb, depth, poses, depth_shift=1000.0):
    import segmentation_oracle as O
    from PIL import Image
    assert (meta["width"], meta["height"], meta["n"]) == (640, 480, len(frames))
    assert np.allclose(meta["extrinsics"], [0, 0, 0, 1, 0, 0, 0])
    for i, f in enumerate(frames):
        full = np.asarray(Image.open(io.BytesIO(jpeg_bytes[i])).convert("RGB")) if jpeg_bytes else f["rgb"]
        want = O.resize_u8_linear(full, 480, 640)
        assert np.array_equal(rgb[i], want), f"frame {i}: colour differs by up to " \
            f"{int(np.abs(rgb[i].astype(int) - want.astype(int)).max())}"
        # eval_one.cc / offline_eval.cc:74: convertTo(CV_32FC1, 1. / factor)
        want_d = f["depth"].astype(np.float32) * np.float32(1.0 / depth_shift)
        assert np.array_equal(depth[i], want_d)
        assert np.allclose(poses[i], expected_pose(f["cam_to_world"]), atol=1e-6)


@pytest.mark.parametrize("subsampling,restart", [("4:2:0", 0), ("4:2:2", 0), ("4:4:4", 0), ("4:2:0", 2)])
def test_sens_stream_jpeg_variants(tmp_path, subsampling, restart):
    import make_sens as M
    frames = M.synthetic_frames(2, color_hw=(121, 163))      # odd sizes: partial MCUs, odd chroma edge
    kw = dict(quality=88, subsampling=subsampling, restart_rows=restart)
    jpegs = [M.encode_jpeg(f["rgb"], **kw) for f in frames]
    sens = tmp_path / "s.sens"
    M.write_sens(sens, frames, jpeg_kw=kw)
    meta, rgb, depth, poses = dump(sens, tmp_path / "dump")
    assert np.allclose(meta["intrinsics"], [577.87, 577.87, 319.5, 239.5])
    assert meta["factor"] == np.float32(1000.0)
    check_stream(frames, jpegs, meta, rgb, depth, poses)


def test_sens_stream_scannet_sized_colour(tmp_path):
    """1296 x 968 colour (the ScanNet recorder's size) resized to 640 x 480, grayscale JPEG too."""
    import make_sens as M
    from PIL import Image
    frames = M.synthetic_frames(1, color_hw=(968, 1296))
    jpegs = [M.encode_jpeg(f["rgb"], quality=92) for f in frames]
    sens = tmp_path / "s.sens"
    M.write_sens(sens, frames, jpeg_kw=dict(quality=92))
    meta, rgb, depth, poses = dump(sens, tmp_path / "dump")
    check_stream(frames, jpegs, meta, rgb, depth, poses)


def test_sens_raw_colour_and_raw_depth(tmp_path):
    import make_sens as M
    frames = M.synthetic_frames(2, color_hw=(480, 640))      # no resize needed
    sens = tmp_path / "s.sens"
    M.write_sens(sens, frames, color_type=0, depth_type=0, depth_shift=500.0)
    meta, rgb, depth, poses = dump(sens, tmp_path / "dump")
    check_stream(frames, None, meta, rgb, depth, poses, depth_shift=500.0)


def test_committed_fixture(tmp_path):
    """tests/golden/tiny.sens + tiny_sens_expected.npz (decoded once with PIL / zlib / the numpy resize
    and committed; generator: the snippet in tests/golden/README_sens.txt): the reader reproduces them
    byte for byte whatever PIL is installed."""
    import zlib
    exp = np.load(ROOT / "tests" / "golden" / "tiny_sens_expected.npz")
    meta, rgb, depth, poses = dump(ROOT / "tests" / "golden" / "tiny.sens", tmp_path / "dump")
    assert meta["n"] == len(exp["rgb_crc"])
    assert np.array_equal(rgb[0], exp["rgb0"])
    assert np.array_equal(depth[0], exp["depth0_u16"].astype(np.float32) * np.float32(1.0 / 1000.0))
    for i in range(meta["n"]):   # the other frames by checksum (keeps the fixture small)
        assert zlib.crc32(rgb[i].tobytes()) == int(exp["rgb_crc"][i])
        assert zlib.crc32(depth[i].tobytes()) == int(exp["depth_crc"][i])
    assert np.allclose(poses, exp["poses"], atol=1e-6)


def test_bad_streams_are_reported(tmp_path):
    import make_sens as M
    p = tmp_path / "bad.sens"
    p.write_bytes(b"\x05\x00\x00\x00" + b"\x00" * 64)
    r = subprocess.run([str(build()), str(p), "--reader-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "version" in r.stderr
    good = tmp_path / "good.sens"
    M.write_sens(good, M.synthetic_frames(1, color_hw=(64, 80)))
    cut = tmp_path / "cut.sens"
    cut.write_bytes(good.read_bytes()[:400])
    r = subprocess.run([str(build()), str(cut), "--reader-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr
