"""ScanNet .sens reader (ra-slam_amd/host: scannet_sens_reader, decode_jpeg, resize_rgb_linear).

The pin is the REFERENCE's own loader: third_party/scannet/sensorData.hpp + RGBDFrame.cc + the vendored
stb_image.h, compiled from /root/reference into oracle/_ref/ref_sens_dump (`make -C oracle ref`, build
container only).  What it decodes from small synthetic streams is committed as tests/golden/sens_ref_*.npz
and tiny_sens_ref.npz (tests/golden/make_sens_ref_golden.py): colour bytes BEFORE cv::resize, 16-bit
depth, stored poses, calibration.  The reader must reproduce them byte for byte
(test_reference_loader_fixtures: runs anywhere).  The remaining tests write fresh streams and compare
with the reference loader run on the spot, so they need oracle/_ref (skipped where it is absent).
After the decode, cv::resize's 8-bit bilinear kernel is checked against its numpy restatement
(oracle/segmentation_oracle.py; parity unpinned: no OpenCV in this image) and SE3::Inverse against
ratsdf.pose (Eigen restated)."""
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


REF_EXE = ROOT / "oracle" / "_ref" / "ref_sens_dump"
needs_ref = pytest.mark.skipif(not REF_EXE.exists(), reason="oracle/_ref not built (no reference tree here)")


def dump(sens, tmp):
    """-> meta, 640x480 colour, depth in metres, poses, colour before the resize"""
    tmp.mkdir(parents=True, exist_ok=True)
    r = subprocess.run([str(build()), str(sens), "--reader-only", "--dump-frames", str(tmp), "--dump-raw-color",
                        "--threads", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    head, ext = (tmp / "meta.txt").read_text().strip().splitlines()
    w, h, n = (int(v) for v in head.split()[:3])
    meta = dict(width=w, height=h, n=n, intrinsics=np.array(head.split()[3:7], dtype=np.float32),
                factor=np.float32(head.split()[7]), extrinsics=np.array(ext.split(), dtype=np.float32))
    rgb = [np.fromfile(tmp / f"{i}.rgb", dtype=np.uint8).reshape(h, w, 3) for i in range(n)]
    depth = [np.fromfile(tmp / f"{i}.depth", dtype=np.float32).reshape(h, w) for i in range(n)]
    poses = np.fromfile(tmp / "poses.bin", dtype=np.float32).reshape(n, 7)
    cw, ch = (int(v) for v in (tmp / "raw_meta.txt").read_text().split())
    raw = [np.fromfile(tmp / f"{i}.color", dtype=np.uint8).reshape(ch, cw, 3) for i in range(n)]
    return meta, rgb, depth, poses, raw


def reference_loader(sens, tmp):
    """the reference's own loader on the same stream: colour before the resize, 16-bit depth"""
    tmp.mkdir(parents=True, exist_ok=True)
    subprocess.run([str(REF_EXE), str(sens), str(tmp)], check=True, timeout=120)
    cw, ch, dw, dh, n = (int(v) for v in (tmp / "meta.txt").read_text().split()[:5])
    color = [np.fromfile(tmp / f"{i}.color", dtype=np.uint8).reshape(ch, cw, 3) for i in range(n)]
    depth = [np.fromfile(tmp / f"{i}.depth", dtype=np.uint16).reshape(dh, dw) for i in range(n)]
    return color, depth


def expected_pose(c2w):
    """scannet_sens_reader.cc:68-75: SE3(matrix).Inverse() with the reference's (Eigen's) arithmetic"""
    from ratsdf import pose as P
    return P.invert(P.pose_from_matrix(c2w.astype(np.float32)))


def check_stream(frames, ref_color, meta, rgb, depth, poses, raw, depth_shift=1000.0):
    import segmentation_oracle as O
    assert (meta["width"], meta["height"], meta["n"]) == (640, 480, len(frames))
    assert np.allclose(meta["extrinsics"], [0, 0, 0, 1, 0, 0, 0])
    for i, f in enumerate(frames):
        full = ref_color[i] if ref_color is not None else f["rgb"]
        assert np.array_equal(raw[i], full), f"frame {i}: decoded colour differs from the reference loader's " \
            f"in {int((raw[i] != full).sum())} bytes (max {int(np.abs(raw[i].astype(int) - full.astype(int)).max())})"
        want = O.resize_u8_linear(full, 480, 640)
        assert np.array_equal(rgb[i], want), f"frame {i}: colour differs by up to " \
            f"{int(np.abs(rgb[i].astype(int) - want.astype(int)).max())}"
        # eval_one.cc / offline_eval.cc:74: convertTo(CV_32FC1, 1. / factor)
        want_d = f["depth"].astype(np.float32) * np.float32(1.0 / depth_shift)
        assert np.array_equal(depth[i], want_d)
        assert np.allclose(poses[i], expected_pose(f["cam_to_world"]), atol=1e-6)


@needs_ref
@pytest.mark.parametrize("subsampling,restart,quality", [("4:2:0", 0, 88), ("4:2:2", 0, 88), ("4:4:4", 0, 88),
                                                         ("4:2:0", 2, 88), ("4:1:1", 0, 75), ("4:2:0", 1, 25),
                                                         ("4:2:2", 3, 100)])
def test_sens_stream_jpeg_variants(tmp_path, subsampling, restart, quality):
    """fresh streams, reference loader run on the spot (build container)"""
    import make_sens as M
    frames = M.synthetic_frames(2, color_hw=(121, 163), seed=quality)   # odd sizes: partial MCUs, odd chroma edge
    kw = dict(quality=quality, subsampling=subsampling, restart_rows=restart)
    sens = tmp_path / "s.sens"
    M.write_sens(sens, frames, jpeg_kw=kw)
    ref_color, ref_depth = reference_loader(sens, tmp_path / "ref")
    meta, rgb, depth, poses, raw = dump(sens, tmp_path / "dump")
    assert np.allclose(meta["intrinsics"], [577.87, 577.87, 319.5, 239.5])
    assert meta["factor"] == np.float32(1000.0)
    check_stream(frames, ref_color, meta, rgb, depth, poses, raw)
    for i, f in enumerate(frames):
        assert np.array_equal(ref_depth[i], f["depth"])      # (stb's inflate == zlib's, as it must)


@needs_ref
@pytest.mark.parametrize("hw", [(1, 1), (1, 9), (9, 1), (2, 3), (8, 8), (16, 16), (17, 33), (15, 31)])
def test_sens_stream_small_and_edge_sizes(tmp_path, hw):
    """one-sample chroma rows / columns, images smaller than an MCU, exact and inexact MCU multiples"""
    import make_sens as M
    for k, sub in enumerate(("4:2:0", "4:2:2", "4:4:4", "4:1:1")):
        frames = M.synthetic_frames(1, color_hw=hw, seed=5 + k)
        sens = tmp_path / f"s{k}.sens"
        M.write_sens(sens, frames, jpeg_kw=dict(quality=93, subsampling=sub))
        ref_color, _ = reference_loader(sens, tmp_path / f"ref{k}")
        raw = dump(sens, tmp_path / f"dump{k}")[4]
        assert np.array_equal(raw[0], ref_color[0]), (hw, sub)


@needs_ref
def test_sens_stream_scannet_sized_colour(tmp_path):
    """1296 x 968 colour (the ScanNet recorder's size) resized to 640 x 480."""
    import make_sens as M
    frames = M.synthetic_frames(1, color_hw=(968, 1296))
    sens = tmp_path / "s.sens"
    M.write_sens(sens, frames, jpeg_kw=dict(quality=92))
    ref_color, _ = reference_loader(sens, tmp_path / "ref")
    meta, rgb, depth, poses, raw = dump(sens, tmp_path / "dump")
    check_stream(frames, ref_color, meta, rgb, depth, poses, raw)


FIXTURES = sorted(p.name for p in (ROOT / "tests" / "golden").glob("sens_ref_*.npz"))


@pytest.mark.parametrize("name", FIXTURES)
def test_reference_loader_fixtures(tmp_path, name):
    """Committed streams + what the reference's loader (stb_image JPEG / zlib, sensorData.hpp container
    parser) returned for them: colour bytes before the resize, 16-bit depth, poses, calibration."""
    import zlib
    fx = np.load(ROOT / "tests" / "golden" / name)
    sens = tmp_path / "s.sens"
    sens.write_bytes(fx["sens"].tobytes())
    meta, rgb, depth, poses, raw = dump(sens, tmp_path / "dump")
    cw, ch, dw, dh, n, shift, fx_, fy_, cx_, cy_ = fx["meta"]
    assert (meta["n"], raw[0].shape[1], raw[0].shape[0]) == (int(n), int(cw), int(ch))
    assert np.array_equal(meta["intrinsics"], np.array([fx_, fy_, cx_, cy_], dtype=np.float32))
    assert meta["factor"] == np.float32(shift)
    for i in range(int(n)):
        assert zlib.crc32(raw[i].tobytes()) == int(fx["color_crc"][i]), f"{name} frame {i}: colour bytes differ"
        if f"color{i}" in fx:
            assert np.array_equal(raw[i], fx[f"color{i}"])
        d16 = np.round(depth[i] * np.float32(shift)).astype(np.uint16)   # exact: the scaling is one multiply
        assert zlib.crc32(d16.tobytes()) == int(fx["depth_crc"][i]), f"{name} frame {i}: depth differs"
        assert np.allclose(poses[i], expected_pose(fx["poses"][i].reshape(4, 4)), atol=1e-6)
    if "color0_rows" in fx:
        assert np.array_equal(raw[0][:fx["color0_rows"].shape[0]], fx["color0_rows"])
    assert np.array_equal(depth[0], fx["depth0"].astype(np.float32) * np.float32(1.0 / shift))
    assert len(FIXTURES) >= 8


def test_sens_raw_colour_and_raw_depth(tmp_path):
    import make_sens as M
    frames = M.synthetic_frames(2, color_hw=(480, 640))      # no resize needed
    sens = tmp_path / "s.sens"
    M.write_sens(sens, frames, color_type=0, depth_type=0, depth_shift=500.0)
    meta, rgb, depth, poses, raw = dump(sens, tmp_path / "dump")
    check_stream(frames, None, meta, rgb, depth, poses, raw, depth_shift=500.0)


def test_committed_fixture(tmp_path):
    """tests/golden/tiny.sens + tiny_sens_ref.npz (the stream as the REFERENCE's loader decodes it:
    tests/golden/make_sens_ref_golden.py): colour before the resize byte for byte, then the resized
    frames against the numpy restatement of cv::resize applied to the reference's colour."""
    import zlib
    import segmentation_oracle as O
    exp = np.load(ROOT / "tests" / "golden" / "tiny_sens_ref.npz")
    meta, rgb, depth, poses, raw = dump(ROOT / "tests" / "golden" / "tiny.sens", tmp_path / "dump")
    assert meta["n"] == len(exp["depth_crc"]) == 3
    for i in range(meta["n"]):
        assert np.array_equal(raw[i], exp[f"color{i}"]), f"frame {i}: {int((raw[i] != exp[f'color{i}']).sum())} bytes"
        assert np.array_equal(rgb[i], O.resize_u8_linear(exp[f"color{i}"], 480, 640))
        d16 = np.round(depth[i] * np.float32(1000.0)).astype(np.uint16)
        assert zlib.crc32(d16.tobytes()) == int(exp["depth_crc"][i])
        assert np.allclose(poses[i], expected_pose(exp["poses"][i].reshape(4, 4)), atol=1e-6)
    assert np.array_equal(depth[0], exp["depth0"].astype(np.float32) * np.float32(1.0 / 1000.0))


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
    # a 64-bit length field that wraps `offset + length` (sens.cc need()): reported, not read out of bounds
    import struct
    blob = bytearray(good.read_bytes())
    name_len_at = 4
    huge = tmp_path / "huge.sens"
    blob[name_len_at:name_len_at + 8] = struct.pack("<Q", 0xFFFFFFFFFFFFFFF0)
    huge.write_bytes(bytes(blob))
    r = subprocess.run([str(build()), str(huge), "--reader-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr
    # colour-size field of frame 0 pushed past the end of the file
    blob = bytearray(good.read_bytes())
    frame0 = 4 + 8 + len(b"synthetic") + 4 * 64 + 6 * 4 + 4 + 8      # header, then cam_to_world + 2 time stamps
    csize_at = frame0 + 64 + 16
    blob[csize_at:csize_at + 8] = struct.pack("<Q", 0xFFFFFFFFFFFFFF00)
    huge.write_bytes(bytes(blob))
    r = subprocess.run([str(build()), str(huge), "--reader-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr


def test_malformed_jpeg_streams_are_reported(tmp_path):
    """JPEG streams cut inside every kind of segment, a stream ending in fill bytes, an SOS of length 2:
    an error naming the frame, never a read past the buffer (the reader runs on whole untrusted files)."""
    import make_sens as M
    import struct
    frames = M.synthetic_frames(1, color_hw=(40, 56))
    jpeg = M.encode_jpeg(frames[0]["rgb"], quality=80)
    sos = jpeg.index(b"\xff\xda")
    bad = [jpeg[:k] for k in (2, 3, 4, 20, sos, sos + 2, sos + 3, sos + 4, len(jpeg) // 2)]
    bad.append(jpeg[:sos] + b"\xff\xff\xff")                                  # fill bytes, then nothing
    bad.append(jpeg[:sos] + b"\xff\xda\x00\x02")                              # SOS with no payload
    bad.append(jpeg[:sos] + b"\xff\xda\x00\x02" + jpeg[sos + 4:])
    for k, b in enumerate(bad):
        sens = tmp_path / f"b{k}.sens"
        M.write_sens(sens, frames, color_type=0)                              # container with raw colour ...
        blob = bytearray(sens.read_bytes())
        # ... rewritten as a JPEG stream: patch the colour type and the frame's colour payload
        hdr = 4 + 8 + len(b"synthetic") + 4 * 64
        blob[hdr:hdr + 4] = struct.pack("<i", 2)
        frame0 = hdr + 6 * 4 + 4 + 8
        csize_at = frame0 + 64 + 16
        (csize, dsize) = struct.unpack("<QQ", blob[csize_at:csize_at + 16])
        payload_at = csize_at + 16
        blob[csize_at:csize_at + 8] = struct.pack("<Q", len(b))
        blob[payload_at:payload_at + csize] = b
        sens.write_bytes(bytes(blob))
        r = subprocess.run([str(build()), str(sens), "--reader-only", "--dump-frames", str(tmp_path)],
                           capture_output=True, text=True, timeout=60)
        # cut inside the entropy-coded data the decoder, like stb, returns what it has; everything cut
        # earlier is an error.  Either way: no crash.
        assert r.returncode in (0, 1), (k, r.returncode, r.stderr)
        if k != 8:
            assert r.returncode == 1 and "JPEG" in r.stderr, (k, r.stderr)
