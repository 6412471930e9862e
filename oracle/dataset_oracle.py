"""CPU restatement of the reference's folder dataset reader -- TEST INFRASTRUCTURE ONLY (tests/ and
bench tooling may import it; the product never does).

Follows utils/offline_data_provider/folder_reader.cc:9-105 (camera_config.yaml, trajectory.txt,
<id>_rgb.png / <id>_depth.png) and main/offline_eval.cc:74 (depth scaling).  The reference decodes
PNGs with cv::imread (libpng) and parses YAML with yaml-cpp; neither is vendored, so this file
restates the published PNG specification (ISO/IEC 15948: zlib stream, five scanline filters) with
numpy + zlib, independently of the C++ reader under ra-slam_amd/host.  Parity unpinned: the
reference holds no fixture for its readers; the golden folder under tests/golden/folder_dataset pins
both implementations against each other and against hand-written expected values.
SE3 arithmetic (Eigen restated in float32) is shared with ratsdf.pose.
"""
import struct
import zlib
from pathlib import Path

import numpy as np


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    if pa <= pb and pa <= pc:
        return a
    return b if pb <= pc else c


def read_png(path):
    """-> (array, bit_depth).  array: H x W x C uint8 / uint16 as stored (palette expanded to RGB)."""
    data = Path(path).read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n", "not a PNG"
    at, idat, palette, ihdr = 8, b"", None, None
    while at < len(data):
        (n,) = struct.unpack(">I", data[at:at + 4])
        typ = data[at + 4:at + 8]
        body = data[at + 8:at + 8 + n]
        (crc,) = struct.unpack(">I", data[at + 8 + n:at + 12 + n])
        assert zlib.crc32(typ + body) & 0xFFFFFFFF == crc, "chunk CRC"
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            palette = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += body
        elif typ == b"IEND":
            break
        at += 12 + n
    w, h, depth, ctype, _, _, interlace = ihdr
    assert interlace == 0 and depth in (8, 16)
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = ch * depth // 8
    stride = bpp * w
    raw = zlib.decompress(idat)
    assert len(raw) == (stride + 1) * h
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        ft = raw[(stride + 1) * y]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=(stride + 1) * y + 1).astype(np.int32)
        cur = np.zeros(stride, dtype=np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                elif ft == 4:
                    p = _paeth(int(a), int(b), int(c))
                else:
                    raise ValueError("bad filter type")
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    if ctype == 3:
        return palette[out.reshape(h, w)], 8
    if depth == 16:
        arr = out.reshape(h, w, ch, 2).astype(np.uint16)
        return (arr[..., 0] << 8) | arr[..., 1], 16
    return out.reshape(h, w, ch), 8


def to_rgb8(arr, depth):
    """cv::imread(path) + COLOR_BGR2RGB: 8-bit, 3 channels, RGB."""
    if depth == 16:
        arr = (arr >> 8).astype(np.uint8)
    if arr.shape[2] <= 2:
        return np.repeat(arr[..., :1], 3, axis=2)
    return np.ascontiguousarray(arr[..., :3])


def depth_to_metres(arr, factor):
    """convertTo(CV_32FC1, 1. / factor): float multiply by (float)(1.0 / factor)."""
    alpha = np.float32(1.0 / float(np.float32(factor)))
    return arr[..., 0].astype(np.float32) * alpha


def parse_yaml(path):
    vals, pending, key = {}, None, None
    for line in Path(path).read_text().splitlines():
        line = line.split("#", 1)[0].strip()
        if not line or line.startswith("%") or line.startswith("---"):
            continue
        if pending is not None:
            pending += " " + line
            if "]" in line:
                vals[key] = pending
                pending = None
            continue
        if ":" not in line:
            continue
        key, val = (s.strip() for s in line.split(":", 1))
        if val.startswith("[") and "]" not in val:
            pending = val
            continue
        vals[key] = val.strip("\"'")
    return vals


def floats(text):
    return [float(np.float32(t)) for t in text.replace("[", " ").replace("]", " ").replace(",", " ").split()]


def read_folder(folder):
    """-> dict(width, height, intrinsics, extrinsics, factor, poses, ids) + lazy frame(i)."""
    from ratsdf import pose as P  # Eigen-restated float32 SE3 helpers

    folder = Path(folder)
    cfg = parse_yaml(folder / "camera_config.yaml")
    intr = tuple(float(np.float32(cfg[k])) for k in ("Camera.fx", "Camera.fy", "Camera.cx", "Camera.cy"))
    ext = P.identity_pose()
    if "Extrinsics" in cfg:
        ext = P.pose_from_matrix(np.array(floats(cfg["Extrinsics"]), dtype=np.float32).reshape(4, 4))
    ids, poses = [], []
    toks = (folder / "trajectory.txt").read_text().split()
    for i in range(0, len(toks) - 12, 13):
        ids.append(int(toks[i]))
        m = np.array([np.float32(t) for t in toks[i + 1:i + 13]], dtype=np.float32).reshape(3, 4)
        poses.append(P.compose(ext, P.pose_from_matrix(m)))
    factor = float(np.float32(cfg["depthmap_factor"]))

    def frame(i):
        rgb = to_rgb8(*read_png(folder / f"{ids[i]}_rgb.png"))
        darr, _ = read_png(folder / f"{ids[i]}_depth.png")
        return rgb, depth_to_metres(darr, factor)

    d0, _ = read_png(folder / f"{ids[0]}_depth.png")
    return dict(width=d0.shape[1], height=d0.shape[0], intrinsics=intr, extrinsics=ext, factor=factor,
                poses=poses, ids=ids, frame=frame)
