"""Writes a small synthetic ScanNet .sens stream (third_party/scannet/sensorData.hpp, version 4):
header, calibration, JPEG colour frames (PIL = the IJG library) and zlib-compressed 16-bit depth
frames.  Used by tests/test_sens_reader.py and to (re)generate tests/golden/tiny.sens:

    python tests/make_sens.py tests/golden/tiny.sens
"""
import io
import struct
import sys
import zlib

import numpy as np


def synthetic_frames(n, color_hw=(121, 162), seed=3):
    """smooth colour images (so that JPEG artefacts stay small) + a depth ramp with holes + poses"""
    rng = np.random.default_rng(seed)
    ch, cw = color_hw
    frames = []
    for i in range(n):
        yy, xx = np.mgrid[0:ch, 0:cw].astype(np.float64)
        rgb = np.stack([127 + 100 * np.sin(xx / 17.0 + i) * np.cos(yy / 13.0),
                        127 + 90 * np.cos(xx / 11.0 - i) * np.sin(yy / 19.0),
                        40 + 1.2 * xx + 0.3 * yy], axis=-1)
        rgb = np.clip(rgb + rng.normal(0, 6, rgb.shape), 0, 255).astype(np.uint8)
        dy, dx = np.mgrid[0:480, 0:640]
        depth = (1200 + dx + 2 * dy + 50 * i).astype(np.uint16)
        depth[rng.random(depth.shape) < 0.02] = 0
        a = 0.1 * i
        c2w = np.array([[np.cos(a), 0, np.sin(a), 0.05 * i], [0, 1, 0, -0.02 * i],
                        [-np.sin(a), 0, np.cos(a), 0.3], [0, 0, 0, 1]], dtype=np.float32)
        frames.append(dict(rgb=rgb, depth=depth, cam_to_world=c2w))
    return frames


def encode_jpeg(rgb, quality=90, subsampling="4:2:0", restart_rows=0):
    from PIL import Image
    buf = io.BytesIO()
    kw = dict(format="JPEG", quality=quality, subsampling=subsampling)
    if restart_rows:
        kw["restart_marker_rows"] = restart_rows
    Image.fromarray(rgb).save(buf, **kw)
    return buf.getvalue()


def encode_png(img, mode=None):
    """img: H x W x 3 (RGB), H x W x 4 (RGBA), H x W uint8 (grey) or H x W uint16 (16-bit grey)"""
    from PIL import Image
    buf = io.BytesIO()
    im = Image.fromarray(img, mode) if mode else Image.fromarray(img)
    im.save(buf, format="PNG")
    return buf.getvalue()


def write_sens(path, frames, intrinsics=(577.87, 577.87, 319.5, 239.5), depth_shift=1000.0,
               color_type=2, depth_type=1, jpeg_kw=None):
    """color_type: 0 raw, 1 PNG, 2 JPEG; depth_type: 0 raw, 1 zlib (COMPRESSION_TYPE_*, include.hpp:259-270)"""
    fx, fy, cx, cy = intrinsics
    K = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    eye = np.eye(4, dtype=np.float32)
    ch, cw = frames[0]["rgb"].shape[:2]
    name = b"synthetic"
    out = [struct.pack("<I", 4), struct.pack("<Q", len(name)), name,
           (K * np.float32(2)).tobytes(), eye.tobytes(),      # colour calibration (not used by the reader)
           K.tobytes(), eye.tobytes(),                        # depth calibration
           struct.pack("<iiIIII", color_type, depth_type, cw, ch, 640, 480), struct.pack("<f", depth_shift),
           struct.pack("<Q", len(frames))]
    for i, f in enumerate(frames):
        color = (encode_jpeg(f["rgb"], **(jpeg_kw or {})) if color_type == 2 else
                 encode_png(f["rgb"]) if color_type == 1 else f["rgb"].tobytes())
        depth = zlib.compress(f["depth"].tobytes(), 6) if depth_type == 1 else f["depth"].tobytes()
        out += [f["cam_to_world"].astype(np.float32).tobytes(), struct.pack("<QQQQ", 1000 * i, 1000 * i + 3,
                                                                          len(color), len(depth)), color, depth]
    out.append(struct.pack("<Q", 0))   # no IMU frames
    with open(path, "wb") as fh:
        fh.write(b"".join(out))


if __name__ == "__main__":
    write_sens(sys.argv[1], synthetic_frames(3))
    print("wrote", sys.argv[1])
