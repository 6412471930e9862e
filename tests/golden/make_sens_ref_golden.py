"""Fixtures that pin the .sens reader to the REFERENCE's own loader (build container only).

    make -C oracle ref                                  # compiles third_party/scannet + stb_image from
                                                        # /root/reference into oracle/_ref/ (git-ignored)
    python tests/golden/make_sens_ref_golden.py         # writes tests/golden/sens_ref_*.npz

For every variant below a small synthetic .sens stream is written (tests/make_sens.py: PIL-encoded JPEG
colour, zlib depth), decoded by oracle/_ref/ref_sens_dump -- ml::SensorData::decompressColorAlloc /
decompressDepthAlloc, i.e. stb_image.h's JPEG and zlib decoders exactly as
utils/offline_data_provider/scannet_sens_reader.cc:44-75 calls them -- and the stream together with
what the reference returned for it is stored as one .npz.  The SSE2 build (what an x86-64 build of the
reference runs) and the STBI_NO_SIMD build must agree byte for byte, else the script stops.
tests/test_sens_reader.py::test_reference_loader_fixtures then feeds the stored streams to
ra-slam_amd/host's reader and compares bytes.  Fixtures are data: stream in, decoded arrays out.
"""
import subprocess
import sys
import tempfile
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import make_sens as M  # noqa: E402

REF = ROOT / "oracle" / "_ref"

# name -> (frames, colour (h, w), jpeg keywords, gray?); a "png" keyword selects TYPE_PNG colour and names
# what the PNG holds: rgb, rgba, gray or gray16
VARIANTS = {
    "png_rgb": (1, (45, 64), dict(png="rgb"), False),
    "png_rgba": (1, (33, 50), dict(png="rgba"), False),                                # alpha is dropped
    "png_gray": (1, (40, 40), dict(png="gray"), False),                                # replicated to 3 channels
    # (16-bit PNG: stb_image v2.08 refuses it -- "1/2/4/8-bit only" -- so there is nothing to pin)
    "420_odd": (2, (97, 130), dict(quality=85, subsampling="4:2:0"), False),          # = tiny.sens's shape
    "422_odd": (2, (121, 163), dict(quality=88, subsampling="4:2:2"), False),
    "444_restart": (2, (61, 75), dict(quality=90, subsampling="4:4:4", restart_rows=1), False),
    "420_restart": (2, (83, 101), dict(quality=70, subsampling="4:2:0", restart_rows=2), False),
    "411": (1, (50, 77), dict(quality=80, subsampling="4:1:1"), False),               # nearest-neighbour path
    "gray": (1, (70, 90), dict(quality=85), True),                                    # 1 component -> 3 equal bytes
    "420_tiny": (1, (2, 2), dict(quality=95, subsampling="4:2:0"), False),            # one chroma sample: w == 1 edge
    "420_q30": (1, (64, 96), dict(quality=30, subsampling="4:2:0"), False),           # coarse tables, clamping
    "scannet_size": (1, (968, 1296), dict(quality=92, subsampling="4:2:0"), False),   # the recorder's frame size
}


def run_ref(exe, sens, out):
    out.mkdir(parents=True, exist_ok=True)
    subprocess.run([str(exe), str(sens), str(out)], check=True)
    cw, ch, dw, dh, n = (int(v) for v in (out / "meta.txt").read_text().split()[:5])
    rest = [float(v) for v in (out / "meta.txt").read_text().split()]   # cw ch dw dh n shift fx fy cx cy
    color = [np.fromfile(out / f"{i}.color", dtype=np.uint8).reshape(ch, cw, 3) for i in range(n)]
    depth = [np.fromfile(out / f"{i}.depth", dtype=np.uint16).reshape(dh, dw) for i in range(n)]
    poses = np.fromfile(out / "poses.bin", dtype=np.float32).reshape(n, 16)
    return color, depth, poses, np.array(rest, dtype=np.float64)


def main():
    if not (REF / "ref_sens_dump").exists():
        raise SystemExit("oracle/_ref/ref_sens_dump missing: run `make -C oracle ref` where /root/reference exists")
    out_dir = ROOT / "tests" / "golden"
    for name, (n, hw, kw, gray) in VARIANTS.items():
        with tempfile.TemporaryDirectory() as td:
            td = Path(td)
            frames = M.synthetic_frames(n, color_hw=hw, seed=11)
            if gray:
                for f in frames:
                    f["rgb"] = f["rgb"][..., 1].copy()  # 2-D array -> PIL mode "L" -> 1-component JPEG
            sens = td / "s.sens"
            if "png" in kw:
                for f in frames:
                    rgb = f["rgb"]
                    if kw["png"] == "rgba":
                        f["rgb"] = np.dstack([rgb, (rgb[..., 0] // 2 + 64).astype(np.uint8)])
                    elif kw["png"] == "gray":
                        f["rgb"] = rgb[..., 1].copy()
                M.write_sens(sens, frames, color_type=1)
            else:
                M.write_sens(sens, frames, jpeg_kw=kw)
            color, depth, poses, meta = run_ref(REF / "ref_sens_dump", sens, td / "simd")
            color_s, depth_s, poses_s, _ = run_ref(REF / "ref_sens_dump_scalar", sens, td / "scalar")
            for a, b in zip(color + depth, color_s + depth_s):
                if not np.array_equal(a, b):
                    raise SystemExit(f"{name}: stb's SSE2 and scalar builds disagree")
            rec = dict(sens=np.frombuffer(sens.read_bytes(), dtype=np.uint8), meta=meta, poses=poses,
                       color_crc=np.array([zlib.crc32(c.tobytes()) for c in color], dtype=np.uint64),
                       depth_crc=np.array([zlib.crc32(d.tobytes()) for d in depth], dtype=np.uint64),
                       depth0=depth[0])
            if name == "scannet_size":      # 3.7 MB of colour: the first rows in full, the rest by checksum
                rec["color0_rows"] = color[0][:48]
            else:
                for i, c in enumerate(color):
                    rec[f"color{i}"] = c
            np.savez_compressed(out_dir / f"sens_ref_{name}.npz", **rec)
            print(f"{name:14s} {n} frame(s) {hw[1]}x{hw[0]}  ->  sens_ref_{name}.npz "
                  f"({(out_dir / f'sens_ref_{name}.npz').stat().st_size // 1024} KiB)")
    # the committed tiny.sens as well (its colour decoded by the reference's loader)
    with tempfile.TemporaryDirectory() as td:
        color, depth, poses, meta = run_ref(REF / "ref_sens_dump", out_dir / "tiny.sens", Path(td) / "t")
        np.savez_compressed(out_dir / "tiny_sens_ref.npz", meta=meta, poses=poses, depth0=depth[0],
                            depth_crc=np.array([zlib.crc32(d.tobytes()) for d in depth], dtype=np.uint64),
                            **{f"color{i}": c for i, c in enumerate(color)})
        print("tiny.sens      ->  tiny_sens_ref.npz")


if __name__ == "__main__":
    main()
