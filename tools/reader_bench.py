#!/usr/bin/env python3
"""Throughput of the folder dataset reader (C++ host layer) on a generated 640x480 dataset, next to
the Python oracle's decoder: tools/reader_bench.py [frames]"""
import subprocess, sys, tempfile, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
from make_dataset import write_folder
import dataset_oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
with tempfile.TemporaryDirectory() as d:
    write_folder(Path(d) / "ds", n=n, scale=1.0, fast=True)
    exe = ROOT / "ra-slam_amd/host/build/ratsdf_offline_eval"
    subprocess.run(["make", "-C", str(ROOT / "ra-slam_amd/host")], check=True, capture_output=True)
    for threads in ("1", "4", "8", "16"):
        r = subprocess.run([str(exe), str(Path(d) / "ds"), "--reader-only", "--threads", threads],
                           capture_output=True, text=True)
        print(f"C++ reader, {threads} thread(s):", r.stderr.strip().splitlines()[-1])
    lib = ROOT / "ra-slam_amd/csrc/build/libratsdf.so"
    for threads in ("1", "8"):
        r = subprocess.run([str(exe), str(Path(d) / "ds"), "--lib", str(lib), "--voxel", "0.005", "--max-depth",
                            "4", "--threads", threads], capture_output=True, text=True)
        print(f"whole harness on the HIP engine, {threads} decoder thread(s):",
              (r.stderr.strip().splitlines() or ["failed"])[-1])
    ds = O.read_folder(Path(d) / "ds")
    t = time.perf_counter()
    for i in range(min(n, 6)):
        ds["frame"](i)
    t = time.perf_counter() - t
    print(f"Python oracle decoder: {min(n, 6) / t:.1f} frames/s")
