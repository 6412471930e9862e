#!/usr/bin/env python3
"""GPU box: only the tsdf_system_path leg of bench.py (TSDFSystem::Integrate from pageable images, C++)."""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "ra-slam_amd"))
import bench
from ratsdf import synthetic
frames = [synthetic.frame("room", i, noise=True, holes=True) for i in range(24)]
for rep in range(3):
    out = bench.bench_tsdf_system(frames, 4.0, 0.005)
    print(json.dumps({k: (v["frames_per_s"], v["h2d_gbps"]) for k, v in out.items() if isinstance(v, dict)}))
