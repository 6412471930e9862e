#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ with the CPU oracle.

The reference cannot be built or run here (CUDA + Eigen + OpenCV) and has no fixtures of its own for
TSDFGrid::Integrate, so these goldens pin the ORACLE's output on seeded synthetic frames: they catch
drift of the oracle (compiler, libm, edits) and give the HIP engine a second, file-based target.
Inputs are regenerated from seeds by ratsdf.synthetic; only expected outputs are stored.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
sys.path.insert(0, str(ROOT / "tests"))

CASES = {
    # name: (scene, frames, cam, scale, voxel, semantic, noise, holes)
    "wall_80x60_2cm_3f": ("wall", 3, "scannet", 0.125, 0.02, True, False, False),
    "room_160x120_2cm_4f": ("room", 4, "scannet", 0.25, 0.02, True, True, True),
    "sphere_160x120_1cm_2f_nosem": ("sphere", 2, "tum", 0.25, 0.01, False, False, False),
}


def run_case(engine_factory, spec):
    from ratsdf import synthetic
    scene, n, cam, scale, vs, sem, noise, holes = spec
    e = engine_factory(vs, 6 * vs)
    stats = []
    for i in range(n):
        f = synthetic.frame(scene, i, cam=cam, scale=scale, semantic=sem, noise=noise, holes=holes)
        e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
        s = e.last_frame_stats()
        stats.append([s[k] for k in ("visible_blocks", "updated_voxels", "allocated_blocks",
                                     "deleted_blocks", "active_blocks")])
    ei, bl = e.dump_directory()
    nf, heap = e.dump_heap()
    tsdf, rgbw, prob = e.dump_voxels(bl["idx"])
    out = dict(stats=np.array(stats, dtype=np.int32), entry_index=ei,
               blocks=np.stack([bl["x"], bl["y"], bl["z"], bl["offset"]], axis=1).astype(np.int16),
               pool_idx=bl["idx"].astype(np.int32), num_free=np.int32(nf),
               heap_tail=heap[max(0, nf - 64):nf].astype(np.int32),
               tsdf=tsdf.astype(np.float32), prob=prob.astype(np.float32),
               rgbw=rgbw.view(np.uint32).reshape(rgbw.shape))
    e.close()
    return out


def main():
    from oracle_binding import load_oracle
    from ratsdf._abi import Engine
    lib = load_oracle()
    for name, spec in CASES.items():
        out = run_case(lambda vs, tr: Engine(lib, vs, tr), spec)
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
