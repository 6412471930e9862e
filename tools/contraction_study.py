#!/usr/bin/env python3
"""How far can FMA contraction move the map?  (CPU only; test infrastructure.)

The reference is built with nvcc's default -fmad=true (CMakeLists.txt:6 sets no -fmad flag), i.e. its
binary contracts a*b+c into fused multiply-adds wherever ptxas sees fit; which ones cannot be known
without nvcc.  The oracle (and with it the HIP engine) fixes the contraction-free evaluation
(-ffp-contract=off) as the canonical one.  This script builds the SAME oracle source a second time with
-ffp-contract=fast -mfma (g++ fuses every a*b+c it finds: one of the many contraction patterns a CUDA
build could have, NOT nvcc's) and compares the two maps on the golden cases and on full-size frames:
  * directory: blocks present in one map only (a pixel pick / block rounding that flipped);
  * voxels of common blocks: max |delta tsdf|, max |delta prob|, voxels whose weight or rgb differ.
It quantifies the distance between two legal floating-point evaluations of the reference's formulas --
the size of the "parity unpinned" caveat of DESIGN.md section 2 -- and is the evidence behind keeping the
contraction-free form (changing the canonical form buys ~15 of 279 VALU instructions per block-wave).
    python tools/contraction_study.py   ->  profiles/r03_contraction_study.txt
"""
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ra-slam_amd"))
sys.path.insert(0, str(ROOT / "tests"))
from oracle_binding import load_oracle  # noqa: E402
from ratsdf import synthetic  # noqa: E402
from ratsdf._abi import Engine, Library  # noqa: E402

CASES = [("wall 80x60 2cm 3f", "wall", 3, "scannet", 0.125, 0.02), ("room 160x120 2cm 4f", "room", 4, "scannet", 0.25, 0.02),
         ("sphere 160x120 1cm 2f (tum)", "sphere", 2, "tum", 0.25, 0.01), ("room 640x480 5mm 6f", "room", 6, "scannet", 1.0, 0.005),
         ("room 1280x720 2mm 2f (l515)", "room", 2, "l515_720p", 1.0, 0.002)]


def build_fma_oracle():
    """The oracle's source compiled with every a*b+c fused (the contraction-free build is the canonical one)."""
    fma = ROOT / "oracle" / "build" / "libratsdf_oracle_fma.so"
    fma.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-mfma", "-fno-fast-math", "-pthread",
                    "-shared", "-o", str(fma), str(ROOT / "oracle" / "ratsdf_oracle.cpp")], check=True)
    return Library(fma, "ratsdf_oracle_")


def compare(A, B, case, threads=8):
    """Both builds on one case -> dict of the distances between the two maps."""
    name, scene, n, cam, scale, vs = case
    ea, eb = Engine(A, vs, 6 * vs, threads=threads), Engine(B, vs, 6 * vs, threads=threads)
    for i in range(n):
        f = synthetic.frame(scene, i, cam=cam, scale=scale, noise=True, holes=True)
        for e in (ea, eb):
            e.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
    _, ba = ea.dump_directory()
    _, bb = eb.dump_directory()
    ka = {(int(x), int(y), int(z)): int(i) for x, y, z, i in zip(ba["x"], ba["y"], ba["z"], ba["idx"])}
    kb = {(int(x), int(y), int(z)): int(i) for x, y, z, i in zip(bb["x"], bb["y"], bb["z"], bb["idx"])}
    common = sorted(set(ka) & set(kb))
    r = dict(name=name, blocks_a=len(ka), blocks_b=len(kb), only_one=len(set(ka) ^ set(kb)), common=len(common),
             same_idx=sum(1 for k in common if ka[k] == kb[k]), dt=0.0, dp=0.0, nw=0, nrgb=0, nvox=0, nbig=0)
    for lo in range(0, len(common), 4096):
        ks = common[lo:lo + 4096]
        ta, ca, pa = ea.dump_voxels(np.array([ka[k] for k in ks], dtype=np.int32))
        tb, cb, pb = eb.dump_voxels(np.array([kb[k] for k in ks], dtype=np.int32))
        r["dt"] = max(r["dt"], float(np.max(np.abs(ta - tb))))
        r["nbig"] += int((np.abs(ta - tb) > 1e-4).sum())
        r["dp"] = max(r["dp"], float(np.max(np.abs(pa - pb))))
        r["nw"] += int((ca["weight"] != cb["weight"]).sum())
        r["nrgb"] += int(((ca["r"] != cb["r"]) | (ca["g"] != cb["g"]) | (ca["b"] != cb["b"])).sum())
        r["nvox"] += ta.size
    ea.close()
    eb.close()
    return r


def main():
    A, B = load_oracle(), build_fma_oracle()
    out = []
    for case in CASES:
        r = compare(A, B, case)
        line = (f"{r['name']:30s} blocks {r['blocks_a']:6d} / {r['blocks_b']:6d}  in one map only {r['only_one']:4d}  "
                f"same pool index {r['same_idx']:6d}/{r['common']:6d}  voxels {r['nvox']:9d}: max|dtsdf| {r['dt']:.3g} "
                f"(> 1e-4 in {r['nbig']} voxels)  max|dprob| {r['dp']:.3g}  weight differs {r['nw']}  rgb differs {r['nrgb']}")
        print(line, flush=True)
        out.append(line)
    (ROOT / "profiles" / "r03_contraction_study.txt").write_text(__doc__ + "\n" + "\n".join(out) + "\n")


if __name__ == "__main__":
    main()
