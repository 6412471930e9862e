"""Compile-time resource guard for the gfx950 kernels (no GPU needed: hipcc cross-compiles).

Measured on the MI355X: 16 bytes of scratch per lane in k_integrate doubled its run time, and one VGPR
above 64 costs it a wave per SIMD.  The build must therefore stay spill-free, and the frame kernels
must keep the occupancy they were tuned for."""
import re
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "ra-slam_amd" / "csrc"


def resource_usage():
    r = subprocess.run(["make", "-C", str(CSRC), "resource-usage"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    kernels, cur = {}, None
    for line in r.stdout.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            continue
        m = re.search(r"(VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            kernels[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    return kernels


def test_kernels_do_not_spill_and_keep_their_occupancy():
    k = resource_usage()
    assert len(k) > 15
    for name, res in k.items():
        assert res.get("ScratchSize", 0) == 0, f"{name} uses scratch memory: {res}"
    # the product build carries ONE form of the frame kernels: k_integrate<2, false> / k_front<false> (the retired
    # variants -- 4 / 8 voxels per lane, the front-tail forms -- exist in the diagnostic build only)
    integrate = [v for n, v in k.items() if "k_integrateILi" in n]
    assert len(integrate) == 1 and integrate[0]["VGPRs"] <= 64 and integrate[0]["Occupancy"] == 8, integrate
    front = [v for n, v in k.items() if "7k_frontILb" in n]
    assert len(front) == 1 and front[0]["Occupancy"] == 8 and front[0]["LDS"] <= 20 * 1024, front
