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
    assert len(k) > 20
    for name, res in k.items():
        assert res.get("ScratchSize", 0) == 0, f"{name} uses scratch memory: {res}"
    # k_integrate<2, false> (the default) and <2, true> (with the front-tail path, RATSDF_FRONT_TAIL=1)
    integrate2 = [v for n, v in k.items() if "k_integrateILi2E" in n]
    assert len(integrate2) == 2 and all(v["VGPRs"] <= 64 and v["Occupancy"] == 8 for v in integrate2)
    # k_front<false> (the default) and k_front<true> (serial role at the launch's tail, RATSDF_FRONT_TAIL=1)
    front = [v for n, v in k.items() if "7k_frontILb" in n]
    assert len(front) == 2 and all(f["Occupancy"] == 8 and f["LDS"] <= 20 * 1024 for f in front)
