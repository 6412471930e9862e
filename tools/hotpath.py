#!/usr/bin/env python3
"""Instruction mix of the voxel-update loop body of k_integrate<2> / k_integrate_g<2> in build/engine.s
(make the listing with: hipcc <FLAGS> -S --cuda-device-only -o build/engine.s ratsdf_engine.hip)."""
import re
import sys

A = sys.argv[1] if len(sys.argv) > 1 else "ra-slam_amd/csrc/build/engine.s"
lines = open(A).read().split("\n")


def extract(prefix):
    s = [i for i, l in enumerate(lines) if l.startswith(prefix) and ": ; @" in l][0]
    e = next(i for i in range(s, len(lines)) if ".amdhsa_next_free_sgpr" in lines[i])
    return lines[s:e + 1]


for name, sym in (("k_integrate<2>", "_ZN6ratsdf11k_integrateILi2ELb0EEEv"), ("k_integrate_g<2>", "_ZN6ratsdf13k_integrate_gILi2ELb0EEEv")):
    f = extract(sym)
    rpi = next(i for i, l in enumerate(f) if "v_cvt_rpi_i32_f32_e64" in l)
    bar = next(i for i in range(rpi, len(f)) if "s_barrier" in f[i])
    st = max(i for i in range(rpi) if ("s_barrier" in f[i] or "Loop Header" in f[i]))
    hot = f[st:bar]
    c = lambda pat: sum(1 for l in hot if re.match(r"\s*" + pat, l))
    div = 0
    inside = False
    for l in hot:
        if "v_div_scale_f32" in l:
            inside = True
        if inside and l.startswith(".LBB"):
            inside = False
        if inside and re.match(r"\s*v_", l):
            div += 1
    res = [l.strip() for l in f if "next_free" in l or "private_segment_fixed_size" in l]
    print(f"{name:18s} loop body {len(hot)} lines: VALU {c('v_')} (rare IEEE-division block {div}) pk {c('v_pk_')} "
          f"readlane {c('v_readlane')} writelane {c('v_writelane')} s_nop {c('s_nop')} SALU {c('s_')} smem {c('s_load')} "
          f"vmem {c('global_')} lds {c('ds_')} | {' '.join(res)}")
    if len(sys.argv) > 2:
        open(sys.argv[2] + "_" + name.replace("<", "").replace(">", "") + ".s", "w").write("\n".join(hot))
