#!/usr/bin/env python3
"""The C call behind TSDFGrid::GatherValid on the bench map (41 MB of records), without the binding's hand-over:
tools/gather_probe.py   (GPU box; RATSDF_LIB selects the library)"""
import sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "ra-slam_amd"))
import numpy as np, ratsdf
from ratsdf import synthetic
gpu = ratsdf.TSDFGrid(0.005, 0.03)
for i in range(45):
    f = synthetic.frame("room", i, noise=True, holes=True)
    gpu.integrate(f["rgb"], f["depth"], f["ht"], f["lt"], 4.0, f["intrinsics"], f["pose"])
fn = gpu.lib.fn["gather_valid"]; free = gpu.lib.fn["free_buffer"]
for rep in range(4):
    p, n = C.c_void_p(), C.c_size_t()
    t0 = time.perf_counter(); st = fn(gpu._h, C.byref(p), C.byref(n)); t1 = time.perf_counter()
    free(p); t2 = time.perf_counter()
    print(f"gather_valid C call {1e3*(t1-t0):.2f} ms for {n.value} records ({n.value*16/1e6:.1f} MB); free {1e3*(t2-t1):.2f} ms")
