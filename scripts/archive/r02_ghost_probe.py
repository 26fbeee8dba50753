#!/usr/bin/env python3
"""Does the slab geometry (D ghost rows per side) by itself slow the window kernel down?  Times the
5-step launch over all owned rows on ghost = 0 / 5 lattices and over a few plane paddings (timing only:
the ghost rows hold zeros)."""
import ctypes as ct, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lattice-boltzmann-method_amd"))
import torch
import pylbm
from pylbm import _ptr
lib = pylbm.Lib(); dev = torch.device("cuda:0")
R = C = 8192; D = 5
prm = pylbm.BgkParams(1.2, 0)

def run(ghost, pad, r0=0, r1=R, n=60):
    rows = R + 2 * ghost
    plane = rows * C + pad
    g = pylbm.Geom(R, C, ghost, plane)
    bc = pylbm.Bc(row_lo=pylbm.EDGE_HALO, row_hi=pylbm.EDGE_HALO) if ghost else pylbm.Bc()
    a = torch.full((9 * plane,), 1.0 / 9, dtype=torch.float64, device=dev)
    b = torch.full((9 * plane,), 1.0 / 9, dtype=torch.float64, device=dev)
    def go(k):
        nonlocal a, b
        for _ in range(k):
            lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), D, r0, r1, None)
            a, b = b, a
    go(150); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); go(n); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    ts.sort()
    ms = ts[2] * 1e3
    print(f"ghost {ghost} pad {pad:7d} rows [{r0},{r1}): {ms:.4f} ms per launch = {(r1 - r0) * C * D / ms / 1e3:.0f} MLUPS", flush=True)
    del a, b

for rep in range(2):
    run(0, 8704)
    run(5, 8704)
for pad in (0, 1088, 4352, 8704 + 512, 17408, 8192 * 10 + 8704, 8192 * 5, 8192 * 10):
    run(0, pad)
run(1, 8704)
run(2, 8704)
run(5, 8704)
run(0, 8704)
