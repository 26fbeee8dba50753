#!/usr/bin/env python3
"""EXPERIMENTS build: the two-columns-per-lane window (tuning sw_cols2 = 1) against the shipped window: bit-identity on small
periodic lattices and slabs, then ms per 5-step launch at 8192^2, alternating.  Run with LBM_HIP_LIB=.../lib_exp/liblbm_hip.so"""
import ctypes as ct
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lattice-boltzmann-method_amd"))
import torch  # noqa: E402
import pylbm  # noqa: E402
from pylbm import _ptr  # noqa: E402

lib = pylbm.Lib()
lib.set_device(0)
dev = torch.device("cuda", 0)
bc = pylbm.Bc.periodic()
w = torch.tensor([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=torch.float64, device=dev)
prm = pylbm.BgkParams(omega=1.2)


def lattice(R, C, ghost=0):
    plane = (R + 2 * ghost) * C
    torch.manual_seed(R * 7 + C)
    return (w.view(9, 1) * (1.0 + 0.05 * torch.rand((9, plane), dtype=torch.float64, device=dev))).contiguous(), plane


for R, C, n in ((96, 256, 5), (200, 1000, 5), (333, 130, 5), (128, 4096, 6), (64, 64, 5)):
    a, plane = lattice(R, C)
    g = pylbm.Geom(R, C, 0, plane, 0)
    out = []
    for v in (0, 1):
        lib.set_tuning(b"sw_cols2", v)
        b = torch.zeros_like(a)
        lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), n, 0, R, None)
        torch.cuda.synchronize()
        out.append(b)
    print("identical", R, C, n, bool(torch.equal(out[0], out[1])), float((out[0] - out[1]).abs().max()), flush=True)
R = C = 8192
a, plane = lattice(R, C)
b = torch.empty_like(a)
g = pylbm.Geom(R, C, 0, plane, 0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), 5, 0, R, None)
    torch.cuda.synchronize()
for rep in range(3):
    for v, n in ((0, 5), (1, 5), (1, 6)):
        lib.set_tuning(b"sw_cols2", v)
        for _ in range(3):
            lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), n, 0, R, None)
            a, b = b, a
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), n, 0, R, None)
            a, b = b, a
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 20
        print("sw_cols2", v, "steps", n, "ms per launch", round(t * 1e3, 4), "MLUPS", round(R * C * n / t / 1e6), flush=True)
