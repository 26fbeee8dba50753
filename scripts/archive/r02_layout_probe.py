#!/usr/bin/env python3
"""Which part of the memory layout moves the window kernel's time on THIS box: the plane stride, the
offset between the two lattices, or the base address?  One arena, lattices carved at chosen offsets."""
import ctypes as ct, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lattice-boltzmann-method_amd"))
import torch
import pylbm
from pylbm import _ptr
lib = pylbm.Lib(); dev = torch.device("cuda:0")
R = C = 8192; D = 5
prm = pylbm.BgkParams(1.2, 0)
MAXPLANE = R * C + (1 << 20)
arena = torch.full((2 * 9 * MAXPLANE + (64 << 20),), 1.0 / 9, dtype=torch.float64, device=dev)
print("arena base %x" % arena.data_ptr(), flush=True)

def run(pad, a_off, gap, n=40, tag=""):
    """a at a_off doubles into the arena, b `gap` doubles after the end of a"""
    plane = R * C + pad
    g = pylbm.Geom(R, C, 0, plane)
    bc = pylbm.Bc()
    a = arena[a_off:a_off + 9 * plane]
    b0 = a_off + 9 * plane + gap
    b = arena[b0:b0 + 9 * plane]
    def go(k):
        nonlocal a, b
        for _ in range(k):
            lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), D, 0, R, None)
            a, b = b, a
    go(60); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); go(n); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    ts.sort()
    ms = ts[1] * 1e3
    print(f"pad {pad:7d} a_off {a_off:9d} gap {gap:9d}: {ms:.4f} ms = {R * C * D / ms / 1e3:.0f} MLUPS {tag}", flush=True)

K = 1024 // 8   # doubles per KiB
run(8704, 0, 0, tag="(reference)")
run(8704, 0, 0, tag="(repeat)")
for gap in (8 * K, 64 * K, 256 * K, 1024 * K, 2048 * K + 64 * K, 16384 * K, 4 * K, 512):
    run(8704, 0, gap)
for a_off in (512, 8 * K, 64 * K, 1024 * K, 2048 * K):
    run(8704, a_off, 0)
for pad in (0, 512, 1088, 2176, 4352, 8704, 13056, 17408, 34816, 40960, 65536, 81920, 131072, 262144 + 1088):
    run(pad, 0, 0)
run(8704, 0, 0, tag="(reference again)")
