#!/usr/bin/env python3
"""What bounds a sliding-window launch when its arithmetic does not?  Time per LAUNCH of lbm_bgk_stream_collide_xn
(n = 2, 3, 5) and lbm_kbc_stream_collide_xn (n = 2, 3) on periodic lattices of several shapes, dense rows and padded rows:
a launch reads and writes every population once whatever n is, so ps per node per launch = the memory-side floor when it
does not move with n.   usage: r04_window_floor_probe.py [RxC,RxC,...]"""
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

shapes = [tuple(int(v) for v in s.split("x")) for s in (sys.argv[1] if len(sys.argv) > 1 else "4096x4096,2048x8192,8192x2048,8192x8192").split(",")]
lib = pylbm.Lib()
lib.set_device(0)
dev = torch.device("cuda", 0)
bc = pylbm.Bc.periodic()
w = torch.tensor([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=torch.float64, device=dev)


def run(fn, prm, n, a, b, g, R, reps=12):
    for _ in range(2):
        fn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), n, 0, R, None)
        a, b = b, a
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), n, 0, R, None)
        a, b = b, a
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for R, C in shapes:
    for pad in (0, 64):
        P = C + pad
        plane = R * P
        g = pylbm.Geom(R, C, 0, plane, P if pad else 0)
        a = (w.view(9, 1) * (1.0 + 0.01 * torch.rand((9, plane), dtype=torch.float64, device=dev))).contiguous()
        b = torch.empty_like(a)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:  # warm the clocks
            run(lib.bgk_stream_collide_xn, pylbm.BgkParams(omega=1.2), 3, a, b, g, R, reps=4)
        row = {"R": R, "C": C, "row_pitch": P}
        for name, fn, prm, ns in (("bgk", lib.bgk_stream_collide_xn, pylbm.BgkParams(omega=1.2), tuple(int(v) for v in os.environ.get("PROBE_BGK_N", "2,3,5").split(","))),
                                  ("kbc", lib.kbc_stream_collide_xn, pylbm.KbcParams(s2=1.6), tuple(int(v) for v in os.environ.get("PROBE_KBC_N", "2,3").split(",")))):
            for n in ns:
                t = run(fn, prm, n, a, b, g, R)
                row["%s_x%d" % (name, n)] = {"ms": round(t * 1e3, 4), "ps_per_node": round(t / (R * C) * 1e12, 2),
                                             "alg_TBs": round(144.0 * R * C / t / 1e12, 3), "MLUPS": round(R * C * n / t / 1e6)}
        print(row, flush=True)
        del a, b
