#!/usr/bin/env python3
"""Config 4 (8192 x 2048), the shipped 16 x 64 tile kernel: tile rows swept alternately top-down / bottom-up ("cg_big_alt" = 1)
against every tile top-down (0), for several XCD patch orders, alternating in one process on one box.
usage: r04_cg_alt_ab.py [xcd,xcd,...] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lattice-boltzmann-method_amd"))
import torch  # noqa: E402
import pylbm  # noqa: E402
import bench  # noqa: E402

xcds = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "802").split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
lib = pylbm.Lib()
lib.set_device(0)
w = bench.Secondary(lib, torch.device("cuda", 0), "cg")
nodes = w.R * w.C


def rate(reps=3):
    w.step(steps)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        w.step(steps)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return round(nodes * steps / sorted(ts)[len(ts) // 2] / 1e6, 1)


t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    w.step(steps)
    torch.cuda.synchronize()
for x in xcds:
    lib.set_tuning(b"cg_big_xcd", x)
    row = {"cg_big_xcd": x, "alt0": [], "alt1": []}
    for _ in range(3):
        for alt in (0, 1):
            lib.set_tuning(b"cg_big_alt", alt)
            row[f"alt{alt}"].append(rate())
    lib.set_tuning(b"cg_big", 0)
    row["tile16x32"] = rate()
    lib.set_tuning(b"cg_big", -1)
    print(json.dumps(row), flush=True)
w.close()
