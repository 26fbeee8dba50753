#!/usr/bin/env python3
"""Config 4 (two-phase 8192 x 2048) through the tile kernel with several nodes per thread (tuning cg_big = shape,
cg_big_xcd = tiles per XCD group), each candidate timed BETWEEN two timings of the default 16 x 32 tile kernel on the
same solver in the same process (box drift shows as the spread of the baseline column).
usage: r04_cg_big_sweep.py [shape,shape,...] [xcd,xcd,...] [steps]"""
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

shapes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4,5,6,7,8").split(",")]
xcds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2").split(",")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lib = pylbm.Lib()
lib.set_device(0)
dev = torch.device("cuda", 0)
w = bench.Secondary(lib, dev, "cg")
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
rows = []
for s in shapes:
    for x in xcds:
        lib.set_tuning(b"cg_big", 0)
        b0 = rate()
        lib.set_tuning(b"cg_big", s)
        lib.set_tuning(b"cg_big_xcd", x)
        v = rate()
        form = int(lib.raw.lbm_cg_last_inner_form())
        lib.set_tuning(b"cg_big", 0)
        b1 = rate()
        rows.append({"cg_big": s, "cg_big_xcd": x, "mlups": v, "form": form, "tile16x32_before": b0, "tile16x32_after": b1})
        print(json.dumps(rows[-1]), flush=True)
w.close()
