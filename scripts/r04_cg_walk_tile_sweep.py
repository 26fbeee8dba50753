#!/usr/bin/env python3
"""Config 4 (8192 x 2048): the walking tile (tuning cg_big = 10) against the shipped tile kernel (cg_big = 2), alternating in
one process, for several chunk heights ("cg_walk_rows") and XCD groupings ("cg_walk_tile_xcd").
usage: r04_cg_walk_tile_sweep.py [rows,rows,...] [xcd,xcd,...] [steps]"""
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

rows_list = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,128,256").split(",")]
xcds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2").split(",")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
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
for r in rows_list:
    for x in xcds:
        lib.set_tuning(b"cg_big", -1)
        b0 = rate()
        lib.set_tuning(b"cg_big", 10)
        lib.set_tuning(b"cg_walk_rows", r)
        lib.set_tuning(b"cg_walk_tile_xcd", x)
        v = rate()
        form = int(lib.raw.lbm_cg_last_inner_form())
        lib.set_tuning(b"cg_big", -1)
        b1 = rate()
        print(json.dumps({"cg_walk_rows": r, "cg_walk_tile_xcd": x, "walking_tile": v, "form": form, "tile_before": b0, "tile_after": b1}), flush=True)
w.close()
