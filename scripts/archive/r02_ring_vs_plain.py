#!/usr/bin/env python3
"""Same process, same lattices (slab geometry with 5 ghost rows): the 5-step launch over all owned rows in ONE
kernel vs the ring's launch-step (edge rows + exchange chain beside the interior).  Separates what the slab
geometry costs from what the concurrent exchange chain costs."""
import ctypes as ct, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lattice-boltzmann-method_amd"))
import torch
import pylbm
from pylbm import _ptr
import bench

class A: pass
a = A(); a.rows = a.cols = 8192; a.omega = 1.2; a.xn = 5; a.edge_rows = 32; a.plane_pad = None
lib = pylbm.Lib(); dev = torch.device("cuda:0")
box = bench.Box(lib, a, 0, 1, dev, with_ring=True)
box.load(bench.taylor_green(lib, 8192, 8192, 0, 8192, dev))
D = 5
bc = pylbm.Bc(row_lo=pylbm.EDGE_HALO, row_hi=pylbm.EDGE_HALO)

def plain(n):
    for _ in range(n):
        src, dst = box.lat[box.cur], box.lat[box.cur ^ 1]
        lib.bgk_stream_collide_xn(_ptr(dst), _ptr(src), ct.byref(box.geom), ct.byref(bc), ct.byref(box.prm), D, 0, 8192, box.stream())
        box.cur ^= 1

def split_no_exchange(n):   # edge rows (one launch) + interior on the same stream, no exchange
    for _ in range(n):
        src, dst = box.lat[box.cur], box.lat[box.cur ^ 1]
        lib.bgk_stream_collide_xn2(_ptr(dst), _ptr(src), ct.byref(box.geom), ct.byref(bc), ct.byref(box.prm), D, 0, 32, 8192 - 32, box.stream())
        lib.bgk_stream_collide_xn(_ptr(dst), _ptr(src), ct.byref(box.geom), ct.byref(bc), ct.byref(box.prm), D, 32, 8192 - 32, box.stream())
        box.cur ^= 1

def ring(n):
    for _ in range(n):
        box.launch(D)

def timeit(fn, n=60):
    fn(40); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(n); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n)
    ts.sort(); return ts[2] * 1e3

for rep in range(2):
    for name, fn in (("one launch over all rows", plain), ("edge launch + interior launch, no exchange", split_no_exchange), ("ring launch-step (self exchange)", ring)):
        ms = timeit(fn)
        print(f"{name:50s} {ms:.4f} ms per launch = {8192 * 8192 * D / ms / 1e3:.0f} MLUPS", flush=True)
    # (the dissection behind profiles/r02_ring_dissect.txt -- the step with its pack / unpack kernels and / or the RCCL call
    #  taken out -- used a diagnostic switch in capi_ring.hip that was removed again: it produced wrong results by design)
