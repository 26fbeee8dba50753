"""lbm_ibm_step (one workgroup) per number of forcing iterations: intercept = gather + source, slope = one iteration"""
import ctypes as ct
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lattice-boltzmann-method_amd"))
import pylbm  # noqa: E402

lib = pylbm.Lib()
dev = torch.device("cuda", 0)
_ptr = lambda t: ct.c_void_p(t.data_ptr())
X, Y, radius = 1200, 4096, 150.0
n = int(round(2 * np.pi * radius))
t = 2 * np.pi * np.arange(n) / n
x, y = X / 2 + 0.3 + radius * np.cos(t), Y / 2 + 0.6 + radius * np.sin(t)
g = pylbm.Geom(X, Y, 0)
rr, cc = torch.meshgrid(torch.arange(X, dtype=torch.float64, device=dev), torch.arange(Y, dtype=torch.float64, device=dev), indexing="ij")
u = torch.stack([0.05 + 0.01 * torch.sin(rr / 7.0), 0.02 * torch.cos(cc / 5.0)]).contiguous()
rho = (1 + 0.02 * torch.sin((rr + cc) / 9.0)).contiguous()
for m_max in (2, 3, 5, 9):
    ib = pylbm.Ibm(lib, x, y, X, Y, m_max=m_max)
    for opt in (0, 1, 3):
        lib.set_tuning(b"ibm_step_opt", opt & 1)
        lib.set_tuning(b"ibm_step_split", opt >> 1)
        p = torch.zeros((9, X, Y), dtype=torch.float64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20):
            lib.ibm_step(ib.h, _ptr(p), ct.byref(g), _ptr(u), _ptr(rho), ct.c_double(1.3), ct.c_double(1.0), ct.c_double(3.0), None)
        e0.record()
        for _ in range(300):
            lib.ibm_step(ib.h, _ptr(p), ct.byref(g), _ptr(u), _ptr(rho), ct.c_double(1.3), ct.c_double(1.0), ct.c_double(3.0), None)
        e1.record()
        torch.cuda.synchronize()
        print(f"{n} markers, {m_max - 1} iteration(s), variant {opt} (0 round 1, 1 lane-major tables, 3 = 1 + source as its own launch): {e0.elapsed_time(e1) / 300 * 1e3:.1f} us per launch", flush=True)
    ib.close()
lib.set_tuning(b"ibm_step_opt", 1)
lib.set_tuning(b"ibm_step_split", 1)
