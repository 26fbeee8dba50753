"""Time of one lbm_ibm_step launch (the one-workgroup forcing of config 5: 942 markers, cylinder d = 300)
with the plain and the batched spread ("ibm_step_opt" 0 / 1), and that both write the same bits."""
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
for X, Y, radius in ((1200, 4096, 150.0), (700, 2048, 80.0), (1400, 4096, 230.0)):
    n = int(round(2 * np.pi * radius))
    t = 2 * np.pi * np.arange(n) / n
    x, y = X / 2 + 0.3 + radius * np.cos(t), Y / 2 + 0.6 + radius * np.sin(t)
    ib = pylbm.Ibm(lib, x, y, X, Y)
    g = pylbm.Geom(X, Y, 0)
    rr, cc = torch.meshgrid(torch.arange(X, dtype=torch.float64, device=dev), torch.arange(Y, dtype=torch.float64, device=dev), indexing="ij")
    u = torch.stack([0.05 + 0.01 * torch.sin(rr / 7.0), 0.02 * torch.cos(cc / 5.0)]).contiguous()
    rho = (1 + 0.02 * torch.sin((rr + cc) / 9.0)).contiguous()
    out = {}
    for opt in (0, 1, 2):
        lib.set_tuning(b"ibm_step_opt", opt & 1)
        lib.set_tuning(b"ibm_step_chain", opt >> 1)
        lib.set_tuning(b"ibm_step_split", 0)
        p = torch.zeros((9, X, Y), dtype=torch.float64, device=dev)
        lib.ibm_step(ib.h, _ptr(p), ct.byref(g), _ptr(u), _ptr(rho), ct.c_double(1.3), ct.c_double(1.0), ct.c_double(3.0), None)
        torch.cuda.synchronize()
        out[opt] = p.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(20):
            lib.ibm_step(ib.h, _ptr(p), ct.byref(g), _ptr(u), _ptr(rho), ct.c_double(1.3), ct.c_double(1.0), ct.c_double(3.0), None)
        e0.record()
        for _ in range(300):
            lib.ibm_step(ib.h, _ptr(p), ct.byref(g), _ptr(u), _ptr(rho), ct.c_double(1.3), ct.c_double(1.0), ct.c_double(3.0), None)
        e1.record()
        torch.cuda.synchronize()
        print(f"{X}x{Y} radius {radius}: {n} markers, variant {opt} (0 plain one-workgroup, 1 batched spread, 2 launch chain): {e0.elapsed_time(e1) / 300 * 1e3:.1f} us per launch", flush=True)
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2]) and float(out[0].abs().max()) > 0
    lib.set_tuning(b"ibm_step_opt", 1)
    lib.set_tuning(b"ibm_step_chain", 0)
    lib.set_tuning(b"ibm_step_split", 1)
    ib.close()
print("same bits")
