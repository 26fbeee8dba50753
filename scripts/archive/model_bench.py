#!/usr/bin/env python3
"""Secondary measurements (not the contract bench): MLUPS of the KBC, colour-gradient and
IBM-cylinder steps at BASELINE config sizes on one GPU, through the solver contexts of the
C ABI.  Prints one JSON line per model."""
import ctypes as ct
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lattice-boltzmann-method_amd"))
import numpy as np
import torch

import pylbm
from pylbm import _ptr

lib = pylbm.Lib()
dev = torch.device("cuda:0")


def timed(step, n, warm=5, min_sec=0.4):
    """seconds per step; the timed region is stretched to >= min_sec so that the clocks have ramped
    (the headline bench gains 13 % between a 10 ms and a 400 ms region, profiles/r01_bench_steps_sweep.log)"""
    step(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    reps = int(min_sec / max(dt, 1e-6))
    if reps > 1:
        step(n * reps // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(n * reps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    return dt / n


def report(name, R, C, dt, bytes_per_lup, extra=None):
    mlups = R * C / dt / 1e6
    out = dict(model=name, rows=R, cols=C, ms_per_step=round(dt * 1e3, 4), MLUPS=round(mlups, 1),
               GBs=round(mlups * 1e6 * bytes_per_lup / 1e9, 1), bytes_per_lup=bytes_per_lup,
               frac_of_8TBs=round(mlups * 1e6 * bytes_per_lup / 8e12, 4))
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)


def smooth_f(R, C):
    r = torch.arange(R, device=dev, dtype=torch.float64).view(-1, 1)
    c = torch.arange(C, device=dev, dtype=torch.float64).view(1, -1)
    u = torch.empty((2, R, C), dtype=torch.float64, device=dev)
    u[0] = 0.03 * torch.sin(2 * np.pi * r / R) * torch.cos(2 * np.pi * c / C)
    u[1] = -0.03 * torch.cos(2 * np.pi * r / R) * torch.sin(2 * np.pi * c / C)
    rho = torch.ones((R, C), dtype=torch.float64, device=dev)
    f = torch.empty((9, R, C), dtype=torch.float64, device=dev)
    lib.equilibrium(_ptr(f), _ptr(u), _ptr(rho), R, C, None)
    torch.cuda.synchronize()
    return f


def bench_single(model, name, R, C, params, n=40, bc=None):
    sv = pylbm.Solver(lib, model, R, C, params, bc=bc)
    f = smooth_f(R, C)
    lib.solver_set_f_soa_dev(sv.h, _ptr(f))
    del f
    dt = timed(lambda k: sv.step(k), n)
    report(name, R, C, dt, 144)
    sv.close()


def bench_cg(R, C, n=20):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    prm = pylbm.cg_params()
    # init on the host like the driver (init_rho_cosine), populations via lbm_cg_equilibrium
    rr = np.arange(R).reshape(-1, 1)
    s = R / 2.0 - 0.1 * C * np.cos(2.0 * 3.141592 * np.arange(C) / C).reshape(1, -1)
    rho_r = 3.0 * (rr < s)
    rho_b = 1.0 * (rr >= s)
    u = np.zeros((R, C, 2))
    d_rr, d_rb = torch.from_numpy(rho_r.astype(np.float64)).to(dev), torch.from_numpy(rho_b.astype(np.float64)).to(dev)
    d_u = torch.zeros((2, R, C), dtype=torch.float64, device=dev)
    f_r = torch.empty((9, R, C), dtype=torch.float64, device=dev)
    f_b = torch.empty((9, R, C), dtype=torch.float64, device=dev)
    lib.cg_equilibrium(_ptr(f_r), _ptr(d_rr), _ptr(d_u), ct.byref(prm.red), R, C, ct.c_longlong(0), None)
    lib.cg_equilibrium(_ptr(f_b), _ptr(d_rb), _ptr(d_u), ct.byref(prm.blue), R, C, ct.c_longlong(0), None)
    torch.cuda.synchronize()
    sv = pylbm.CgSolver(lib, R, C, prm)
    sv.set_state(np.moveaxis(f_r.cpu().numpy(), 0, -1), np.moveaxis(f_b.cpu().numpy(), 0, -1), rho_r, rho_b, u)
    del f_r, f_b
    for strip in [x for x in os.environ.get("LBM_CG_STRIP", "").split(",") if x]:
        for rows in os.environ.get("LBM_CG_ROWS", "64").split(","):
            lib.set_tuning(b"cg_fused", 1)
            lib.set_tuning(b"cg_strip", int(strip))
            lib.set_tuning(b"cg_rows", int(rows))
            dt = timed(lambda k: sv.step(k), n, warm=3)
            report("colour-gradient MRT (one launch per step, column strips, %s wave(s)/block, %s rows/chunk)" % (strip, rows), R, C, dt, 288)
    lib.set_tuning(b"cg_rows", -1)
    lib.set_tuning(b"cg_strip", 0)
    for s2 in [x for x in os.environ.get("LBM_CG_STRIP2", "").split(",") if x]:
        for rows in os.environ.get("LBM_CG_ROWS2", "64").split(","):
            lib.set_tuning(b"cg_fused", 1)
            lib.set_tuning(b"cg_strip2", int(s2))
            lib.set_tuning(b"cg_rows2", int(rows))
            for xo in os.environ.get("LBM_CG_STRIP_XCD", "1").split(","):
                lib.set_tuning(b"cg_strip_xcd", int(xo))
                dt = timed(lambda k: sv.step(k), n, warm=3)
                report("colour-gradient MRT (inner rectangle: register-ring strips, %s wave(s)/block, %s rows/chunk, xcd order %s; frame: tiles)" % (s2, rows, xo), R, C, dt, 288)
            lib.set_tuning(b"cg_strip_xcd", -1)
    lib.set_tuning(b"cg_strip2", 0)
    lib.set_tuning(b"cg_rows2", -1)
    tiles = os.environ.get("LBM_CG_TILES", "4").split(",")
    for tile in tiles:
        for xcd in os.environ.get("LBM_CG_XCD", "2").split(","):
            lib.set_tuning(b"cg_fused", 1)
            lib.set_tuning(b"cg_tile", int(tile))
            lib.set_tuning(b"cg_xcd", int(xcd))
            for mg in os.environ.get("LBM_CG_MERGE", "-1").split(","):
                lib.set_tuning(b"cg_merge", int(mg))
                dt = timed(lambda k: sv.step(k), n, warm=3)
                report("colour-gradient MRT (fused, one launch per step, tile %s, xcd order %s, merge %s)" % (tile, xcd, mg), R, C, dt, 288)
            lib.set_tuning(b"cg_merge", -1)
    lib.set_tuning(b"cg_xcd", -1)
    lib.set_tuning(b"cg_fused", 0)
    dt = timed(lambda k: sv.step(k), n, warm=3)
    report("colour-gradient MRT (two-pass, reference operation order)", R, C, dt, 496)
    lib.set_tuning(b"cg_fused", -1)
    lib.set_tuning(b"cg_strip", -1)
    lib.set_tuning(b"cg_strip2", -1)
    lib.set_tuning(b"cg_tile", -1)
    sv.close()


def bench_cylinder(X, Y, n=30):
    omega, u_in = 1.0 / 0.55, 0.04
    m = int(round(np.pi * 300))
    t = 2 * np.pi * np.arange(m) / m
    x, y = X / 4.0 + 150 * np.cos(t), Y / 2.0 + 150 * np.sin(t)
    bc = pylbm.Bc.periodic()
    bc.row_lo = bc.row_hi = pylbm.EDGE_ABB_VELOCITY
    bc.col_lo = bc.col_hi = pylbm.EDGE_SPECULAR
    bc.uw_r = u_in
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 0, 1), bc=bc)
    ib = pylbm.Ibm(lib, x, y, X, Y)
    sv.attach_ibm(ib)
    u = torch.zeros((2, X, Y), dtype=torch.float64, device=dev); u[0] = u_in
    rho = torch.ones((X, Y), dtype=torch.float64, device=dev)
    f = torch.empty((9, X, Y), dtype=torch.float64, device=dev)
    lib.incomp_equilibrium(_ptr(f), _ptr(u), _ptr(rho), X, Y, None)
    lib.solver_set_f_soa_dev(sv.h, _ptr(f))
    del f, u, rho
    dt = timed(lambda k: sv.step(k), n)
    report("BGK + IBM cylinder (d=300, %d markers)" % m, X, Y, dt, 144, dict(note="5 steps per block: forced band around the ROI on a shrinking trapezoid (single steps), rows farther away through the 5-step window"))
    sv.close(); ib.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["bgk", "walls", "kbc", "cg", "ibm"]
    if "bgk" in which:
        bench_single(pylbm.MODEL_BGK, "BGK (solver context)", 8192, 8192, pylbm.BgkParams(1.2, 0))
    if "walls" in which:
        for name, rows, cols in (("channel: bounce-back columns, periodic rows", False, True),
                                 ("closed box: bounce-back rows and columns", True, True)):
            bc = pylbm.Bc.periodic()
            if cols:
                bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
            if rows:
                bc.row_lo = bc.row_hi = pylbm.EDGE_BOUNCE_BACK
            for depth in os.environ.get("LBM_WALL_DEPTH", "5,1").split(","):
                for split in os.environ.get("LBM_WALL_SPLIT", "-1").split(","):
                    lib.set_tuning(b"solver_depth", int(depth))
                    lib.set_tuning(b"solver_depth_walls", int(depth))
                    lib.set_tuning(b"sw_split", int(split))
                    bench_single(pylbm.MODEL_BGK, "BGK %s, %s step(s) per launch, sw_split %s" % (name, depth, split), 8192, 8192,
                                 pylbm.BgkParams(1.2, 0), bc=bc)
            lib.set_tuning(b"sw_split", -1)
            lib.set_tuning(b"solver_depth", -1)
            lib.set_tuning(b"solver_depth_walls", -1)
    if "pressure" in which:   # Poiseuille channel at config-2 size: pressure-periodic rows, bounce-back columns
        from math import sqrt
        omega = 1.0 / (sqrt(3.0 / 16.0) + 0.5)
        H = W = int(os.environ.get("LBM_PRESSURE_SIZE", "8192"))
        grad = 8.0 * (1.0 / 3.0) * (1.0 / omega - 0.5) * 0.1 / (W * W)
        bc = pylbm.Bc(col_lo=pylbm.EDGE_BOUNCE_BACK, col_hi=pylbm.EDGE_BOUNCE_BACK, pressure_rows=1,
                      rho_inlet=1.0 + 3.0 * (H - 1) * grad, rho_outlet=1.0)
        for depth in os.environ.get("LBM_PRESSURE_DEPTH", "5,3,1").split(","):
            lib.set_tuning(b"pressure_depth", int(depth))
            bench_single(pylbm.MODEL_BGK, "BGK Poiseuille channel (pressure rows + bounce-back columns, incompressible), %s step(s) per block" % depth,
                         H, W, pylbm.BgkParams(omega, 1), bc=bc)
        lib.set_tuning(b"pressure_depth", -1)
    if "kbc" in which:
        for depth in os.environ.get("LBM_KBC_DEPTH", "3").split(","):
            lib.set_tuning(b"kbc_depth", int(depth))
            for size in os.environ.get("LBM_KBC_SIZE", "4096").split(","):
                for rows in os.environ.get("LBM_SW_ROWS", "-1").split(","):
                    lib.set_tuning(b"sw_rows", int(rows))
                    bench_single(pylbm.MODEL_KBC, "KBC (reassociated collision, %s step(s) per launch%s)" % (
                        depth, "" if int(rows) < 0 else ", %s rows per wave" % rows), int(size), int(size),
                        pylbm.KbcParams(1.0 / (0.5 + 3 * 1.70766666e-4)))
            lib.set_tuning(b"sw_rows", -1)
        lib.set_tuning(b"kbc_depth", -1)
        lib.set_tuning(b"kbc_fast", 0)
        bench_single(pylbm.MODEL_KBC, "KBC (reference operation order)", 4096, 4096,
                     pylbm.KbcParams(1.0 / (0.5 + 3 * 1.70766666e-4)))
        lib.set_tuning(b"kbc_fast", -1)
    if "cg" in which:
        bench_cg(8192, 2048)
    if "ibm" in which:
        # 2048: one 8-GPU slab of config 5; 16384: config 5 whole on one GPU
        for rows in os.environ.get("LBM_IBM_SIZES", "2048,4096,16384").split(","):
            bench_cylinder(int(rows), 4096)
