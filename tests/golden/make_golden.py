#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref, built by
`make -C oracle ref` from /root/reference, CPU libtorch).  Runs only in the build
container; the fixtures it writes are plain data (inputs + expected outputs) and are
committed.  Re-run:  python tests/golden/make_golden.py [--skip-dsf]

Sources of each fixture
  solver_units.npz  solver::{calc_rho,calc_u,calc_incomp_u,equilibrium,incomp_equilibrium,
                    collision,advect} + the BGK periodic loop, via ref_capi.cpp
  hpt_21x21.npz     test/horizontal_poiseuille_test.cpp  main(), unmodified (oracle/_ref/hpt)
  ddm_21x21.npz     test/decompose_domain.cpp            main(), unmodified (oracle/_ref/ddm)
  dsf_128.npz       test/ulbm_double_shear_flow.cpp      main(), unmodified (oracle/_ref/dsf)
  kbc_units.npz     ulbm::d2q9::kbc collide/advect via ref_capi.cpp
  diff5.npz         differential::x / ::y via ref_capi.cpp
  upo_units.npz     the loop of test/ulbm_poiseuille.cpp on the reference's own kbc class and
                    solver::incomp_equilibrium, sequenced by ref_capi.cpp (ref_upo_steps)
  ddl_512.npz       test/decompose_domain_loop.cpp main(), unmodified (oracle/_ref/ddl): 50000 steps, ~25 min
                    -- only with --ddl-dir pointing at a finished run's output files
  upo_128.npz       test/ulbm_poiseuille.cpp main(), unmodified (oracle/_ref/upo): 300000 steps, hours
                    of CPU time -- only with --upo-dir pointing at a finished run's output files
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyoracle import REF_DIR, Oracle, Ref, build_ref  # noqa: E402


def load_pt(path):
    return list(torch.jit.load(path).parameters())[0].detach().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB")


def random_state(rng, R, C, r):
    rho = 1 + 0.01 * rng.standard_normal((R, C))
    u = 0.05 * rng.standard_normal((R, C, 2))
    f = r.equilibrium(u, rho) * (1 + 0.01 * rng.standard_normal((R, C, 9)))
    return rho, u, f


def gen_solver_units(r, o):
    rng = np.random.default_rng(1)
    out = {}
    for tag, (R, C) in dict(a=(37, 23), b=(64, 48)).items():
        rho, u, f = random_state(rng, R, C, r)
        out[f"{tag}_rho_in"], out[f"{tag}_u_in"], out[f"{tag}_f_in"] = rho, u, f
        out[f"{tag}_calc_rho"] = r.calc_rho(f)
        out[f"{tag}_calc_u"] = r.calc_u(f, out[f"{tag}_calc_rho"])
        out[f"{tag}_calc_incomp_u"] = r.calc_incomp_u(f)
        out[f"{tag}_equilibrium"] = r.equilibrium(u, rho)
        out[f"{tag}_incomp_equilibrium"] = r.incomp_equilibrium(u, rho)
        out[f"{tag}_collision_w1.2"] = r.collision(f, out[f"{tag}_equilibrium"], 1.2)
        out[f"{tag}_advect"] = r.advect(f)
        for n in (1, 10, 100):
            for inc in (0, 1):
                ff, rr, uu = r.bgk_periodic_steps(f, 1.2, n, bool(inc))
                out[f"{tag}_bgk{n}_inc{inc}_f"] = ff
                out[f"{tag}_bgk{n}_inc{inc}_rho"] = rr
                out[f"{tag}_bgk{n}_inc{inc}_u"] = uu
    save("solver_units.npz", **out)


def gen_hpt():
    with tempfile.TemporaryDirectory() as d:
        log = subprocess.run([os.path.join(REF_DIR, "hpt")], cwd=d, capture_output=True,
                             text=True, check=True).stdout
        l2 = float(re.search(r"L2=([0-9.eE+-]+)", log).group(1))
        ux, uy = load_pt(f"{d}/hpt-ux.pt"), load_pt(f"{d}/hpt-uy.pt")
        rho = load_pt(f"{d}/hpt-ps.pt") * 3.0
        fs = load_pt(f"{d}/hpt-fs.pt")
    T = ux.shape[-1]
    steps = np.array([1, 2, 3, 10, 100, 1000, T - 1])
    # snapshot t holds f_adve BEFORE step t and the moments computed DURING step t-1
    save("hpt_21x21.npz", steps=steps, l2_printed=np.float64(l2), T=np.int64(T),
         fs=np.ascontiguousarray(fs[..., steps]), ux=np.ascontiguousarray(ux[..., steps]),
         uy=np.ascontiguousarray(uy[..., steps]), rho=np.ascontiguousarray(rho[..., steps]))


def _run_main(exe, prefix, steps_fn, out_name, rho_from_ps=True):
    """run an unmodified reference main that saves <prefix>-{fs,ux,uy,ps}.pt stacks over all T
    steps and keep a few time slices (snapshot t = f_adve entering step t, moments of step t-1)"""
    with tempfile.TemporaryDirectory() as d:
        log = subprocess.run([os.path.join(REF_DIR, exe)], cwd=d, capture_output=True, text=True,
                             check=True).stdout
        ux, uy = load_pt(f"{d}/{prefix}-ux.pt"), load_pt(f"{d}/{prefix}-uy.pt")
        rho = load_pt(f"{d}/{prefix}-ps.pt") * 3.0
        fs = load_pt(f"{d}/{prefix}-fs.pt")
    m = re.search(r"last t=(\d+)", log)
    T = ux.shape[-1]
    steps = steps_fn(T, int(m.group(1)) if m else None)
    save(out_name, steps=steps, T=np.int64(T), last_t=np.int64(int(m.group(1)) if m else -1),
         fs=np.ascontiguousarray(fs[..., steps]), ux=np.ascontiguousarray(ux[..., steps]),
         uy=np.ascontiguousarray(uy[..., steps]), rho=np.ascontiguousarray(rho[..., steps]))


def gen_sbt():  # test/specular_boundary_test.cpp, 51 x 51, T = 10000 (1.9 GB of snapshots in RAM)
    _run_main("sbt", "sbt", lambda T, last: np.array([1, 2, 10, 100, 1000, T - 1]), "sbt_51x51.npz")


def gen_gt():   # test/gravity_test.cpp, 21 x 21; stops by its own convergence rule ("last t=")
    _run_main("gt", "gt", lambda T, last: np.array([1, 2, 10, 100, 1000, last]), "gt_21x21.npz")


def gen_ddm():
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([os.path.join(REF_DIR, "ddm")], cwd=d, capture_output=True, check=True)
        out = {}
        steps = None
        for blk in "AB":
            fs = load_pt(f"{d}/{blk}-domain-decomp-hpt-fs.pt")
            steps = np.array([1, 2, 10, 100, fs.shape[-1] - 1])
            out[f"{blk}_fs"] = np.ascontiguousarray(fs[..., steps])
            for k in ("ux", "uy", "rho"):
                out[f"{blk}_{k}"] = np.ascontiguousarray(
                    load_pt(f"{d}/{blk}-domain-decomp-hpt-{k}.pt")[..., steps])
    save("ddm_21x21.npz", steps=steps, **out)


def gen_dsf():
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([os.path.join(REF_DIR, "dsf")], cwd=d, capture_output=True, check=True)
        idx = np.array([1, 2, 5, 10, 20, 50])  # snapshot index i <-> time step 10*i
        out = {k: np.ascontiguousarray(load_pt(f"{d}/ulbm-double-shear-flow-{k}.pt")[..., idx])
               for k in ("ux", "uy", "rho")}
    save("dsf_128.npz", snap_index=idx, snapshot_period=np.int64(10), **out)


def gen_kbc(r, o):
    rng = np.random.default_rng(7)
    R, C = 24, 20
    rho, u, f = random_state(rng, R, C, r)
    m0 = r.calc_rho(f)
    m1 = r.calc_u(f, m0)
    s2 = 1.0 / (0.5 + 3.0 * 1.70766666e-4)
    f1, m01, m11, coll = r.kbc_steps(f, m0, m1, s2, 1, want_coll=True)
    out = dict(f_in=f, m0_in=m0, m1_in=m1, s2=np.float64(s2), coll1=coll, f1=f1, m0_1=m01, m1_1=m11)
    # shear-layer start exactly as the driver does it (eval_equilibrium with ux2 = uy2 = 0)
    sm0, sm1 = o.kbc_shear_init(32, 32)
    out["shear_m0"], out["shear_m1"] = sm0, sm1
    for n in (0, 1, 5, 20):
        ff, a, b = r.kbc_steps(None, sm0, sm1, s2, n, init_from_moments=True)
        out[f"shear{n}_f"], out[f"shear{n}_m0"], out[f"shear{n}_m1"] = ff, a, b
    save("kbc_units.npz", **out)


def upo_params(H, W):
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    p_grad = 8.0 * nu * 0.05 / (W * W)
    return s2, 3.0 * (H - 1) * p_grad + 1.0, 1.0      # ulbm_poiseuille.cpp:70-83


def gen_upo_units(r):
    out = {}
    for tag, (H, W), steps in (("a", (24, 20), (1, 2, 10, 100)), ("b", (128, 128), (50,))):
        s2, rin, rout = upo_params(H, W)
        out[f"{tag}_shape"] = np.array([H, W])
        for n in steps:
            f, m0, m1 = r.upo_steps(H, W, s2, rin, rout, n)
            out[f"{tag}_{n}_f"], out[f"{tag}_{n}_m0"], out[f"{tag}_{n}_m1"] = f, m0, m1
    save("upo_units.npz", **out)


def gen_upo_main(d):
    """snapshots of the unmodified main (index i <-> state BEFORE iteration 100 i, :109-117)"""
    idx = np.array([1, 2, 5, 10, 50, 200, 1000, 2999])
    pre = "ulbm-poiseuillehpt-"            # file_prefix + "hpt-ux.pt" (:151-154)
    out = {k: np.ascontiguousarray(load_pt(os.path.join(d, f"{pre}{k}.pt"))[..., idx]) for k in ("ux", "uy", "rho")}
    save("upo_128.npz", snap_index=idx, snapshot_period=np.int64(100), **out)


def gen_ddl_main(d):
    """snapshots of the unmodified test/decompose_domain_loop.cpp main (oracle/_ref/ddl, L = 512,
    T = 50000, ~25 min): index i <-> state BEFORE iteration 50 i (:112-132), i.e. the moments
    computed in iteration 50 i - 1, with F already added to A's u on the force rows (:114).
    Full fields for i = 1, 2, 10; every 4th row / column for i = 100, 999 (size)."""
    out = dict(full_index=np.array([1, 2, 10]), strided_index=np.array([100, 999]), stride=np.int64(4),
               snapshot_period=np.int64(50))
    for blk in "ABCD":
        for k in ("ux", "uy", "rho"):
            a = load_pt(os.path.join(d, f"{blk}-domain-decomp-hpt-{k}.pt"))
            out[f"{blk}_{k}_full"] = np.ascontiguousarray(a[..., [1, 2, 10]])
            out[f"{blk}_{k}_strided"] = np.ascontiguousarray(a[::4, ::4][..., [100, 999]])
    save("ddl_512.npz", **out)


def gen_diff(r):
    rng = np.random.default_rng(3)
    psi = rng.standard_normal((19, 31))
    lin = np.add.outer(8.0 * np.arange(12), 1.0 * np.arange(9))
    save("diff5.npz", psi=psi, dx=r.diff_x(psi), dy=r.diff_y(psi), lin=lin, lin_dx=r.diff_x(lin),
         lin_dy=r.diff_y(lin))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-dsf", action="store_true", help="skip the ~minutes-long shear-flow run")
    ap.add_argument("--upo-dir", default="", help="directory holding the .pt files of a finished oracle/_ref/upo run")
    ap.add_argument("--only-upo", action="store_true")
    ap.add_argument("--ddl-dir", default="", help="directory holding the .pt files of a finished oracle/_ref/ddl run")
    a = ap.parse_args()
    build_ref()
    r, o = Ref(), Oracle()
    gen_upo_units(r)
    if a.upo_dir:
        gen_upo_main(a.upo_dir)
    if a.ddl_dir:
        gen_ddl_main(a.ddl_dir)
    if a.only_upo:
        sys.exit(0)
    gen_solver_units(r, o)
    gen_kbc(r, o)
    gen_diff(r)
    gen_hpt()
    gen_ddm()
    gen_gt()
    gen_sbt()
    if not a.skip_dsf:
        gen_dsf()
