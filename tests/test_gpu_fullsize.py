"""Full-size parity for BASELINE configs 4 and 5 against the CPU oracle: the launches
that only the benchmark scripts used to exercise -- the 8192 x 2048 two-phase step with its XCD tile
order and inner / frame split, and the 16384 x 4096 immersed-boundary 5-step block -- compared with the
OpenMP oracle on the GPU box's host cores.  A handful of steps each, so the suite stays short."""
import time

import numpy as np
import pytest
from conftest import relerr

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
import pyoracle  # noqa: E402
from gpu_util import bits_equal, ulp_diff  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


def test_config5_fullsize_ibm_block_vs_oracle(lib, oracle):
    """16384 x 4096, cylinder of diameter 300 (942 markers) at (rows / 4, cols / 2) as SURVEY 8(d) states
    config 5: the first iteration + ONE 5-step immersed-boundary block (band in forced single steps, far
    rows through the wall-carrying 5-step window) == 6 oracle iterations bit for bit."""
    X, Y, omega, u_in = 16384, 4096, 1.0 / 0.55, 0.04
    m = int(round(np.pi * 300))
    t = 2 * np.pi * np.arange(m) / m
    x, y = X / 4.0 + 0.37 + 150.0 * np.cos(t), Y / 2.0 + 0.21 + 150.0 * np.sin(t)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    del u0
    bc = pylbm.Bc(row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY, col_lo=pylbm.EDGE_SPECULAR,
                  col_hi=pylbm.EDGE_SPECULAR, uw_r=u_in)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 0, 1, form=pylbm.FORM_REFERENCE_ORDER), bc=bc)
    ib = pylbm.Ibm(lib, x, y, X, Y)
    sv.attach_ibm(ib)
    sv.set_f(f0)
    sv.step(6)
    assert lib.raw.lbm_solver_block_launches(sv.h) == 1      # the 5 steps after the first went as ONE block
    f = sv.get_f()
    Fs = ib.surface_force()
    sv.close(); ib.close()
    t0 = time.time()
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, 6)
    print(f"oracle: 6 iterations of 16384 x 4096 in {time.time() - t0:.1f} s on {oracle.max_threads()} threads")
    del f0, uo, rhoo
    assert bits_equal(f, fo), ulp_diff(f, fo)
    assert np.allclose(Fs, Fso, rtol=1e-10) and Fso[0] < 0


@pytest.mark.parametrize("mode", ["two_pass", "fused"])
def test_config4_fullsize_two_phase_step_vs_oracle(lib, oracle, mode):
    """8192 x 2048 (rows x cols), init_rho_cosine, 3 iterations: the reference-order two-pass step BITWISE,
    the fused one-launch step (inner tiles through the boundary-free instantiation, XCD-paired tile order,
    frame launch) within 1e-11 of the oracle on f, rho, u."""
    R, C, n = 8192, 2048, 3
    lib.set_tuning(b"cg_fused", 1 if mode == "fused" else 0)
    try:
        po = pyoracle.cg_params(R, C)
        pg = pylbm.cg_params()
        s0 = oracle.cg_init(po)
        sv = pylbm.CgSolver(lib, R, C, pg)
        sv.set_state(s0["f_r"], s0["f_b"], s0["rho_r"], s0["rho_b"], s0["u"])
        sv.step(n)
        got = sv.get_state()
        sv.close()
        t0 = time.time()
        want = oracle.cg_steps(po, s0, n)
        print(f"oracle: {n} two-phase iterations of {R} x {C} in {time.time() - t0:.1f} s")
        for k in ("f_r", "f_b", "rho_r", "rho_b", "u"):
            if mode == "two_pass":
                assert bits_equal(got[k], want[k]), (k, ulp_diff(got[k], want[k]))
            else:
                assert relerr(got[k], want[k]) < 1e-11, (k, relerr(got[k], want[k]))
    finally:
        lib.set_tuning(b"cg_fused", -1)


@pytest.mark.parametrize("form", [41, 42])
def test_config4_fullsize_inner_kernels_leave_the_same_bits(lib, oracle, form):
    """8192 x 2048, 4 iterations through lbm_cg_solver_step: the default (round 4: 16 x 64 tiles, two nodes per thread,
    patches of 8 x 2 tiles per XCD -- form 102), round 3's 16 x 32 tile kernel (cg_big = 0) and the opt-in walking block
    (cg_big = 0, cg_strip2 = 41 / 42: 22 chunks of 34 strips, XCD-contiguous order, 32-bit plane offsets) leave the SAME
    BITS -- populations of both colours and every observable field -- and each is the form that ran."""
    R, C, n = 8192, 2048, 4
    po = pyoracle.cg_params(R, C)
    pg = pylbm.cg_params()
    s0 = oracle.cg_init(po)
    got = {}
    try:
        for big, strip2, want in ((-1, -1, 102), (0, 0, 0), (0, form, form)):
            lib.set_tuning(b"cg_big", big)
            lib.set_tuning(b"cg_strip2", strip2)
            sv = pylbm.CgSolver(lib, R, C, pg)
            sv.set_state(s0["f_r"], s0["f_b"], s0["rho_r"], s0["rho_b"], s0["u"])
            sv.step(n)
            got[want] = sv.get_state()
            assert lib.raw.lbm_cg_last_inner_form() == want
            sv.close()
    finally:
        lib.set_tuning(b"cg_strip2", -1)
        lib.set_tuning(b"cg_big", -1)
    for f in (0, form):
        for k in got[102]:
            assert bits_equal(got[f][k], got[102][k]), (f, k, ulp_diff(got[f][k], got[102][k]))
