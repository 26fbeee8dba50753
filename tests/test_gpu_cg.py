"""GPU parity tests for the colour-gradient two-phase MRT step (BASELINE config 4) through the
C ABI.  The reference operators live in test/mrtcg_rayleigh_taylor.cpp, which cannot be
compiled here (toml++ absent) -- the oracle for this path is "parity unpinned" except for its
sub-operators differential::x/y and solver::advect/calc_u (pinned in test_oracle_golden.py).
Bar vs the oracle: bitwise (same expression order, -ffp-contract=off); stated tolerance for
the north star: 1e-9 relative on rho, u after <= 200 steps."""
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


def run_pair(lib, oracle, R, C, steps_list, gravity=6.25e-6, sigma=0.1):
    po = pyoracle.cg_params(R, C, sigma=sigma, gravity=gravity)
    pg = pylbm.cg_params(sigma=sigma, gravity=gravity)
    s0 = oracle.cg_init(po)
    sv = pylbm.CgSolver(lib, R, C, pg)
    sv.set_state(s0["f_r"], s0["f_b"], s0["rho_r"], s0["rho_b"], s0["u"])
    done = 0
    for n in steps_list:
        sv.step(n - done)
        done = n
        yield n, sv.get_state(), oracle.cg_steps(po, s0, n)
    sv.close()


@pytest.mark.parametrize("R,C", [(64, 32), (37, 45)])
def test_cg_steps_vs_oracle(lib, oracle, R, C):
    for n, got, want in run_pair(lib, oracle, R, C, [1, 2, 5, 50]):
        for k in ("f_r", "f_b", "rho_r", "rho_b", "u", "psi", "s_nu"):
            assert relerr(got[k], want[k]) < 1e-12, (n, k, relerr(got[k], want[k]))
            assert bits_equal(got[k], want[k]), (n, k, ulp_diff(got[k], want[k]))


def test_cg_256x128_200_steps_tolerance(lib, oracle):
    """north-star tolerance case: 256 x 128 (the shipped TOML's domain), 200 steps"""
    for n, got, want in run_pair(lib, oracle, 256, 128, [200]):
        assert relerr(got["rho_r"], want["rho_r"]) < 1e-9
        assert relerr(got["rho_b"], want["rho_b"]) < 1e-9
        assert relerr(got["u"], want["u"]) < 1e-9
        # physics sanity: total mass of each colour is conserved by recolouring + bounce-back
        for k in ("rho_r", "rho_b"):
            assert abs(got[k].sum() - want[k].sum()) / want[k].sum() < 1e-12


def test_cg_large_box_vs_oracle(lib, oracle):
    """1024 x 512 (fast interior tiles, many blocks): 10 steps, bitwise vs the oracle; the
    driver's model is only approximately mass-conserving per colour (its source term is added
    unweighted to each colour), so mass is checked to 1e-9, not to rounding."""
    R, C = 1024, 512
    for n, got, want in run_pair(lib, oracle, R, C, [10]):
        for k in ("f_r", "f_b", "rho_r", "rho_b", "u"):
            assert bits_equal(got[k], want[k]), (k, ulp_diff(got[k], want[k]))
        s0 = oracle.cg_init(pyoracle.cg_params(R, C))
        for k in ("rho_r", "rho_b"):
            assert abs(got[k].sum() - s0[k].sum()) / s0[k].sum() < 1e-9
        assert (got["psi"][: R // 4] > 0.99).all() and (got["psi"][-R // 4:] < -0.99).all()
