"""GPU parity for SURVEY 8(f) row 1 -- the remaining single-phase BGK drivers expressed as
boundary / force presets of the same fused kernels: specular_boundary_test and gravity_test
(pinned by golden vectors of their unmodified mains), free_stream_test (needs toml++: oracle only)."""
import numpy as np
import pytest
from conftest import golden, relerr

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
from gpu_util import bits_equal, ulp_diff  # noqa: E402
from test_oracle_golden import sbt_constants  # noqa: E402

W9 = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


def rest_state(H, W):
    f = np.empty((H, W, 9))
    f[...] = W9            # incomp_equilibrium(u = 0, rho = 1)
    return f


def test_specular_boundary_driver(lib, oracle):
    g = golden("sbt_51x51.npz")
    omega, rin, rout = sbt_constants()
    bc = pylbm.Bc(col_lo=pylbm.EDGE_SPECULAR, col_hi=pylbm.EDGE_SPECULAR, pressure_rows=1,
                  rho_inlet=rin, rho_outlet=rout)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, 51, 51, pylbm.BgkParams(omega, 0), bc=bc)
    sv.set_f(rest_state(51, 51))
    done = 0
    for k, t in enumerate(g["steps"]):
        sv.step(int(t) - done, record_moments=True)
        done = int(t)
        f = sv.get_f()
        assert relerr(f, g["fs"][..., k]) < 1e-12, t                       # unmodified reference main
        if t <= 1000:
            assert bits_equal(f, oracle.sbt_run(51, 51, done, omega, rin, rout)["f"]), t
    rho, u = sv.moments()
    assert relerr(u[..., 0], g["ux"][..., -1]) < 1e-10
    sv.close()


def test_gravity_driver(lib, oracle):
    g = golden("gt_21x21.npz")
    omega = 1.0 / (np.sqrt(3.0 / 16.0) + 0.5)
    bc = pylbm.Bc(col_lo=pylbm.EDGE_BOUNCE_BACK, col_hi=pylbm.EDGE_BOUNCE_BACK, pressure_rows=1,
                  rho_inlet=1.0, rho_outlet=1.0)
    prm = pylbm.BgkParams(omega, 1, force=(-0.0003, 0.0))
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, 21, 21, prm, bc=bc)
    sv.set_f(rest_state(21, 21))
    done = 0
    for k, t in enumerate(g["steps"]):
        sv.step(int(t) - done, record_moments=True)
        done = int(t)
        f = sv.get_f()
        assert relerr(f, g["fs"][..., k]) < 1e-11, t
        o = oracle.gravity_run(21, 21, done, omega, -0.0003, 0.0, check_convergence=False)
        assert bits_equal(f, o["f"]), (t, ulp_diff(f, o["f"]))
    rho, u = sv.moments()
    assert relerr(u[..., 0], g["ux"][..., -1]) < 1e-10
    assert bits_equal(u, o["u"]) and bits_equal(rho, o["rho"])   # u includes the += Fg shift
    sv.close()


@pytest.mark.parametrize("X,Y", [(40, 30), (96, 128)])
def test_free_stream_driver(lib, oracle, X, Y):
    omega, uw = 1.0 / 0.55, 0.1
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = 0.1                             # free_stream_test.cpp:52
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    bc = pylbm.Bc(row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY,
                  col_lo=pylbm.EDGE_SPECULAR, col_hi=pylbm.EDGE_SPECULAR, uw_r=uw)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 1), bc=bc)
    sv.set_f(f0)
    sv.step(60, record_moments=True)
    f = sv.get_f()
    fo, uo, rhoo = oracle.free_stream_steps(f0, omega, uw, 60)
    assert bits_equal(f, fo), ulp_diff(f, fo)
    rho, u = sv.moments()
    assert bits_equal(u, uo) and bits_equal(rho, rhoo)
    assert np.all(np.isfinite(u))
    sv.close()
