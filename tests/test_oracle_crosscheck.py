"""CPU suite: the two INDEPENDENT restatements of the reference files that cannot be compiled
here (toml++): oracle/lbm_oracle.cpp (per-node C++) vs oracle/torch_restatement.py (the reference's
own tensor operations, statement by statement).  Agreement to rounding = no transcription slip in
either; it is not a pin to the reference itself."""
import numpy as np
import pytest
from conftest import relerr

import pyoracle
import torch_restatement as tr


@pytest.mark.parametrize("R,C,steps", [(24, 16, 1), (24, 16, 12), (31, 20, 5)])
def test_colour_gradient_restatements_agree(oracle, R, C, steps):
    red, blue = (3.0, 0.7, 0.04, 0.7), (1.0, 0.1, 0.04, -0.7)
    p = pyoracle.cg_params(R, C, red=red, blue=blue, sigma=0.1, gravity=6.25e-6)
    a = oracle.cg_steps(p, oracle.cg_init(p), steps)
    b = tr.cg_run(R, C, red, blue, 0.1, 6.25e-6, steps)
    for k in ("f_r", "f_b", "rho_r", "rho_b", "u", "psi", "s_nu"):
        assert relerr(a[k], b[k]) < 1e-12, (k, relerr(a[k], b[k]))


def circle(cx, cy, radius):
    n = int(round(2 * np.pi * radius))
    t = 2 * np.pi * np.arange(n) / n
    return cx + radius * np.cos(t), cy + radius * np.sin(t)


def test_ibm_force_restatements_agree(oracle):
    import torch
    X, Y = 40, 36
    x, y = circle(20.3, 17.6, 6.0)
    rng = np.random.default_rng(2)
    u = 0.05 * rng.standard_normal((X, Y, 2))
    rho = 1 + 0.02 * rng.standard_normal((X, Y))
    a = oracle.ibm_force(x, y, u, rho)
    ib = tr.Ibm(list(x), list(y))
    b = ib.eulerian_force_density(torch.from_numpy(u), torch.from_numpy(rho).unsqueeze(-1)).numpy()
    assert (ib.rows.start, ib.rows.stop, ib.cols.start, ib.cols.stop) == oracle.ibm_roi(x, y)
    assert relerr(a, b) < 1e-13


@pytest.mark.parametrize("steps", [1, 6])
def test_cylinder_driver_restatements_agree(oracle, steps):
    X, Y, omega, u_in = 48, 40, 1.0 / 0.56, 0.05
    x, y = circle(14.4, 20.2, 5.0)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, steps)
    ft, ut, rhot, Fst = tr.cylinder_run(list(x), list(y), X, Y, omega, u_in, steps)
    assert relerr(fo, ft) < 1e-12 and relerr(uo, ut) < 1e-11 and relerr(rhoo, rhot) < 1e-13
    assert np.allclose(Fso, Fst, rtol=1e-10, atol=1e-16)


@pytest.mark.parametrize("steps", [1, 8])
def test_static_droplet_restatements_agree(oracle, steps):
    """test/mrtcg_static_droplet.cpp (sigmoid droplet, Fg = (0, -6.25e-6) shifts u
    only, source term commented out, sigma = 0.1)."""
    R, C = 64, 64
    red, blue = (3.0, 0.7, 0.04, 0.7), (1.0, 0.1, 0.04, -0.7)
    p = pyoracle.cg_params(R, C, red=red, blue=blue, sigma=0.1, gravity=0.0, gravity_c=-6.25e-6, add_source=0)
    a = oracle.cg_steps(p, oracle.cg_init_droplet(p), steps)
    b = tr.cg_run(R, C, red, blue, 0.1, -6.25e-6, steps, droplet=True)
    for k in ("f_r", "f_b", "rho_r", "rho_b", "u", "psi", "s_nu"):
        # smooth sigmoid interface: the minority colour is ~1e-11 inside the droplet, which
        # amplifies relative rounding noise between the two summation orders
        assert relerr(a[k], b[k]) < 1e-10, (k, relerr(a[k], b[k]))
