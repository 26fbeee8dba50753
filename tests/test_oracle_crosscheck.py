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


# ---- known answers of the PUBLISHED algorithm (multi-direct forcing with Peskin's 4-point kernel, src/ibm.cpp:39-45,
# :158-190): the files that hold it cannot be compiled here, so these pin the restatement to what the method itself
# guarantees -- independent of either restatement's code ----------------------------------------------------------
def test_ibm_kernel_is_a_partition_of_unity_uniform_field_known_answer(oracle):
    """phi sums to 1 over a marker's 4 x 4 box wherever the marker sits, so in a UNIFORM field every marker
    interpolates exactly (rho0, u0), its force is f_j = -2 rho0 u0 (ibm.cpp:176-178), and spreading moves all
    of it onto the lattice: after ONE forcing iteration sum_nodes F = -2 rho0 u0 N_markers."""
    X, Y = 64, 56
    for cx, cy, radius in ((30.3, 27.6, 9.0), (31.0, 28.5, 12.25), (29.87, 26.11, 5.5)):
        x, y = circle(cx, cy, radius)
        u0, rho0 = np.array([0.043, -0.017]), 1.013
        u = np.broadcast_to(u0, (X, Y, 2)).copy()
        rho = np.full((X, Y), rho0)
        F = oracle.ibm_force(x, y, u, rho, m_max=2)
        total = F.reshape(-1, 2).sum(0)
        want = -2.0 * rho0 * u0 * len(x)
        assert np.allclose(total, want, rtol=1e-12, atol=0), (total, want)
    # no flow, no force
    assert not oracle.ibm_force(x, y, np.zeros((X, Y, 2)), np.ones((X, Y)), m_max=5).any()


def test_ibm_forcing_is_linear_in_the_velocity(oracle):
    """f_j = -2 rho_j u_j and u <- u + F / (2 rho) are linear in u at fixed rho, through all m_max - 1
    iterations: F(a u1 + b u2) = a F(u1) + b F(u2)."""
    X, Y = 48, 44
    x, y = circle(23.4, 21.7, 8.0)
    rng = np.random.default_rng(9)
    u1, u2 = 0.05 * rng.standard_normal((X, Y, 2)), 0.05 * rng.standard_normal((X, Y, 2))
    rho = 1 + 0.03 * rng.standard_normal((X, Y))
    F1, F2 = oracle.ibm_force(x, y, u1, rho), oracle.ibm_force(x, y, u2, rho)
    F12 = oracle.ibm_force(x, y, 0.7 * u1 - 1.9 * u2, rho)
    assert relerr(F12, 0.7 * F1 - 1.9 * F2) < 1e-12


def test_ibm_forcing_moves_with_the_boundary(oracle):
    """Markers and fields shifted together by whole lattice units (2 rows, 3 columns: exact in floating point for
    these coordinates): the force field shifts with them, bit for bit -- the ROI origin and the box of a marker
    are floor-based (ibm.cpp:20-37, :124-153), nothing else knows where the boundary is."""
    X, Y = 56, 52
    x, y = circle(24.25, 22.5, 7.0)
    x, y = np.round(x * 1024) / 1024, np.round(y * 1024) / 1024   # shifts by integers stay exact
    rng = np.random.default_rng(4)
    u = 0.05 * rng.standard_normal((X, Y, 2))
    rho = 1 + 0.02 * rng.standard_normal((X, Y))
    F0 = oracle.ibm_force(x, y, u, rho)
    F1 = oracle.ibm_force(x + 2, y + 3, np.roll(u, (2, 3), axis=(0, 1)), np.roll(rho, (2, 3), axis=(0, 1)))
    r0 = oracle.ibm_roi(x, y)
    r1 = oracle.ibm_roi(x + 2, y + 3)
    assert r1 == (r0[0] + 2, r0[1] + 2, r0[2] + 3, r0[3] + 3)
    assert np.array_equal(F0, F1)


def test_colour_gradient_collision_conserves_what_the_method_conserves(oracle):
    """Known answers of the colour-gradient MRT method itself: the single-phase operator relaxes towards an equilibrium
    with the node's own rho_k and momentum, the perturbation operator carries neither mass nor momentum, recolouring
    redistributes the total population between the colours with rho_k fixed (mrtcg_rayleigh_taylor.cpp:212-336).  So
    at EVERY node, after the whole collision: each colour's mass unchanged, and -- without gravity -- the total
    momentum unchanged.  Developed Rayleigh-Taylor state (interface, velocities), walls included."""
    cx = np.array([0, 1, 0, -1, 0, 1, -1, -1, 1.0])
    cy = np.array([0, 0, 1, 0, -1, 1, 1, -1, -1.0])
    red, blue = (3.0, 0.7, 0.04, 0.7), (1.0, 0.1, 0.04, -0.7)
    R, C = 48, 32
    # develop a flow under gravity, then look at one collision with gravity switched off
    pg = pyoracle.cg_params(R, C, red=red, blue=blue, sigma=0.1, gravity=6.25e-6)
    s = oracle.cg_steps(pg, oracle.cg_init(pg), 40)
    st = {k: s[k] for k in ("f_r", "f_b", "rho_r", "rho_b", "u")}
    assert np.abs((st["f_r"] + st["f_b"]) @ cx).max() > 1e-5          # there IS momentum to conserve
    p0 = pyoracle.cg_params(R, C, red=red, blue=blue, sigma=0.1, gravity=0.0)
    out = oracle.cg_steps(p0, st, 1, want_col=True)
    for k in ("r", "b"):
        assert np.abs(out["col_" + k].sum(-1) - st["f_" + k].sum(-1)).max() < 1e-14, k
    tot0, tot1 = st["f_r"] + st["f_b"], out["col_r"] + out["col_b"]
    assert np.abs(tot1 @ cx - tot0 @ cx).max() < 1e-15
    assert np.abs(tot1 @ cy - tot0 @ cy).max() < 1e-15
    # and the moments the step works with ARE the zeroth moments of its input
    assert np.abs(st["rho_r"] - st["f_r"].sum(-1)).max() < 1e-14


def test_colour_rest_equilibrium_weights_known_answer(oracle):
    """Known answer for `colour` (src/colour.cpp:11-64): the rest equilibrium of fluid k is rho_k phi^k with
    phi^k = (alpha_k, (1 - alpha_k) / 5 x 4, (1 - alpha_k) / 20 x 4) -- a partition of unity whose second moment gives
    the colour's sound speed 3 (1 - alpha_k) / 5 --, and the shipped parameters balance the two pressures across an
    interface at rest: rho_r / rho_b = (1 - alpha_b) / (1 - alpha_r)  (mrtcg-rayleigh-taylor-gamma3.toml: 3 / 1, 0.7 / 0.1)."""
    red, blue = (3.0, 0.7, 0.04, 0.7), (1.0, 0.1, 0.04, -0.7)
    p = pyoracle.cg_params(48, 32, red=red, blue=blue, sigma=0.1, gravity=0.0)
    s0 = oracle.cg_init(p)
    for key, (rho0, alpha, _, _) in (("r", red), ("b", blue)):
        rho, f = s0["rho_" + key], s0["f_" + key]
        i = np.unravel_index(np.argmax(rho), rho.shape)
        assert rho[i] == rho0
        phi = f[i] / rho[i]
        want = np.array([alpha] + [(1 - alpha) / 5] * 4 + [(1 - alpha) / 20] * 4)
        assert np.allclose(phi, want, rtol=1e-14, atol=0), (phi, want)
        assert abs(phi.sum() - 1) < 1e-15
    assert np.all(s0["u"] == 0)
    assert abs(red[0] / blue[0] - (1 - blue[1]) / (1 - red[1])) < 1e-12
