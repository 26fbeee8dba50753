"""Physics known-answers for the rows whose oracle cannot be pinned to the reference here (toml++ absent:
colour gradient, immersed boundary) and a long-horizon bound for the reassociated BGK collision.
These are PROPERTY checks, not parity: they show that the kernels compute the physics the models are
built for (Laplace law, cylinder drag, mass conservation, viscous decay); the parity of those rows stays
"unpinned" (DESIGN 7).  Bands are wide enough for the models' known discretisation errors and narrow
enough to catch a wrong stencil, a wrong weight or a lost population."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
import pyoracle  # noqa: E402
from pylbm import _ptr  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


def _cg_state(lib, prm, rho_r, rho_b):
    """populations at rest in equilibrium with the given densities (mrtcg_static_droplet.cpp:456-459)"""
    R, C = rho_r.shape
    d = torch.device("cuda:0")
    u = np.zeros((R, C, 2))
    d_rr, d_rb = torch.from_numpy(rho_r).to(d), torch.from_numpy(rho_b).to(d)
    d_u = torch.zeros((2, R, C), dtype=torch.float64, device=d)
    f_r = torch.empty((9, R, C), dtype=torch.float64, device=d)
    f_b = torch.empty_like(f_r)
    lib.cg_equilibrium(_ptr(f_r), _ptr(d_rr), _ptr(d_u), ct.byref(prm.red), R, C, ct.c_longlong(0), None)
    lib.cg_equilibrium(_ptr(f_b), _ptr(d_rb), _ptr(d_u), ct.byref(prm.blue), R, C, ct.c_longlong(0), None)
    torch.cuda.synchronize()
    return (np.ascontiguousarray(np.moveaxis(f_r.cpu().numpy(), 0, -1)),
            np.ascontiguousarray(np.moveaxis(f_b.cpu().numpy(), 0, -1)), u)


def test_laplace_law_static_droplet(lib):
    """mrtcg_static_droplet physics: a circular droplet of the red fluid at rest; the pressure jump across its
    interface must scale as 1 / R (Laplace, 2-D: dp = sigma_st / R): dp * R is the same for three radii within
    2 % (measured: 0.4 %), and equals the surface tension the perturbation operator is built for: with A_k = 4.5
    sigma s_nu for both colours (mrtcg_static_droplet.cpp / mrtcg_rayleigh_taylor.cpp:450-452) the
    Reis-Phillips relation sigma_st = (2/9)(A_r + A_b) / s_nu gives 2 sigma -- within 3 % (measured: 0.1 %)."""
    N, sigma = 192, 0.1
    prm = pylbm.cg_params(sigma=sigma, gravity=0.0, gravity_c=0.0, add_source=0)
    cs2 = [3.0 * (1.0 - 0.7) / 5.0, 3.0 * (1.0 - 0.1) / 5.0]      # colour.cpp:37
    rr, cc = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    s = np.hypot(rr - N / 2.0, cc - N / 2.0)
    sig = lambda x: 1.0 / (1.0 + np.exp(-x))
    out = []
    for radius in (24.0, 36.0, 48.0):
        rho_r = 3.0 * (1.0 - sig(s - radius))
        rho_b = 1.0 * sig(s - radius)
        f_r, f_b, u = _cg_state(lib, prm, rho_r, rho_b)
        sv = pylbm.CgSolver(lib, N, N, prm)
        sv.set_state(f_r, f_b, rho_r, rho_b, u)
        sv.step(60000)
        p = 0.0
        for _ in range(8):     # the closed box rings with slowly damped pressure waves: average them out
            sv.step(2500)
            st = sv.get_state()
            p = p + (cs2[0] * st["rho_r"] + cs2[1] * st["rho_b"]) / 8.0
        sv.close()
        p_in = p[s < 0.5 * radius].mean()
        p_out = p[(s > radius + 25) & (rr > 8) & (rr < N - 8)].mean()
        r_eff = np.sqrt((st["psi"] > 0).sum() / np.pi)
        umax = np.abs(st["u"]).max()
        out.append((radius, r_eff, p_in - p_out, (p_in - p_out) * r_eff, umax))
        assert abs(r_eff - radius) < 1.5 and umax < 5e-3, out[-1]          # the droplet stays put, spurious currents small
    st_eff = np.array([o[3] for o in out])
    print("laplace:", out)
    assert st_eff.min() > 0
    assert (st_eff.max() - st_eff.min()) / st_eff.mean() < 0.02, st_eff    # dp * R independent of R
    assert abs(st_eff.mean() / (2.0 * sigma) - 1.0) < 0.03, st_eff         # ... and equal to 2 sigma


def test_two_phase_mass_of_each_colour_is_conserved(lib, oracle):
    """Collision (MRT + perturbation + recolouring + source) moves mass between directions and, node by node,
    between nothing else: the mass of EACH colour is invariant.  On a fully periodic box -- where streaming is
    a permutation -- that must hold to rounding (1e-12 over 4000 steps).  With the driver's own edges the
    reference's fix-ups are NOT a permutation at the four corners (mrtcg_rayleigh_taylor.cpp:517-531, SURVEY
    Q5: same-row column copies beside bounce-back rows), so the colours drift -- by less than 1e-4 over 4000
    steps, the oracle's own behaviour (test_gpu_cg.py holds the bits to it)."""
    R, C = 256, 128
    po = pyoracle.cg_params(R, C)
    s0 = oracle.cg_init(po)
    for edges, bound in (("periodic", 1e-12), ("driver", 1e-4)):
        bc = pylbm.Bc() if edges == "periodic" else None
        sv = pylbm.CgSolver(lib, R, C, pylbm.cg_params(), bc=bc)
        sv.set_state(s0["f_r"], s0["f_b"], s0["rho_r"], s0["rho_b"], s0["u"])
        m0 = (s0["f_r"].sum(), s0["f_b"].sum())
        sv.step(4000)
        st = sv.get_state()
        sv.close()
        m1 = (st["f_r"].sum(), st["f_b"].sum())
        print("colour masses:", edges, m0, m1)
        for a, b in zip(m0, m1):
            assert abs(a - b) / a < bound, (edges, a, b)
        assert np.isfinite(st["u"]).all() and np.abs(st["u"]).max() < 0.05


def test_cylinder_drag_coefficient_re20(lib, oracle):
    """cylinder_test geometry (velocity inlet / outlet rows, specular side walls, immersed cylinder) at Re = 20
    with the STANDARD Guo coefficients (3, 9) -- the driver's (1/3, 1/9), SURVEY Q4, weaken the body force
    ninefold and have no literature counterpart.  The driver's source term carries Guo's prefactor (1 - omega/2)
    while its equilibrium uses the velocity WITHOUT the half-force shift (cylinder_test.cpp:104-119), so the
    momentum the fluid receives per step is (1 - omega/2) F_s: that is the drag.  C_d = 2 (1 - omega/2) |F_r| /
    (rho u^2 D): the unbounded-flow value is 2.0-2.1; with 5 % blockage between free-slip walls and the
    direct-forcing boundary's slightly larger effective diameter: 2.0 < C_d < 2.7.  Lift by symmetry ~ 0."""
    X, Y, D, u_in, Re = 640, 400, 20.0, 0.04, 20.0
    nu = u_in * D / Re
    omega = 1.0 / (3.0 * nu + 0.5)
    m = int(round(np.pi * D))
    t = 2 * np.pi * np.arange(m) / m
    x, y = X / 4.0 + 0.5 * D * np.cos(t), Y / 2.0 + 0.5 * D * np.sin(t)   # centred on the channel axis
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    bc = pylbm.Bc(row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY, col_lo=pylbm.EDGE_SPECULAR,
                  col_hi=pylbm.EDGE_SPECULAR, uw_r=u_in)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 0, 1), bc=bc)
    ib = pylbm.Ibm(lib, x, y, X, Y)
    sv.attach_ibm(ib, guo_a=3.0, guo_b=9.0)
    sv.set_f(f0)
    cds = []
    for _ in range(6):
        sv.step(5000)
        F = ib.surface_force()          # force density the boundary exerts on the fluid, summed over the ROI
        k = 2.0 * (1.0 - 0.5 * omega) / (u_in * u_in * D)
        cds.append((-k * F[0], k * F[1]))
    sv.close(); ib.close()
    print("C_d, C_l history:", cds)
    cd, cl = cds[-1]
    assert abs(cds[-1][0] - cds[-2][0]) / cd < 0.01, cds      # steady
    assert 2.0 < cd < 2.7, cds
    assert abs(cl) < 0.03 * cd, cds


def _moments_dev(f):
    rho = f.sum(0)
    jx = f[1] - f[3] + f[5] - f[6] - f[7] + f[8]
    jy = f[2] - f[4] + f[5] + f[6] - f[7] - f[8]
    return rho, jx / rho, jy / rho


def _taylor_green(lib, R, C, U):
    d = torch.device("cuda:0")
    r = torch.arange(R, device=d, dtype=torch.float64).view(-1, 1)
    c = torch.arange(C, device=d, dtype=torch.float64).view(1, -1)
    kr, kc = 2 * np.pi / R, 2 * np.pi / C
    u0 = torch.empty((2, R, C), dtype=torch.float64, device=d)
    u0[0] = U * torch.sin(kr * r) * torch.cos(kc * c)
    u0[1] = -U * torch.cos(kr * r) * torch.sin(kc * c)
    rho0 = torch.ones((R, C), dtype=torch.float64, device=d)
    f0 = torch.empty((9, R, C), dtype=torch.float64, device=d)
    lib.equilibrium(_ptr(f0), _ptr(u0), _ptr(rho0), R, C, None)
    return f0, float((u0 * u0).sum())


def _run_box(lib, f0, omega, n, fast):
    R, C = f0.shape[1:]
    lib.set_tuning(b"bgk_fast", fast)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(omega, 0))
    lib.solver_set_f_soa_dev(sv.h, _ptr(f0))
    sv.step(n)
    assert lib.raw.lbm_solver_block_launches(sv.h) >= (n - 1) // 5      # fused launches, not single steps
    out = torch.empty_like(f0)
    lib.solver_get_f_soa_dev(sv.h, _ptr(out))
    torch.cuda.synchronize()
    sv.close()
    lib.set_tuning(b"bgk_fast", -1)
    rho, ux, uy = _moments_dev(out)
    return ux, uy, float((ux * ux + uy * uy).sum()), float(rho.sum())


def test_reassociated_bgk_long_horizon_vs_reference_order(lib):
    """BASELINE config 2 (8192 x 8192, omega = 1.2, Taylor-Green vortex, U = 0.04): 5000 steps with the default
    reassociated collision (BgkFastModel) and with the reference operation order -- the two velocity fields
    agree to 1e-10 of U (rounding does not accumulate: the flow is smooth and decaying) and mass is conserved
    to 1e-12.  (nothing tested the fast model beyond 100 steps.)"""
    R = C = 8192
    omega, U, n = 1.2, 0.04, 5000
    f0, _ = _taylor_green(lib, R, C, U)
    a = _run_box(lib, f0, omega, n, 1)
    b = _run_box(lib, f0, omega, n, 0)
    du = max(float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max()))
    print(f"fast vs reference order after {n} steps at 8192^2: max |du| / U = {du / U:.3e}; mass {a[3] / (R * C) - 1:.2e} / {b[3] / (R * C) - 1:.2e}")
    assert du / U < 1e-10
    assert abs(a[3] / (R * C) - 1.0) < 1e-12 and abs(b[3] / (R * C) - 1.0) < 1e-12


@pytest.mark.parametrize("fast", [1, 0])
def test_taylor_green_decays_at_the_viscous_rate(lib, fast):
    """512 x 512 periodic box, omega = 1.2, U = 0.02: after 6000 steps the kinetic energy has decayed by
    exp(-2 nu (k_r^2 + k_c^2) t) with nu = (1/omega - 1/2)/3 (solver.cpp's BGK); within 1 %.  (On the 8192^2 box
    the decay over any affordable horizon is smaller than the energy the initial acoustic transient carries.)"""
    R = C = 512
    omega, U, n = 1.2, 0.02, 6000
    f0, e0 = _taylor_green(lib, R, C, U)
    _, _, e1, mass = _run_box(lib, f0, omega, n, fast)
    k = 2 * np.pi / R
    nu = (1.0 / omega - 0.5) / 3.0
    want = 2.0 * nu * 2.0 * k * k * n
    got = -np.log(e1 / e0)
    print(f"viscous decay exponent after {n} steps: {got:.6f} vs {want:.6f} (fast={fast})")
    assert abs(got / want - 1.0) < 0.01, (got, want)
    assert abs(mass / (R * C) - 1.0) < 1e-12


def test_reassociated_kbc_long_horizon_vs_reference_order(lib):
    """The default KBC collision (190 f64 operations, k22 folded into H~_0, gamma's weights without a reciprocal,
    S~ + gamma H~ formed first: csrc/kbc.hpp KbcFastModel) against the reference's operation order (KbcModel,
    src/ulbm.cpp:91-228) over a long horizon: 3000 steps of a decaying Taylor-Green vortex on 256 x 256, s2 = 1.6
    -- a laminar flow, so that the difference measures accumulated rounding and not the divergence of two
    chaotic trajectories.  Mass agrees to 1e-13, the velocity field to 1e-11 of U (measured 4e-14), and the comparison is not
    vacuous (the vortex is still there)."""
    R = C = 256
    U, s2, n = 0.03, 1.6, 3000
    f0, _ = _taylor_green(lib, R, C, U)
    finals = []
    for form in (pylbm.FORM_DEFAULT, pylbm.FORM_REFERENCE_ORDER):
        sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(s2, form=form))
        lib.solver_set_f_soa_dev(sv.h, _ptr(f0))
        sv.step(n)
        out = torch.empty_like(f0)
        lib.solver_get_f_soa_dev(sv.h, _ptr(out))
        torch.cuda.synchronize()
        finals.append(_moments_dev(out))
        sv.close()
    (rf, uxf, uyf), (rr, uxr, uyr) = finals
    assert bool(torch.isfinite(uxf).all()) and bool(torch.isfinite(uxr).all())
    assert abs(float(rf.sum()) / float(rr.sum()) - 1.0) < 1e-13
    du = max(float((uxf - uxr).abs().max()), float((uyf - uyr).abs().max()))
    print(f"KBC default vs reference order after {n} steps at {R}^2: max |du| / U = {du / U:.3e}; max |u| / U = {float(uxr.abs().max()) / U:.3f}")
    assert du / U < 1e-11      # measured 4.1e-14
    assert float(uxr.abs().max()) > 0.05 * U
