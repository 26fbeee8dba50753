"""CPU suite: the oracle (oracle/lbm_oracle.cpp) against golden vectors produced by the
UNMODIFIED reference (tests/golden/make_golden.py).  Tolerances: the oracle sums in a
fixed q order, libtorch's reductions/dgemm do not promise one -> a few ulp per step."""
import numpy as np
import pytest
from conftest import golden, relerr

from pyoracle import hpt_params


@pytest.mark.parametrize("tag", ["a", "b"])
def test_solver_unit_functions(oracle, tag):
    g = golden("solver_units.npz")
    f, u, rho = g[f"{tag}_f_in"], g[f"{tag}_u_in"], g[f"{tag}_rho_in"]
    assert relerr(oracle.calc_rho(f), g[f"{tag}_calc_rho"]) < 1e-15
    assert relerr(oracle.calc_u(f, g[f"{tag}_calc_rho"]), g[f"{tag}_calc_u"]) < 1e-14
    assert relerr(oracle.calc_incomp_u(f), g[f"{tag}_calc_incomp_u"]) < 1e-14
    assert relerr(oracle.equilibrium(u, rho), g[f"{tag}_equilibrium"]) < 1e-15
    assert relerr(oracle.incomp_equilibrium(u, rho), g[f"{tag}_incomp_equilibrium"]) < 1e-15
    assert relerr(oracle.collision(f, g[f"{tag}_equilibrium"], 1.2), g[f"{tag}_collision_w1.2"]) < 1e-15
    # streaming is a pure permutation: bit-exact node indexing
    assert np.array_equal(oracle.advect(f), g[f"{tag}_advect"])


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("n", [1, 10, 100])
@pytest.mark.parametrize("inc", [0, 1])
def test_bgk_periodic_loop(oracle, tag, n, inc):
    g = golden("solver_units.npz")
    f, rho, u = oracle.bgk_periodic_steps(g[f"{tag}_f_in"], 1.2, n, bool(inc))
    assert relerr(f, g[f"{tag}_bgk{n}_inc{inc}_f"]) < 1e-13
    assert relerr(rho, g[f"{tag}_bgk{n}_inc{inc}_rho"]) < 1e-13
    assert relerr(u, g[f"{tag}_bgk{n}_inc{inc}_u"]) < 1e-12


def test_poiseuille_main_known_answers(oracle):
    """test/horizontal_poiseuille_test.cpp: snapshots of the unmodified main + its L2 assert."""
    g = golden("hpt_21x21.npz")
    steps = g["steps"]
    for k, t in enumerate(steps):
        p = hpt_params(21, 21, int(t), check_convergence=1)
        out = oracle.hpt_run(p)
        assert out["steps"] == t  # the reference never hit its early exit (T = 8301 steps run)
        # fs[..., t] = f_adve entering step t
        assert relerr(out["f"], g["fs"][..., k]) < 1e-12, t
    # moments saved at snapshot t are those computed during step t-1
    T = int(g["T"])
    out = oracle.hpt_run(hpt_params(21, 21, T - 1))
    k = len(steps) - 1
    assert relerr(out["u"][..., 0], g["ux"][..., k]) < 1e-11
    assert relerr(out["rho"], g["rho"][..., k]) < 1e-13
    full = oracle.hpt_run(hpt_params(21, 21, T))
    assert full["l2"] <= 1e-11  # the reference's only assertion (:172-175)
    assert abs(full["l2"] - float(g["l2_printed"])) < 5e-14
    # known-answer centre-row profile recorded in SURVEY.md 8c
    ka = [0.009585127953269833, 0.07972021053814406, 0.10309857139975714, 0.079720210538144,
          0.00958512795326982]
    assert np.allclose(full["u"][10, [0, 5, 10, 15, 20], 0], ka, rtol=1e-11, atol=0)


def test_decompose_domain_main(oracle):
    g = golden("ddm_21x21.npz")
    p = hpt_params(21, 21, 500)
    for k, t in enumerate(g["steps"]):
        o = oracle.ddm_run(21, 21, int(t), p.omega, p.rho_inlet, p.rho_outlet)
        assert relerr(o["fA"], g["A_fs"][..., k]) < 1e-12
        assert relerr(o["fB"], g["B_fs"][..., k]) < 1e-12


def test_kbc_collide_and_steps(oracle):
    g = golden("kbc_units.npz")
    s2 = float(g["s2"])
    coll, gamma = oracle.kbc_collide(g["f_in"], g["m0_in"], g["m1_in"], s2)
    assert relerr(coll, g["coll1"]) < 1e-13
    assert np.all(np.isfinite(gamma))
    f1, m0, m1 = oracle.kbc_steps(g["f_in"], g["m0_in"], g["m1_in"], s2, 1)
    assert relerr(f1, g["f1"]) < 1e-13 and relerr(m0, g["m0_1"]) < 1e-14 and relerr(m1, g["m1_1"]) < 1e-12
    # the driver's initialisation: eval_equilibrium with ux2 = uy2 = 0 (ctor state)
    f0 = oracle.kbc_equilibrium(g["shear_m0"], g["shear_m1"], use_zero_u2=True)
    assert relerr(f0, g["shear0_f"]) < 1e-15
    for n in (1, 5, 20):
        f, m0, m1 = oracle.kbc_steps(f0, g["shear_m0"], g["shear_m1"], s2, n)
        assert relerr(f, g[f"shear{n}_f"]) < 1e-12
        assert relerr(m1, g[f"shear{n}_m1"]) < 1e-11


def test_double_shear_main_snapshots(oracle):
    """test/ulbm_double_shear_flow.cpp unmodified main, 128x128, snapshots every 10 steps."""
    try:
        g = golden("dsf_128.npz")
    except FileNotFoundError:
        pytest.skip("dsf_128.npz not generated")
    s2 = 1.0 / (0.5 + 3.0 * 1.70766666e-4)
    m0, m1 = oracle.kbc_shear_init(128, 128)
    f = oracle.kbc_equilibrium(m0, m1, use_zero_u2=True)
    t = 0
    for k, i in enumerate(g["snap_index"]):
        n = int(i) * int(g["snapshot_period"]) - t
        f, m0, m1 = oracle.kbc_steps(f, m0, m1, s2, n)
        t += n
        tol = 1e-11 if t <= 100 else 1e-9
        assert relerr(m1[..., 0], g["ux"][..., k]) < tol, t
        assert relerr(m1[..., 1], g["uy"][..., k]) < tol, t
        assert relerr(m0, g["rho"][..., k]) < tol, t


def test_differential(oracle):
    g = golden("diff5.npz")
    assert relerr(oracle.diff_x(g["psi"]), g["dx"]) < 1e-14
    assert relerr(oracle.diff_y(g["psi"]), g["dy"]) < 1e-14
    assert relerr(oracle.diff_x(g["lin"]), g["lin_dx"]) < 1e-14
    assert relerr(oracle.diff_y(g["lin"]), g["lin_dy"]) < 1e-14
    # interior of a linear field: exact derivative (8 per row, 1 per column)
    assert np.allclose(oracle.diff_x(g["lin"])[2:-2, 2:-2], 8.0, rtol=1e-13)
    assert np.allclose(oracle.diff_y(g["lin"])[2:-2, 2:-2], 1.0, rtol=1e-13)


def test_oracle_against_live_reference(oracle, ref):
    """Where oracle/_ref exists (build container), re-check on fresh random inputs."""
    rng = np.random.default_rng(11)
    R, C = 29, 41
    rho = 1 + 0.02 * rng.standard_normal((R, C))
    u = 0.04 * rng.standard_normal((R, C, 2))
    f = ref.equilibrium(u, rho) * (1 + 0.02 * rng.standard_normal((R, C, 9)))
    assert np.array_equal(oracle.advect(f), ref.advect(f))
    a = oracle.bgk_periodic_steps(f, 1.7, 25)
    b = ref.bgk_periodic_steps(f, 1.7, 25)
    assert all(relerr(x, y) < 1e-12 for x, y in zip(a, b))
    psi = rng.standard_normal((R, C))
    assert relerr(oracle.diff_x(psi), ref.diff_x(psi)) < 1e-14


def sbt_constants(H=51, W=51):
    """test/specular_boundary_test.cpp:50-66"""
    tau = np.sqrt(3.0 / 16.0) + 0.5
    nu = (2.0 * tau - 1.0) / 6.0
    p_grad = 8.0 * nu * 0.1 / (W * W)
    return 1.0 / tau, 3.0 * (H - 1) * p_grad + 1.0, 1.0   # omega, rho_inlet, rho_outlet


def test_specular_boundary_main(oracle):
    """unmodified main of test/specular_boundary_test.cpp (51 x 51)."""
    g = golden("sbt_51x51.npz")
    omega, rin, rout = sbt_constants()
    for k, t in enumerate(g["steps"]):
        o = oracle.sbt_run(51, 51, int(t), omega, rin, rout)
        assert relerr(o["f"], g["fs"][..., k]) < 1e-12, t
    assert relerr(o["u"][..., 0], g["ux"][..., -1]) < 1e-10   # moments of iteration t-1


def test_gravity_main(oracle):
    """unmodified main of test/gravity_test.cpp (21 x 21, Fg = (-3e-4, 0)); it
    stops by its own convergence rule at t = 8301."""
    g = golden("gt_21x21.npz")
    omega = 1.0 / (np.sqrt(3.0 / 16.0) + 0.5)
    for k, t in enumerate(g["steps"][:-1]):
        o = oracle.gravity_run(21, 21, int(t), omega, -0.0003, 0.0, check_convergence=False)
        assert relerr(o["f"], g["fs"][..., k]) < 1e-12, t
    full = oracle.gravity_run(21, 21, int(g["T"]), omega, -0.0003, 0.0, check_convergence=True)
    assert full["steps"] == int(g["last_t"]) == 8301           # same early exit as the reference
    assert relerr(full["f"], g["fs"][..., -1]) < 1e-11
    assert relerr(full["u"][..., 0], g["ux"][..., -1]) < 1e-10


def test_ulbm_poiseuille_loop_vs_reference_classes(oracle):
    """orc_upo_steps (test/ulbm_poiseuille.cpp:104-141) against the same loop run on the reference's
    own ulbm::d2q9::kbc + solver::incomp_equilibrium (fixture upo_units.npz, ref_upo_steps)."""
    g = golden("upo_units.npz")
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    for tag, steps in (("a", (1, 2, 10, 100)), ("b", (50,))):
        H, W = (int(v) for v in g[f"{tag}_shape"])
        rin = 3.0 * (H - 1) * (8.0 * nu * 0.05 / (W * W)) + 1.0
        for n in steps:
            f, m0, m1 = oracle.upo_steps(H, W, s2, rin, 1.0, n)
            assert relerr(f, g[f"{tag}_{n}_f"]) < 1e-12, (tag, n)
            assert relerr(m0, g[f"{tag}_{n}_m0"]) < 1e-12, (tag, n)
            assert np.abs(m1 - g[f"{tag}_{n}_m1"]).max() < 1e-12, (tag, n)


def _ddl_expect(g, blk, kind, j):
    return {k: g[f"{blk}_{k}_{kind}"][..., j] for k in ("ux", "uy", "rho")}


def test_decompose_domain_loop_vs_unmodified_main(oracle):
    """orc_ddl_run against the snapshots of the unmodified test/decompose_domain_loop.cpp main
    (L = 512): snapshot i = moments computed in iteration 50 i - 1, A's u_x on the force rows
    already carrying the + F of :114."""
    g = golden("ddl_512.npz")
    L, L4 = 512, 128
    for j, i in enumerate(g["full_index"][:2]):
        got = oracle.ddl_run(L, 50 * int(i))
        for k, blk in enumerate("ABCD"):
            want = _ddl_expect(g, blk, "full", j)
            ux = got["u"][k][..., 0].copy()
            if blk == "A":
                ux[L4 + 5:L4 + 55] += 3e-3
            assert np.abs(ux - want["ux"]).max() < 1e-13, (i, blk)
            assert np.abs(got["u"][k][..., 1] - want["uy"]).max() < 1e-13, (i, blk)
            assert relerr(got["rho"][k], want["rho"]) < 1e-14, (i, blk)


def test_ulbm_poiseuille_vs_unmodified_main(oracle):
    """orc_upo_steps against the snapshots of the unmodified test/ulbm_poiseuille.cpp main (128 x 128,
    300000 iterations, fixture upo_128.npz): snapshot i = kbc.m0 / kbc.m1 after 100 i iterations."""
    g = golden("upo_128.npz")
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    rin = 3.0 * 127 * (8.0 * nu * 0.05 / (128 * 128)) + 1.0
    state, done = None, 0
    for j, i in enumerate(g["snap_index"][:4]):                   # t = 100, 200, 500, 1000
        n = 100 * int(i)
        state = oracle.upo_steps(128, 128, s2, rin, 1.0, n - done, state=state)
        done = n
        _, m0, m1 = state
        assert relerr(m0, g["rho"][..., j]) < 1e-12, n                 # observed 5e-14 .. 1.5e-13
        assert np.abs(m1[..., 0] - g["ux"][..., j]).max() < 1e-12, n
        assert np.abs(m1[..., 1] - g["uy"][..., j]).max() < 1e-12, n
