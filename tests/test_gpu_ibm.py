"""GPU parity tests for the immersed-boundary path (BASELINE config 5) through the C ABI.
src/ibm.cpp and test/cylinder_test.cpp need toml++ and cannot be compiled here: the oracle of
this path is "parity unpinned" (restatement only; its solver:: sub-steps are pinned).  The
spread is a gather in marker order, so the kernels are compared BITWISE with the oracle; the
north-star tolerance (1e-8 relative, stated for atomic-order effects) is asserted as well."""
import ctypes as ct

import numpy as np
import pytest
from conftest import relerr

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
from gpu_util import bits_equal, dev, download_aos, ulp_diff, upload_soa  # noqa: E402
from pylbm import _ptr  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


def circle(cx, cy, radius, spacing=1.0):
    n = int(round(2 * np.pi * radius / spacing))
    t = 2 * np.pi * np.arange(n) / n
    return cx + radius * np.cos(t), cy + radius * np.sin(t)


def test_eulerian_force_density_vs_oracle(lib, oracle):
    X, Y = 80, 70
    x, y = circle(40.3, 35.6, 12.0)
    rng = np.random.default_rng(4)
    rr, cc = np.meshgrid(np.arange(X), np.arange(Y), indexing="ij")
    u = np.stack([0.05 + 0.01 * np.sin(rr / 7.0), 0.02 * np.cos(cc / 5.0)], axis=-1)
    rho = 1 + 0.02 * np.sin((rr + cc) / 9.0) + 1e-3 * rng.standard_normal((X, Y))
    ib = pylbm.Ibm(lib, x, y, X, Y)
    assert ib.roi() == oracle.ibm_roi(x, y)
    r0, r1, c0, c1 = ib.roi()
    ud, rd = upload_soa(lib, u), upload_soa(lib, rho)
    F = torch.empty((2, r1 - r0, c1 - c0), dtype=torch.float64, device=dev())
    lib.ibm_force(ib.h, _ptr(ud), _ptr(rd), _ptr(F), None)
    got = download_aos(lib, F)
    want = oracle.ibm_force(x, y, u, rho)
    assert np.abs(want).max() > 1e-3  # the boundary does act on the flow
    assert bits_equal(got, want), ulp_diff(got, want)
    fs = ib.surface_force()
    assert np.allclose(fs, want.reshape(-1, 2).sum(0), rtol=1e-12, atol=1e-15)
    ib.close()


def cylinder_solver(lib, X, Y, omega, u_in, x, y):
    bc = pylbm.Bc.periodic()
    bc.row_lo = bc.row_hi = pylbm.EDGE_ABB_VELOCITY      # cylinder_test.cpp:135-154
    bc.col_lo = bc.col_hi = pylbm.EDGE_SPECULAR          # :157-163
    bc.uw_r, bc.uw_c = u_in, 0.0
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 0, 1), bc=bc)
    ib = pylbm.Ibm(lib, x, y, X, Y)
    sv.attach_ibm(ib)
    return sv, ib


@pytest.mark.parametrize("X,Y,radius", [(120, 90, 10.0), (96, 51, 6.5)])
def test_cylinder_driver_steps_vs_oracle(lib, oracle, X, Y, radius):
    """test/cylinder_test.cpp:85-164 on a small lattice: IBM forcing + Guo source + anti-bounce-
    back inlet/outlet + specular walls.  (Y = 51: generic kernel; Y = 90: fast kernel + edge pass.)"""
    omega, u_in = 1.0 / 0.55, 0.05
    x, y = circle(X / 4.0 + 0.37, Y / 2.0 + 0.21, radius)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))   # :85
    sv, ib = cylinder_solver(lib, X, Y, omega, u_in, x, y)
    sv.set_f(f0)
    done = 0
    for n in (1, 2, 5, 30):
        sv.step(n - done, record_moments=True)
        done = n
        f = sv.get_f()
        rho, u = sv.moments()
        fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, n)
        assert relerr(f, fo) < 1e-8 and relerr(u, uo) < 1e-8          # north-star tolerance
        assert bits_equal(f, fo), (n, ulp_diff(f, fo))                 # what we actually reach
        assert bits_equal(rho, rhoo) and bits_equal(u, uo)
        assert np.allclose(ib.surface_force(), Fso, rtol=1e-11, atol=1e-16)
    sv.close(); ib.close()


def test_cylinder_config5_scale_vs_oracle(lib, oracle):
    """A slab-sized piece of config 5 (2048 x 1024, cylinder diameter 300 like parameters.toml's
    l = 300): 4 steps, bitwise vs the oracle."""
    X, Y, omega, u_in = 2048, 1024, 1.0 / 0.55, 0.04
    x, y = circle(X / 4.0, Y / 2.0, 150.0)
    assert len(x) > 900
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    sv, ib = cylinder_solver(lib, X, Y, omega, u_in, x, y)
    sv.set_f(f0)
    sv.step(4, record_moments=True)
    f = sv.get_f()
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, 4)
    assert bits_equal(f, fo), ulp_diff(f, fo)
    assert np.allclose(ib.surface_force(), Fso, rtol=1e-10)
    assert Fso[0] < 0  # the cylinder is dragged downstream: the fluid feels -drag
    sv.close(); ib.close()


@pytest.mark.parametrize("owner", [0, 1])
def test_cylinder_two_slabs_equal_single_block(lib, oracle, owner):
    """Config 5 over slabs, emulated on one GPU: a chain of 2 slabs (anti-bounce-back inlet on
    slab 0, outlet on slab 1, HALO seam, specular side walls); the immersed boundary is created in
    slab-local coordinates on the slab that owns its ROI.  Equals the oracle's single block bitwise."""
    from pylbm.slab import TO_NEXT, TO_PREV
    X, Y, n = 128, 96, 9
    R = X // 2
    omega, u_in = 1.0 / 0.55, 0.05
    x, y = circle(32.37 + owner * R, 48.21, 10.0)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, n)
    d = dev()
    prm = pylbm.BgkParams(omega, 0, 1)
    a, b = 1.0 / 3.0, 1.0 / 9.0

    def mkbc(lo, hi):
        bc = pylbm.Bc(row_lo=lo, row_hi=hi, col_lo=pylbm.EDGE_SPECULAR, col_hi=pylbm.EDGE_SPECULAR)
        bc.uw_r = u_in
        return bc
    ABB, HALO = pylbm.EDGE_ABB_VELOCITY, pylbm.EDGE_HALO
    flat, bc_flat = pylbm.Geom(X, Y, 0), mkbc(ABB, ABB)
    # first iteration on the whole block: collide, IBM force, Guo source on the ROI
    ib_flat = pylbm.Ibm(lib, x, y, X, Y)
    f0d = upload_soa(lib, f0)
    p0 = torch.empty((9, X, Y), dtype=torch.float64, device=d)
    rho = torch.empty((X, Y), dtype=torch.float64, device=d)
    u = torch.empty((2, X, Y), dtype=torch.float64, device=d)
    lib.bgk_collide(_ptr(p0), _ptr(f0d), ct.byref(flat), ct.byref(bc_flat), ct.byref(prm), _ptr(rho), _ptr(u), None)
    lib.ibm_force(ib_flat.h, _ptr(u), _ptr(rho), None, None)
    lib.ibm_add_source(ib_flat.h, _ptr(p0), ct.byref(flat), _ptr(u), ct.c_double(omega), ct.c_double(a), ct.c_double(b), None)
    torch.cuda.synchronize()
    # slabs: ghost = 1; the cylinder (rows ~20..45) belongs to slab 0
    geom = pylbm.Geom(R, Y, 1)
    bcs = [mkbc(ABB, HALO), mkbc(HALO, ABB)]
    ib0 = pylbm.Ibm(lib, x, y, R, Y, row_offset=owner * R)   # global marker coordinates, slab-local ROI
    assert 1 <= ib0.roi()[0] and ib0.roi()[1] <= R - 1
    lat = [[torch.zeros((9, R + 2, Y), dtype=torch.float64, device=d) for _ in range(2)] for _ in range(2)]
    mom = [(torch.empty((R, Y), dtype=torch.float64, device=d), torch.empty((2, R, Y), dtype=torch.float64, device=d)) for _ in range(2)]

    def halo(cur):
        for q in TO_NEXT:
            lat[1][cur][q, 0] = lat[0][cur][q, R]
        for q in TO_PREV:
            lat[0][cur][q, R + 1] = lat[1][cur][q, 1]
    for s in range(2):
        lat[s][0][:, 1:R + 1] = p0[:, s * R:(s + 1) * R]
    halo(0)
    cur = 0
    for _ in range(n - 1):
        for s in range(2):
            src, dst = lat[s][cur], lat[s][cur ^ 1]
            rs, us = mom[s]
            for r0, r1 in ((0, 1), (R - 1, R), (1, R - 1)):
                lib.bgk_stream_collide(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bcs[s]), ct.byref(prm),
                                       r0, r1, _ptr(rs), _ptr(us), None)
            if s == owner:
                lib.ibm_force(ib0.h, _ptr(us), _ptr(rs), None, None)
                lib.ibm_add_source(ib0.h, _ptr(dst), ct.byref(geom), _ptr(us), ct.c_double(omega),
                                   ct.c_double(a), ct.c_double(b), None)
        torch.cuda.synchronize()
        cur ^= 1
        halo(cur)
    P = torch.cat([lat[0][cur][:, 1:R + 1], lat[1][cur][:, 1:R + 1]], dim=1).contiguous()
    out = torch.empty_like(P)
    lib.stream(_ptr(out), _ptr(P), ct.byref(flat), ct.byref(bc_flat), None)
    got = download_aos(lib, out)
    assert bits_equal(got, fo), ulp_diff(got, fo)
    assert np.allclose(ib0.surface_force(), Fso, rtol=1e-11, atol=1e-16)
    ib0.close(); ib_flat.close()


def test_moving_boundary_extension(lib, oracle):
    """BASELINE config 5 says "moving boundary"; the reference's boundary is stationary (Q10).  The
    extension lbm_ibm_set_velocity gives the markers a uniform velocity U_b (f_j = 2 rho_j (U_b - u_j)):
    (0, 0) must reproduce the stationary run bit for bit; with U_b = the inflow velocity the
    boundary no longer resists the stream (surface force ~ 0); with U_b against it the drag grows."""
    X, Y, omega, u_in = 160, 128, 1.0 / 0.6, 0.04
    t = 2 * np.pi * np.arange(60) / 60
    x, y = X / 3.0 + 10 * np.cos(t), Y / 2.0 + 10 * np.sin(t)
    bc = pylbm.Bc.periodic()
    bc.row_lo = bc.row_hi = pylbm.EDGE_ABB_VELOCITY
    bc.col_lo = bc.col_hi = pylbm.EDGE_SPECULAR
    bc.uw_r = u_in
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))

    def run(Ub):
        sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 0, 1), bc=bc)
        ib = pylbm.Ibm(lib, x, y, X, Y)
        if Ub is not None:
            lib.ibm_set_velocity(ib.h, ct.c_double(Ub[0]), ct.c_double(Ub[1]))
        sv.attach_ibm(ib)
        sv.set_f(f0)
        sv.step(60)
        f, Fs = sv.get_f(), ib.surface_force()
        sv.close(); ib.close()
        return f, Fs

    f_ref, Fs_ref = run(None)
    f_zero, Fs_zero = run((0.0, 0.0))
    assert bits_equal(f_zero, f_ref) and np.array_equal(Fs_zero, Fs_ref)
    _, Fs_co = run((u_in, 0.0))
    _, Fs_counter = run((-u_in, 0.0))
    assert Fs_ref[0] < 0                                   # F acts on the fluid: the stationary cylinder resists the +r stream
    assert abs(Fs_co[0]) < 0.05 * abs(Fs_ref[0])           # co-moving markers: almost no force
    assert Fs_counter[0] < 1.5 * Fs_ref[0]                 # counter-moving: larger resistance (more negative)


@pytest.mark.parametrize("cx_frac", [0.5, 0.12])
def test_forced_band_blocks_equal_single_steps(lib, oracle, cx_frac):
    """lbm_solver_step with an immersed boundary advances D steps per block (forced band around the ROI
    on a shrinking trapezoid + D-step window for the rows farther away; tuning ibm_depth = 5 default, 3)
    or one step per launch (ibm_depth = 1): same bits after 13 steps (two blocks + singles), and equal
    to the oracle.  cx_frac = 0.12 puts the ROI too close to the inlet for a block: silent fallback."""
    X, Y, omega, u_in, radius = 160, 128, 1.0 / 0.55, 0.05, 8.0
    x, y = circle(X * cx_frac + 0.3, Y / 2.0 - 0.4, radius)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    res = {}
    try:
        for depth in (1, 5, 3):
            lib.set_tuning(b"ibm_depth", depth)
            sv, ib = cylinder_solver(lib, X, Y, omega, u_in, x, y)
            sv.set_f(f0)
            sv.step(13, record_moments=False)
            res[depth] = (sv.get_f(), ib.surface_force())
            sv.close(); ib.close()
    finally:
        lib.set_tuning(b"ibm_depth", -1)
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, 13)
    for depth, (f, Fs) in res.items():
        assert bits_equal(f, fo), (depth, ulp_diff(f, fo))
        assert np.allclose(Fs, Fso, rtol=1e-11, atol=1e-16)


def test_cylinder_with_reassociated_delta_form(lib, oracle):
    """Opt-in (tuning bgk_fast_delta = 1): the cylinder preset's delta-form collision through the
    reassociated model -- not bitwise any more, 1e-10 relative on f after 13 steps (north star: 1e-8)."""
    X, Y, omega, u_in, radius = 160, 128, 1.0 / 0.55, 0.05, 8.0
    x, y = circle(X * 0.5 + 0.3, Y / 2.0 - 0.4, radius)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, 13)
    try:
        lib.set_tuning(b"bgk_fast_delta", 1)
        lib.set_tuning(b"bgk_fast", 1)
        sv, ib = cylinder_solver(lib, X, Y, omega, u_in, x, y)
        sv.set_f(f0)
        sv.step(13, record_moments=False)
        f, Fs = sv.get_f(), ib.surface_force()
        sv.close(); ib.close()
    finally:
        lib.set_tuning(b"bgk_fast_delta", -1)
        lib.set_tuning(b"bgk_fast", -1)
    assert relerr(f, fo) < 1e-10, relerr(f, fo)
    assert np.allclose(Fs, Fso, rtol=1e-8, atol=1e-14)
