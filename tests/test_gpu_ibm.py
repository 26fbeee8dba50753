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


def cylinder_solver(lib, X, Y, omega, u_in, x, y, form=pylbm.FORM_REFERENCE_ORDER):
    """the cylinder preset; the parity tests hold it to the oracle BITWISE, i.e. in the reference's operation order
    (the library's default is the reassociated collision: test_cylinder_with_the_default_collision)"""
    bc = pylbm.Bc.periodic()
    bc.row_lo = bc.row_hi = pylbm.EDGE_ABB_VELOCITY      # cylinder_test.cpp:135-154
    bc.col_lo = bc.col_hi = pylbm.EDGE_SPECULAR          # :157-163
    bc.uw_r, bc.uw_c = u_in, 0.0
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 0, 1, form=form), bc=bc)
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
    prm = pylbm.BgkParams(omega, 0, 1, form=pylbm.FORM_REFERENCE_ORDER)
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
        sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, pylbm.BgkParams(omega, 0, 1, form=pylbm.FORM_REFERENCE_ORDER), bc=bc)
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


@pytest.mark.parametrize("tune,cy_frac", [
    ({"ibm_box": 0}, 0.55),                          # full-width band (round 2, first half)
    ({"ibm_box": 1, "ibm_box_overlap": 0}, 0.55),    # forced box, window launch behind the chain on one stream
    ({"ibm_box": 1, "ibm_step_chain": 1}, 0.55),     # box; forcing as the 10-launch chain
    ({"ibm_box": 1, "ibm_step_opt": 0, "ibm_step_split": 0}, 0.55),  # box; the one-workgroup kernel of round 1
    ({"ibm_box": 1, "bg_priority": 1}, 0.55),
    ({"ibm_box": 1, "ibm_chain_kernel": 1}, 0.55),   # box; the chain as ONE launch of 16 workgroups with grid barriers
    ({"ibm_box": 1, "ibm_chain_kernel": 1, "ibm_chain_wgs": 5}, 0.55),
    ({"ibm_box": 1, "ibm_chain_kernel": 1, "ibm_chain_wgs": 40, "ibm_box_overlap": 0}, 0.55),
    ({"ibm_box": 1}, 0.12),                          # box would touch the wall columns: the band takes over
    ({"ibm_box": 1}, 0.87),
])
def test_forced_box_variants_equal_the_oracle(lib, oracle, tune, cy_frac):
    """The forced BOX of an immersed-boundary block (rows and columns ROI +- 2 D on a small lattice of its own,
    D-step window over all rows beside it) and every switch around it: same bits as the oracle after 16 steps
    (1 + three 5-step blocks), same surface force; the blocks really ran (lbm_solver_block_launches)."""
    if tune.get("ibm_chain_kernel") and not lib.raw.lbm_build_has_experiments():
        pytest.skip("the one-launch forcing chain is an experiment: make -C lattice-boltzmann-method_amd/csrc EXPERIMENTS=1")
    X, Y, omega, u_in, radius = 176, 200, 1.0 / 0.55, 0.05, 9.0
    x, y = circle(X * 0.45 + 0.3, Y * cy_frac - 0.4, radius)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, 16)
    try:
        for k, v in tune.items():
            lib.set_tuning(k.encode(), v)
        sv, ib = cylinder_solver(lib, X, Y, omega, u_in, x, y)
        sv.set_f(f0)
        sv.step(16, record_moments=False)
        f, Fs = sv.get_f(), ib.surface_force()
        blocks = int(lib.raw.lbm_solver_block_launches(sv.h))
        sv.close(); ib.close()
    finally:
        for k in tune:
            lib.set_tuning(k.encode(), -1)
    assert blocks == 3, blocks
    assert bits_equal(f, fo), ulp_diff(f, fo)
    assert np.allclose(Fs, Fso, rtol=1e-11, atol=1e-16)


def test_cylinder_with_the_default_collision(lib, oracle):
    """LBM_FORM_DEFAULT (round 4: one rule -- the reassociated collision everywhere, the delta form included): the
    cylinder preset is not bitwise then, 1e-10 relative on f after 13 steps (north star: 1e-8); and DEFAULT and
    LBM_FORM_REASSOCIATED are the same bits."""
    X, Y, omega, u_in, radius = 160, 128, 1.0 / 0.55, 0.05, 8.0
    x, y = circle(X * 0.5 + 0.3, Y / 2.0 - 0.4, radius)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    fo, uo, rhoo, Fso = oracle.cylinder_steps(x, y, f0, omega, u_in, 13)
    res = {}
    for form in (pylbm.FORM_DEFAULT, pylbm.FORM_REASSOCIATED):
        sv, ib = cylinder_solver(lib, X, Y, omega, u_in, x, y, form=form)
        sv.set_f(f0)
        sv.step(13, record_moments=False)
        res[form] = (sv.get_f(), ib.surface_force())
        sv.close(); ib.close()
    f, Fs = res[pylbm.FORM_DEFAULT]
    assert relerr(f, fo) < 1e-10, relerr(f, fo)
    assert not bits_equal(f, fo)                       # (it really is the other operation order)
    assert np.allclose(Fs, Fso, rtol=1e-8, atol=1e-14)
    assert bits_equal(f, res[pylbm.FORM_REASSOCIATED][0])


# ---- config 5 over slabs in blocks of D steps, the boundary anywhere -- also across a seam -------------
def _slab_block_run(lib, oracle, X, Y, n_slabs, cx, radius, D, n_blocks, start_from_pre=False):
    """the run through lbm_slab_ibm_* on n_slabs emulated slabs (one GPU, messages moved by device copies)
    and the same run on ONE block through the solver context; returns (P_slabs, P_block, Fs_slabs, Fs_block,
    slab objects' roles)"""
    R = X // n_slabs
    omega, u_in = 1.0 / 0.55, 0.05
    x, y = circle(cx, Y / 2 + 0.21, radius)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    prm = pylbm.BgkParams(omega, 0, 1, form=pylbm.FORM_REFERENCE_ORDER)
    bc = pylbm.Bc(row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY, col_lo=pylbm.EDGE_SPECULAR,
                  col_hi=pylbm.EDGE_SPECULAR, uw_r=u_in)
    n = 1 + D * n_blocks
    # one block through the solver context (collide-first + forced blocks, == single steps == oracle elsewhere)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, prm, bc=bc)
    ibw = pylbm.Ibm(lib, x, y, X, Y)
    sv.attach_ibm(ibw)
    sv.set_f(f0)
    sv.step(n)
    want, Fw = sv.get_f(), ibw.surface_force()
    sv.close(); ibw.close()
    # first iteration on the whole block: collide, forcing, source (cylinder_test.cpp:103-127)
    d = dev()
    flat = pylbm.Geom(X, Y, 0)
    ib_flat = pylbm.Ibm(lib, x, y, X, Y)
    f0d = upload_soa(lib, f0)
    p0 = torch.empty((9, X, Y), dtype=torch.float64, device=d)
    rho = torch.empty((X, Y), dtype=torch.float64, device=d)
    u = torch.empty((2, X, Y), dtype=torch.float64, device=d)
    lib.bgk_collide(_ptr(p0), _ptr(f0d), ct.byref(flat), ct.byref(bc), ct.byref(prm), _ptr(rho), _ptr(u), None)
    lib.ibm_force(ib_flat.h, _ptr(u), _ptr(rho), None, None)
    lib.ibm_add_source(ib_flat.h, _ptr(p0), ct.byref(flat), _ptr(u), ct.c_double(omega), ct.c_double(1 / 3), ct.c_double(1 / 9), None)
    torch.cuda.synchronize()
    ib_flat.close()
    geom = pylbm.Geom(R, Y, D)
    slabs = [pylbm.SlabIbm(lib, geom, s * R, X, bc, prm, D, x, y) for s in range(n_slabs)]
    lat = [[torch.zeros((9, R + 2 * D, Y), dtype=torch.float64, device=d) for _ in range(2)] for _ in range(n_slabs)]
    for s in range(n_slabs):
        lat[s][0][:, D:R + D] = p0[:, s * R:(s + 1) * R]
    z = lambda k: torch.zeros(max(int(k), 1), dtype=torch.float64, device=d)

    def exchange(pack, finish, counts):
        """pack on every slab, move next->prev / prev->next, finish on every slab"""
        bufs = []
        for s, sl in enumerate(slabs):
            c = counts(sl)
            b = dict(sp=z(c[0][0]), rp=z(c[0][1]), sn=z(c[1][0]), rn=z(c[1][1]))
            bufs.append(b)
            pack(s, sl, b)
        torch.cuda.synchronize()
        for s in range(n_slabs - 1):
            assert bufs[s]["sn"].numel() == bufs[s + 1]["rp"].numel() and bufs[s + 1]["sp"].numel() == bufs[s]["rn"].numel()
            bufs[s + 1]["rp"].copy_(bufs[s]["sn"])
            bufs[s]["rn"].copy_(bufs[s + 1]["sp"])
        torch.cuda.synchronize()
        for s, sl in enumerate(slabs):
            finish(s, sl, bufs[s])
        torch.cuda.synchronize()

    cur = 0
    if start_from_pre:   # the first iteration over the slabs themselves, from the PRE-collision state
        pre = [torch.zeros((9, R + 2 * D, Y), dtype=torch.float64, device=d) for _ in range(n_slabs)]
        for s in range(n_slabs):
            pre[s][:, D:R + D] = f0d[:, s * R:(s + 1) * R]
            lat[s][0].zero_()
        exchange(lambda s, sl, b: lib.slab_ibm_prime_pack(sl.h, _ptr(pre[s]), _ptr(b["sp"]), _ptr(b["sn"]), None),
                 lambda s, sl, b: lib.slab_ibm_start_finish(sl.h, _ptr(lat[s][0]), _ptr(pre[s]), _ptr(b["rp"]), _ptr(b["rn"]), None),
                 lambda sl: (sl.prime_counts(0), sl.prime_counts(1)))
        for s in range(n_slabs):   # == the whole-block first iteration, ghost rows aside
            assert torch.equal(lat[s][0][:, D:R + D], p0[:, s * R:(s + 1) * R]), s
    else:
        exchange(lambda s, sl, b: lib.slab_ibm_prime_pack(sl.h, _ptr(lat[s][cur]), _ptr(b["sp"]), _ptr(b["sn"]), None),
                 lambda s, sl, b: lib.slab_ibm_prime_finish(sl.h, _ptr(lat[s][cur]), _ptr(b["rp"]), _ptr(b["rn"]), None),
                 lambda sl: (sl.prime_counts(0), sl.prime_counts(1)))
    for _ in range(n_blocks):
        m = lambda sl: ((sl.msg_doubles,) * 2, (sl.msg_doubles,) * 2)
        exchange(lambda s, sl, b: lib.slab_ibm_block_compute(sl.h, _ptr(lat[s][cur ^ 1]), _ptr(lat[s][cur]), _ptr(b["sp"]), _ptr(b["sn"]), None),
                 lambda s, sl, b: lib.slab_ibm_block_finish(sl.h, _ptr(lat[s][cur ^ 1]), _ptr(b["rp"]), _ptr(b["rn"]), None), m)
        cur ^= 1
    P = torch.cat([lat[s][cur][:, D:R + D] for s in range(n_slabs)], dim=1).contiguous()
    out = torch.empty_like(P)
    lib.stream(_ptr(out), _ptr(P), ct.byref(flat), ct.byref(bc), None)
    got = download_aos(lib, out)
    owners = [s for s, sl in enumerate(slabs) if sl.owner]
    Fs = [slabs[s].surface_force() for s in owners]
    roles = [(sl.owner, sl.straddle_prev, sl.straddle_next) for sl in slabs]
    for sl in slabs:
        sl.close()
    return got, want, Fs, Fw, roles


@pytest.mark.parametrize("case", ["on_the_seam", "inside_slab0", "band_into_ghost_rows", "three_slabs_seam_1_2",
                                  "off_centre_on_seam", "depth3_on_seam"])
def test_cylinder_blocks_over_slabs_equal_single_block(lib, oracle, case):
    """the cylinder centred ON a seam (the BASELINE geometry: centre at rows/4 of 8 slabs),
    in blocks of D = 5 steps: both co-owners run the band chain, swap the band's outer rows, nothing else
    changes -- populations and surface force equal the single block bit for bit.  Also the boundary wholly
    inside one slab, the band reaching into (but its valid rows not across) the ghost rows, three slabs
    with a non-owner, an off-centre straddle and D = 3."""
    X, Y, n_slabs, radius, D, nb = 256, 96, 2, 10.0, 5, 3
    if case == "on_the_seam":
        cx, want_roles = 128.3, [(1, 0, 1), (1, 1, 0)]
    elif case == "inside_slab0":
        cx, want_roles = 60.4, [(1, 0, 0), (0, 0, 0)]
    elif case == "band_into_ghost_rows":
        cx, want_roles = 107.4, [(1, 0, 0), (0, 0, 0)]
    elif case == "three_slabs_seam_1_2":
        X, n_slabs, cx, want_roles = 384, 3, 255.6, [(0, 0, 0), (1, 0, 1), (1, 1, 0)]
    elif case == "off_centre_on_seam":
        cx, want_roles = 137.8, [(1, 0, 1), (1, 1, 0)]
    else:
        cx, D, nb, want_roles = 128.3, 3, 4, [(1, 0, 1), (1, 1, 0)]
    got, want, Fs, Fw, roles = _slab_block_run(lib, oracle, X, Y, n_slabs, cx, radius, D, nb,
                                               start_from_pre=case in ("on_the_seam", "three_slabs_seam_1_2", "band_into_ghost_rows"))
    assert [tuple(int(v) for v in r) for r in roles] == want_roles, roles
    assert bits_equal(got, want), (case, ulp_diff(got, want))
    for F in Fs:                      # every co-owner holds the same forcing, bit for bit
        assert np.array_equal(F, Fw), (case, F, Fw)


def test_ring_ibm_block_wrappers_on_a_self_ring(lib, oracle):
    """lbm_ring_ibm_start / lbm_ring_bgk_block_ibm (RCCL transport of the slab blocks) on hardware: one rank whose
    two neighbours are itself (periodic self ring), playing the MIDDLE slab of a 3-slab domain that owns the
    cylinder.  The messages it receives are its own -- physically meaningless, but the same bytes a plain device
    copy of its send buffers delivers: the RCCL run must equal that emulation bit for bit (message sizes, order of
    the sends / receives, stream ordering of compute -> exchange -> finish)."""
    X, Y, R, D = 384, 96, 128, 5
    omega, u_in = 1.0 / 0.55, 0.05
    x, y = circle(R + 64.3, Y / 2 + 0.21, 10.0)
    prm = pylbm.BgkParams(omega, 0, 1, form=pylbm.FORM_REFERENCE_ORDER)
    bc = pylbm.Bc(row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY, col_lo=pylbm.EDGE_SPECULAR,
                  col_hi=pylbm.EDGE_SPECULAR, uw_r=u_in)
    u0 = np.zeros((R, Y, 2)); u0[..., 0] = u_in
    f0 = upload_soa(lib, oracle.incomp_equilibrium(u0, np.ones((R, Y))))
    d = dev()
    geom = pylbm.Geom(R, Y, D)
    n_msg = 9 * D * Y

    def fresh():
        sl = pylbm.SlabIbm(lib, geom, R, X, bc, prm, D, x, y)
        assert (sl.owner, sl.straddle_prev, sl.straddle_next) == (1, 0, 0)
        pre = torch.zeros((9, R + 2 * D, Y), dtype=torch.float64, device=d)
        pre[:, D:R + D] = f0
        lat = [torch.zeros((9, R + 2 * D, Y), dtype=torch.float64, device=d) for _ in range(2)]
        return sl, pre, lat

    # emulation: what a self ring delivers is this slab's own send buffers
    sl, pre, lat = fresh()
    sp, sn = (torch.zeros(n_msg, dtype=torch.float64, device=d) for _ in range(2))
    lib.slab_ibm_prime_pack(sl.h, _ptr(pre), _ptr(sp), _ptr(sn), None)
    torch.cuda.synchronize()
    rp, rn = sn.clone(), sp.clone()       # from prev (= self): its send_next; from next: its send_prev
    lib.slab_ibm_start_finish(sl.h, _ptr(lat[0]), _ptr(pre), _ptr(rp), _ptr(rn), None)
    cur = 0
    for _ in range(3):
        lib.slab_ibm_block_compute(sl.h, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), _ptr(sp), _ptr(sn), None)
        torch.cuda.synchronize()
        rp, rn = sn.clone(), sp.clone()
        lib.slab_ibm_block_finish(sl.h, _ptr(lat[cur ^ 1]), _ptr(rp), _ptr(rn), None)
        cur ^= 1
    torch.cuda.synchronize()
    want, Fw = lat[cur].clone(), sl.surface_force()
    sl.close()
    assert bool(torch.isfinite(want).all())
    # the same through RCCL
    sl, pre, lat = fresh()
    ident = (ct.c_ubyte * 128)()
    lib.ring_unique_id(ident)
    ring = ct.c_void_p()
    lib.ring_create(ct.byref(ring), ident, 0, 1, ct.byref(geom), 1)
    try:
        lib.ring_ibm_start(ring, sl.h, _ptr(lat[0]), _ptr(pre), None)
        cur = 0
        for _ in range(3):
            lib.ring_bgk_block_ibm(ring, sl.h, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), 16, None)
            cur ^= 1
        torch.cuda.synchronize()
        assert torch.equal(lat[cur], want)
        assert np.array_equal(sl.surface_force(), Fw)
    finally:
        lib.ring_destroy(ring)
        sl.close()


def test_uniform_field_known_answer_on_the_gpu(lib, oracle):
    """Known answer of the method itself (not of any restatement): Peskin's kernel is a partition of unity, so in
    a uniform field one forcing iteration puts exactly -2 rho0 u0 per marker onto the lattice (ibm.cpp:176-182);
    both forms of the forcing (launch chain: lbm_ibm_force; one workgroup: lbm_ibm_step) and lbm_ibm_surface_force."""
    X, Y = 96, 88
    x, y = circle(47.3, 41.6, 14.0)
    u0, rho0 = np.array([0.043, -0.017]), 1.013
    u = np.broadcast_to(u0, (X, Y, 2)).copy()
    rho = np.full((X, Y), rho0)
    ib = pylbm.Ibm(lib, x, y, X, Y, m_max=2)
    r0, r1, c0, c1 = ib.roi()
    ud, rd = upload_soa(lib, u), upload_soa(lib, rho)
    F = torch.empty((2, r1 - r0, c1 - c0), dtype=torch.float64, device=dev())
    lib.ibm_force(ib.h, _ptr(ud), _ptr(rd), _ptr(F), None)
    want = -2.0 * rho0 * u0 * len(x)
    got = F.sum(dim=(1, 2)).cpu().numpy()
    assert np.allclose(got, want, rtol=1e-12, atol=0), (got, want)
    assert np.allclose(ib.surface_force(), want, rtol=1e-12, atol=0)
    p = torch.zeros((9, X, Y), dtype=torch.float64, device=dev())
    g = pylbm.Geom(X, Y, 0)
    lib.ibm_step(ib.h, _ptr(p), ct.byref(g), _ptr(ud), _ptr(rd), ct.c_double(1.3), ct.c_double(3.0), ct.c_double(9.0), None)
    assert np.allclose(ib.surface_force(), want, rtol=1e-12, atol=0)
    # Guo's source with a = 1 / cs^2 = 3, b = 1 / cs^4 = 9 (cylinder_test.cpp:116-127): sum_q S_q = 0 and
    # sum_q c_q S_q = (1 - omega / 2) F at every node, hence over the lattice
    cx = torch.tensor([0, 1, 0, -1, 0, 1, -1, -1, 1], dtype=torch.float64, device=dev())
    cy = torch.tensor([0, 0, 1, 0, -1, 1, 1, -1, -1], dtype=torch.float64, device=dev())
    mom = torch.stack([(p * cx[:, None, None]).sum(), (p * cy[:, None, None]).sum()]).cpu().numpy()
    assert abs(float(p.sum())) < 1e-12 * abs(want).max()          # no mass
    assert np.allclose(mom, (1 - 0.5 * 1.3) * want, rtol=1e-10), (mom, (1 - 0.5 * 1.3) * want)
    ib.close()
