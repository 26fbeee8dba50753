"""GPU parity tests for the KBC central-moment collision (src/ulbm.cpp; BASELINE config 3),
through the C ABI.  Bars: bitwise vs the oracle when both start from the same moments;
<= 1e-11 relative (north star: "stated tolerance for MRT") vs golden vectors of the unmodified
reference for <= 100 steps, 1e-9 up to 500 steps of the (chaotic) shear layer."""
import ctypes as ct

import numpy as np
import pytest
from conftest import golden, relerr

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
from gpu_util import bits_equal, dev, download_aos, ulp_diff, upload_soa  # noqa: E402
from pylbm import _ptr  # noqa: E402

S2 = 1.0 / (0.5 + 3.0 * 1.70766666e-4)  # ulbm_double_shear_flow.cpp:75-76


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


@pytest.fixture(params=["reference_order", "fast"])
def mode(request, lib):
    """reference_order: kbc.hpp KbcModel, -ffp-contract=off: BITWISE vs the oracle.
    fast (the default): KbcFastModel, same mathematics reassociated (raw-moment butterfly, one
    back-transform per part, product-form equilibrium, reciprocals, FMA): stated tolerance 1e-12
    relative (L2 over the field) for <= 20 steps, 1e-10 for <= 500 steps."""
    lib.set_tuning(b"kbc_fast", 1 if request.param == "fast" else 0)
    yield request.param
    lib.set_tuning(b"kbc_fast", -1)


def check(got, want, mode, what, tol=1e-12):
    if mode == "reference_order":
        assert bits_equal(got, want), (what, ulp_diff(got, want))
    else:
        assert relerr(got, want) < tol, (what, relerr(got, want))


def test_kbc_collide_unit_vs_reference_and_oracle(lib, oracle):
    g = golden("kbc_units.npz")
    f, m0, m1 = g["f_in"], g["m0_in"], g["m1_in"]
    R, C = m0.shape
    out = torch.empty((9, R, C), dtype=torch.float64, device=dev())
    prm = pylbm.KbcParams(float(g["s2"]))
    fd, m0d, m1d = upload_soa(lib, f), upload_soa(lib, m0), upload_soa(lib, m1)  # keep alive
    lib.kbc_collide_given_moments(_ptr(out), _ptr(fd), _ptr(m0d), _ptr(m1d), ct.byref(prm), R, C, None)
    got = download_aos(lib, out)
    want, _ = oracle.kbc_collide(f, m0, m1, float(g["s2"]))
    assert bits_equal(got, want), ulp_diff(got, want)
    assert relerr(got, g["coll1"]) < 1e-13  # kbc::collide of the unmodified reference


def test_kbc_driver_initialisation(lib):
    """eval_equilibrium as the driver calls it: ux2 = uy2 = 0 left by the ctor."""
    g = golden("kbc_units.npz")
    m0, m1 = g["shear_m0"], g["shear_m1"]
    R, C = m0.shape
    out = torch.empty((9, R, C), dtype=torch.float64, device=dev())
    m0d, m1d = upload_soa(lib, m0), upload_soa(lib, m1)  # keep alive
    lib.kbc_equilibrium(_ptr(out), _ptr(m0d), _ptr(m1d), R, C, 1, None)
    assert relerr(download_aos(lib, out), g["shear0_f"]) < 1e-15


def test_kbc_solver_steps_vs_golden_and_oracle(lib, oracle, mode):
    g = golden("kbc_units.npz")
    f0 = g["shear0_f"]
    R, C = g["shear_m0"].shape
    sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(S2))
    sv.set_f(f0)
    # the oracle fed with the moments OF f0 performs exactly the kernel's arithmetic
    m0 = oracle.calc_rho(f0)
    m1 = oracle.calc_u(f0, m0)
    done = 0
    for n in (1, 5, 20):
        sv.step(n - done, record_moments=True)
        done = n
        f = sv.get_f()
        fo, m0o, m1o = oracle.kbc_steps(f0, m0, m1, S2, n)
        check(f, fo, mode, n)
        assert relerr(f, g[f"shear{n}_f"]) < 1e-12, n  # reference (starts from the IC moments)
    sv.close()


def test_double_shear_main_snapshots(lib, oracle, mode):
    """test/ulbm_double_shear_flow.cpp unmodified main (128 x 128): snapshots every 10 steps."""
    try:
        g = golden("dsf_128.npz")
    except FileNotFoundError:
        pytest.skip("dsf_128.npz not generated")
    m0, m1 = oracle.kbc_shear_init(128, 128)
    R = C = 128
    fdev = torch.empty((9, R, C), dtype=torch.float64, device=dev())
    m0d, m1d = upload_soa(lib, m0), upload_soa(lib, m1)  # keep alive
    lib.kbc_equilibrium(_ptr(fdev), _ptr(m0d), _ptr(m1d), R, C, 1, None)
    sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(S2))
    lib.solver_set_f_soa_dev(sv.h, _ptr(fdev))
    t = 0
    for k, i in enumerate(g["snap_index"]):
        # snapshot i holds m0/m1 as they stand after 10*i iterations: moments of the streamed state
        n = int(i) * int(g["snapshot_period"]) - t
        sv.step(n)
        t += n
        f = sv.get_f()
        rho = oracle.calc_rho(f)
        u = oracle.calc_u(f, rho)
        tol = 1e-11 if t <= 100 else 1e-9
        assert relerr(u[..., 0], g["ux"][..., k]) < tol, t
        assert relerr(u[..., 1], g["uy"][..., k]) < tol, t
        assert relerr(rho, g["rho"][..., k]) < tol, t
    sv.close()


def test_config3_size_equals_tiled_small_box(lib, oracle, mode):
    """BASELINE config 3 size (4096 x 4096 KBC): tiled 64 x 64 state == the oracle's 64 x 64 box."""
    rng = np.random.default_rng(9)
    rho = 1 + 0.01 * rng.standard_normal((64, 64))
    u = 0.03 * rng.standard_normal((64, 64, 2))
    tile = oracle.equilibrium(u, rho)
    m0 = oracle.calc_rho(tile)
    m1 = oracle.calc_u(tile, m0)
    want, _, _ = oracle.kbc_steps(tile, m0, m1, S2, 5)
    R = C = 4096
    big = upload_soa(lib, tile).repeat(1, R // 64, C // 64).contiguous()
    sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(S2))
    lib.solver_set_f_soa_dev(sv.h, _ptr(big))
    sv.step(5)
    lib.solver_get_f_soa_dev(sv.h, _ptr(big))
    torch.cuda.synchronize()
    blocks = big.view(9, R // 64, 64, C // 64, 64)
    first = blocks[:, 0, :, 0, :].contiguous()
    check(download_aos(lib, first), want, mode, "tile")
    assert bool((blocks == first.view(9, 1, 64, 1, 64)).all())
    sv.close()


def _upo_case(H, W):
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    rin = 3.0 * (H - 1) * (8.0 * nu * 0.05 / (W * W)) + 1.0        # ulbm_poiseuille.cpp:70-83
    bc = pylbm.Bc.periodic()
    bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK                  # :126-132
    bc.pressure_rows, bc.rho_inlet, bc.rho_outlet = 1, rin, 1.0     # :36-58, :122
    return s2, rin, bc


@pytest.mark.parametrize("H,W", [(24, 20), (128, 128)])
def test_ulbm_poiseuille_preset_vs_oracle_and_reference(lib, oracle, H, W):
    """SURVEY 8(f) row 1, test/ulbm_poiseuille.cpp: KBC + pressure-periodic rows (imposed density through
    solver::incomp_equilibrium, f_equi = iequi_f.pow(-1)) + bounce-back columns, started like the
    driver from adve_f = 0 with held moments m0 = 1, m1 = 0.  Bitwise vs the oracle; vs the
    reference's own classes (tests/golden/upo_units.npz) to rounding."""
    s2, rin, bc = _upo_case(H, W)
    g = golden("upo_units.npz")
    tag, steps = ("a", (1, 2, 10, 100)) if (H, W) == (24, 20) else ("b", (50,))
    sv = pylbm.Solver(lib, pylbm.MODEL_KBC, H, W, pylbm.KbcParams(s2), bc=bc)
    sv.set_f(np.zeros((H, W, 9)))
    sv.set_moments(np.ones((H, W)), np.zeros((H, W, 2)))
    done = 0
    for n in steps:
        sv.step(n - done, record_moments=False)
        done = n
        got = sv.get_f()
        want_f, want_m0, want_m1 = oracle.upo_steps(H, W, s2, rin, 1.0, n)
        assert bits_equal(got, want_f), (n, ulp_diff(got, want_f))
        assert relerr(got, g[f"{tag}_{n}_f"]) < 1e-12, (n, relerr(got, g[f"{tag}_{n}_f"]))
    sv.close()


def test_kbc_multi_step_launches_equal_single_steps(lib, oracle):
    """lbm_kbc_stream_collide_xn (register sliding window, reassociated collision): D steps in one
    launch == D single-step launches bit for bit (same per-node arithmetic), D = 2, 3, 4; and the
    solver context, which fuses 4 steps per launch by default (21 = 5 x 4 + 1), against the oracle."""
    rng = np.random.default_rng(21)
    R, C = 96, 192
    rho = 1 + 0.01 * rng.standard_normal((R, C))
    u = 0.03 * rng.standard_normal((R, C, 2))
    f0 = oracle.equilibrium(u, rho) * (1 + 0.01 * rng.standard_normal((R, C, 9)))
    g = pylbm.Geom(R, C, 0)
    prm = pylbm.KbcParams(S2)
    p0 = upload_soa(lib, f0)
    a, b = torch.empty_like(p0), torch.empty_like(p0)
    for D in (2, 3, 4):
        src = p0.clone()
        for _ in range(D):
            lib.kbc_stream_collide(_ptr(a), _ptr(src), ct.byref(g), None, ct.byref(prm), 0, R, None, None, None)
            src, a = a, src
        lib.kbc_stream_collide_xn(_ptr(b), _ptr(p0), ct.byref(g), None, ct.byref(prm), D, 0, R, None)
        torch.cuda.synchronize()
        assert torch.equal(b, src), (D, float((b - src).abs().max()))
    sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, prm)
    sv.set_f(f0)
    sv.step(21)
    m0 = oracle.calc_rho(f0)
    want, _, _ = oracle.kbc_steps(f0, m0, oracle.calc_u(f0, m0), S2, 21)
    assert relerr(sv.get_f(), want) < 1e-12
    sv.close()


def test_native_ring_kbc_self_exchange(lib, oracle):
    """lbm_ring_kbc_step (C++ slab ring on RCCL, csrc/capi_ring.hip): one periodic rank whose
    neighbours are itself, 3-step launches and single steps; equals the single-block solver bit for
    bit (same collision, same kernels) and the oracle to the reassociated tolerance."""
    d = dev()
    R, C = 96, 256
    rng = np.random.default_rng(5)
    rho = 1 + 0.01 * rng.standard_normal((R, C))
    u = 0.03 * rng.standard_normal((R, C, 2))
    f0 = oracle.equilibrium(u, rho) * (1 + 0.01 * rng.standard_normal((R, C, 9)))
    prm = pylbm.KbcParams(S2)
    flat = pylbm.Geom(R, C, 0)
    ident = (ct.c_ubyte * 128)()
    for depth, launches, period in ((3, 3, 1), (3, 5, 2), (2, 7, 3), (1, 4, 1)):   # (the last case feeds the oracle check below)
        n = depth * launches
        # reference: the same number of single steps on one ghost-free block
        a = upload_soa(lib, f0)
        tmp = torch.empty_like(a)
        lib.kbc_collide(_ptr(tmp), _ptr(a), ct.byref(flat), None, ct.byref(prm), None, None, None)
        first = tmp.clone()
        for _ in range(n):
            lib.kbc_stream_collide(_ptr(a), _ptr(tmp), ct.byref(flat), None, ct.byref(prm), 0, R, None, None, None)
            a, tmp = tmp, a
        torch.cuda.synchronize()
        want = tmp
        G = depth * period   # period > 1: one exchange per `period` launches, the others use ghost rows up
        g = pylbm.Geom(R, C, G)
        lat = [torch.zeros((9, R + 2 * G, C), dtype=torch.float64, device=d) for _ in range(2)]
        lat[0][:, G:G + R] = first
        lat[1].fill_(float("nan"))
        ring = ct.c_void_p()
        lib.ring_unique_id(ident)
        lib.ring_create(ct.byref(ring), ident, 0, 1, ct.byref(g), 1)
        try:
            torch.cuda.synchronize()
            lib.ring_exchange(ring, _ptr(lat[0]), None)
            lib.ring_join(ring, None)
            cur = 0
            for _ in range(launches):
                lib.ring_kbc_step(ring, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), None, ct.byref(prm), depth, 16, None)
                cur ^= 1
            torch.cuda.synchronize()
            assert torch.equal(lat[cur][:, G:G + R], want), (depth, period, float((lat[cur][:, G:G + R] - want).abs().max()))
        finally:
            lib.ring_destroy(ring)
    m0 = oracle.calc_rho(f0)
    fo, _, _ = oracle.kbc_steps(f0, m0, oracle.calc_u(f0, m0), S2, 5)   # 1 collide + 4 stream-collides, streamed
    out = torch.empty((9, R, C), dtype=torch.float64, device=d)
    lib.stream(_ptr(out), _ptr(want.contiguous()), ct.byref(flat), None, None)
    assert relerr(download_aos(lib, out), fo) < 1e-12


def test_ulbm_poiseuille_vs_unmodified_main_snapshots(lib):
    """The ulbm_poiseuille preset against the snapshots of the UNMODIFIED test/ulbm_poiseuille.cpp main
    (tests/golden/upo_128.npz; 300000 iterations took ~3.5 h on one CPU core): moments after
    100 ... 100000 iterations.  Tolerance 1e-12 up to 1000 iterations, 1e-10 after (the flow is still
    accelerating; no chaotic growth)."""
    g = golden("upo_128.npz")
    s2, rin, bc = _upo_case(128, 128)
    sv = pylbm.Solver(lib, pylbm.MODEL_KBC, 128, 128, pylbm.KbcParams(s2), bc=bc)
    sv.set_f(np.zeros((128, 128, 9)))
    sv.set_moments(np.ones((128, 128)), np.zeros((128, 128, 2)))
    done = 0
    for j, i in enumerate(g["snap_index"]):
        n = 100 * int(i)
        if n > 100000:
            break
        sv.step(n - done, record_moments=False)
        done = n
        f = sv.get_f()
        m0 = f.sum(-1)
        ux = (f[..., 1] - f[..., 3] + f[..., 5] - f[..., 6] - f[..., 7] + f[..., 8]) / m0
        uy = (f[..., 2] - f[..., 4] + f[..., 5] + f[..., 6] - f[..., 7] - f[..., 8]) / m0
        tol = 1e-12 if n <= 1000 else 1e-10
        assert relerr(m0, g["rho"][..., j]) < tol, n
        assert np.abs(ux - g["ux"][..., j]).max() < tol and np.abs(uy - g["uy"][..., j]).max() < tol, n
    assert done == 100000
    sv.close()


def test_kbc_sliding_window_carries_walls(lib, oracle):
    """KBC multi-step launches on wall-bounded single blocks (bounce-back columns; closed box):
    D = 2, 3 steps in one launch == single-step launches (interior kernel + edge pass), bit for bit."""
    rng = np.random.default_rng(8)
    R, C = 96, 150
    rho = 1 + 0.01 * rng.standard_normal((R, C))
    u = 0.03 * rng.standard_normal((R, C, 2))
    f0 = oracle.equilibrium(u, rho) * (1 + 0.01 * rng.standard_normal((R, C, 9)))
    g = pylbm.Geom(R, C, 0)
    prm = pylbm.KbcParams(S2)
    p0 = upload_soa(lib, f0)
    a, b = torch.empty_like(p0), torch.empty_like(p0)
    for rows_too in (False, True):
        bc = pylbm.Bc.periodic()
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
        if rows_too:
            bc.row_lo = bc.row_hi = pylbm.EDGE_BOUNCE_BACK
        for D in (2, 3):
            src = p0.clone()
            for _ in range(D):
                lib.kbc_stream_collide(_ptr(a), _ptr(src), ct.byref(g), ct.byref(bc), ct.byref(prm), 0, R, None, None, None)
                src, a = a, src
            lib.kbc_stream_collide_xn(_ptr(b), _ptr(p0), ct.byref(g), ct.byref(bc), ct.byref(prm), D, 0, R, None)
            torch.cuda.synchronize()
            assert torch.equal(b, src), (rows_too, D, float((b - src).abs().max()))


def test_two_solvers_with_different_collision_forms_coexist(lib, oracle):
    """the collision form is a field of the parameter structs, not process-wide state -- a
    reference-order solver and a reassociated one, BGK and KBC, stepped in turns inside one process.  Each equals the
    same solver run alone bit for bit; the reference-order ones equal the oracle bit for bit; the process-wide knobs
    stay at their defaults throughout."""
    R, C, n = 96, 128, 12
    rng = np.random.default_rng(5)
    rho = 1 + 0.01 * rng.standard_normal((R, C))
    u = 0.03 * rng.standard_normal((R, C, 2))
    f0 = oracle.equilibrium(u, rho)
    s2 = 1.0 / (0.5 + 3 * 1.7e-4)

    def make(model, form):
        prm = pylbm.BgkParams(1.3, 0, form=form) if model == pylbm.MODEL_BGK else pylbm.KbcParams(s2, form)
        sv = pylbm.Solver(lib, model, R, C, prm)
        sv.set_f(f0)
        return sv

    kinds = [(pylbm.MODEL_BGK, pylbm.FORM_REFERENCE_ORDER), (pylbm.MODEL_BGK, pylbm.FORM_REASSOCIATED),
             (pylbm.MODEL_KBC, pylbm.FORM_REFERENCE_ORDER), (pylbm.MODEL_KBC, pylbm.FORM_REASSOCIATED)]
    alone = []
    for k in kinds:
        sv = make(*k)
        sv.step(n)
        alone.append(sv.get_f())
        sv.close()
    svs = [make(*k) for k in kinds]
    for chunk in (1, 5, 3, 3):                 # turns of different lengths: multi-step launches and remainders interleave
        for sv in svs:
            sv.step(chunk)
    together = [sv.get_f() for sv in svs]
    for sv in svs:
        sv.close()
    for k, a, t in zip(kinds, alone, together):
        assert bits_equal(a, t), (k, ulp_diff(a, t))
    assert not bits_equal(together[0], together[1]) and not bits_equal(together[2], together[3])   # the forms really differ
    assert relerr(together[1], together[0]) < 1e-12 and relerr(together[3], together[2]) < 1e-11
    f_o, _, _ = oracle.bgk_periodic_steps(f0, 1.3, n)
    assert bits_equal(together[0], f_o), ulp_diff(together[0], f_o)
    for key in (b"bgk_fast", b"kbc_fast"):
        assert lib.raw.lbm_get_tuning(key) == 0      # unset (lbm_get_tuning reports 0 for keys nobody set)


@pytest.mark.parametrize("case", ["periodic", "pressure_rows"])
def test_kbc_solver_pads_its_rows_and_leaves_the_same_state(lib, oracle, tmp_path, case):
    """lbm_solver_create pads the rows of a KBC solver's lattices where C * 8 bytes is a multiple of 4 KiB
    (lbm_default_row_pitch, round 4; tuning "row_pad" = 0: dense rows): the same populations and moments bit for bit --
    multi-step windows + single steps, the held-moments start and the pressure-row blocks with their seam copies,
    get / set on the device, checkpoint save on a padded solver -> load on a dense one"""
    R, C = 96, 1024
    assert lib.raw.lbm_default_row_pitch(C) == C + 64
    if case == "periodic":
        m0, m1 = oracle.kbc_shear_init(R, C)
        f0 = oracle.kbc_equilibrium(m0, m1)
        s2, bc, n = 1.0 / (0.5 + 3 * 1.7e-4), None, 11
    else:
        s2, rin, bc = _upo_case(R, C)
        f0, n = np.zeros((R, C, 9)), 1 + 2 * 4 + 1
    got = {}
    try:
        for pad in (0, -1):
            lib.set_tuning(b"row_pad", pad)
            sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(s2), bc=bc)
            sv.set_f(f0)
            if case == "pressure_rows":
                sv.set_moments(np.ones((R, C)), np.zeros((R, C, 2)))
            sv.step(n, record_moments=True)
            rho, u = sv.moments()
            if pad == -1:
                sv.checkpoint_save(tmp_path / "ck.bin")
            got[pad] = (sv.get_f(), rho, u)
            sv.close()
        # the padded solver's checkpoint restarts a dense one
        lib.set_tuning(b"row_pad", 0)
        sv = pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(s2), bc=bc)
        sv.checkpoint_load(tmp_path / "ck.bin")
        f_ck = sv.get_f()
        sv.close()
    finally:
        lib.set_tuning(b"row_pad", -1)
    for a, b in zip(got[-1], got[0]):
        assert bits_equal(a, b), ulp_diff(a, b)
    assert bits_equal(f_ck, got[0][0])
