"""Pressure-periodic rows (test/horizontal_poiseuille_test.cpp:25-45, test/ulbm_poiseuille.cpp:36-58,
test/specular_boundary_test.cpp, test/gravity_test.cpp) at multi-step speed: lbm_solver_step advances D
steps per block -- rows at least D away from the virtual rows 0 / R-1 through the D-step window, the 2 D
rows on either side of the seam in single steps on a small periodic lattice (capi_solver.hip
solver_pressure_block).  Same kernels per node as single steps, so every depth must give the same BITS as
"pressure_depth" = 1 (one step per launch), which in turn is held to the oracle / the unmodified
reference mains elsewhere (test_gpu_bgk.py, test_gpu_presets.py, test_gpu_kbc.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
from gpu_util import bits_equal, ulp_diff  # noqa: E402
from pyoracle import hpt_params  # noqa: E402

W9 = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


def perturbed_rest(H, W, seed):
    """rest state + a smooth perturbation, so that a wrong row / column shows up at once"""
    rng = np.random.default_rng(seed)
    f = np.empty((H, W, 9))
    f[...] = W9
    return f * (1 + 0.01 * rng.standard_normal((H, W, 9)))


def run(lib, model, H, W, prm, bc, f0, n, depth, moments=None):
    lib.set_tuning(b"pressure_depth", depth)
    sv = pylbm.Solver(lib, model, H, W, prm, bc=bc)
    sv.set_f(f0)
    if moments is not None:
        sv.set_moments(*moments)
    sv.step(n)
    f = sv.get_f()
    sv.close()
    lib.set_tuning(b"pressure_depth", -1)
    return f


@pytest.mark.parametrize("case", ["poiseuille_bb", "specular", "gravity", "periodic_cols"])
@pytest.mark.parametrize("H,W", [(64, 64), (96, 150), (257, 200)])
def test_bgk_pressure_blocks_equal_single_steps(lib, case, H, W):
    p = hpt_params(H, W, 0)
    bc = pylbm.Bc(pressure_rows=1, rho_inlet=p.rho_inlet, rho_outlet=p.rho_outlet)
    prm = pylbm.BgkParams(p.omega, 1)
    if case == "poiseuille_bb":
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    elif case == "specular":                      # specular_boundary_test.cpp: compressible equilibrium
        bc.col_lo = bc.col_hi = pylbm.EDGE_SPECULAR
        prm = pylbm.BgkParams(p.omega, 0)
    elif case == "gravity":                       # gravity_test.cpp: body force, rho_in = rho_out
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
        bc.rho_inlet = bc.rho_outlet = 1.0
        prm = pylbm.BgkParams(p.omega, 1, force=(-0.0003, 0.0))
    f0 = perturbed_rest(H, W, seed=H + W)
    n = 23                                        # first step singly, then 5-blocks and a remainder
    want = run(lib, pylbm.MODEL_BGK, H, W, prm, bc, f0, n, 1)
    for depth in (2, 3, 5):
        if H < 6 * depth + 8:
            continue
        got = run(lib, pylbm.MODEL_BGK, H, W, prm, bc, f0, n, depth)
        assert bits_equal(got, want), (case, H, W, depth, ulp_diff(got, want))
    got = run(lib, pylbm.MODEL_BGK, H, W, prm, bc, f0, n, -1)   # the default
    assert bits_equal(got, want), (case, H, W, "default", ulp_diff(got, want))


def test_bgk_pressure_blocks_vs_oracle_config1(lib, oracle):
    """BASELINE config 1 (256 x 64) through 5-step blocks == the oracle bit for bit, and the blocks
    really ran (the default depth is 5; a silent fallback to single steps would pass otherwise)"""
    p = hpt_params(256, 64, 203, check_convergence=0)
    bc = pylbm.Bc(col_lo=pylbm.EDGE_BOUNCE_BACK, col_hi=pylbm.EDGE_BOUNCE_BACK, pressure_rows=1,
                  rho_inlet=p.rho_inlet, rho_outlet=p.rho_outlet)
    f0 = np.empty((256, 64, 9))
    f0[...] = W9
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, 256, 64, pylbm.BgkParams(p.omega, 1), bc=bc)
    sv.set_f(f0)
    assert lib.raw.lbm_solver_block_launches(sv.h) == 0
    sv.step(203)
    assert lib.raw.lbm_solver_block_launches(sv.h) == 41     # 1 single + 40 x 5 + one block of 2
    f = sv.get_f()
    sv.close()
    o = oracle.hpt_run(p)
    assert bits_equal(f, o["f"]), ulp_diff(f, o["f"])


@pytest.mark.parametrize("H,W", [(64, 64), (128, 128)])
def test_kbc_pressure_blocks_equal_single_steps(lib, H, W):
    """ulbm_poiseuille preset: KBC + pressure rows + bounce-back columns, 2 steps per block (the far rows
    through the reference-order 2-step window)"""
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    rin = 3.0 * (H - 1) * (8.0 * nu * 0.05 / (W * W)) + 1.0
    bc = pylbm.Bc(col_lo=pylbm.EDGE_BOUNCE_BACK, col_hi=pylbm.EDGE_BOUNCE_BACK, pressure_rows=1,
                  rho_inlet=rin, rho_outlet=1.0)
    f0 = np.zeros((H, W, 9))
    mom = (np.ones((H, W)), np.zeros((H, W, 2)))
    want = run(lib, pylbm.MODEL_KBC, H, W, pylbm.KbcParams(s2), bc, f0, 12, 1, mom)
    got = run(lib, pylbm.MODEL_KBC, H, W, pylbm.KbcParams(s2), bc, f0, 12, 2, mom)
    assert bits_equal(got, want), (H, W, ulp_diff(got, want))
    got = run(lib, pylbm.MODEL_KBC, H, W, pylbm.KbcParams(s2), bc, f0, 12, -1, mom)
    assert bits_equal(got, want), (H, W, "default", ulp_diff(got, want))


# ---- pressure-periodic rows over slabs (second half) -----------------------------------------
@pytest.mark.parametrize("case,n_slabs,D", [("poiseuille_bb", 2, 5), ("poiseuille_bb", 3, 5), ("specular", 2, 3),
                                            ("gravity", 3, 4), ("periodic_cols", 4, 2)])
def test_pressure_rows_over_emulated_slabs_equal_single_block(lib, case, n_slabs, D):
    """The channel presets over a PERIODIC ring of slabs, emulated on one GPU (messages moved by device copies):
    the two end slabs replicate the small seam lattice and swap the rows at distance [D, 2D) from the virtual rows
    per block; middle slabs are ordinary.  Start-up from the pre-collision state + 3 blocks == the single block
    stepped one step per launch, bit for bit."""
    import ctypes as ct
    from gpu_util import dev, download_aos, upload_soa
    from pylbm import _ptr
    R, W = 64, 150
    H = R * n_slabs
    p = hpt_params(H, W, 0)
    bc = pylbm.Bc(pressure_rows=1, rho_inlet=p.rho_inlet, rho_outlet=p.rho_outlet)
    prm = pylbm.BgkParams(p.omega, 1)
    if case == "poiseuille_bb":
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    elif case == "specular":
        bc.col_lo = bc.col_hi = pylbm.EDGE_SPECULAR
        prm = pylbm.BgkParams(p.omega, 0)
    elif case == "gravity":
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
        bc.rho_inlet = bc.rho_outlet = 1.0
        prm = pylbm.BgkParams(p.omega, 1, force=(-0.0003, 0.0))
    f0 = perturbed_rest(H, W, seed=H + W + D)
    nb = 3
    want = run(lib, pylbm.MODEL_BGK, H, W, prm, bc, f0, 1 + nb * D, 1)
    d = dev()
    geom = pylbm.Geom(R, W, D)
    lib.raw.lbm_slab_pressure_msg_doubles.restype = ct.c_longlong
    slabs = []
    for s in range(n_slabs):
        h = ct.c_void_p()
        lib.slab_pressure_create(ct.byref(h), ct.byref(geom), s * R, H, ct.byref(bc), ct.byref(prm), D)
        slabs.append(h)
    f0d = upload_soa(lib, f0)
    pre = [torch.zeros((9, R + 2 * D, W), dtype=torch.float64, device=d) for _ in range(n_slabs)]
    lat = [[torch.zeros((9, R + 2 * D, W), dtype=torch.float64, device=d) for _ in range(2)] for _ in range(n_slabs)]
    for s in range(n_slabs):
        pre[s][:, D:R + D] = f0d[:, s * R:(s + 1) * R]

    def exchange(pack, finish, start):
        bufs = []
        for s, h in enumerate(slabs):
            n = [int(lib.raw.lbm_slab_pressure_msg_doubles(h, side, start)) for side in (0, 1)]
            b = dict(sp=torch.zeros(n[0], dtype=torch.float64, device=d), rp=torch.zeros(n[0], dtype=torch.float64, device=d),
                     sn=torch.zeros(n[1], dtype=torch.float64, device=d), rn=torch.zeros(n[1], dtype=torch.float64, device=d))
            bufs.append(b)
            pack(s, h, b)
        torch.cuda.synchronize()
        for s in range(n_slabs):   # periodic ring
            nx = (s + 1) % n_slabs
            assert bufs[s]["sn"].numel() == bufs[nx]["rp"].numel() and bufs[nx]["sp"].numel() == bufs[s]["rn"].numel()
            bufs[nx]["rp"].copy_(bufs[s]["sn"])
            bufs[s]["rn"].copy_(bufs[nx]["sp"])
        torch.cuda.synchronize()
        for s, h in enumerate(slabs):
            finish(s, h, bufs[s])
        torch.cuda.synchronize()

    exchange(lambda s, h, b: lib.slab_pressure_start_pack(h, _ptr(pre[s]), _ptr(b["sp"]), _ptr(b["sn"]), None),
             lambda s, h, b: lib.slab_pressure_start_finish(h, _ptr(lat[s][0]), _ptr(pre[s]), _ptr(b["rp"]), _ptr(b["rn"]), None), 1)
    cur = 0
    for _ in range(nb):
        exchange(lambda s, h, b: lib.slab_pressure_block_compute(h, _ptr(lat[s][cur ^ 1]), _ptr(lat[s][cur]), _ptr(b["sp"]), _ptr(b["sn"]), None),
                 lambda s, h, b: lib.slab_pressure_block_finish(h, _ptr(lat[s][cur ^ 1]), _ptr(b["rp"]), _ptr(b["rn"]), None), 0)
        cur ^= 1
    P = torch.cat([lat[s][cur][:, D:R + D] for s in range(n_slabs)], dim=1).contiguous()
    out = torch.empty_like(P)
    flat = pylbm.Geom(H, W, 0)
    lib.stream(_ptr(out), _ptr(P), ct.byref(flat), ct.byref(bc), None)
    got = download_aos(lib, out)
    for h in slabs:
        lib.slab_pressure_destroy(h)
    assert bits_equal(got, want), (case, n_slabs, D, ulp_diff(got, want))


@pytest.mark.parametrize("n_slabs", [2, 3])
def test_kbc_pressure_rows_over_emulated_slabs_equal_single_block(lib, n_slabs):
    """ulbm_poiseuille (KBC + pressure rows + bounce-back columns, start from adve_f = 0 on held moments,
    test/ulbm_poiseuille.cpp:36-58, :85-139) over a periodic ring of slabs in 2-step blocks: the end slabs replicate the
    small seam lattice (its first iteration on the held moments of both sides), far rows through the reference-order
    2-step window; start-up + 4 blocks == the single block stepped one step per launch, bit for bit."""
    import ctypes as ct
    from gpu_util import dev, download_aos, upload_soa
    from pylbm import _ptr
    R, W, D, nb = 64, 128, 2, 4
    H = R * n_slabs
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    rin = 3.0 * (H - 1) * (8.0 * nu * 0.05 / (W * W)) + 1.0
    bc = pylbm.Bc(col_lo=pylbm.EDGE_BOUNCE_BACK, col_hi=pylbm.EDGE_BOUNCE_BACK, pressure_rows=1, rho_inlet=rin, rho_outlet=1.0)
    prm = pylbm.KbcParams(s2)
    rng = np.random.default_rng(3)
    m0 = 1.0 + 0.001 * rng.standard_normal((H, W))          # held moments that differ from row to row: a wrong row shows
    m1 = 0.001 * rng.standard_normal((H, W, 2))
    f0 = np.zeros((H, W, 9))
    want = run(lib, pylbm.MODEL_KBC, H, W, prm, bc, f0, 1 + nb * D, 1, (m0, m1))
    d = dev()
    geom = pylbm.Geom(R, W, D)
    lib.raw.lbm_slab_pressure_msg_doubles.restype = ct.c_longlong
    slabs = []
    for s_ in range(n_slabs):
        h = ct.c_void_p()
        lib.slab_pressure_create_kbc(ct.byref(h), ct.byref(geom), s_ * R, H, ct.byref(bc), ct.byref(prm))
        slabs.append(h)
    m0d = torch.from_numpy(m0).to(d)
    m1d = torch.from_numpy(np.ascontiguousarray(np.moveaxis(m1, -1, 0))).to(d)
    pre = [torch.zeros((9, R + 2 * D, W), dtype=torch.float64, device=d) for _ in range(n_slabs)]
    lat = [[torch.zeros((9, R + 2 * D, W), dtype=torch.float64, device=d) for _ in range(2)] for _ in range(n_slabs)]
    sm0 = [m0d[s_ * R:(s_ + 1) * R].contiguous() for s_ in range(n_slabs)]
    sm1 = [m1d[:, s_ * R:(s_ + 1) * R].contiguous() for s_ in range(n_slabs)]

    def exchange(pack, finish, start):
        bufs = []
        for s_, h in enumerate(slabs):
            n = [int(lib.raw.lbm_slab_pressure_msg_doubles(h, side, start)) for side in (0, 1)]
            b = dict(sp=torch.zeros(n[0], dtype=torch.float64, device=d), rp=torch.zeros(n[0], dtype=torch.float64, device=d),
                     sn=torch.zeros(n[1], dtype=torch.float64, device=d), rn=torch.zeros(n[1], dtype=torch.float64, device=d))
            bufs.append(b)
            pack(s_, h, b)
        torch.cuda.synchronize()
        for s_ in range(n_slabs):
            nx = (s_ + 1) % n_slabs
            assert bufs[s_]["sn"].numel() == bufs[nx]["rp"].numel() and bufs[nx]["sp"].numel() == bufs[s_]["rn"].numel()
            bufs[nx]["rp"].copy_(bufs[s_]["sn"])
            bufs[s_]["rn"].copy_(bufs[nx]["sp"])
        torch.cuda.synchronize()
        for s_, h in enumerate(slabs):
            finish(s_, h, bufs[s_])
        torch.cuda.synchronize()

    exchange(lambda s_, h, b: lib.slab_pressure_start_pack_kbc(h, _ptr(pre[s_]), _ptr(sm0[s_]), _ptr(sm1[s_]), _ptr(b["sp"]), _ptr(b["sn"]), None),
             lambda s_, h, b: lib.slab_pressure_start_finish_kbc(h, _ptr(lat[s_][0]), _ptr(pre[s_]), _ptr(sm0[s_]), _ptr(sm1[s_]),
                                                                 _ptr(b["rp"]), _ptr(b["rn"]), None), 1)
    # complete halos of the post-collision state over every seam of the ring (what lbm_ring_pressure_start_kbc ends with)
    full = 100 + D
    msg = torch.empty(lib.raw.lbm_halo_rows(full) * W, dtype=torch.float64, device=d)
    for s_ in range(n_slabs):
        nx = (s_ + 1) % n_slabs
        lib.halo_pack(_ptr(msg), _ptr(lat[s_][0]), ct.byref(geom), full, 1, None)
        lib.halo_unpack(_ptr(lat[nx][0]), _ptr(msg), ct.byref(geom), full, 0, None)
        lib.halo_pack(_ptr(msg), _ptr(lat[nx][0]), ct.byref(geom), full, 0, None)
        lib.halo_unpack(_ptr(lat[s_][0]), _ptr(msg), ct.byref(geom), full, 1, None)
    cur = 0
    for _ in range(nb):
        exchange(lambda s_, h, b: lib.slab_pressure_block_compute(h, _ptr(lat[s_][cur ^ 1]), _ptr(lat[s_][cur]), _ptr(b["sp"]), _ptr(b["sn"]), None),
                 lambda s_, h, b: lib.slab_pressure_block_finish(h, _ptr(lat[s_][cur ^ 1]), _ptr(b["rp"]), _ptr(b["rn"]), None), 0)
        cur ^= 1
    P = torch.cat([lat[s_][cur][:, D:R + D] for s_ in range(n_slabs)], dim=1).contiguous()
    out = torch.empty_like(P)
    flat = pylbm.Geom(H, W, 0)
    lib.stream(_ptr(out), _ptr(P), ct.byref(flat), ct.byref(bc), None)
    got = download_aos(lib, out)
    for h in slabs:
        lib.slab_pressure_destroy(h)
    assert bits_equal(got, want), (n_slabs, ulp_diff(got, want))
