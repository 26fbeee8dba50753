"""GPU parity tests for the BGK path, all through the C ABI (liblbm_hip.so).

Bars: streaming / layout = bit-exact (pure index work).  Collision arithmetic: the kernels
are built with -ffp-contract=off and evaluate the reference expressions in the reference
order, so they are compared BITWISE with the oracle (oracle/lbm_oracle.cpp, itself within a
few ulp/step of the unmodified reference, see test_oracle_golden.py) and to <= 1e-12
relative with the golden vectors of the unmodified reference (north star: 1e-6)."""
import ctypes as ct

import numpy as np
import pytest
from conftest import golden, relerr

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
from gpu_util import bits_equal as _bits_equal, dev, download_aos, ulp_diff, upload_soa  # noqa: E402
from pylbm import _ptr  # noqa: E402
from pyoracle import hpt_params  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1, "no HIP device visible"
    return lib


# Every test runs under both collision implementations of the plain compressible model:
#   reference_order  (tuning bgk_fast = 0) solver.cpp's operation order, -ffp-contract=off: the GPU
#                    results are compared BITWISE with the oracle;
#   fast             (the default) d2q9.hpp BgkFastModel -- reciprocal instead of two divisions,
#                    equilibrium split into even / odd parts, FMA per expression: stated tolerance
#                    1e-12 relative (L2 over the field) against the oracle for the <= 100 steps
#                    these tests run (north star: rho, u within 1e-6 for BGK).
# Incompressible / delta-form / forced models and lattices with pressure rows always run in the
# reference order, so their comparisons stay bitwise in both modes.
_MODE = {"fast": False}


def bits_equal(a, b):
    if _bits_equal(a, b):
        return True
    return _MODE["fast"] and relerr(a, b) < 1e-12


@pytest.fixture(autouse=True, params=["reference_order", "fast"])
def _collision_mode(request, lib):
    _MODE["fast"] = request.param == "fast"
    lib.set_tuning(b"bgk_fast", 1 if _MODE["fast"] else 0)
    yield
    lib.reset_tuning()
    _MODE["fast"] = False


def random_state(oracle, R, C, seed):
    rng = np.random.default_rng(seed)
    rho = 1 + 0.01 * rng.standard_normal((R, C))
    u = 0.05 * rng.standard_normal((R, C, 2))
    return oracle.equilibrium(u, rho) * (1 + 0.01 * rng.standard_normal((R, C, 9)))


@pytest.mark.parametrize("shape", [(1, 1, 9), (7, 5, 9), (37, 23, 9), (300, 257, 9), (64, 48, 2), (33, 65, 1)])
def test_layout_roundtrip_bit_exact(lib, shape):
    rng = np.random.default_rng(0)
    a = rng.standard_normal(shape)
    soa = upload_soa(lib, a)
    R, C, Q = shape
    # bit-exact node indexing: AoS ((r*C)+c)*Q+q  <->  SoA q*R*C + r*C + c
    assert bits_equal(soa.cpu().numpy(), np.moveaxis(a, -1, 0))
    assert bits_equal(download_aos(lib, soa), a)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_unfused_operators_vs_oracle_and_golden(lib, oracle, tag):
    g = golden("solver_units.npz")
    f, u, rho = g[f"{tag}_f_in"], g[f"{tag}_u_in"], g[f"{tag}_rho_in"]
    R, C, _ = f.shape
    fd, ud, rd = upload_soa(lib, f), upload_soa(lib, u), upload_soa(lib, rho)
    out_s = torch.empty((R, C), dtype=torch.float64, device=dev())
    out_v = torch.empty((2, R, C), dtype=torch.float64, device=dev())
    out_f = torch.empty((9, R, C), dtype=torch.float64, device=dev())

    lib.calc_rho(_ptr(out_s), _ptr(fd), R, C, None)
    got = download_aos(lib, out_s)
    assert bits_equal(got, oracle.calc_rho(f)) and relerr(got, g[f"{tag}_calc_rho"]) < 1e-15
    rho_ref = upload_soa(lib, g[f"{tag}_calc_rho"])
    lib.calc_u(_ptr(out_v), _ptr(fd), _ptr(rho_ref), R, C, None)
    got = download_aos(lib, out_v)
    assert bits_equal(got, oracle.calc_u(f, g[f"{tag}_calc_rho"])) and relerr(got, g[f"{tag}_calc_u"]) < 1e-14
    lib.calc_incomp_u(_ptr(out_v), _ptr(fd), R, C, None)
    got = download_aos(lib, out_v)
    assert bits_equal(got, oracle.calc_incomp_u(f)) and relerr(got, g[f"{tag}_calc_incomp_u"]) < 1e-14
    lib.equilibrium(_ptr(out_f), _ptr(ud), _ptr(rd), R, C, None)
    got = download_aos(lib, out_f)
    assert bits_equal(got, oracle.equilibrium(u, rho)) and relerr(got, g[f"{tag}_equilibrium"]) < 1e-15
    lib.incomp_equilibrium(_ptr(out_f), _ptr(ud), _ptr(rd), R, C, None)
    got = download_aos(lib, out_f)
    assert bits_equal(got, oracle.incomp_equilibrium(u, rho))
    assert relerr(got, g[f"{tag}_incomp_equilibrium"]) < 1e-15
    feq = upload_soa(lib, g[f"{tag}_equilibrium"])
    lib.collision(_ptr(out_f), _ptr(fd), _ptr(feq), ct.c_double(1.2), R, C, None)
    got = download_aos(lib, out_f)
    assert bits_equal(got, oracle.collision(f, g[f"{tag}_equilibrium"], 1.2))
    assert relerr(got, g[f"{tag}_collision_w1.2"]) < 1e-15
    lib.advect(_ptr(out_f), _ptr(fd), R, C, None)
    assert bits_equal(download_aos(lib, out_f), g[f"{tag}_advect"])  # bit-exact vs the reference


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("inc", [0, 1])
def test_bgk_periodic_solver_vs_oracle_and_golden(lib, oracle, tag, inc):
    g = golden("solver_units.npz")
    f0 = g[f"{tag}_f_in"]
    R, C, _ = f0.shape
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(1.2, inc))
    sv.set_f(f0)
    assert bits_equal(sv.get_f(), f0)
    done = 0
    for n in (1, 10, 100):
        sv.step(n - done, record_moments=True)
        done = n
        f = sv.get_f()
        rho, u = sv.moments()
        fo, rho_o, u_o = oracle.bgk_periodic_steps(f0, 1.2, n, bool(inc))
        assert bits_equal(f, fo), (n, ulp_diff(f, fo))
        assert bits_equal(rho, rho_o) and bits_equal(u, u_o)
        assert relerr(f, g[f"{tag}_bgk{n}_inc{inc}_f"]) < 1e-13
        assert relerr(rho, g[f"{tag}_bgk{n}_inc{inc}_rho"]) < 1e-13
        assert relerr(u, g[f"{tag}_bgk{n}_inc{inc}_u"]) < 1e-12
    sv.close()


@pytest.mark.parametrize("R,C", [(40, 64), (33, 512), (16, 1030), (130, 256)])
def test_interior_kernel_variants_bit_identical(lib, oracle, R, C):
    """Every tuning variant of the fused pull kernel == generic kernel == oracle, bitwise."""
    f0 = random_state(oracle, R, C, seed=R * 1000 + C)
    want, _, _ = oracle.bgk_periodic_steps(f0, 1.6, 7)
    lib.set_tuning(b"solver_depth", 1)   # one step per launch: this test is about the single-step variants
    combos = [(0, 0, 0, 256, 1), (1, 0, 0, 256, 1), (2, 0, 0, 256, 1), (2, 3, 0, 256, 1),
              (1, 3, 0, 256, 1), (2, 1, 64, 256, 1), (2, 2, 7, 256, 1), (1, 0, 5, 256, 1),
              (3, 3, 0, 256, 1), (3, 0, 0, 128, 1), (3, 1, 0, 512, 2), (3, 2, 0, 1024, 1),
              (3, 3, 0, 256, 4), (3, 3, 0, 128, 2)]
    for variant, nt, cap, block, rows in combos:
        lib.set_tuning(b"variant", variant)
        lib.set_tuning(b"nt", nt)
        lib.set_tuning(b"grid_cap", cap)
        lib.set_tuning(b"block", block)
        lib.set_tuning(b"rows", rows)
        lib.set_tuning(b"xcd_swizzle", (block // 128 + rows) % 2)
        sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(1.6, 0))
        sv.set_f(f0)
        sv.step(7)
        got = sv.get_f()
        sv.close()
        assert bits_equal(got, want), (variant, nt, cap, block, rows, ulp_diff(got, want))


def hpt_bc(p):
    bc = pylbm.Bc.periodic()
    bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    bc.pressure_rows = 1
    bc.rho_inlet, bc.rho_outlet = p.rho_inlet, p.rho_outlet
    return bc


def run_hpt_gpu(lib, p, steps):
    H, W = p.H, p.W
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, H, W, pylbm.BgkParams(p.omega, 1), bc=hpt_bc(p))
    f0 = np.empty((H, W, 9))
    f0[...] = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)  # incomp_equilibrium(u=0, rho=1)
    sv.set_f(f0)
    return sv


def test_poiseuille_main_golden_and_l2(lib, oracle):
    """config 1 (plumbing): test/horizontal_poiseuille_test.cpp at its own size, 21x21."""
    g = golden("hpt_21x21.npz")
    p = hpt_params(21, 21, int(g["T"]))
    sv = run_hpt_gpu(lib, p, 0)
    f0 = oracle.hpt_run(hpt_params(21, 21, 0))["f"]
    assert bits_equal(sv.get_f(), f0)
    done = 0
    for k, t in enumerate(g["steps"]):
        sv.step(int(t) - done, record_moments=True)
        done = int(t)
        f = sv.get_f()
        assert relerr(f, g["fs"][..., k]) < 1e-12, t          # unmodified reference main
        o = oracle.hpt_run(hpt_params(21, 21, done, check_convergence=0))
        assert bits_equal(f, o["f"]), (t, ulp_diff(f, o["f"]))  # oracle, bitwise
    # after T-1 iterations rho/u are what the reference saved in its last snapshot
    rho, u = sv.moments()
    k = len(g["steps"]) - 1
    assert relerr(u[..., 0], g["ux"][..., k]) < 1e-11
    assert relerr(rho, g["rho"][..., k]) < 1e-13
    sv.step(1, record_moments=True)  # iteration T: the state the L2 assertion looks at
    rho, u = sv.moments()
    W = 21
    y = np.arange(1, W + 1) - 0.5
    ua = -4.0 * p.u_max / (W * W) * y * (y - W)
    err = np.sqrt(((u[1:-1, :, 0] - ua) ** 2).sum(axis=1)) / np.sqrt((ua ** 2).sum())
    l2 = err.sum() / 21
    assert l2 <= 1e-11                                         # the reference's own assertion
    assert abs(l2 - float(g["l2_printed"])) < 5e-14
    sv.close()


def test_poiseuille_config1_256x64_vs_oracle(lib, oracle):
    """BASELINE config 1: 256x64 BGK Poiseuille; fast interior kernel + edge pass + pressure rows."""
    p = hpt_params(256, 64, 600, check_convergence=0)
    sv = run_hpt_gpu(lib, p, 0)
    sv.step(600, record_moments=True)
    o = oracle.hpt_run(p)
    f = sv.get_f()
    rho, u = sv.moments()
    assert bits_equal(f, o["f"]), ulp_diff(f, o["f"])
    assert bits_equal(rho, o["rho"]) and bits_equal(u, o["u"])
    sv.close()


def test_fullsize_box_equals_tiled_small_box(lib, oracle):
    """BASELINE config 2 size (8192 x 8192): a periodic box whose initial state has period
    64 x 64 must stay periodic and equal the 64 x 64 periodic box the oracle can run --
    checks every index computation of the full-size launch, bitwise."""
    tile = random_state(oracle, 64, 64, seed=5)
    want, _, _ = oracle.bgk_periodic_steps(tile, 1.2, 6)
    R = C = 8192
    t = upload_soa(lib, tile)                         # [9,64,64]
    big = t.repeat(1, R // 64, C // 64).contiguous()  # [9,8192,8192] SoA
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(1.2, 0))
    lib.solver_set_f_soa_dev(sv.h, _ptr(big))
    sv.step(6)
    lib.solver_get_f_soa_dev(sv.h, _ptr(big))
    torch.cuda.synchronize()
    blocks = big.view(9, R // 64, 64, C // 64, 64)
    ref_tile = blocks[:, 0, :, 0, :].contiguous()
    assert bits_equal(download_aos(lib, ref_tile), want)
    assert bool((blocks == ref_tile.view(9, 1, 64, 1, 64)).all())  # every tile identical
    # conservation at full size (exact in exact arithmetic; ~1e-13 relative in f64)
    mass0 = float(tile.sum()) * (R // 64) * (C // 64)
    assert abs(float(big.sum()) - mass0) / mass0 < 1e-12
    sv.close()


def test_two_slabs_on_one_gpu_equal_single_block(lib, oracle):
    """The ghost-row (slab) code path of the fused kernel: 2 slabs emulated on one GPU, halo =
    3 populations per side copied by hand (what SlabRing sends over RCCL), boundary rows and
    interior rows launched separately as in the overlap schedule.  Must equal the single
    periodic block bitwise (the contract of test/decompose_domain.cpp:181-187)."""
    from pylbm.slab import TO_NEXT, TO_PREV
    Rg, C, n = 96, 256, 6
    R = Rg // 2
    f0 = random_state(oracle, Rg, C, seed=77)
    want, _, _ = oracle.bgk_periodic_steps(f0, 1.4, n)
    prm = pylbm.BgkParams(1.4, 0)
    flat = pylbm.Geom(Rg, C, 0)
    p0 = torch.empty((9, Rg, C), dtype=torch.float64, device=dev())
    f0d = upload_soa(lib, f0)
    lib.bgk_collide(_ptr(p0), _ptr(f0d), ct.byref(flat), None, ct.byref(prm), None, None, None)
    torch.cuda.synchronize()
    geom = pylbm.Geom(R, C, 1)
    bc = pylbm.Bc.periodic()
    bc.row_lo = bc.row_hi = pylbm.EDGE_HALO
    lat = [[torch.zeros((9, R + 2, C), dtype=torch.float64, device=dev()) for _ in range(2)] for _ in range(2)]

    def halo(cur):
        for s in range(2):
            nxt = prv = 1 - s  # ring of two
            for q in TO_NEXT:
                lat[nxt][cur][q, 0] = lat[s][cur][q, R]
            for q in TO_PREV:
                lat[prv][cur][q, R + 1] = lat[s][cur][q, 1]

    for s in range(2):
        lat[s][0][:, 1:R + 1] = p0[:, s * R:(s + 1) * R]
    halo(0)
    cur = 0
    for _ in range(n - 1):
        for s in range(2):
            src, dst = lat[s][cur], lat[s][cur ^ 1]
            for r0, r1 in ((0, 1), (R - 1, R), (1, R - 1)):
                lib.bgk_stream_collide(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc),
                                       ct.byref(prm), r0, r1, None, None, None)
        torch.cuda.synchronize()
        cur ^= 1
        halo(cur)
    p = torch.cat([lat[0][cur][:, 1:R + 1], lat[1][cur][:, 1:R + 1]], dim=1).contiguous()
    out = torch.empty_like(p)
    lib.stream(_ptr(out), _ptr(p), ct.byref(flat), None, None)
    got = download_aos(lib, out)
    assert bits_equal(got, want), ulp_diff(got, want)


def test_slab_ring_rccl_self_exchange(lib, oracle):
    """SlabRing's real transport on the GPU box: torch.distributed backend nccl (= RCCL) with
    world_size 1 and forced ghost rows, so every halo row is an RCCL send/recv to self posted
    through the same batch_isend_irecv call the 8-GPU run uses.  Result == single block."""
    import torch.distributed as dist
    from pylbm.slab import SlabRing
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    d = dev()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=d)
    try:
        R, C, n = 64, 256, 8
        f0 = random_state(oracle, R, C, seed=31)
        want, _, _ = oracle.bgk_periodic_steps(f0, 1.1, n)
        prm = pylbm.BgkParams(1.1, 0)
        ring = SlabRing(lib, R, C, 0, 1, d, periodic=True, force_ghost=True)
        assert ring.ghost == 1 and ring.next_rank == 0 and ring.prev_rank == 0
        f0d = upload_soa(lib, f0)
        ring.load_precollision(f0d, lambda dst, src, geom: lib.bgk_collide(
            _ptr(dst), _ptr(src), ct.byref(geom), None, ct.byref(prm), None, None, ring.stream_ptr()))

        def step_rows(dst, src, geom, bc, r0, r1):
            lib.bgk_stream_collide(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc), ct.byref(prm),
                                   r0, r1, None, None, ring.stream_ptr())
        for _ in range(n - 1):
            ring.step(step_rows)
        torch.cuda.synchronize()
        p = ring.owned().contiguous()
        out = torch.empty_like(p)
        lib.stream(_ptr(out), _ptr(p), ct.byref(pylbm.Geom(R, C, 0)), None, None)
        got = download_aos(lib, out)
        assert bits_equal(got, want), ulp_diff(got, want)

        # depth 2: two time steps per launch, 9 halo rows per side per launch
        want2, _, _ = oracle.bgk_periodic_steps(f0, 1.1, 1 + 2 * 4)
        ring2 = SlabRing(lib, R, C, 0, 1, d, periodic=True, force_ghost=True, depth=2)
        assert ring2.ghost == 2
        ring2.load_precollision(f0d, lambda dst, src, geom: lib.bgk_collide(
            _ptr(dst), _ptr(src), ct.byref(geom), None, ct.byref(prm), None, None, ring2.stream_ptr()))

        def step_rows_x2(dst, src, geom, bc, r0, r1):
            lib.bgk_stream_collide_x2(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc),
                                      ct.byref(prm), r0, r1, ring2.stream_ptr())
        lib.set_tuning(b"tb_rows", 8)
        ring2.autotune(step_rows_x2, steps=0, edge_rows=8)   # both schedules once: 2 launches
        for _ in range(2):
            ring2.step(step_rows_x2, edge_rows=8)
        torch.cuda.synchronize()
        p = ring2.owned().contiguous()
        lib.stream(_ptr(out), _ptr(p), ct.byref(pylbm.Geom(R, C, 0)), None, None)
        got = download_aos(lib, out)
        assert bits_equal(got, want2), ulp_diff(got, want2)

        # depth 5: sliding-window launches, 36 halo rows per side per launch
        want5, _, _ = oracle.bgk_periodic_steps(f0, 1.1, 1 + 5 * 4)
        ring5 = SlabRing(lib, R, C, 0, 1, d, periodic=True, force_ghost=True, depth=5)
        ring5.load_precollision(f0d, lambda dst, src, geom: lib.bgk_collide(
            _ptr(dst), _ptr(src), ct.byref(geom), None, ct.byref(prm), None, None, ring5.stream_ptr()))

        def step_rows_x5(dst, src, geom, bc, r0, r1):
            lib.bgk_stream_collide_xn(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc),
                                      ct.byref(prm), 5, r0, r1, ring5.stream_ptr())
        ring5.autotune(step_rows_x5, steps=0, edge_rows=16)   # both schedules once: 2 launches
        for _ in range(2):
            ring5.step(step_rows_x5, edge_rows=16)
        torch.cuda.synchronize()
        p = ring5.owned().contiguous()
        lib.stream(_ptr(out), _ptr(p), ct.byref(pylbm.Geom(R, C, 0)), None, None)
        got = download_aos(lib, out)
        assert bits_equal(got, want5), ulp_diff(got, want5)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R,C", [(40, 128), (37, 64), (5, 192), (130, 256)])
def test_two_step_temporal_blocking_bit_identical(lib, oracle, R, C):
    """lbm_bgk_stream_collide_x2 (two steps per launch through an LDS tile) == two single
    steps == oracle, bitwise, for every tile shape; rows not a multiple of the tile included."""
    f0 = random_state(oracle, R, C, seed=R + C)
    n_pairs = 3
    want, _, _ = oracle.bgk_periodic_steps(f0, 1.5, 1 + 2 * n_pairs)
    prm = pylbm.BgkParams(1.5, 0)
    g = pylbm.Geom(R, C, 0)
    f0d = upload_soa(lib, f0)
    for tr, blk in [(4, 256), (6, 256), (8, 256), (8, 512), (12, 256), (12, 512), (14, 512),
                    (16, 512), (16, 1024), (30, 1024)]:
        lib.set_tuning(b"tb_rows", tr)
        lib.set_tuning(b"tb_block", blk)
        a = torch.empty((9, R, C), dtype=torch.float64, device=dev())
        b = torch.empty_like(a)
        lib.bgk_collide(_ptr(a), _ptr(f0d), ct.byref(g), None, ct.byref(prm), None, None, None)
        for _ in range(n_pairs):
            lib.bgk_stream_collide_x2(_ptr(b), _ptr(a), ct.byref(g), None, ct.byref(prm), 0, R, None)
            a, b = b, a
        out = torch.empty_like(a)
        lib.stream(_ptr(out), _ptr(a), ct.byref(g), None, None)
        got = download_aos(lib, out)
        assert bits_equal(got, want), (tr, blk, ulp_diff(got, want))
    lib.set_tuning(b"tb_rows", -1)
    lib.set_tuning(b"tb_block", -1)


def test_two_slabs_two_step_launches_equal_single_block(lib, oracle):
    """Temporal blocking across a seam: 2 slabs with TWO ghost rows emulated on one GPU, halo per
    pair of steps = the 9 rows per side SlabRing(depth=2) sends, edge tiles and interior tiles
    launched separately.  Must equal the single periodic block bitwise."""
    from pylbm.slab import HALO_TO_NEXT, HALO_TO_PREV
    Rg, C, pairs, TR = 96, 128, 3, 8
    R, G = Rg // 2, 2
    f0 = random_state(oracle, Rg, C, seed=78)
    want, _, _ = oracle.bgk_periodic_steps(f0, 1.4, 1 + 2 * pairs)
    prm = pylbm.BgkParams(1.4, 0)
    flat = pylbm.Geom(Rg, C, 0)
    p0 = torch.empty((9, Rg, C), dtype=torch.float64, device=dev())
    f0d = upload_soa(lib, f0)
    lib.bgk_collide(_ptr(p0), _ptr(f0d), ct.byref(flat), None, ct.byref(prm), None, None, None)
    torch.cuda.synchronize()
    geom = pylbm.Geom(R, C, G)
    bc = pylbm.Bc.periodic()
    bc.row_lo = bc.row_hi = pylbm.EDGE_HALO
    lat = [[torch.zeros((9, R + 2 * G, C), dtype=torch.float64, device=dev()) for _ in range(2)] for _ in range(2)]

    def halo(cur):
        for s in range(2):
            o = 1 - s
            for pops, k in HALO_TO_NEXT[2]:
                for q in pops:
                    lat[o][cur][q, G - 1 - k] = lat[s][cur][q, G + R - 1 - k]
            for pops, k in HALO_TO_PREV[2]:
                for q in pops:
                    lat[o][cur][q, G + R + k] = lat[s][cur][q, G + k]

    for s in range(2):
        lat[s][0][:, G:G + R] = p0[:, s * R:(s + 1) * R]
    halo(0)
    lib.set_tuning(b"tb_rows", TR)
    cur = 0
    for _ in range(pairs):
        for s in range(2):
            src, dst = lat[s][cur], lat[s][cur ^ 1]
            for r0, r1 in ((0, TR), (R - TR, R), (TR, R - TR)):
                lib.bgk_stream_collide_x2(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc),
                                          ct.byref(prm), r0, r1, None)
        torch.cuda.synchronize()
        cur ^= 1
        halo(cur)
    p = torch.cat([lat[0][cur][:, G:G + R], lat[1][cur][:, G:G + R]], dim=1).contiguous()
    out = torch.empty_like(p)
    lib.stream(_ptr(out), _ptr(p), ct.byref(flat), None, None)
    got = download_aos(lib, out)
    assert bits_equal(got, want), ulp_diff(got, want)
    # odd step count: one trailing SINGLE-step launch on the same depth-2 ghost geometry
    # (what bench.py does when --steps is odd), edge rows = the ghost depth
    for s in range(2):
        src, dst = lat[s][cur], lat[s][cur ^ 1]
        for r0, r1 in ((0, G), (R - G, R), (G, R - G)):
            lib.bgk_stream_collide(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc), ct.byref(prm),
                                   r0, r1, None, None, None)
    torch.cuda.synchronize()
    cur ^= 1
    halo(cur)
    p = torch.cat([lat[0][cur][:, G:G + R], lat[1][cur][:, G:G + R]], dim=1).contiguous()
    lib.stream(_ptr(out), _ptr(p), ct.byref(flat), None, None)
    want_odd, _, _ = oracle.bgk_periodic_steps(f0, 1.4, 2 + 2 * pairs)
    got = download_aos(lib, out)
    assert bits_equal(got, want_odd), ulp_diff(got, want_odd)


@pytest.mark.parametrize("R,C", [(40, 128), (37, 64), (96, 200), (130, 256)])
def test_sliding_window_temporal_blocking_bit_identical(lib, oracle, R, C):
    """lbm_bgk_stream_collide_xn (register sliding window, D = 2..6 steps per launch) == D single
    steps == oracle, bitwise; partial last strip (C not a multiple of the strip width), row
    chunks that do not divide R, 1/2/4 waves per workgroup, chunk height fixed or chosen by the launcher."""
    f0 = random_state(oracle, R, C, seed=3 * R + C)
    prm = pylbm.BgkParams(1.5, 0)
    g = pylbm.Geom(R, C, 0)
    f0d = upload_soa(lib, f0)
    p0 = torch.empty((9, R, C), dtype=torch.float64, device=dev())
    lib.bgk_collide(_ptr(p0), _ptr(f0d), ct.byref(g), None, ct.byref(prm), None, None, None)
    out = torch.empty_like(p0)
    for depth, rows, waves in [(2, 256, 4), (3, 16, 4), (4, 11, 2), (5, 32, 2), (6, 256, 4), (2, 7, 1), (4, 256, 4),
                               (5, -1, 2), (3, -1, 4)]:   # -1: rows fitted to the resident wave slots
        lib.set_tuning(b"sw_rows", rows)
        lib.set_tuning(b"sw_waves", waves)
        a, b = p0.clone(), torch.empty_like(p0)
        for _ in range(2):
            lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), None, ct.byref(prm), depth, 0, R, None)
            a, b = b, a
        lib.stream(_ptr(out), _ptr(a), ct.byref(g), None, None)
        got = download_aos(lib, out)
        want, _, _ = oracle.bgk_periodic_steps(f0, 1.5, 1 + 2 * depth)
        assert bits_equal(got, want), (depth, rows, waves, ulp_diff(got, want))
    lib.set_tuning(b"sw_rows", -1)
    lib.set_tuning(b"sw_waves", -1)


@pytest.mark.parametrize("depth", [3, 5])
def test_two_slabs_sliding_window_launches_equal_single_block(lib, oracle, depth):
    """D-step launches across a seam: 2 slabs with D ghost rows emulated on one GPU; halo per
    launch = the 9(D-1) rows per side SlabRing(depth=D) sends; edge rows and interior rows launched
    separately; a trailing single step on the same geometry.  Equals the single block bitwise."""
    from pylbm.slab import HALO_TO_NEXT, HALO_TO_PREV
    Rg, C, launches, E = 128, 128, 2, 16
    R, G = Rg // 2, depth
    f0 = random_state(oracle, Rg, C, seed=79 + depth)
    prm = pylbm.BgkParams(1.4, 0)
    flat = pylbm.Geom(Rg, C, 0)
    p0 = torch.empty((9, Rg, C), dtype=torch.float64, device=dev())
    f0d = upload_soa(lib, f0)
    lib.bgk_collide(_ptr(p0), _ptr(f0d), ct.byref(flat), None, ct.byref(prm), None, None, None)
    torch.cuda.synchronize()
    geom = pylbm.Geom(R, C, G)
    bc = pylbm.Bc(row_lo=pylbm.EDGE_HALO, row_hi=pylbm.EDGE_HALO)
    lat = [[torch.zeros((9, R + 2 * G, C), dtype=torch.float64, device=dev()) for _ in range(2)] for _ in range(2)]

    def halo(cur):
        for s in range(2):
            o = 1 - s
            for pops, k in HALO_TO_NEXT[depth]:
                for q in pops:
                    lat[o][cur][q, G - 1 - k] = lat[s][cur][q, G + R - 1 - k]
            for pops, k in HALO_TO_PREV[depth]:
                for q in pops:
                    lat[o][cur][q, G + R + k] = lat[s][cur][q, G + k]

    for s in range(2):
        lat[s][0][:, G:G + R] = p0[:, s * R:(s + 1) * R]
    halo(0)
    lib.set_tuning(b"sw_rows", 24)
    cur = 0
    for _ in range(launches):
        for s in range(2):
            src, dst = lat[s][cur], lat[s][cur ^ 1]
            for r0, r1 in ((0, E), (R - E, R), (E, R - E)):
                lib.bgk_stream_collide_xn(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc),
                                          ct.byref(prm), depth, r0, r1, None)
        torch.cuda.synchronize()
        cur ^= 1
        halo(cur)
    for s in range(2):   # one trailing single step
        src, dst = lat[s][cur], lat[s][cur ^ 1]
        for r0, r1 in ((0, G), (R - G, R), (G, R - G)):
            lib.bgk_stream_collide(_ptr(dst), _ptr(src), ct.byref(geom), ct.byref(bc), ct.byref(prm),
                                   r0, r1, None, None, None)
    torch.cuda.synchronize()
    cur ^= 1
    p = torch.cat([lat[0][cur][:, G:G + R], lat[1][cur][:, G:G + R]], dim=1).contiguous()
    out = torch.empty_like(p)
    lib.stream(_ptr(out), _ptr(p), ct.byref(flat), None, None)
    got = download_aos(lib, out)
    want, _, _ = oracle.bgk_periodic_steps(f0, 1.4, 1 + launches * depth + 1)
    assert bits_equal(got, want), ulp_diff(got, want)
    lib.set_tuning(b"sw_rows", -1)


@pytest.mark.parametrize("n,record", [(2, False), (7, False), (7, True), (13, True), (23, False)])
def test_solver_context_fuses_steps_transparently(lib, oracle, n, record):
    """lbm_solver_step groups driver iterations into multi-step launches on periodic BGK blocks
    (first iteration and the moment-recording one stay single): same results, bitwise."""
    R, C = 72, 128
    f0 = random_state(oracle, R, C, seed=n)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(1.3, 0))
    sv.set_f(f0)
    sv.step(n, record_moments=record)
    got = sv.get_f()
    want, rho_o, u_o = oracle.bgk_periodic_steps(f0, 1.3, n)
    assert bits_equal(got, want), ulp_diff(got, want)
    if record:
        rho, u = sv.moments()
        assert bits_equal(rho, rho_o) and bits_equal(u, u_o)
    sv.close()


def test_native_ring_self_exchange(lib, oracle):
    """The C++ slab ring (csrc/capi_ring.hip: dlopen'd RCCL, packed halo, edge stream): one rank,
    periodic, so both neighbours are the rank itself and every launch-step does a real
    ncclSend/ncclRecv pair per side.  Depth 5 (sliding window) and depth 1 (single-step kernel),
    bitwise against the single-block oracle."""
    d = dev()
    R, C = 96, 256
    f0 = random_state(oracle, R, C, seed=77)
    prm = pylbm.BgkParams(1.3, 0)
    ident = (ct.c_ubyte * 128)()
    lib.ring_unique_id(ident)
    for depth, launches in ((5, 3), (1, 6), (3, 2)):
        g = pylbm.Geom(R, C, depth)
        plane = (R + 2 * depth) * C
        lat = [torch.zeros(9 * plane, dtype=torch.float64, device=d) for _ in range(2)]
        ring = ct.c_void_p()
        if depth != 5:
            lib.ring_unique_id(ident)
        lib.ring_create(ct.byref(ring), ident, 0, 1, ct.byref(g), 1)
        try:
            f0d = upload_soa(lib, f0)           # [9][R][C] without ghosts
            flat = pylbm.Geom(R, C, 0)
            tmp = torch.empty_like(f0d)
            lib.bgk_collide(_ptr(tmp), _ptr(f0d), ct.byref(flat), None, ct.byref(prm), None, None, None)
            lat[0].view(9, R + 2 * depth, C)[:, depth:depth + R].copy_(tmp.view(9, R, C))
            torch.cuda.synchronize()
            lib.ring_exchange(ring, _ptr(lat[0]), None)
            lib.ring_join(ring, None)
            cur = 0
            for _ in range(launches):
                lib.ring_bgk_step(ring, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), None, ct.byref(prm),
                                  depth, 16, None)
                cur ^= 1
            torch.cuda.synchronize()
            p = lat[cur].view(9, R + 2 * depth, C)[:, depth:depth + R].contiguous()
            out = torch.empty_like(p)
            lib.stream(_ptr(out), _ptr(p), ct.byref(flat), None, None)
            got = download_aos(lib, out)
            want, _, _ = oracle.bgk_periodic_steps(f0, 1.3, 1 + depth * launches)
            assert bits_equal(got, want), (depth, ulp_diff(got, want))
        finally:
            lib.ring_destroy(ring)


@pytest.mark.parametrize("case", ["bb_cols", "closed_box", "channel_abb_specular", "channel_delta"])
def test_sliding_window_carries_walls(lib, oracle, case):
    """Multi-step launches on wall-bounded lattices: bounce-back / specular columns, bounce-back /
    anti-bounce-back-velocity rows are applied inside the register sliding window at every level
    (level 1: full boundary gather from memory; levels 2..D: fix-ups from the node's own ring row).
    D steps in one launch == D single-step launches (interior kernel + edge pass) bit for bit,
    D = 2..5, lattice sizes with partial strips and partial chunks."""
    bc = pylbm.Bc.periodic()
    prm = pylbm.BgkParams(1.3, 0)
    if case == "bb_cols":
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    elif case == "closed_box":
        bc.col_lo = bc.col_hi = bc.row_lo = bc.row_hi = pylbm.EDGE_BOUNCE_BACK
    else:
        bc.row_lo = bc.row_hi = pylbm.EDGE_ABB_VELOCITY
        bc.col_lo = bc.col_hi = pylbm.EDGE_SPECULAR
        bc.uw_r = 0.04
        if case == "channel_delta":
            prm = pylbm.BgkParams(1.3, 0, 1, form=pylbm.FORM_REFERENCE_ORDER)
    for R, C in ((96, 150), (70, 64), (130, 420)):
        f0 = random_state(oracle, R, C, seed=3)
        g = pylbm.Geom(R, C, 0)
        p0 = upload_soa(lib, f0)
        a, b = torch.empty_like(p0), torch.empty_like(p0)
        for D in (2, 3, 4, 5):
            if R < 4 * D + 8:
                continue
            src = p0.clone()
            for _ in range(D):
                lib.bgk_stream_collide(_ptr(a), _ptr(src), ct.byref(g), ct.byref(bc), ct.byref(prm), 0, R, None, None, None)
                src, a = a, src
            # split = 1 (default): wall-free interior through the plain instantiation, frame through the
            # wall-carrying one on the helper stream; 0: one wall-carrying launch
            for rows, split in ((64, 1), (24, 1), (-1, 1), (64, 0), (64, 2), (24, 2), (-1, 2)):   # 2: frame + interior in ONE dispatch (the default)
                lib.set_tuning(b"sw_rows", rows)
                lib.set_tuning(b"sw_split", split)
                b.zero_()
                lib.bgk_stream_collide_xn(_ptr(b), _ptr(p0), ct.byref(g), ct.byref(bc), ct.byref(prm), D, 0, R, None)
                torch.cuda.synchronize()
                if not torch.equal(b, src):
                    bad = (b != src).nonzero()
                    raise AssertionError((case, R, C, D, rows, float((b - src).abs().max()), bad[:6].tolist(), int(bad.shape[0]),
                                          sorted(set(bad[:, 2].tolist()))[:12], sorted(set(bad[:, 1].tolist()))[:12], sorted(set(bad[:, 0].tolist()))))
    lib.set_tuning(b"sw_rows", -1)
    lib.set_tuning(b"sw_split", -1)


@pytest.mark.parametrize("depth", [2, 4])
def test_two_slabs_with_walls_multi_step_equal_single_block(lib, oracle, depth):
    """Wall-bounded channel over slabs with multi-step launches: 2 slabs emulated on one GPU (chain:
    bounce-back rows at the global ends, bounce-back columns everywhere), COMPLETE ghost rows moved
    with lbm_halo_pack / _unpack(LBM_HALO_FULL(depth)) -- the fix-ups of a ghost-row wall node read
    that node's own populations.  Equals the single block advanced by single-step launches."""
    Rg, C, launches = 128, 150, 3
    R, D = Rg // 2, depth
    f0 = random_state(oracle, Rg, C, seed=11)
    prm = pylbm.BgkParams(1.25, 0)
    bc_g = pylbm.Bc.periodic()
    bc_g.row_lo = bc_g.row_hi = bc_g.col_lo = bc_g.col_hi = pylbm.EDGE_BOUNCE_BACK
    flat = pylbm.Geom(Rg, C, 0)
    p0 = upload_soa(lib, f0)
    a, b = p0.clone(), torch.empty_like(p0)
    for _ in range(D * launches):
        lib.bgk_stream_collide(_ptr(b), _ptr(a), ct.byref(flat), ct.byref(bc_g), ct.byref(prm), 0, Rg, None, None, None)
        a, b = b, a
    torch.cuda.synchronize()
    want = a
    geom = pylbm.Geom(R, C, D)
    bcs = []
    for s in range(2):
        bb = pylbm.Bc.periodic()
        bb.col_lo = bb.col_hi = pylbm.EDGE_BOUNCE_BACK
        bb.row_lo = pylbm.EDGE_BOUNCE_BACK if s == 0 else pylbm.EDGE_HALO
        bb.row_hi = pylbm.EDGE_HALO if s == 0 else pylbm.EDGE_BOUNCE_BACK
        bcs.append(bb)
    lat = [[torch.zeros((9, R + 2 * D, C), dtype=torch.float64, device=dev()) for _ in range(2)] for _ in range(2)]
    H = 100 + D
    msg = torch.empty(lib.raw.lbm_halo_rows(H) * C, dtype=torch.float64, device=dev())
    assert lib.raw.lbm_halo_rows(H) == 9 * D

    def halo(cur):
        lib.halo_pack(_ptr(msg), _ptr(lat[0][cur]), ct.byref(geom), H, 1, None)
        lib.halo_unpack(_ptr(lat[1][cur]), _ptr(msg), ct.byref(geom), H, 0, None)
        lib.halo_pack(_ptr(msg), _ptr(lat[1][cur]), ct.byref(geom), H, 0, None)
        lib.halo_unpack(_ptr(lat[0][cur]), _ptr(msg), ct.byref(geom), H, 1, None)

    for s in range(2):
        lat[s][0][:, D:D + R] = p0[:, s * R:(s + 1) * R]
    halo(0)
    cur = 0
    for _ in range(launches):
        for s in range(2):
            for r0, r1 in ((0, 16), (R - 16, R), (16, R - 16)):
                lib.bgk_stream_collide_xn(_ptr(lat[s][cur ^ 1]), _ptr(lat[s][cur]), ct.byref(geom), ct.byref(bcs[s]),
                                          ct.byref(prm), D, r0, r1, None)
        cur ^= 1
        halo(cur)
    torch.cuda.synchronize()
    got = torch.cat([lat[0][cur][:, D:D + R], lat[1][cur][:, D:D + R]], dim=1)
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("rowmode", ["bounce_back", "abb_velocity"])
@pytest.mark.parametrize("depth", [2, 5])
def test_window_row_walls_with_periodic_columns(lib, oracle, rowmode, depth):
    """Wall ROWS + PERIODIC columns in the multi-step window (round 3 fix): the lanes of a strip that lie beyond the
    lattice hold wrapped columns -- real nodes of the wall rows, whose fix-ups were skipped, so the corners fed
    garbage (over slabs: NaN from the unused ghost rows behind the wall) into the valid columns.  One block with
    every frame / interior split and two emulated slabs == single steps, bit for bit."""
    mode = pylbm.EDGE_BOUNCE_BACK if rowmode == "bounce_back" else pylbm.EDGE_ABB_VELOCITY
    Rg, C, launches = 128, 150, 2
    R, D = Rg // 2, depth
    f0 = random_state(oracle, Rg, C, seed=11)
    prm = pylbm.BgkParams(1.25, 0)
    bc_g = pylbm.Bc.periodic()
    bc_g.row_lo = bc_g.row_hi = mode
    bc_g.uw_r = 0.03
    flat = pylbm.Geom(Rg, C, 0)
    p0 = upload_soa(lib, f0)
    a, b = p0.clone(), torch.empty_like(p0)
    for _ in range(D * launches):
        lib.bgk_stream_collide(_ptr(b), _ptr(a), ct.byref(flat), ct.byref(bc_g), ct.byref(prm), 0, Rg, None, None, None)
        a, b = b, a
    torch.cuda.synchronize()
    want = a
    try:
        for split in (0, 1, 2):
            lib.set_tuning(b"sw_split", split)
            x, y = p0.clone(), torch.empty_like(p0)
            for _ in range(launches):
                lib.bgk_stream_collide_xn(_ptr(y), _ptr(x), ct.byref(flat), ct.byref(bc_g), ct.byref(prm), D, 0, Rg, None)
                x, y = y, x
            torch.cuda.synchronize()
            assert torch.equal(x, want), (split, float((x - want).abs().max()))
    finally:
        lib.set_tuning(b"sw_split", -1)
    geom = pylbm.Geom(R, C, D)
    bcs = []
    for s in range(2):
        bb = pylbm.Bc.periodic()
        bb.uw_r = 0.03
        bb.row_lo = mode if s == 0 else pylbm.EDGE_HALO
        bb.row_hi = pylbm.EDGE_HALO if s == 0 else mode
        bcs.append(bb)
    lat = [[torch.zeros((9, R + 2 * D, C), dtype=torch.float64, device=dev()) for _ in range(2)] for _ in range(2)]
    H = 100 + D
    msg = torch.empty(lib.raw.lbm_halo_rows(H) * C, dtype=torch.float64, device=dev())

    def halo(cur):
        lib.halo_pack(_ptr(msg), _ptr(lat[0][cur]), ct.byref(geom), H, 1, None)
        lib.halo_unpack(_ptr(lat[1][cur]), _ptr(msg), ct.byref(geom), H, 0, None)
        lib.halo_pack(_ptr(msg), _ptr(lat[1][cur]), ct.byref(geom), H, 0, None)
        lib.halo_unpack(_ptr(lat[0][cur]), _ptr(msg), ct.byref(geom), H, 1, None)

    for s in range(2):
        lat[s][0][:, D:D + R] = p0[:, s * R:(s + 1) * R]
    halo(0)
    cur = 0
    for _ in range(launches):
        for s in range(2):
            for r0, r1 in ((0, 16), (R - 16, R), (16, R - 16)):
                lib.bgk_stream_collide_xn(_ptr(lat[s][cur ^ 1]), _ptr(lat[s][cur]), ct.byref(geom), ct.byref(bcs[s]),
                                          ct.byref(prm), D, r0, r1, None)
        cur ^= 1
        halo(cur)
    torch.cuda.synchronize()
    got = torch.cat([lat[0][cur][:, D:D + R], lat[1][cur][:, D:D + R]], dim=1)
    assert torch.equal(got, want), float((got - want).abs().max())


def test_native_ring_with_wall_columns(lib, oracle):
    """lbm_ring_bgk_step on a channel (bounce-back columns, periodic rows): one rank, self send/recv
    of COMPLETE ghost rows, 4-step launches; equals the single block advanced step by step."""
    R, C, D, launches = 96, 200, 4, 3
    f0 = random_state(oracle, R, C, seed=12)
    prm = pylbm.BgkParams(1.2, 0)
    bc = pylbm.Bc.periodic()
    bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    flat = pylbm.Geom(R, C, 0)
    p0 = upload_soa(lib, f0)
    a, b = p0.clone(), torch.empty_like(p0)
    for _ in range(D * launches):
        lib.bgk_stream_collide(_ptr(b), _ptr(a), ct.byref(flat), ct.byref(bc), ct.byref(prm), 0, R, None, None, None)
        a, b = b, a
    torch.cuda.synchronize()
    g = pylbm.Geom(R, C, D)
    lat = [torch.zeros((9, R + 2 * D, C), dtype=torch.float64, device=dev()) for _ in range(2)]
    lat[0][:, D:D + R] = p0
    ident = (ct.c_ubyte * 128)()
    ring = ct.c_void_p()
    lib.ring_unique_id(ident)
    lib.ring_create(ct.byref(ring), ident, 0, 1, ct.byref(g), 1)
    try:
        torch.cuda.synchronize()
        lib.ring_exchange_full(ring, _ptr(lat[0]), None)
        lib.ring_join(ring, None)
        cur = 0
        for _ in range(launches):
            lib.ring_bgk_step(ring, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), ct.byref(bc), ct.byref(prm), D, 16, None)
            cur ^= 1
        torch.cuda.synchronize()
        got = lat[cur][:, D:D + R]
        assert torch.equal(got, a), float((got - a).abs().max())
    finally:
        lib.ring_destroy(ring)


def test_wall_split_launches_replay_from_a_captured_graph(lib, oracle):
    """The interior / frame split forks onto a helper stream and queries the occupancy of its kernels:
    both must be legal inside a stream capture (lbm_graph_*).  Two captured 5-step launches on a closed
    box, replayed three times == six direct launches, bit for bit."""
    R, C, D = 96, 200, 5
    bc = pylbm.Bc.periodic()
    bc.col_lo = bc.col_hi = bc.row_lo = bc.row_hi = pylbm.EDGE_BOUNCE_BACK
    prm = pylbm.BgkParams(1.3, 0)
    g = pylbm.Geom(R, C, 0)
    p0 = upload_soa(lib, random_state(oracle, R, C, seed=11))
    a, b = p0.clone(), torch.empty_like(p0)
    for _ in range(3):
        lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), D, 0, R, None)
        lib.bgk_stream_collide_xn(_ptr(a), _ptr(b), ct.byref(g), ct.byref(bc), ct.byref(prm), D, 0, R, None)
    torch.cuda.synchronize()
    want = a.clone()
    st = ct.c_void_p()
    lib.stream_create(ct.byref(st))
    graph = ct.c_void_p()
    c, d = p0.clone(), torch.empty_like(p0)
    torch.cuda.synchronize()
    try:
        lib.graph_begin_capture(st)
        lib.bgk_stream_collide_xn(_ptr(d), _ptr(c), ct.byref(g), ct.byref(bc), ct.byref(prm), D, 0, R, st)
        lib.bgk_stream_collide_xn(_ptr(c), _ptr(d), ct.byref(g), ct.byref(bc), ct.byref(prm), D, 0, R, st)
        lib.graph_end_capture(st, ct.byref(graph))
        lib.graph_launch(graph, 3, st)
        lib.stream_sync(st)
        assert torch.equal(c, want), float((c - want).abs().max())
    finally:
        if graph:
            lib.graph_destroy(graph)
        lib.stream_destroy(st)


@pytest.mark.parametrize("axis", ["cols", "rows"])
def test_mixed_wall_and_periodic_axis_falls_back_to_single_steps(lib, oracle, axis):
    """A wall on one side of an axis and PERIODIC on the other: the single-step gather
    handles it, the multi-step window does not (column clamp / row wrap are per axis) -- the
    launcher must refuse it and lbm_solver_step must fall back, so that n fused driver iterations
    equal n single-step launches bit for bit."""
    R, C = 96, 150
    bc = pylbm.Bc.periodic()
    if axis == "cols":
        bc.col_lo = pylbm.EDGE_BOUNCE_BACK
    else:
        bc.row_hi = pylbm.EDGE_BOUNCE_BACK
    prm = pylbm.BgkParams(1.3, 0)
    f0 = random_state(oracle, R, C, seed=5)
    g = pylbm.Geom(R, C, 0)
    p0 = upload_soa(lib, f0)
    a, b = torch.empty_like(p0), torch.empty_like(p0)
    with pytest.raises(pylbm.LbmError, match="both edges of an axis"):
        lib.bgk_stream_collide_xn(_ptr(b), _ptr(p0), ct.byref(g), ct.byref(bc), ct.byref(prm), 3, 0, R, None)
    n = 7
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, prm, bc=bc)
    sv.set_f(f0)
    sv.step(n)
    got = sv.get_f()
    sv.close()
    # the same through explicit launches: collide-only, n - 1 single steps, stream at the end
    lib.bgk_collide(_ptr(a), _ptr(p0), ct.byref(g), ct.byref(bc), ct.byref(prm), None, None, None)
    for _ in range(n - 1):
        lib.bgk_stream_collide(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), 0, R, None, None, None)
        a, b = b, a
    lib.stream(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), None)
    torch.cuda.synchronize()
    assert _bits_equal(got, download_aos(lib, b))


def test_ghost_rows_reject_periodic_row_edges(lib):
    """with ghost rows nothing wraps: a PERIODIC (or NULL = periodic) bc on a ghost-row geometry would
    read ghost rows nobody fills -- rejected with an error, not silently computed"""
    R, C = 64, 64
    g = pylbm.Geom(R, C, 1)
    a = torch.zeros((9, R + 2, C), dtype=torch.float64, device=dev())
    b = torch.zeros_like(a)
    prm = pylbm.BgkParams(1.0, 0)
    with pytest.raises(pylbm.LbmError, match="ghost rows"):
        lib.bgk_stream_collide(_ptr(b), _ptr(a), ct.byref(g), None, ct.byref(prm), 0, R, None, None, None)
    bc = pylbm.Bc.periodic()
    bc.row_lo = pylbm.EDGE_HALO   # row_hi stays PERIODIC
    with pytest.raises(pylbm.LbmError, match="ghost rows"):
        lib.bgk_stream_collide(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), 0, R, None, None, None)
    bc.row_hi = pylbm.EDGE_HALO
    lib.bgk_stream_collide(_ptr(b), _ptr(a), ct.byref(g), ct.byref(bc), ct.byref(prm), 0, R, None, None, None)
    torch.cuda.synchronize()


@pytest.mark.parametrize("R,C", [(64, 256), (96, 300), (130, 512), (75, 1000)])
def test_paired_strip_and_deep_prefetch_windows_bit_identical(lib, oracle, R, C):
    """The two further forms of the 5-step window: "sw_pair" = 2 / 4 (the waves of a workgroup own adjacent
    windows and hand each other the edge columns of every level through LDS; levels lag by two rows, one
    barrier per iteration) and "sw_pf2" = 1 (level-1 rows prefetched two iterations ahead).  Same arithmetic
    per node: 2 launches == 10 single steps == oracle bit for bit; partial last group, chunks that do not divide
    R, column counts that wrap inside a group."""
    if not lib.raw.lbm_build_has_experiments():
        pytest.skip("paired strips / two-row prefetch are experiments: make -C lattice-boltzmann-method_amd/csrc EXPERIMENTS=1")
    f0 = random_state(oracle, R, C, seed=7 * R + C)
    prm = pylbm.BgkParams(1.5, 0)
    g = pylbm.Geom(R, C, 0)
    f0d = upload_soa(lib, f0)
    p0 = torch.empty((9, R, C), dtype=torch.float64, device=dev())
    lib.bgk_collide(_ptr(p0), _ptr(f0d), ct.byref(g), None, ct.byref(prm), None, None, None)
    out = torch.empty_like(p0)
    want, _, _ = oracle.bgk_periodic_steps(f0, 1.5, 11)
    try:
        for pair, pf2, rows in [(2, 0, 32), (2, 0, -1), (4, 0, 17), (4, 0, -1), (0, 1, 32), (0, 1, -1), (2, 0, 256)]:
            lib.set_tuning(b"sw_pair", pair)
            lib.set_tuning(b"sw_pf2", pf2)
            lib.set_tuning(b"sw_rows", rows)
            a, b = p0.clone(), torch.zeros_like(p0)
            for _ in range(2):
                lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(g), None, ct.byref(prm), 5, 0, R, None)
                a, b = b, a
            lib.stream(_ptr(out), _ptr(a), ct.byref(g), None, None)
            got = download_aos(lib, out)
            assert bits_equal(got, want), (pair, pf2, rows, ulp_diff(got, want))
    finally:
        for k in (b"sw_pair", b"sw_pf2", b"sw_rows"):
            lib.set_tuning(k, -1)


@pytest.mark.parametrize("walls", [False, True])
@pytest.mark.parametrize("period,launches", [(2, 4), (2, 3), (3, 7)])
def test_native_ring_one_exchange_per_several_launches(lib, oracle, walls, period, launches):
    """lbm_ring_bgk_step on a closed ring whose slabs carry ghost = period x D rows: only every period-th
    launch exchanges (period x D rows deep); the launches in between run as ONE plain launch over the owned
    rows plus the ghost rows the later launches of the period still read.  One rank, self send / recv;
    owned rows after any number of launches (also one that stops inside a period) == the single block
    advanced by the same window kernel, bit for bit.  With bounce-back columns the message carries complete
    ghost rows (LBM_HALO_FULL)."""
    R, C, D = 128, 200, 4
    G = period * D
    f0 = random_state(oracle, R, C, seed=21)
    prm = pylbm.BgkParams(1.2, 0)
    bc = pylbm.Bc.periodic()
    if walls:
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    flat = pylbm.Geom(R, C, 0)
    p0 = upload_soa(lib, f0)
    a, b = p0.clone(), torch.empty_like(p0)
    for _ in range(launches):
        lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(flat), ct.byref(bc), ct.byref(prm), D, 0, R, None)
        a, b = b, a
    torch.cuda.synchronize()
    g = pylbm.Geom(R, C, G)
    lat = [torch.zeros((9, R + 2 * G, C), dtype=torch.float64, device=dev()) for _ in range(2)]
    lat[0][:, G:G + R] = p0
    lat[1].fill_(float("nan"))          # nothing may be read that was not written
    ident = (ct.c_ubyte * 128)()
    ring = ct.c_void_p()
    lib.ring_unique_id(ident)
    lib.ring_create(ct.byref(ring), ident, 0, 1, ct.byref(g), 1)
    try:
        torch.cuda.synchronize()
        (lib.ring_exchange_full if walls else lib.ring_exchange)(ring, _ptr(lat[0]), None)
        lib.ring_join(ring, None)
        cur = 0
        for _ in range(launches):
            lib.ring_bgk_step(ring, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), ct.byref(bc), ct.byref(prm), D, 16, None)
            cur ^= 1
        torch.cuda.synchronize()
        got = lat[cur][:, G:G + R]
        assert torch.equal(got, a), float((got - a).abs().max())
        # "ring_period" = 1 switches the scheme off: every launch exchanges, same result
        lib.set_tuning(b"ring_period", 1)
        lat[0][:, G:G + R] = p0
        torch.cuda.synchronize()
        (lib.ring_exchange_full if walls else lib.ring_exchange)(ring, _ptr(lat[0]), None)
        lib.ring_join(ring, None)
        cur = 0
        for _ in range(launches):
            lib.ring_bgk_step(ring, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), ct.byref(bc), ct.byref(prm), D, 16, None)
            cur ^= 1
        torch.cuda.synchronize()
        assert torch.equal(lat[cur][:, G:G + R], a)
    finally:
        lib.set_tuning(b"ring_period", -1)
        lib.ring_destroy(ring)


@pytest.mark.parametrize("D,period", [(5, 2), (3, 3)])
def test_three_slabs_one_exchange_per_several_launches(lib, oracle, D, period):
    """The schedule of the ring with ghost = period x D rows, on three DIFFERENT slabs of a periodic box
    emulated on one GPU (the self ring above cannot tell a slab from its neighbour): messages of the
    partial depth-(period x D) halo (9 (period D - 1) rows) moved with lbm_halo_pack / _unpack, the
    in-between launches over a widened geometry (owned rows + the ghost rows still needed), the last of
    a period over the owned rows.  Equals the single block, bit for bit."""
    Rg, C, S = 192, 128, 3
    R, G = Rg // S, period * D
    launches = 2 * period + 1
    f0 = random_state(oracle, Rg, C, seed=5)
    prm = pylbm.BgkParams(1.4, 0)
    bc = pylbm.Bc.periodic()
    flat = pylbm.Geom(Rg, C, 0)
    p0 = upload_soa(lib, f0)
    a, b = p0.clone(), torch.empty_like(p0)
    for _ in range(launches):
        lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(flat), ct.byref(bc), ct.byref(prm), D, 0, Rg, None)
        a, b = b, a
    torch.cuda.synchronize()
    geom = pylbm.Geom(R, C, G)
    hb = pylbm.Bc.periodic()
    hb.row_lo = hb.row_hi = pylbm.EDGE_HALO
    lat = [[torch.full((9, R + 2 * G, C), float("nan"), dtype=torch.float64, device=dev()) for _ in range(2)] for _ in range(S)]
    assert lib.raw.lbm_halo_rows(G) == 9 * (G - 1)
    msg = torch.empty(lib.raw.lbm_halo_rows(G) * C, dtype=torch.float64, device=dev())

    def halo(cur):
        for s in range(S):
            n = (s + 1) % S
            lib.halo_pack(_ptr(msg), _ptr(lat[s][cur]), ct.byref(geom), G, 1, None)
            lib.halo_unpack(_ptr(lat[n][cur]), _ptr(msg), ct.byref(geom), G, 0, None)
            lib.halo_pack(_ptr(msg), _ptr(lat[n][cur]), ct.byref(geom), G, 0, None)
            lib.halo_unpack(_ptr(lat[s][cur]), _ptr(msg), ct.byref(geom), G, 1, None)

    for s in range(S):
        lat[s][0][:, G:G + R] = p0[:, s * R:(s + 1) * R]
    halo(0)
    cur, phase = 0, 0
    for _ in range(launches):
        left = period - 1 - phase
        for s in range(S):
            if left > 0:
                wide = pylbm.Geom(R + 2 * left * D, C, G - left * D, (R + 2 * G) * C)
                lib.bgk_stream_collide_xn(_ptr(lat[s][cur ^ 1]), _ptr(lat[s][cur]), ct.byref(wide), ct.byref(hb),
                                          ct.byref(prm), D, 0, R + 2 * left * D, None)
            else:
                lib.bgk_stream_collide_xn(_ptr(lat[s][cur ^ 1]), _ptr(lat[s][cur]), ct.byref(geom), ct.byref(hb),
                                          ct.byref(prm), D, 0, R, None)
        cur ^= 1
        if left > 0:
            phase += 1
        else:
            halo(cur)
            phase = 0
    torch.cuda.synchronize()
    got = torch.cat([lat[s][cur][:, G:G + R] for s in range(S)], dim=1)
    assert torch.equal(got, a), float((got - a).abs().max())


def test_native_ring_mixed_depths_between_exchanges(lib, oracle):
    """The ring keeps count of the ghost rows that are still current: launches of varying depth on 12 ghost
    rows skip the exchange whenever this launch AND one more of the same depth fit into what is left, and
    exchange otherwise.  Self ring; owned rows after every launch == the single block advanced as far."""
    R, C, G = 128, 192, 12
    depths = [4, 3, 5, 4, 2, 5, 5, 3]
    f0 = random_state(oracle, R, C, seed=31)
    prm = pylbm.BgkParams(1.1, 0)
    bc = pylbm.Bc.periodic()
    flat = pylbm.Geom(R, C, 0)
    p0 = upload_soa(lib, f0)
    g = pylbm.Geom(R, C, G)
    lat = [torch.zeros((9, R + 2 * G, C), dtype=torch.float64, device=dev()) for _ in range(2)]
    lat[0][:, G:G + R] = p0
    lat[1].fill_(float("nan"))
    ident = (ct.c_ubyte * 128)()
    ring = ct.c_void_p()
    lib.ring_unique_id(ident)
    lib.ring_create(ct.byref(ring), ident, 0, 1, ct.byref(g), 1)
    try:
        torch.cuda.synchronize()
        lib.ring_exchange(ring, _ptr(lat[0]), None)
        lib.ring_join(ring, None)
        a, b = p0.clone(), torch.empty_like(p0)
        cur = 0
        for d in depths:
            lib.ring_bgk_step(ring, _ptr(lat[cur ^ 1]), _ptr(lat[cur]), ct.byref(bc), ct.byref(prm), d, 16, None)
            cur ^= 1
            lib.bgk_stream_collide_xn(_ptr(b), _ptr(a), ct.byref(flat), ct.byref(bc), ct.byref(prm), d, 0, R, None)
            a, b = b, a
            torch.cuda.synchronize()
            got = lat[cur][:, G:G + R]
            assert torch.equal(got, a), (d, float((got - a).abs().max()))
    finally:
        lib.ring_destroy(ring)


def test_row_padded_layout_converters_and_row_copies(lib):
    """lbm_geom.row_pitch ("row padding, hidden behind the ABI"): the pitched converters map the reference's node
    ((r * C) + c) * 9 + q to q * plane_stride + r * row_pitch + c and back bit for bit, touch no padding, and
    lbm_lattice_copy_rows moves row ranges between a padded ghost-row lattice and dense planes"""
    import ctypes as ct
    R, C, P, G = 37, 50, 64, 3
    rng = np.random.default_rng(5)
    f = rng.standard_normal((R, C, 9))
    aos = torch.from_numpy(f).to("cuda:0")
    plane = R * P + 24
    soa = torch.full((9 * plane,), 777.0, dtype=torch.float64, device="cuda:0")
    lib.aos_to_soa_pitched(_ptr(soa), _ptr(aos), R, C, 9, ct.c_longlong(plane), P, None)
    torch.cuda.synchronize()
    view = soa.cpu().numpy()
    for q in (0, 4, 8):
        assert np.array_equal(view[q * plane:q * plane + R * P].reshape(R, P)[:, :C], f[..., q])
    assert int((view == 777.0).sum()) == 9 * plane - 9 * R * C          # the padding is untouched
    back = torch.empty_like(aos)
    lib.soa_to_aos_pitched(_ptr(back), _ptr(soa), R, C, 9, ct.c_longlong(plane), P, None)
    torch.cuda.synchronize()
    assert torch.equal(back, aos)
    # rows [5, 25) of the padded planes -> rows [-2, 18) of a ghost-row lattice with another pitch -> dense planes
    src_g = pylbm.Geom(R, C, 0, plane, P)
    ghosted = pylbm.Geom(R, C, G, (R + 2 * G) * 58, 58)
    lat = torch.zeros(9 * (R + 2 * G) * 58, dtype=torch.float64, device="cuda:0")
    lib.lattice_copy_rows(_ptr(lat), ct.byref(ghosted), -2, _ptr(soa), ct.byref(src_g), 5, 20, None)
    dense_g = pylbm.Geom(20, C, 0)
    dense = torch.empty((9, 20, C), dtype=torch.float64, device="cuda:0")
    lib.lattice_copy_rows(_ptr(dense), ct.byref(dense_g), 0, _ptr(lat), ct.byref(pylbm.Geom(R, C, G, (R + 2 * G) * 58, 58)), -2, 20, None)
    torch.cuda.synchronize()
    assert np.array_equal(dense.cpu().numpy(), np.moveaxis(f[5:25], -1, 0))
    with pytest.raises(pylbm.LbmError, match="row_pitch"):
        lib.lattice_copy_rows(_ptr(dense), ct.byref(pylbm.Geom(20, C, 0, 0, C - 2)), 0, _ptr(lat), ct.byref(ghosted), 0, 20, None)
