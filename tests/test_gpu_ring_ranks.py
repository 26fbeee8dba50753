"""The C++ slab ring (csrc/capi_ring.hip) across REAL process boundaries on the one GPU of the box: 2 and 3
ranks, each a process of its own (tests/ring_rank.py, started before it touches the GPU), joined by the
peer-mapped transport (LBM_RING_IPC: hipIpcMemHandle_t windows + sequence words, capi_ring_ipc.hip) -- RCCL
refuses two ranks on one device.  Every wrapper the multi-GPU drivers use is held BITWISE to the single
block: lbm_ring_bgk_step (5 steps per launch, one exchange per launch and per two launches, mixed depths,
wall columns), lbm_ring_kbc_step, lbm_ring_cg_step, lbm_ring_ibm_start / lbm_ring_bgk_block_ibm with the
cylinder on the seam, lbm_ring_pressure_start / lbm_ring_bgk_block_pressure.
Reference contract: the block binding of test/decompose_domain.cpp:181-187 and the cross-block pressure rows
of :50-73 (two blocks there; n slabs with D ghost rows here)."""
import ctypes as ct
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
from gpu_util import bits_equal, dev, download_aos, ulp_diff, upload_soa  # noqa: E402
from pylbm import _ptr  # noqa: E402
from pyoracle import hpt_params  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
IPC = 1
W9 = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


def hexof(s):
    return bytes(s).hex()


def run_ranks(lib, tmp_path, case, n, cfg, arrays, timeout=300):
    """start n rank processes on `case`, wait, return their outputs (a failing rank ends the others)"""
    work = str(tmp_path)
    cfg = dict(cfg, transport=IPC)
    json.dump(cfg, open(os.path.join(work, "cfg.json"), "w"))
    ident = (ct.c_ubyte * 128)()
    lib.ring_unique_id_ex(ident, IPC)
    open(os.path.join(work, "id.bin"), "wb").write(bytes(ident))
    for k, a in arrays.items():
        np.save(os.path.join(work, k + ".npy"), a)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(n):
        log = open(os.path.join(work, f"rank{r}.log"), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.join(HERE, "ring_rank.py"), case, str(r), str(n), work],
                                       stdout=log, stderr=subprocess.STDOUT, env=env), log))
    t0, failed = time.time(), None
    while any(p.poll() is None for p, _ in procs):
        bad = [r for r, (p, _) in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.time() - t0 > timeout:
            failed = f"rank(s) {bad} failed" if bad else f"timed out after {timeout} s"
            for p, _ in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p, log in procs:
        p.wait()
        log.close()
    bad = [r for r, (p, _) in enumerate(procs) if p.returncode != 0]
    if failed or bad:
        logs = "\n".join(f"--- rank {r} (rc {procs[r][0].returncode}) ---\n" + open(os.path.join(work, f"rank{r}.log")).read()[-3000:]
                         for r in range(n))
        raise AssertionError(f"{case}: {failed or bad}\n{logs}")
    return [np.load(os.path.join(work, f"out_{r}.npz")) for r in range(n)]


def perturbed_rest(H, W, seed):
    rng = np.random.default_rng(seed)
    f = np.empty((H, W, 9))
    f[...] = W9
    return f * (1 + 0.01 * rng.standard_normal((H, W, 9)))


# ---- BGK / KBC box -------------------------------------------------------------------------------------------------
def box_reference(lib, model, f0, bc, prm, total):
    """whole box on one block: collide-only launch, then `total` single steps (ghost 0, wrap inside the block);
    returns (p0, p_final) as SoA numpy"""
    H, C, _ = f0.shape
    flat = pylbm.Geom(H, C, 0)
    pre = upload_soa(lib, f0)
    a, b = torch.empty_like(pre), torch.empty_like(pre)
    coll = lib.bgk_collide if model == "bgk" else lib.kbc_collide
    step = lib.bgk_stream_collide if model == "bgk" else lib.kbc_stream_collide
    coll(_ptr(a), _ptr(pre), ct.byref(flat), ct.byref(bc), ct.byref(prm), None, None, None)
    torch.cuda.synchronize()
    p0 = a.cpu().numpy()
    for _ in range(total):
        step(_ptr(b), _ptr(a), ct.byref(flat), ct.byref(bc), ct.byref(prm), 0, H, None, None, None)
        a, b = b, a
    torch.cuda.synchronize()
    return p0, a.cpu().numpy()


@pytest.mark.parametrize("n,ghost,depths,walls", [
    (2, 5, [5, 5, 5], 0),            # one exchange per launch
    (2, 10, [5, 5, 5, 5, 5], 0),     # one exchange per two launches, ending inside a period
    (3, 10, [5, 5, 5, 5], 0),
    (2, 10, [2, 2, 2, 5, 3], 0),     # depths that use the ghost rows up unevenly: the deeper launch refreshes first
    (3, 5, [5, 4, 1, 5], 1),         # bounce-back columns: complete ghost rows travel
    (2, 10, [5, 5, 5], 1),
])
def test_bgk_ring_across_processes_equals_one_block(lib, tmp_path, n, ghost, depths, walls):
    R, C = 96, 160
    f0 = perturbed_rest(R * n, C, seed=n + ghost + len(depths))
    bc = pylbm.Bc()
    if walls:
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    prm = pylbm.BgkParams(1.2, 0)
    p0, want = box_reference(lib, "bgk", f0, bc, prm, sum(depths))
    outs = run_ranks(lib, tmp_path, "bgk", n, dict(R=R, C=C, ghost=ghost, depths=depths, walls=walls, bc=hexof(bc), prm=hexof(prm)),
                     dict(p0=p0))
    got = np.concatenate([o["P"] for o in outs], axis=1)
    assert bits_equal(got, want), ulp_diff(got, want)


@pytest.mark.parametrize("n,ghost,depths", [(2, 3, [3, 3, 3]), (3, 6, [3, 3, 3, 3, 3]), (2, 6, [2, 2, 3, 1, 3])])
def test_kbc_ring_across_processes_equals_one_block(lib, tmp_path, n, ghost, depths):
    R, C = 96, 128
    H = R * n
    rng = np.random.default_rng(7)
    f0 = perturbed_rest(H, C, seed=3) * (1 + 0.002 * rng.standard_normal((H, C, 1)))
    bc = pylbm.Bc()
    prm = pylbm.KbcParams(1.0 / (0.5 + 3 * 1.7e-4))
    p0, want = box_reference(lib, "kbc", f0, bc, prm, sum(depths))
    outs = run_ranks(lib, tmp_path, "kbc", n, dict(R=R, C=C, ghost=ghost, depths=depths, bc=hexof(bc), prm=hexof(prm)), dict(p0=p0))
    got = np.concatenate([o["P"] for o in outs], axis=1)
    assert bits_equal(got, want), ulp_diff(got, want)


# ---- two-phase chain -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 3])
def test_cg_ring_across_processes_equals_one_block(lib, oracle, tmp_path, n):
    """lbm_ring_cg_step: the fused two-phase step on a CHAIN of slabs (3 ghost rows, both colours in one message);
    mrtcg_rayleigh_taylor.cpp:413-478 with its walls (:495-533) on the outer slabs"""
    import pyoracle
    R, C, steps = 48, 64, 9
    Rg = R * n
    s0 = oracle.cg_init(pyoracle.cg_params(Rg, C))
    pg = pylbm.cg_params()
    flat, bcf = pylbm.Geom(Rg, C, 0), pylbm.Bc()
    lib.raw.lbm_cg_default_bc(ct.byref(bcf))
    d = dev()
    f_r, f_b = upload_soa(lib, s0["f_r"]), upload_soa(lib, s0["f_b"])
    rr, rb, uu = upload_soa(lib, s0["rho_r"]), upload_soa(lib, s0["rho_b"]), upload_soa(lib, s0["u"])
    a = [torch.empty((9, Rg, C), dtype=torch.float64, device=d) for _ in range(2)]
    b = [torch.empty_like(a[0]) for _ in range(2)]
    lib.cg_collide(_ptr(a[0]), _ptr(a[1]), _ptr(f_r), _ptr(f_b), _ptr(rr), _ptr(rb), _ptr(uu), ct.byref(flat), ct.byref(bcf),
                   ct.byref(pg), None, None, None)
    torch.cuda.synchronize()
    p0 = [x.cpu().numpy() for x in a]
    for _ in range(steps):
        lib.cg_step_fused(_ptr(b[0]), _ptr(b[1]), _ptr(a[0]), _ptr(a[1]), ct.byref(flat), ct.byref(bcf), ct.byref(pg), 0, Rg,
                          None, None, None, None, None, None)
        a, b = b, a
    torch.cuda.synchronize()
    outs = run_ranks(lib, tmp_path, "cg", n, dict(R=R, C=C, steps=steps), dict(p0r=p0[0], p0b=p0[1]))
    for k, key in ((0, "Pr"), (1, "Pb")):
        got = np.concatenate([o[key] for o in outs], axis=1)
        assert bits_equal(got, a[k].cpu().numpy()), (key, ulp_diff(got, a[k].cpu().numpy()))


# ---- immersed boundary over a chain, the cylinder on a seam -----------------------------------------------------------
def circle(cx, cy, radius):
    k = int(np.ceil(2 * np.pi * radius))
    t = 2 * np.pi * np.arange(k) / k
    return cx + radius * np.cos(t), cy + radius * np.sin(t)


@pytest.mark.parametrize("n,cx,want_roles,cols", [
    (2, 128.3, [(1, 0, 1), (1, 1, 0)], "specular"),
    (3, 255.6, [(0, 0, 0), (1, 0, 1), (1, 1, 0)], "specular"),
    (3, 200.4, [(0, 0, 0), (1, 0, 0), (0, 0, 0)], "periodic"),   # non-owners beside an owner with periodic columns
])
def test_ibm_blocks_across_processes_equal_one_block(lib, oracle, tmp_path, n, cx, want_roles, cols):
    """lbm_ring_ibm_start + lbm_ring_bgk_block_ibm over a chain of slabs (cylinder_test.cpp:88-164 / ibm.cpp:158-190 over
    the block binding): co-owners swap the band's outer rows in place of the seam's halo, non-owners run the overlapped
    window schedule; populations and surface force equal the single block bit for bit"""
    X, Y, D, nb, radius = 128 * n, 96, 5, 3, 10.0
    omega, u_in = 1.0 / 0.55, 0.05
    x, y = circle(cx, Y / 2 + 0.21, radius)
    u0 = np.zeros((X, Y, 2)); u0[..., 0] = u_in
    f0 = oracle.incomp_equilibrium(u0, np.ones((X, Y)))
    prm = pylbm.BgkParams(omega, 0, 1, form=pylbm.FORM_REFERENCE_ORDER)
    edge = pylbm.EDGE_SPECULAR if cols == "specular" else pylbm.EDGE_PERIODIC
    bc = pylbm.Bc(row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY, col_lo=edge, col_hi=edge, uw_r=u_in)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, X, Y, prm, bc=bc)
    ibw = pylbm.Ibm(lib, x, y, X, Y)
    sv.attach_ibm(ibw)
    sv.set_f(f0)
    sv.step(1 + D * nb)
    want, Fw = sv.get_f(), ibw.surface_force()
    sv.close(); ibw.close()
    outs = run_ranks(lib, tmp_path, "ibm", n, dict(X=X, Y=Y, D=D, blocks=nb, bc=hexof(bc), prm=hexof(prm)),
                     dict(f0=upload_soa(lib, f0).cpu().numpy(), x=x, y=y))
    assert [tuple(int(v) for v in o["roles"]) for o in outs] == want_roles
    P = torch.from_numpy(np.concatenate([o["P"] for o in outs], axis=1)).to(dev())
    out = torch.empty_like(P)
    flat = pylbm.Geom(X, Y, 0)
    lib.stream(_ptr(out), _ptr(P), ct.byref(flat), ct.byref(bc), None)
    got = download_aos(lib, out)
    assert bits_equal(got, want), ulp_diff(got, want)
    for o in outs:
        if o["roles"][0]:
            assert np.array_equal(o["Fs"], Fw), (o["Fs"], Fw)


# ---- pressure-periodic rows over a periodic ring ------------------------------------------------------------------------
@pytest.mark.parametrize("case,n,D", [("poiseuille_bb", 2, 5), ("poiseuille_bb", 3, 5), ("gravity", 3, 4), ("periodic_cols", 2, 2)])
def test_pressure_rows_across_processes_equal_one_block(lib, tmp_path, case, n, D):
    """lbm_ring_pressure_start + lbm_ring_bgk_block_pressure: the virtual rows 0 / Rg-1 of
    horizontal_poiseuille_test.cpp:25-45 sit on the two end slabs of the periodic ring (decompose_domain.cpp:50-73
    puts them on two blocks the same way); start-up + 3 blocks == the single block stepped one step per launch"""
    R, W, nb = 64, 150, 3
    H = R * n
    p = hpt_params(H, W, 0)
    bc = pylbm.Bc(pressure_rows=1, rho_inlet=p.rho_inlet, rho_outlet=p.rho_outlet)
    prm = pylbm.BgkParams(p.omega, 1)
    if case == "poiseuille_bb":
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
    elif case == "gravity":
        bc.col_lo = bc.col_hi = pylbm.EDGE_BOUNCE_BACK
        bc.rho_inlet = bc.rho_outlet = 1.0
        prm = pylbm.BgkParams(p.omega, 1, force=(-0.0003, 0.0))
    f0 = perturbed_rest(H, W, seed=H + W + D)
    lib.set_tuning(b"pressure_depth", 1)
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, H, W, prm, bc=bc)
    sv.set_f(f0)
    sv.step(1 + nb * D)
    want = sv.get_f()
    sv.close()
    lib.set_tuning(b"pressure_depth", -1)
    outs = run_ranks(lib, tmp_path, "pressure", n, dict(R=R, W=W, D=D, blocks=nb, bc=hexof(bc), prm=hexof(prm)),
                     dict(f0=upload_soa(lib, f0).cpu().numpy()))
    P = torch.from_numpy(np.concatenate([o["P"] for o in outs], axis=1)).to(dev())
    out = torch.empty_like(P)
    flat = pylbm.Geom(H, W, 0)
    lib.stream(_ptr(out), _ptr(P), ct.byref(flat), ct.byref(bc), None)
    got = download_aos(lib, out)
    assert bits_equal(got, want), (case, n, D, ulp_diff(got, want))


@pytest.mark.parametrize("n", [2, 3])
def test_kbc_pressure_rows_across_processes_equal_one_block(lib, tmp_path, n):
    """lbm_ring_pressure_start_kbc + lbm_ring_bgk_block_pressure on KBC slabs (ulbm_poiseuille.cpp:36-58, :85-139 over the
    ring): held moments per rank, 12-plane start-up message across the pressure seam, 2-step blocks"""
    R, W, nb, D = 64, 128, 4, 2
    H = R * n
    nu = 1e-4
    s2 = 1.0 / (0.5 + 3.0 * nu)
    rin = 3.0 * (H - 1) * (8.0 * nu * 0.05 / (W * W)) + 1.0
    bc = pylbm.Bc(col_lo=pylbm.EDGE_BOUNCE_BACK, col_hi=pylbm.EDGE_BOUNCE_BACK, pressure_rows=1, rho_inlet=rin, rho_outlet=1.0)
    prm = pylbm.KbcParams(s2)
    rng = np.random.default_rng(4)
    m0 = 1.0 + 0.001 * rng.standard_normal((H, W))
    m1 = 0.001 * rng.standard_normal((H, W, 2))
    lib.set_tuning(b"pressure_depth", 1)
    sv = pylbm.Solver(lib, pylbm.MODEL_KBC, H, W, prm, bc=bc)
    sv.set_f(np.zeros((H, W, 9)))
    sv.set_moments(m0, m1)
    sv.step(1 + nb * D)
    want = sv.get_f()
    sv.close()
    lib.set_tuning(b"pressure_depth", -1)
    outs = run_ranks(lib, tmp_path, "pressure_kbc", n, dict(R=R, W=W, blocks=nb, bc=hexof(bc), prm=hexof(prm)),
                     dict(m0=m0, m1=np.ascontiguousarray(np.moveaxis(m1, -1, 0))))
    P = torch.from_numpy(np.concatenate([o["P"] for o in outs], axis=1)).to(dev())
    out = torch.empty_like(P)
    flat = pylbm.Geom(H, W, 0)
    lib.stream(_ptr(out), _ptr(P), ct.byref(flat), ct.byref(bc), None)
    got = download_aos(lib, out)
    assert bits_equal(got, want), (n, ulp_diff(got, want))


def test_a_rank_that_never_arrives_is_reported_not_waited_for_forever(lib, tmp_path):
    """the bounded device-side wait: rank 1 of 2 maps the windows and leaves; rank 0's launches drain and
    lbm_ring_status says which neighbour failed"""
    R, C = 64, 128
    f0 = perturbed_rest(2 * R, C, seed=1)
    bc, prm = pylbm.Bc(), pylbm.BgkParams(1.2, 0)
    p0, _ = box_reference(lib, "bgk", f0, bc, prm, 0)
    with pytest.raises(AssertionError, match="never delivered"):
        run_ranks(lib, tmp_path, "bgk", 2, dict(R=R, C=C, ghost=5, depths=[5], bc=hexof(bc), prm=hexof(prm),
                                                desert=1, tuning=dict(ring_ipc_timeout_ms=1500)), dict(p0=p0))


def test_a_queue_of_launches_behind_a_deserter_drains_in_one_time_limit(lib, tmp_path):
    """the failure of a bounded wait is sticky (capi_ring_ipc.hip): eight queued launch-steps (each with up to four waits)
    behind a neighbour that left take ONE time limit to drain, not 32 of them"""
    R, C = 64, 128
    f0 = perturbed_rest(2 * R, C, seed=1)
    bc, prm = pylbm.Bc(), pylbm.BgkParams(1.2, 0)
    p0, _ = box_reference(lib, "bgk", f0, bc, prm, 0)
    with pytest.raises(AssertionError, match="never delivered"):
        run_ranks(lib, tmp_path, "bgk", 2, dict(R=R, C=C, ghost=5, depths=[5] * 8, bc=hexof(bc), prm=hexof(prm),
                                                desert=1, tuning=dict(ring_ipc_timeout_ms=1500)), dict(p0=p0))
    drain = float(open(os.path.join(str(tmp_path), "drain_0.txt")).read())
    assert 1.0 < drain < 4.0, drain


def test_refused_uncached_window_fails_unless_accepted(lib, tmp_path):
    """the receive window is allocated uncached; a refusal (forced here by "ring_ipc_force_cached") fails the creation with
    a sentence, and with "ring_ipc_cached_ok" the ring runs on a cached window, says so and is still bitwise right"""
    R, C = 64, 128
    f0 = perturbed_rest(2 * R, C, seed=2)
    bc, prm = pylbm.Bc(), pylbm.BgkParams(1.2, 0)
    p0, want = box_reference(lib, "bgk", f0, bc, prm, 10)
    cfg = dict(R=R, C=C, ghost=5, depths=[5, 5], bc=hexof(bc), prm=hexof(prm))
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    with pytest.raises(AssertionError, match="uncached allocation of the receive window was refused"):
        run_ranks(lib, tmp_path / "a", "bgk", 2, dict(cfg, tuning=dict(ring_ipc_force_cached=1)), dict(p0=p0))
    outs = run_ranks(lib, tmp_path / "b", "bgk", 2, dict(cfg, expect_cached=1, tuning=dict(ring_ipc_force_cached=1, ring_ipc_cached_ok=1)),
                     dict(p0=p0))
    got = np.concatenate([o["P"] for o in outs], axis=1)
    assert bits_equal(got, want)
