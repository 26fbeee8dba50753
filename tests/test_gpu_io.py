"""GPU suite, SURVEY 8(f) row 3: asynchronous snapshots (.npy) and bitwise checkpoint / restart."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import pylbm  # noqa: E402
from gpu_util import bits_equal  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    lib = pylbm.Lib()
    assert lib.device_count() >= 1
    return lib


def state(oracle, R, C, seed):
    rng = np.random.default_rng(seed)
    rho = 1 + 0.01 * rng.standard_normal((R, C))
    u = 0.04 * rng.standard_normal((R, C, 2))
    return oracle.equilibrium(u, rho)


@pytest.mark.parametrize("model", ["bgk", "bgk-walls", "kbc"])
def test_checkpoint_restart_is_bitwise(lib, oracle, tmp_path, model):
    R, C = 80, 128
    f0 = state(oracle, R, C, 1)

    def make():
        if model == "kbc":
            return pylbm.Solver(lib, pylbm.MODEL_KBC, R, C, pylbm.KbcParams(1.9))
        bc = None
        if model == "bgk-walls":
            bc = pylbm.Bc(col_lo=pylbm.EDGE_BOUNCE_BACK, col_hi=pylbm.EDGE_BOUNCE_BACK,
                          row_lo=pylbm.EDGE_ABB_VELOCITY, row_hi=pylbm.EDGE_ABB_VELOCITY, uw_r=0.03)
        return pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(1.4, 0), bc=bc)
    a = make()
    a.set_f(f0)
    a.step(23)
    a.checkpoint_save(tmp_path / "ck.bin")
    a.step(17, record_moments=True)
    fa = a.get_f()
    rho_a, u_a = a.moments()
    b = make()
    b.checkpoint_load(tmp_path / "ck.bin")
    b.step(17, record_moments=True)
    assert bits_equal(b.get_f(), fa)
    rho_b, u_b = b.moments()
    assert bits_equal(rho_a, rho_b) and bits_equal(u_a, u_b)
    # a checkpoint of another shape is refused
    c = pylbm.Solver(lib, pylbm.MODEL_BGK, R + 1, C, pylbm.BgkParams(1.4, 0))
    with pytest.raises(pylbm.LbmError, match="holds model"):
        c.checkpoint_load(tmp_path / "ck.bin")
    for s in (a, b, c):
        s.close()


def test_async_snapshot_npy(lib, oracle, tmp_path):
    R, C = 96, 160
    sv = pylbm.Solver(lib, pylbm.MODEL_BGK, R, C, pylbm.BgkParams(1.2, 0))
    sv.set_f(state(oracle, R, C, 2))
    sn = ct.c_void_p()
    lib.snapshot_create(ct.byref(sn), sv.h)
    sv.step(10, record_moments=True)
    rho10, u10 = sv.moments()
    lib.snapshot_record(sn)
    sv.step(25)                       # keeps stepping while the snapshot drains
    lib.snapshot_write_npy(sn, str(tmp_path / "rho.npy").encode(), str(tmp_path / "u.npy").encode())
    assert bits_equal(np.load(tmp_path / "rho.npy"), rho10)
    assert bits_equal(np.load(tmp_path / "u.npy"), u10)
    step = ct.c_longlong()
    lib.snapshot_host(sn, None, None, ct.byref(step))
    assert step.value == 10
    lib.snapshot_destroy(sn)
    sv.close()


def test_two_phase_async_snapshot(lib, oracle, tmp_path):
    """lbm_cg_snapshot_*: rho_r, rho_b, u of the current two-phase state, copied out on a private
    stream while the solver keeps stepping; .npy files read back == get_state() at that step."""
    import ctypes as ct
    import pyoracle
    R, C = 96, 64
    po = pyoracle.cg_params(R, C)
    s0 = oracle.cg_init(po)
    sv = pylbm.CgSolver(lib, R, C, pylbm.cg_params())
    sv.set_state(s0["f_r"], s0["f_b"], s0["rho_r"], s0["rho_b"], s0["u"])
    sn = ct.c_void_p()
    lib.cg_snapshot_create(ct.byref(sn), sv.h)
    try:
        sv.step(7)
        want = sv.get_state()
        lib.cg_snapshot_record(sn)
        sv.step(5)                                  # keeps running while the copy is in flight
        lib.cg_snapshot_write_npy(sn, str(tmp_path / "rr.npy").encode(), str(tmp_path / "rb.npy").encode(),
                                  str(tmp_path / "u.npy").encode())
        step = ct.c_longlong()
        lib.cg_snapshot_host(sn, None, None, None, ct.byref(step))
        assert step.value == 7
        assert bits_equal(np.load(tmp_path / "rr.npy"), want["rho_r"])
        assert bits_equal(np.load(tmp_path / "rb.npy"), want["rho_b"])
        assert bits_equal(np.load(tmp_path / "u.npy"), want["u"])
        assert not bits_equal(sv.get_state()["rho_r"], want["rho_r"])   # the solver has moved on
    finally:
        lib.cg_snapshot_destroy(sn)
        sv.close()
