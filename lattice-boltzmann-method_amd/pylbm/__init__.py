"""ctypes binding of liblbm_hip.so (the C ABI in include/lbm_hip.h) for tests and bench.py.

The product's host side is C++ (lattice-boltzmann-method_amd/include/lbm/*.hpp, mirroring the
reference's headers); this module is the thin Python door onto the same C ABI.  There is no
CPU fallback: if the HIP library is missing, import fails loudly.
"""
import ctypes as ct
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.dirname(_HERE)
REPO = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("LBM_HIP_LIB", os.path.join(PKG_DIR, "lib", "liblbm_hip.so"))
HEADER = os.path.join(REPO, "include", "lbm_hip.h")

EDGE_PERIODIC, EDGE_HALO, EDGE_BOUNCE_BACK, EDGE_SPECULAR, EDGE_ABB_VELOCITY, EDGE_WRAP_NOSHIFT = range(6)
HALO_TWO_PHASE = -3  # lbm_halo_pack / _unpack depth code of the colour-gradient step (21 rows)
MODEL_BGK, MODEL_KBC = 0, 1
FORM_DEFAULT, FORM_REFERENCE_ORDER, FORM_REASSOCIATED = 0, 1, 2   # field `form` of the parameter structs (LBM_FORM_*)
RING_DEFAULT, RING_RCCL, RING_IPC = -1, 0, 1
CG_PART_FRAME, CG_PART_INNER = 1, 2   # lbm_cg_step_fused_part (LBM_CG_PART_*)

_dp = ct.POINTER(ct.c_double)


class Geom(ct.Structure):
    _fields_ = [("R", ct.c_int), ("C", ct.c_int), ("ghost", ct.c_int),
                ("plane_stride", ct.c_longlong), ("row_pitch", ct.c_int)]

    def __init__(self, R=0, C=0, ghost=0, plane_stride=0, row_pitch=0):
        super().__init__(R, C, ghost, plane_stride, row_pitch)


class Bc(ct.Structure):
    _fields_ = [("row_lo", ct.c_int), ("row_hi", ct.c_int), ("col_lo", ct.c_int),
                ("col_hi", ct.c_int), ("pressure_rows", ct.c_int), ("rho_inlet", ct.c_double),
                ("rho_outlet", ct.c_double), ("uw_r", ct.c_double), ("uw_c", ct.c_double)]

    def __init__(self, row_lo=0, row_hi=0, col_lo=0, col_hi=0, pressure_rows=0, rho_inlet=1.0,
                 rho_outlet=1.0, uw_r=0.0, uw_c=0.0):
        super().__init__(row_lo, row_hi, col_lo, col_hi, pressure_rows, rho_inlet, rho_outlet, uw_r, uw_c)

    @staticmethod
    def periodic():
        return Bc()


class BgkParams(ct.Structure):
    _fields_ = [("omega", ct.c_double), ("incompressible", ct.c_int), ("delta_form", ct.c_int),
                ("force_mode", ct.c_int), ("force_r", ct.c_double), ("force_c", ct.c_double),
                ("guo_a", ct.c_double), ("guo_b", ct.c_double), ("form", ct.c_int)]

    def __init__(self, omega=1.0, incompressible=0, delta_form=0, force=None, guo=(1.0 / 3.0, 1.0 / 9.0), form=FORM_DEFAULT):
        """force=(Fr, Fc): the body force of test/gravity_test.cpp (u += F, Guo-type source); form: FORM_*"""
        if force is None:
            super().__init__(omega, incompressible, delta_form, 0, 0.0, 0.0, 0.0, 0.0, form)
        else:
            super().__init__(omega, incompressible, 1, 1, force[0], force[1], guo[0], guo[1], form)


class KbcParams(ct.Structure):
    _fields_ = [("s2", ct.c_double), ("form", ct.c_int)]

    def __init__(self, s2=1.0, form=FORM_DEFAULT):
        super().__init__(s2, form)


class CgColour(ct.Structure):
    _fields_ = [("rho_0", ct.c_double), ("alpha", ct.c_double), ("nu", ct.c_double),
                ("beta", ct.c_double)]


class CgParams(ct.Structure):
    _fields_ = [("red", CgColour), ("blue", CgColour), ("sigma", ct.c_double),
                ("gravity_r", ct.c_double), ("gravity_c", ct.c_double), ("add_source", ct.c_int),
                ("delta", ct.c_double), ("form", ct.c_int)]


class LbmError(RuntimeError):
    pass


def declared_symbols(header=HEADER):
    """Every function the C header declares (used by the ABI-completeness test)."""
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z0-9_]+)\s*\(", txt)))


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise LbmError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C lattice-boltzmann-method_amd/csrc` (no CPU fallback exists)")
    lib = ct.CDLL(path)
    lib.lbm_last_error_string.restype = ct.c_char_p
    lib.lbm_default_plane_pad.restype = ct.c_longlong
    lib.lbm_solver_block_launches.restype = ct.c_longlong
    lib.lbm_slab_ibm_msg_doubles.restype = ct.c_longlong
    lib.lbm_cg_solver_pair_launches.restype = ct.c_longlong
    lib.lbm_slab_pressure_msg_doubles.restype = ct.c_longlong
    return lib


class Lib:
    """Checked calls: any non-zero status raises LbmError with the library's message."""

    def __init__(self, path=LIB_PATH):
        self.raw = load_library(path)

    def __getattr__(self, name):
        fn = getattr(self.raw, "lbm_" + name)

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise LbmError(f"lbm_{name} -> {rc}: {self.raw.lbm_last_error_string().decode()}")
            return rc

        return call

    def device_count(self):
        return self.raw.lbm_device_count()

    def default_plane_pad(self, R, C):
        return int(self.raw.lbm_default_plane_pad(R, C))

    def reset_tuning(self):
        for k in (b"variant", b"nt", b"grid_cap", b"block", b"rows", b"xcd_swizzle", b"tb_rows", b"tb_block", b"tb_order", b"sw_rows", b"sw_waves", b"solver_depth", b"cg_fused", b"cg_tile", b"cg_xcd", b"kbc_fast", b"kbc_depth", b"bgk_fast", b"cg_strip", b"cg_rows", b"cg_split", b"sw_split", b"solver_depth_walls", b"ibm_depth", b"ibm_gate", b"bgk_fast_delta", b"pressure_depth", b"halo_grid", b"cg_strip2", b"cg_rows2", b"sw_pair", b"sw_pf2", b"cg_strip_xcd", b"cg_merge", b"cg_frame_beside", b"ring_period", b"ibm_step_opt", b"ibm_step_split", b"ibm_step_chain", b"ibm_box", b"ibm_box_overlap", b"bg_priority", b"ibm_chain_kernel", b"ibm_chain_wgs", b"ibm_box_sole", b"sw_ldsring", b"ring_ipc_timeout_ms", b"cg_big", b"cg_big_xcd", b"ring_cg_parts", b"ring_ipc_force_cached", b"ring_ipc_cached_ok", b"row_pad", b"cg_walk_rows", b"cg_walk_tile_xcd", b"sw_cols2"):
            self.set_tuning(k, -1)


def _ptr(t):
    """torch CUDA tensor (float64, contiguous) or int address -> double*"""
    if t is None:
        return ct.cast(None, _dp)
    if isinstance(t, int):
        return ct.cast(t, _dp)
    # strided lattice views (padded planes) are fine: the geometry carries the plane stride
    assert t.stride(-1) == 1 and str(t.dtype) == "torch.float64", (t.dtype, t.stride())
    return ct.cast(t.data_ptr(), _dp)


def _hptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def _stream(s):
    return ct.c_void_p(0 if s is None else int(s))


class Solver:
    """Python face of lbm_solver (single block).  numpy AoS in/out, reference layout."""

    def __init__(self, lib, model, R, C, params, bc=None, stream=None):
        self.lib, self.R, self.C = lib, R, C
        self.g = Geom(R, C, 0)
        self.bc = bc if bc is not None else Bc.periodic()
        self.params = params
        self.h = ct.c_void_p()
        lib.solver_create(ct.byref(self.h), model, ct.byref(self.g), ct.byref(self.bc),
                          ct.byref(params), _stream(stream))

    def close(self):
        if self.h:
            self.lib.solver_destroy(self.h)
            self.h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_f(self, f):
        f = np.ascontiguousarray(f, dtype=np.float64)
        assert f.shape == (self.R, self.C, 9)
        self.lib.solver_set_f_aos(self.h, _hptr(f))

    def set_moments(self, rho, u):
        """KBC: the first iteration collides on these held moments (ulbm_poiseuille.cpp:85-86)"""
        rho = np.ascontiguousarray(rho, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert rho.shape == (self.R, self.C) and u.shape == (self.R, self.C, 2)
        self.lib.solver_set_moments_aos(self.h, _hptr(rho), _hptr(u))

    def get_f(self):
        f = np.empty((self.R, self.C, 9))
        self.lib.solver_get_f_aos(self.h, _hptr(f))
        return f

    def step(self, n, record_moments=False):
        self.lib.solver_step(self.h, int(n), int(bool(record_moments)))

    def moments(self):
        rho = np.empty((self.R, self.C))
        u = np.empty((self.R, self.C, 2))
        self.lib.solver_get_moments_aos(self.h, _hptr(rho), _hptr(u))
        return rho, u

    def sync(self):
        self.lib.solver_sync(self.h)

    def checkpoint_save(self, path):
        self.lib.solver_checkpoint_save(self.h, str(path).encode())

    def checkpoint_load(self, path):
        self.lib.solver_checkpoint_load(self.h, str(path).encode())

    def attach_ibm(self, ibm, guo_a=1.0 / 3.0, guo_b=1.0 / 9.0):
        """defaults: the (1/3, 1/9) the cylinder driver uses (cylinder_test.cpp:66-67)"""
        self._ibm = ibm  # keep alive
        self.lib.solver_attach_ibm(self.h, ibm.h, ct.c_double(guo_a), ct.c_double(guo_b))

    def lattices(self):
        a, b, g = _dp(), _dp(), Geom()
        self.lib.solver_lattices(self.h, ct.byref(a), ct.byref(b), ct.byref(g))
        return ct.cast(a, ct.c_void_p).value, ct.cast(b, ct.c_void_p).value, g


def cg_params(red=(3.0, 0.7, 0.04, 0.7), blue=(1.0, 0.1, 0.04, -0.7), sigma=0.1, gravity=6.25e-6,
              delta=0.1, gravity_c=0.0, add_source=1, form=FORM_DEFAULT):
    """[red]/[blue] of mrtcg-rayleigh-taylor-gamma3.toml as (rho_0, alpha, nu, beta); sigma and
    gravity are this build's recorded choices for the keys the shipped TOML lacks."""
    return CgParams(CgColour(*red), CgColour(*blue), sigma, gravity, gravity_c, add_source, delta, form)


class CgSolver:
    """Python face of lbm_cg_solver: the two-phase driver loop, numpy AoS in/out."""

    def __init__(self, lib, R, C, params, bc=None, stream=None):
        self.lib, self.R, self.C, self.params = lib, R, C, params
        self.g = Geom(R, C, 0)
        self.h = ct.c_void_p()
        lib.cg_solver_create(ct.byref(self.h), ct.byref(self.g), ct.byref(bc) if bc is not None else None,
                             ct.byref(params), _stream(stream))

    def close(self):
        if self.h:
            self.lib.cg_solver_destroy(self.h)
            self.h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, f_r, f_b, rho_r, rho_b, u):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (f_r, f_b, rho_r, rho_b, u)]
        self.lib.cg_solver_set_state(self.h, *[_hptr(a) for a in arrs])

    def step(self, n):
        self.lib.cg_solver_step(self.h, int(n))

    def get_state(self):
        R, C = self.R, self.C
        out = dict(f_r=np.empty((R, C, 9)), f_b=np.empty((R, C, 9)), rho_r=np.empty((R, C)),
                   rho_b=np.empty((R, C)), u=np.empty((R, C, 2)), psi=np.empty((R, C)),
                   s_nu=np.empty((R, C)))
        self.lib.cg_solver_get_state(self.h, *[_hptr(out[k]) for k in
                                               ("f_r", "f_b", "rho_r", "rho_b", "u", "psi", "s_nu")])
        return out


class Ibm:
    """Python face of lbm_ibm (immersed boundary, stationary markers)."""

    def __init__(self, lib, x, y, X, Y, m_max=5, row_offset=0):
        self.lib = lib
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.h = ct.c_void_p()
        lib.ibm_create_slab(ct.byref(self.h), _hptr(x), _hptr(y), len(x), int(m_max), int(X), int(Y),
                            int(row_offset))

    def roi(self):
        v = [ct.c_int() for _ in range(4)]
        self.lib.ibm_roi(self.h, *[ct.byref(i) for i in v])
        return tuple(i.value for i in v)

    def surface_force(self, stream=None):
        out = np.zeros(2)
        self.lib.ibm_surface_force(self.h, _hptr(out), _stream(stream))
        return out

    def close(self):
        if self.h:
            self.lib.ibm_destroy(self.h)
            self.h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlabIbm:
    """Python face of lbm_slab_ibm: one BGK row slab of a domain with an immersed boundary, advanced in
    blocks of `depth` steps (capi_slab_ibm.hip).  The transport between slabs is the caller's."""

    def __init__(self, lib, geom, slab_row0, rows_global, bc_global, prm, depth, x, y, m_max=5,
                 guo=(1.0 / 3.0, 1.0 / 9.0)):
        self.lib, self.geom = lib, geom
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.h = ct.c_void_p()
        lib.slab_ibm_create(ct.byref(self.h), ct.byref(geom), int(slab_row0), int(rows_global), ct.byref(bc_global),
                            ct.byref(prm), int(depth), _hptr(x), _hptr(y), len(x), int(m_max),
                            ct.c_double(guo[0]), ct.c_double(guo[1]))
        v = [ct.c_int() for _ in range(5)]
        lib.slab_ibm_info(self.h, *[ct.byref(i) for i in v])
        self.owner, self.straddle_prev, self.straddle_next, self.b0, self.b1 = (i.value for i in v)
        self.msg_doubles = int(lib.raw.lbm_slab_ibm_msg_doubles(self.h))

    def prime_counts(self, side):
        a, b = ct.c_longlong(), ct.c_longlong()
        self.lib.slab_ibm_prime_counts(self.h, int(side), ct.byref(a), ct.byref(b))
        return a.value, b.value

    def surface_force(self):
        out = np.zeros(2)
        self.lib.slab_ibm_surface_force(self.h, _hptr(out), None)
        return out

    def close(self):
        if self.h:
            self.lib.slab_ibm_destroy(self.h)
            self.h = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
