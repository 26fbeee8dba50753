"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for

  * ``oracle/_build/liblbm_oracle.so``  (the CPU restatement, ``Oracle``), and
  * ``oracle/_ref/libref_lbm.so``       (the unmodified reference on CPU libtorch, ``Ref``).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Arrays use the reference layout: f[R,C,9], rho[R,C], u[R,C,2], float64.
"""
import ctypes as ct
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.environ.get("LBM_ORACLE_LIB", os.path.join(_HERE, "_build", "liblbm_oracle.so"))  # (the sanitizer build: make san)
REF_SO = os.path.join(_HERE, "_ref", "libref_lbm.so")
REF_DIR = os.path.join(_HERE, "_ref")

_dp = ct.POINTER(ct.c_double)


def _p(a):
    if a is None:
        return ct.cast(None, _dp)
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return ORACLE_SO


def build_ref():
    """Needs /root/reference (build container only)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref_build"), "-j4"])
    return REF_SO


class ColourParams(ct.Structure):
    _fields_ = [("rho_0", ct.c_double), ("alpha", ct.c_double), ("nu", ct.c_double),
                ("beta", ct.c_double)]


class CgParams(ct.Structure):
    _fields_ = [("R", ct.c_int), ("C", ct.c_int), ("red", ColourParams), ("blue", ColourParams),
                ("sigma", ct.c_double), ("g_r", ct.c_double), ("g_c", ct.c_double),
                ("add_source", ct.c_int), ("delta", ct.c_double)]


class HptParams(ct.Structure):
    _fields_ = [("H", ct.c_int), ("W", ct.c_int), ("T", ct.c_int), ("omega", ct.c_double),
                ("u_max", ct.c_double), ("rho_inlet", ct.c_double), ("rho_outlet", ct.c_double),
                ("check_convergence", ct.c_int)]


class IbmMarkers(ct.Structure):
    _fields_ = [("n_markers", ct.c_int), ("x", _dp), ("y", _dp), ("m_max", ct.c_int)]


def cg_params(R, C, red=(3.0, 0.7, 0.04, 0.7), blue=(1.0, 0.1, 0.04, -0.7), sigma=0.1,
              gravity=6.25e-6, delta=0.1, gravity_c=0.0, add_source=1):
    """Defaults: [red]/[blue] of mrtcg-rayleigh-taylor-gamma3.toml (rho_0, alpha, nu, beta);
    sigma/gravity are the builder's recorded choices."""
    return CgParams(R, C, ColourParams(*red), ColourParams(*blue), sigma, gravity, gravity_c, add_source, delta)


def hpt_params(H=21, W=21, T=8301, check_convergence=1):
    """test/horizontal_poiseuille_test.cpp:50-67 with H, W, T replaceable."""
    tau = np.sqrt(3.0 / 16.0) + 0.5
    omega = 1.0 / tau
    u_max = 1.030985714e-1
    nu = (2.0 * tau - 1.0) / 6.0
    p_grad = 8.0 * nu * u_max / (W * W)
    rho_outlet = 1.0
    rho_inlet = 3.0 * (H - 1) * p_grad + rho_outlet
    return HptParams(H, W, T, omega, u_max, rho_inlet, rho_outlet, check_convergence)


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build_oracle()
        self.lib = ct.CDLL(path)
        self.lib.orc_max_threads.restype = ct.c_int
        self.lib.orc_hpt_run.restype = ct.c_int
        self.lib.orc_gravity_run.restype = ct.c_int

    def max_threads(self):
        return self.lib.orc_max_threads()

    def set_threads(self, n):
        self.lib.orc_set_threads(int(n))

    # -- solver:: -----------------------------------------------------------
    def calc_rho(self, f):
        f = _c(f); R, C, _ = f.shape
        rho = np.empty((R, C)); self.lib.orc_calc_rho(_p(rho), _p(f), R, C); return rho

    def calc_u(self, f, rho):
        f = _c(f); rho = _c(rho); R, C, _ = f.shape
        u = np.empty((R, C, 2)); self.lib.orc_calc_u(_p(u), _p(f), _p(rho), R, C); return u

    def calc_incomp_u(self, f):
        f = _c(f); R, C, _ = f.shape
        u = np.empty((R, C, 2)); self.lib.orc_calc_incomp_u(_p(u), _p(f), R, C); return u

    def equilibrium(self, u, rho):
        u = _c(u); rho = _c(rho); R, C, _ = u.shape
        e = np.empty((R, C, 9)); self.lib.orc_equilibrium(_p(e), _p(u), _p(rho), R, C); return e

    def incomp_equilibrium(self, u, rho):
        u = _c(u); rho = _c(rho); R, C, _ = u.shape
        e = np.empty((R, C, 9)); self.lib.orc_incomp_equilibrium(_p(e), _p(u), _p(rho), R, C)
        return e

    def collision(self, f, feq, omega):
        f = _c(f); feq = _c(feq); R, C, _ = f.shape
        o = np.empty((R, C, 9))
        self.lib.orc_collision(_p(o), _p(f), _p(feq), ct.c_double(omega), R, C); return o

    def advect(self, f):
        f = _c(f); R, C, _ = f.shape
        g = np.empty((R, C, 9)); self.lib.orc_advect(_p(g), _p(f), R, C); return g

    def bgk_periodic_steps(self, f, omega, nsteps, incompressible=False):
        f = _c(f).copy(); R, C, _ = f.shape
        rho = np.empty((R, C)); u = np.empty((R, C, 2))
        self.lib.orc_bgk_periodic_steps(_p(f), _p(rho), _p(u), R, C, ct.c_double(omega),
                                        int(incompressible), int(nsteps))
        return f, rho, u

    # -- drivers ------------------------------------------------------------
    def hpt_run(self, p):
        H, W = p.H, p.W
        f = np.empty((H, W, 9)); u = np.empty((H, W, 2)); rho = np.empty((H, W))
        l2 = ct.c_double(0.0)
        steps = self.lib.orc_hpt_run(ct.byref(p), _p(f), _p(u), _p(rho), ct.byref(l2))
        return dict(steps=steps, f=f, u=u, rho=rho, l2=l2.value)

    def sbt_run(self, H, W, T, omega, rho_inlet, rho_outlet):
        f = np.empty((H, W, 9)); u = np.empty((H, W, 2)); rho = np.empty((H, W))
        self.lib.orc_sbt_run(H, W, int(T), ct.c_double(omega), ct.c_double(rho_inlet),
                             ct.c_double(rho_outlet), _p(f), _p(u), _p(rho))
        return dict(f=f, u=u, rho=rho)

    def free_stream_steps(self, f, omega, uw, nsteps):
        f = _c(f).copy(); X, Y, _ = f.shape
        u = np.zeros((X, Y, 2)); rho = np.ones((X, Y))
        self.lib.orc_free_stream_steps(_p(f), _p(u), _p(rho), X, Y, ct.c_double(omega),
                                       ct.c_double(uw), int(nsteps))
        return f, u, rho

    def gravity_run(self, H, W, T, omega, Fr, Fc, rho_inlet=1.0, rho_outlet=1.0, check_convergence=True):
        f = np.empty((H, W, 9)); u = np.empty((H, W, 2)); rho = np.empty((H, W))
        steps = self.lib.orc_gravity_run(H, W, int(T), ct.c_double(omega), ct.c_double(Fr),
                                         ct.c_double(Fc), ct.c_double(rho_inlet),
                                         ct.c_double(rho_outlet), int(check_convergence), _p(f), _p(u), _p(rho))
        return dict(steps=steps, f=f, u=u, rho=rho)

    def ddm_run(self, H, W, T, omega, rho_inlet, rho_outlet):
        out = {k: np.empty(s) for k, s in dict(fA=(H, W, 9), fB=(H, W, 9), uA=(H, W, 2),
                                                 uB=(H, W, 2), rhoA=(H, W), rhoB=(H, W)).items()}
        self.lib.orc_ddm_run(H, W, T, ct.c_double(omega), ct.c_double(rho_inlet),
                             ct.c_double(rho_outlet), _p(out["fA"]), _p(out["fB"]),
                             _p(out["uA"]), _p(out["uB"]), _p(out["rhoA"]), _p(out["rhoB"]))
        return out

    # -- KBC ----------------------------------------------------------------
    def kbc_equilibrium(self, m0, m1, use_zero_u2=False):
        m0 = _c(m0); m1 = _c(m1); R, C = m0.shape
        e = np.empty((R, C, 9))
        self.lib.orc_kbc_equilibrium(_p(e), _p(m0), _p(m1), int(use_zero_u2), R, C); return e

    def kbc_collide(self, f, m0, m1, s2):
        f = _c(f); m0 = _c(m0); m1 = _c(m1); R, C = m0.shape
        o = np.empty((R, C, 9)); g = np.empty((R, C))
        self.lib.orc_kbc_collide(_p(o), _p(f), _p(m0), _p(m1), ct.c_double(s2), R, C, _p(g))
        return o, g

    def kbc_steps(self, f, m0, m1, s2, nsteps):
        f = _c(f).copy(); m0 = _c(m0).copy(); m1 = _c(m1).copy(); R, C = m0.shape
        self.lib.orc_kbc_steps(_p(f), _p(m0), _p(m1), R, C, ct.c_double(s2), int(nsteps))
        return f, m0, m1

    def ddl_run(self, L, nsteps, omega=None, Fr=3e-3):
        """test/decompose_domain_loop.cpp: returns {"f": [A..D], "rho": [...], "u": [...]}"""
        if omega is None:
            omega = 1.0 / (np.sqrt(3.0 / 16.0) + 0.5)
        shapes = [(L, L // 4), (L // 4, L // 2), (L, L // 4), (L // 4, L // 2)]
        f = [np.empty(s + (9,)) for s in shapes]
        rho = [np.empty(s) for s in shapes]
        u = [np.empty(s + (2,)) for s in shapes]
        P4 = ct.POINTER(ct.c_double) * 4
        self.lib.orc_ddl_run(int(L), int(nsteps), ct.c_double(omega), ct.c_double(Fr), *[_p(a) for a in f],
                             P4(*[_p(a) for a in rho]), P4(*[_p(a) for a in u]))
        return dict(f=f, rho=rho, u=u)

    def sed_steps(self, X, Y, omega, u_in, nsteps, w_s=3e-3, Cw=1e-3, state=None):
        """test/rectangle_sedimentation_test.cpp loop; state = dict(f, g, rho, u, C) or None (start)"""
        if state is None:
            st = dict(f=np.empty((X, Y, 9)), g=np.empty((X, Y, 9)), rho=np.empty((X, Y)),
                      u=np.empty((X, Y, 2)), C=np.empty((X, Y)))
        else:
            st = {k: _c(v).copy() for k, v in state.items()}
        self.lib.orc_sed_steps(X, Y, ct.c_double(omega), ct.c_double(u_in), ct.c_double(w_s), ct.c_double(Cw),
                               int(state is None), int(nsteps), _p(st["f"]), _p(st["g"]), _p(st["rho"]),
                               _p(st["u"]), _p(st["C"]))
        return st

    def upo_steps(self, H, W, s2, rho_inlet, rho_outlet, nsteps, state=None):
        """test/ulbm_poiseuille.cpp loop; state = (f, m0, m1) to continue, None = the driver's start"""
        if state is None:
            f, m0, m1 = np.zeros((H, W, 9)), np.ones((H, W)), np.zeros((H, W, 2))
        else:
            f, m0, m1 = (_c(a).copy() for a in state)
        self.lib.orc_upo_steps(_p(f), _p(m0), _p(m1), H, W, ct.c_double(s2), ct.c_double(rho_inlet),
                               ct.c_double(rho_outlet), int(state is None), int(nsteps))
        return f, m0, m1

    def kbc_shear_init(self, R, C, u_max=0.02, alpha=80.0, delta=0.05):
        m0 = np.empty((R, C)); m1 = np.empty((R, C, 2))
        self.lib.orc_kbc_shear_init(_p(m0), _p(m1), R, C, ct.c_double(u_max), ct.c_double(alpha),
                                    ct.c_double(delta))
        return m0, m1

    # -- differential ---------------------------------------------------------
    def diff_x(self, psi):
        psi = _c(psi); R, C = psi.shape
        o = np.empty((R, C)); self.lib.orc_diff_x(_p(o), _p(psi), R, C); return o

    def diff_y(self, psi):
        psi = _c(psi); R, C = psi.shape
        o = np.empty((R, C)); self.lib.orc_diff_y(_p(o), _p(psi), R, C); return o

    # -- colour gradient --------------------------------------------------------
    def cg_init(self, p):
        R, C = p.R, p.C
        s = dict(f_r=np.empty((R, C, 9)), f_b=np.empty((R, C, 9)), rho_r=np.empty((R, C)),
                 rho_b=np.empty((R, C)), u=np.empty((R, C, 2)))
        self.lib.orc_cg_init(ct.byref(p), _p(s["f_r"]), _p(s["f_b"]), _p(s["rho_r"]),
                             _p(s["rho_b"]), _p(s["u"]))
        return s

    def cg_init_droplet(self, p):
        R, C = p.R, p.C
        s = dict(f_r=np.empty((R, C, 9)), f_b=np.empty((R, C, 9)), rho_r=np.empty((R, C)),
                 rho_b=np.empty((R, C)), u=np.empty((R, C, 2)))
        self.lib.orc_cg_init_droplet(ct.byref(p), _p(s["f_r"]), _p(s["f_b"]), _p(s["rho_r"]),
                                     _p(s["rho_b"]), _p(s["u"]))
        return s

    def cg_steps(self, p, state, nsteps, want_col=False):
        R, C = p.R, p.C
        s = {k: _c(v).copy() for k, v in state.items() if k in ("f_r", "f_b", "rho_r", "rho_b", "u")}
        psi = np.empty((R, C)); snu = np.empty((R, C))
        col_r = np.empty((R, C, 9)) if want_col else None
        col_b = np.empty((R, C, 9)) if want_col else None
        self.lib.orc_cg_steps(ct.byref(p), _p(s["f_r"]), _p(s["f_b"]), _p(s["rho_r"]),
                              _p(s["rho_b"]), _p(s["u"]), int(nsteps), _p(psi), _p(snu),
                              _p(col_r), _p(col_b))
        s.update(psi=psi, s_nu=snu)
        if want_col:
            s.update(col_r=col_r, col_b=col_b)
        return s

    # -- IBM ------------------------------------------------------------------
    @staticmethod
    def _markers(x, y, m_max=5):
        x = _c(x); y = _c(y)
        return IbmMarkers(len(x), _p(x), _p(y), m_max), (x, y)

    def ibm_roi(self, x, y):
        mk, keep = self._markers(x, y)
        v = [ct.c_int() for _ in range(4)]
        self.lib.orc_ibm_roi(ct.byref(mk), *[ct.byref(i) for i in v])
        return tuple(i.value for i in v)

    def ibm_force(self, x, y, u, rho, m_max=5):
        mk, keep = self._markers(x, y, m_max)
        u = _c(u); rho = _c(rho); X, Y = rho.shape
        r0, r1, c0, c1 = self.ibm_roi(x, y)
        F = np.empty((r1 - r0, c1 - c0, 2))
        self.lib.orc_ibm_force(ct.byref(mk), _p(u), _p(rho), X, Y, _p(F))
        return F

    def cylinder_steps(self, x, y, f, omega, u_in, nsteps, m_max=5):
        mk, keep = self._markers(x, y, m_max)
        f = _c(f).copy(); X, Y, _ = f.shape
        u = np.zeros((X, Y, 2)); rho = np.ones((X, Y)); Fs = np.zeros(2)
        self.lib.orc_cylinder_steps(ct.byref(mk), _p(f), _p(u), _p(rho), X, Y,
                                    ct.c_double(omega), ct.c_double(u_in), int(nsteps), _p(Fs))
        return f, u, rho, Fs


class Ref:
    """The unmodified reference (oracle/_ref/libref_lbm.so).  Present in the build container
    after `make -C oracle ref`; on the GPU box only as the prebuilt file."""

    def __init__(self, path=REF_SO):
        self.lib = ct.CDLL(path)
        self.lib.ref_num_threads.restype = ct.c_int

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def num_threads(self):
        return self.lib.ref_num_threads()

    def set_threads(self, n):
        self.lib.ref_set_num_threads(int(n))

    def calc_rho(self, f):
        f = _c(f); R, C, _ = f.shape
        rho = np.empty((R, C)); self.lib.ref_calc_rho(_p(rho), _p(f), R, C); return rho

    def calc_u(self, f, rho):
        f = _c(f); rho = _c(rho); R, C, _ = f.shape
        u = np.empty((R, C, 2)); self.lib.ref_calc_u(_p(u), _p(f), _p(rho), R, C); return u

    def calc_incomp_u(self, f):
        f = _c(f); R, C, _ = f.shape
        u = np.empty((R, C, 2)); self.lib.ref_calc_incomp_u(_p(u), _p(f), R, C); return u

    def equilibrium(self, u, rho):
        u = _c(u); rho = _c(rho); R, C, _ = u.shape
        e = np.empty((R, C, 9)); self.lib.ref_equilibrium(_p(e), _p(u), _p(rho), R, C); return e

    def incomp_equilibrium(self, u, rho):
        u = _c(u); rho = _c(rho); R, C, _ = u.shape
        e = np.empty((R, C, 9)); self.lib.ref_incomp_equilibrium(_p(e), _p(u), _p(rho), R, C)
        return e

    def collision(self, f, feq, omega):
        f = _c(f); feq = _c(feq); R, C, _ = f.shape
        o = np.empty((R, C, 9))
        self.lib.ref_collision(_p(o), _p(f), _p(feq), ct.c_double(omega), R, C); return o

    def advect(self, f):
        f = _c(f); R, C, _ = f.shape
        g = np.empty((R, C, 9)); self.lib.ref_advect(_p(g), _p(f), R, C); return g

    def bgk_periodic_steps(self, f, omega, nsteps, incompressible=False):
        f = _c(f).copy(); R, C, _ = f.shape
        rho = np.empty((R, C)); u = np.empty((R, C, 2))
        self.lib.ref_bgk_periodic_steps(_p(f), _p(rho), _p(u), R, C, ct.c_double(omega),
                                        int(incompressible), int(nsteps))
        return f, rho, u

    def kbc_steps(self, f, m0, m1, s2, nsteps, init_from_moments=False, want_coll=False):
        m0 = _c(m0).copy(); m1 = _c(m1).copy(); R, C = m0.shape
        f = np.zeros((R, C, 9)) if f is None else _c(f).copy()
        coll = np.empty((R, C, 9)) if want_coll else None
        self.lib.ref_kbc_steps(_p(f), _p(m0), _p(m1), R, C, ct.c_double(s2),
                               int(init_from_moments), int(nsteps), _p(coll))
        return (f, m0, m1, coll) if want_coll else (f, m0, m1)

    def upo_steps(self, H, W, s2, rho_inlet, rho_outlet, nsteps, state=None):
        if state is None:
            f, m0, m1 = np.zeros((H, W, 9)), np.ones((H, W)), np.zeros((H, W, 2))
        else:
            f, m0, m1 = (_c(a).copy() for a in state)
        self.lib.ref_upo_steps(_p(f), _p(m0), _p(m1), H, W, ct.c_double(s2), ct.c_double(rho_inlet),
                               ct.c_double(rho_outlet), int(state is None), int(nsteps))
        return f, m0, m1

    def diff_x(self, psi):
        psi = _c(psi); R, C = psi.shape
        o = np.empty((R, C)); self.lib.ref_diff_x(_p(o), _p(psi), R, C); return o

    def diff_y(self, psi):
        psi = _c(psi); R, C = psi.shape
        o = np.empty((R, C)); self.lib.ref_diff_y(_p(o), _p(psi), R, C); return o
