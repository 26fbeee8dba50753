"""TEST INFRASTRUCTURE: a SECOND, independent restatement of the two reference files that cannot
be compiled here (they include toml++): test/mrtcg_rayleigh_taylor.cpp (+ src/colour.cpp,
src/differential.cpp) and src/ibm.cpp (+ the loop body of test/cylinder_test.cpp).

Unlike oracle/lbm_oracle.cpp (per-node C++ loops) this file follows the reference statement by
statement with the SAME tensor operations (matmul, conv2d with replicate padding, where /
masked_fill, slice assignment), on CPU torch float64.  The two restatements were written from the
source separately; tests/test_oracle_crosscheck.py requires the two to agree to rounding.  This does
not pin the oracle to the reference (only a compiled reference could) but it makes a transcription
slip in either one visible.
"""
import math

import torch
import torch.nn.functional as F

torch.set_default_dtype(torch.float64)

# test/mrtcg_rayleigh_taylor.cpp:130-178
M = torch.tensor([[1, 1, 1, 1, 1, 1, 1, 1, 1], [-4, -1, -1, -1, -1, 2, 2, 2, 2], [4, -2, -2, -2, -2, 1, 1, 1, 1],
                  [0, 1, 0, -1, 0, 1, -1, -1, 1], [0, -2, 0, 2, 0, 1, -1, -1, 1], [0, 0, 1, 0, -1, 1, 1, -1, -1],
                  [0, 0, -2, 0, 2, 1, 1, -1, -1], [0, 1, -1, 1, -1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, -1, 1, -1]],
                 dtype=torch.float64)
Mi = (1.0 / 36.0) * torch.tensor(
    [[4, -4, 4, 0, 0, 0, 0, 0, 0], [4, -1, -2, 6, -6, 0, 0, 9, 0], [4, -1, -2, 0, 0, 6, -6, -9, 0],
     [4, -1, -2, -6, 6, 0, 0, 9, 0], [4, -1, -2, 0, 0, -6, 6, -9, 0], [4, 2, 1, 6, 3, 6, 3, 0, 9],
     [4, 2, 1, -6, -3, 6, 3, 0, -9], [4, 2, 1, -6, -3, -6, -3, 0, 9], [4, 2, 1, 6, 3, -6, -3, 0, -9]],
    dtype=torch.float64)
B = torch.tensor([-4.0 / 27.0] + [2.0 / 27.0] * 4 + [5.0 / 108.0] * 4)
W = torch.tensor([4.0 / 9.0] + [1.0 / 9.0] * 4 + [1.0 / 36.0] * 4)
E = torch.tensor([[0.0, 1.0, 0.0, -1.0, 0.0, 1.0, -1.0, -1.0, 1.0],
                  [0.0, 0.0, 1.0, 0.0, -1.0, 1.0, 1.0, -1.0, -1.0]])
unit_E = E / torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0, math.sqrt(2), math.sqrt(2), math.sqrt(2), math.sqrt(2)])


class Differential:  # src/differential.hpp:9-40, src/differential.cpp:3-33
    xi = (1.0 / 5040.0) * torch.tensor([[1.0, 32.0, 84.0, 32.0, 1.0], [32.0, 448.0, 960.0, 448.0, 32.0],
                                         [84.0, 960.0, 0.0, 960.0, 84.0], [32.0, 448.0, 960.0, 448.0, 32.0],
                                         [1.0, 32.0, 84.0, 32.0, 1.0]])
    ky = torch.tensor([[-2.0, -1.0, 0.0, 1.0, 2.0]] * 5)
    kx = -torch.tensor([[2.0] * 5, [1.0] * 5, [0.0] * 5, [-1.0] * 5, [-2.0] * 5])

    def _conv(self, psi, k):
        w = (self.xi * k).reshape(1, 1, 5, 5)
        p = F.pad(psi.reshape(1, 1, *psi.shape[:2]), (2, 2, 2, 2), mode="replicate")
        return F.conv2d(p, w).squeeze(0).squeeze(0)

    def x(self, psi):
        return self._conv(psi, self.kx)

    def y(self, psi):
        return self._conv(psi, self.ky)


D = Differential()


class Colour:  # src/colour.cpp:11-64
    def __init__(self, rho_0, alpha, nu, beta, R, C):
        self.rho_0, self.alpha, self.nu, self.beta = rho_0, alpha, nu, beta
        self.cs2 = 3.0 * (1.0 - alpha) / 5.0
        a, b = 0.2 * (1.0 - alpha), 0.05 * (1.0 - alpha)
        self.phi = torch.tensor([alpha, a, a, a, a, b, b, b, b])
        self.eta = (1.0 + 0.5 * (3.0 * self.cs2 - 1.0) * (3.0 * E.mul(E).sum(0) - 4.0)).expand(R, C, 9).clone()
        self.rho = torch.zeros(R, C, 1)
        self.adv_f = torch.zeros(R, C, 9)
        self.C = torch.zeros(R, C, 9)


def relaxation_function(red, blue, delta):  # :34-101 (the "tau" arguments receive omegas)
    r_omega, b_omega = 1.0 / (0.5 + red.nu / red.cs2), 1.0 / (0.5 + blue.nu / blue.cs2)
    s1 = 2.0 * r_omega * b_omega / (r_omega + b_omega)
    s2 = 2.0 * (r_omega - s1) / delta
    s3 = -s2 / (2.0 * delta)
    t2 = 2.0 * (s1 - b_omega) / delta
    t3 = t2 / (2.0 * delta)

    def ev(s_nu, psi_):
        psi = psi_.squeeze(-1).clone()
        s_nu = s_nu.masked_fill(psi > delta, r_omega)
        s_nu = torch.where((delta >= psi) * (psi > 0.0), s1 + s2 * psi + s3 * psi * psi, s_nu)
        s_nu = torch.where((0.0 >= psi) * (psi >= -delta), s1 + t2 * psi + t3 * psi * psi, s_nu)
        return s_nu.masked_fill(psi < -delta, b_omega)
    return ev


def eval_equilibrium(k_rho, k_phi, k_eta, u):  # :233-247
    return k_rho * (k_phi + W.mul(3.0 * u.matmul(E) * k_eta + 9.0 * u.matmul(E).pow(2)
                                  - 3.0 * u.mul(u).sum(-1).unsqueeze(-1)))


def advect(f):  # src/solver.cpp:76-131, slice by slice
    g = f.clone()
    g[1:, :, 1] = f[:-1, :, 1]; g[0, :, 1] = f[-1, :, 1]
    g[:, 1:, 2] = f[:, :-1, 2]; g[:, 0, 2] = f[:, -1, 2]
    g[:-1, :, 3] = f[1:, :, 3]; g[-1, :, 3] = f[0, :, 3]
    g[:, :-1, 4] = f[:, 1:, 4]; g[:, -1, 4] = f[:, 0, 4]
    g[1:, 1:, 5] = f[:-1, :-1, 5]; g[0, 1:, 5] = f[-1, :-1, 5]; g[1:, 0, 5] = f[:-1, -1, 5]; g[0, 0, 5] = f[-1, -1, 5]
    g[:-1, 1:, 6] = f[1:, :-1, 6]; g[-1, 1:, 6] = f[0, :-1, 6]; g[:-1, 0, 6] = f[1:, -1, 6]; g[-1, 0, 6] = f[0, -1, 6]
    g[:-1, :-1, 7] = f[1:, 1:, 7]; g[-1, :-1, 7] = f[0, 1:, 7]; g[:-1, -1, 7] = f[1:, 0, 7]; g[-1, -1, 7] = f[0, 0, 7]
    g[1:, :-1, 8] = f[:-1, 1:, 8]; g[0, :-1, 8] = f[-1, 1:, 8]; g[1:, -1, 8] = f[:-1, 0, 8]; g[0, -1, 8] = f[-1, 0, 8]
    return g


def apply_boundary_conditions(adv_f, col_f):  # :495-533
    adv_f[1:-1, 0, 2] = col_f[1:-1, -1, 2]; adv_f[1:-1, 0, 5] = col_f[1:-1, -1, 5]; adv_f[1:-1, 0, 6] = col_f[1:-1, -1, 6]
    adv_f[1:-1, -1, 4] = col_f[1:-1, 0, 4]; adv_f[1:-1, -1, 8] = col_f[1:-1, 0, 8]; adv_f[1:-1, -1, 7] = col_f[1:-1, 0, 7]
    adv_f[-1, :, 3] = col_f[-1, :, 1]; adv_f[-1, :, 7] = col_f[-1, :, 5]; adv_f[-1, :, 6] = col_f[-1, :, 8]
    adv_f[0, :, 1] = col_f[0, :, 3]; adv_f[0, :, 5] = col_f[0, :, 7]; adv_f[0, :, 8] = col_f[0, :, 6]


def cg_run(R, C, red, blue, sigma, gravity, steps, delta=0.1, droplet=False):
    """main() of test/mrtcg_rayleigh_taylor.cpp:364-477 for `steps` iterations; droplet=True: main()
    of test/mrtcg_static_droplet.cpp:396-528 (sigmoid droplet, Fg = (0, gravity) shifts u only,
    u shifted before the initial equilibrium, no source term; sigma = 0.1 there)."""
    r, b = Colour(*red, R, C), Colour(*blue, R, C)
    middle = R / 2.0  # init_rho_cosine :182-210
    for k, invert in ((r, True), (b, False)):
        for c in range(C):
            s = middle - 0.1 * C * math.cos(2.0 * 3.141592 * c / C)
            for rr in range(R):
                if droplet:  # init_rho_droplet, mrtcg_static_droplet.cpp:182-204
                    d = math.sqrt((rr - middle) * (rr - middle) + (c - middle) * (c - middle))
                    sg = 1.0 / (1.0 + math.exp(-(1.0 * (d - 25.0))))
                    ans = 1.0 - sg if invert else sg
                else:
                    ans = (1.0 if rr < s else 0.0) if invert else (1.0 if rr >= s else 0.0)
                k.rho[rr, c, 0] = k.rho_0 * ans
    relax = relaxation_function(r, b, delta)
    u = torch.zeros(R, C, 2)
    s_nu = torch.zeros(R, C)
    S = torch.diagflat(torch.tensor([0.0, 1.25, 1.14, 0.0, 1.6, 0.0, 1.6, 0.0, 0.0])).repeat(R, C, 1, 1)
    Fg = torch.tensor([[0.0], [gravity]]) if droplet else torch.tensor([[gravity], [0.0]])
    rho = r.rho + b.rho
    if droplet:
        u = u + 0.5 * Fg.t() / rho  # mrtcg_static_droplet.cpp:457
    r.adv_f = eval_equilibrium(r.rho, r.phi, r.eta, u)
    b.adv_f = eval_equilibrium(b.rho, b.phi, b.eta, u)
    phase = torch.zeros(R, C, 1)
    for _ in range(steps):
        r_equ = eval_equilibrium(r.rho, r.phi, r.eta, u)
        b_equ = eval_equilibrium(b.rho, b.phi, b.eta, u)
        phase = (r.rho / r.rho_0 - b.rho / b.rho_0) / (r.rho / r.rho_0 + b.rho / b.rho_0)  # :212-225
        s_nu = relax(s_nu, phase)
        for k in (r, b):  # update_C :320-336
            DxQx = D.x((1.8 * k.alpha - 0.8) * k.rho.squeeze(-1) * u[..., 0])
            DyQy = D.y((1.8 * k.alpha - 0.8) * k.rho.squeeze(-1) * u[..., 1])
            k.C[..., 1] = 3.0 * (1.0 - 0.5 * 1.25) * (DxQx + DyQy)
            k.C[..., 7] = (1.0 - 0.5 * s_nu) * (DxQx - DyQy)
        S[..., 7, 7] = s_nu  # update_S :227-231
        S[..., 8, 8] = s_nu
        om1 = []
        for k, equ in ((r, r_equ), (b, b_equ)):  # eval_mrt_operator :249-261
            om1.append(Mi.matmul(S.matmul(M.matmul((equ - k.adv_f).unsqueeze(-1))) + k.C.unsqueeze(-1)).squeeze(-1))
        grad = torch.stack([D.x(phase.squeeze(-1)), D.y(phase.squeeze(-1))], dim=-1)  # :443
        grad_norm = torch.sqrt(grad[..., 0].pow(2) + grad[..., 1].pow(2)).unsqueeze(-1)
        xi = 0.5 * grad_norm * (W.mul((grad.matmul(E) / (1e-20 + grad_norm)).pow(2)) - B)  # :290-300
        A = 4.5 * sigma * s_nu.unsqueeze(-1)
        om2 = A * xi
        kappa = (r.rho * b.rho * grad.matmul(unit_E) * (r.rho * r.phi + b.rho * b.phi)) / (rho.pow(2) * (1e-20 + grad_norm))
        total_f = r.adv_f + om1[0] + om2 + b.adv_f + om1[1] + om2
        om3_r = r.rho * total_f / rho + r.beta * kappa
        om3_b = b.rho * total_f / rho + b.beta * kappa
        force = (1 - 0.5 * s_nu.unsqueeze(-1)) * ((3.0 + 9.0 * u.matmul(E)) * Fg.t().matmul(E) - 3.0 * u.matmul(Fg)) * W
        r_col, b_col = (om3_r, om3_b) if droplet else (om3_r + force, om3_b + force)
        r.adv_f, b.adv_f = advect(r_col), advect(b_col)
        apply_boundary_conditions(r.adv_f, r_col)
        apply_boundary_conditions(b.adv_f, b_col)
        r.rho = r.adv_f.sum(-1).unsqueeze(-1)
        b.rho = b.adv_f.sum(-1).unsqueeze(-1)
        rho = r.rho + b.rho
        u = (r.adv_f + b.adv_f).matmul(E.t()) / rho  # solver::calc_u
        u = u + 0.5 * Fg.t() / rho
    return dict(f_r=r.adv_f.numpy(), f_b=b.adv_f.numpy(), rho_r=r.rho.squeeze(-1).numpy(),
                rho_b=b.rho.squeeze(-1).numpy(), u=u.numpy(), psi=phase.squeeze(-1).numpy(), s_nu=s_nu.numpy())


# ---------------------------------------------------------------------------------------------
# src/ibm.cpp
# ---------------------------------------------------------------------------------------------
STENCIL = torch.tensor([[0, 1, 2, 3] * 4, [0] * 4 + [1] * 4 + [2] * 4 + [3] * 4], dtype=torch.float64)  # :11-13


def calc_phi_scalar(r_):  # :39-45
    r = abs(r_)
    if r <= 1:
        return 0.125 * (3.0 - 2.0 * r + math.sqrt(1.0 + 4.0 * r - 4.0 * r * r))
    if r <= 2:
        return 0.125 * (5.0 - 2.0 * r - math.sqrt(-7.0 + 12.0 * r - 4.0 * r * r))
    return 0.0


class Marker:  # :15-57
    def __init__(self, x, y):
        r = torch.tensor([[x], [y]])
        s = r - (STENCIL + torch.floor(r) - 1.0)
        a = torch.tensor([[calc_phi_scalar(float(v)) for v in row] for row in s])
        self.phi = a[0] * a[1]
        self.rows = slice(int(math.floor(x)) - 1, int(math.floor(x)) + 3)
        self.cols = slice(int(math.floor(y)) - 1, int(math.floor(y)) + 3)


class Ibm:  # :59-190
    def __init__(self, xs, ys, m_max=5):
        r_min = min(int(math.floor(x) - 2) for x in xs); r_max = max(int(math.floor(x) + 2) for x in xs)
        c_min = min(int(math.floor(y) - 2) for y in ys); c_max = max(int(math.floor(y) + 2) for y in ys)
        self.rows, self.cols, self.m_max = slice(r_min, r_max + 1), slice(c_min, c_max + 1), m_max
        self.markers = [Marker(x - r_min, y - c_min) for x, y in zip(xs, ys)]
        self.shape = (r_max - r_min + 1, c_max - c_min + 1)

    def eulerian_force_density(self, u_0, rho_0):
        u = u_0[self.rows, self.cols].clone()
        rho = rho_0[self.rows, self.cols].clone()
        Fa = torch.zeros(*self.shape, 2, self.m_max)
        for n in range(1, self.m_max):
            for m in self.markers:
                box = u[m.rows, m.cols].clone().reshape(16, 2)
                uj = m.phi.matmul(box)
                rhoj = m.phi.matmul(rho[m.rows, m.cols].reshape(16, 1))
                fj = -2.0 * rhoj * uj
                Fa[m.rows, m.cols, :, n] += m.phi.reshape(4, 4, 1) * fj.unsqueeze(1).t()
            u += 0.5 * Fa[..., n] / rho
        return Fa.sum(3)


def cylinder_run(xs, ys, X, Y, omega, u_in, steps):
    """test/cylinder_test.cpp:85-164 for `steps` iterations."""
    c, Ew = E, W
    u = torch.zeros(X, Y, 2); u[..., 0] = u_in
    rho = torch.ones(X, Y, 1)
    f_adve = (rho + 3.0 * u.matmul(c)) * Ew  # incomp_equilibrium
    ib = Ibm(xs, ys)
    ics2, ics4 = 1.0 / 3.0, 1.0 / 9.0
    u_w = torch.zeros(Y, 2); u_w[:, 0] = u_in
    opp = [0, 3, 4, 1, 2, 7, 8, 5, 6]
    F_s = torch.zeros(2)
    for _ in range(steps):
        rho = f_adve.sum(-1, keepdim=True)
        u = f_adve.matmul(c.t()) / rho
        u_u = (u * u).sum(-1, keepdim=True)
        c_u = u.matmul(c)
        f_equi = (rho * (1.0 + 3.0 * c_u + 4.5 * c_u.pow(2) - 1.5 * u_u)) * Ew
        equi_populations = -omega * (f_adve - f_equi)
        Fr = ib.eulerian_force_density(u, rho)
        F_s = Fr.reshape(-1, 2).sum(0)
        u_roi = u[ib.rows, ib.cols]
        S = ((1 - 0.5 * omega) * ((ics2 + ics4 * u_roi.matmul(c)) * Fr.matmul(c) - ics2 * (u_roi * Fr).sum(2).unsqueeze(2)) * Ew)
        f_coll = f_adve + equi_populations
        f_coll[ib.rows, ib.cols] += S
        f_adve = advect(f_coll)
        abb = (2.0 + 9.0 * u_w.matmul(c).pow(2.0) - 3.0 * u_w.mul(u_w).sum(1).unsqueeze(1)) * Ew
        for row in (0, -1):
            for q in range(1, 9):
                f_adve[row, :, opp[q]] = -f_coll[row, :, q] + abb[:, q]
        f_adve[:, -1, 4] = f_coll[:, -1, 2]; f_adve[:, -1, 7] = f_coll[:, -1, 6]; f_adve[:, -1, 8] = f_coll[:, -1, 5]
        f_adve[:, 0, 2] = f_coll[:, 0, 4]; f_adve[:, 0, 5] = f_coll[:, 0, 8]; f_adve[:, 0, 6] = f_coll[:, 0, 7]
    return f_adve.numpy(), u.numpy(), rho.squeeze(-1).numpy(), F_s.numpy()
