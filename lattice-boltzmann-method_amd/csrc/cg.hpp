// Colour-gradient MRT two-phase step (BASELINE config 4): device restatement of the loop body
// of test/mrtcg_rayleigh_taylor.cpp:431-477 together with src/colour.cpp and the 5x5 stencils
// of src/differential.{hpp,cpp}.  The reference runs ~400 ATen launches per step, a per-node
// 9x9 S tensor (81 doubles/node) and 10 padded conv2d calls; here a step is two kernels:
//
//   pass A  k_cg_stream_moments : stream both colours (pull + the driver's BCs), reduce to
//           rho_r, rho_b, u (= (sum f c)/rho + Fg/(2 rho))                 reads 18, writes 4
//   pass B  k_cg_stream_collide : LDS tile of (psi, Q_r, Q_b) with a +-2 halo built from the
//           4 macroscopic fields (replicate-clamped at the global edges); 5x5
//           derivatives from LDS; re-stream the 18 populations; MRT relaxation with the
//           constant M / M^-1 unrolled and the diagonal S in registers; perturbation,
//           recolouring, gravity source; write both colours               reads 18+4h, writes 18
//
// = 496 B/LUP + stencil halo ("two-pass" figure of SURVEY 8d).  Every non-standard formula of
// the driver is kept (Q5 column copy, Q6 equilibrium, Q7 unweighted source, Q11 omega blend).
#pragma once
#include "d2q9.hpp"

namespace lbm {

struct CgColour {
  double rho_0, alpha, beta, qcoef;  // qcoef = 1.8 alpha - 0.8 (update_C, :326-327)
  double phi[Q], eta[Q];             // src/colour.cpp:49-64
};
struct CgConsts {
  CgColour k[2];  // 0 = red, 1 = blue
  double sigma, gr, gc;  // Fg = (gr, gc)
  int add_source;
  double delta, r_omega, b_omega, s1, s2, s3, t2, t3;  // relaxation_function :34-101
  double unitx[Q], unity[Q];                           // unit_E :176-178
};

inline CgConsts make_cg_consts(const lbm_cg_params& p) {
  CgConsts c;
  const lbm_cg_colour* src[2] = {&p.red, &p.blue};
  double omega[2];
  for (int i = 0; i < 2; ++i) {
    CgColour& k = c.k[i];
    k.rho_0 = src[i]->rho_0;
    k.alpha = src[i]->alpha;
    k.beta = src[i]->beta;
    k.qcoef = 1.8 * k.alpha - 0.8;
    const double cs2 = 3.0 * (1.0 - k.alpha) / 5.0;  // colour.cpp:37
    const double a = 0.2 * (1.0 - k.alpha), b = 0.05 * (1.0 - k.alpha);
    const double ph[Q] = {k.alpha, a, a, a, a, b, b, b, b};
    for (int q = 0; q < Q; ++q) {
      k.phi[q] = ph[q];
      const double e2 = (double)(icx(q) * icx(q) + icy(q) * icy(q));
      k.eta[q] = 1.0 + 0.5 * (3.0 * cs2 - 1.0) * (3.0 * e2 - 4.0);  // colour.cpp:49-54
    }
    omega[i] = 1.0 / (0.5 + src[i]->nu / cs2);  // init_omega :57-58
  }
  c.sigma = p.sigma;
  c.gr = p.gravity_r;
  c.gc = p.gravity_c;
  c.add_source = p.add_source;
  c.delta = p.delta;
  c.r_omega = omega[0];
  c.b_omega = omega[1];
  c.s1 = 2.0 * c.r_omega * c.b_omega / (c.r_omega + c.b_omega);
  c.s2 = 2.0 * (c.r_omega - c.s1) / c.delta;
  c.s3 = -c.s2 / (2.0 * c.delta);
  c.t2 = 2.0 * (c.s1 - c.b_omega) / c.delta;
  c.t3 = c.t2 / (2.0 * c.delta);
  for (int q = 0; q < Q; ++q) {
    const double d = q < 5 ? 1.0 : std::sqrt(2);
    c.unitx[q] = (double)icx(q) / d;
    c.unity[q] = (double)icy(q) / d;
  }
  return c;
}

// eval_equilibrium, mrtcg_rayleigh_taylor.cpp:233-247 (9 (c.u)^2 - 3 u.u: SURVEY Q6)
__device__ __forceinline__ void cg_feq(double (&e)[Q], double rho_k, const CgColour& k, double ux,
                                       double uy) {
  const double uu = ux * ux + uy * uy;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double cu = ux * (double)icx(q) + uy * (double)icy(q);
    e[q] = rho_k * (k.phi[q] + wq(q) * (3.0 * cu * k.eta[q] + 9.0 * (cu * cu) - 3.0 * uu));
  }
}

__device__ __forceinline__ double cg_psi(const CgConsts& c, double rr, double rb) {  // :212-225
  return (rr / c.k[0].rho_0 - rb / c.k[1].rho_0) / (rr / c.k[0].rho_0 + rb / c.k[1].rho_0);
}
__device__ __forceinline__ double cg_snu(const CgConsts& c, double psi) {  // :84-100
  double v = 0.0;
  if (psi > c.delta) v = c.r_omega;
  if (c.delta >= psi && psi > 0.0) v = c.s1 + c.s2 * psi + c.s3 * psi * psi;
  if (0.0 >= psi && psi >= -c.delta) v = c.s1 + c.t2 * psi + c.t3 * psi * psi;
  if (psi < -c.delta) v = c.b_omega;
  return v;
}

// M and 36 M^-1 of the driver (:130-140, :146-156), compile-time tables
__device__ __forceinline__ constexpr double cg_M(int a, int q) {
  constexpr double M[Q][Q] = {{1, 1, 1, 1, 1, 1, 1, 1, 1},     {-4, -1, -1, -1, -1, 2, 2, 2, 2},
                              {4, -2, -2, -2, -2, 1, 1, 1, 1}, {0, 1, 0, -1, 0, 1, -1, -1, 1},
                              {0, -2, 0, 2, 0, 1, -1, -1, 1},  {0, 0, 1, 0, -1, 1, 1, -1, -1},
                              {0, 0, -2, 0, 2, 1, 1, -1, -1},  {0, 1, -1, 1, -1, 0, 0, 0, 0},
                              {0, 0, 0, 0, 0, 1, -1, 1, -1}};
  return M[a][q];
}
__device__ __forceinline__ constexpr double cg_Mi36(int q, int a) {
  constexpr double Mi[Q][Q] = {{4, -4, 4, 0, 0, 0, 0, 0, 0},    {4, -1, -2, 6, -6, 0, 0, 9, 0},
                               {4, -1, -2, 0, 0, 6, -6, -9, 0}, {4, -1, -2, -6, 6, 0, 0, 9, 0},
                               {4, -1, -2, 0, 0, -6, 6, -9, 0}, {4, 2, 1, 6, 3, 6, 3, 0, 9},
                               {4, 2, 1, -6, -3, 6, 3, 0, -9},  {4, 2, 1, -6, -3, -6, -3, 0, 9},
                               {4, 2, 1, 6, 3, -6, -3, 0, -9}};
  return Mi[q][a];
}
__device__ __forceinline__ constexpr double cg_B(int q) {  // :158-163
  return q == 0 ? -4.0 / 27.0 : (q < 5 ? 2.0 / 27.0 : 5.0 / 108.0);
}
// 5x5 isotropic stencil weights (xi / 5040, differential.hpp:9-16)
__device__ __forceinline__ constexpr double cg_xi(int i, int j) {
  constexpr double XI[5][5] = {{1.0, 32.0, 84.0, 32.0, 1.0},
                               {32.0, 448.0, 960.0, 448.0, 32.0},
                               {84.0, 960.0, 0.0, 960.0, 84.0},
                               {32.0, 448.0, 960.0, 448.0, 32.0},
                               {1.0, 32.0, 84.0, 32.0, 1.0}};
  return XI[i][j];
}

// Macroscopic fields of a slab carry CG_MG = 2 ghost rows per side when the populations carry
// ghost rows (lbm_geom.ghost = 3): pass A recomputes rho_r, rho_b, u on rows -2..R+1 from the
// populations' 3 ghost rows, so one population exchange per step feeds both the streaming and
// the 5x5 stencils.  Single block (ghost = 0): no macro ghost rows.
struct MacroIdx {
  int mg;   // macro ghost rows per side (0 or 2)
  int C;
  long n;   // doubles per macro plane = (R + 2*mg) * C
  __host__ __device__ long at(int r, int c) const { return (long)(r + mg) * C + c; }
};
inline MacroIdx make_macro_idx(const Geom& g) {
  const int mg = g.ghost ? 2 : 0;
  return MacroIdx{mg, g.C, (long)(g.R + 2 * mg) * g.C};
}
// rows on which the macroscopic fields exist / the stencil may read: clamped at GLOBAL edges
// only (replicate padding); a HALO edge continues into the neighbour's rows
__host__ __device__ inline int cg_row_lo(const Geom& g, const Bc& bc) {
  return (g.ghost && bc.row_lo == LBM_EDGE_HALO) ? -2 : 0;
}
__host__ __device__ inline int cg_row_hi(const Geom& g, const Bc& bc) {  // inclusive
  return (g.ghost && bc.row_hi == LBM_EDGE_HALO) ? g.R + 1 : g.R - 1;
}

// ---- pass A ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cg_stream_moments(
    double* __restrict__ rho_r, double* __restrict__ rho_b, double* __restrict__ u,
    const double* __restrict__ p_r, const double* __restrict__ p_b, Geom g, Bc bc, double grav_r,
    double grav_c, MacroIdx mi, int row_lo, int row_hi /* exclusive */) {
  const long n = (long)(row_hi - row_lo) * g.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = row_lo + (int)(i / g.C), c = (int)(i % g.C);
    double fr[Q], fb[Q];
    gather_bc(fr, p_r, g, bc, r, c);
    gather_bc(fb, p_b, g, bc, r, c);
    double rr, rb, jx, jy, ft[Q];
    BgkModel::moments(fr, rr, jx, jy);  // r.rho = r.adv_f.sum(-1), :472
    BgkModel::moments(fb, rb, jx, jy);
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] = fr[q] + fb[q];  // calc_u(u, r.adv_f + b.adv_f, rho) :476
    double dummy;
    BgkModel::moments(ft, dummy, jx, jy);
    const double rt = rr + rb;  // :474
    const long o = mi.at(r, c);
    rho_r[o] = rr;
    rho_b[o] = rb;
    u[o] = jx / rt + 0.5 * grav_r / rt;      // :477  u + 0.5 Fg^T / rho
    u[mi.n + o] = jy / rt + 0.5 * grav_c / rt;
  }
}

// ---- pass B ---------------------------------------------------------------------------------
#ifndef LBM_CG_WAVES
#define LBM_CG_WAVES 1  // waves per SIMD the collide kernel is register-budgeted for (measured)
#endif
constexpr int CG_TR = 8, CG_TC = 32;               // tile of nodes per 256-thread block
constexpr int CG_LR = CG_TR + 4, CG_LC = CG_TC + 4;  // with the +-2 stencil halo

template <bool FROM_POST, bool WITH_FIELDS>
__global__ __launch_bounds__(256, LBM_CG_WAVES) void k_cg_collide(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, const double* __restrict__ rho_r,
    const double* __restrict__ rho_b, const double* __restrict__ u, Geom g, Bc bc, CgConsts cc,
    double* __restrict__ psi_out, double* __restrict__ snu_out, MacroIdx mi, int row_begin,
    int row_end) {
  __shared__ double s_psi[CG_LR][CG_LC + 1];
  __shared__ double s_q[4][CG_LR][CG_LC + 1];  // Qx_r, Qy_r, Qx_b, Qy_b
  const int tiles_c = (g.C + CG_TC - 1) / CG_TC;
  const int r_base = row_begin + (blockIdx.x / tiles_c) * CG_TR, c_base = (blockIdx.x % tiles_c) * CG_TC;
  const long n = mi.n;
  const int rlo = cg_row_lo(g, bc), rhi = cg_row_hi(g, bc);
  // stage the stencil inputs; replicate padding = clamp (differential.cpp:5-9) at the edges of
  // the GLOBAL domain, neighbour rows across a slab seam
  for (int i = threadIdx.x; i < CG_LR * CG_LC; i += 256) {
    const int lr = i / CG_LC, lc = i % CG_LC;
    int gr = r_base + lr - 2, gc = c_base + lc - 2;
    gr = gr < rlo ? rlo : (gr > rhi ? rhi : gr);
    gc = gc < 0 ? 0 : (gc > g.C - 1 ? g.C - 1 : gc);
    const long o = mi.at(gr, gc);
    const double rr = rho_r[o], rb = rho_b[o], ux = u[o], uy = u[n + o];
    s_psi[lr][lc] = cg_psi(cc, rr, rb);
    s_q[0][lr][lc] = cc.k[0].qcoef * rr * ux;  // (1.8 alpha - 0.8) * rho_k * u_x  (:326)
    s_q[1][lr][lc] = cc.k[0].qcoef * rr * uy;
    s_q[2][lr][lc] = cc.k[1].qcoef * rb * ux;
    s_q[3][lr][lc] = cc.k[1].qcoef * rb * uy;
  }
  __syncthreads();
  const int tr = threadIdx.x / CG_TC, tc = threadIdx.x % CG_TC;
  const int r = r_base + tr, c = c_base + tc;
  if (r >= row_end || c >= g.C) return;

  // 5x5 cross-correlations, taps in (i, j) row-major order as conv2d lays them out.  The row loop is
  // NOT unrolled (the weights come from a constant table, same values as the folded constants):
  // fully unrolled, the scheduler hoists all 100 LDS reads above the collision and the kernel needs
  // 274 VGPRs + AGPR spills; the accumulation order per sum is unchanged, so results are too.
  double gx = 0.0, gy = 0.0, dxq[2] = {0.0, 0.0}, dyq[2] = {0.0, 0.0};
  {
    constexpr double kx[5][5] = {
        {(1.0 / 5040.0) * cg_xi(0, 0), (1.0 / 5040.0) * cg_xi(0, 1), (1.0 / 5040.0) * cg_xi(0, 2), (1.0 / 5040.0) * cg_xi(0, 3), (1.0 / 5040.0) * cg_xi(0, 4)},
        {(1.0 / 5040.0) * cg_xi(1, 0), (1.0 / 5040.0) * cg_xi(1, 1), (1.0 / 5040.0) * cg_xi(1, 2), (1.0 / 5040.0) * cg_xi(1, 3), (1.0 / 5040.0) * cg_xi(1, 4)},
        {(1.0 / 5040.0) * cg_xi(2, 0), (1.0 / 5040.0) * cg_xi(2, 1), (1.0 / 5040.0) * cg_xi(2, 2), (1.0 / 5040.0) * cg_xi(2, 3), (1.0 / 5040.0) * cg_xi(2, 4)},
        {(1.0 / 5040.0) * cg_xi(3, 0), (1.0 / 5040.0) * cg_xi(3, 1), (1.0 / 5040.0) * cg_xi(3, 2), (1.0 / 5040.0) * cg_xi(3, 3), (1.0 / 5040.0) * cg_xi(3, 4)},
        {(1.0 / 5040.0) * cg_xi(4, 0), (1.0 / 5040.0) * cg_xi(4, 1), (1.0 / 5040.0) * cg_xi(4, 2), (1.0 / 5040.0) * cg_xi(4, 3), (1.0 / 5040.0) * cg_xi(4, 4)}};
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
      const double di = (double)(i - 2);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const double wx = kx[i][j] * di;                  // d/d(row)  ("x")
        const double wy = kx[i][j] * (double)(j - 2);     // d/d(col)  ("y")
        if (i != 2) {
          gx += wx * s_psi[tr + i][tc + j];
          dxq[0] += wx * s_q[0][tr + i][tc + j];
          dxq[1] += wx * s_q[2][tr + i][tc + j];
        }
        if (j != 2) {
          gy += wy * s_psi[tr + i][tc + j];
          dyq[0] += wy * s_q[1][tr + i][tc + j];
          dyq[1] += wy * s_q[3][tr + i][tc + j];
        }
      }
    }
  }

  const long o = mi.at(r, c);
  const double rr = rho_r[o], rb = rho_b[o], ux = u[o], uy = u[n + o];
  const double rt = rr + rb;
  const double psi = s_psi[tr + 2][tc + 2];
  const double s_nu = cg_snu(cc, psi);
  const double S[Q] = {0.0, 1.25, 1.14, 0.0, 1.6, 0.0, 1.6, s_nu, s_nu};  // :384-386, :227-231

  const double gnorm = sqrt(gx * gx + gy * gy);  // :444-447
  const double A = 4.5 * cc.sigma * s_nu;        // :450
  const long lo = g.at(r, c);
  // Omega2 (perturbation, identical for both colours) first, then total_f accumulated colour by
  // colour in the driver's left-to-right order (:455)
  //   total_f = r.adv_f + r.omega1 + r.omega2 + b.adv_f + b.omega1 + b.omega2
  // so that only ONE colour's populations are live at a time (register pressure).
  double om2[Q], tot[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double gE = gx * (double)icx(q) + gy * (double)icy(q);
    const double t1 = gE / (1e-20 + gnorm);
    const double xi = 0.5 * gnorm * (wq(q) * (t1 * t1) - cg_B(q));  // eval_xi :290-300
    om2[q] = A * xi;                                                  // :263-273
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double f[Q], m[Q];
    if (FROM_POST) {
      gather_bc(f, k ? in_b : in_r, g, bc, r, c);
    } else {
      const double* in = k ? in_b : in_r;
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = in[q * g.plane + lo];
    }
    {
      double d[Q];
      cg_feq(d, k ? rb : rr, cc.k[k], ux, uy);  // :431-432
#pragma unroll
      for (int q = 0; q < Q; ++q) d[q] = d[q] - f[q];
#pragma unroll
      for (int a = 0; a < Q; ++a) {              // eval_mrt_operator :249-261
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < Q; ++q) s += cg_M(a, q) * d[q];
        double Ck = 0.0;                         // update_C :320-336
        if (a == 1) Ck = 3.0 * (1.0 - 0.5 * 1.25) * (dxq[k] + dyq[k]);
        if (a == 7) Ck = (1.0 - 0.5 * s_nu) * (dxq[k] - dyq[k]);
        m[a] = S[a] * s + Ck;
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double om1 = 0.0;
#pragma unroll
      for (int a = 0; a < Q; ++a) om1 += ((1.0 / 36.0) * cg_Mi36(q, a)) * m[a];
      if (k == 0) tot[q] = f[q] + om1 + om2[q];
      else tot[q] = tot[q] + f[q] + om1 + om2[q];
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double gU = gx * cc.unitx[q] + gy * cc.unity[q];
    const double kappa = (rr * rb * gU * (rr * cc.k[0].phi[q] + rb * cc.k[1].phi[q])) /
                         ((rt * rt) * (1e-20 + gnorm));               // eval_kappa :302-318
    const double cu = ux * (double)icx(q) + uy * (double)icy(q);
    const double FgE = cc.gr * (double)icx(q) + cc.gc * (double)icy(q);
    const double uFg = ux * cc.gr + uy * cc.gc;
    const double Fq = (1 - 0.5 * s_nu) * ((3.0 + 9.0 * cu) * FgE - 3.0 * uFg) * wq(q);  // :460-462
    const double o3r = rr * tot[q] / rt + cc.k[0].beta * kappa;  // eval_rec_operator :275-288
    const double o3b = rb * tot[q] / rt + cc.k[1].beta * kappa;
    pn_r[q * g.plane + lo] = cc.add_source ? o3r + Fq : o3r;  // :463 (source commented out in
    pn_b[q * g.plane + lo] = cc.add_source ? o3b + Fq : o3b;  // :464  mrtcg_static_droplet.cpp:513)
  }
  if (WITH_FIELDS) {
    const long oo = (long)r * g.C + c;  // diagnostics carry no ghost rows
    psi_out[oo] = psi;
    snu_out[oo] = s_nu;
  }
}

}  // namespace lbm
