// TEST INFRASTRUCTURE ONLY -- see lbm_oracle.h.  Plain C++17 + OpenMP, no torch, no HIP.
// CPU restatement of the reference hot path in the reference's own AoS layout
// f[R][C][9].  Operation order follows the reference expressions where rounding
// could depend on it; libtorch's internal reduction / dgemm orders are unknown, so
// agreement with oracle/_ref is to a few ulp per step, not bitwise (tests state it).
#include "lbm_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// src/solver.cpp:12-21 -- NB: the reference calls the weights "E" and the velocities "c".
const double W9[9] = {4.0 / 9.0, 1.0 / 9.0, 1.0 / 9.0, 1.0 / 9.0, 1.0 / 9.0,
                      1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0};
const double CX[9] = {0.0, 1.0, 0.0, -1.0, 0.0, 1.0, -1.0, -1.0, 1.0};
const double CY[9] = {0.0, 0.0, 1.0, 0.0, -1.0, 1.0, 1.0, -1.0, -1.0};
const int ICX[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
const int ICY[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};

inline size_t nid(int r, int c, int C) { return (size_t)r * C + c; }

// ---- per-node pieces of solver:: ------------------------------------------------
inline double node_rho(const double* f) {  // solver.cpp:23-26
  double s = 0.0;
  for (int q = 0; q < 9; ++q) s += f[q];
  return s;
}
inline void node_mom(const double* f, double& jx, double& jy) {  // matmul(f, c^T), solver.cpp:30,36
  jx = 0.0;
  jy = 0.0;
  for (int q = 0; q < 9; ++q) {
    jx += f[q] * CX[q];
    jy += f[q] * CY[q];
  }
}
inline void node_feq(double* feq, double rho, double ux, double uy) {  // solver.cpp:51-62
  const double u_u = ux * ux + uy * uy;
  for (int q = 0; q < 9; ++q) {
    const double c_u = ux * CX[q] + uy * CY[q];
    const double A = 1.0 + 3.0 * c_u + 4.5 * (c_u * c_u) - 1.5 * u_u;
    feq[q] = (rho * A) * W9[q];
  }
}
inline void node_feq_incomp(double* feq, double rho, double ux, double uy) {  // solver.cpp:39-49
  for (int q = 0; q < 9; ++q) {
    const double c_u = ux * CX[q] + uy * CY[q];
    feq[q] = (rho + 3.0 * c_u) * W9[q];
  }
}

void advect(double* g, const double* f, int R, int C) {  // solver.cpp:76-131 (pull form)
#pragma omp parallel for schedule(static)
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      double* gn = g + nid(r, c, C) * 9;
      for (int q = 0; q < 9; ++q) {
        int rs = r - ICX[q], cs = c - ICY[q];
        if (rs < 0) rs += R;
        if (rs >= R) rs -= R;
        if (cs < 0) cs += C;
        if (cs >= C) cs -= C;
        gn[q] = f[nid(rs, cs, C) * 9 + q];
      }
    }
}

// Tiny grids (the reference's 21x21 cases): OpenMP fork/join would dominate.
struct small_grid_guard {
  int old;
  explicit small_grid_guard(size_t nodes) {
#ifdef _OPENMP
    old = omp_get_max_threads();
    // one thread per 8192 nodes at most: on a 128-thread host the fork/join of a full team costs
    // more than a whole pass over a 128 x 128 block
    const int want = (int)(nodes / 8192);
    if (want < old) omp_set_num_threads(want < 1 ? 1 : want);
#else
    (void)nodes; old = 1;
#endif
  }
  ~small_grid_guard() {
#ifdef _OPENMP
    omp_set_num_threads(old);
#endif
  }
};

}  // namespace

extern "C" {

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

void orc_calc_rho(double* rho, const double* f, int R, int C) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C; ++i) rho[i] = node_rho(f + i * 9);
}
void orc_calc_u(double* u, const double* f, const double* rho, int R, int C) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C; ++i) {
    double jx, jy;
    node_mom(f + i * 9, jx, jy);
    u[2 * i] = jx / rho[i];
    u[2 * i + 1] = jy / rho[i];
  }
}
void orc_calc_incomp_u(double* u, const double* f, int R, int C) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C; ++i) node_mom(f + i * 9, u[2 * i], u[2 * i + 1]);
}
void orc_equilibrium(double* feq, const double* u, const double* rho, int R, int C) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C; ++i) node_feq(feq + i * 9, rho[i], u[2 * i], u[2 * i + 1]);
}
void orc_incomp_equilibrium(double* feq, const double* u, const double* rho, int R, int C) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C; ++i)
    node_feq_incomp(feq + i * 9, rho[i], u[2 * i], u[2 * i + 1]);
}
void orc_collision(double* fc, const double* f, const double* feq, double omega, int R, int C) {
  // solver.cpp:65-74
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C * 9; ++i) fc[i] = (1.0 - omega) * f[i] + omega * feq[i];
}
void orc_advect(double* g, const double* f, int R, int C) { advect(g, f, R, C); }

void orc_bgk_periodic_steps(double* f, double* rho_out, double* u_out, int R, int C,
                            double omega, int incompressible, int nsteps) {
  const size_t N = (size_t)R * C;
  small_grid_guard sg(N);
  std::vector<double> fc(N * 9), rho(N, 1.0), u(N * 2, 0.0);
  for (int t = 0; t < nsteps; ++t) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
      const double* fn = f + i * 9;
      double feq[9], jx, jy;
      const double rh = node_rho(fn);
      node_mom(fn, jx, jy);
      double ux = jx, uy = jy;
      if (incompressible) {
        node_feq_incomp(feq, rh, ux, uy);
      } else {
        ux = jx / rh;
        uy = jy / rh;
        node_feq(feq, rh, ux, uy);
      }
      rho[i] = rh;
      u[2 * i] = ux;
      u[2 * i + 1] = uy;
      for (int q = 0; q < 9; ++q) fc[i * 9 + q] = (1.0 - omega) * fn[q] + omega * feq[q];
    }
    advect(f, fc.data(), R, C);
  }
  if (rho_out) std::memcpy(rho_out, rho.data(), N * sizeof(double));
  if (u_out) std::memcpy(u_out, u.data(), N * 2 * sizeof(double));
}

// ---------------------------------------------------------------------------------
// test/horizontal_poiseuille_test.cpp
// ---------------------------------------------------------------------------------
static void hpt_pressure_rows(double* f_coll, const double* f_equi, const double* u, int H, int W,
                              double rho_inlet, double rho_outlet) {
  // :25-45.  virtual inlet row 0 <- outlet row H-2; virtual outlet row H-1 <- inlet row 1.
  for (int c = 0; c < W; ++c) {
    double te[9];
    size_t src = nid(H - 2, c, W);
    node_feq_incomp(te, rho_inlet * 1.0, u[2 * src], u[2 * src + 1]);
    for (int q = 0; q < 9; ++q)
      f_coll[nid(0, c, W) * 9 + q] = (te[q] + f_coll[src * 9 + q]) - f_equi[src * 9 + q];
  }
  for (int c = 0; c < W; ++c) {
    double te[9];
    size_t src = nid(1, c, W);
    node_feq_incomp(te, rho_outlet * 1.0, u[2 * src], u[2 * src + 1]);
    for (int q = 0; q < 9; ++q)
      f_coll[nid(H - 1, c, W) * 9 + q] = (te[q] + f_coll[src * 9 + q]) - f_equi[src * 9 + q];
  }
}

static void bounce_back_columns(double* f_adve, const double* f_coll, int H, int W) {
  // :146-152
  for (int r = 0; r < H; ++r) {
    const size_t e = nid(r, W - 1, W) * 9, w = nid(r, 0, W) * 9;
    f_adve[e + 4] = f_coll[e + 2];
    f_adve[e + 7] = f_coll[e + 5];
    f_adve[e + 8] = f_coll[e + 6];
    f_adve[w + 2] = f_coll[w + 4];
    f_adve[w + 5] = f_coll[w + 7];
    f_adve[w + 6] = f_coll[w + 8];
  }
}

int orc_hpt_run(const orc_hpt_params* p, double* f_adve, double* u, double* rho, double* l2_out) {
  const int H = p->H, W = p->W;
  const size_t N = (size_t)H * W;
  small_grid_guard sg(N);
  std::vector<double> f_equi(N * 9, 0.0), f_coll(N * 9, 0.0);
  for (size_t i = 0; i < N; ++i) {  // :80-82, :91
    rho[i] = 1.0;
    u[2 * i] = u[2 * i + 1] = 0.0;
  }
  orc_incomp_equilibrium(f_adve, u, rho, H, W);
  double old_mean = 1.0;  // old_u = ones_like(rho), :97
  int t = 0;
  for (; t < p->T; ++t) {
    if (p->check_convergence && t % 100 == 1) {  // :113-126
      double m = 0.0;
      for (size_t i = 0; i < N; ++i) m += u[2 * i];
      m /= (double)N;
      const double diff = std::fabs(m / old_mean - 1.0);
      if (diff < 1e-12) break;
      old_mean = m;
    }
    orc_calc_rho(rho, f_adve, H, W);                              // :130
    orc_calc_incomp_u(u, f_adve, H, W);                           // :131
    orc_incomp_equilibrium(f_equi.data(), u, rho, H, W);          // :134
    orc_collision(f_coll.data(), f_adve, f_equi.data(), p->omega, H, W);  // :137
    hpt_pressure_rows(f_coll.data(), f_equi.data(), u, H, W, p->rho_inlet, p->rho_outlet);  // :140
    advect(f_adve, f_coll.data(), H, W);                          // :143
    bounce_back_columns(f_adve, f_coll.data(), H, W);             // :146-152
  }
  if (l2_out) {  // :163-170
    std::vector<double> ua(W);
    double den = 0.0;
    for (int c = 0; c < W; ++c) {
      const double y = (1.0 + c) - 0.5;
      ua[c] = -4.0 * p->u_max / ((double)W * W) * y * (y - W);
      den += ua[c] * ua[c];
    }
    den = 1.0 / std::sqrt(den);
    double sum = 0.0;
    for (int r = 1; r < H - 1; ++r) {
      double e = 0.0;
      for (int c = 0; c < W; ++c) {
        const double d = u[2 * nid(r, c, W)] - ua[c];
        e += d * d;
      }
      sum += std::sqrt(e) * den;
    }
    *l2_out = (1.0 / H) * sum;
  }
  return t;
}

// ---------------------------------------------------------------------------------
// specular_boundary_test, free_stream_test, gravity_test
// ---------------------------------------------------------------------------------
static void specular_columns(double* f_adve, const double* f_coll, int H, int W) {
  // specular_boundary_test.cpp:121-127 == cylinder_test.cpp:157-163
  for (int r = 0; r < H; ++r) {
    const size_t e = nid(r, W - 1, W) * 9, w = nid(r, 0, W) * 9;
    f_adve[e + 4] = f_coll[e + 2];
    f_adve[e + 7] = f_coll[e + 6];
    f_adve[e + 8] = f_coll[e + 5];
    f_adve[w + 2] = f_coll[w + 4];
    f_adve[w + 5] = f_coll[w + 8];
    f_adve[w + 6] = f_coll[w + 7];
  }
}

void orc_sbt_run(int H, int W, int T, double omega, double rho_inlet, double rho_outlet,
                 double* f_adve, double* u, double* rho) {
  const size_t N = (size_t)H * W;
  small_grid_guard sg(N);
  std::vector<double> f_equi(N * 9), f_coll(N * 9);
  for (size_t i = 0; i < N; ++i) {
    rho[i] = 1.0;
    u[2 * i] = u[2 * i + 1] = 0.0;
  }
  orc_incomp_equilibrium(f_adve, u, rho, H, W);  // :87 (incompressible init, compressible loop)
  for (int t = 0; t < T; ++t) {
    orc_calc_rho(rho, f_adve, H, W);             // :104
    orc_calc_u(u, f_adve, rho, H, W);            // :105
    orc_equilibrium(f_equi.data(), u, rho, H, W);                       // :108
    orc_collision(f_coll.data(), f_adve, f_equi.data(), omega, H, W);   // :111
    for (int e = 0; e < 2; ++e) {                // periodic_boundary_condition :23-45 (compressible feq)
      const int dst = e ? H - 1 : 0, src = e ? 1 : H - 2;
      for (int c = 0; c < W; ++c) {
        double te[9];
        const size_t s = nid(src, c, W);
        node_feq(te, (e ? rho_outlet : rho_inlet) * 1.0, u[2 * s], u[2 * s + 1]);
        for (int q = 0; q < 9; ++q)
          f_coll[nid(dst, c, W) * 9 + q] = (te[q] + f_coll[s * 9 + q]) - f_equi[s * 9 + q];
      }
    }
    advect(f_adve, f_coll.data(), H, W);                 // :117
    specular_columns(f_adve, f_coll.data(), H, W);       // :120-127
  }
}

void orc_free_stream_steps(double* f_adve, double* u, double* rho, int X, int Y, double omega,
                           double uw, int nsteps) {
  const size_t N = (size_t)X * Y;
  small_grid_guard sg(N);
  std::vector<double> f_equi(N * 9), f_coll(N * 9);
  static const int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
  double abb[9];  // free_stream_test.cpp:104, :115 with u_w = (uw, 0)
  for (int q = 0; q < 9; ++q) {
    const double cu = uw * CX[q] + 0.0 * CY[q];
    abb[q] = (2.0 + 9.0 * std::pow(cu, 2.0) - 3.0 * (uw * uw + 0.0 * 0.0)) * W9[q];
  }
  for (int t = 0; t < nsteps; ++t) {
    orc_calc_rho(rho, f_adve, X, Y);                                    // :90
    orc_calc_incomp_u(u, f_adve, X, Y);                                 // :91
    orc_incomp_equilibrium(f_equi.data(), u, rho, X, Y);                // :94
    orc_collision(f_coll.data(), f_adve, f_equi.data(), omega, X, Y);   // :97
    advect(f_adve, f_coll.data(), X, Y);                                // :100
    for (int e = 0; e < 2; ++e) {                                       // :102-124
      const int r = e ? X - 1 : 0;
      for (int c = 0; c < Y; ++c) {
        const size_t n = nid(r, c, Y) * 9;
        for (int q = 1; q < 9; ++q) f_adve[n + OPP[q]] = -f_coll[n + q] + abb[q];
      }
    }
    specular_columns(f_adve, f_coll.data(), X, Y);                      // :126-133
  }
}

int orc_gravity_run(int H, int W, int T, double omega, double Fr, double Fc, double rho_inlet,
                    double rho_outlet, int check_convergence, double* f_adve, double* u, double* rho) {
  const size_t N = (size_t)H * W;
  small_grid_guard sg(N);
  std::vector<double> f_equi(N * 9), f_coll(N * 9);
  const double ics2 = 1.0 / 3.0, ics4 = 1.0 / 9.0;  // gravity_test.cpp:81-82
  for (size_t i = 0; i < N; ++i) {
    rho[i] = 1.0;
    u[2 * i] = u[2 * i + 1] = 0.0;
  }
  orc_incomp_equilibrium(f_adve, u, rho, H, W);  // :103
  double old_mean = 1.0;
  int t = 0;
  for (; t < T; ++t) {
    if (check_convergence && t % 100 == 1) {  // :126-139
      double m = 0.0;
      for (size_t i = 0; i < N; ++i) m += u[2 * i];
      m /= (double)N;
      if (std::fabs(m / old_mean - 1.0) < 1e-12) break;
      old_mean = m;
    }
    orc_calc_rho(rho, f_adve, H, W);         // :143
    orc_calc_incomp_u(u, f_adve, H, W);      // :144
    for (size_t i = 0; i < N; ++i) {         // u += Fg.t()  :146
      u[2 * i] += Fr;
      u[2 * i + 1] += Fc;
    }
    orc_incomp_equilibrium(f_equi.data(), u, rho, H, W);  // :150
    for (size_t i = 0; i < N; ++i) {
      const double ux = u[2 * i], uy = u[2 * i + 1];
      const double uF = ux * Fr + uy * Fc;
      for (int q = 0; q < 9; ++q) {
        const double eqp = -omega * (f_adve[i * 9 + q] - f_equi[i * 9 + q]);  // :151
        const double cu = ux * CX[q] + uy * CY[q], cF = Fr * CX[q] + Fc * CY[q];
        const double S = ((1 - 0.5 * omega) * ((ics2 + ics4 * cu) * cF - ics2 * uF) * W9[q]);  // :154
        f_coll[i * 9 + q] = f_adve[i * 9 + q] + eqp + S;  // :158-160
      }
    }
    hpt_pressure_rows(f_coll.data(), f_equi.data(), u, H, W, rho_inlet, rho_outlet);  // :164 (incompressible)
    advect(f_adve, f_coll.data(), H, W);                   // :168
    bounce_back_columns(f_adve, f_coll.data(), H, W);      // :171-177
  }
  return t;
}

// ---------------------------------------------------------------------------------
// test/decompose_domain.cpp
// ---------------------------------------------------------------------------------
void orc_ddm_run(int H, int W, int T, double omega, double rho_inlet, double rho_outlet,
                 double* fA, double* fB, double* uA, double* uB, double* rhoA, double* rhoB) {
  const size_t N = (size_t)H * W;
  small_grid_guard sg(N);
  std::vector<double> eA(N * 9), eB(N * 9), cA(N * 9), cB(N * 9);
  for (size_t i = 0; i < N; ++i) {  // :105-108
    rhoA[i] = rhoB[i] = 1.0;
    uA[2 * i] = uA[2 * i + 1] = uB[2 * i] = uB[2 * i + 1] = 0.0;
  }
  orc_equilibrium(fA, uA, rhoA, H, W);  // :122-123
  orc_equilibrium(fB, uB, rhoB, H, W);
  for (int t = 0; t < T; ++t) {
    orc_calc_rho(rhoA, fA, H, W);  // :141-152
    orc_calc_rho(rhoB, fB, H, W);
    orc_calc_u(uA, fA, rhoA, H, W);
    orc_calc_u(uB, fB, rhoB, H, W);
    orc_equilibrium(eA.data(), uA, rhoA, H, W);
    orc_equilibrium(eB.data(), uB, rhoB, H, W);
    orc_collision(cA.data(), fA, eA.data(), omega, H, W);
    orc_collision(cB.data(), fB, eB.data(), omega, H, W);
    // periodic_boundary_condition(A, B, ...) :50-73 (compressible equilibrium here)
    for (int c = 0; c < W; ++c) {
      double te[9];
      const size_t s = nid(H - 2, c, W);
      node_feq(te, rho_inlet * 1.0, uB[2 * s], uB[2 * s + 1]);
      for (int q = 0; q < 9; ++q)
        cA[nid(0, c, W) * 9 + q] = (te[q] + cB[s * 9 + q]) - eB[s * 9 + q];
    }
    for (int c = 0; c < W; ++c) {
      double te[9];
      const size_t s = nid(1, c, W);
      node_feq(te, rho_outlet * 1.0, uA[2 * s], uA[2 * s + 1]);
      for (int q = 0; q < 9; ++q)
        cB[nid(H - 1, c, W) * 9 + q] = (te[q] + cA[s * 9 + q]) - eA[s * 9 + q];
    }
    advect(fA, cA.data(), H, W);  // :159-160
    advect(fB, cB.data(), H, W);
    bounce_back_columns(fA, cA.data(), H, W);  // :163-178
    bounce_back_columns(fB, cB.data(), H, W);
    // bind :181-187
    for (int c = 0; c < W; ++c) fA[nid(H - 1, c, W) * 9 + 3] = cB[nid(0, c, W) * 9 + 3];
    for (int c = 1; c < W; ++c) fA[nid(H - 1, c, W) * 9 + 6] = cB[nid(0, c - 1, W) * 9 + 6];
    for (int c = 0; c < W - 1; ++c) fA[nid(H - 1, c, W) * 9 + 7] = cB[nid(0, c + 1, W) * 9 + 7];
    for (int c = 0; c < W; ++c) fB[nid(0, c, W) * 9 + 1] = cA[nid(H - 1, c, W) * 9 + 1];
    for (int c = 1; c < W; ++c) fB[nid(0, c, W) * 9 + 5] = cA[nid(H - 1, c - 1, W) * 9 + 5];
    for (int c = 0; c < W - 1; ++c) fB[nid(0, c, W) * 9 + 8] = cA[nid(H - 1, c + 1, W) * 9 + 8];
  }
}

// ---------------------------------------------------------------------------------
// test/decompose_domain_loop.cpp: four blocks A (L x L/4), B (L/4 x L/2), C (L x L/4),
// D (L/4 x L/2) closed into a square loop channel by column-seam bindings; momentum source in A
// ---------------------------------------------------------------------------------
namespace {
struct Blk {
  int R, C;
  double *f, *coll;  // adve_f, coll_f  [R][C][9]
};
// dst.f[r0 .. r1, col, q] = src.coll[s0 .., scol, sq]; negative indices count from the end
// (the torch slices of :192-261 written out; `none` = an open slice end)
const int none = 1 << 30;
inline int norm(int i, int n) { return i == none ? n : (i < 0 ? i + n : i); }
void bind_rows(const Blk& d, int r0, int r1, int col, int q, const Blk& s, int s0, int scol, int sq) {
  r0 = norm(r0, d.R);
  r1 = norm(r1, d.R);
  s0 = norm(s0, s.R);
  col = norm(col, d.C);
  scol = norm(scol, s.C);
  for (int k = 0; k < r1 - r0; ++k)
    d.f[nid(r0 + k, col, d.C) * 9 + q] = s.coll[nid(s0 + k, scol, s.C) * 9 + sq];
}
void wall_rows(const Blk& b) {  // top :192-195 / bottom :196-199 of every block
  for (int c = 0; c < b.C; ++c) {
    const size_t t = nid(0, c, b.C) * 9, m = nid(b.R - 1, c, b.C) * 9;
    b.f[t + 8] = b.coll[t + 6];
    b.f[t + 1] = b.coll[t + 3];
    b.f[t + 5] = b.coll[t + 7];
    b.f[m + 7] = b.coll[m + 5];
    b.f[m + 3] = b.coll[m + 1];
    b.f[m + 6] = b.coll[m + 8];
  }
}
void wall_col(const Blk& b, int r0, int r1, bool left) {  // :200-207, :224-231
  r0 = norm(r0, b.R);
  r1 = norm(r1, b.R);
  const int col = left ? 0 : b.C - 1;
  for (int r = r0; r < r1; ++r) {
    const size_t i = nid(r, col, b.C) * 9;
    if (left) {
      b.f[i + 2] = b.coll[i + 4];
      b.f[i + 5] = b.coll[i + 7];
      b.f[i + 6] = b.coll[i + 8];
    } else {
      b.f[i + 4] = b.coll[i + 2];
      b.f[i + 7] = b.coll[i + 5];
      b.f[i + 8] = b.coll[i + 6];
    }
  }
}
}  // namespace

// nsteps iterations of :112-262 from the driver's start state (m_0 = 1, m_1 = 0, adve_f = feq).
// Outputs: adve_f of the four blocks and rho / u as computed in the LAST iteration (what the
// snapshot taken at t = nsteps stores, before the driver adds F to A's u on the force rows, :114).
void orc_ddl_run(int L, int nsteps, double omega, double Fr, double* fA, double* fB, double* fC,
                 double* fD, double* rho[4], double* u[4]) {
  const int L2 = L / 2, L4 = L / 4;
  small_grid_guard sg((size_t)L * L4);
  const int Rs[4] = {L, L4, L, L4}, Cs[4] = {L4, L2, L4, L2};
  double* fs[4] = {fA, fB, fC, fD};
  std::vector<double> coll[4], feq[4];
  Blk b[4];
  for (int k = 0; k < 4; ++k) {
    const size_t N = (size_t)Rs[k] * Cs[k];
    coll[k].resize(N * 9);
    feq[k].resize(N * 9);
    b[k] = Blk{Rs[k], Cs[k], fs[k], coll[k].data()};
    for (size_t i = 0; i < N; ++i) {  // :77-80
      rho[k][i] = 1.0;
      u[k][2 * i] = u[k][2 * i + 1] = 0.0;
    }
    orc_equilibrium(fs[k], u[k], rho[k], Rs[k], Cs[k]);  // :105-108
  }
  const Blk &A = b[0], &B = b[1], &C = b[2], &D = b[3];
  const int f0 = L4 + 5, f1 = L4 + 55;  // force_idx :63
  for (int t = 0; t < nsteps; ++t) {
    for (int k = 0; k < 4; ++k) {  // :134-150
      orc_calc_rho(rho[k], fs[k], Rs[k], Cs[k]);
      orc_calc_u(u[k], fs[k], rho[k], Rs[k], Cs[k]);
      orc_equilibrium(feq[k].data(), u[k], rho[k], Rs[k], Cs[k]);
    }
    // A: adve + (-omega (adve - equi)), then + S on the force rows (:152-160); S with (3, 9)
    {
      const size_t N = (size_t)A.R * A.C;
#pragma omp parallel for schedule(static)
      for (long i = 0; i < (long)(N * 9); ++i) A.coll[i] = A.f[i] + (-omega * (A.f[i] - feq[0][i]));
      for (int r = f0; r < f1; ++r)
        for (int c = 0; c < A.C; ++c) {
          const size_t i = nid(r, c, A.C);
          const double ux = u[0][2 * i], uy = u[0][2 * i + 1];
          const double uF = ux * Fr + uy * 0.0;
          for (int q = 0; q < 9; ++q) {
            const double uc = ux * CX[q] + uy * CY[q];
            const double Fc = Fr * CX[q] + 0.0 * CY[q];
            const double S = ((1 - 0.5 * omega) * ((3.0 + 9.0 * uc) * Fc - 3.0 * uF)) * W9[q];
            A.coll[i * 9 + q] = A.coll[i * 9 + q] + S;
          }
        }
    }
    for (int k = 1; k < 4; ++k) orc_collision(coll[k].data(), fs[k], feq[k].data(), omega, Rs[k], Cs[k]);  // :161-163
    for (int k = 0; k < 4; ++k) advect(fs[k], coll[k].data(), Rs[k], Cs[k]);                               // :166-169
    // no-slip walls :173-231 (in the driver's order: later assignments win at shared corners)
    wall_rows(A);
    wall_col(A, L4, -L4, true);
    wall_col(A, 1, -1, false);
    wall_rows(B);
    wall_rows(C);
    wall_col(C, 1, -1, true);
    wall_col(C, L4, -L4, false);
    wall_rows(D);
    // bindings :235-261
    bind_rows(A, -L4, -1, 0, 6, B, 1, -1, 6);
    bind_rows(A, -L4, none, 0, 2, B, 0, -1, 2);
    bind_rows(A, -L4 + 1, none, 0, 5, B, 0, -1, 5);
    bind_rows(B, 1, none, -1, 8, A, -L4, 0, 8);
    bind_rows(B, 0, none, -1, 4, A, -L4, 0, 4);
    bind_rows(B, 0, -1, -1, 7, A, -L4 + 1, 0, 7);
    bind_rows(B, 0, -1, 0, 6, C, -L4 + 1, -1, 6);
    bind_rows(B, 0, none, 0, 2, C, -L4, -1, 2);
    bind_rows(B, 1, none, 0, 5, C, -L4, -1, 5);
    bind_rows(C, -L4, -1, -1, 7, B, 1, 0, 7);
    bind_rows(C, -L4, none, -1, 4, B, 0, 0, 4);
    bind_rows(C, -L4 + 1, none, -1, 8, B, 0, 0, 8);
    bind_rows(C, 0, L4 - 1, -1, 7, D, 1, 0, 7);
    bind_rows(C, 0, L4, -1, 4, D, 0, 0, 4);
    bind_rows(C, 1, L4, -1, 8, D, 0, 0, 8);
    bind_rows(D, 0, -1, 0, 6, C, 1, -1, 6);
    bind_rows(D, 0, none, 0, 2, C, 0, -1, 2);
    bind_rows(D, 1, none, 0, 5, C, 0, -1, 5);
    bind_rows(D, 0, -1, -1, 7, A, 1, 0, 7);
    bind_rows(D, 0, none, -1, 4, A, 0, 0, 4);
    bind_rows(D, 1, none, -1, 8, A, 0, 0, 8);
    bind_rows(A, 0, L4 - 1, 0, 6, D, 1, -1, 6);
    bind_rows(A, 0, L4, 0, 2, D, 0, -1, 2);
    bind_rows(A, 1, L4, 0, 5, D, 0, -1, 5);
  }
}

// ---------------------------------------------------------------------------------
// test/rectangle_sedimentation_test.cpp: fluid f + sediment concentration g (advection-diffusion
// with settling velocity w_s), anti-bounce-back inlet / outlet columns, specular top, no-slip
// bottom, a three-sided rectangular obstacle at hard-coded coordinates (:71-73)
// ---------------------------------------------------------------------------------
namespace {
// the eight (dst, src) population pairs of :139-146 etc., in the driver's order
const int ABB_DST[8] = {3, 4, 1, 2, 7, 8, 5, 6}, ABB_SRC[8] = {1, 2, 3, 4, 5, 6, 7, 8};
inline void abb_terms(double* t, double uw0, double uw1) {  // :135 / :149
  const double uu = uw0 * uw0 + uw1 * uw1;
  for (int q = 0; q < 9; ++q) {
    const double uc = uw0 * CX[q] + uw1 * CY[q];
    t[q] = ((2.0 + 9.0 * (uc * uc)) - 3.0 * uu) * W9[q];
  }
}
}  // namespace

// nsteps iterations of :106-237; init != 0: build the driver's start state first (:79-103).
// State: f, g = f_adve, g_adve [X][Y][9]; rho [X][Y], u [X][Y][2], C [X][Y].
void orc_sed_steps(int X, int Y, double omega, double u_in, double w_s, double Cw, int init,
                   int nsteps, double* f, double* g, double* rho, double* u, double* C) {
  const size_t N = (size_t)X * Y;
  small_grid_guard sg(N);
  const int R23 = X - 151, C28 = 200, C38 = 250;  // :71-73 (R23 = -151 counted from the end)
  std::vector<double> C_w(X, 0.0);
  for (int r = X - 50; r < X; ++r) C_w[r] = Cw;  // :93
  if (init) {
    for (size_t i = 0; i < N; ++i) {
      rho[i] = 1.0;
      u[2 * i] = 0.0;
      u[2 * i + 1] = u_in;  // :83
      C[i] = 0.0;
    }
    for (int r = 0; r < X; ++r) C[nid(r, 0, Y)] = C_w[r];  // :94
    orc_equilibrium(g, u, C, X, Y);                         // :95
    orc_incomp_equilibrium(f, u, rho, X, Y);                // :100
    orc_calc_rho(rho, f, X, Y);                             // :103-104
    orc_calc_u(u, f, rho, X, Y);
  }
  std::vector<double> fe(N * 9), fc(N * 9), ge(N * 9), gc(N * 9), us(N * 2);
  for (int t = 0; t < nsteps; ++t) {
    orc_equilibrium(fe.data(), u, rho, X, Y);  // :123
    for (size_t i = 0; i < 2 * N; ++i) us[i] = u[i] + w_s;
    orc_equilibrium(ge.data(), us.data(), C, X, Y);        // :124
    orc_collision(fc.data(), f, fe.data(), omega, X, Y);   // :130
    orc_collision(gc.data(), g, ge.data(), omega / 1.0, X, Y);
    // zero gradient (:136-140): top row, then the outlet column
    for (int c = 0; c < Y; ++c)
      for (int q = 0; q < 9; ++q) gc[nid(0, c, Y) * 9 + q] = gc[nid(1, c, Y) * 9 + q];
    for (int r = 1; r < X - 1; ++r)
      for (int q = 0; q < 9; ++q) gc[nid(r, Y - 1, Y) * 9 + q] = gc[nid(r, Y - 2, Y) * 9 + q];
    advect(f, fc.data(), X, Y);  // :143-144
    advect(g, gc.data(), X, Y);
    {  // inlet, fixed wall velocity (0, u_in) (:148-159)
      double a[9];
      abb_terms(a, 0.0, u_in);
      for (int r = 1; r < X - 1; ++r) {
        const size_t i = nid(r, 0, Y) * 9;
        for (int k = 0; k < 8; ++k) f[i + ABB_DST[k]] = -fc[i + ABB_SRC[k]] + a[ABB_SRC[k]];
      }
    }
    for (int r = 0; r < X; ++r) {  // outlet, extrapolated wall velocity (:161-170)
      const size_t l = nid(r, Y - 1, Y), m = nid(r, Y - 2, Y);
      double a[9];
      abb_terms(a, 1.5 * u[2 * l] - 0.5 * u[2 * m], 1.5 * u[2 * l + 1] - 0.5 * u[2 * m + 1]);
      for (int k = 0; k < 8; ++k) f[l * 9 + ABB_DST[k]] = -fc[l * 9 + ABB_SRC[k]] + a[ABB_SRC[k]];
    }
    for (int c = 0; c < Y; ++c) {  // specular top (:173-175), no-slip bottom (:178-180)
      const size_t a = nid(0, c, Y) * 9, b = nid(X - 1, c, Y) * 9;
      f[a + 8] = fc[a + 7];
      f[a + 1] = fc[a + 3];
      f[a + 5] = fc[a + 6];
      f[b + 7] = fc[b + 5];
      f[b + 3] = fc[b + 1];
      f[b + 6] = fc[b + 8];
    }
    for (int r = R23 + 1; r < X - 1; ++r) {  // first wall (:184-186)
      const size_t i = nid(r, C28, Y) * 9;
      f[i + 8] = fc[i + 6];
      f[i + 4] = fc[i + 2];
      f[i + 7] = fc[i + 5];
    }
    for (int c = C28; c < C38 + 1; ++c) {  // ceiling (:188-190)
      const size_t i = nid(R23, c, Y) * 9;
      f[i + 6] = fc[i + 8];
      f[i + 3] = fc[i + 1];
      f[i + 7] = fc[i + 5];
    }
    for (int r = R23 + 1; r < X - 1; ++r) {  // second wall (:192-194)
      const size_t i = nid(r, C38, Y) * 9;
      f[i + 5] = fc[i + 7];
      f[i + 2] = fc[i + 4];
      f[i + 6] = fc[i + 8];
    }
    orc_calc_rho(rho, f, X, Y);  // :197-198
    orc_calc_u(u, f, rho, X, Y);
    for (int r = 1; r < X - 1; ++r) {  // concentration inlet (:202-217)
      const size_t i = nid(r, 0, Y);
      const double w0 = u[2 * i] + w_s, w1 = u[2 * i + 1] + w_s, uu = w0 * w0 + w1 * w1;
      for (int k = 0; k < 8; ++k) {
        const int q = ABB_SRC[k];
        const double uc = w0 * CX[q] + w1 * CY[q];
        const double ga = ((((1.0 + 3.0 * uc) + 4.5 * (uc * uc)) - 1.5 * uu) * W9[q]) * C_w[r];
        g[i * 9 + ABB_DST[k]] = -gc[i * 9 + q] + 2.0 * ga;
      }
    }
    for (int r = R23 + 1; r < X; ++r) {  // first wall (to the last row, :220-222)
      const size_t i = nid(r, C28, Y) * 9;
      g[i + 8] = -gc[i + 6];
      g[i + 4] = -gc[i + 2];
      g[i + 7] = -gc[i + 5];
    }
    for (int c = C28; c < C38 + 1; ++c) {  // ceiling (:224-226)
      const size_t i = nid(R23, c, Y) * 9;
      g[i + 6] = -gc[i + 8];
      g[i + 3] = -gc[i + 1];
      g[i + 7] = -gc[i + 5];
    }
    for (int r = R23 + 1; r < X - 1; ++r) {  // second wall (:228-230)
      const size_t i = nid(r, C38, Y) * 9;
      g[i + 5] = -gc[i + 7];
      g[i + 2] = -gc[i + 4];
      g[i + 6] = -gc[i + 8];
    }
    for (int c = 0; c < Y; ++c) {  // bottom (:232-234)
      const size_t i = nid(X - 1, c, Y) * 9;
      g[i + 6] = gc[i + 8];
      g[i + 3] = gc[i + 1];
      g[i + 7] = gc[i + 5];
    }
    orc_calc_rho(C, g, X, Y);  // :235
  }
}

// ---------------------------------------------------------------------------------
// ulbm::d2q9::kbc
// ---------------------------------------------------------------------------------
namespace {
const double cs2 = 1.0 / 3.0, cs4 = 1.0 / 9.0;  // ulbm.hpp:26-27

inline void kbc_feq_poly(double* e, double ux, double uy, double ux2, double uy2) {
  // ulbm.cpp:234-242 == :252-260
  e[0] = 2.0 * cs2 * (0.5 * ux2 + 0.5 * uy2 - 1.0) + cs4 + ux2 * uy2 - ux2 - uy2 + 1.0;
  e[1] = 0.5 * (-cs2 * (ux2 + uy2 + ux - 1.0) - cs4 - ux2 * uy2 + ux2 - uy2 * ux + ux);
  e[2] = 0.5 * (-cs2 * (ux2 + uy2 + uy - 1.0) - cs4 - ux2 * uy2 - ux2 * uy + uy2 + uy);
  e[3] = 0.5 * (-cs2 * (ux2 + uy2 - ux - 1.0) - cs4 - ux2 * uy2 + ux2 + uy2 * ux - ux);
  e[4] = 0.5 * (-cs2 * (ux2 + uy2 - uy - 1.0) - cs4 - ux2 * uy2 + ux2 * uy + uy2 - uy);
  e[5] = 0.25 * (cs2 * (ux2 + uy2 + ux + uy) + cs4 + ux2 * uy2 + ux2 * uy + uy2 * ux + ux * uy);
  e[6] = 0.25 * (cs2 * (ux2 + uy2 - ux + uy) + cs4 + ux2 * uy2 + ux2 * uy - uy2 * ux - ux * uy);
  e[7] = 0.25 * (cs2 * (ux2 + uy2 - ux - uy) + cs4 + ux2 * uy2 - ux2 * uy - uy2 * ux + ux * uy);
  e[8] = 0.25 * (cs2 * (ux2 + uy2 + ux - uy) + cs4 + ux2 * uy2 - ux2 * uy + uy2 * ux - ux * uy);
}

inline double kbc_node_collide(double* coll, const double* f, double m0, double ux, double uy,
                               double s2) {
  const double is2 = 1.0 / s2;
  const double ux2 = ux * ux, uy2 = uy * uy;  // :150-155
  // eval_central_momenta :265-320
  double cT[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = 0; q < 9; ++q) {
    const double cmx = CX[q] - ux, cmy = CY[q] - uy;
    const double cmx2 = cmx * cmx, cmy2 = cmy * cmy;
    cT[0] += f[q];
    cT[1] += f[q] * cmx;
    cT[2] += f[q] * cmy;
    cT[3] += f[q] * (cmx2 + cmy2);
    cT[4] += f[q] * (cmx2 - cmy2);
    cT[5] += f[q] * cmx * cmy;
    cT[6] += f[q] * cmx2 * cmy;
    cT[7] += f[q] * cmx * cmy2;
    cT[8] += f[q] * cmx2 * cmy2;
  }
  const double C3 = cT[3], C4 = cT[4], C5 = cT[5], C6 = cT[6], C7 = cT[7], C8 = cT[8];
  const double D3 = C3 - 2.0 * cs2 * m0;
  double ds[9], dh[9], ie[9];
  // eval_delta_s :157-192
  ds[0] = -0.5 * C4 * (ux2 - uy2) + 4.0 * C5 * ux * uy - cs4 * m0 - m0 * (ux2 * uy2 - ux2 - uy2 + 1) + D3 * (0.5 * ux2 + 0.5 * uy2 - 1.0);
  ds[1] = 0.25 * C4 * (ux2 - uy2 + ux + 1) - C5 * uy * (2.0 * ux + 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 + uy2 * ux - ux) - 0.25 * D3 * (ux2 + uy2 + ux - 1.0);
  ds[2] = -0.25 * C4 * (-ux2 + uy2 + uy + 1) - C5 * ux * (2.0 * uy + 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - uy2 + ux2 * uy - uy) - 0.25 * D3 * (ux2 + uy2 + uy - 1.0);
  ds[3] = 0.25 * C4 * (ux2 - uy2 - ux + 1) - C5 * uy * (2.0 * ux - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 - uy2 * ux + ux) - 0.25 * D3 * (ux2 + uy2 - ux - 1.0);
  ds[4] = 0.25 * C4 * (ux2 - uy2 + uy - 1) - C5 * ux * (2.0 * uy - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - uy2 - ux2 * uy + uy) - 0.25 * D3 * (ux2 + uy2 - uy - 1.0);
  ds[5] = -0.125 * C4 * (ux2 - uy2 + ux - uy) + C5 * (ux * uy + 0.5 * ux + 0.5 * uy + 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 * uy + uy2 * ux + ux * uy) + 0.125 * D3 * (ux2 + uy2 + ux + uy);
  ds[6] = 0.125 * C4 * (-ux2 + uy2 + ux + uy) + C5 * (ux * uy + 0.5 * ux - 0.5 * uy - 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 * uy - uy2 * ux - ux * uy) + 0.125 * D3 * (ux2 + uy2 - ux + uy);
  ds[7] = -0.125 * C4 * (ux2 - uy2 - ux + uy) + C5 * (ux * uy - 0.5 * ux - 0.5 * uy + 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 * uy - uy2 * ux + ux * uy) + 0.125 * D3 * (ux2 + uy2 - ux - uy);
  ds[8] = -0.125 * C4 * (ux2 - uy2 + ux + uy) + C5 * (ux * uy - 0.5 * ux + 0.5 * uy - 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 * uy + uy2 * ux - ux * uy) + 0.125 * D3 * (ux2 + uy2 + ux - uy);
  // eval_delta_h :194-228, rows 5-8 keep the reference's "ux2+uy" 
  dh[0] = 2.0 * C6 * uy + 2.0 * C7 * ux + C8 - 2.0 * cs2 * m0 * (0.5 * ux2 + 0.5 * uy2 - 1.0) - cs4 * m0 - m0 * (ux2 * uy2 - ux2 - uy2 + 1.0);
  dh[1] = -C6 * uy - C7 * (ux + 0.5) - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 + ux - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 + uy2 * ux - ux);
  dh[2] = -C6 * (uy + 0.5) - C7 * ux - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 + uy - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 + ux2 * uy - uy2 - uy);
  dh[3] = -C6 * uy - C7 * (ux - 0.5) - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 - ux - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 - uy2 * ux + ux);
  dh[4] = -C6 * (uy - 0.5) - C7 * ux - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 - uy - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 * uy - uy2 + uy);
  dh[5] = C6 * (0.5 * uy + 0.25) + C7 * (0.5 * ux + 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 + ux + uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 + uy + uy2 * ux + ux * uy);
  dh[6] = C6 * (0.5 * uy + 0.25) + C7 * (0.5 * ux - 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 - ux + uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 + uy - uy2 * ux - ux * uy);
  dh[7] = C6 * (0.5 * uy - 0.25) + C7 * (0.5 * ux - 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 - ux - uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 + uy - uy2 * ux + ux * uy);
  dh[8] = C6 * (0.5 * uy - 0.25) + C7 * (0.5 * ux + 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 + ux - uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 + uy + uy2 * ux - ux * uy);
  // eval_iequilibrium :230-246
  kbc_feq_poly(ie, ux, uy, ux2, uy2);
  for (int q = 0; q < 9; ++q) ie[q] = 1.0 / (ie[q] * m0);
  // eval_gamma :138-148
  double num = 0.0, den = 0.0;
  for (int q = 0; q < 9; ++q) {
    num += ds[q] * dh[q] * ie[q];
    den += dh[q] * dh[q] * ie[q];
  }
  const double gamma = is2 - (1.0 - is2) * num / den;
  // collide :98-125
  cT[0] += -m0;
  cT[3] += -2.0 * cs2 * m0;
  cT[8] += -cs4 * m0;
  const double S[9] = {1.0, 1.0, 1.0, s2, s2, s2, gamma * s2, gamma * s2, gamma * s2};
  for (int k = 0; k < 9; ++k) cT[k] *= S[k];
  const double T0 = cT[0], T1 = cT[1], T2 = cT[2], T3 = cT[3], T4 = cT[4], T5 = cT[5],
               T6 = cT[6], T7 = cT[7], T8 = cT[8];
  double i0 = T0;
  double i1 = T0 * ux + T1;
  double i2 = T0 * uy + T2;
  double i3 = T0 * (ux2 + uy2) + 2.0 * T1 * ux + 2.0 * T2 * uy + T3;
  double i4 = T0 * (ux2 - uy2) + 2.0 * T1 * ux - 2.0 * T2 * uy + T4;
  double i5 = T0 * ux * uy + T1 * uy + T2 * ux + T5;
  double i6 = T0 * ux2 * uy + 2.0 * T1 * ux * uy + T2 * ux2 + 0.5 * T3 * uy + 0.5 * T4 * uy + 2.0 * T5 * ux + T6;
  double i7 = T0 * ux * uy2 + T1 * uy2 + 2.0 * T2 * ux * uy + 0.5 * T3 * ux - 0.5 * T4 * ux + 2.0 * T5 * uy + T7;
  double i8 = T0 * ux2 * uy2 + 2.0 * T1 * ux * uy2 + 2.0 * T2 * ux2 * uy + 0.5 * T3 * (ux2 + uy2) - 0.5 * T4 * (ux2 - uy2) + 4.0 * T5 * ux * uy + 2.0 * T6 * uy + 2.0 * T7 * ux + T8;
  (void)i0;
  double o[9];
  o[0] = i0 - i3 + i8;
  o[1] = 0.5 * i1 + 0.25 * i3 + 0.25 * i4 - 0.5 * i7 - 0.5 * i8;
  o[2] = 0.5 * i2 + 0.25 * i3 - 0.25 * i4 - 0.5 * i6 - 0.5 * i8;
  o[3] = -0.5 * i1 + 0.25 * i3 + 0.25 * i4 + 0.5 * i7 - 0.5 * i8;
  o[4] = -0.5 * i2 + 0.25 * i3 - 0.25 * i4 + 0.5 * i6 - 0.5 * i8;
  o[5] = 0.25 * (i5 + i6 + i7 + i8);
  o[6] = 0.25 * (-i5 + i6 - i7 + i8);
  o[7] = 0.25 * (i5 - i6 - i7 + i8);
  o[8] = 0.25 * (-i5 - i6 + i7 + i8);
  for (int q = 0; q < 9; ++q) coll[q] = o[q] * -1.0 + f[q];
  return gamma;
}
}  // namespace

void orc_kbc_equilibrium(double* feq, const double* m0, const double* m1, int use_zero_u2, int R,
                         int C) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C; ++i) {
    const double ux = m1[2 * i], uy = m1[2 * i + 1];
    const double ux2 = use_zero_u2 ? 0.0 : ux * ux, uy2 = use_zero_u2 ? 0.0 : uy * uy;
    double e[9];
    kbc_feq_poly(e, ux, uy, ux2, uy2);
    for (int q = 0; q < 9; ++q) feq[i * 9 + q] = e[q] * m0[i];  // :262
  }
}

void orc_kbc_collide(double* coll, const double* f, const double* m0, const double* m1, double s2,
                     int R, int C, double* gamma_out) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)R * C; ++i) {
    const double g = kbc_node_collide(coll + i * 9, f + i * 9, m0[i], m1[2 * i], m1[2 * i + 1], s2);
    if (gamma_out) gamma_out[i] = g;
  }
}

void orc_kbc_steps(double* f, double* m0, double* m1, int R, int C, double s2, int nsteps) {
  const size_t N = (size_t)R * C;
  small_grid_guard sg(N);
  std::vector<double> coll(N * 9);
  for (int t = 0; t < nsteps; ++t) {
    orc_kbc_collide(coll.data(), f, m0, m1, s2, R, C, nullptr);  // :119
    advect(f, coll.data(), R, C);                                // :120 (== solver::advect)
    orc_calc_rho(m0, f, R, C);                                   // :141
    orc_calc_u(m1, f, m0, R, C);                                 // :142
  }
}

// test/ulbm_poiseuille.cpp:104-141 loop body, nsteps times, from the driver's start state if
// init != 0 (adve_f = 0 from the ctor, m0 = 1 (:86), m1 = 0): kbc.collide(); pressure-periodic rows
// with solver::incomp_equilibrium for the imposed density and kbc.iequi_f.pow(-1) as f_equi
// (:36-58, :122); advect; halfway bounce-back columns (:126-132); m0 = sum f, m1 = f c^T / m0.
void orc_upo_steps(double* f, double* m0, double* m1, int H, int W, double s2, double rho_inlet,
                   double rho_outlet, int init, int nsteps) {
  const size_t N = (size_t)H * W;
  small_grid_guard sg(N);
  if (init) {
    std::fill(f, f + N * 9, 0.0);
    std::fill(m0, m0 + N, 1.0);
    std::fill(m1, m1 + N * 2, 0.0);
  }
  std::vector<double> coll(N * 9);
  for (int t = 0; t < nsteps; ++t) {
    orc_kbc_collide(coll.data(), f, m0, m1, s2, H, W, nullptr);  // :121
    // periodic_boundary_condition(kbc.coll_f, kbc.iequi_f.pow(-1), kbc.m1, kbc.m0, ...) :122
    // (reads rows 1 and H-2 of coll, writes rows 0 and H-1: no overlap for H >= 4)
    for (int e = 0; e < 2; ++e) {
      const int dst = e ? H - 1 : 0, src = e ? 1 : H - 2;
      const double rho_bc = (e ? rho_outlet : rho_inlet) * 1.0;  // rho_inlet * temp_rho (ones)
      for (int c = 0; c < W; ++c) {
        const size_t i = nid(src, c, W);
        const double ux = m1[2 * i], uy = m1[2 * i + 1];
        double te[9], pe[9];
        node_feq_incomp(te, rho_bc, ux, uy);                 // :50 / :54
        kbc_feq_poly(pe, ux, uy, ux * ux, uy * uy);          // eval_iequilibrium, ulbm.cpp:230-246
        for (int q = 0; q < 9; ++q) {
          const double iequi = 1.0 / (pe[q] * m0[i]);        // what collide() stored
          const double fequi = 1.0 / iequi;                  // .pow(-1)
          coll[nid(dst, c, W) * 9 + q] = (te[q] + coll[i * 9 + q]) - fequi;  // :51 / :55
        }
      }
    }
    advect(f, coll.data(), H, W);  // :123
    for (int r = 0; r < H; ++r) {  // :126-132
      const size_t a = nid(r, W - 1, W) * 9, b = nid(r, 0, W) * 9;
      f[a + 4] = coll[a + 2];
      f[a + 7] = coll[a + 5];
      f[a + 8] = coll[a + 6];
      f[b + 2] = coll[b + 4];
      f[b + 5] = coll[b + 7];
      f[b + 6] = coll[b + 8];
    }
    orc_calc_rho(m0, f, H, W);      // :136
    orc_calc_u(m1, f, m0, H, W);    // :138
  }
}

void orc_kbc_shear_init(double* m0, double* m1, int R, int C, double u_max, double alpha,
                        double delta) {
  // ulbm_double_shear_flow.cpp:42-63 ("R" used for both dimensions, 6.2832 for 2*pi)
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      const size_t i = nid(r, c, C);
      m1[2 * i] = u_max * std::tanh(alpha * (0.25 * R - std::abs(c - 0.5 * R)));
      m0[i] = 1.0;
      m1[2 * i + 1] = u_max * delta * std::sin(6.2832 * (r + 0.25 * R) / R);
    }
}

// ---------------------------------------------------------------------------------
// differential
// ---------------------------------------------------------------------------------
namespace {
const double XI[5][5] = {{1.0, 32.0, 84.0, 32.0, 1.0},
                         {32.0, 448.0, 960.0, 448.0, 32.0},
                         {84.0, 960.0, 0.0, 960.0, 84.0},
                         {32.0, 448.0, 960.0, 448.0, 32.0},
                         {1.0, 32.0, 84.0, 32.0, 1.0}};
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// conv2d (cross-correlation) with weight (xi*kernel), replicate padding 2
// differential.hpp:9-40, differential.cpp:3-33.  dir 0: d/d(row) ("x"), 1: d/d(col) ("y").
void diff5(double* out, const double* psi, int R, int C, int dir) {
  double w[5][5];
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j)
      w[i][j] = ((1.0 / 5040.0) * XI[i][j]) * (dir == 0 ? (double)(i - 2) : (double)(j - 2));
#pragma omp parallel for schedule(static)
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      double s = 0.0;
      for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j)
          s += w[i][j] * psi[nid(clampi(r + i - 2, 0, R - 1), clampi(c + j - 2, 0, C - 1), C)];
      out[nid(r, c, C)] = s;
    }
}
}  // namespace

void orc_diff_x(double* out, const double* psi, int R, int C) { diff5(out, psi, R, C, 0); }
void orc_diff_y(double* out, const double* psi, int R, int C) { diff5(out, psi, R, C, 1); }

// ---------------------------------------------------------------------------------
// colour-gradient MRT (test/mrtcg_rayleigh_taylor.cpp + src/colour.cpp)
// ---------------------------------------------------------------------------------
namespace {
// :130-140 and :146-156
const double Mm[9][9] = {{1, 1, 1, 1, 1, 1, 1, 1, 1},     {-4, -1, -1, -1, -1, 2, 2, 2, 2},
                         {4, -2, -2, -2, -2, 1, 1, 1, 1}, {0, 1, 0, -1, 0, 1, -1, -1, 1},
                         {0, -2, 0, 2, 0, 1, -1, -1, 1},  {0, 0, 1, 0, -1, 1, 1, -1, -1},
                         {0, 0, -2, 0, 2, 1, 1, -1, -1},  {0, 1, -1, 1, -1, 0, 0, 0, 0},
                         {0, 0, 0, 0, 0, 1, -1, 1, -1}};
const double Mi36[9][9] = {{4, -4, 4, 0, 0, 0, 0, 0, 0},     {4, -1, -2, 6, -6, 0, 0, 9, 0},
                           {4, -1, -2, 0, 0, 6, -6, -9, 0},  {4, -1, -2, -6, 6, 0, 0, 9, 0},
                           {4, -1, -2, 0, 0, -6, 6, -9, 0},  {4, 2, 1, 6, 3, 6, 3, 0, 9},
                           {4, 2, 1, -6, -3, 6, 3, 0, -9},   {4, 2, 1, -6, -3, -6, -3, 0, 9},
                           {4, 2, 1, 6, 3, -6, -3, 0, -9}};
const double Bq[9] = {-4.0 / 27.0, 2.0 / 27.0, 2.0 / 27.0, 2.0 / 27.0, 2.0 / 27.0,
                      5.0 / 108.0, 5.0 / 108.0, 5.0 / 108.0, 5.0 / 108.0};  // :158-163

struct colour_consts {  // src/colour.cpp:11-64
  double rho_0, alpha, nu, beta, cs2, phi[9], eta[9];
  explicit colour_consts(const orc_colour_params& p)
      : rho_0(p.rho_0), alpha(p.alpha), nu(p.nu), beta(p.beta) {
    cs2 = 3.0 * (1.0 - alpha) / 5.0;
    const double a = 0.2 * (1.0 - alpha), b = 0.05 * (1.0 - alpha);
    const double ph[9] = {alpha, a, a, a, a, b, b, b, b};
    for (int q = 0; q < 9; ++q) {
      phi[q] = ph[q];
      const double e2 = CX[q] * CX[q] + CY[q] * CY[q];
      eta[q] = 1.0 + 0.5 * (3.0 * cs2 - 1.0) * (3.0 * e2 - 4.0);
    }
  }
  double omega() const { return 1.0 / (0.5 + nu / cs2); }  // :57-58 of the driver
};

struct relax_fn {  // mrtcg_rayleigh_taylor.cpp:34-101 (blends omegas)
  double delta, r_omega, b_omega, s1, s2, s3, t2, t3;
  relax_fn(const colour_consts& r, const colour_consts& b, double d) : delta(d) {
    r_omega = r.omega();
    b_omega = b.omega();
    s1 = 2.0 * r_omega * b_omega / (r_omega + b_omega);
    s2 = 2.0 * (r_omega - s1) / delta;
    s3 = -s2 / (2.0 * delta);
    t2 = 2.0 * (s1 - b_omega) / delta;
    t3 = t2 / (2.0 * delta);
  }
  double eval(double psi, double prev) const {
    double v = prev;
    if (psi > delta) v = r_omega;
    if (delta >= psi && psi > 0.0) v = s1 + s2 * psi + s3 * psi * psi;
    if (0.0 >= psi && psi >= -delta) v = s1 + t2 * psi + t3 * psi * psi;
    if (psi < -delta) v = b_omega;
    return v;
  }
};

inline void cg_feq(double* e, double rho_k, const colour_consts& k, double ux, double uy) {
  // :233-247 (9(c.u)^2 - 3u.u)
  const double uu = ux * ux + uy * uy;
  for (int q = 0; q < 9; ++q) {
    const double cu = ux * CX[q] + uy * CY[q];
    e[q] = rho_k * (k.phi[q] + W9[q] * (3.0 * cu * k.eta[q] + 9.0 * (cu * cu) - 3.0 * uu));
  }
}

void cg_boundary(double* adv, const double* col, int R, int C) {
  // apply_boundary_conditions :495-533 (column copies WITHOUT row shift)
  for (int r = 1; r < R - 1; ++r) {
    const size_t w = nid(r, 0, C) * 9, e = nid(r, C - 1, C) * 9;
    adv[w + 2] = col[e + 2];
    adv[w + 5] = col[e + 5];
    adv[w + 6] = col[e + 6];
    adv[e + 4] = col[w + 4];
    adv[e + 8] = col[w + 8];
    adv[e + 7] = col[w + 7];
  }
  for (int c = 0; c < C; ++c) {
    const size_t b = nid(R - 1, c, C) * 9, t = nid(0, c, C) * 9;
    adv[b + 3] = col[b + 1];
    adv[b + 7] = col[b + 5];
    adv[b + 6] = col[b + 8];
    adv[t + 1] = col[t + 3];
    adv[t + 5] = col[t + 7];
    adv[t + 8] = col[t + 6];
  }
}
}  // namespace

void orc_cg_init(const orc_cg_params* p, double* f_r, double* f_b, double* rho_r, double* rho_b,
                 double* u) {
  const int R = p->R, C = p->C;
  const colour_consts kr(p->red), kb(p->blue);
  const double middle = R / 2.0;  // :182-210
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      const double s = middle - 0.1 * C * std::cos(2.0 * 3.141592 * c / C);
      const size_t i = nid(r, c, C);
      rho_r[i] = kr.rho_0 * ((r < s) ? 1.0 : 0.0);
      rho_b[i] = kb.rho_0 * ((r >= s) ? 1.0 : 0.0);
      u[2 * i] = u[2 * i + 1] = 0.0;
      cg_feq(f_r + i * 9, rho_r[i], kr, 0.0, 0.0);  // :409-410
      cg_feq(f_b + i * 9, rho_b[i], kb, 0.0, 0.0);
    }
}

void orc_cg_init_droplet(const orc_cg_params* p, double* f_r, double* f_b, double* rho_r,
                         double* rho_b, double* u) {
  const int R = p->R, C = p->C;
  const colour_consts kr(p->red), kb(p->blue);
  const double center = R / 2.0, radius = 25.0;  // mrtcg_static_droplet.cpp:188-189
  auto sigmoid = [](double x) { return 1.0 / (1.0 + std::exp(-x)); };  // :180
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      const double s = std::sqrt((r - center) * (r - center) + (c - center) * (c - center));
      const size_t i = nid(r, c, C);
      rho_r[i] = kr.rho_0 * (1.0 - sigmoid(1.0 * (s - radius)));  // invert = true  (:197)
      rho_b[i] = kb.rho_0 * sigmoid(1.0 * (s - radius));          // :198
      const double rho = rho_r[i] + rho_b[i];                      // :456
      u[2 * i] = 0.0 + 0.5 * p->g_r / rho;                         // :457
      u[2 * i + 1] = 0.0 + 0.5 * p->g_c / rho;
      cg_feq(f_r + i * 9, rho_r[i], kr, u[2 * i], u[2 * i + 1]);  // :458-459
      cg_feq(f_b + i * 9, rho_b[i], kb, u[2 * i], u[2 * i + 1]);
    }
}

void orc_cg_steps(const orc_cg_params* p, double* f_r, double* f_b, double* rho_r, double* rho_b,
                  double* u, int nsteps, double* psi_out, double* snu_out, double* col_r_out,
                  double* col_b_out) {
  const int R = p->R, C = p->C;
  const size_t N = (size_t)R * C;
  small_grid_guard sg(N);
  const colour_consts kr(p->red), kb(p->blue);
  const relax_fn relax(kr, kb, p->delta);
  const double g_r = p->g_r, g_c = p->g_c, sigma = p->sigma;
  std::vector<double> rho(N), psi(N, 0.0), snu(N, 0.0), Qx(N), Qy(N), DxQx_r(N), DyQy_r(N),
      DxQx_b(N), DyQy_b(N), gx(N), gy(N), col_r(N * 9), col_b(N * 9);
  for (size_t i = 0; i < N; ++i) rho[i] = rho_r[i] + rho_b[i];  // :407 / :474
  double unitx[9], unity[9];  // :176-178
  for (int q = 0; q < 9; ++q) {
    const double d = (q < 5) ? 1.0 : std::sqrt(2);
    unitx[q] = CX[q] / d;
    unity[q] = CY[q] / d;
  }
  for (int t = 0; t < nsteps; ++t) {
    // :434-435 phase field and relaxation blend
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
      psi[i] = (rho_r[i] / kr.rho_0 - rho_b[i] / kb.rho_0) / (rho_r[i] / kr.rho_0 + rho_b[i] / kb.rho_0);
      snu[i] = relax.eval(psi[i], snu[i]);
    }
    // :436-437 update_C: 5x5 derivatives of Q = (1.8 alpha - 0.8) rho_k u
    for (int k = 0; k < 2; ++k) {
      const colour_consts& kc = k ? kb : kr;
      const double* rk = k ? rho_b : rho_r;
      for (size_t i = 0; i < N; ++i) {
        Qx[i] = (1.8 * kc.alpha - 0.8) * rk[i] * u[2 * i];
        Qy[i] = (1.8 * kc.alpha - 0.8) * rk[i] * u[2 * i + 1];
      }
      diff5(k ? DxQx_b.data() : DxQx_r.data(), Qx.data(), R, C, 0);
      diff5(k ? DyQy_b.data() : DyQy_r.data(), Qy.data(), R, C, 1);
    }
    diff5(gx.data(), psi.data(), R, C, 0);  // :443
    diff5(gy.data(), psi.data(), R, C, 1);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
      const double ux = u[2 * i], uy = u[2 * i + 1];
      const double s_nu = snu[i];
      const double S[9] = {0.0, 1.25, 1.14, 0.0, 1.6, 0.0, 1.6, s_nu, s_nu};  // :384-386, :227-231
      double om1[2][9];
      for (int k = 0; k < 2; ++k) {
        const colour_consts& kc = k ? kb : kr;
        const double* fk = (k ? f_b : f_r) + i * 9;
        const double rk = k ? rho_b[i] : rho_r[i];
        double feq[9], m[9], Ck[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        cg_feq(feq, rk, kc, ux, uy);  // :431-432
        const double dxq = k ? DxQx_b[i] : DxQx_r[i], dyq = k ? DyQy_b[i] : DyQy_r[i];
        Ck[1] = 3.0 * (1.0 - 0.5 * 1.25) * (dxq + dyq);  // :329-331
        Ck[7] = (1.0 - 0.5 * s_nu) * (dxq - dyq);         // :332-334
        for (int a = 0; a < 9; ++a) {                      // :249-261
          double s = 0.0;
          for (int q = 0; q < 9; ++q) s += Mm[a][q] * (feq[q] - fk[q]);
          m[a] = S[a] * s + Ck[a];
        }
        for (int q = 0; q < 9; ++q) {
          double s = 0.0;
          for (int a = 0; a < 9; ++a) s += ((1.0 / 36.0) * Mi36[q][a]) * m[a];
          om1[k][q] = s;
        }
      }
      const double gnorm = std::sqrt(gx[i] * gx[i] + gy[i] * gy[i]);  // :444-447
      const double A = 4.5 * sigma * s_nu;                            // :450
      const double rr = rho_r[i], rb = rho_b[i], rt = rho[i];
      double* cr = col_r.data() + i * 9;
      double* cb = col_b.data() + i * 9;
      for (int q = 0; q < 9; ++q) {
        const double gE = gx[i] * CX[q] + gy[i] * CY[q];
        const double t1 = gE / (1e-20 + gnorm);
        const double xi = 0.5 * gnorm * (W9[q] * (t1 * t1) - Bq[q]);  // :290-300
        const double om2 = A * xi;                                     // :263-273
        const double gU = gx[i] * unitx[q] + gy[i] * unity[q];
        const double kappa = (rr * rb * gU * (rr * kr.phi[q] + rb * kb.phi[q])) /
                             ((rt * rt) * (1e-20 + gnorm));            // :302-318
        const double tot = f_r[i * 9 + q] + om1[0][q] + om2 + f_b[i * 9 + q] + om1[1][q] + om2;  // :455
        const double cu = ux * CX[q] + uy * CY[q];
        const double FgE = g_r * CX[q] + g_c * CY[q];
        const double uFg = ux * g_r + uy * g_c;
        const double Fq = (1 - 0.5 * s_nu) * ((3.0 + 9.0 * cu) * FgE - 3.0 * uFg) * W9[q];  // :460-462
        const double o3r = rr * tot / rt + kr.beta * kappa;  // :275-288
        const double o3b = rb * tot / rt + kb.beta * kappa;
        cr[q] = p->add_source ? o3r + Fq : o3r;  // :463 / mrtcg_static_droplet.cpp:513
        cb[q] = p->add_source ? o3b + Fq : o3b;  // :464 / :514
      }
    }
    advect(f_r, col_r.data(), R, C);  // :466-467
    advect(f_b, col_b.data(), R, C);
    cg_boundary(f_r, col_r.data(), R, C);  // :469-470
    cg_boundary(f_b, col_b.data(), R, C);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {  // :472-477
      rho_r[i] = node_rho(f_r + i * 9);
      rho_b[i] = node_rho(f_b + i * 9);
      rho[i] = rho_r[i] + rho_b[i];
      double jx = 0.0, jy = 0.0;
      for (int q = 0; q < 9; ++q) {
        const double ft = f_r[i * 9 + q] + f_b[i * 9 + q];
        jx += ft * CX[q];
        jy += ft * CY[q];
      }
      u[2 * i] = jx / rho[i] + 0.5 * g_r / rho[i];
      u[2 * i + 1] = jy / rho[i] + 0.5 * g_c / rho[i];
    }
  }
  if (psi_out) std::memcpy(psi_out, psi.data(), N * sizeof(double));
  if (snu_out) std::memcpy(snu_out, snu.data(), N * sizeof(double));
  if (col_r_out) std::memcpy(col_r_out, col_r.data(), N * 9 * sizeof(double));
  if (col_b_out) std::memcpy(col_b_out, col_b.data(), N * 9 * sizeof(double));
}

// ---------------------------------------------------------------------------------
// immersed boundary (src/ibm.cpp) and cylinder driver (test/cylinder_test.cpp)
// ---------------------------------------------------------------------------------
namespace {
inline double peskin4(double r_) {  // ibm.cpp:39-45
  const double r = std::abs(r_);
  if (r <= 1) return 0.125 * (3.0 - 2.0 * r + std::sqrt(1.0 + 4.0 * r - 4.0 * r * r));
  else if (r <= 2) return 0.125 * (5.0 - 2.0 * r - std::sqrt(-7.0 + 12.0 * r - 4.0 * r * r));
  return 0.0;
}
struct marker_t {
  int row0, col0;  // box = rows [row0,row0+4) x cols [col0,col0+4) of the ROI
  double phi[16];  // indexed k = i*4 + j over the box flattened [row i][col j]
};
marker_t make_marker(double x, double y) {  // ibm.cpp:20-37 (x, y relative to the ROI origin)
  marker_t m;
  const double fx = std::floor(x), fy = std::floor(y);
  for (int k = 0; k < 16; ++k) {
    // stencil row 0 = k%4 pairs with x, row 1 = k/4 pairs with y (ibm.cpp:11-13,26) while the
    // box is flattened [row=k/4][col=k%4] (:171-181): kernels transposed, SURVEY Q9.
    const double sx = x - ((double)(k % 4) + fx - 1.0);
    const double sy = y - ((double)(k / 4) + fy - 1.0);
    m.phi[k] = peskin4(sx) * peskin4(sy);
  }
  m.row0 = (int)fx - 1;
  m.col0 = (int)fy - 1;
  return m;
}
}  // namespace

void orc_ibm_roi(const orc_ibm_markers* m, int* r0, int* r1, int* c0, int* c1) {
  long r_min = 1000000, r_max = 0, c_min = 1000000, c_max = 0;  // ibm.cpp:124-153
  for (int i = 0; i < m->n_markers; ++i) {
    const int fx = (int)std::floor(m->x[i]), fy = (int)std::floor(m->y[i]);
    r_min = std::min<long>(r_min, fx - 2);
    r_max = std::max<long>(r_max, fx + 2);
    c_min = std::min<long>(c_min, fy - 2);
    c_max = std::max<long>(c_max, fy + 2);
  }
  *r0 = (int)r_min;
  *r1 = (int)r_max + 1;
  *c0 = (int)c_min;
  *c1 = (int)c_max + 1;
}

void orc_ibm_force(const orc_ibm_markers* mk, const double* u_0, const double* rho_0, int X, int Y,
                   double* F_out) {
  (void)X;
  int r0, r1, c0, c1;
  orc_ibm_roi(mk, &r0, &r1, &c0, &c1);
  const int RR = r1 - r0, RC = c1 - c0;
  std::vector<marker_t> ms;
  for (int i = 0; i < mk->n_markers; ++i) ms.push_back(make_marker(mk->x[i] - r0, mk->y[i] - c0));
  std::vector<double> u((size_t)RR * RC * 2), rho((size_t)RR * RC), Fn((size_t)RR * RC * 2);
  for (int r = 0; r < RR; ++r)  // :163-164
    for (int c = 0; c < RC; ++c) {
      const size_t s = nid(r0 + r, c0 + c, Y), d = nid(r, c, RC);
      u[2 * d] = u_0[2 * s];
      u[2 * d + 1] = u_0[2 * s + 1];
      rho[d] = rho_0[s];
    }
  std::fill(F_out, F_out + (size_t)RR * RC * 2, 0.0);
  for (int n = 1; n < mk->m_max; ++n) {  // :166-187
    std::fill(Fn.begin(), Fn.end(), 0.0);
    for (const marker_t& m : ms) {
      double ujx = 0.0, ujy = 0.0, rhoj = 0.0;
      for (int k = 0; k < 16; ++k) {
        const size_t d = nid(m.row0 + k / 4, m.col0 + k % 4, RC);
        ujx += m.phi[k] * u[2 * d];
        ujy += m.phi[k] * u[2 * d + 1];
        rhoj += m.phi[k] * rho[d];
      }
      const double fjx = -2.0 * rhoj * ujx, fjy = -2.0 * rhoj * ujy;
      for (int k = 0; k < 16; ++k) {
        const size_t d = nid(m.row0 + k / 4, m.col0 + k % 4, RC);
        Fn[2 * d] += m.phi[k] * fjx;
        Fn[2 * d + 1] += m.phi[k] * fjy;
      }
    }
    for (size_t d = 0; d < (size_t)RR * RC; ++d) {
      u[2 * d] += 0.5 * Fn[2 * d] / rho[d];
      u[2 * d + 1] += 0.5 * Fn[2 * d + 1] / rho[d];
      F_out[2 * d] += Fn[2 * d];  // torch::sum(F, 3), :189
      F_out[2 * d + 1] += Fn[2 * d + 1];
    }
  }
}

void orc_cylinder_steps(const orc_ibm_markers* mk, double* f_adve, double* u, double* rho, int X,
                        int Y, double omega, double u_in, int nsteps, double* Fs) {
  const size_t N = (size_t)X * Y;
  small_grid_guard sg(N);
  int r0, r1, c0, c1;
  orc_ibm_roi(mk, &r0, &r1, &c0, &c1);
  const int RR = r1 - r0, RC = c1 - c0;
  std::vector<double> f_equi(N * 9), f_coll(N * 9), F((size_t)RR * RC * 2);
  const double ics2 = 1.0 / 3.0, ics4 = 1.0 / 9.0;  // cylinder_test.cpp:66-67
  double abb[9];                                    // :135, :146 with u_w = (u_in, 0)
  for (int q = 0; q < 9; ++q) {
    const double cu = u_in * CX[q] + 0.0 * CY[q];
    abb[q] = (2.0 + 9.0 * std::pow(cu, 2.0) - 3.0 * (u_in * u_in + 0.0 * 0.0)) * W9[q];
  }
  static const int OPP[9] = {0, 3, 4, 1, 2, 7, 8, 5, 6};
  for (int t = 0; t < nsteps; ++t) {
    orc_calc_rho(rho, f_adve, X, Y);                // :103
    orc_calc_u(u, f_adve, rho, X, Y);               // :104
    orc_equilibrium(f_equi.data(), u, rho, X, Y);   // :107
    orc_ibm_force(mk, u, rho, X, Y, F.data());      // :110
    if (Fs) {                                       // :112
      Fs[0] = Fs[1] = 0.0;
      for (size_t d = 0; d < (size_t)RR * RC; ++d) {
        Fs[0] += F[2 * d];
        Fs[1] += F[2 * d + 1];
      }
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N * 9; ++i)
      f_coll[i] = f_adve[i] + (-omega * (f_adve[i] - f_equi[i]));  // :108, :123-125
    for (int r = 0; r < RR; ++r)                                   // :116-119, :127
      for (int c = 0; c < RC; ++c) {
        const size_t s = nid(r0 + r, c0 + c, Y), d = nid(r, c, RC);
        const double ux = u[2 * s], uy = u[2 * s + 1], Fx = F[2 * d], Fy = F[2 * d + 1];
        const double uF = ux * Fx + uy * Fy;
        for (int q = 0; q < 9; ++q) {
          const double cu = ux * CX[q] + uy * CY[q], cF = Fx * CX[q] + Fy * CY[q];
          f_coll[s * 9 + q] += ((1 - 0.5 * omega) * ((ics2 + ics4 * cu) * cF - ics2 * uF) * W9[q]);
        }
      }
    advect(f_adve, f_coll.data(), X, Y);  // :130
    for (int e = 0; e < 2; ++e) {         // :135-154 anti-bounce-back inlet row 0, outlet row X-1
      const int r = e ? X - 1 : 0;
      for (int c = 0; c < Y; ++c) {
        const size_t n = nid(r, c, Y) * 9;
        for (int q = 1; q < 9; ++q) f_adve[n + OPP[q]] = -f_coll[n + q] + abb[q];
      }
    }
    for (int r = 0; r < X; ++r) {  // :157-163 specular columns
      const size_t e = nid(r, Y - 1, Y) * 9, w = nid(r, 0, Y) * 9;
      f_adve[e + 4] = f_coll[e + 2];
      f_adve[e + 7] = f_coll[e + 6];
      f_adve[e + 8] = f_coll[e + 5];
      f_adve[w + 2] = f_coll[w + 4];
      f_adve[w + 5] = f_coll[w + 8];
      f_adve[w + 6] = f_coll[w + 7];
    }
  }
}

}  // extern "C"
