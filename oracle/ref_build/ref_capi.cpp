// TEST INFRASTRUCTURE ONLY.  C ABI over the *unmodified* reference sources
// (compiled from /root/reference/src by oracle/ref_build/Makefile, CPU libtorch).
// Used (a) to generate tests/golden/*.npz, (b) to validate oracle/lbm_oracle.cpp,
// (c) as bench.py's cpu_baseline leg (kind "reference").  Never linked into or
// called by the product library.
//
// All arrays are host f64 in the reference's own layout: f[R][C][9],
// rho[R][C] (stands for the reference's [R,C,1]), u[R][C][2].
#include <torch/torch.h>
#include <cstring>
#include <memory>

#include "solver.hpp"        // /root/reference/src/solver.hpp
#include "ulbm.hpp"          // /root/reference/src/ulbm.hpp
#include "differential.hpp"  // /root/reference/src/differential.hpp

namespace {
using torch::Tensor;
const auto f64 = torch::TensorOptions().dtype(torch::kDouble);

Tensor wrap(const double* p, std::initializer_list<int64_t> shape) {
  return torch::from_blob(const_cast<double*>(p), shape, f64).clone();
}
void unwrap(double* dst, const Tensor& t) {
  Tensor c = t.contiguous();
  std::memcpy(dst, c.data_ptr<double>(), sizeof(double) * c.numel());
}
void set_f64_default() {
  torch::set_default_dtype(caffe2::scalarTypeToTypeMeta(torch::kDouble));
}
}  // namespace

extern "C" {

int ref_num_threads() { return at::get_num_threads(); }
void ref_set_num_threads(int n) { at::set_num_threads(n); }

// ---- solver:: unit functions (src/solver.cpp:23-131) -------------------------
void ref_calc_rho(double* rho, const double* f, int R, int C) {
  set_f64_default();
  Tensor r = torch::zeros({R, C, 1}, f64);
  solver::calc_rho(r, wrap(f, {R, C, 9}));
  unwrap(rho, r);
}
void ref_calc_u(double* u, const double* f, const double* rho, int R, int C) {
  set_f64_default();
  Tensor uu = torch::zeros({R, C, 2}, f64);
  solver::calc_u(uu, wrap(f, {R, C, 9}), wrap(rho, {R, C, 1}));
  unwrap(u, uu);
}
void ref_calc_incomp_u(double* u, const double* f, int R, int C) {
  set_f64_default();
  Tensor uu = torch::zeros({R, C, 2}, f64);
  solver::calc_incomp_u(uu, wrap(f, {R, C, 9}));
  unwrap(u, uu);
}
void ref_equilibrium(double* feq, const double* u, const double* rho, int R, int C) {
  set_f64_default();
  Tensor e = torch::zeros({R, C, 9}, f64);
  solver::equilibrium(e, wrap(u, {R, C, 2}), wrap(rho, {R, C, 1}));
  unwrap(feq, e);
}
void ref_incomp_equilibrium(double* feq, const double* u, const double* rho, int R, int C) {
  set_f64_default();
  Tensor e = torch::zeros({R, C, 9}, f64);
  solver::incomp_equilibrium(e, wrap(u, {R, C, 2}), wrap(rho, {R, C, 1}));
  unwrap(feq, e);
}
void ref_collision(double* fc, const double* f, const double* feq, double omega, int R, int C) {
  set_f64_default();
  Tensor o = torch::zeros({R, C, 9}, f64);
  solver::collision(o, wrap(f, {R, C, 9}), wrap(feq, {R, C, 9}), omega);
  unwrap(fc, o);
}
void ref_advect(double* g, const double* f, int R, int C) {
  set_f64_default();
  Tensor o = torch::zeros({R, C, 9}, f64);
  solver::advect(o, wrap(f, {R, C, 9}));
  unwrap(g, o);
}

// ---- BGK periodic box: the loop every driver hand-writes ----------------------
// calc_rho -> calc_u -> equilibrium -> collision -> advect  (compressible), or
// calc_rho -> calc_incomp_u -> incomp_equilibrium -> ...    (incompressible),
// e.g. test/horizontal_poiseuille_test.cpp:130-143 without the BC fix-ups.
// f is updated in place (f_adve after n steps); rho/u (optional) receive the
// moments of the LAST pre-collision state, as the drivers' snapshots would.
void ref_bgk_periodic_steps(double* f, double* rho_out, double* u_out, int R, int C,
                            double omega, int incompressible, int nsteps) {
  set_f64_default();
  Tensor f_adve = wrap(f, {R, C, 9});
  Tensor f_equi = torch::zeros_like(f_adve), f_coll = torch::zeros_like(f_adve);
  Tensor u = torch::zeros({R, C, 2}, f64), rho = torch::ones({R, C, 1}, f64);
  for (int t = 0; t < nsteps; ++t) {
    solver::calc_rho(rho, f_adve);
    if (incompressible) {
      solver::calc_incomp_u(u, f_adve);
      solver::incomp_equilibrium(f_equi, u, rho);
    } else {
      solver::calc_u(u, f_adve, rho);
      solver::equilibrium(f_equi, u, rho);
    }
    solver::collision(f_coll, f_adve, f_equi, omega);
    solver::advect(f_adve, f_coll);
  }
  unwrap(f, f_adve);
  if (rho_out) unwrap(rho_out, rho);
  if (u_out) unwrap(u_out, u);
}

// ---- ulbm::d2q9::kbc (src/ulbm.cpp) ---------------------------------------------
// Runs the body of test/ulbm_double_shear_flow.cpp:119-142 (collide, advect,
// moment update; the explicit periodic edge copies at :124-138 rewrite values
// advect() already produced and are therefore omitted) starting from moments
// (m0, m1): adve_f = eval_equilibrium(m0, m1) when init_from_moments != 0,
// otherwise from the given f.
void ref_kbc_steps(double* f, double* m0, double* m1, int R, int C, double s2,
                   int init_from_moments, int nsteps, double* coll_out) {
  set_f64_default();
  ulbm::d2q9::kbc k{R, C, s2};
  k.m0 = wrap(m0, {R, C});
  k.m1 = wrap(m1, {R, C, 2});
  if (init_from_moments) {
    // eval_equilibrium reads ux2/uy2, which only eval_gamma refreshes (private);
    // the reference driver calls it with ux2 = uy2 = 0 left by the ctor
    // (ulbm_double_shear_flow.cpp:96) -- reproduced as is.
    k.eval_equilibrium(k.adve_f);
  } else {
    k.adve_f = wrap(f, {R, C, 9});
  }
  const Tensor c = solver::c;
  for (int t = 0; t < nsteps; ++t) {
    k.collide();
    if (coll_out && t == nsteps - 1) unwrap(coll_out, k.coll_f);
    k.advect();
    k.m0 = k.adve_f.sum(-1).detach().clone();
    k.m1 = (torch::matmul(k.adve_f, c.transpose(0, 1)) / k.m0.unsqueeze(-1)).detach().clone();
  }
  unwrap(f, k.adve_f);
  unwrap(m0, k.m0);
  unwrap(m1, k.m1);
}

// The loop body of test/ulbm_poiseuille.cpp:104-141 driven through the reference's own
// ulbm::d2q9::kbc and solver::incomp_equilibrium (this harness only sequences the calls the way
// that main does; the unmodified main itself is oracle/_ref/upo, 300000 steps).
void ref_upo_steps(double* f, double* m0, double* m1, int H, int W, double s2, double rho_inlet,
                   double rho_outlet, int init, int nsteps) {
  using torch::indexing::Ellipsis;
  using torch::indexing::Slice;
  set_f64_default();
  ulbm::d2q9::kbc k{H, W, s2};
  if (init) {
    k.m0.fill_(1.0);
  } else {
    k.adve_f = wrap(f, {H, W, 9});
    k.m0 = wrap(m0, {H, W});
    k.m1 = wrap(m1, {H, W, 2});
  }
  const Tensor c = solver::c;
  for (int t = 0; t < nsteps; ++t) {
    k.collide();
    {
      Tensor f_equi = k.iequi_f.pow(-1);
      Tensor temp_equi = torch::zeros({1, W, 9});
      Tensor temp_rho = torch::ones({1, W, 1});
      solver::incomp_equilibrium(temp_equi, k.m1.index({-2, Ellipsis}).unsqueeze(0), rho_inlet * temp_rho);
      k.coll_f.index({0, Ellipsis}) =
          (temp_equi + k.coll_f.index({-2, Ellipsis}) - f_equi.index({-2, Ellipsis})).squeeze(0).clone().detach();
      solver::incomp_equilibrium(temp_equi, k.m1.index({1, Ellipsis}).unsqueeze(0), rho_outlet * temp_rho);
      k.coll_f.index({-1, Ellipsis}) =
          (temp_equi + k.coll_f.index({1, Ellipsis}) - f_equi.index({1, Ellipsis})).squeeze(0).clone().detach();
    }
    k.advect();
    k.adve_f.index({Slice(), -1, 4}) = k.coll_f.index({Slice(), -1, 2}).clone().detach();
    k.adve_f.index({Slice(), -1, 7}) = k.coll_f.index({Slice(), -1, 5}).clone().detach();
    k.adve_f.index({Slice(), -1, 8}) = k.coll_f.index({Slice(), -1, 6}).clone().detach();
    k.adve_f.index({Slice(), 0, 2}) = k.coll_f.index({Slice(), 0, 4}).clone().detach();
    k.adve_f.index({Slice(), 0, 5}) = k.coll_f.index({Slice(), 0, 7}).clone().detach();
    k.adve_f.index({Slice(), 0, 6}) = k.coll_f.index({Slice(), 0, 8}).clone().detach();
    k.m0 = k.adve_f.sum(-1).detach().clone();
    k.m1 = (torch::matmul(k.adve_f, c.transpose(0, 1)) / k.m0.unsqueeze(-1)).detach().clone();
  }
  unwrap(f, k.adve_f);
  unwrap(m0, k.m0);
  unwrap(m1, k.m1);
}

// ---- differential (src/differential.cpp:23-33) -----------------------------------
void ref_diff_x(double* out, const double* psi, int R, int C) {
  set_f64_default();
  static differential D{};
  unwrap(out, D.x(wrap(psi, {R, C})));
}
void ref_diff_y(double* out, const double* psi, int R, int C) {
  set_f64_default();
  static differential D{};
  unwrap(out, D.y(wrap(psi, {R, C})));
}

}  // extern "C"
