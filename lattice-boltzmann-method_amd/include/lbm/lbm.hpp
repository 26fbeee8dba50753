// C++ host facade of the MI355X D2Q9 engine: the reference's own API names --
//   solver::{E, c, calc_rho, calc_u, calc_incomp_u, equilibrium, incomp_equilibrium,
//            collision, advect}                         (src/solver.hpp:8-36)
//   struct domain                                       (src/domain.hpp:5-15)
//   ulbm::d2q9::kbc                                     (src/ulbm.hpp:12-89)
//   class differential, class colour, class ibm         (src/differential.hpp, colour.hpp, ibm.hpp)
// re-created over the C ABI (include/lbm_hip.h, liblbm_hip.so) so that the reference's drivers
// can be restated line by line (see ../../drivers/), plus lbm::Solver / lbm::CgSolver: the
// fused time loops the drivers should call instead of seven unfused operators per step.
//
// What replaces torch::Tensor: lbm::Field -- an owning handle on a device array in the
// engine's SoA layout [Q][R][C]; host transfers use the REFERENCE layout [R][C][Q]
// (bit-exact node indexing via lbm_aos_to_soa / lbm_soa_to_aos).
// Errors: every failing C-ABI call becomes std::runtime_error (the reference throws
// c10::Error / std::runtime_error in the same places).  Header-only; needs no HIP headers.
#pragma once
#include <array>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/lbm_hip.h"
#include "toml.hpp"

namespace lbm {

inline void check(int rc) {
  if (rc != LBM_OK) throw std::runtime_error(std::string("lbm: ") + lbm_last_error_string());
}

// Device array [Q][R][C] of doubles (Q = 9 populations, 2 velocity components, 1 scalar).
class Field {
 public:
  Field() = default;
  Field(int R, int C, int Q) : R_(R), C_(C), Q_(Q) {
    void* p = nullptr;
    check(lbm_malloc(&p, bytes()));
    d_ = static_cast<double*>(p);
    check(lbm_memset(d_, 0, bytes(), nullptr));  // torch::zeros
  }
  Field(const Field& o) : Field(o.R_, o.C_, o.Q_) {
    if (o.d_) check(lbm_memcpy_d2d(d_, o.d_, bytes(), nullptr));
  }
  Field(Field&& o) noexcept { swap(o); }
  Field& operator=(Field o) noexcept {
    swap(o);
    return *this;
  }
  ~Field() {
    if (d_) lbm_free(d_);
  }
  void swap(Field& o) noexcept {
    std::swap(d_, o.d_);
    std::swap(R_, o.R_);
    std::swap(C_, o.C_);
    std::swap(Q_, o.Q_);
  }
  int rows() const { return R_; }
  int cols() const { return C_; }
  int q() const { return Q_; }
  size_t numel() const { return (size_t)R_ * C_ * Q_; }
  size_t bytes() const { return numel() * sizeof(double); }
  double* data() { return d_; }
  const double* data() const { return d_; }

  void fill(double v) {
    std::vector<double> h(numel(), v);
    check(lbm_memcpy_h2d(d_, h.data(), bytes(), nullptr));
    check(lbm_stream_sync(nullptr));
  }
  // host <-> device in the reference layout [R][C][Q]
  void from_host(const std::vector<double>& aos) {
    if (aos.size() != numel()) throw std::runtime_error("lbm::Field::from_host: size mismatch");
    Field tmp(R_, C_, Q_);
    check(lbm_memcpy_h2d(tmp.d_, aos.data(), bytes(), nullptr));
    check(lbm_aos_to_soa(d_, tmp.d_, R_, C_, Q_, nullptr));
    check(lbm_stream_sync(nullptr));
  }
  std::vector<double> to_host() const {
    std::vector<double> aos(numel());
    Field tmp(R_, C_, Q_);
    check(lbm_soa_to_aos(tmp.d_, d_, R_, C_, Q_, nullptr));
    check(lbm_memcpy_d2h(aos.data(), tmp.d_, bytes(), nullptr));
    check(lbm_stream_sync(nullptr));
    return aos;
  }

 private:
  double* d_ = nullptr;
  int R_ = 0, C_ = 0, Q_ = 0;
};

inline void require_same(const Field& a, const Field& b, const char* what) {
  if (a.rows() != b.rows() || a.cols() != b.cols())
    throw std::runtime_error(std::string("shape mismatch in ") + what);
}

// ---- fused time loops ---------------------------------------------------------------------
struct BoundarySet : lbm_bc {
  BoundarySet() : lbm_bc{0, 0, 0, 0, 0, 1.0, 1.0, 0.0, 0.0} {}
};

class Solver {  // single-phase BGK / KBC block, wraps lbm_solver
 public:
  static Solver bgk(int R, int C, double omega, bool incompressible, const lbm_bc& bc = BoundarySet(),
                    bool delta_form = false) {
    lbm_bgk_params p{omega, incompressible ? 1 : 0, delta_form ? 1 : 0, 0, 0.0, 0.0, 0.0, 0.0, LBM_FORM_DEFAULT};
    return Solver(LBM_MODEL_BGK, R, C, &p, bc);
  }
  // full parameter block (e.g. the body force of test/gravity_test.cpp: force_mode = 1)
  static Solver bgk(int R, int C, const lbm_bgk_params& p, const lbm_bc& bc) {
    return Solver(LBM_MODEL_BGK, R, C, &p, bc);
  }
  static Solver kbc(int R, int C, double s2, const lbm_bc& bc = BoundarySet()) {
    lbm_kbc_params p{s2, LBM_FORM_DEFAULT};
    return Solver(LBM_MODEL_KBC, R, C, &p, bc);
  }
  Solver(Solver&& o) noexcept : h_(o.h_), R_(o.R_), C_(o.C_) { o.h_ = nullptr; }
  Solver(const Solver&) = delete;
  ~Solver() {
    if (h_) lbm_solver_destroy(h_);
  }
  void set_f(const std::vector<double>& f_aos) { check(lbm_solver_set_f_aos(h_, f_aos.data())); }
  void set_f(const Field& f_soa) { check(lbm_solver_set_f_soa_dev(h_, f_soa.data())); }
  // KBC: the moments the driver HOLDS for its first collide (test/ulbm_poiseuille.cpp:85-86)
  void set_moments(const std::vector<double>& rho, const std::vector<double>& u) {
    check(lbm_solver_set_moments_aos(h_, rho.data(), u.data()));
  }
  std::vector<double> get_f() {
    std::vector<double> f((size_t)R_ * C_ * 9);
    check(lbm_solver_get_f_aos(h_, f.data()));
    return f;
  }
  void step(int n, bool record_moments = false) { check(lbm_solver_step(h_, n, record_moments)); }
  // rho [R][C], u [R][C][2] as the reference's tensors hold them after the iterations run so far
  std::pair<std::vector<double>, std::vector<double>> moments() {
    std::vector<double> rho((size_t)R_ * C_), u((size_t)R_ * C_ * 2);
    check(lbm_solver_get_moments_aos(h_, rho.data(), u.data()));
    return {std::move(rho), std::move(u)};
  }
  void attach(lbm_ibm* ib, double a = 1.0 / 3.0, double b = 1.0 / 9.0) {
    check(lbm_solver_attach_ibm(h_, ib, a, b));
  }
  void sync() { check(lbm_solver_sync(h_)); }
  lbm_solver* handle() { return h_; }
  int rows() const { return R_; }
  int cols() const { return C_; }

 private:
  Solver(int model, int R, int C, const void* prm, const lbm_bc& bc) : R_(R), C_(C) {
    lbm_geom g{R, C, 0, 0, 0};
    check(lbm_solver_create(&h_, model, &g, &bc, prm, nullptr));
  }
  lbm_solver* h_ = nullptr;
  int R_, C_;
};

}  // namespace lbm

// ===============================================================================================
// solver:: -- src/solver.hpp.  NB the reference's naming: E = the 9 lattice WEIGHTS, c = the
// lattice velocities [2][9] (src/solver.cpp:12-21).
// ===============================================================================================
namespace solver {
using lbm::Field;

inline const std::array<double, 9> E = {4.0 / 9.0,  1.0 / 9.0,  1.0 / 9.0,  1.0 / 9.0, 1.0 / 9.0,
                                        1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0, 1.0 / 36.0};
inline const std::array<std::array<double, 9>, 2> c = {{{0.0, 1.0, 0.0, -1.0, 0.0, 1.0, -1.0, -1.0, 1.0},
                                                         {0.0, 0.0, 1.0, 0.0, -1.0, 1.0, 1.0, -1.0, -1.0}}};

// Out-params are written in place (the reference re-binds them to fresh tensors, SURVEY Q2;
// callers cannot tell the difference except through stale aliases).
inline void calc_rho(Field& rho, const Field& f) {
  lbm::require_same(rho, f, "calc_rho");
  lbm::check(lbm_calc_rho(rho.data(), f.data(), f.rows(), f.cols(), nullptr));
}
inline void calc_u(Field& u, const Field& f, const Field& rho) {
  lbm::require_same(u, f, "calc_u");
  lbm::check(lbm_calc_u(u.data(), f.data(), rho.data(), f.rows(), f.cols(), nullptr));
}
inline void calc_incomp_u(Field& u, const Field& f) {
  lbm::require_same(u, f, "calc_incomp_u");
  lbm::check(lbm_calc_incomp_u(u.data(), f.data(), f.rows(), f.cols(), nullptr));
}
inline void collision(Field& f_coll, const Field& f_curr, const Field& f_equi, const double omega) {
  lbm::require_same(f_coll, f_curr, "collision");
  lbm::check(lbm_collision(f_coll.data(), f_curr.data(), f_equi.data(), omega, f_curr.rows(),
                           f_curr.cols(), nullptr));
}
inline void equilibrium(Field& f_eq, const Field& u, const Field& rho) {
  lbm::require_same(f_eq, u, "equilibrium");
  lbm::check(lbm_equilibrium(f_eq.data(), u.data(), rho.data(), u.rows(), u.cols(), nullptr));
}
inline void incomp_equilibrium(Field& f_eq, const Field& u, const Field& rho) {
  lbm::require_same(f_eq, u, "incomp_equilibrium");
  lbm::check(lbm_incomp_equilibrium(f_eq.data(), u.data(), rho.data(), u.rows(), u.cols(), nullptr));
}
inline void advect(Field& g, const Field& f) {
  lbm::require_same(g, f, "advect");
  lbm::check(lbm_advect(g.data(), f.data(), f.rows(), f.cols(), nullptr));
}
}  // namespace solver

// ===============================================================================================
// struct domain -- src/domain.hpp:5-15, src/domain.cpp:3-12
// ===============================================================================================
struct domain {
  const int R;
  const int C;
  lbm::Field adve_f, equi_f, coll_f;
  lbm::Field m_0;  // typically density
  lbm::Field m_1;  // typically momentum
  domain(int R, int C, int Q = 9)
      : R{R}, C{C}, adve_f(R, C, Q), equi_f(R, C, Q), coll_f(R, C, Q), m_0(R, C, 1), m_1(R, C, 2) {}
};

// ===============================================================================================
// ulbm::d2q9::kbc -- src/ulbm.hpp:12-89.  The 17 [R,C,9] scratch tensors of the reference do not
// exist: collide() is one kernel with the nine central moments in registers.
// ===============================================================================================
namespace ulbm::d2q9 {
class kbc {
 public:
  lbm::Field coll_f, adve_f;
  lbm::Field m0;  // [R,C]
  lbm::Field m1;  // [R,C,2]
  kbc(int R, int C, double s2) : coll_f(R, C, 9), adve_f(R, C, 9), m0(R, C, 1), m1(R, C, 2), s2{s2} {}
  void collide() {  // ulbm.cpp:91-126, with the moments the members hold
    lbm_kbc_params p{s2, LBM_FORM_DEFAULT};
    lbm::check(lbm_kbc_collide_given_moments(coll_f.data(), adve_f.data(), m0.data(), m1.data(), &p,
                                             adve_f.rows(), adve_f.cols(), nullptr));
    warm_ = true;
  }
  void advect() { solver::advect(adve_f, coll_f); }  // ulbm.cpp:322-379 == solver::advect
  // ulbm.cpp:248-263.  Before the first collide() the reference's ux2/uy2 members are still the
  // ctor's zeros (that is how the driver initialises, ulbm_double_shear_flow.cpp:96).
  void eval_equilibrium(lbm::Field& equi_f) {
    lbm::check(lbm_kbc_equilibrium(equi_f.data(), m0.data(), m1.data(), equi_f.rows(), equi_f.cols(),
                                   warm_ ? 0 : 1, nullptr));
  }
  // the driver's moment update (ulbm_double_shear_flow.cpp:141-142)
  void update_moments() {
    solver::calc_rho(m0, adve_f);
    solver::calc_u(m1, adve_f, m0);
  }

 private:
  const double s2;
  bool warm_ = false;
};
}  // namespace ulbm::d2q9

// ===============================================================================================
// class differential -- src/differential.hpp:6-53
// ===============================================================================================
class differential {
 public:
  lbm::Field x(const lbm::Field& psi) const { return run(psi, 0); }
  lbm::Field y(const lbm::Field& psi) const { return run(psi, 1); }
  void grad(lbm::Field& ans, const lbm::Field& psi) const {  // ans: [R,C,2]
    const size_t n = (size_t)psi.rows() * psi.cols();
    lbm::check(lbm_diff5(ans.data(), psi.data(), psi.rows(), psi.cols(), 0, nullptr));
    lbm::check(lbm_diff5(ans.data() + n, psi.data(), psi.rows(), psi.cols(), 1, nullptr));
  }

 private:
  static lbm::Field run(const lbm::Field& psi, int dir) {
    lbm::Field out(psi.rows(), psi.cols(), 1);
    lbm::check(lbm_diff5(out.data(), psi.data(), psi.rows(), psi.cols(), dir, nullptr));
    return out;
  }
};

// ===============================================================================================
// class colour -- src/colour.hpp:9-42: per-fluid parameters from a TOML table.  The per-node
// constant tensors of the reference (eta [R,C,9]) are kernel constants here.
// ===============================================================================================
class colour {
 public:
  const double rho_0, alpha, A, nu, mu, beta, cs2, ics2, rlx;
  std::array<double, 9> phi, eta;
  explicit colour(const lbm::toml::node_view& tbl)
      : rho_0{try_double(tbl, "initial_density")},
        alpha{try_double(tbl, "alpha")},
        A{try_double(tbl, "interfacial_tension_control")},
        nu{try_double(tbl, "kinematic_viscosity")},
        mu{nu * rho_0},
        beta{try_double(tbl, "interface_thickness_control")},
        cs2{3.0 * (1.0 - alpha) / 5.0},
        ics2{1.0 / cs2},
        rlx{1.0 / (0.5 + nu / cs2)} {
    const double a = 0.2 * (1.0 - alpha), b = 0.05 * (1.0 - alpha);
    phi = {alpha, a, a, a, a, b, b, b, b};
    for (int q = 0; q < 9; ++q) {
      const double e2 = solver::c[0][q] * solver::c[0][q] + solver::c[1][q] * solver::c[1][q];
      eta[q] = 1.0 + 0.5 * (3.0 * cs2 - 1.0) * (3.0 * e2 - 4.0);
    }
  }
  lbm_cg_colour abi() const { return lbm_cg_colour{rho_0, alpha, nu, beta}; }

 private:
  static double try_double(const lbm::toml::node_view& tbl, const std::string& name) {
    std::optional<double> op = tbl[name].value<double>();
    if (op.has_value()) return op.value();
    throw std::runtime_error(name + "not defined in parameters file");  // sic, colour.cpp:45
  }
};

namespace lbm {
// two-phase driver loop (test/mrtcg_rayleigh_taylor.cpp:413-478), wraps lbm_cg_solver
class CgSolver {
 public:
  // Fg = (gravity_r, gravity_c); add_source = false: Fg only shifts u (static-droplet driver)
  CgSolver(int R, int C, const colour& red, const colour& blue, double sigma, double gravity_r,
           double delta = 0.1, double gravity_c = 0.0, bool add_source = true)
      : R_(R), C_(C) {
    lbm_geom g{R, C, 0, 0, 0};
    lbm_cg_params p{red.abi(), blue.abi(), sigma, gravity_r, gravity_c, add_source ? 1 : 0, delta, LBM_FORM_DEFAULT};
    check(lbm_cg_solver_create(&h_, &g, nullptr, &p, nullptr));
  }
  CgSolver(const CgSolver&) = delete;
  ~CgSolver() {
    if (h_) lbm_cg_solver_destroy(h_);
  }
  void set_state(const std::vector<double>& f_r, const std::vector<double>& f_b,
                 const std::vector<double>& rho_r, const std::vector<double>& rho_b,
                 const std::vector<double>& u) {
    check(lbm_cg_solver_set_state(h_, f_r.data(), f_b.data(), rho_r.data(), rho_b.data(), u.data()));
  }
  void step(int n) { check(lbm_cg_solver_step(h_, n)); }
  struct State {
    std::vector<double> rho_r, rho_b, u, psi, s_nu;
  };
  State macroscopic() {
    const size_t n = (size_t)R_ * C_;
    State s{std::vector<double>(n), std::vector<double>(n), std::vector<double>(2 * n),
            std::vector<double>(n), std::vector<double>(n)};
    check(lbm_cg_solver_get_state(h_, nullptr, nullptr, s.rho_r.data(), s.rho_b.data(), s.u.data(),
                                  s.psi.data(), s.s_nu.data()));
    return s;
  }

 private:
  lbm_cg_solver* h_ = nullptr;
  int R_, C_;
};
}  // namespace lbm

// ===============================================================================================
// class ibm -- src/ibm.hpp:21-34: markers from the TOML arrays x, y of table `name`
// ===============================================================================================
class ibm {
 public:
  ibm(const lbm::toml::table& tbl, const std::string& name, int X, int Y, int m_max = 5) {
    auto x = tbl[name]["x"].as_array();
    auto y = tbl[name]["y"].as_array();
    if (!x || !y) throw std::runtime_error("Cannot parse x coordinate");  // ibm.cpp:90
    if (x->size() != y->size()) throw std::runtime_error("Cannot parse y coordinate");
    lbm::check(lbm_ibm_create(&h_, x->values().data(), y->values().data(), (int)x->size(), m_max, X, Y));
    lbm::check(lbm_ibm_roi(h_, &rows.first, &rows.second, &cols.first, &cols.second));
  }
  ibm(const ibm&) = delete;
  ~ibm() {
    if (h_) lbm_ibm_destroy(h_);
  }
  // eulerian_force_density (ibm.cpp:158-190): u [X,Y,2], rho [X,Y,1] -> F [ROI_r, ROI_c, 2]
  lbm::Field eulerian_force_density(const lbm::Field& u_0, const lbm::Field& rho_0) {
    lbm::Field F(rows.second - rows.first, cols.second - cols.first, 2);
    lbm::check(lbm_ibm_force(h_, u_0.data(), rho_0.data(), F.data(), nullptr));
    return F;
  }
  std::array<double, 2> surface_force() {
    std::array<double, 2> fs{};
    lbm::check(lbm_ibm_surface_force(h_, fs.data(), nullptr));
    return fs;
  }
  std::pair<int, int> rows, cols;  // region of interest [first, second)
  lbm_ibm* handle() { return h_; }

 private:
  lbm_ibm* h_ = nullptr;
};
