// Restatement of test/decompose_domain.cpp (the halo contract's specification): the
// Poiseuille channel of horizontal_poiseuille_test cut into two blocks A (upstream) and B, glued by
// (1) the pressure-periodic virtual rows taken ACROSS the blocks (:50-73, compressible equilibrium),
// (2) three populations per interface row copied each way after advect (:181-187).
// Operator level on purpose -- this driver exists to show the binding semantics the slab ring
// generalises: per block calc_rho, calc_u, equilibrium, collision, advect (solver:: facade), the two
// cross-block rows (lbm_pressure_row), and ONE gather launch (lbm_links_*) for the bounce-back
// columns of both blocks and the six binding slices.
//   usage: decompose_domain [--T 500] [--H 21] [--W 21] [--dump prefix]
// Dumps (raw f64) per block X in A, B: <prefix>-X-f.f64 [H][W][9] = adve_f, -X-rho.f64, -X-u.f64 = m_0, m_1
// as held at the top of iteration T (what the reference's snapshot T stores).
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

int main(int argc, char** argv) {
  const int T = std::stoi(arg_value(argc, argv, "--T", "500"));
  const int H = std::stoi(arg_value(argc, argv, "--H", "21"));
  const int W = std::stoi(arg_value(argc, argv, "--W", "21"));
  const std::string dump = arg_value(argc, argv, "--dump", "");
  const double tau = std::sqrt(3.0 / 16.0) + 0.5, omega = 1.0 / tau;   // :79-80
  const double u_max = 1.030985714E-1, nu = (2.0 * tau - 1.0) / 6.0;   // :86-87
  const double p_grad = 8.0 * nu * u_max / (W * W);
  const double rho_outlet = 1.0, rho_inlet = 3.0 * (H - 1) * p_grad + rho_outlet;   // :93-95
  std::cout.precision(17);
  std::cout << "T=" << T << "\nH=" << H << "; W=" << W << "\nomega=" << omega << "\nrho_inlet=" << rho_inlet << std::endl;
  if (lbm_device_count() < 1) {
    std::cerr << "no HIP device available\n";
    return 2;
  }
  try {
    struct Block {
      lbm::Field adve, coll, equi, u, rho;
      Block(int H, int W) : adve(H, W, 9), coll(H, W, 9), equi(H, W, 9), u(H, W, 2), rho(H, W, 1) {}
    };
    Block A(H, W), B(H, W);
    const lbm_geom g{H, W, 0, 0, 0};
    for (Block* b : {&A, &B}) {  // m_0 = 1, m_1 = 0, adve_f = equilibrium (:105-123)
      b->rho.fill(1.0);
      b->u.fill(0.0);
      solver::equilibrium(b->adve, b->u, b->rho);
    }
    // walls + bindings as one link table: lattice 0 = A, 1 = B; dst = adve_f, src = coll_f
    lbm_links* links = nullptr;
    const lbm_geom geoms[2] = {g, g};
    lbm::check(lbm_links_create(&links, 2, geoms));
    for (int k = 0; k < 2; ++k) {  // bounce-back columns of each block (:163-178)
      const int pairs[3][2] = {{4, 2}, {7, 5}, {8, 6}};
      for (auto& p : pairs) lbm::check(lbm_links_add(links, k, p[0], 0, W - 1, 1, 0, k, p[1], 0, W - 1, 1, 0, H));
      const int pairs0[3][2] = {{2, 4}, {5, 7}, {6, 8}};
      for (auto& p : pairs0) lbm::check(lbm_links_add(links, k, p[0], 0, 0, 1, 0, k, p[1], 0, 0, 1, 0, H));
    }
    // A's last row <- B's first row (populations moving towards -r), B's first row <- A's last row (:181-187)
    lbm::check(lbm_links_add(links, 0, 3, H - 1, 0, 0, 1, 1, 3, 0, 0, 0, 1, W));
    lbm::check(lbm_links_add(links, 0, 6, H - 1, 1, 0, 1, 1, 6, 0, 0, 0, 1, W - 1));
    lbm::check(lbm_links_add(links, 0, 7, H - 1, 0, 0, 1, 1, 7, 0, 1, 0, 1, W - 1));
    lbm::check(lbm_links_add(links, 1, 1, 0, 0, 0, 1, 0, 1, H - 1, 0, 0, 1, W));
    lbm::check(lbm_links_add(links, 1, 5, 0, 1, 0, 1, 0, 5, H - 1, 0, 0, 1, W - 1));
    lbm::check(lbm_links_add(links, 1, 8, 0, 0, 0, 1, 0, 8, H - 1, 1, 0, 1, W - 1));
    lbm::check(lbm_links_finalize(links));
    double* adve[2] = {A.adve.data(), B.adve.data()};
    const double* coll[2] = {A.coll.data(), B.coll.data()};
    for (int t = 0; t < T; ++t) {
      for (Block* b : {&A, &B}) {  // :141-152
        solver::calc_rho(b->rho, b->adve);
        solver::calc_u(b->u, b->adve, b->rho);
        solver::equilibrium(b->equi, b->u, b->rho);
        solver::collision(b->coll, b->adve, b->equi, omega);
      }
      // periodic_boundary_condition(A, B, ...) :50-73: inlet of A from B's outlet row, outlet of B from A's inlet row
      lbm::check(lbm_pressure_row(A.coll.data(), &g, 0, B.coll.data(), B.equi.data(), B.u.data(), &g, H - 2, rho_inlet, 0, nullptr));
      lbm::check(lbm_pressure_row(B.coll.data(), &g, H - 1, A.coll.data(), A.equi.data(), A.u.data(), &g, 1, rho_outlet, 0, nullptr));
      solver::advect(A.adve, A.coll);  // :159-160
      solver::advect(B.adve, B.coll);
      lbm::check(lbm_links_apply(links, adve, coll, nullptr));
    }
    lbm_links_destroy(links);
    double mass = 0.0;
    const char* names = "AB";
    int k = 0;
    for (Block* b : {&A, &B}) {
      const auto rh = b->rho.to_host();
      for (double v : rh) mass += v;
      if (!dump.empty()) {
        const std::string pre = dump + "-" + names[k];
        dump_f64(pre + "-f.f64", b->adve.to_host());
        dump_f64(pre + "-rho.f64", rh);
        dump_f64(pre + "-u.f64", b->u.to_host());
      }
      ++k;
    }
    std::cout << "steps=" << T << "\nmass=" << mass << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
