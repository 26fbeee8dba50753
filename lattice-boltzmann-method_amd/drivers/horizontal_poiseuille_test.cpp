// Restatement of test/horizontal_poiseuille_test.cpp on the MI355X engine (BASELINE config 1).
// Same flow parameters (:50-67), same convergence rule (:113-126), same L2 check (:163-175);
// the hand-written loop body (:130-152: calc_rho, calc_incomp_u, incomp_equilibrium, collision,
// pressure-periodic rows, advect, halfway bounce-back columns) is ONE fused launch per step.
//   usage: horizontal_poiseuille_test [--H 21] [--W 21] [--T 8301] [--dump prefix]
#include <cassert>
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

using std::cout;
using std::endl;

int main(int argc, char** argv) {
  const int T = std::stoi(arg_value(argc, argv, "--T", "8301"));
  const int H = std::stoi(arg_value(argc, argv, "--H", "21"));
  const int W = std::stoi(arg_value(argc, argv, "--W", "21"));
  const std::string dump = arg_value(argc, argv, "--dump", "");
  cout << "T=" << T << "\nH=" << H << "; W=" << W << endl;
  const double tau = std::sqrt(3.0 / 16.0) + 0.5;
  const double omega = 1.0 / tau;
  cout << "omega=" << omega << endl;
  const double u_max = 1.030985714E-1;
  const double nu = (2.0 * tau - 1.0) / 6.0;
  cout << "nu=" << nu << endl;
  cout << "Re=" << W * u_max / nu << endl;
  const double p_grad = 8.0 * nu * u_max / (W * W);
  cout << "grad(p)=" << p_grad << endl;
  const double rho_outlet = 1.0;
  const double rho_inlet = 3.0 * (H - 1) * p_grad + rho_outlet;
  cout << "rho_inlet=" << rho_inlet << endl;
  if (lbm_device_count() < 1) {
    std::cerr << "no HIP device available\n";
    return 2;
  }
  try {
    // Tensors (:78-82) and initialisation (:91)
    lbm::Field f_adve(H, W, 9), u(H, W, 2), rho(H, W, 1);
    rho.fill(1.0);
    solver::incomp_equilibrium(f_adve, u, rho);

    lbm::BoundarySet bc;                                  // :140 and :146-152
    bc.col_lo = bc.col_hi = LBM_EDGE_BOUNCE_BACK;
    bc.pressure_rows = 1;
    bc.rho_inlet = rho_inlet;
    bc.rho_outlet = rho_outlet;
    lbm::Solver sv = lbm::Solver::bgk(H, W, omega, /*incompressible=*/true, bc);
    sv.set_f(f_adve);

    const int t_interval = 100;
    const double tolerance = 1e-12;
    double old_mean = 1.0;  // old_u = ones_like(rho)
    std::vector<double> uh((size_t)H * W * 2, 0.0), rhoh((size_t)H * W, 1.0);
    int t = 0;
    auto advance = [&](int n) {
      if (n <= 0) return;
      sv.step(n, true);
      auto m = sv.moments();
      rhoh = std::move(m.first);
      uh = std::move(m.second);
      t += n;
    };
    cout << "main loop starts" << endl;
    while (t < T) {
      if (t % t_interval == 1) {  // :113-126, u = the moments of iteration t-1
        double mean = 0.0;
        for (size_t i = 0; i < (size_t)H * W; ++i) mean += uh[2 * i];
        mean /= (double)H * W;
        if (std::fabs(mean / old_mean - 1.0) < tolerance) {
          cout << "last t=" << t << endl;
          break;
        }
        old_mean = mean;
      }
      const int next = (t == 0) ? 1 : t + t_interval;  // t is 0 or == 1 (mod t_interval)
      advance(std::min(next, T) - t);
    }
    // Evaluate final velocity profile (:163-175)
    std::vector<double> ua(W);
    double den = 0.0;
    for (int c = 0; c < W; ++c) {
      const double y = (1.0 + c) - 0.5;
      ua[c] = -4.0 * u_max / ((double)W * W) * y * (y - W);
      den += ua[c] * ua[c];
    }
    den = 1.0 / std::sqrt(den);
    double sum = 0.0;
    for (int r = 1; r < H - 1; ++r) {
      double e = 0.0;
      for (int c = 0; c < W; ++c) {
        const double d = uh[2 * ((size_t)r * W + c)] - ua[c];
        e += d * d;
      }
      sum += std::sqrt(e) * den;
    }
    const double l2 = (1.0 / H) * sum;
    cout.precision(5);
    cout << "steps=" << t << "\nL2=" << l2 << endl;
    dump_f64(dump.empty() ? "" : dump + "-u.f64", uh);
    dump_f64(dump.empty() ? "" : dump + "-rho.f64", rhoh);
    dump_f64(dump.empty() ? "" : dump + "-f.f64", sv.get_f());
    if (!(l2 <= 1e-11)) {
      std::cerr << "Large L2 error" << endl;  // the reference's assert (:172)
      return 1;
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << endl;
    return 3;
  }
  return 0;
}
