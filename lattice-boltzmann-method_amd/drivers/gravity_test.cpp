// Restatement of test/gravity_test.cpp: incompressible BGK channel driven by a
// body force Fg = (-0.0003, 0): u += Fg (:146), delta-form relaxation plus the Guo-type source with
// 1/3, 1/9 (:151-160), pressure-periodic rows with rho_in = rho_out (:164), halfway bounce-back
// columns (:171-177), the reference's convergence rule (:126-139).
//   usage: gravity_test [--H 21] [--W 21] [--T 10000] [--dump prefix]
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

int main(int argc, char** argv) {
  const int T = std::stoi(arg_value(argc, argv, "--T", "10000"));
  const int H = std::stoi(arg_value(argc, argv, "--H", "21"));
  const int W = std::stoi(arg_value(argc, argv, "--W", "21"));
  const std::string dump = arg_value(argc, argv, "--dump", "");
  const double tau = std::sqrt(3.0 / 16.0) + 0.5, omega = 1.0 / tau;
  const double rho_outlet = 1.0, rho_inlet = rho_outlet;  // :74-75
  std::cout << "T=" << T << "\nH=" << H << "; W=" << W << "\nomega=" << omega << "\nFg=(-0.0003, 0)" << std::endl;
  if (lbm_device_count() < 1) {
    std::cerr << "no HIP device available\n";
    return 2;
  }
  try {
    lbm::Field f_adve(H, W, 9), u(H, W, 2), rho(H, W, 1);
    rho.fill(1.0);
    solver::incomp_equilibrium(f_adve, u, rho);  // :103
    lbm::BoundarySet bc;
    bc.col_lo = bc.col_hi = LBM_EDGE_BOUNCE_BACK;
    bc.pressure_rows = 1;
    bc.rho_inlet = rho_inlet;
    bc.rho_outlet = rho_outlet;
    const lbm_bgk_params prm{omega, /*incompressible=*/1, /*delta_form=*/1, /*force_mode=*/1,
                             -0.0003, 0.0, 1.0 / 3.0, 1.0 / 9.0, LBM_FORM_DEFAULT};  // Fg :87, ics2/ics4 :81-82
    lbm::Solver sv = lbm::Solver::bgk(H, W, prm, bc);
    sv.set_f(f_adve);
    std::vector<double> uh((size_t)H * W * 2, 0.0);
    double old_mean = 1.0;
    int t = 0;
    while (t < T) {
      if (t % 100 == 1) {  // :126-139
        double mean = 0.0;
        for (size_t i = 0; i < (size_t)H * W; ++i) mean += uh[2 * i];
        mean /= (double)H * W;
        if (std::fabs(mean / old_mean - 1.0) < 1e-12) {
          std::cout << "last t=" << t << std::endl;
          break;
        }
        old_mean = mean;
      }
      const int next = (t == 0) ? 1 : t + 100;
      sv.step(std::min(next, T) - t, true);
      t = std::min(next, T);
      uh = sv.moments().second;
    }
    std::cout.precision(17);
    std::cout << "steps=" << t << "\nux_centre=" << uh[2 * ((size_t)(H / 2) * W + W / 2)] << std::endl;
    dump_f64(dump.empty() ? "" : dump + "-u.f64", uh);
    dump_f64(dump.empty() ? "" : dump + "-f.f64", sv.get_f());
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
