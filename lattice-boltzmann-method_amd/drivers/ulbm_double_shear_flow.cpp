// Restatement of test/ulbm_double_shear_flow.cpp (BASELINE config 3: KBC doubly periodic shear
// layer).  Initial conditions exactly as :42-63, kbc.eval_equilibrium as the driver
// calls it (:96), then collide + advect + moment update fused into one launch per step.
//   usage: ulbm_double_shear_flow [--H 128] [--W 128] [--T 10000] [--snapshot 10] [--dump prefix]
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

using std::cout;
using std::endl;

static void set_initial_conditions(std::vector<double>& rho, std::vector<double>& u, int R, int C,
                                   const double u_max, const double alpha = 80.0,
                                   const double delta = 0.05) {
  for (int r = 0; r < R; r++)
    for (int c = 0; c < C; c++) {
      const size_t i = (size_t)r * C + c;
      u[2 * i] = u_max * std::tanh(alpha * (0.25 * R - std::abs(c - 0.5 * R)));
      rho[i] = 1.0;
      u[2 * i + 1] = u_max * delta * std::sin(6.2832 * (r + 0.25 * R) / R);
    }
}

int main(int argc, char** argv) {
  const int T = std::stoi(arg_value(argc, argv, "--T", "10000"));
  const int snapshot_period = std::stoi(arg_value(argc, argv, "--snapshot", "10"));
  const int H = std::stoi(arg_value(argc, argv, "--H", "128"));
  const int W = std::stoi(arg_value(argc, argv, "--W", "128"));
  const std::string dump = arg_value(argc, argv, "--dump", "");
  const double nu = 1.70766666E-4;
  const double omega = 1.0 / (0.5 + 3.0 * nu);
  const double u_max = 0.02;
  cout << "T=" << T << "\nH=" << H << "; W=" << W << "\nnu=" << nu << "\nomega=" << omega
       << "\ntau=" << 1.0 / omega << "\nu_max=" << u_max << "\nRe=" << W * u_max / nu << endl;
  if (lbm_device_count() < 1) {
    std::cerr << "no HIP device available\n";
    return 2;
  }
  try {
    ulbm::d2q9::kbc kbc{H, W, omega};
    std::vector<double> m0((size_t)H * W), m1((size_t)H * W * 2);
    set_initial_conditions(m0, m1, H, W, u_max);
    kbc.m0.from_host(m0);
    kbc.m1.from_host(m1);
    kbc.eval_equilibrium(kbc.adve_f);  // :96

    lbm::Solver sv = lbm::Solver::kbc(H, W, omega);
    sv.set_f(kbc.adve_f);
    cout << "main loop starts" << endl;
    int snaps = 0;
    for (int t = 0; t < T; t += snapshot_period) {
      const int n = std::min(snapshot_period, T - t);
      sv.step(n);
      ++snaps;
    }
    // m0, m1 as the driver leaves them after T iterations (:141-142)
    kbc.adve_f.from_host(sv.get_f());
    kbc.update_moments();
    auto rho = kbc.m0.to_host();
    auto u = kbc.m1.to_host();
    double mass = 0.0, ke = 0.0;
    for (size_t i = 0; i < rho.size(); ++i) {
      mass += rho[i];
      ke += 0.5 * rho[i] * (u[2 * i] * u[2 * i] + u[2 * i + 1] * u[2 * i + 1]);
    }
    cout.precision(17);
    cout << "steps=" << T << "\nmass=" << mass << "\nkinetic_energy=" << ke << endl;
    dump_f64(dump.empty() ? "" : dump + "-u.f64", u);
    dump_f64(dump.empty() ? "" : dump + "-rho.f64", rho);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << endl;
    return 3;
  }
  return 0;
}
