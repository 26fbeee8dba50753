// Restatement of test/ulbm_poiseuille.cpp: Poiseuille flow with the entropic
// KBC collision -- pressure-periodic rows 0 / H-1 (:36-58; imposed density through
// solver::incomp_equilibrium, f_equi = kbc.iequi_f.pow(-1)), halfway bounce-back on the first and
// last column (:126-132), started from adve_f = 0 with the held moments m0 = 1, m1 = 0 (:85-86).
//   usage: ulbm_poiseuille [--T 300000] [--snapshot 100] [--H 128] [--W 128] [--dump prefix]
// Dumps (raw f64): <prefix>-u.f64 [H][W][2], <prefix>-rho.f64 [H][W] = kbc.m1 / kbc.m0 after T
// iterations (:136-139).
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

using std::cout;
using std::endl;

int main(int argc, char** argv) {
  const int T = std::stoi(arg_value(argc, argv, "--T", "300000"));
  const int snapshot_period = std::stoi(arg_value(argc, argv, "--snapshot", "100"));
  const int H = std::stoi(arg_value(argc, argv, "--H", "128"));
  const int W = std::stoi(arg_value(argc, argv, "--W", "128"));
  const std::string dump = arg_value(argc, argv, "--dump", "");
  const double nu = 1E-4;                          // :70
  const double omega = 1.0 / (0.5 + 3.0 * nu);     // :71
  const double u_max = 0.05;                       // :76
  const double p_grad = 8.0 * nu * u_max / (W * W);
  const double rho_outlet = 1.0;
  const double rho_inlet = 3.0 * (H - 1) * p_grad + rho_outlet;  // :80-83
  cout << "T=" << T << "\nH=" << H << "; W=" << W << "\nnu=" << nu << "\nomega=" << omega
       << "\ntau=" << 1.0 / omega << "\nu_max=" << u_max << "\nRe=" << W * u_max / nu
       << "\ngrad(p)=" << p_grad << "\nrho_inlet=" << rho_inlet << endl;
  if (lbm_device_count() < 1) {
    std::cerr << "no HIP device available\n";
    return 2;
  }
  try {
    lbm::BoundarySet bc;
    bc.col_lo = bc.col_hi = LBM_EDGE_BOUNCE_BACK;
    bc.pressure_rows = 1;
    bc.rho_inlet = rho_inlet;
    bc.rho_outlet = rho_outlet;
    lbm::Solver sv = lbm::Solver::kbc(H, W, omega, bc);
    const size_t n = (size_t)H * W;
    sv.set_f(std::vector<double>(n * 9, 0.0));                             // ctor zeros
    sv.set_moments(std::vector<double>(n, 1.0), std::vector<double>(n * 2, 0.0));  // :86
    cout << "main loop starts" << endl;
    for (int t = 0; t < T; t += snapshot_period) sv.step(std::min(snapshot_period, T - t));
    // m0, m1 as the driver holds them after T iterations
    ulbm::d2q9::kbc kbc{H, W, omega};
    kbc.adve_f.from_host(sv.get_f());
    kbc.update_moments();
    auto rho = kbc.m0.to_host();
    auto u = kbc.m1.to_host();
    double mass = 0.0, u_mid = 0.0;
    for (size_t i = 0; i < rho.size(); ++i) mass += rho[i];
    for (int r = 0; r < H; ++r) u_mid += u[2 * ((size_t)r * W + W / 2)] / H;  // mean u_x on the centre line
    cout.precision(17);
    cout << "steps=" << T << "\nmass=" << mass << "\nu_centre=" << u_mid << endl;
    dump_f64(dump.empty() ? "" : dump + "-u.f64", u);
    dump_f64(dump.empty() ? "" : dump + "-rho.f64", rho);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << endl;
    return 3;
  }
  return 0;
}
