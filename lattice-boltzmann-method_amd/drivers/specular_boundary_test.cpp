// Restatement of test/specular_boundary_test.cpp: compressible BGK channel,
// pressure-periodic inlet/outlet rows (compressible equilibrium, :23-45), specular side walls
// (:121-127).  The reference only saves snapshots; this driver prints the centre-row profile.
//   usage: specular_boundary_test [--H 51] [--W 51] [--T 10000] [--dump prefix]
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

int main(int argc, char** argv) {
  const int T = std::stoi(arg_value(argc, argv, "--T", "10000"));
  const int H = std::stoi(arg_value(argc, argv, "--H", "51"));
  const int W = std::stoi(arg_value(argc, argv, "--W", "51"));
  const std::string dump = arg_value(argc, argv, "--dump", "");
  const double tau = std::sqrt(3.0 / 16.0) + 0.5, omega = 1.0 / tau, u_max = 0.1;
  const double nu = (2.0 * tau - 1.0) / 6.0, p_grad = 8.0 * nu * u_max / (W * W);
  const double rho_outlet = 1.0, rho_inlet = 3.0 * (H - 1) * p_grad + rho_outlet;
  std::cout << "T=" << T << "\nH=" << H << "; W=" << W << "\nomega=" << omega << "\nnu=" << nu
            << "\nRe=" << W * u_max / nu << "\ngrad(p)=" << p_grad << "\nrho_inlet=" << rho_inlet << std::endl;
  if (lbm_device_count() < 1) {
    std::cerr << "no HIP device available\n";
    return 2;
  }
  try {
    lbm::Field f_adve(H, W, 9), u(H, W, 2), rho(H, W, 1);
    rho.fill(1.0);
    solver::incomp_equilibrium(f_adve, u, rho);  // :87
    lbm::BoundarySet bc;
    bc.col_lo = bc.col_hi = LBM_EDGE_SPECULAR;
    bc.pressure_rows = 1;
    bc.rho_inlet = rho_inlet;
    bc.rho_outlet = rho_outlet;
    lbm::Solver sv = lbm::Solver::bgk(H, W, omega, /*incompressible=*/false, bc);
    sv.set_f(f_adve);
    sv.step(T, true);
    auto m = sv.moments();
    std::cout.precision(17);
    std::cout << "steps=" << T << "\nux_centre=" << m.second[2 * ((size_t)(H / 2) * W + W / 2)] << std::endl;
    dump_f64(dump.empty() ? "" : dump + "-u.f64", m.second);
    dump_f64(dump.empty() ? "" : dump + "-rho.f64", m.first);
    dump_f64(dump.empty() ? "" : dump + "-f.f64", sv.get_f());
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
