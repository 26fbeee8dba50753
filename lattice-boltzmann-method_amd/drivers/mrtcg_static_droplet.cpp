// Restatement of test/mrtcg_static_droplet.cpp: the colour-gradient operators of
// the Rayleigh-Taylor driver with a droplet initial state (:182-204), sigma = 0.1 hard-coded (:439),
// Fg = (0, -6.25e-6) acting only as the velocity shift u += Fg/(2 rho) (:452, :457, :527) -- the
// source term is commented out there (:513-514).  Reads [domain], [red], [blue] from the TOML file
// (the reference's mrtcg-rayleigh-taylor-gamma3.toml is directly usable).
//   usage: mrtcg_static_droplet params.toml [--steps N] [--dump prefix]
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

using std::cerr;
using std::cout;
using std::endl;

static double sigmoid(double x) { return 1.0 / (1.0 + std::exp(-x)); }

static std::vector<double> init_rho_droplet(int R, int C, double rho_0, bool invert) {  // :182-204
  std::vector<double> rho((size_t)R * C);
  const double center = R / 2.0, radius = 25.0;
  for (int r = 0; r < R; r++)
    for (int c = 0; c < C; c++) {
      double s = std::sqrt((r - center) * (r - center) + (c - center) * (c - center));
      double ans = 0.0;
      if (invert) ans = 1.0 - sigmoid(1.0 * (s - radius));
      else ans = sigmoid(1.0 * (s - radius));
      rho[(size_t)r * C + c] = rho_0 * ans;
    }
  return rho;
}

int main(int argc, char* argv[]) {
  if (argc < 2) {
    cerr << "usage: " << argv[0] << " params.toml [--steps N] [--dump prefix]\n";
    return 1;
  }
  lbm::toml::table tbl;
  try {
    tbl = lbm::toml::parse_file(argv[1]);
  } catch (const lbm::toml::parse_error& err) {
    cerr << "Parsing failed:\n" << err.what() << "\n";
    return 1;
  }
  try {
    auto need_int = [&](const char* key) {
      auto v = tbl["domain"][key].value<int>();
      if (!v) throw std::runtime_error(std::string(key) + "not defined in parameters file");
      return *v;
    };
    const int R = need_int("rows"), C = need_int("columns");
    int T = need_int("time_steps");
    const int nr_snapshots = need_int("nr_snapshots");
    T = std::stoi(arg_value(argc, argv, "--steps", std::to_string(T)));
    const std::string dump = arg_value(argc, argv, "--dump", "");
    colour r{tbl["red"]}, b{tbl["blue"]};
    const double sigma = 0.1, g_c = -6.25e-6;  // :439, :452
    if (lbm_device_count() < 1) {
      cerr << "no HIP device available\n";
      return 2;
    }
    auto rho_r = init_rho_droplet(R, C, r.rho_0, true);   // :420-421
    auto rho_b = init_rho_droplet(R, C, b.rho_0, false);
    std::vector<double> u((size_t)R * C * 2);
    for (size_t i = 0; i < (size_t)R * C; ++i) {          // :456-457
      const double rho = rho_r[i] + rho_b[i];
      u[2 * i] = 0.0 + 0.5 * 0.0 / rho;
      u[2 * i + 1] = 0.0 + 0.5 * g_c / rho;
    }
    lbm::Field d_rr(R, C, 1), d_rb(R, C, 1), d_u(R, C, 2), f_r(R, C, 9), f_b(R, C, 9);
    d_rr.from_host(rho_r);
    d_rb.from_host(rho_b);
    d_u.from_host(u);
    const lbm_cg_colour cr = r.abi(), cb = b.abi();
    lbm::check(lbm_cg_equilibrium(f_r.data(), d_rr.data(), d_u.data(), &cr, R, C, 0, nullptr));  // :458-459
    lbm::check(lbm_cg_equilibrium(f_b.data(), d_rb.data(), d_u.data(), &cb, R, C, 0, nullptr));
    lbm::CgSolver sv(R, C, r, b, sigma, /*gravity_r=*/0.0, 0.1, /*gravity_c=*/g_c, /*add_source=*/false);
    sv.set_state(f_r.to_host(), f_b.to_host(), rho_r, rho_b, u);
    cout << "main loop" << endl;
    const int period = std::max(1, T / std::max(1, nr_snapshots));
    for (int t = 0; t < T; t += period) sv.step(std::min(period, T - t));
    auto s = sv.macroscopic();
    double mr = 0.0, mb = 0.0;
    for (size_t i = 0; i < s.rho_r.size(); ++i) {
      mr += s.rho_r[i];
      mb += s.rho_b[i];
    }
    cout.precision(17);
    cout << "steps=" << T << "\nmass_red=" << mr << "\nmass_blue=" << mb << endl;
    if (!dump.empty()) {
      dump_f64(dump + "-rho_r.f64", s.rho_r);
      dump_f64(dump + "-rho_b.f64", s.rho_b);
      dump_f64(dump + "-u.f64", s.u);
      dump_f64(dump + "-phase.f64", s.psi);
    }
  } catch (const std::exception& e) {
    cerr << "error: " << e.what() << endl;
    return 3;
  }
  return 0;
}
