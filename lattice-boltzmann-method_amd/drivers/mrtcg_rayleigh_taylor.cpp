// Restatement of test/mrtcg_rayleigh_taylor.cpp (BASELINE config 4): colour-gradient MRT
// Rayleigh-Taylor instability.  Same TOML input (argv[1]): [general] sigma, gravity_magnitude,
// name (:360-362), [domain] rows, columns, time_steps, nr_snapshots (:103-117), [red]/[blue]
// (src/colour.cpp:11-20).  init_rho_cosine as :182-210; the ~400-launch loop body (:431-477) is
// two fused kernels per step.
//   usage: mrtcg_rayleigh_taylor params.toml [--steps N] [--dump prefix]
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

using std::cerr;
using std::cout;
using std::endl;

template <typename T>
T try_value(const lbm::toml::node_view& tbl, const std::string name) {  // :26-32
  std::optional<T> op = tbl[name].template value<T>();
  if (op.has_value()) return op.value();
  else throw std::runtime_error(name + "not defined in parameters file");
}

static std::vector<double> init_rho_cosine(int R, int C, double rho_0, bool invert) {  // :182-210
  std::vector<double> rho((size_t)R * C);
  const double middle = R / 2.0;
  for (int r = 0; r < R; r++)
    for (int c = 0; c < C; c++) {
      double s = middle - 0.1 * C * std::cos(2.0 * 3.141592 * c / C);
      double ans = 0.0;
      if (invert) {
        if (r < s) ans = 1.0;  // red fluid
      } else {
        if (r >= s) ans = 1.0;  // blue fluid
      }
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
    const double sigma = try_value<double>(tbl["general"], "sigma");
    const double gravity_magnitude = try_value<double>(tbl["general"], "gravity_magnitude");
    const std::string name = try_value<std::string>(tbl["general"], "name");
    const int R = try_value<int>(tbl["domain"], "rows");
    const int C = try_value<int>(tbl["domain"], "columns");
    int T = try_value<int>(tbl["domain"], "time_steps");
    const int nr_snapshots = try_value<int>(tbl["domain"], "nr_snapshots");
    T = std::stoi(arg_value(argc, argv, "--steps", std::to_string(T)));
    const std::string dump = arg_value(argc, argv, "--dump", "");
    cout << "DOMAIN parameters:\nR=" << R << "\nC=" << C << "\nT=" << T << "\nnr_snapshots=" << nr_snapshots << endl;
    colour r{tbl["red"]};
    colour b{tbl["blue"]};
    if (lbm_device_count() < 1) {
      cerr << "no HIP device available\n";
      return 2;
    }
    // Init. densities (:372-373) and populations (:407-410, u = 0)
    auto rho_r = init_rho_cosine(R, C, r.rho_0, true);
    auto rho_b = init_rho_cosine(R, C, b.rho_0, false);
    std::vector<double> u((size_t)R * C * 2, 0.0);
    lbm::Field d_rr(R, C, 1), d_rb(R, C, 1), d_u(R, C, 2), f_r(R, C, 9), f_b(R, C, 9);
    d_rr.from_host(rho_r);
    d_rb.from_host(rho_b);
    const lbm_cg_colour cr = r.abi(), cb = b.abi();
    lbm::check(lbm_cg_equilibrium(f_r.data(), d_rr.data(), d_u.data(), &cr, R, C, 0, nullptr));
    lbm::check(lbm_cg_equilibrium(f_b.data(), d_rb.data(), d_u.data(), &cb, R, C, 0, nullptr));

    lbm::CgSolver sv(R, C, r, b, sigma, gravity_magnitude, 0.1);  // delta hard-coded at :375
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
    cout << "name=" << name << "\nsteps=" << T << "\nmass_red=" << mr << "\nmass_blue=" << mb << endl;
    if (!dump.empty()) {
      dump_f64(dump + "-rho_r.f64", s.rho_r);
      dump_f64(dump + "-rho_b.f64", s.rho_b);
      dump_f64(dump + "-u.f64", s.u);
      dump_f64(dump + "-phase.f64", s.psi);
      dump_f64(dump + "-snu.f64", s.s_nu);
    }
  } catch (const std::exception& e) {
    cerr << "error: " << e.what() << endl;
    return 3;
  }
  return 0;
}
