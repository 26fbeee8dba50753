// Restatement of test/free_stream_test.cpp: incompressible BGK, uniform stream
// u = (0.1, 0), anti-bounce-back inlet/outlet rows with u_w = (0.1, 0) (:102-124), specular side
// walls (:126-133); lattice size and omega from params.toml (argv[1]) like the reference.
//   usage: free_stream_test params.toml [--steps N] [--dump prefix]
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "../include/lbm/params.hpp"
#include "common.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage: " << argv[0] << " params.toml [--steps N] [--dump prefix]\n";
    return 1;
  }
  lbm::toml::table tbl;
  try {
    tbl = lbm::toml::parse_file(argv[1]);
  } catch (const lbm::toml::parse_error& err) {
    std::cerr << "Parsing failed:\n" << err.what() << "\n";
    return 1;
  }
  try {
    const params::flow fp{tbl};
    const params::lattice lp{tbl, fp};
    const params::simulation sp{tbl, lp};
    std::cout << fp << "\n" << lp << "\n" << sp << "\n";
    const int steps = std::stoi(arg_value(argc, argv, "--steps", std::to_string(sp.total_steps)));
    const std::string dump = arg_value(argc, argv, "--dump", "");
    if (lbm_device_count() < 1) {
      std::cerr << "no HIP device available\n";
      return 2;
    }
    lbm::Field f_adve(lp.X, lp.Y, 9), u(lp.X, lp.Y, 2), rho(lp.X, lp.Y, 1);
    rho.fill(1.0);
    {
      std::vector<double> uh((size_t)lp.X * lp.Y * 2, 0.0);
      for (size_t i = 0; i < (size_t)lp.X * lp.Y; ++i) uh[2 * i] = 0.1;  // :52
      u.from_host(uh);
    }
    solver::incomp_equilibrium(f_adve, u, rho);  // :76
    lbm::BoundarySet bc;
    bc.row_lo = bc.row_hi = LBM_EDGE_ABB_VELOCITY;
    bc.col_lo = bc.col_hi = LBM_EDGE_SPECULAR;
    bc.uw_r = 0.1;  // u_w, :67
    lbm::Solver sv = lbm::Solver::bgk(lp.X, lp.Y, lp.omega, /*incompressible=*/true, bc);
    sv.set_f(f_adve);
    sv.step(steps, true);
    auto m = sv.moments();
    double mean = 0.0;
    for (size_t i = 0; i < m.first.size(); ++i) mean += m.second[2 * i];
    std::cout.precision(17);
    std::cout << "steps=" << steps << "\nmean_ux=" << mean / (double)m.first.size() << std::endl;
    dump_f64(dump.empty() ? "" : dump + "-u.f64", m.second);
    dump_f64(dump.empty() ? "" : dump + "-f.f64", sv.get_f());
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
