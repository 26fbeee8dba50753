// Restatement of test/cylinder_test.cpp (BASELINE config 5): flow past an immersed-boundary
// cylinder.  argv[1]: [flow] [lattice] [simulation] tables (src/params.cpp); argv[2]: boundary
// TOML with table "cylinder-a" holding the marker arrays x, y (src/ibm.cpp:78-102).  The
// interactive prompt of the reference (:79-82) is dropped.
//   usage: cylinder_test params.toml boundary.toml [--steps N] [--dump prefix]
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "../include/lbm/params.hpp"
#include "common.hpp"

using std::cerr;
using std::cout;

int main(int argc, char* argv[]) {
  if (argc < 3) {
    cerr << "usage: " << argv[0] << " params.toml boundary.toml [--steps N] [--dump prefix]\n";
    return 1;
  }
  lbm::toml::table tbl, tbl_boundary;
  try {
    tbl = lbm::toml::parse_file(argv[1]);
    tbl_boundary = lbm::toml::parse_file(argv[2]);
  } catch (const lbm::toml::parse_error& err) {
    cerr << "Parsing failed:\n" << err.what() << "\n";
    return 1;
  }
  try {
    const params::flow fp{tbl};
    cout << fp << "\n";
    const params::lattice lp{tbl, fp};
    cout << lp << "\n";
    const params::simulation sp{tbl, lp};
    cout << sp << "\n";
    const int steps = std::stoi(arg_value(argc, argv, "--steps", std::to_string(sp.total_steps)));
    const std::string dump = arg_value(argc, argv, "--dump", "");
    if (lbm_device_count() < 1) {
      cerr << "no HIP device available\n";
      return 2;
    }
    // Tensors (:49-53), u[...,0] = lp.u (:75), f_adve = incomp_equilibrium(u, rho) (:85)
    lbm::Field f_adve(lp.X, lp.Y, 9), u(lp.X, lp.Y, 2), rho(lp.X, lp.Y, 1);
    rho.fill(1.0);
    {
      std::vector<double> uh((size_t)lp.X * lp.Y * 2, 0.0);
      for (size_t i = 0; i < (size_t)lp.X * lp.Y; ++i) uh[2 * i] = lp.u;
      u.from_host(uh);
    }
    solver::incomp_equilibrium(f_adve, u, rho);

    ibm ib{tbl_boundary, "cylinder-a", lp.X, lp.Y};  // :62
    cout << "Immersed boundary data:\nrows=[" << ib.rows.first << ", " << ib.rows.second << ") cols=["
         << ib.cols.first << ", " << ib.cols.second << ")\n";

    lbm::BoundarySet bc;
    bc.row_lo = bc.row_hi = LBM_EDGE_ABB_VELOCITY;  // :135-154, u_w = (lp.u, 0)
    bc.col_lo = bc.col_hi = LBM_EDGE_SPECULAR;      // :157-163
    bc.uw_r = lp.u;
    bc.uw_c = 0.0;
    lbm::Solver sv = lbm::Solver::bgk(lp.X, lp.Y, lp.omega, /*incompressible=*/false, bc,
                                      /*delta_form=*/true);  // :103-108, :123-125
    sv.attach(ib.handle(), 1.0 / 3.0, 1.0 / 9.0);             // ics2, ics4 of :66-67
    sv.set_f(f_adve);
    std::array<double, 2> F_s{};
    for (int t = 0; t < steps;) {
      const int n = std::min(sp.snapshot_steps > 0 ? sp.snapshot_steps : steps, steps - t);
      sv.step(n, true);
      t += n;
      F_s = ib.surface_force();
      cout << t << "; t=" << t * lp.dt << " s  F_s=(" << F_s[0] << ", " << F_s[1] << ")\n";
    }
    if (!dump.empty()) {
      auto m = sv.moments();
      dump_f64(dump + "-rho.f64", m.first);
      dump_f64(dump + "-u.f64", m.second);
      dump_f64(dump + "-f.f64", sv.get_f());
    }
    cout.precision(17);
    cout << "steps=" << steps << "\nFs_r=" << F_s[0] << "\nFs_c=" << F_s[1] << "\n";
  } catch (const std::exception& e) {
    cerr << "error: " << e.what() << "\n";
    return 3;
  }
  return 0;
}
