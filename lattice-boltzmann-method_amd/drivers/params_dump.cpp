// Host-only helper for the CPU test-suite: parse a TOML file with lbm::toml, build
// params::flow / lattice / simulation and print every field as JSON.  No GPU, no liblbm_hip.
#include <iostream>

#include "../include/lbm/params.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  try {
    auto tbl = lbm::toml::parse_file(argv[1]);
    const params::flow fp{tbl};
    const params::lattice lp{tbl, fp};
    std::cout.precision(17);
    std::cout << "{\"flow\": {\"nu\": " << fp.nu << ", \"u\": " << fp.u << ", \"l\": " << fp.l
              << ", \"rho_0\": " << fp.rho_0 << ", \"Re\": " << fp.Re << "},\n \"lattice\": {\"tau\": "
              << lp.tau << ", \"omega\": " << lp.omega << ", \"Re\": " << lp.Re << ", \"nu\": " << lp.nu
              << ", \"l\": " << lp.l << ", \"dx\": " << lp.dx << ", \"dt\": " << lp.dt << ", \"T\": " << lp.T
              << ", \"u\": " << lp.u << ", \"X\": " << lp.X << ", \"Y\": " << lp.Y << "}";
    if (tbl.contains("simulation")) {
      const params::simulation sp{tbl, lp};
      std::cout << ",\n \"simulation\": {\"total_steps\": " << sp.total_steps << ", \"snapshot_steps\": "
                << sp.snapshot_steps << ", \"total_snapshots\": " << sp.total_snapshots
                << ", \"file_prefix\": \"" << sp.file_prefix << "\"}";
    }
    if (argc > 2) {  // arrays of a named table (IBM marker files)
      auto x = tbl[argv[2]]["x"].as_array();
      std::cout << ",\n \"n_x\": " << (x ? x->size() : 0);
    }
    std::cout << "}\n";
  } catch (const std::exception& e) {
    std::cout << "{\"error\": \"" << e.what() << "\"}\n";
    return 2;
  }
  return 0;
}
