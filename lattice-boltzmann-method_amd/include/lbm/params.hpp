// params::flow / lattice / simulation -- the reference's params.toml surface
// (src/params.hpp:9-49, src/params.cpp:7-120): same TOML tables and keys, same derived
// quantities, same std::runtime_error texts on a missing key.  Host-only, no GPU.
#pragma once
#include <cmath>
#include <iostream>
#include <optional>
#include <stdexcept>
#include <string>

#include "toml.hpp"

namespace params {

namespace detail {
inline double need(const lbm::toml::table& tbl, const char* sect, const char* key) {
  std::optional<double> v = tbl[sect][key].value<double>();
  if (v.has_value()) return v.value();
  throw std::runtime_error(std::string(key) + " not defined in parameters file");
}
}  // namespace detail

struct flow {  // src/params.cpp:7-29
  double nu{0}, u{0}, l{0}, rho_0{0}, Re{0};
  explicit flow(const lbm::toml::table& tbl) {
    rho_0 = detail::need(tbl, "flow", "initial_density");
    nu = detail::need(tbl, "flow", "kinematic_viscosity");
    u = detail::need(tbl, "flow", "characteristic_velocity");
    l = detail::need(tbl, "flow", "characteristic_length");
    Re = u * l / nu;
  }
};

struct lattice {  // src/params.cpp:31-66
  const double cs2{1.0 / 3.0};
  double tau, omega, Re, nu;
  int l;
  double dx, dt;
  int T;
  double u;
  int Y, X;
  lattice(const lbm::toml::table& tbl, const flow& fp) {
    tau = detail::need(tbl, "lattice", "relaxation_time");
    dx = detail::need(tbl, "lattice", "lattice_spacing");
    const double x_mult = detail::need(tbl, "lattice", "x_multiplier");
    const double y_mult = detail::need(tbl, "lattice", "y_multiplier");
    // characteristic length -> nearest odd integer (:52-54)
    if ((int)std::ceil(fp.l / dx) % 2 != 0) l = std::ceil(fp.l / dx);
    else l = std::floor(fp.l / dx);
    omega = 1.0 / tau;
    Re = fp.Re;
    nu = cs2 * (tau - 0.5);
    u = fp.Re * nu / l;
    dt = cs2 * (tau - 0.5) * (dx * dx) / fp.nu;
    T = std::ceil(1.0 / dt);
    X = std::ceil(l * x_mult);
    Y = std::ceil(l * y_mult);
  }
};

struct simulation {  // src/params.cpp:95-120
  double stop_time, snapshot_period;
  int total_steps, snapshot_steps, total_snapshots;
  std::string file_prefix;
  simulation(const lbm::toml::table& tbl, const lattice& lp) {
    stop_time = detail::need(tbl, "simulation", "stop_time");
    snapshot_period = detail::need(tbl, "simulation", "snapshot_period");
    std::optional<std::string> p = tbl["simulation"]["file_prefix"].value<std::string>();
    if (p.has_value()) file_prefix = p.value();
    else throw std::runtime_error("file_prefix not defined in parameters file");
    total_steps = std::ceil(stop_time * lp.T);
    snapshot_steps = std::ceil(snapshot_period * lp.T);
    total_snapshots = std::ceil((total_steps + 0.0) / snapshot_steps);
  }
  bool snapshot(int step) const { return step % snapshot_steps == 0; }
};

inline std::ostream& operator<<(std::ostream& os, const flow& p) {  // :68-76
  return os << "Flow parameters:\n" << "nu=" << p.nu << " m2/s\n" << "u=" << p.u << " m/s\n"
            << "l=" << p.l << " m\n" << "rho_0=" << p.rho_0 << " kg/m3\n" << "Re=" << p.Re << std::endl;
}
inline std::ostream& operator<<(std::ostream& os, const lattice& p) {  // :78-92
  return os << "Lattice parameters:" << std::endl << "Re=" << p.Re << "\n" << "tau=" << p.tau << "\n"
            << "omega=" << p.omega << "\n" << "dx=" << p.dx << " m\n" << "l=" << p.l << "\n"
            << "nu=" << p.nu << "\n" << "u=" << p.u << "\n" << "dt=" << p.dt << "s\n"
            << "T=" << p.T << "\n" << "X=" << p.X << "\n" << "Y=" << p.Y << std::endl;
}
inline std::ostream& operator<<(std::ostream& os, const simulation& p) {  // :122-128
  return os << "Simulation parameters:\n" << "stop time: " << p.stop_time << " s (" << p.total_steps
            << " steps)\n" << "saving results each " << p.snapshot_period << " s (" << p.snapshot_steps
            << " steps)\n" << "for a total of " << p.total_snapshots << " snapshots" << std::endl;
}

}  // namespace params
