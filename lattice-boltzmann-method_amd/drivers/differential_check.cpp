// Test helper for the facade class `differential` (src/differential.hpp:48-51,
// src/differential.cpp:23-39): reads psi [R][C] as raw f64, writes differential::x(psi),
// differential::y(psi) and differential::grad(psi) ([R][C][2], reference layout) as raw f64.
//   differential_check R C psi.bin dx.bin dy.bin grad.bin
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "../include/lbm/lbm.hpp"

static std::vector<double> slurp(const char* path, size_t n) {
  std::vector<double> v(n);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(v.data(), sizeof(double), n, f) != n) throw std::runtime_error(std::string("cannot read ") + path);
  std::fclose(f);
  return v;
}
static void dump(const char* path, const std::vector<double>& v) {
  FILE* f = std::fopen(path, "wb");
  if (!f || std::fwrite(v.data(), sizeof(double), v.size(), f) != v.size()) throw std::runtime_error(std::string("cannot write ") + path);
  std::fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 7) {
    std::cerr << "usage: differential_check R C psi.bin dx.bin dy.bin grad.bin\n";
    return 1;
  }
  try {
    const int R = std::atoi(argv[1]), C = std::atoi(argv[2]);
    lbm::Field psi(R, C, 1);
    psi.from_host(slurp(argv[3], (size_t)R * C));
    differential diff;
    dump(argv[4], diff.x(psi).to_host());
    dump(argv[5], diff.y(psi).to_host());
    lbm::Field g(R, C, 2);
    diff.grad(g, psi);
    dump(argv[6], g.to_host());
  } catch (const std::exception& e) {
    std::cerr << e.what() << "\n";
    return 2;
  }
  return 0;
}
