// shared by the drivers: raw snapshot writer (the reference uses torch::save, utils.cpp:21-29)
#pragma once
#include <cstdio>
#include <string>
#include <vector>

inline void dump_f64(const std::string& path, const std::vector<double>& a) {
  if (path.empty()) return;
  if (FILE* f = std::fopen(path.c_str(), "wb")) {
    std::fwrite(a.data(), sizeof(double), a.size(), f);
    std::fclose(f);
  }
}
inline std::string arg_value(int argc, char** argv, const std::string& key, const std::string& dflt) {
  for (int i = 1; i + 1 < argc; ++i)
    if (key == argv[i]) return argv[i + 1];
  return dflt;
}
