// shared by the slab-ring drivers: status check, the file rendezvous that distributes the RCCL
// unique id (no MPI in this image), and the fork-one-rank-per-GPU launcher.
#pragma once
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lbm_hip.h"

inline void check(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + ": " + lbm_last_error_string());
}

inline bool read_file(const std::string& path, void* buf, size_t n) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  size_t got = std::fread(buf, 1, n, f);
  std::fclose(f);
  return got == n;
}
inline void write_file_atomic(const std::string& path, const void* buf, size_t n) {
  std::string tmp = path + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot write " + tmp);
  std::fwrite(buf, 1, n, f);
  std::fclose(f);
  std::rename(tmp.c_str(), path.c_str());
}
inline void wait_file(const std::string& path, void* buf, size_t n, double timeout_s = 120) {
  auto t0 = std::chrono::steady_clock::now();
  while (!read_file(path, buf, n)) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      throw std::runtime_error("timed out waiting for " + path);
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
}


// fork BEFORE anything touches the GPU; every child is an ordinary one-GPU process
template <class F>
int spawn_ranks(int n, F&& run) {
  std::vector<pid_t> kids;
  for (int r = 0; r < n; ++r) {
    pid_t pid = fork();
    if (pid == 0) {
      int rc = 1;
      try {
        rc = run(r);
      } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %d: %s\n", r, e.what());
      }
      std::fflush(nullptr);
      _exit(rc);
    }
    kids.push_back(pid);
  }
  int worst = 0;
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) worst = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
  }
  return worst;
}
inline void cleanup_ring_files(const std::string& id_file, int n) {
  for (const char* suf : {"", ".tmp"}) std::remove((id_file + suf).c_str());
  for (int r = 0; r < n; ++r)
    for (const char* suf : {".t", ".f", ".g"}) std::remove((id_file + suf + std::to_string(r)).c_str());
}
// rank 0 creates the id and publishes it; the others wait for the file
inline void share_unique_id(unsigned char (&id)[128], int rank, int world, const std::string& id_file) {
  if (rank == 0) {
    check(lbm_ring_unique_id(id), "lbm_ring_unique_id");
    if (world > 1) write_file_atomic(id_file, id, sizeof id);
  } else {
    wait_file(id_file, id, sizeof id);
  }
}
// slowest rank's time, gathered on rank 0 through files
inline double max_time_over_ranks(double sec, int rank, int world, const std::string& id_file) {
  double tmax = sec;
  if (world > 1) {
    write_file_atomic(id_file + ".t" + std::to_string(rank), &sec, sizeof sec);
    if (rank == 0)
      for (int r = 1; r < world; ++r) {
        double t;
        wait_file(id_file + ".t" + std::to_string(r), &t, sizeof t);
        tmax = t > tmax ? t : tmax;
      }
  }
  return tmax;
}
