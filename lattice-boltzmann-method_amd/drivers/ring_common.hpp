// shared by the slab-ring drivers: status check, the file rendezvous that distributes the RCCL
// unique id (no MPI in this image), and the fork-one-rank-per-GPU launcher.
#pragma once
#include <signal.h>
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


// fork BEFORE anything touches the GPU; every child is an ordinary one-GPU process.  Children are
// reaped in exit order: the first one that fails takes its peers with it (SIGTERM, then SIGKILL
// after a grace period) -- a rank blocked in ncclCommInitRank / send / recv on a dead peer would
// otherwise hold its GPU until an outer timeout.
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
    if (pid < 0) {
      std::perror("fork");
      for (pid_t k : kids) kill(k, SIGKILL);  // (all still our un-reaped children here)
      for (pid_t k : kids) waitpid(k, nullptr, 0);
      return 1;
    }
    kids.push_back(pid);
  }
  int worst = 0;
  size_t left = kids.size();
  bool killing = false, killed_hard = false;
  auto t_kill = std::chrono::steady_clock::now();
  while (left > 0) {
    int st = 0;
    pid_t k = waitpid(-1, &st, killing ? WNOHANG : 0);
    if (k > 0) {
      --left;
      for (pid_t& o : kids)
        if (o == k) o = -1;  // reaped: its pid may be handed to an unrelated process from now on -- never signal it again
      const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
      if (rc != 0 && worst == 0) worst = rc;
      if (rc != 0 && !killing) {  // first failure: stop the survivors
        killing = true;
        t_kill = std::chrono::steady_clock::now();
        for (pid_t o : kids)
          if (o > 0) kill(o, SIGTERM);
      }
    } else if (k == 0) {  // survivors still running after SIGTERM
      if (!killed_hard && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_kill).count() > 5.0) {
        killed_hard = true;  // once
        for (pid_t o : kids)
          if (o > 0) kill(o, SIGKILL);
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    } else {
      break;  // ECHILD: nothing left to wait for
    }
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
// after the timed launches (and their sync): a rank whose ring has given up on a neighbour -- a bounded wait of the
// peer-mapped transport, an asynchronous RCCL error -- has void lattices and void timings: it says so as JSON and the
// driver returns 4 instead of printing a rate (its peers are ended by the launcher)
inline int ring_failed(lbm_ring* ring, const char* driver, int rank) {
  if (lbm_ring_status(ring) == 0) return 0;
  std::string msg = lbm_last_error_string();
  for (char& ch : msg)
    if (ch == '"' || ch == '\\') ch = '\'';  // (the message goes into a JSON string)
  std::printf("{\"driver\": \"%s\", \"rank\": %d, \"error\": \"%s\"}\n", driver, rank, msg.c_str());
  std::fflush(stdout);
  return 4;
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
