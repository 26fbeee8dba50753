// Periodic BGK box, slab-decomposed along r over the GPUs of one node: the C++ host of the
// multi-GPU path (north star: "halo exchange via RCCL send/recv over xGMI overlapped with interior
// collision on a second HIP stream").  One process per GPU.
//
//   slab_ring_box --spawn N [...]          fork N ranks on this node (rank i -> GPU i)
//   RANK=i WORLD_SIZE=N LOCAL_RANK=i slab_ring_box --id-file /tmp/x [...]   under any launcher
//
// Options: --rows R (per GPU, weak scaling) --cols C --steps K (launch-steps timed) --warmup W
//          --depth D (time steps per launch, 1..6; KBC: 1..4) --period P (launches per halo exchange,
//          ghost rows = P x D; default 2) --edge-rows E --omega w
//          --model bgk|kbc (kbc: the entropic KBC collision with s2 = omega, config 3 over slabs)
//          --check 1 (N ranks vs rank 0 recomputing the whole box: small sizes only)
//          --desert R (fault injection: rank R joins the ring and leaves; the others must report it and return 4)
//
// The block binding this generalises: test/decompose_domain.cpp:181-187 (3 populations per
// interface row, one row per step); here 9(D-1) rows per side per D-step launch (3 for D = 1).
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lbm_hip.h"
#include "common.hpp"
#include "ring_common.hpp"

namespace {

// Taylor-Green-like smooth field on the GLOBAL box -> compressible equilibrium (solver.cpp:51-62)
// for global row gr, column c; written in the reference's operation order so that the N-rank and
// the 1-rank initial states are the same bits.
void init_node(double* f9, int gr, int c, int Rg, int C) {
  const double pi = 3.14159265358979323846;
  const double x = 2 * pi * gr / Rg, y = 2 * pi * c / C;
  const double u0 = 0.04 * std::sin(x) * std::cos(y), u1 = -0.04 * std::cos(x) * std::sin(y);
  const double rho = 1.0 + 0.01 * std::cos(2 * x);
  static const double w[9] = {4. / 9, 1. / 9, 1. / 9, 1. / 9, 1. / 9, 1. / 36, 1. / 36, 1. / 36, 1. / 36};
  static const int cx[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1}, cy[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
  const double uu = u0 * u0 + u1 * u1;
  for (int q = 0; q < 9; ++q) {
    const double cu = cx[q] * u0 + cy[q] * u1;
    f9[q] = w[q] * rho * (1.0 + 3.0 * cu + 4.5 * cu * cu - 1.5 * uu);
  }
}

struct Args {
  int rows = 8192, cols = 8192, steps = 20, warmup = 5, depth = 5, period = 2, edge_rows = 32, check = 0, desert = -1;
  double omega = 1.2;
  bool kbc = false;
  std::string id_file;
};

// post-collision slab lattice [9][R+2D][C] of rows [row0, row0+R) of an Rg x C box
double* make_slab(const Args& a, int R, int row0, int Rg, const lbm_geom& g, const lbm_bgk_params& prm) {
  const int C = a.cols, D = g.ghost;
  const size_t plane = (size_t)(R + 2 * D) * C;
  std::vector<double> h(9 * plane, 0.0);
  double f9[9];
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      init_node(f9, row0 + r, c, Rg, C);
      for (int q = 0; q < 9; ++q) h[q * plane + (size_t)(r + D) * C + c] = f9[q];
    }
  double *pre = nullptr, *post = nullptr;
  check(lbm_malloc((void**)&pre, 9 * plane * sizeof(double)), "lbm_malloc");
  check(lbm_malloc((void**)&post, 9 * plane * sizeof(double)), "lbm_malloc");
  check(lbm_memcpy_h2d(pre, h.data(), 9 * plane * sizeof(double), nullptr), "h2d");
  check(lbm_memset(post, 0, 9 * plane * sizeof(double), nullptr), "memset");
  // the collide-only launch that opens the post-collision-resident loop (ghost rows: collide of
  // zeros stays in the ghost rows and is overwritten by the first exchange)
  if (a.kbc) {
    lbm_kbc_params kp{prm.omega, LBM_FORM_DEFAULT};
    check(lbm_kbc_collide(post, pre, &g, nullptr, &kp, nullptr, nullptr, nullptr), "lbm_kbc_collide");
  } else {
    check(lbm_bgk_collide(post, pre, &g, nullptr, &prm, nullptr, nullptr, nullptr), "lbm_bgk_collide");
  }
  check(lbm_stream_sync(nullptr), "sync");
  lbm_free(pre);
  return post;
}

int run_rank(const Args& a, int rank, int world, int local_rank) {
  check(lbm_set_device(std::getenv("LBM_ONE_GPU") ? 0 : local_rank), "lbm_set_device");
  const int R = a.rows, C = a.cols, D = a.depth, Rg = R * world;
  // ghost = period x D rows: lbm_ring_bgk_step / _kbc_step exchange once per `period` launches
  const int G = D * (D < 2 ? 1 : a.period);
  lbm_geom g{R, C, G, 0, 0};
  lbm_bgk_params prm{};
  prm.omega = a.omega;
  lbm_kbc_params kprm{a.omega, LBM_FORM_DEFAULT};

  unsigned char id[128];
  if (rank == 0) {
    check(lbm_ring_unique_id(id), "lbm_ring_unique_id");
    if (world > 1) write_file_atomic(a.id_file, id, sizeof id);
  } else {
    wait_file(a.id_file, id, sizeof id);
  }
  lbm_ring* ring = nullptr;
  check(lbm_ring_create(&ring, id, rank, world, &g, /*periodic=*/1), "lbm_ring_create");
  if (rank == a.desert) {  // joined, mapped, gone
    std::fflush(nullptr);
    _exit(0);
  }

  const size_t plane = (size_t)(R + 2 * G) * C;
  double* lat[2];
  lat[0] = make_slab(a, R, rank * R, Rg, g, prm);
  check(lbm_malloc((void**)&lat[1], 9 * plane * sizeof(double)), "lbm_malloc");
  check(lbm_memset(lat[1], 0, 9 * plane * sizeof(double), nullptr), "memset");
  check(lbm_ring_exchange(ring, lat[0], nullptr), "lbm_ring_exchange");
  check(lbm_ring_join(ring, nullptr), "lbm_ring_join");

  int cur = 0;
  auto launch = [&]() {
    if (a.kbc) check(lbm_ring_kbc_step(ring, lat[cur ^ 1], lat[cur], nullptr, &kprm, D, a.edge_rows, nullptr), "lbm_ring_kbc_step");
    else check(lbm_ring_bgk_step(ring, lat[cur ^ 1], lat[cur], nullptr, &prm, D, a.edge_rows, nullptr), "lbm_ring_bgk_step");
    cur ^= 1;
  };
  for (int i = 0; i < a.warmup; ++i) launch();
  check(lbm_stream_sync(nullptr), "sync");
  // (each rank starts its clock after its own warm-up; the neighbour exchanges keep ranks in step)
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < a.steps; ++i) launch();
  check(lbm_stream_sync(nullptr), "sync");
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (const int failed = ring_failed(ring, "slab_ring_box", rank)) return failed;

  // gather the per-rank times through files (no MPI here); value = all nodes / slowest rank
  double tmax = sec;
  if (world > 1) {
    write_file_atomic(a.id_file + ".t" + std::to_string(rank), &sec, sizeof sec);
    if (rank == 0)
      for (int r = 1; r < world; ++r) {
        double t;
        wait_file(a.id_file + ".t" + std::to_string(r), &t, sizeof t);
        tmax = std::max(tmax, t);
      }
  }

  int bad = 0;
  if (a.check) {
    // every rank dumps its owned rows; rank 0 recomputes the whole box on its own GPU (ghost 0,
    // periodic wrap inside the block) and compares bit for bit
    std::vector<double> h(9 * plane);
    check(lbm_memcpy_d2h(h.data(), lat[cur], h.size() * sizeof(double), nullptr), "d2h");
    check(lbm_stream_sync(nullptr), "sync");
    std::vector<double> own((size_t)9 * R * C);
    for (int q = 0; q < 9; ++q)
      std::memcpy(&own[(size_t)q * R * C], &h[q * plane + (size_t)G * C], (size_t)R * C * sizeof(double));
    write_file_atomic(a.id_file + ".f" + std::to_string(rank), own.data(), own.size() * sizeof(double));
    if (rank == 0) {
      Args whole = a;
      lbm_geom gw{Rg, C, 0, 0, 0};
      double* p = make_slab(whole, Rg, 0, Rg, gw, prm);
      double* q2 = nullptr;
      const size_t n = (size_t)9 * Rg * C;
      check(lbm_malloc((void**)&q2, n * sizeof(double)), "lbm_malloc");
      const int total = (a.warmup + a.steps) * D;
      for (int t = 0; t < total; ++t) {
        if (a.kbc) check(lbm_kbc_stream_collide(q2, p, &gw, nullptr, &kprm, 0, Rg, nullptr, nullptr, nullptr), "ref step");
        else check(lbm_bgk_stream_collide(q2, p, &gw, nullptr, &prm, 0, Rg, nullptr, nullptr, nullptr), "ref step");
        std::swap(p, q2);
      }
      std::vector<double> want(n);
      check(lbm_memcpy_d2h(want.data(), p, n * sizeof(double), nullptr), "d2h");
      check(lbm_stream_sync(nullptr), "sync");
      for (int r = 0; r < world; ++r) {
        wait_file(a.id_file + ".f" + std::to_string(r), own.data(), own.size() * sizeof(double));
        for (int q = 0; q < 9; ++q)
          if (std::memcmp(&own[(size_t)q * R * C], &want[(size_t)q * Rg * C + (size_t)r * R * C],
                          (size_t)R * C * sizeof(double)) != 0)
            ++bad;
      }
      lbm_free(p);
      lbm_free(q2);
    }
  }

  if (rank == 0) {
    const double lups = (double)Rg * C * D * a.steps / tmax;
    std::printf("{\"driver\": \"slab_ring_box\", \"model\": \"%s\", \"n_gpus\": %d, \"rows_per_gpu\": %d, \"cols\": %d, "
                "\"depth\": %d, \"ghost_rows\": %d, \"launches\": %d, \"ms_per_launch\": %.4f, \"mlups\": %.1f, "
                "\"transport\": \"rccl send/recv (C++ ring)\"%s}\n",
                a.kbc ? "kbc" : "bgk", world, R, C, D, G, a.steps, 1e3 * tmax / a.steps, lups / 1e6,
                a.check ? (bad ? ", \"check\": \"MISMATCH\"" : ", \"check\": \"bitwise equal to one block\"") : "");
    std::fflush(stdout);
  }
  lbm_ring_destroy(ring);
  lbm_free(lat[0]);
  lbm_free(lat[1]);
  return bad ? 3 : 0;
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  a.rows = std::atoi(arg_value(argc, argv, "--rows", "8192").c_str());
  a.cols = std::atoi(arg_value(argc, argv, "--cols", "8192").c_str());
  a.steps = std::atoi(arg_value(argc, argv, "--steps", "20").c_str());
  a.warmup = std::atoi(arg_value(argc, argv, "--warmup", "5").c_str());
  a.depth = std::atoi(arg_value(argc, argv, "--depth", "5").c_str());
  a.period = std::max(1, std::min(3, std::atoi(arg_value(argc, argv, "--period", "2").c_str())));
  a.edge_rows = std::atoi(arg_value(argc, argv, "--edge-rows", "32").c_str());
  a.check = std::atoi(arg_value(argc, argv, "--check", "0").c_str());
  a.desert = std::atoi(arg_value(argc, argv, "--desert", "-1").c_str());
  a.omega = std::atof(arg_value(argc, argv, "--omega", "1.2").c_str());
  a.kbc = arg_value(argc, argv, "--model", "bgk") == "kbc";
  if (a.kbc && a.depth > 4) a.depth = 3;
  a.id_file = arg_value(argc, argv, "--id-file", "/tmp/lbm_ring_id." + std::to_string((long)getpid()));
  const int spawn = std::atoi(arg_value(argc, argv, "--spawn", "0").c_str());
  // --transport rccl|ipc: what carries the ring's messages (lbm_ring_unique_id / lbm_ring_create follow the environment);
  // --one-gpu 1: every rank on GPU 0 (with ipc: N real ranks on one device, which RCCL refuses)
  const std::string transport = arg_value(argc, argv, "--transport", "");
  if (!transport.empty()) setenv("LBM_RING_TRANSPORT", transport.c_str(), 1);
  if (std::atoi(arg_value(argc, argv, "--one-gpu", "0").c_str())) setenv("LBM_ONE_GPU", "1", 1);
  try {
    if (spawn > 0) {
      cleanup_ring_files(a.id_file, spawn);  // a stale id file of a killed run must not be picked up
      const int rc = spawn_ranks(spawn, [&](int r) { return run_rank(a, r, spawn, r); });
      cleanup_ring_files(a.id_file, spawn);
      return rc;
    }
    const char* er = std::getenv("RANK");
    const char* ew = std::getenv("WORLD_SIZE");
    const char* el = std::getenv("LOCAL_RANK");
    const int rank = er ? std::atoi(er) : 0, world = ew ? std::atoi(ew) : 1;
    return run_rank(a, rank, world, el ? std::atoi(el) : rank);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "slab_ring_box: %s\n", e.what());
    return 1;
  }
}
