// Flow past an immersed-boundary cylinder (BASELINE config 5, test/cylinder_test.cpp) slab-decomposed
// along the streamwise direction r over the GPUs of one node: a CHAIN of slabs -- the global rows 0
// and Rg-1 are the anti-bounce-back velocity inlet / outlet (:135-154), the columns are specular
// walls (:157-163) -- with the cylinder owned by one slab (lbm_ibm_create_slab).  One ghost row,
// 3 populations per side per step (the block binding of test/decompose_domain.cpp:181-187), one
// packed message per neighbour, exchange overlapped with the interior rows AND the forcing.
// C++ host on lbm_ring_bgk_step_ibm; one process per GPU.
//
//   slab_ring_cylinder --spawn N [--rows R_per_gpu] [--cols C] [--steps K] [--warmup W]
//                      [--diameter D] [--edge-rows E] [--check 1]
//   RANK=i WORLD_SIZE=N LOCAL_RANK=i slab_ring_cylinder --id-file /tmp/x ...
//
// tau = 0.55 (parameters.toml), u_in = 0.04 (SURVEY 8d allows a smaller u for the benchmark), markers
// on a circle of the given diameter with spacing ~1 centred in the middle of slab N/4 (SURVEY 8e:
// "place the cylinder away from seams and assert") and at column C/2, generated deterministically.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "ring_common.hpp"

namespace {

struct Args {
  int rows = 2048, cols = 4096, steps = 50, warmup = 5, edge_rows = 8, check = 0, diameter = 300;
  std::string id_file;
};
const double kTau = 0.55, kUin = 0.04, kGuoA = 1.0 / 3.0, kGuoB = 1.0 / 9.0;  // cylinder_test.cpp:66-67

void cylinder_markers(int Rg, int C, int R, int world, int diameter, std::vector<double>& x,
                      std::vector<double>& y, int& owner) {
  owner = world / 4;
  const double cx = owner * (double)R + R / 2.0, cy = C / 2.0, rad = diameter / 2.0;
  const int m = (int)std::lround(3.14159265358979323846 * diameter);
  x.resize(m);
  y.resize(m);
  for (int i = 0; i < m; ++i) {
    const double t = 2.0 * 3.14159265358979323846 * i / m;
    x[i] = cx + rad * std::cos(t);
    y[i] = cy + rad * std::sin(t);
  }
  (void)Rg;
}

lbm_bc global_bc() {
  lbm_bc bc{};
  bc.row_lo = bc.row_hi = LBM_EDGE_ABB_VELOCITY;
  bc.col_lo = bc.col_hi = LBM_EDGE_SPECULAR;
  bc.rho_inlet = bc.rho_outlet = 1.0;
  bc.uw_r = kUin;
  bc.uw_c = 0.0;
  return bc;
}

// pre-collision lattice [9][R+2G][C]: incompressible equilibrium of rho = 1, u = (u_in, 0) (:75,:85)
double* uniform_inflow(int R, int C, int G) {
  const size_t n = (size_t)R * C, plane = (size_t)(R + 2 * G) * C;
  std::vector<double> hu(2 * n, 0.0), hr(n, 1.0);
  for (size_t i = 0; i < n; ++i) hu[i] = kUin;
  double *du, *dr, *dense, *lat;
  check(lbm_malloc((void**)&du, 2 * n * 8), "lbm_malloc");
  check(lbm_malloc((void**)&dr, n * 8), "lbm_malloc");
  check(lbm_malloc((void**)&dense, 9 * n * 8), "lbm_malloc");
  check(lbm_malloc((void**)&lat, 9 * plane * 8), "lbm_malloc");
  check(lbm_memcpy_h2d(du, hu.data(), 2 * n * 8, nullptr), "h2d");
  check(lbm_memcpy_h2d(dr, hr.data(), n * 8, nullptr), "h2d");
  check(lbm_incomp_equilibrium(dense, du, dr, R, C, nullptr), "lbm_incomp_equilibrium");
  check(lbm_memset(lat, 0, 9 * plane * 8, nullptr), "memset");
  for (int q = 0; q < 9; ++q)
    check(lbm_memcpy_d2d(lat + q * plane + (size_t)G * C, dense + q * n, n * 8, nullptr), "d2d");
  check(lbm_stream_sync(nullptr), "sync");
  for (double* p : {du, dr, dense}) lbm_free(p);
  return lat;
}

int run_rank(const Args& a, int rank, int world, int local_rank) {
  check(lbm_set_device(local_rank), "lbm_set_device");
  const int R = a.rows, C = a.cols, Rg = R * world, G = 1;
  lbm_geom g{R, C, G, 0};
  lbm_bgk_params prm{};
  prm.omega = 1.0 / kTau;
  prm.incompressible = 0;
  prm.delta_form = 1;  // :123-125
  lbm_bc bc = global_bc(), slab_bc = bc;
  if (rank > 0) slab_bc.row_lo = LBM_EDGE_HALO;
  if (rank < world - 1) slab_bc.row_hi = LBM_EDGE_HALO;

  std::vector<double> mx, my;
  int owner = 0;
  cylinder_markers(Rg, C, R, world, a.diameter, mx, my, owner);
  lbm_ibm* ib = nullptr;
  if (rank == owner)
    check(lbm_ibm_create_slab(&ib, mx.data(), my.data(), (int)mx.size(), 5, R, C, owner * R), "lbm_ibm_create_slab");

  unsigned char id[128];
  share_unique_id(id, rank, world, a.id_file);
  lbm_ring* ring = nullptr;
  check(lbm_ring_create(&ring, id, rank, world, &g, /*periodic=*/0), "lbm_ring_create");

  const size_t n = (size_t)R * C, plane = (size_t)(R + 2 * G) * C;
  double *lat[2], *rho, *u;
  double* pre = uniform_inflow(R, C, G);
  check(lbm_malloc((void**)&lat[0], 9 * plane * 8), "lbm_malloc");
  check(lbm_malloc((void**)&lat[1], 9 * plane * 8), "lbm_malloc");
  check(lbm_malloc((void**)&rho, n * 8), "lbm_malloc");
  check(lbm_malloc((void**)&u, 2 * n * 8), "lbm_malloc");
  for (double* p : {lat[0], lat[1]}) check(lbm_memset(p, 0, 9 * plane * 8, nullptr), "memset");
  // first iteration (:103-127): moments, collision in delta form, forcing + source on the owner
  check(lbm_bgk_collide(lat[0], pre, &g, &slab_bc, &prm, rho, u, nullptr), "lbm_bgk_collide");
  if (ib) {
    check(lbm_ibm_force(ib, u, rho, nullptr, nullptr), "lbm_ibm_force");
    check(lbm_ibm_add_source(ib, lat[0], &g, u, prm.omega, kGuoA, kGuoB, nullptr), "lbm_ibm_add_source");
  }
  check(lbm_ring_exchange(ring, lat[0], nullptr), "lbm_ring_exchange");
  check(lbm_ring_join(ring, nullptr), "lbm_ring_join");
  check(lbm_stream_sync(nullptr), "sync");
  lbm_free(pre);

  int cur = 0;
  auto step = [&]() {
    check(lbm_ring_bgk_step_ibm(ring, lat[cur ^ 1], lat[cur], &bc, &prm, a.edge_rows, ib, kGuoA, kGuoB, rho, u,
                                nullptr), "lbm_ring_bgk_step_ibm");
    cur ^= 1;
  };
  for (int i = 0; i < a.warmup; ++i) step();
  check(lbm_stream_sync(nullptr), "sync");
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < a.steps; ++i) step();
  check(lbm_stream_sync(nullptr), "sync");
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double tmax = max_time_over_ranks(sec, rank, world, a.id_file);
  double Fs[2] = {0, 0};
  if (ib) {
    check(lbm_ibm_surface_force(ib, Fs, nullptr), "lbm_ibm_surface_force");
    write_file_atomic(a.id_file + ".g" + std::to_string(rank), Fs, sizeof Fs);
  }

  int bad = 0;
  if (a.check) {
    std::vector<double> h(9 * plane), own((size_t)9 * n);
    check(lbm_memcpy_d2h(h.data(), lat[cur], h.size() * 8, nullptr), "d2h");
    check(lbm_stream_sync(nullptr), "sync");
    for (int q = 0; q < 9; ++q) std::memcpy(&own[(size_t)q * n], &h[q * plane + (size_t)G * C], n * 8);
    write_file_atomic(a.id_file + ".f" + std::to_string(rank), own.data(), own.size() * 8);
    if (rank == 0) {
      // the same run as ONE block through the solver context (collide-first, fused steps, forcing)
      lbm_geom gw{Rg, C, 0, 0};
      lbm_solver* sv = nullptr;
      check(lbm_solver_create(&sv, LBM_MODEL_BGK, &gw, &bc, &prm, nullptr), "lbm_solver_create");
      lbm_ibm* ibw = nullptr;
      check(lbm_ibm_create(&ibw, mx.data(), my.data(), (int)mx.size(), 5, Rg, C), "lbm_ibm_create");
      check(lbm_solver_attach_ibm(sv, ibw, kGuoA, kGuoB), "lbm_solver_attach_ibm");
      double* prew = uniform_inflow(Rg, C, 0);
      check(lbm_solver_set_f_soa_dev(sv, prew), "lbm_solver_set_f_soa_dev");
      check(lbm_solver_step(sv, 1 + a.warmup + a.steps, 0), "lbm_solver_step");
      double *cl = nullptr, *ol = nullptr;
      lbm_geom gg;
      check(lbm_solver_lattices(sv, &cl, &ol, &gg), "lbm_solver_lattices");
      const long long ps = gg.plane_stride ? gg.plane_stride : (long long)Rg * C;
      std::vector<double> want((size_t)Rg * C);
      for (int q = 0; q < 9; ++q) {
        check(lbm_memcpy_d2h(want.data(), cl + q * ps, want.size() * 8, nullptr), "d2h");
        check(lbm_solver_sync(sv), "sync");
        check(lbm_stream_sync(nullptr), "sync");
        for (int r = 0; r < world; ++r) {
          wait_file(a.id_file + ".f" + std::to_string(r), own.data(), own.size() * 8);
          if (std::memcmp(&own[(size_t)q * n], &want[(size_t)r * n], n * 8) != 0) ++bad;
        }
      }
      double Fw[2], Fo[2];
      check(lbm_ibm_surface_force(ibw, Fw, nullptr), "lbm_ibm_surface_force");
      wait_file(a.id_file + ".g" + std::to_string(owner), Fo, sizeof Fo);
      if (std::memcmp(Fw, Fo, sizeof Fw) != 0) ++bad;
      lbm_solver_destroy(sv);
      lbm_ibm_destroy(ibw);
      lbm_free(prew);
    }
  }
  if (rank == 0) {
    double Fo[2] = {0, 0};
    if (world > 1 || ib) {
      if (ib) std::memcpy(Fo, Fs, sizeof Fo);
      else wait_file(a.id_file + ".g" + std::to_string(owner), Fo, sizeof Fo);
    }
    std::printf("{\"driver\": \"slab_ring_cylinder\", \"n_gpus\": %d, \"rows_per_gpu\": %d, \"cols\": %d, "
                "\"markers\": %d, \"owner_rank\": %d, \"steps\": %d, \"ms_per_step\": %.4f, \"mlups\": %.1f, "
                "\"Fs\": [%.17g, %.17g], \"transport\": \"rccl send/recv (C++ ring)\"%s}\n",
                world, R, C, (int)mx.size(), owner, a.steps, 1e3 * tmax / a.steps,
                (double)Rg * C * a.steps / tmax / 1e6, Fo[0], Fo[1],
                a.check ? (bad ? ", \"check\": \"MISMATCH\"" : ", \"check\": \"bitwise equal to one block\"") : "");
    std::fflush(stdout);
  }
  lbm_ring_destroy(ring);
  if (ib) lbm_ibm_destroy(ib);
  for (double* p : {lat[0], lat[1], rho, u}) lbm_free(p);
  return bad ? 3 : 0;
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  a.rows = std::atoi(arg_value(argc, argv, "--rows", "2048").c_str());
  a.cols = std::atoi(arg_value(argc, argv, "--cols", "4096").c_str());
  a.steps = std::atoi(arg_value(argc, argv, "--steps", "50").c_str());
  a.warmup = std::atoi(arg_value(argc, argv, "--warmup", "5").c_str());
  a.edge_rows = std::atoi(arg_value(argc, argv, "--edge-rows", "8").c_str());
  a.diameter = std::atoi(arg_value(argc, argv, "--diameter", "300").c_str());
  a.check = std::atoi(arg_value(argc, argv, "--check", "0").c_str());
  a.id_file = arg_value(argc, argv, "--id-file", "/tmp/lbm_ring_id." + std::to_string((long)getpid()));
  const int spawn = std::atoi(arg_value(argc, argv, "--spawn", "0").c_str());
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
    std::fprintf(stderr, "slab_ring_cylinder: %s\n", e.what());
    return 1;
  }
}
