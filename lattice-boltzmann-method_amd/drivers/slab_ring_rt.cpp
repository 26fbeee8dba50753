// Rayleigh-Taylor two-phase run (BASELINE config 4) slab-decomposed along r over the GPUs of one
// node: a CHAIN of slabs (rows 0 and R-1 of the global domain are the driver's bounce-back walls,
// test/mrtcg_rayleigh_taylor.cpp:525-531), 3 ghost rows per colour, ONE packed message per
// neighbour per step (both colours), exchange overlapped with the interior rows.  C++ host on
// lbm_ring_* + lbm_cg_step_fused; one process per GPU.
//
//   slab_ring_rt --spawn N [--rows R_per_gpu] [--cols C] [--steps K] [--warmup W] [--edge-rows E]
//                [--check 1]   (rank 0 recomputes the whole domain as one block: small sizes only)
//                [--transport rccl|ipc] [--one-gpu 1]  (ipc + one-gpu: N real ranks sharing GPU 0 through the
//                peer-mapped transport -- RCCL refuses two ranks on one device)
//   slab_ring_rt --emulate N ...   ONE process / one GPU playing all N slabs of the chain in turn (edge rows, interior
//                rows, pack -> device copy -> unpack of the 3 ghost rows of both colours, per-slab time by HIP events;
//                --check 1: bitwise against the single block) -- BASELINE config 4 = --emulate 4 --rows 2048 --cols 2048
//   RANK=i WORLD_SIZE=N LOCAL_RANK=i slab_ring_rt --id-file /tmp/x ...     under any launcher
//
// Parameters: [red]/[blue] of mrtcg-rayleigh-taylor-gamma3.toml, sigma = 0.1, g = 6.25e-6
// (SURVEY 8d, C4); initial state = init_rho_cosine (:182-210), u = 0, f = feq.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "ring_common.hpp"

namespace {

struct Args {
  int rows = 2048, cols = 2048, steps = 50, warmup = 5, edge_rows = 16, check = 0, emulate = 0, one_gpu = 0, parts = 1;
  std::string id_file;
};

lbm_cg_params rt_params() {
  lbm_cg_params p{};
  p.red = lbm_cg_colour{3.0, 0.7, 0.04, 0.7};
  p.blue = lbm_cg_colour{1.0, 0.1, 0.04, -0.7};
  p.sigma = 0.1;
  p.gravity_r = 6.25e-6;
  p.gravity_c = 0.0;
  p.add_source = 1;
  p.delta = 0.1;
  return p;
}

// geometry of a slab's lattices: rows padded off a power-of-two stride like the solver contexts' (lbm_default_row_pitch:
// +3 % for the two-phase kernel at 2048 columns); `dense` = as the reference holds its tensors
lbm_geom slab_geom(int R, int C, int G, bool dense = false) {
  const int pitch = dense ? C : lbm_default_row_pitch(C);
  return lbm_geom{R, C, G, (long long)(R + 2 * G) * pitch, pitch > C ? pitch : 0};
}
size_t plane_of(const lbm_geom& g) { return (size_t)g.plane_stride; }
// the owned rows of a lattice as 9 dense planes [9][R][C] on the host
void owned_rows_to_host(double* host, const double* lattice, const lbm_geom& g, double* dense_dev) {
  const lbm_geom d{g.R, g.C, 0, 0, 0};
  check(lbm_lattice_copy_rows(dense_dev, &d, 0, lattice, &g, 0, g.R, nullptr), "lbm_lattice_copy_rows");
  check(lbm_memcpy_d2h(host, dense_dev, (size_t)9 * g.R * g.C * 8, nullptr), "d2h");
  check(lbm_stream_sync(nullptr), "sync");
}

// Post-collision lattices of rows [row0, row0 + R) of an Rg x C domain, ghost rows G (0 or 3):
// densities by init_rho_cosine, f = feq(rho_k, u = 0), then the driver's first collision
// (lbm_cg_collide on the given rho, u).  The macroscopic arrays of a slab carry 2 ghost rows.
void make_slab(int R, int C, int row0, int Rg, const lbm_geom& g, const lbm_bc& bc,
               const lbm_cg_params& prm, double** post_r, double** post_b) {
  const int G = g.ghost, mg = G ? 2 : 0;
  const size_t plane = plane_of(g), mplane = (size_t)(R + 2 * mg) * C;
  std::vector<double> hr(mplane), hb(mplane);
  for (int r = -mg; r < R + mg; ++r) {
    int gr = row0 + r;
    gr = gr < 0 ? 0 : (gr > Rg - 1 ? Rg - 1 : gr);
    for (int c = 0; c < C; ++c) {
      const double s = Rg / 2.0 - 0.1 * C * std::cos(2.0 * 3.141592 * c / C);  // :196-199
      const bool red = gr < s;
      hr[(size_t)(r + mg) * C + c] = red ? prm.red.rho_0 : 0.0;
      hb[(size_t)(r + mg) * C + c] = red ? 0.0 : prm.blue.rho_0;
    }
  }
  double *d_rr, *d_rb, *d_u, *pre_r, *pre_b;
  check(lbm_malloc((void**)&d_rr, mplane * 8), "lbm_malloc");
  check(lbm_malloc((void**)&d_rb, mplane * 8), "lbm_malloc");
  check(lbm_malloc((void**)&d_u, 2 * mplane * 8), "lbm_malloc");
  check(lbm_malloc((void**)&pre_r, 9 * plane * 8), "lbm_malloc");
  check(lbm_malloc((void**)&pre_b, 9 * plane * 8), "lbm_malloc");
  check(lbm_malloc((void**)post_r, 9 * plane * 8), "lbm_malloc");
  check(lbm_malloc((void**)post_b, 9 * plane * 8), "lbm_malloc");
  check(lbm_memcpy_h2d(d_rr, hr.data(), mplane * 8, nullptr), "h2d");
  check(lbm_memcpy_h2d(d_rb, hb.data(), mplane * 8, nullptr), "h2d");
  check(lbm_memset(d_u, 0, 2 * mplane * 8, nullptr), "memset");
  for (double* p : {pre_r, pre_b, *post_r, *post_b}) check(lbm_memset(p, 0, 9 * plane * 8, nullptr), "memset");
  // feq on the owned rows (u = 0: dense zeros) as dense planes, then into the ghosted, row-padded lattices
  {
    double* eq = nullptr;
    check(lbm_malloc((void**)&eq, (size_t)9 * R * C * 8), "lbm_malloc");
    const lbm_geom dense{R, C, 0, 0, 0};
    check(lbm_cg_equilibrium(eq, d_rr + (size_t)mg * C, d_u, &prm.red, R, C, 0, nullptr), "lbm_cg_equilibrium");
    check(lbm_lattice_copy_rows(pre_r, &g, 0, eq, &dense, 0, R, nullptr), "lbm_lattice_copy_rows");
    check(lbm_cg_equilibrium(eq, d_rb + (size_t)mg * C, d_u, &prm.blue, R, C, 0, nullptr), "lbm_cg_equilibrium");
    check(lbm_lattice_copy_rows(pre_b, &g, 0, eq, &dense, 0, R, nullptr), "lbm_lattice_copy_rows");
    check(lbm_stream_sync(nullptr), "sync");
    lbm_free(eq);
  }
  check(lbm_cg_collide(*post_r, *post_b, pre_r, pre_b, d_rr, d_rb, d_u, &g, &bc, &prm, nullptr, nullptr, nullptr), "lbm_cg_collide");
  check(lbm_stream_sync(nullptr), "sync");
  for (double* p : {d_rr, d_rb, d_u, pre_r, pre_b}) lbm_free(p);
}

int run_rank(const Args& a, int rank, int world, int local_rank) {
  check(lbm_set_device(a.one_gpu ? 0 : local_rank), "lbm_set_device");
  const int R = a.rows, C = a.cols, Rg = R * world, G = 3;
  const lbm_cg_params prm = rt_params();
  const lbm_geom g = slab_geom(R, C, G);
  lbm_bc bc;
  lbm_cg_default_bc(&bc);
  if (rank > 0) bc.row_lo = LBM_EDGE_HALO;
  if (rank < world - 1) bc.row_hi = LBM_EDGE_HALO;

  unsigned char id[128];
  share_unique_id(id, rank, world, a.id_file);
  lbm_ring* ring = nullptr;
  check(lbm_ring_create(&ring, id, rank, world, &g, /*periodic=*/0), "lbm_ring_create");

  const size_t plane = plane_of(g);
  double* lat[2][2];
  make_slab(R, C, rank * R, Rg, g, bc, prm, &lat[0][0], &lat[0][1]);
  for (int k = 0; k < 2; ++k) {
    check(lbm_malloc((void**)&lat[1][k], 9 * plane * 8), "lbm_malloc");
    check(lbm_memset(lat[1][k], 0, 9 * plane * 8, nullptr), "memset");
  }
  check(lbm_ring_exchange2(ring, lat[0][0], lat[0][1], nullptr), "lbm_ring_exchange2");
  check(lbm_ring_join(ring, nullptr), "lbm_ring_join");

  int cur = 0;
  auto step = [&]() {
    check(lbm_ring_cg_step(ring, lat[cur ^ 1][0], lat[cur ^ 1][1], lat[cur][0], lat[cur][1], nullptr, &prm,
                           a.edge_rows, nullptr), "lbm_ring_cg_step");
    cur ^= 1;
  };
  for (int i = 0; i < a.warmup; ++i) step();
  check(lbm_stream_sync(nullptr), "sync");
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < a.steps; ++i) step();
  check(lbm_stream_sync(nullptr), "sync");
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (const int failed = ring_failed(ring, "slab_ring_rt", rank)) return failed;
  const double tmax = max_time_over_ranks(sec, rank, world, a.id_file);

  int bad = 0;
  if (a.check) {
    std::vector<double> own((size_t)18 * R * C);
    double* dense_dev = nullptr;
    check(lbm_malloc((void**)&dense_dev, (size_t)9 * R * C * 8), "lbm_malloc");
    for (int k = 0; k < 2; ++k) owned_rows_to_host(&own[(size_t)k * 9 * R * C], lat[cur][k], g, dense_dev);
    lbm_free(dense_dev);
    write_file_atomic(a.id_file + ".f" + std::to_string(rank), own.data(), own.size() * 8);
    if (rank == 0) {
      const lbm_geom gw = slab_geom(Rg, C, 0, /*dense=*/true);
      lbm_bc bw;
      lbm_cg_default_bc(&bw);
      double *p[2], *q2[2];
      make_slab(Rg, C, 0, Rg, gw, bw, prm, &p[0], &p[1]);
      const size_t n = (size_t)9 * Rg * C;
      for (int k = 0; k < 2; ++k) check(lbm_malloc((void**)&q2[k], n * 8), "lbm_malloc");
      for (int t = 0; t < a.warmup + a.steps; ++t) {
        check(lbm_cg_step_fused(q2[0], q2[1], p[0], p[1], &gw, &bw, &prm, 0, Rg, nullptr, nullptr, nullptr,
                                nullptr, nullptr, nullptr), "lbm_cg_step_fused");
        std::swap(p[0], q2[0]);
        std::swap(p[1], q2[1]);
      }
      std::vector<double> want(n);
      for (int k = 0; k < 2; ++k) {
        check(lbm_memcpy_d2h(want.data(), p[k], n * 8, nullptr), "d2h");
        check(lbm_stream_sync(nullptr), "sync");
        for (int r = 0; r < world; ++r) {
          wait_file(a.id_file + ".f" + std::to_string(r), own.data(), own.size() * 8);
          for (int q = 0; q < 9; ++q)
            if (std::memcmp(&own[((size_t)k * 9 + q) * R * C], &want[(size_t)q * Rg * C + (size_t)r * R * C],
                            (size_t)R * C * 8) != 0)
              ++bad;
        }
      }
      for (int k = 0; k < 2; ++k) {
        lbm_free(p[k]);
        lbm_free(q2[k]);
      }
    }
  }
  if (rank == 0) {
    std::printf("{\"driver\": \"slab_ring_rt\", \"n_gpus\": %d, \"rows_per_gpu\": %d, \"cols\": %d, \"steps\": %d, "
                "\"ms_per_step\": %.4f, \"mlups\": %.1f, \"transport\": \"%s (C++ ring)\"%s}\n",
                world, R, C, a.steps, 1e3 * tmax / a.steps, (double)Rg * C * a.steps / tmax / 1e6,
                lbm_ring_transport(ring) == LBM_RING_IPC ? "peer-mapped windows" : "rccl send/recv",
                a.check ? (bad ? ", \"check\": \"MISMATCH\"" : ", \"check\": \"bitwise equal to one block\"") : "");
    std::fflush(stdout);
  }
  lbm_ring_destroy(ring);
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 2; ++k) lbm_free(lat[b][k]);
  return bad ? 3 : 0;
}

// --emulate N: every slab of the chain in turn on ONE GPU; the messages of lbm_ring_cg_step (LBM_HALO_TWO_PHASE: 21 rows per
// colour and side) travel by device copies.  Same kernels on the same row ranges as a rank of the ring runs.
int run_emulated(const Args& a, int N) {
  check(lbm_set_device(0), "lbm_set_device");
  const int R = a.rows, C = a.cols, Rg = R * N, G = 3, E = a.edge_rows < G ? G : a.edge_rows;
  if (2 * E >= R) throw std::runtime_error("--edge-rows too large for these slabs");
  const lbm_cg_params prm = rt_params();
  const lbm_geom g = slab_geom(R, C, G);
  const size_t plane = plane_of(g), msg = (size_t)lbm_halo_rows(LBM_HALO_TWO_PHASE) * C;
  struct Slab {
    lbm_bc bc;
    double* lat[2][2];   // [buffer][colour]
    double* buf[2][2];   // [side][send / recv], both colours back to back
    double ms = 0;
  };
  std::vector<Slab> S(N);
  for (int r = 0; r < N; ++r) {
    lbm_cg_default_bc(&S[r].bc);
    if (r > 0) S[r].bc.row_lo = LBM_EDGE_HALO;
    if (r < N - 1) S[r].bc.row_hi = LBM_EDGE_HALO;
    make_slab(R, C, r * R, Rg, g, S[r].bc, prm, &S[r].lat[0][0], &S[r].lat[0][1]);
    for (int k = 0; k < 2; ++k) {
      check(lbm_malloc((void**)&S[r].lat[1][k], 9 * plane * 8), "lbm_malloc");
      check(lbm_memset(S[r].lat[1][k], 0, 9 * plane * 8, nullptr), "memset");
    }
    for (int side = 0; side < 2; ++side)
      for (int k = 0; k < 2; ++k) check(lbm_malloc((void**)&S[r].buf[side][k], 2 * msg * 8), "lbm_malloc");
  }
  // the ring's two streams: edge (frame of the slab, pack, exchange) beside main (the inner rectangle)
  lbm_stream_t edge = nullptr;
  check(lbm_stream_create(&edge), "lbm_stream_create");
  void *ev_fork = nullptr, *ev_join = nullptr;
  check(lbm_event_create(&ev_fork), "lbm_event_create");
  check(lbm_event_create(&ev_join), "lbm_event_create");
  auto pack = [&](int r, int cur, lbm_stream_t st) {
    for (int k = 0; k < 2; ++k) {
      if (r > 0) check(lbm_halo_pack(S[r].buf[0][0] + k * msg, S[r].lat[cur][k], &g, LBM_HALO_TWO_PHASE, 0, st), "lbm_halo_pack");
      if (r < N - 1) check(lbm_halo_pack(S[r].buf[1][0] + k * msg, S[r].lat[cur][k], &g, LBM_HALO_TWO_PHASE, 1, st), "lbm_halo_pack");
    }
  };
  auto deliver = [&]() {
    for (int r = 0; r + 1 < N; ++r) {
      check(lbm_memcpy_d2d(S[r + 1].buf[0][1], S[r].buf[1][0], 2 * msg * 8, nullptr), "d2d");
      check(lbm_memcpy_d2d(S[r].buf[1][1], S[r + 1].buf[0][0], 2 * msg * 8, nullptr), "d2d");
    }
  };
  auto unpack = [&](int r, int cur) {
    for (int k = 0; k < 2; ++k) {
      if (r > 0) check(lbm_halo_unpack(S[r].lat[cur][k], S[r].buf[0][1] + k * msg, &g, LBM_HALO_TWO_PHASE, 0, nullptr), "lbm_halo_unpack");
      if (r < N - 1) check(lbm_halo_unpack(S[r].lat[cur][k], S[r].buf[1][1] + k * msg, &g, LBM_HALO_TWO_PHASE, 1, nullptr), "lbm_halo_unpack");
    }
  };
  int cur = 0;
  for (int r = 0; r < N; ++r) pack(r, cur, nullptr);
  deliver();
  for (int r = 0; r < N; ++r) unpack(r, cur);
  std::vector<void*> ev(2 * N, nullptr);
  for (auto& e : ev) check(lbm_event_create(&e), "lbm_event_create");
  for (int i = 0; i < a.warmup + a.steps; ++i) {
    for (int r = 0; r < N; ++r) {  // a slab's step as lbm_ring_cg_step enqueues it
      check(lbm_event_record(ev[2 * r], nullptr), "event");
      auto rows = [&](int r0, int r1) {
        check(lbm_cg_step_fused(S[r].lat[cur ^ 1][0], S[r].lat[cur ^ 1][1], S[r].lat[cur][0], S[r].lat[cur][1], &g, &S[r].bc, &prm, r0, r1,
                                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "lbm_cg_step_fused");
      };
      auto part = [&](int which, lbm_stream_t st) {
        check(lbm_cg_step_fused_part(S[r].lat[cur ^ 1][0], S[r].lat[cur ^ 1][1], S[r].lat[cur][0], S[r].lat[cur][1], &g, &S[r].bc, &prm, which, E,
                                     nullptr, nullptr, nullptr, nullptr, nullptr, st), "lbm_cg_step_fused_part");
      };
      if (N > 1 && a.parts) {
        // frame (wall / copy columns + the first and last edge rows) and the messages on the edge stream, the inner
        // rectangle on the main stream beside them; the step ends when both have
        check(lbm_event_record(ev_fork, nullptr), "event");
        check(lbm_stream_wait_event(edge, ev_fork), "wait");
        part(LBM_CG_PART_FRAME, edge);
        part(LBM_CG_PART_INNER, nullptr);
        pack(r, cur ^ 1, edge);
        check(lbm_event_record(ev_join, edge), "event");
        check(lbm_stream_wait_event(nullptr, ev_join), "wait");
      } else if (N > 1) {  // round 3: three row ranges, each a frame + inner pair
        rows(0, E);
        rows(R - E, R);
        rows(E, R - E);
        pack(r, cur ^ 1, nullptr);
      } else {
        rows(0, R);
        pack(r, cur ^ 1, nullptr);
      }
      check(lbm_event_record(ev[2 * r + 1], nullptr), "event");
    }
    deliver();
    for (int r = 0; r < N; ++r) unpack(r, cur ^ 1);
    for (int r = 0; r < N; ++r) {
      float m = 0;
      check(lbm_event_elapsed_ms(&m, ev[2 * r], ev[2 * r + 1]), "elapsed");
      if (i >= a.warmup) S[r].ms += m;
    }
    cur ^= 1;
  }
  for (auto& e : ev) lbm_event_destroy(e);
  lbm_event_destroy(ev_fork);
  lbm_event_destroy(ev_join);
  check(lbm_stream_sync(edge), "sync");
  lbm_stream_destroy(edge);
  int bad = 0;
  if (a.check) {
    const lbm_geom gw = slab_geom(Rg, C, 0, /*dense=*/true);
    lbm_bc bw;
    lbm_cg_default_bc(&bw);
    double *p[2], *q2[2];
    make_slab(Rg, C, 0, Rg, gw, bw, prm, &p[0], &p[1]);
    const size_t n = (size_t)9 * Rg * C;
    for (int k = 0; k < 2; ++k) check(lbm_malloc((void**)&q2[k], n * 8), "lbm_malloc");
    for (int t = 0; t < a.warmup + a.steps; ++t) {
      check(lbm_cg_step_fused(q2[0], q2[1], p[0], p[1], &gw, &bw, &prm, 0, Rg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "lbm_cg_step_fused");
      std::swap(p[0], q2[0]);
      std::swap(p[1], q2[1]);
    }
    std::vector<double> want((size_t)R * C), got((size_t)9 * R * C);
    double* dense_dev = nullptr;
    check(lbm_malloc((void**)&dense_dev, (size_t)9 * R * C * 8), "lbm_malloc");
    for (int k = 0; k < 2; ++k)
      for (int r = 0; r < N; ++r) {
        owned_rows_to_host(got.data(), S[r].lat[cur][k], g, dense_dev);
        for (int q = 0; q < 9; ++q) {
          check(lbm_memcpy_d2h(want.data(), p[k] + (size_t)q * Rg * C + (size_t)r * R * C, want.size() * 8, nullptr), "d2h");
          check(lbm_stream_sync(nullptr), "sync");
          if (std::memcmp(want.data(), &got[(size_t)q * R * C], want.size() * 8) != 0) ++bad;
        }
      }
    lbm_free(dense_dev);
    for (int k = 0; k < 2; ++k) {
      lbm_free(p[k]);
      lbm_free(q2[k]);
    }
  }
  double slowest = 0;
  for (int r = 0; r < N; ++r) slowest = std::max(slowest, S[r].ms / a.steps);
  std::printf("{\"driver\": \"slab_ring_rt\", \"mode\": \"emulated chain on one GPU\", \"slabs\": %d, \"rows_per_slab\": %d, \"cols\": %d, "
              "\"global_rows\": %d, \"steps\": %d, \"edge_rows\": %d, \"message_rows_per_colour_and_side\": %d, \"slowest_slab_ms_per_step\": %.4f, "
              "\"chain_mlups_at_the_slowest_slabs_pace\": %.1f, \"per_slab\": [",
              N, R, C, Rg, a.steps, E, lbm_halo_rows(LBM_HALO_TWO_PHASE), slowest, (double)Rg * C / slowest / 1e3);
  for (int r = 0; r < N; ++r)
    std::printf("%s{\"slab\": %d, \"ms_per_step\": %.4f, \"mlups\": %.1f}", r ? ", " : "", r, S[r].ms / a.steps, (double)R * C / (S[r].ms / a.steps) / 1e3);
  std::printf("]%s}\n", a.check ? (bad ? ", \"check\": \"MISMATCH\"" : ", \"check\": \"bitwise equal to one block\"") : "");
  std::fflush(stdout);
  for (auto& sb : S)
    for (int x = 0; x < 2; ++x)
      for (int k = 0; k < 2; ++k) {
        lbm_free(sb.lat[x][k]);
        lbm_free(sb.buf[x][k]);
      }
  return bad ? 3 : 0;
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  a.rows = std::atoi(arg_value(argc, argv, "--rows", "2048").c_str());
  a.cols = std::atoi(arg_value(argc, argv, "--cols", "2048").c_str());
  a.steps = std::atoi(arg_value(argc, argv, "--steps", "50").c_str());
  a.warmup = std::atoi(arg_value(argc, argv, "--warmup", "5").c_str());
  a.edge_rows = std::atoi(arg_value(argc, argv, "--edge-rows", "16").c_str());
  a.parts = std::atoi(arg_value(argc, argv, "--parts", "1").c_str());  // 0: the emulated chain with round 3's three row ranges per step
  a.check = std::atoi(arg_value(argc, argv, "--check", "0").c_str());
  a.emulate = std::atoi(arg_value(argc, argv, "--emulate", "0").c_str());
  a.one_gpu = std::atoi(arg_value(argc, argv, "--one-gpu", "0").c_str());
  const std::string transport = arg_value(argc, argv, "--transport", "");
  if (!transport.empty()) setenv("LBM_RING_TRANSPORT", transport.c_str(), 1);  // lbm_ring_unique_id / lbm_ring_create follow it
  a.id_file = arg_value(argc, argv, "--id-file", "/tmp/lbm_ring_id." + std::to_string((long)getpid()));
  const int spawn = std::atoi(arg_value(argc, argv, "--spawn", "0").c_str());
  try {
    if (a.emulate > 0) return run_emulated(a, a.emulate);
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
    std::fprintf(stderr, "slab_ring_rt: %s\n", e.what());
    return 1;
  }
}
