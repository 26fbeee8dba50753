// Flow past an immersed-boundary cylinder (BASELINE config 5, test/cylinder_test.cpp) slab-decomposed
// along the streamwise direction r over the GPUs of one node: a CHAIN of slabs -- the global rows 0
// and Rg-1 are the anti-bounce-back velocity inlet / outlet (:135-154), the columns are specular
// walls (:157-163).  The cylinder sits where SURVEY 8(d) puts it: centre at (rows/4, cols/2) -- on 8
// slabs of 2048 rows its ROI STRADDLES the seam at row 4096.  Time advances in blocks of D = 5 steps
// (lbm_ring_bgk_block_ibm): slabs without boundary rows run the 5-step window with one exchange of
// complete ghost rows per block, overlapped with their interior; the one or two slabs that own rows of
// the band around the ROI run the band's forced single steps in a compact replica beside their far
// rows and, across a seam inside the band, swap its outer rows instead of the halo.
// C++ host; one process per GPU.
//
//   slab_ring_cylinder --spawn N [--rows R_per_gpu] [--cols C] [--steps K] [--warmup W]
//                      [--diameter D] [--centre-row r] [--depth 5] [--edge-rows E] [--check 1]
//   slab_ring_cylinder --emulate N ...   ONE process / one GPU playing all N slabs of the chain in turn
//                      (messages moved by device copies; per-slab time per block reported; --check 1
//                      compares with the single-block run bit for bit -- the 8 x 2048 x 4096 layout of
//                      BASELINE config 5 fits one MI355X several times over)
//   slab heights (all modes): the domain has N x --rows rows; by default the LIBRARY cuts it (lbm_slab_ibm_plan_rows: the
//                      slab that holds the forced band pays its chain on top of its rows, so it gets the band and little
//                      else, and all slabs finish a block together); --uniform 1: N equal slabs (the cylinder then
//                      straddles a seam of the BASELINE layout); --slab-rows r0,r1,...: heights by hand;
//                      --costs far_us_per_row,owner_us,owner_us_per_row: the planner's cost model instead of its table
//                      --form reference|reassociated: the collision's operation order (default: the library's = reassociated; reference = bitwise to the oracle)
//   RANK=i WORLD_SIZE=N LOCAL_RANK=i slab_ring_cylinder --id-file /tmp/x ...
//
// tau = 0.55 (parameters.toml), u_in = 0.04 (SURVEY 8d allows a smaller u for the benchmark), markers
// on a circle of the given diameter with spacing ~1, generated deterministically.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "ring_common.hpp"

namespace {

struct Args {
  int rows = 2048, cols = 4096, steps = 50, warmup = 5, edge_rows = 32, check = 0, diameter = 300;
  int depth = 5, centre_row = -1, emulate = 0, uniform = 0;
  int form = LBM_FORM_DEFAULT;  // --form reference|reassociated: the collision's operation order (lbm_bgk_params.form)
  std::string id_file;
  std::string slab_rows;  // comma-separated slab heights (default: planned by the library)
  std::string costs;      // far_us_per_row,owner_us,owner_us_per_row for the planner
};
const double kTau = 0.55, kUin = 0.04, kGuoA = 1.0 / 3.0, kGuoB = 1.0 / 9.0;  // cylinder_test.cpp:66-67

void cylinder_markers(int Rg, int C, int centre_row, int diameter, std::vector<double>& x, std::vector<double>& y) {
  // centre at (rows / 4, cols / 2) unless told otherwise
  const double cx = (centre_row >= 0 ? centre_row : Rg / 4) + 0.37, cy = C / 2.0 + 0.21, rad = diameter / 2.0;
  const int m = (int)std::lround(3.14159265358979323846 * diameter);
  x.resize(m);
  y.resize(m);
  for (int i = 0; i < m; ++i) {
    const double t = 2.0 * 3.14159265358979323846 * i / m;
    x[i] = cx + rad * std::cos(t);
    y[i] = cy + rad * std::sin(t);
  }
}

std::vector<double> parse_list(const std::string& s) {
  std::vector<double> v;
  size_t pos = 0;
  while (pos < s.size()) {
    const size_t end = s.find(',', pos);
    v.push_back(std::atof(s.substr(pos, end == std::string::npos ? std::string::npos : end - pos).c_str()));
    pos = end == std::string::npos ? s.size() : end + 1;
  }
  return v;
}

// heights of the N slabs of a domain of N x --rows rows: by hand, equal, or planned by the library (the default)
std::vector<int> slab_heights(const Args& a, int N, const std::vector<double>& mx, std::string& how, double& predicted_us) {
  const int Rg = a.rows * N;
  std::vector<int> rows(N, a.rows);
  predicted_us = 0;
  if (!a.slab_rows.empty()) {
    const std::vector<double> v = parse_list(a.slab_rows);
    if ((int)v.size() != N) throw std::runtime_error("--slab-rows needs one height per slab");
    for (int r = 0; r < N; ++r) {
      rows[r] = (int)v[r];
      if (rows[r] <= 0) throw std::runtime_error("--slab-rows needs one positive height per slab");
    }
    how = "given";
  } else if (a.uniform || N == 1) {
    how = "uniform";
  } else {
    const std::vector<double> c = parse_list(a.costs);
    if (!c.empty() && c.size() != 3) throw std::runtime_error("--costs needs far_us_per_row,owner_us,owner_us_per_row");
    check(lbm_slab_ibm_plan_rows(rows.data(), N, Rg, a.cols, a.depth, mx.data(), (int)mx.size(), c.empty() ? nullptr : c.data(), &predicted_us),
          "lbm_slab_ibm_plan_rows");
    how = "planned (lbm_slab_ibm_plan_rows)";
  }
  return rows;
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
  check(lbm_set_device(std::getenv("LBM_ONE_GPU") ? 0 : local_rank), "lbm_set_device");
  const bool rehearse = false;
  const int vr = rank, vw = world;
  const int D = a.depth, C = a.cols, Rg = a.rows * vw, G = D;
  lbm_bgk_params prm{};
  prm.omega = 1.0 / kTau;
  prm.incompressible = 0;
  prm.delta_form = 1;  // :123-125
  prm.form = a.form;
  lbm_bc bc = global_bc();

  std::vector<double> mx, my;
  cylinder_markers(Rg, C, a.centre_row, a.diameter, mx, my);
  std::string how;
  double predicted_us = 0;
  const std::vector<int> heights = slab_heights(a, vw, mx, how, predicted_us);  // the same on every rank
  std::vector<int> row0s(vw, 0);
  for (int r = 1; r < vw; ++r) row0s[r] = row0s[r - 1] + heights[r - 1];
  if (row0s[vw - 1] + heights[vw - 1] != Rg) throw std::runtime_error("slab heights do not add up to the domain");
  const int R = heights[vr];
  lbm_geom g{R, C, G, 0, 0};
  lbm_slab_ibm* sl = nullptr;
  check(lbm_slab_ibm_create(&sl, &g, row0s[vr], Rg, &bc, &prm, D, mx.data(), my.data(), (int)mx.size(), 5, kGuoA, kGuoB),
        "lbm_slab_ibm_create");
  int owner = 0, sp = 0, sn = 0, b0 = 0, b1 = 0;
  check(lbm_slab_ibm_info(sl, &owner, &sp, &sn, &b0, &b1), "lbm_slab_ibm_info");

  unsigned char id[128];
  share_unique_id(id, rank, world, a.id_file);
  lbm_ring* ring = nullptr;
  check(lbm_ring_create(&ring, id, rank, world, &g, /*periodic=*/rehearse ? 1 : 0), "lbm_ring_create");

  const size_t n = (size_t)R * C, plane = (size_t)(R + 2 * G) * C;
  double* lat[2];
  double* pre = uniform_inflow(R, C, G);
  check(lbm_malloc((void**)&lat[0], 9 * plane * 8), "lbm_malloc");
  check(lbm_malloc((void**)&lat[1], 9 * plane * 8), "lbm_malloc");
  for (double* p : {lat[0], lat[1]}) check(lbm_memset(p, 0, 9 * plane * 8, nullptr), "memset");
  // first iteration (:103-127): moments, collision in delta form, forcing + source on the band's owners
  check(lbm_ring_ibm_start(ring, sl, lat[0], pre, nullptr), "lbm_ring_ibm_start");
  check(lbm_stream_sync(nullptr), "sync");
  lbm_free(pre);

  int cur = 0;
  auto block = [&]() {
    check(lbm_ring_bgk_block_ibm(ring, sl, lat[cur ^ 1], lat[cur], a.edge_rows, nullptr), "lbm_ring_bgk_block_ibm");
    cur ^= 1;
  };
  const int wb = (a.warmup + D - 1) / D, nb = (a.steps + D - 1) / D, steps = nb * D;
  for (int i = 0; i < wb; ++i) block();
  check(lbm_stream_sync(nullptr), "sync");
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < nb; ++i) block();
  check(lbm_stream_sync(nullptr), "sync");
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (const int failed = ring_failed(ring, "slab_ring_cylinder", rank)) return failed;
  const double tmax = max_time_over_ranks(sec, rank, world, a.id_file);
  double Fs[2] = {0, 0};
  if (owner) {
    check(lbm_slab_ibm_surface_force(sl, Fs, nullptr), "lbm_slab_ibm_surface_force");
    write_file_atomic(a.id_file + ".g" + std::to_string(rank), Fs, sizeof Fs);
  }
  int first_owner = -1;  // every rank can tell who owns band rows: valid rows [b0 + D, b1 - D)
  for (int r = 0; r < vw && first_owner < 0; ++r)
    if (b0 + D < row0s[r] + heights[r] && b1 - D > row0s[r]) first_owner = r;

  int bad = 0;
  if (a.check && !rehearse) {
    std::vector<double> h(9 * plane), own((size_t)9 * n);
    check(lbm_memcpy_d2h(h.data(), lat[cur], h.size() * 8, nullptr), "d2h");
    check(lbm_stream_sync(nullptr), "sync");
    for (int q = 0; q < 9; ++q) std::memcpy(&own[(size_t)q * n], &h[q * plane + (size_t)G * C], n * 8);
    write_file_atomic(a.id_file + ".f" + std::to_string(rank), own.data(), own.size() * 8);
    if (rank == 0) {
      // the same run as ONE block through the solver context (collide-first, forced blocks)
      lbm_geom gw{Rg, C, 0, 0, 0};
      lbm_solver* sv = nullptr;
      check(lbm_solver_create(&sv, LBM_MODEL_BGK, &gw, &bc, &prm, nullptr), "lbm_solver_create");
      lbm_ibm* ibw = nullptr;
      check(lbm_ibm_create(&ibw, mx.data(), my.data(), (int)mx.size(), 5, Rg, C), "lbm_ibm_create");
      check(lbm_solver_attach_ibm(sv, ibw, kGuoA, kGuoB), "lbm_solver_attach_ibm");
      double* prew = uniform_inflow(Rg, C, 0);
      check(lbm_solver_set_f_soa_dev(sv, prew), "lbm_solver_set_f_soa_dev");
      check(lbm_solver_step(sv, 1 + (wb + nb) * D, 0), "lbm_solver_step");
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
          const size_t nr = (size_t)heights[r] * C;
          std::vector<double> theirs((size_t)9 * nr);
          wait_file(a.id_file + ".f" + std::to_string(r), theirs.data(), theirs.size() * 8);
          if (std::memcmp(&theirs[(size_t)q * nr], &want[(size_t)row0s[r] * C], nr * 8) != 0) ++bad;
        }
      }
      double Fw[2], Fo[2];
      check(lbm_ibm_surface_force(ibw, Fw, nullptr), "lbm_ibm_surface_force");
      wait_file(a.id_file + ".g" + std::to_string(first_owner), Fo, sizeof Fo);
      if (std::memcmp(Fw, Fo, sizeof Fw) != 0) ++bad;
      lbm_solver_destroy(sv);
      lbm_ibm_destroy(ibw);
      lbm_free(prew);
    }
  }
  if (rank == 0) {
    double Fo[2] = {0, 0};
    if (owner) std::memcpy(Fo, Fs, sizeof Fo);
    else if (!rehearse && first_owner >= 0 && first_owner < world) wait_file(a.id_file + ".g" + std::to_string(first_owner), Fo, sizeof Fo);
    std::printf("{\"driver\": \"slab_ring_cylinder\", \"n_gpus\": %d, \"rows_of_rank_0\": %d, \"cols\": %d, \"global_rows\": %d, "
                "\"markers\": %d, \"band_rows\": [%d, %d], \"first_owner_rank\": %d, \"this_rank\": {\"rank\": %d, \"owner\": %d, "
                "\"straddle_prev\": %d, \"straddle_next\": %d}, \"steps_per_block\": %d, \"steps\": %d, \"ms_per_step\": %.4f, "
                "\"mlups_per_gpu\": %.1f, \"mlups\": %.1f, \"Fs\": [%.17g, %.17g], \"transport\": \"rccl send/recv (C++ ring)%s\"%s}\n",
                world, R, C, Rg, (int)mx.size(), b0, b1, first_owner, vr, owner, sp, sn, D, steps, 1e3 * tmax / steps,
                (double)Rg / world * C * steps / tmax / 1e6, (double)Rg * C * steps / tmax / 1e6, Fo[0], Fo[1],
                rehearse ? ", one GPU rehearsing one slab with self send/recv" : "",
                a.check && !rehearse ? (bad ? ", \"check\": \"MISMATCH\"" : ", \"check\": \"bitwise equal to one block\"") : "");
    std::fflush(stdout);
  }
  lbm_ring_destroy(ring);
  lbm_slab_ibm_destroy(sl);
  for (double* p : {lat[0], lat[1]}) lbm_free(p);
  return bad ? 3 : 0;
}

// --emulate N: every slab of the chain in turn on ONE GPU, messages by device copies
int run_emulated(const Args& a, int N) {
  check(lbm_set_device(0), "lbm_set_device");
  const int D = a.depth, C = a.cols, G = D;
  // slab heights: uniform, or as listed (a load-balanced decomposition gives the slabs that share the forced band
  // fewer rows: their block carries the band's chain on top of their far rows)
  std::vector<int> row0(N, 0);
  std::vector<double> mx, my;
  {
    int total = 0;
    if (!a.slab_rows.empty()) for (double v : parse_list(a.slab_rows)) total += (int)v;
    else total = a.rows * N;
    cylinder_markers(total, C, a.centre_row, a.diameter, mx, my);
  }
  std::string how;
  double predicted_us = 0;
  std::vector<int> rows = slab_heights(a, N, mx, how, predicted_us);
  int Rg = 0;
  for (int r = 0; r < N; ++r) row0[r] = Rg, Rg += rows[r];
  lbm_bgk_params prm{};
  prm.omega = 1.0 / kTau;
  prm.delta_form = 1;
  prm.form = a.form;
  lbm_bc bc = global_bc();
  struct Slab {
    int R = 0, row0 = 0;
    size_t n = 0, plane = 0;
    lbm_geom g{};
    lbm_slab_ibm* sl = nullptr;
    double* lat[2] = {nullptr, nullptr};
    double* buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [side][send / recv], sized for the priming messages
    int owner = 0, sp = 0, sn = 0;
    double ms = 0;
  };
  std::vector<Slab> S(N);
  int b0 = 0, b1 = 0;
  for (int r = 0; r < N; ++r) {
    S[r].R = rows[r];
    S[r].row0 = row0[r];
    S[r].n = (size_t)rows[r] * C;
    S[r].plane = (size_t)(rows[r] + 2 * G) * C;
    S[r].g = lbm_geom{rows[r], C, G, 0, 0};
    check(lbm_slab_ibm_create(&S[r].sl, &S[r].g, S[r].row0, Rg, &bc, &prm, D, mx.data(), my.data(), (int)mx.size(), 5, kGuoA, kGuoB), "lbm_slab_ibm_create");
    check(lbm_slab_ibm_info(S[r].sl, &S[r].owner, &S[r].sp, &S[r].sn, &b0, &b1), "lbm_slab_ibm_info");
    for (int k = 0; k < 2; ++k) {
      check(lbm_malloc((void**)&S[r].lat[k], 9 * S[r].plane * 8), "lbm_malloc");
      check(lbm_memset(S[r].lat[k], 0, 9 * S[r].plane * 8, nullptr), "memset");
    }
    for (int side = 0; side < 2; ++side) {
      long long cs = 0, cr = 0;
      check(lbm_slab_ibm_prime_counts(S[r].sl, side, &cs, &cr), "lbm_slab_ibm_prime_counts");
      const long long m = lbm_slab_ibm_msg_doubles(S[r].sl);
      check(lbm_malloc((void**)&S[r].buf[side][0], (size_t)std::max(cs, m) * 8), "lbm_malloc");
      check(lbm_malloc((void**)&S[r].buf[side][1], (size_t)std::max(cr, m) * 8), "lbm_malloc");
    }
  }
  auto deliver = [&](bool priming) {  // next's recv_prev <- my send_next; my recv_next <- next's send_prev
    for (int r = 0; r + 1 < N; ++r) {
      long long s_dn = lbm_slab_ibm_msg_doubles(S[r].sl), s_up = s_dn, dummy = 0;
      if (priming) {
        check(lbm_slab_ibm_prime_counts(S[r].sl, 1, &s_dn, &dummy), "counts");
        check(lbm_slab_ibm_prime_counts(S[r + 1].sl, 0, &s_up, &dummy), "counts");
      }
      check(lbm_memcpy_d2d(S[r + 1].buf[0][1], S[r].buf[1][0], (size_t)s_dn * 8, nullptr), "d2d");
      check(lbm_memcpy_d2d(S[r].buf[1][1], S[r + 1].buf[0][0], (size_t)s_up * 8, nullptr), "d2d");
    }
  };
  {  // first iteration (:103-127) from the uniform inflow state
    std::vector<double*> pre(N);
    for (int r = 0; r < N; ++r) {
      pre[r] = uniform_inflow(S[r].R, C, G);
      check(lbm_slab_ibm_prime_pack(S[r].sl, pre[r], S[r].buf[0][0], S[r].buf[1][0], nullptr), "lbm_slab_ibm_prime_pack");
    }
    deliver(true);
    for (int r = 0; r < N; ++r) {
      check(lbm_slab_ibm_start_finish(S[r].sl, S[r].lat[0], pre[r], S[r].buf[0][1], S[r].buf[1][1], nullptr), "lbm_slab_ibm_start_finish");
      check(lbm_stream_sync(nullptr), "sync");
      lbm_free(pre[r]);
    }
  }
  int cur = 0;
  const int wb = (a.warmup + D - 1) / D, nb = (a.steps + D - 1) / D, steps = nb * D;
  // Per-slab time of a block = HIP events on the caller's stream around its compute and finish calls (the helper
  // streams inside are joined before the call returns its stream).  All slabs of a block are enqueued back to back and
  // read at the end of the block: each slab runs alone on the GPU, and while it does the host is already enqueuing
  // the next one -- what a rank of a running chain sees, not the host's launch latency of a cold start per block.
  std::vector<void*> ev(4 * N, nullptr);
  for (auto& e : ev) check(lbm_event_create(&e), "lbm_event_create");
  for (int i = 0; i < wb + nb; ++i) {
    for (int r = 0; r < N; ++r) {
      check(lbm_event_record(ev[4 * r], nullptr), "event");
      check(lbm_slab_ibm_block_compute(S[r].sl, S[r].lat[cur ^ 1], S[r].lat[cur], S[r].buf[0][0], S[r].buf[1][0], nullptr), "lbm_slab_ibm_block_compute");
      check(lbm_event_record(ev[4 * r + 1], nullptr), "event");
    }
    deliver(false);
    for (int r = 0; r < N; ++r) {
      check(lbm_event_record(ev[4 * r + 2], nullptr), "event");
      check(lbm_slab_ibm_block_finish(S[r].sl, S[r].lat[cur ^ 1], S[r].buf[0][1], S[r].buf[1][1], nullptr), "lbm_slab_ibm_block_finish");
      check(lbm_event_record(ev[4 * r + 3], nullptr), "event");
    }
    for (int r = 0; r < N; ++r) {
      float m0 = 0, m1 = 0;
      check(lbm_event_elapsed_ms(&m0, ev[4 * r], ev[4 * r + 1]), "elapsed");
      check(lbm_event_elapsed_ms(&m1, ev[4 * r + 2], ev[4 * r + 3]), "elapsed");
      if (i >= wb) S[r].ms += m0 + m1;
    }
    cur ^= 1;
  }
  for (auto& e : ev) lbm_event_destroy(e);
  double Fs[2] = {0, 0};
  int first_owner = -1, bad = 0;
  for (int r = 0; r < N; ++r)
    if (S[r].owner) {
      double F[2];
      check(lbm_slab_ibm_surface_force(S[r].sl, F, nullptr), "lbm_slab_ibm_surface_force");
      if (first_owner < 0) first_owner = r, std::memcpy(Fs, F, sizeof F);
      else if (std::memcmp(F, Fs, sizeof F) != 0) ++bad;  // co-owners hold the same forcing
    }
  if (a.check) {
    lbm_geom gw{Rg, C, 0, 0, 0};
    lbm_solver* sv = nullptr;
    check(lbm_solver_create(&sv, LBM_MODEL_BGK, &gw, &bc, &prm, nullptr), "lbm_solver_create");
    lbm_ibm* ibw = nullptr;
    check(lbm_ibm_create(&ibw, mx.data(), my.data(), (int)mx.size(), 5, Rg, C), "lbm_ibm_create");
    check(lbm_solver_attach_ibm(sv, ibw, kGuoA, kGuoB), "lbm_solver_attach_ibm");
    double* prew = uniform_inflow(Rg, C, 0);
    check(lbm_solver_set_f_soa_dev(sv, prew), "lbm_solver_set_f_soa_dev");
    check(lbm_solver_step(sv, 1 + (wb + nb) * D, 0), "lbm_solver_step");
    double *cl = nullptr, *ol = nullptr;
    lbm_geom gg;
    check(lbm_solver_lattices(sv, &cl, &ol, &gg), "lbm_solver_lattices");
    const long long ps = gg.plane_stride ? gg.plane_stride : (long long)Rg * C;
    std::vector<double> want, got;
    for (int q = 0; q < 9; ++q)
      for (int r = 0; r < N; ++r) {
        const size_t n = S[r].n;
        want.resize(n);
        got.resize(n);
        check(lbm_memcpy_d2h(want.data(), cl + q * ps + (size_t)S[r].row0 * C, n * 8, nullptr), "d2h");
        check(lbm_memcpy_d2h(got.data(), S[r].lat[cur] + q * S[r].plane + (size_t)G * C, n * 8, nullptr), "d2h");
        check(lbm_solver_sync(sv), "sync");
        check(lbm_stream_sync(nullptr), "sync");
        if (std::memcmp(want.data(), got.data(), n * 8) != 0) ++bad;
      }
    double Fw[2];
    check(lbm_ibm_surface_force(ibw, Fw, nullptr), "lbm_ibm_surface_force");
    if (std::memcmp(Fw, Fs, sizeof Fw) != 0) ++bad;
    lbm_solver_destroy(sv);
    lbm_ibm_destroy(ibw);
    lbm_free(prew);
  }
  double slowest = 0;
  for (int r = 0; r < N; ++r) slowest = std::max(slowest, S[r].ms / nb);
  std::printf("{\"driver\": \"slab_ring_cylinder\", \"mode\": \"emulated chain on one GPU\", \"slabs\": %d, \"slab_heights\": \"%s\", "
              "\"planner_predicted_ms_per_block\": %.4f, "
              "\"cols\": %d, \"global_rows\": %d, \"markers\": %d, \"band_rows\": [%d, %d], \"steps_per_block\": %d, \"steps\": %d, "
              "\"slowest_slab_ms_per_block\": %.4f, \"chain_mlups_at_the_slowest_slabs_pace\": %.1f, \"per_slab\": [",
              N, how.c_str(), predicted_us / 1e3, C, Rg, (int)mx.size(), b0, b1, D, steps, slowest, (double)Rg * C * D / slowest / 1e3);
  for (int r = 0; r < N; ++r)
    std::printf("%s{\"slab\": %d, \"rows\": %d, \"owner\": %d, \"straddle_prev\": %d, \"straddle_next\": %d, \"ms_per_block\": %.4f, \"mlups\": %.1f}",
                r ? ", " : "", r, S[r].R, S[r].owner, S[r].sp, S[r].sn, S[r].ms / nb, (double)S[r].R * C * D / (S[r].ms / nb) / 1e3);
  std::printf("], \"Fs\": [%.17g, %.17g]%s}\n", Fs[0], Fs[1],
              a.check ? (bad ? ", \"check\": \"MISMATCH\"" : ", \"check\": \"bitwise equal to one block\"") : "");
  std::fflush(stdout);
  for (auto& sb : S) {
    lbm_slab_ibm_destroy(sb.sl);
    for (double* p : {sb.lat[0], sb.lat[1], sb.buf[0][0], sb.buf[0][1], sb.buf[1][0], sb.buf[1][1]}) lbm_free(p);
  }
  return bad ? 3 : 0;
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  a.rows = std::atoi(arg_value(argc, argv, "--rows", "2048").c_str());
  a.cols = std::atoi(arg_value(argc, argv, "--cols", "4096").c_str());
  a.steps = std::atoi(arg_value(argc, argv, "--steps", "50").c_str());
  a.warmup = std::atoi(arg_value(argc, argv, "--warmup", "5").c_str());
  a.edge_rows = std::atoi(arg_value(argc, argv, "--edge-rows", "32").c_str());
  a.depth = std::atoi(arg_value(argc, argv, "--depth", "5").c_str());
  a.centre_row = std::atoi(arg_value(argc, argv, "--centre-row", "-1").c_str());
  a.slab_rows = arg_value(argc, argv, "--slab-rows", "");
  a.costs = arg_value(argc, argv, "--costs", "");
  a.uniform = std::atoi(arg_value(argc, argv, "--uniform", "0").c_str());
  {
    const std::string f = arg_value(argc, argv, "--form", "");
    if (f == "reassociated") a.form = LBM_FORM_REASSOCIATED;
    else if (f == "reference") a.form = LBM_FORM_REFERENCE_ORDER;
    else if (!f.empty()) {
      std::fprintf(stderr, "--form reference|reassociated\n");
      return 2;
    }
    // the planner's built-in table is measured with the reference-order collision; the reassociated one -- the library's
    // default since round 4 -- walks a far row in 0.157 us instead of 0.180 (4096 columns, 5 steps) and its owner block
    // costs 326 us + 0.12 us per row: unless --costs says otherwise
    if (a.form != LBM_FORM_REFERENCE_ORDER && a.costs.empty()) a.costs = "0.157,326,0.12";
  }
  a.emulate = std::atoi(arg_value(argc, argv, "--emulate", "0").c_str());
  a.diameter = std::atoi(arg_value(argc, argv, "--diameter", "300").c_str());
  a.check = std::atoi(arg_value(argc, argv, "--check", "0").c_str());
  a.id_file = arg_value(argc, argv, "--id-file", "/tmp/lbm_ring_id." + std::to_string((long)getpid()));
  const int spawn = std::atoi(arg_value(argc, argv, "--spawn", "0").c_str());
  // --transport rccl|ipc: what carries the ring's messages (lbm_ring_unique_id / lbm_ring_create follow the environment);
  // --one-gpu 1: every rank on GPU 0 (with ipc: N real ranks on one device, which RCCL refuses)
  const std::string transport = arg_value(argc, argv, "--transport", "");
  if (!transport.empty()) setenv("LBM_RING_TRANSPORT", transport.c_str(), 1);
  if (std::atoi(arg_value(argc, argv, "--one-gpu", "0").c_str())) setenv("LBM_ONE_GPU", "1", 1);
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
    std::fprintf(stderr, "slab_ring_cylinder: %s\n", e.what());
    return 1;
  }
}
