// Restatement of test/decompose_domain_loop.cpp: four blocks -- A (L x L/4),
// B (L/4 x L/2), C (L x L/4), D (L/4 x L/2) -- closed into a square loop channel.  Every block is
// periodic inside solver::advect; no-slip walls and the block-to-block bindings then overwrite
// the populations that crossed a wall or a seam (:173-261).  A momentum source F = (3e-3, 0) with
// the (3, 9) coefficients acts on rows [L/4+5, L/4+55) of A (:63, :152-160).
//
// Engine mapping: per block and step one collide launch (with rho, u), one periodic stream launch;
// ALL ~70 slice assignments of a step are one lbm_links_apply gather (the table is built once, in
// the driver's order, later assignments winning at shared elements).
// --graph 1 captures the 14 launches of a step into a HIP graph and replays it (same results; measured
// 1.25 s vs 1.08 s of plain launches for 20000 steps at L = 512: the step is bound by the device-side
// dispatch of ~4 us kernels, not by host launch cost, so plain launches stay the default).
//   usage: decompose_domain_loop [--L 512] [--T 50000] [--graph 0] [--dump prefix]
// Dumps (raw f64) per block X in A..D: <prefix>-X-rho.f64 [R][C], <prefix>-X-u.f64 [R][C][2] = m_0, m_1
// as computed in the last iteration (what the reference's snapshot at t = T holds, before :114 adds
// F to A's u on the force rows), <prefix>-X-f.f64 [R][C][9] = adve_f.
#include <cmath>
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "common.hpp"

namespace {
const int kNone = 1 << 30;  // an open slice end
struct Block {
  int R, C;
  lbm_geom g;
  double *adve, *coll, *rho, *u;
};
int norm(int i, int n) { return i == kNone ? n : (i < 0 ? i + n : i); }

struct Topology {
  lbm_links* t;
  Block* b;
  // dst.adve[r0:r1, col, q] = src.coll[s0:.., scol, sq]
  void rows(int d, int r0, int r1, int col, int q, int s, int s0, int scol, int sq) {
    r0 = norm(r0, b[d].R);
    r1 = norm(r1, b[d].R);
    lbm::check(lbm_links_add(t, d, q, r0, norm(col, b[d].C), 1, 0, s, sq, norm(s0, b[s].R), norm(scol, b[s].C), 1, 0, r1 - r0));
  }
  // dst.adve[row, :, q] = dst.coll[row, :, sq]
  void wall_row(int d, int row, int q, int sq) {
    row = norm(row, b[d].R);
    lbm::check(lbm_links_add(t, d, q, row, 0, 0, 1, d, sq, row, 0, 0, 1, b[d].C));
  }
  void top_bottom(int d) {  // :173-180 and the same for B, C, D
    wall_row(d, 0, 8, 6);
    wall_row(d, 0, 1, 3);
    wall_row(d, 0, 5, 7);
    wall_row(d, -1, 7, 5);
    wall_row(d, -1, 3, 1);
    wall_row(d, -1, 6, 8);
  }
  void left(int d, int r0, int r1) {
    rows(d, r0, r1, 0, 2, d, r0, 0, 4);
    rows(d, r0, r1, 0, 5, d, r0, 0, 7);
    rows(d, r0, r1, 0, 6, d, r0, 0, 8);
  }
  void right(int d, int r0, int r1) {
    rows(d, r0, r1, -1, 4, d, r0, -1, 2);
    rows(d, r0, r1, -1, 7, d, r0, -1, 5);
    rows(d, r0, r1, -1, 8, d, r0, -1, 6);
  }
};
}  // namespace

int main(int argc, char** argv) {
  const int L = std::stoi(arg_value(argc, argv, "--L", "512"));
  const int T = std::stoi(arg_value(argc, argv, "--T", "50000"));
  const std::string dump = arg_value(argc, argv, "--dump", "");
  const bool use_graph = std::stoi(arg_value(argc, argv, "--graph", "0")) != 0;
  const int L2 = L / 2, L4 = L / 4;
  const double tau = std::sqrt(3.0 / 16.0) + 0.5, omega = 1.0 / tau;  // :44-45
  const double nu = (2.0 * tau - 1.0) / 6.0, u_max = 0.1;
  std::cout << "T=" << T << "\nomega=" << omega << "\nnu=" << nu << "\nRe=" << L4 * u_max / nu << std::endl;
  if (L % 4 || L4 + 55 > L) {
    std::cerr << "L must be a multiple of 4 with L/4 + 55 <= L (the force window, :63)\n";
    return 1;
  }
  if (lbm_device_count() < 1) {
    std::cerr << "no HIP device available\n";
    return 2;
  }
  try {
    enum { A, B, C, D };
    const int shape[4][2] = {{L, L4}, {L4, L2}, {L, L4}, {L4, L2}};
    Block b[4];
    lbm_geom geoms[4];
    for (int k = 0; k < 4; ++k) {
      b[k].R = shape[k][0];
      b[k].C = shape[k][1];
      b[k].g = geoms[k] = lbm_geom{b[k].R, b[k].C, 0, 0, 0};
      const size_t n = (size_t)b[k].R * b[k].C;
      for (double** p : {&b[k].adve, &b[k].coll}) lbm::check(lbm_malloc((void**)p, n * 9 * sizeof(double)));
      lbm::check(lbm_malloc((void**)&b[k].rho, n * sizeof(double)));
      lbm::check(lbm_malloc((void**)&b[k].u, n * 2 * sizeof(double)));
      // m_0 = 1, m_1 = 0 (:77-80); adve_f = equilibrium (:105-108)
      std::vector<double> ones(n, 1.0);
      lbm::check(lbm_memcpy_h2d(b[k].rho, ones.data(), n * sizeof(double), nullptr));
      lbm::check(lbm_memset(b[k].u, 0, n * 2 * sizeof(double), nullptr));
      lbm::check(lbm_equilibrium(b[k].adve, b[k].u, b[k].rho, b[k].R, b[k].C, nullptr));
    }
    lbm_links* links = nullptr;
    lbm::check(lbm_links_create(&links, 4, geoms));
    Topology tp{links, b};
    // walls, in the driver's order (:173-231)
    tp.top_bottom(A);
    tp.left(A, L4, -L4);
    tp.right(A, 1, -1);
    tp.top_bottom(B);
    tp.top_bottom(C);
    tp.left(C, 1, -1);
    tp.right(C, L4, -L4);
    tp.top_bottom(D);
    // bindings (:235-261): column seams with the +-1 row shift of the diagonal populations
    tp.rows(A, -L4, -1, 0, 6, B, 1, -1, 6);
    tp.rows(A, -L4, kNone, 0, 2, B, 0, -1, 2);
    tp.rows(A, -L4 + 1, kNone, 0, 5, B, 0, -1, 5);
    tp.rows(B, 1, kNone, -1, 8, A, -L4, 0, 8);
    tp.rows(B, 0, kNone, -1, 4, A, -L4, 0, 4);
    tp.rows(B, 0, -1, -1, 7, A, -L4 + 1, 0, 7);
    tp.rows(B, 0, -1, 0, 6, C, -L4 + 1, -1, 6);
    tp.rows(B, 0, kNone, 0, 2, C, -L4, -1, 2);
    tp.rows(B, 1, kNone, 0, 5, C, -L4, -1, 5);
    tp.rows(C, -L4, -1, -1, 7, B, 1, 0, 7);
    tp.rows(C, -L4, kNone, -1, 4, B, 0, 0, 4);
    tp.rows(C, -L4 + 1, kNone, -1, 8, B, 0, 0, 8);
    tp.rows(C, 0, L4 - 1, -1, 7, D, 1, 0, 7);
    tp.rows(C, 0, L4, -1, 4, D, 0, 0, 4);
    tp.rows(C, 1, L4, -1, 8, D, 0, 0, 8);
    tp.rows(D, 0, -1, 0, 6, C, 1, -1, 6);
    tp.rows(D, 0, kNone, 0, 2, C, 0, -1, 2);
    tp.rows(D, 1, kNone, 0, 5, C, 0, -1, 5);
    tp.rows(D, 0, -1, -1, 7, A, 1, 0, 7);
    tp.rows(D, 0, kNone, -1, 4, A, 0, 0, 4);
    tp.rows(D, 1, kNone, -1, 8, A, 0, 0, 8);
    tp.rows(A, 0, L4 - 1, 0, 6, D, 1, -1, 6);
    tp.rows(A, 0, L4, 0, 2, D, 0, -1, 2);
    tp.rows(A, 1, L4, 0, 5, D, 0, -1, 5);
    lbm::check(lbm_links_finalize(links));
    std::cout << "links=" << lbm_links_count(links) << std::endl;

    lbm_bgk_params pA{omega, 0, 1, 0, 0.0, 0.0, 0.0, 0.0, LBM_FORM_DEFAULT};  // A: adve + (-omega (adve - equi)) (:152-158)
    lbm_bgk_params pX{omega, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, LBM_FORM_DEFAULT};  // B, C, D: solver::collision (:161-163)
    double* adve[4] = {b[A].adve, b[B].adve, b[C].adve, b[D].adve};
    const double* coll[4] = {b[A].coll, b[B].coll, b[C].coll, b[D].coll};
    std::cout << "main loop starts" << std::endl;
    lbm_stream_t st = nullptr;
    lbm::check(lbm_stream_create(&st));
    lbm::check(lbm_stream_sync(nullptr));  // initialisation ran on the default stream
    auto one_step = [&]() {
      for (int k = 0; k < 4; ++k)
        lbm::check(lbm_bgk_collide(b[k].coll, b[k].adve, &b[k].g, nullptr, k == A ? &pA : &pX, b[k].rho, b[k].u, st));
      lbm::check(lbm_bgk_add_force_rows(b[A].coll, &b[A].g, b[A].u, omega, 3E-3, 0.0, 3.0, 9.0, L4 + 5, L4 + 55, st));
      for (int k = 0; k < 4; ++k) lbm::check(lbm_advect(b[k].adve, b[k].coll, b[k].R, b[k].C, st));
      lbm::check(lbm_links_apply(links, adve, coll, st));
    };
    if (use_graph && T > 0) {
      lbm_graph* graph = nullptr;
      lbm::check(lbm_graph_begin_capture(st));
      one_step();
      lbm::check(lbm_graph_end_capture(st, &graph));
      lbm::check(lbm_graph_launch(graph, T, st));
      lbm::check(lbm_stream_sync(st));
      lbm_graph_destroy(graph);
    } else {
      for (int t = 0; t < T; ++t) one_step();
      lbm::check(lbm_stream_sync(st));
    }
    lbm_stream_destroy(st);
    const char* names = "ABCD";
    double mass = 0.0;
    for (int k = 0; k < 4; ++k) {
      const size_t n = (size_t)b[k].R * b[k].C;
      lbm::Field rho(b[k].R, b[k].C, 1), u(b[k].R, b[k].C, 2), f(b[k].R, b[k].C, 9);
      lbm::check(lbm_memcpy_d2d(rho.data(), b[k].rho, n * sizeof(double), nullptr));
      lbm::check(lbm_memcpy_d2d(u.data(), b[k].u, n * 2 * sizeof(double), nullptr));
      lbm::check(lbm_memcpy_d2d(f.data(), b[k].adve, n * 9 * sizeof(double), nullptr));
      const auto rh = rho.to_host();
      for (double v : rh) mass += v;
      if (!dump.empty()) {
        const std::string pre = dump + "-" + names[k];
        dump_f64(pre + "-rho.f64", rh);
        dump_f64(pre + "-u.f64", u.to_host());
        dump_f64(pre + "-f.f64", f.to_host());
      }
    }
    std::cout.precision(17);
    std::cout << "steps=" << T << "\nmass=" << mass << std::endl;
    lbm_links_destroy(links);
    for (int k = 0; k < 4; ++k)
      for (double* p : {b[k].adve, b[k].coll, b[k].rho, b[k].u}) lbm_free(p);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
