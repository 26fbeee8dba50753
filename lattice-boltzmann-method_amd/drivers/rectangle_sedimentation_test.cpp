// Restatement of test/rectangle_sedimentation_test.cpp: a fluid distribution f and
// a sediment-concentration distribution g (advection-diffusion with the settling velocity w_s added
// to BOTH velocity components, :124) in a channel with anti-bounce-back inlet / outlet columns, a
// specular top, a no-slip bottom and a three-sided rectangular obstacle at hard-coded coordinates
// (:71-73: the lattice needs more than 151 rows and 250 columns).
//   usage: rectangle_sedimentation_test params.toml [--steps N] [--dump prefix]
//
// Engine mapping (operator level, like the reference's loop): per step one fused collide launch for
// f (with rho, u), the unfused equilibrium / collision pair for g (its equilibrium takes u + w_s,
// not its own moments), two periodic stream launches, and TWO gather launches (lbm_links_*) for the
// ~50 slice assignments of the boundary conditions: plain copies, sign-flipped copies, and
// "-f_coll + wall term" with per-row terms that lbm_wall_terms refreshes from the step's u.
// Dumps (raw f64): <prefix>-rho.f64 [X][Y], -u.f64 [X][Y][2], -C.f64 [X][Y], -f.f64 / -g.f64 [X][Y][9].
#include <iostream>

#include "../include/lbm/lbm.hpp"
#include "../include/lbm/params.hpp"
#include "common.hpp"

namespace {
const int kNone = 1 << 30;
int norm(int i, int n) { return i == kNone ? n : (i < 0 ? i + n : i); }
// the eight (dst, src) population pairs of the anti-bounce-back columns, in the driver's order
const int kDst[8] = {3, 4, 1, 2, 7, 8, 5, 6}, kSrc[8] = {1, 2, 3, 4, 5, 6, 7, 8};

struct Table {
  lbm_links* t;
  int X, Y;
  // adve[r0:r1, col, q] = scale * coll[r0:r1, col, sq] (+ addends[add0 + 9 (r - r0_of_terms) ...])
  void rows(int r0, int r1, int col, int q, int sq, double scale = 1.0, long long add0 = -1) {
    r0 = norm(r0, X);
    r1 = norm(r1, X);
    col = norm(col, Y);
    lbm::check(lbm_links_add_affine(t, 0, q, r0, col, 1, 0, 0, sq, r0, col, 1, 0, r1 - r0, scale,
                                    add0 < 0 ? -1 : add0 + 9LL * r0, 9));
  }
  void cols(int row, int c0, int c1, int q, int sq, double scale = 1.0) {
    row = norm(row, X);
    lbm::check(lbm_links_add_affine(t, 0, q, row, c0, 0, 1, 0, sq, row, c0, 0, 1, c1 - c0, scale, -1, 0));
  }
};
}  // namespace

int main(int argc, char* argv[]) {
  if (argc < 2) {
    std::cerr << "usage: " << argv[0] << " params.toml [--steps N] [--dump prefix]\n";
    return 1;
  }
  lbm::toml::table tbl;
  try {
    tbl = lbm::toml::parse_file(argv[1]);
  } catch (const lbm::toml::parse_error& err) {
    std::cerr << "Parsing failed:\n" << err.what() << "\n";
    return 1;
  }
  try {
    const params::flow fp{tbl};
    std::cout << fp << "\n";
    const params::lattice lp{tbl, fp};
    std::cout << lp << "\n";
    const params::simulation sp{tbl, lp};
    std::cout << sp << "\n";
    const int steps = std::stoi(arg_value(argc, argv, "--steps", std::to_string(sp.total_steps)));
    const std::string dump = arg_value(argc, argv, "--dump", "");
    const int X = lp.X, Y = lp.Y;
    const int R23 = -151, C28 = 200, C38 = 250;  // :71-73
    if (X <= 152 || Y <= 251) {
      std::cerr << "the obstacle of :71-73 needs X > 152 and Y > 251 (got " << X << " x " << Y << ")\n";
      return 1;
    }
    if (lbm_device_count() < 1) {
      std::cerr << "no HIP device available\n";
      return 2;
    }
    const double w_s = 3e-3, scalar_C_w = 1e-3;  // :91-92
    const size_t n = (size_t)X * Y;
    lbm::Field f_adve(X, Y, 9), f_coll(X, Y, 9), g_adve(X, Y, 9), g_coll(X, Y, 9), g_equi(X, Y, 9);
    lbm::Field u(X, Y, 2), u_new(X, Y, 2), u_s(X, Y, 2), rho(X, Y, 1), rho_new(X, Y, 1), C(X, Y, 1);
    // C_w: the last 50 rows of the inlet column carry sediment (:93-94)
    std::vector<double> C_w(X, 0.0), Ch(n, 0.0);
    for (int r = X - 50; r < X; ++r) C_w[r] = scalar_C_w;
    for (int r = 0; r < X; ++r) Ch[(size_t)r * Y] = C_w[r];
    C.from_host(Ch);
    {
      std::vector<double> uh(n * 2, 0.0);
      for (size_t i = 0; i < n; ++i) uh[2 * i + 1] = lp.u;  // :83
      u.from_host(uh);
    }
    rho.fill(1.0);
    solver::equilibrium(g_adve, u, C);           // :95
    solver::incomp_equilibrium(f_adve, u, rho);  // :100
    solver::calc_rho(rho, f_adve);               // :103-104
    solver::calc_u(u, f_adve, rho);

    // wall-term rows: [0, 9X) inlet (fixed), [9X, 18X) outlet (per step), [18X, 27X) concentration inlet
    double *terms = nullptr, *d_Cw = nullptr;
    lbm::check(lbm_malloc((void**)&terms, 27 * (size_t)X * sizeof(double)));
    lbm::check(lbm_malloc((void**)&d_Cw, (size_t)X * sizeof(double)));
    lbm::check(lbm_memcpy_h2d(d_Cw, C_w.data(), (size_t)X * sizeof(double), nullptr));
    {  // inlet: fixed wall velocity (0, lp.u) (:135) -- the same arithmetic on the host
      std::vector<double> a((size_t)9 * X);
      const double uu = 0.0 * 0.0 + lp.u * lp.u;
      for (int q = 0; q < 9; ++q) {
        const double uc = 0.0 * solver::c[0][q] + lp.u * solver::c[1][q];
        const double t = ((2.0 + 9.0 * (uc * uc)) - 3.0 * uu) * solver::E[q];
        for (int r = 0; r < X; ++r) a[(size_t)r * 9 + q] = t;
      }
      lbm::check(lbm_memcpy_h2d(terms, a.data(), a.size() * sizeof(double), nullptr));
    }
    const lbm_geom geom{X, Y, 0, 0, 0};
    // f: post-advect assignments in the driver's order (:148-194)
    lbm_links* lf = nullptr;
    lbm::check(lbm_links_create(&lf, 1, &geom));
    Table tf{lf, X, Y};
    for (int k = 0; k < 8; ++k) tf.rows(1, -1, 0, kDst[k], kSrc[k], -1.0, 0 * 9LL * X + kSrc[k]);
    for (int k = 0; k < 8; ++k) tf.rows(0, kNone, -1, kDst[k], kSrc[k], -1.0, 9LL * X + kSrc[k]);
    tf.cols(0, 0, Y, 8, 7);  // specular top
    tf.cols(0, 0, Y, 1, 3);
    tf.cols(0, 0, Y, 5, 6);
    tf.cols(-1, 0, Y, 7, 5);  // bottom: no slip
    tf.cols(-1, 0, Y, 3, 1);
    tf.cols(-1, 0, Y, 6, 8);
    tf.rows(R23 + 1, -1, C28, 8, 6);  // rectangle: first wall, ceiling, second wall
    tf.rows(R23 + 1, -1, C28, 4, 2);
    tf.rows(R23 + 1, -1, C28, 7, 5);
    tf.cols(R23, C28, C38 + 1, 6, 8);
    tf.cols(R23, C28, C38 + 1, 3, 1);
    tf.cols(R23, C28, C38 + 1, 7, 5);
    tf.rows(R23 + 1, -1, C38, 5, 7);
    tf.rows(R23 + 1, -1, C38, 2, 4);
    tf.rows(R23 + 1, -1, C38, 6, 8);
    lbm::check(lbm_links_finalize(lf));
    // g: post-advect assignments (:210-234)
    lbm_links* lg = nullptr;
    lbm::check(lbm_links_create(&lg, 1, &geom));
    Table tg{lg, X, Y};
    for (int k = 0; k < 8; ++k) tg.rows(1, -1, 0, kDst[k], kSrc[k], -1.0, 18LL * X + kSrc[k]);
    tg.rows(R23 + 1, kNone, C28, 8, 6, -1.0);
    tg.rows(R23 + 1, kNone, C28, 4, 2, -1.0);
    tg.rows(R23 + 1, kNone, C28, 7, 5, -1.0);
    tg.cols(R23, C28, C38 + 1, 6, 8, -1.0);
    tg.cols(R23, C28, C38 + 1, 3, 1, -1.0);
    tg.cols(R23, C28, C38 + 1, 7, 5, -1.0);
    tg.rows(R23 + 1, -1, C38, 5, 7, -1.0);
    tg.rows(R23 + 1, -1, C38, 2, 4, -1.0);
    tg.rows(R23 + 1, -1, C38, 6, 8, -1.0);
    tg.cols(-1, 0, Y, 6, 8);
    tg.cols(-1, 0, Y, 3, 1);
    tg.cols(-1, 0, Y, 7, 5);
    lbm::check(lbm_links_finalize(lg));
    // zero gradient on g_coll BEFORE propagation (:136-140): in place, second after first
    lbm_links *z1 = nullptr, *z2 = nullptr;
    lbm::check(lbm_links_create(&z1, 1, &geom));
    lbm::check(lbm_links_create(&z2, 1, &geom));
    for (int q = 0; q < 9; ++q) {
      lbm::check(lbm_links_add(z1, 0, q, 0, 0, 0, 1, 0, q, 1, 0, 0, 1, Y));
      lbm::check(lbm_links_add(z2, 0, q, 1, Y - 1, 1, 0, 0, q, 1, Y - 2, 1, 0, X - 2));
    }
    lbm::check(lbm_links_finalize(z1));
    lbm::check(lbm_links_finalize(z2));
    std::cout << "links: f " << lbm_links_count(lf) << ", g " << lbm_links_count(lg) << std::endl;

    lbm_bgk_params prm{lp.omega, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, LBM_FORM_DEFAULT};
    double* fa[1] = {f_adve.data()};
    const double* fc[1] = {f_coll.data()};
    double* ga[1] = {g_adve.data()};
    double* gcw[1] = {g_coll.data()};
    const double* gc[1] = {g_coll.data()};
    std::cout << "main loop\n";
    for (int t = 0; t < steps; ++t) {
      // :123-131  f: equilibrium + collision from its own moments (= the held u, rho); g: equilibrium(u + w_s, C)
      lbm::check(lbm_bgk_collide(f_coll.data(), f_adve.data(), &geom, nullptr, &prm, rho.data(), u.data(), nullptr));
      lbm::check(lbm_axpb(u_s.data(), u.data(), 1.0, w_s, (long long)(2 * n), nullptr));
      solver::equilibrium(g_equi, u_s, C);
      solver::collision(g_coll, g_adve, g_equi, lp.omega / 1.0);
      lbm::check(lbm_links_apply(z1, gcw, gc, nullptr));
      lbm::check(lbm_links_apply(z2, gcw, gc, nullptr));
      // outlet wall velocity 1.5 u[:, -1] - 0.5 u[:, -2] from the step's u (:161-162)
      lbm::check(lbm_wall_terms(terms + 9 * (size_t)X, u.data(), X, Y, Y - 1, 1.5, Y - 2, -0.5, 0.0, 0, nullptr, 1.0, nullptr));
      solver::advect(f_adve, f_coll);  // :143-144
      solver::advect(g_adve, g_coll);
      lbm::check(lbm_links_apply_affine(lf, fa, fc, terms, nullptr));
      solver::calc_rho(rho_new, f_adve);  // :197-198
      solver::calc_u(u_new, f_adve, rho_new);
      // concentration inlet from the NEW u (:202-217), twice the term as the driver writes 2.0 * g_abb_bc
      lbm::check(lbm_wall_terms(terms + 18 * (size_t)X, u_new.data(), X, Y, 0, 1.0, 0, 0.0, w_s, 1, d_Cw, 2.0, nullptr));
      lbm::check(lbm_links_apply_affine(lg, ga, gc, terms, nullptr));
      solver::calc_rho(C, g_adve);  // :235
    }
    // the state the driver holds after the loop: rho, u from the last :197-198
    if (steps > 0) {
      rho = rho_new;
      u = u_new;
    }
    const auto Chost = C.to_host();
    double mass_c = 0.0;
    for (double v : Chost) mass_c += v;
    std::cout.precision(17);
    std::cout << "steps=" << steps << "\nsediment_mass=" << mass_c << std::endl;
    if (!dump.empty()) {
      dump_f64(dump + "-rho.f64", rho.to_host());
      dump_f64(dump + "-u.f64", u.to_host());
      dump_f64(dump + "-C.f64", Chost);
      dump_f64(dump + "-f.f64", f_adve.to_host());
      dump_f64(dump + "-g.f64", g_adve.to_host());
    }
    for (lbm_links* l : {lf, lg, z1, z2}) lbm_links_destroy(l);
    lbm_free(terms);
    lbm_free(d_Cw);
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << "\n";
    return 3;
  }
  return 0;
}
