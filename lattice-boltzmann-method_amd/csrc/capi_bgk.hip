// C ABI, part 2: the BGK hot loop (solver.cpp:23-131 fused) and stream-only.
#include "launch.hpp"

namespace lbm {

// Pressure-periodic virtual rows (horizontal_poiseuille_test.cpp:25-45; compressible variant
// decompose_domain.cpp:25-48): the post-collision populations of row 0 are replaced by
//   feq(rho_inlet, u[R-2]) + f_coll[R-2] - f_equi[R-2]
// and those of row R-1 by feq(rho_outlet, u[1]) + f_coll[1] - f_equi[1].  The source node is
// re-collided here with exactly the arithmetic of the main kernel, so f_coll and f_equi are
// the values the reference holds.  One thread per (edge, column).
template <bool FROM_POST>
__global__ __launch_bounds__(256) void k_bgk_pressure_rows(double* __restrict__ pn,
                                                           const double* __restrict__ in, Geom g,
                                                           Bc bc, BgkModel m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * g.C) return;
  const int e = i / g.C, c = i % g.C;
  const int dst = e ? g.R - 1 : 0, src = e ? 1 : g.R - 2;
  const double rho_bc = (e ? bc.rho_outlet : bc.rho_inlet) * 1.0;
  double f[Q], feq[Q], te[Q], rho, ux, uy;
  if (FROM_POST) {
    gather_bc(f, in, g, bc, src, c);
  } else {
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q] = in[q * g.plane + g.at(src, c)];
  }
  m.collide(f, rho, ux, uy, feq);
  m.feq(te, rho_bc, ux, uy);
  const long o = g.at(dst, c);
#pragma unroll
  for (int q = 0; q < Q; ++q) pn[q * g.plane + o] = (te[q] + f[q]) - feq[q];
}

// f = stream(p) with all boundary fix-ups (solver::advect + post-advect BCs).
__global__ __launch_bounds__(256) void k_stream_only(double* __restrict__ f,
                                                     const double* __restrict__ p, Geom g, Bc bc) {
  const long n_nodes = (long)g.R * g.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes;
       i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / g.C), c = (int)(i % g.C);
    double v[Q];
    gather_bc(v, p, g, bc, r, c);
    const long o = g.at(r, c);
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q * g.plane + o] = v[q];
  }
}

static int launch_pressure_rows(bool from_post, double* pn, const double* in, const Geom& g,
                                const Bc& bc, const BgkModel& m, hipStream_t st) {
  const int n = 2 * g.C;
  if (from_post) LBM_KLAUNCH(k_bgk_pressure_rows<true>, dim3((n + 255) / 256), dim3(256), 0, st, pn, in, g, bc, m);
  else LBM_KLAUNCH(k_bgk_pressure_rows<false>, dim3((n + 255) / 256), dim3(256), 0, st, pn, in, g, bc, m);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

// The reassociated collision (d2q9.hpp BgkFastModel) serves the plain compressible model on every
// launch path -- first collide, single step, multi-step -- so that results do not depend on how a
// run is cut into launches.  tuning "bgk_fast" = 0 selects the reference operation order
// (bit-identical to the oracle); lattices with pressure rows keep it throughout (those rows
// re-collide their source rows in that order).
static bool use_fast_bgk(const lbm_bgk_params* prm, const lbm_bc* bc) {
  // ONE rule for what LBM_FORM_DEFAULT means (round 4): the REASSOCIATED collision wherever a model has one -- the plain
  // compressible BGK model, and its delta form f - omega (f - feq) too (the same polynomial as (1 - omega) f + omega feq:
  // 1e-10 against the oracle after 13 steps of the cylinder preset, tests/test_gpu_ibm.py), as KBC and the two-phase step
  // do.  LBM_FORM_REFERENCE_ORDER in the parameters (or "bgk_fast" = 0 / "bgk_fast_delta" = 0 process-wide) selects the
  // reference's operation order, bitwise equal to the oracle -- that is what the parity tests ask for.
  const bool wanted = prm->form == LBM_FORM_DEFAULT ? (tuning("bgk_fast", 1) && (!prm->delta_form || tuning("bgk_fast_delta", 1)))
                                                    : prm->form == LBM_FORM_REASSOCIATED;
  return wanted && !prm->force_mode && !prm->incompressible && !(bc && bc->pressure_rows);
}

static int check_bgk(const char* fn, const lbm_bgk_params* prm) {
  LBM_REQUIRE(prm, "%s: NULL params", fn);
  LBM_REQUIRE(prm->omega > 0.0 && prm->omega < 2.0, "%s: omega=%g outside (0, 2)", fn, prm->omega);
  LBM_REQUIRE(prm->force_mode == 0 || prm->force_mode == 1, "%s: force_mode=%d", fn, prm->force_mode);
  LBM_REQUIRE(prm->form >= LBM_FORM_DEFAULT && prm->form <= LBM_FORM_REASSOCIATED, "%s: form=%d (LBM_FORM_*)", fn, prm->form);
  return LBM_OK;
}

}  // namespace lbm

using namespace lbm;

extern "C" {

int lbm_bgk_collide(double* p, const double* f, const lbm_geom* g, const lbm_bc* bc,
                    const lbm_bgk_params* prm, double* rho, double* u, lbm_stream_t s) {
  int rc = check_bgk("lbm_bgk_collide", prm);
  if (rc) return rc;
  if (use_fast_bgk(prm, bc))
    return launch_collide_only("lbm_bgk_collide", p, f, g, bc, BgkFastModel(prm->omega), rho, u, as_stream(s));
  const BgkModel m{prm->omega, prm->incompressible, prm->delta_form, prm->force_mode, prm->force_r, prm->force_c, prm->guo_a, prm->guo_b};
  rc = launch_collide_only("lbm_bgk_collide", p, f, g, bc, m, rho, u, as_stream(s));
  if (rc) return rc;
  if (bc && bc->pressure_rows) {
    LBM_REQUIRE(p != f, "lbm_bgk_collide: pressure rows need distinct in/out lattices");
    LBM_REQUIRE(g->ghost == 0, "lbm_bgk_collide: pressure rows are single-block only");
    return launch_pressure_rows(false, p, f, make_geom(*g), make_bc(bc), m, as_stream(s));
  }
  return LBM_OK;
}

int lbm_bgk_stream_collide(double* p_new, const double* p_old, const lbm_geom* g,
                           const lbm_bc* bc, const lbm_bgk_params* prm, int row_begin,
                           int row_end, double* rho, double* u, lbm_stream_t s) {
  int rc = check_bgk("lbm_bgk_stream_collide", prm);
  if (rc) return rc;
  if (use_fast_bgk(prm, bc))
    return launch_stream_collide("lbm_bgk_stream_collide", p_new, p_old, g, bc, BgkFastModel(prm->omega),
                                 row_begin, row_end, rho, u, as_stream(s));
  const BgkModel m{prm->omega, prm->incompressible, prm->delta_form, prm->force_mode, prm->force_r, prm->force_c, prm->guo_a, prm->guo_b};
  rc = launch_stream_collide("lbm_bgk_stream_collide", p_new, p_old, g, bc, m, row_begin, row_end,
                             rho, u, as_stream(s));
  if (rc) return rc;
  if (bc && bc->pressure_rows) {
    LBM_REQUIRE(g->ghost == 0 && row_begin == 0 && row_end == g->R,
                "lbm_bgk_stream_collide: pressure rows need the whole single block");
    return launch_pressure_rows(true, p_new, p_old, make_geom(*g), make_bc(bc), m, as_stream(s));
  }
  return LBM_OK;
}

int lbm_bgk_stream_collide_x2(double* p_new, const double* p_old, const lbm_geom* g,
                              const lbm_bc* bc, const lbm_bgk_params* prm, int row_begin,
                              int row_end, lbm_stream_t s) {
  int rc = check_bgk("lbm_bgk_stream_collide_x2", prm);
  if (rc) return rc;
  if (use_fast_bgk(prm, bc))
    return launch_stream_collide_x2("lbm_bgk_stream_collide_x2", p_new, p_old, g, bc, BgkFastModel(prm->omega),
                                    row_begin, row_end, as_stream(s));
  const BgkModel m{prm->omega, prm->incompressible, prm->delta_form, prm->force_mode, prm->force_r, prm->force_c, prm->guo_a, prm->guo_b};
  return launch_stream_collide_x2("lbm_bgk_stream_collide_x2", p_new, p_old, g, bc, m, row_begin,
                                  row_end, as_stream(s));
}

static int bgk_xn(const char* fn, double* p_new, const double* p_old, const lbm_geom* g, const lbm_bc* bc,
                  const lbm_bgk_params* prm, int n_steps, int row_begin, int row_end, int second_begin,
                  lbm_stream_t s, bool allow_fast = true) {
  int rc = check_bgk(fn, prm);
  if (rc) return rc;
  if (!prm->force_mode) {  // compile-time model: no mode branches inside the unrolled window
    const int key = (prm->incompressible ? 2 : 0) | (prm->delta_form ? 1 : 0);
    if (allow_fast && use_fast_bgk(prm, bc))  // leaner collision: best at one 2-wave block per SIMD pair (146.6 k vs 137 k MLUPS)
      return launch_stream_collide_sw(fn, p_new, p_old, g, bc, BgkFastModel(prm->omega), n_steps, row_begin, row_end, as_stream(s), 2, second_begin);
    switch (key) {
      case 0: return launch_stream_collide_sw(fn, p_new, p_old, g, bc, BgkModelT<0, 0>{prm->omega}, n_steps, row_begin, row_end, as_stream(s), 4, second_begin);
      case 1: return launch_stream_collide_sw(fn, p_new, p_old, g, bc, BgkModelT<0, 1>{prm->omega}, n_steps, row_begin, row_end, as_stream(s), 4, second_begin);
      case 2: return launch_stream_collide_sw(fn, p_new, p_old, g, bc, BgkModelT<1, 0>{prm->omega}, n_steps, row_begin, row_end, as_stream(s), 4, second_begin);
      default: return launch_stream_collide_sw(fn, p_new, p_old, g, bc, BgkModelT<1, 1>{prm->omega}, n_steps, row_begin, row_end, as_stream(s), 4, second_begin);
    }
  }
  const BgkModel m{prm->omega, prm->incompressible, prm->delta_form, prm->force_mode, prm->force_r, prm->force_c, prm->guo_a, prm->guo_b};
  // runtime-mode model (body force): uncapped 2-wave blocks (the 4-wave variants spill 280-410 VGPRs)
  return launch_stream_collide_sw(fn, p_new, p_old, g, bc, m, n_steps, row_begin, row_end, as_stream(s), 2, second_begin);
}

int lbm_bgk_stream_collide_xn(double* p_new, const double* p_old, const lbm_geom* g,
                              const lbm_bc* bc, const lbm_bgk_params* prm, int n_steps,
                              int row_begin, int row_end, lbm_stream_t s) {
  return bgk_xn("lbm_bgk_stream_collide_xn", p_new, p_old, g, bc, prm, n_steps, row_begin, row_end, -1, s);
}

// the same on TWO row ranges of equal height in ONE launch: [row_begin, row_end) and
// [row_begin2, row_begin2 + (row_end - row_begin)) -- the edge rows at both ends of a slab, which as two
// launches on one stream would run one after the other (the second queued behind a grid-filling
// interior launch); wall-bounded lattices fall back to two launches
int lbm_bgk_stream_collide_xn2(double* p_new, const double* p_old, const lbm_geom* g,
                               const lbm_bc* bc, const lbm_bgk_params* prm, int n_steps,
                               int row_begin, int row_end, int row_begin2, lbm_stream_t s) {
  LBM_REQUIRE(row_begin2 >= 0, "lbm_bgk_stream_collide_xn2: row_begin2=%d", row_begin2);
  return bgk_xn("lbm_bgk_stream_collide_xn2", p_new, p_old, g, bc, prm, n_steps, row_begin, row_end, row_begin2, s);
}

int lbm_stream(double* f, const double* p, const lbm_geom* g, const lbm_bc* bc, lbm_stream_t s) {
  int rc = validate_geom_bc("lbm_stream", g, bc);
  if (rc) return rc;
  LBM_REQUIRE(f && p && f != p, "lbm_stream: NULL or aliased lattices");
  const Geom gg = make_geom(*g);
  const long n = (long)gg.R * gg.C;
  LBM_KLAUNCH(k_stream_only, dim3(capped_grid((n + 255) / 256)), dim3(256), 0, as_stream(s),
                     f, p, gg, make_bc(bc));
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

}  // extern "C"

// internal (internal.hpp): the far rows of a lattice with pressure rows -- such lattices run the
// reference operation order on every path (use_fast_bgk), and so must their window launches although
// the bc they are given has the pressure rows taken out
int lbm::bgk_stream_collide_xn_ref(double* p_new, const double* p_old, const lbm_geom* g, const lbm_bc* bc,
                                   const lbm_bgk_params* prm, int n_steps, int row_begin, int row_end, hipStream_t st) {
  return bgk_xn("bgk_stream_collide_xn_ref", p_new, p_old, g, bc, prm, n_steps, row_begin, row_end, -1, (lbm_stream_t)st, false);
}

// internal: lbm_bgk_collide pinned to the reference operation order, no pressure rows (slabs of a lattice with
// pressure-periodic rows collide their rows -- ghost rows included -- this way; the virtual rows come from the seam lattice)
int lbm::bgk_collide_ref(double* p, const double* f, const lbm_geom* g, const lbm_bgk_params* prm, hipStream_t st) {
  int rc = check_bgk("bgk_collide_ref", prm);
  if (rc) return rc;
  const BgkModel m{prm->omega, prm->incompressible, prm->delta_form, prm->force_mode, prm->force_r, prm->force_c, prm->guo_a, prm->guo_b};
  return launch_collide_only("bgk_collide_ref", p, f, g, nullptr, m, nullptr, nullptr, st);
}

bool lbm::bgk_uses_fast_model(const lbm_bgk_params* prm, const lbm_bc* bc) { return prm && use_fast_bgk(prm, bc); }
