// C ABI, part 4: KBC ("MRT moment-space collision" of BASELINE config 3), src/ulbm.cpp.
#include "kbc.hpp"
#include "launch.hpp"

namespace lbm {

// kbc::eval_equilibrium (ulbm.cpp:248-263).  zero_u2 != 0 reproduces the state in which the
// driver calls it (ulbm_double_shear_flow.cpp:96): the ctor left ux2 = uy2 = 0.
__global__ __launch_bounds__(256) void k_kbc_equilibrium(double* __restrict__ feq,
                                                         const double* __restrict__ m0,
                                                         const double* __restrict__ m1, long n,
                                                         int zero_u2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double ux = m1[i], uy = m1[n + i];
    const double ux2 = zero_u2 ? 0.0 : ux * ux, uy2 = zero_u2 ? 0.0 : uy * uy;
    double e[Q];
    KbcModel::feq_poly(e, ux, uy, ux2, uy2);
    const double r = m0[i];
#pragma unroll
    for (int q = 0; q < Q; ++q) feq[q * n + i] = e[q] * r;
  }
}

// collide() with caller-supplied moments (unit parity with kbc::collide on arbitrary state)
__global__ __launch_bounds__(256) void k_kbc_collide_given(double* __restrict__ out,
                                                           const double* __restrict__ f,
                                                           const double* __restrict__ m0,
                                                           const double* __restrict__ m1, long n,
                                                           KbcModel m) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = f[q * n + i];
    m.collide_with(v, m0[i], m1[i], m1[n + i]);
#pragma unroll
    for (int q = 0; q < Q; ++q) out[q * n + i] = v[q];
  }
}

// the same on a lattice with its own plane stride (solver context); m0 [R][C], m1 [2][R][C] dense
__global__ __launch_bounds__(256) void k_kbc_collide_first(double* __restrict__ pn,
                                                           const double* __restrict__ in, Geom g,
                                                           const double* __restrict__ m0,
                                                           const double* __restrict__ m1, KbcModel m) {
  const long n = (long)g.R * g.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long o = g.at((int)(i / g.C), (int)(i % g.C));
    double v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = in[q * g.plane + o];
    m.collide_with(v, m0[i], m1[i], m1[n + i]);
#pragma unroll
    for (int q = 0; q < Q; ++q) pn[q * g.plane + o] = v[q];
  }
}

// Pressure-periodic virtual rows of test/ulbm_poiseuille.cpp:36-58: row 0 <- feq_inc(rho_inlet,
// u[R-2]) + f_coll[R-2] - f_equi[R-2], row R-1 <- feq_inc(rho_outlet, u[1]) + f_coll[1] - f_equi[1],
// where feq_inc is solver::incomp_equilibrium (:50,:54) and f_equi = kbc.iequi_f.pow(-1) (:122),
// i.e. the reciprocal of the reciprocal the collision stored -- kept as two divisions.
// MODE 0: `in` is the pre-collision state and the moments are GIVEN (first iteration of the
// driver, which starts from adve_f = 0 with m0 = 1; "dense single block" above refers to m0/m1); MODE 1: `in` holds post-collision
// populations, streamed at read time, moments recomputed (:136-139).
template <int MODE>
__global__ __launch_bounds__(256) void k_kbc_pressure_rows(double* __restrict__ pn,
                                                           const double* __restrict__ in, Geom g,
                                                           Bc bc, KbcModel m,
                                                           const double* __restrict__ m0,
                                                           const double* __restrict__ m1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * g.C) return;
  const int e = i / g.C, c = i % g.C;
  const int dst = e ? g.R - 1 : 0, src = e ? 1 : g.R - 2;
  const double rho_bc = (e ? bc.rho_outlet : bc.rho_inlet) * 1.0;
  double f[Q], fe[Q], te[Q], rho, ux, uy;
  if (MODE == 1) {
    gather_bc(f, in, g, bc, src, c);
    double jx, jy;
    BgkModel::moments(f, rho, jx, jy);
    ux = jx / rho;
    uy = jy / rho;
  } else {
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q] = in[q * g.plane + g.at(src, c)];
    const long n = (long)g.R * g.C, o = (long)src * g.C + c;
    rho = m0[o];
    ux = m1[o];
    uy = m1[n + o];
  }
  KbcModel::feq_poly(fe, ux, uy, ux * ux, uy * uy);  // eval_iequilibrium :230-246
#pragma unroll
  for (int q = 0; q < Q; ++q) fe[q] = 1.0 / (1.0 / (fe[q] * rho));
  m.collide_with(f, rho, ux, uy);
  const BgkModel inc{1.0, 1, 0, 0, 0.0, 0.0, 0.0, 0.0};
  inc.feq(te, rho_bc, ux, uy);
  const long o = g.at(dst, c);
#pragma unroll
  for (int q = 0; q < Q; ++q) pn[q * g.plane + o] = (te[q] + f[q]) - fe[q];
}

static int launch_kbc_pressure_rows(int mode, double* pn, const double* in, const lbm_geom* lg,
                                    const lbm_bc* lbc, const KbcModel& m, const double* m0,
                                    const double* m1, hipStream_t st) {
  LBM_REQUIRE(lg->ghost == 0 && pn != in, "lbm_kbc: pressure rows need a single block and distinct lattices");
  const Geom g = make_geom(*lg);
  const Bc bc = make_bc(lbc);
  const int n = 2 * g.C;
  if (mode) LBM_KLAUNCH(k_kbc_pressure_rows<1>, dim3((n + 255) / 256), dim3(256), 0, st, pn, in, g, bc, m, m0, m1);
  else LBM_KLAUNCH(k_kbc_pressure_rows<0>, dim3((n + 255) / 256), dim3(256), 0, st, pn, in, g, bc, m, m0, m1);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

// launch_stream_collide_sw restricted to the instantiations KBC needs (2 waves per block so that
// the register allocator is not capped: the collision keeps ~100 doubles live next to the ring)
static int launch_stream_collide_sw_kbc(const char* fn, double* pn, const double* po,
                                        const lbm_geom* lg, const lbm_bc* lbc, const KbcFastModel& m,
                                        int depth, int row_begin, int row_end, hipStream_t st) {
  int rc = validate_geom_bc(fn, lg, lbc);
  if (rc) return rc;
  LBM_REQUIRE(pn && po && pn != po, "%s: NULL or aliased lattices", fn);
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= lg->R,
              "%s: row range [%d, %d) outside [0, %d)", fn, row_begin, row_end, lg->R);
  const Bc bc = make_bc(lbc);
  auto carried = [](int m) { return m == LBM_EDGE_PERIODIC || m == LBM_EDGE_HALO || bc_is_wall(m); };
  LBM_REQUIRE(carried(bc.row_lo) && carried(bc.row_hi) && carried(bc.col_lo) && carried(bc.col_hi) && !bc.pressure_rows,
              "%s: multi-step launches carry periodic / halo / bounce-back / specular / velocity edges only", fn);
  LBM_REQUIRE(!bc_mixed_axis(bc), "%s: multi-step launches need both edges of an axis walled or neither", fn);
  const bool walls = bc_needs_edge_pass(bc);
  LBM_REQUIRE(!walls || (lg->ghost == 0 && depth <= 3), "%s: wall-carrying launches are single-block only, 2..3 steps", fn);
  LBM_REQUIRE(lg->ghost == 0 || lg->ghost >= depth, "%s: ghost=%d rows, need 0 or >= %d", fn, lg->ghost, depth);
  LBM_REQUIRE(lg->R >= 4 * depth + 8 && lg->C >= 64, "%s: lattice %dx%d too small for %d-step launches", fn, lg->R, lg->C, depth);
  if (row_begin == row_end) return LBM_OK;
  const Geom g = make_geom(*lg);
  const int nrows = row_end - row_begin;
  const int W = sw_strip_width(depth, sw_full_strips<KbcFastModel>::value);
  const int strips = (g.C + W - 1) / W;
  LBM_REQUIRE((long)strips * ((nrows + 31) / 32) < (1L << 30), "%s: lattice too large for one launch", fn);
  // rows per wave: "sw_rows" if set, else fitted to the resident wave slots of the instance (launch.hpp)
#define LBM_KBC_SW(...)                                                                               \
  {                                                                                                   \
    int rpc = tuning("sw_rows", -1);                                                                  \
    if (rpc <= 0) {                                                                                   \
      const long slots = sw_wave_slots((const void*)k_stream_collide_sw<__VA_ARGS__>, 128);           \
      rpc = slots > 0 ? sw_pick_rows(nrows, strips, depth, slots) : 64;                               \
    }                                                                                                 \
    if (rpc > nrows) rpc = nrows;                                                                     \
    const int n_waves = strips * ((nrows + rpc - 1) / rpc);                                           \
    LBM_KLAUNCH((k_stream_collide_sw<__VA_ARGS__>), dim3((n_waves + 1) / 2), dim3(128), 0, st, pn, po, g, m, row_begin, row_end, rpc, strips, n_waves, LBM_KBC_SW_TAIL); \
  }
  if (walls) {  // plain instantiation on the wall-free interior, wall-carrying one on the frame (launch.hpp)
    rc = depth == 2 ? sw_launch_walls<KbcFastModel, 2>(pn, po, g, m, bc, row_begin, row_end, st)
                    : sw_launch_walls<KbcFastModel, 3>(pn, po, g, m, bc, row_begin, row_end, st);
    if (rc) return rc;
    LBM_CHECK_LAUNCH();
    return LBM_OK;
  }
#define LBM_KBC_SW_TAIL tuning("sw_xcd", 0)
  // "sw_ldsring" (default 1): the ring of the 2- / 3-step window in wave-private LDS instead of registers (d2q9.hpp LDSR)
  const bool ldsr = tuning("sw_ldsring", 1) != 0;
  if (depth == 2 && ldsr) LBM_KBC_SW(KbcFastModel, 2, 2, true, false, false, true)
  else if (depth == 3 && ldsr) LBM_KBC_SW(KbcFastModel, 3, 2, true, false, false, true)
  else if (depth == 4 && ldsr) LBM_KBC_SW(KbcFastModel, 4, 2, true, false, false, true)
  else if (depth == 2) LBM_KBC_SW(KbcFastModel, 2, 2, true)
  else if (depth == 3) LBM_KBC_SW(KbcFastModel, 3, 2, true)
  else LBM_KBC_SW(KbcFastModel, 4, 2, true)
#undef LBM_KBC_SW_TAIL
#undef LBM_KBC_SW
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

}  // namespace lbm
// Two steps per launch in the reference operation order (KbcModel): the far rows of a lattice with
// pressure-periodic rows (capi_solver.hip solver_pressure_block).  ~940 f64 operations per collision and
// a register ring: one wave per SIMD with scratch spills -- it exists for bit-parity with the single-step
// path, not for speed.
int lbm::kbc_stream_collide_x2_ref(double* pn, const double* po, const lbm_geom* lg, const lbm_bc* lbc,
                                   const lbm_kbc_params* prm, int row_begin, int row_end, hipStream_t st) {
  const char* fn = "kbc_stream_collide_x2_ref";
  int rc = validate_geom_bc(fn, lg, lbc);
  if (rc) return rc;
  LBM_REQUIRE(pn && po && pn != po && prm, "%s: bad argument", fn);
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= lg->R, "%s: row range", fn);
  const Bc bc = make_bc(lbc);
  // (slabs: ghost >= 2 rows behind HALO edges -- the far rows of a slab of a lattice with pressure rows, capi_slab_pressure.hip)
  auto carried = [](int m) { return m == LBM_EDGE_PERIODIC || m == LBM_EDGE_HALO || bc_is_wall(m); };
  LBM_REQUIRE((lg->ghost == 0 || lg->ghost >= 2) && carried(bc.row_lo) && carried(bc.row_hi) && carried(bc.col_lo) && carried(bc.col_hi) &&
                  !bc.pressure_rows && !bc_mixed_axis(bc) && lg->R >= 16 && lg->C >= 64,
              "%s: periodic / halo / wall edges only (ghost rows: 0 or >= 2)", fn);
  if (row_begin == row_end) return LBM_OK;
  const Geom g = make_geom(*lg);
  const KbcModel m{prm->s2};
  if (bc_needs_edge_pass(bc)) {
    rc = sw_launch_walls<KbcModel, 2>(pn, po, g, m, bc, row_begin, row_end, st);
    if (rc) return rc;
  } else {
    const int W = sw_strip_width(2, sw_full_strips<KbcModel>::value), strips = (g.C + W - 1) / W;
    sw_launch_part<KbcModel, 2, false>(pn, po, g, m, bc, row_begin, row_end, 0, strips, 0, st);
  }
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}
namespace lbm {

static int check_kbc(const char* fn, const lbm_kbc_params* prm) {
  LBM_REQUIRE(prm, "%s: NULL params", fn);
  LBM_REQUIRE(prm->s2 > 0.0 && prm->s2 <= 2.0, "%s: s2=%g outside (0, 2]", fn, prm->s2);
  LBM_REQUIRE(prm->form >= LBM_FORM_DEFAULT && prm->form <= LBM_FORM_REASSOCIATED, "%s: form=%d (LBM_FORM_*)", fn, prm->form);
  return LBM_OK;
}

// the reassociated collision unless the parameters (or, for LBM_FORM_DEFAULT, the process-wide knob) ask for the reference order
bool kbc_uses_fast_model(const lbm_kbc_params* prm) {
  return prm->form == LBM_FORM_DEFAULT ? tuning("kbc_fast", 1) != 0 : prm->form == LBM_FORM_REASSOCIATED;
}

}  // namespace lbm

using namespace lbm;

extern "C" {

int lbm_kbc_equilibrium(double* feq, const double* m0, const double* m1, int R, int C,
                        int zero_u2, lbm_stream_t s) {
  LBM_REQUIRE(feq && m0 && m1 && R > 0 && C > 0, "lbm_kbc_equilibrium: bad argument");
  const long n = (long)R * C;
  LBM_KLAUNCH(k_kbc_equilibrium, dim3(capped_grid((n + 255) / 256)), dim3(256), 0,
                     as_stream(s), feq, m0, m1, n, zero_u2);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_kbc_collide_given_moments(double* coll, const double* f, const double* m0,
                                  const double* m1, const lbm_kbc_params* prm, int R, int C,
                                  lbm_stream_t s) {
  int rc = check_kbc("lbm_kbc_collide_given_moments", prm);
  if (rc) return rc;
  LBM_REQUIRE(coll && f && m0 && m1 && R > 0 && C > 0, "lbm_kbc_collide_given_moments: bad argument");
  const long n = (long)R * C;
  LBM_KLAUNCH(k_kbc_collide_given, dim3(capped_grid((n + 255) / 256)), dim3(256), 0,
                     as_stream(s), coll, f, m0, m1, n, KbcModel{prm->s2});
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_kbc_collide(double* p, const double* f, const lbm_geom* g, const lbm_bc* bc,
                    const lbm_kbc_params* prm, double* rho, double* u, lbm_stream_t s) {
  int rc = check_kbc("lbm_kbc_collide", prm);
  if (rc) return rc;
  LBM_REQUIRE(!(bc && bc->pressure_rows), "lbm_kbc_collide: pressure rows need the moments of the source rows: use lbm_kbc_collide_first");
  if (kbc_uses_fast_model(prm))
    return launch_collide_only("lbm_kbc_collide", p, f, g, bc, KbcFastModel(prm->s2), rho, u, as_stream(s));
  return launch_collide_only("lbm_kbc_collide", p, f, g, bc, KbcModel{prm->s2}, rho, u, as_stream(s));
}

int lbm_kbc_collide_first(double* p, const double* f, const double* m0, const double* m1,
                          const lbm_geom* g, const lbm_bc* bc, const lbm_kbc_params* prm,
                          lbm_stream_t s) {
  int rc = check_kbc("lbm_kbc_collide_first", prm);
  if (rc) return rc;
  rc = validate_geom_bc("lbm_kbc_collide_first", g, bc);
  if (rc) return rc;
  LBM_REQUIRE(p && f && m0 && m1 && p != f, "lbm_kbc_collide_first: bad argument");
  LBM_REQUIRE(g->ghost == 0, "lbm_kbc_collide_first: single block only");
  const long n = (long)g->R * g->C;
  LBM_KLAUNCH(k_kbc_collide_first, dim3(capped_grid((n + 255) / 256)), dim3(256), 0, as_stream(s), p, f,
              make_geom(*g), m0, m1, KbcModel{prm->s2});
  LBM_CHECK_LAUNCH();
  if (bc && bc->pressure_rows)
    return launch_kbc_pressure_rows(0, p, f, g, bc, KbcModel{prm->s2}, m0, m1, as_stream(s));
  return LBM_OK;
}

int lbm_kbc_stream_collide(double* p_new, const double* p_old, const lbm_geom* g,
                           const lbm_bc* bc, const lbm_kbc_params* prm, int row_begin,
                           int row_end, double* rho, double* u, lbm_stream_t s) {
  int rc = check_kbc("lbm_kbc_stream_collide", prm);
  if (rc) return rc;
  // the reassociated collision (kbc.hpp) unless the caller asks for the reference operation order
  // (tuning "kbc_fast" = 0) or the pressure rows -- which re-collide their source rows in that order
  // -- are in use
  if (kbc_uses_fast_model(prm) && !(bc && bc->pressure_rows))
    return launch_stream_collide("lbm_kbc_stream_collide", p_new, p_old, g, bc, KbcFastModel(prm->s2),
                                 row_begin, row_end, rho, u, as_stream(s));
  rc = launch_stream_collide("lbm_kbc_stream_collide", p_new, p_old, g, bc, KbcModel{prm->s2},
                             row_begin, row_end, rho, u, as_stream(s));
  if (rc) return rc;
  if (bc && bc->pressure_rows) {
    LBM_REQUIRE(row_begin == 0 && row_end == g->R, "lbm_kbc_stream_collide: pressure rows need the whole block");
    return launch_kbc_pressure_rows(1, p_new, p_old, g, bc, KbcModel{prm->s2}, nullptr, nullptr, as_stream(s));
  }
  return LBM_OK;
}

// n_steps = 2..4 time steps in ONE launch through the register sliding window of the BGK path
// (d2q9.hpp k_stream_collide_sw, here with the reassociated KBC collision): periodic / halo
// edges only, whole block or a row range of a slab with ghost >= n_steps.  The reference-order
// model (tuning "kbc_fast" = 0) has no multi-step instantiation: callers fall back to single steps.
int lbm_kbc_stream_collide_xn(double* p_new, const double* p_old, const lbm_geom* g,
                              const lbm_bc* bc, const lbm_kbc_params* prm, int n_steps,
                              int row_begin, int row_end, lbm_stream_t s) {
  int rc = check_kbc("lbm_kbc_stream_collide_xn", prm);
  if (rc) return rc;
  LBM_REQUIRE(kbc_uses_fast_model(prm), "lbm_kbc_stream_collide_xn: multi-step launches exist for the reassociated collision only (form = LBM_FORM_REASSOCIATED, or the default with kbc_fast = 1)");
  LBM_REQUIRE(n_steps >= 2 && n_steps <= 4, "lbm_kbc_stream_collide_xn: %d steps per launch (supported: 2..4)", n_steps);
  return launch_stream_collide_sw_kbc("lbm_kbc_stream_collide_xn", p_new, p_old, g, bc,
                                      KbcFastModel(prm->s2), n_steps, row_begin, row_end, as_stream(s));
}

}  // extern "C"
