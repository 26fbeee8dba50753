// C ABI, part 3: solver context = the driver loop of the reference's test/*.cpp mains
// (single block).  Model dispatch lives here; kernels are in d2q9.hpp / kbc.hpp.
#include <new>

#include "kbc.hpp"
#include "launch.hpp"

struct lbm_solver {
  int model;
  lbm_geom g;
  lbm_bc bc;
  lbm_bgk_params bgk;
  lbm_kbc_params kbc;
  hipStream_t st;
  double* lat[2];   // SoA lattices [9][R+2g][C]
  double* stage;    // AoS staging [R][C][9] (also SoA scratch for get_f)
  double* rho;      // [R][C]
  double* u;        // [2][R][C]
  int cur;          // lat[cur] holds the state
  bool post;        // state is post-collision (P-form); false: pre-collision f_adve
  bool have_moments;
  long steps;
  lbm_ibm* ibm;  // optional immersed boundary (not owned)
  double guo_a, guo_b;
};

using namespace lbm;

static int solver_collide_first(lbm_solver* sv, double* rho, double* u) {
  double* dst = sv->lat[sv->cur ^ 1];
  const double* src = sv->lat[sv->cur];
  if (sv->model == LBM_MODEL_BGK)
    return lbm_bgk_collide(dst, src, &sv->g, &sv->bc, &sv->bgk, rho, u, sv->st);
  return lbm_kbc_collide(dst, src, &sv->g, &sv->bc, &sv->kbc, rho, u, sv->st);
}
static int solver_fused(lbm_solver* sv, double* rho, double* u) {
  double* dst = sv->lat[sv->cur ^ 1];
  const double* src = sv->lat[sv->cur];
  if (sv->model == LBM_MODEL_BGK)
    return lbm_bgk_stream_collide(dst, src, &sv->g, &sv->bc, &sv->bgk, 0, sv->g.R, rho, u, sv->st);
  return lbm_kbc_stream_collide(dst, src, &sv->g, &sv->bc, &sv->kbc, 0, sv->g.R, rho, u, sv->st);
}

extern "C" {

int lbm_solver_create(lbm_solver** out, int model, const lbm_geom* g, const lbm_bc* bc,
                      const void* params, lbm_stream_t s) {
  LBM_REQUIRE(out && g && params, "lbm_solver_create: NULL argument");
  LBM_REQUIRE(model == LBM_MODEL_BGK || model == LBM_MODEL_KBC, "lbm_solver_create: model=%d", model);
  LBM_REQUIRE(g->ghost == 0, "lbm_solver_create: single block only (ghost=0)");
  int rc = validate_geom_bc("lbm_solver_create", g, bc);
  if (rc) return rc;
  lbm_solver* sv = new (std::nothrow) lbm_solver();
  LBM_REQUIRE(sv, "lbm_solver_create: out of host memory");
  sv->model = model;
  sv->g = *g;
  if (bc) sv->bc = *bc;
  else sv->bc = lbm_bc{0, 0, 0, 0, 0, 1.0, 1.0, 0.0, 0.0};
  if (model == LBM_MODEL_BGK) sv->bgk = *static_cast<const lbm_bgk_params*>(params);
  else sv->kbc = *static_cast<const lbm_kbc_params*>(params);
  sv->st = as_stream(s);
  sv->cur = 0;
  sv->post = false;
  sv->have_moments = false;
  sv->steps = 0;
  sv->ibm = nullptr;
  sv->guo_a = sv->guo_b = 0.0;
  const size_t n = (size_t)g->R * g->C;
  // Population planes are padded off their natural (often power-of-two) stride: 18 streams at
  // the same offset modulo 2^k hit the same HBM channels (+9 % MLUPS at 8192^2).
  sv->g.plane_stride = (long long)n + lbm_default_plane_pad(g->R, g->C);
  const size_t lat_doubles = (size_t)sv->g.plane_stride * 9;
  sv->lat[0] = sv->lat[1] = sv->stage = sv->rho = sv->u = nullptr;
  hipError_t e = hipMalloc(&sv->lat[0], lat_doubles * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->lat[1], lat_doubles * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->stage, n * 9 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->rho, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->u, n * 2 * sizeof(double));
  if (e != hipSuccess) {
    set_error("lbm_solver_create: hipMalloc failed: %s", hipGetErrorString(e));
    lbm_solver_destroy(sv);
    return LBM_ERR_HIP;
  }
  *out = sv;
  return LBM_OK;
}

int lbm_solver_destroy(lbm_solver* sv) {
  if (!sv) return LBM_OK;
  for (double* p : {sv->lat[0], sv->lat[1], sv->stage, sv->rho, sv->u})
    if (p) (void)hipFree(p);
  delete sv;
  return LBM_OK;
}

int lbm_solver_set_f_soa_dev(lbm_solver* sv, const double* f_dev) {
  LBM_REQUIRE(sv && f_dev, "lbm_solver_set_f_soa_dev: NULL argument");
  const size_t plane_bytes = (size_t)sv->g.R * sv->g.C * sizeof(double);
  LBM_CHECK_HIP(hipMemcpy2DAsync(sv->lat[sv->cur], (size_t)sv->g.plane_stride * sizeof(double), f_dev,
                                 plane_bytes, plane_bytes, 9, hipMemcpyDeviceToDevice, sv->st));
  sv->post = false;
  sv->have_moments = false;
  return LBM_OK;
}

int lbm_solver_set_f_aos(lbm_solver* sv, const double* f_host) {
  LBM_REQUIRE(sv && f_host, "lbm_solver_set_f_aos: NULL argument");
  const size_t bytes = (size_t)sv->g.R * sv->g.C * 9 * sizeof(double);
  LBM_CHECK_HIP(hipMemcpyAsync(sv->stage, f_host, bytes, hipMemcpyHostToDevice, sv->st));
  int rc = lbm_aos_to_soa_ex(sv->lat[sv->cur], sv->stage, sv->g.R, sv->g.C, 9, sv->g.plane_stride, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));  // f_host may be reused by the caller
  sv->post = false;
  sv->have_moments = false;
  return LBM_OK;
}

// the reference's f_adve as padded SoA: the state itself (pre-collision) or its streamed image
// in the other lattice, which is dead between steps
static int solver_f_adve(lbm_solver* sv, const double** out) {
  *out = sv->lat[sv->cur];
  if (!sv->post) return LBM_OK;
  int rc = lbm_stream(sv->lat[sv->cur ^ 1], sv->lat[sv->cur], &sv->g, &sv->bc, sv->st);
  *out = sv->lat[sv->cur ^ 1];
  return rc;
}

int lbm_solver_get_f_soa_dev(lbm_solver* sv, double* f_dev) {
  LBM_REQUIRE(sv && f_dev, "lbm_solver_get_f_soa_dev: NULL argument");
  const double* src;
  int rc = solver_f_adve(sv, &src);
  if (rc) return rc;
  const size_t plane_bytes = (size_t)sv->g.R * sv->g.C * sizeof(double);
  LBM_CHECK_HIP(hipMemcpy2DAsync(f_dev, plane_bytes, src, (size_t)sv->g.plane_stride * sizeof(double),
                                 plane_bytes, 9, hipMemcpyDeviceToDevice, sv->st));
  return LBM_OK;
}

int lbm_solver_get_f_aos(lbm_solver* sv, double* f_host) {
  LBM_REQUIRE(sv && f_host, "lbm_solver_get_f_aos: NULL argument");
  const double* src;
  int rc = solver_f_adve(sv, &src);
  if (rc) return rc;
  rc = lbm_soa_to_aos_ex(sv->stage, src, sv->g.R, sv->g.C, 9, sv->g.plane_stride, sv->st);
  if (rc) return rc;
  const size_t bytes = (size_t)sv->g.R * sv->g.C * 9 * sizeof(double);
  LBM_CHECK_HIP(hipMemcpyAsync(f_host, sv->stage, bytes, hipMemcpyDeviceToHost, sv->st));
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  return LBM_OK;
}

// periodic BGK block without per-step observers: steps can be fused several per launch
static bool solver_can_fuse_steps(const lbm_solver* sv) {
  const lbm_bc& b = sv->bc;
  auto plain = [](int m) { return m == LBM_EDGE_PERIODIC; };
  return sv->model == LBM_MODEL_BGK && !sv->ibm && !b.pressure_rows && plain(b.row_lo) &&
         plain(b.row_hi) && plain(b.col_lo) && plain(b.col_hi) && sv->g.C >= 64;
}

int lbm_solver_step(lbm_solver* sv, int n, int record_moments) {
  LBM_REQUIRE(sv && n >= 0, "lbm_solver_step: bad argument (n=%d)", n);
  const int max_depth = tuning("solver_depth", 5);
  for (int i = 0; i < n;) {
    // temporal blocking: D driver iterations in one launch (bit-identical); the iteration that
    // must record moments, and the first one on a pre-collision state, run singly
    const int fusable = n - i - (record_moments ? 1 : 0);
    int depth = fusable < max_depth ? fusable : max_depth;
    while (depth >= 2 && sv->g.R < 4 * depth + 8) --depth;
    if (sv->post && depth >= 2 && solver_can_fuse_steps(sv)) {
      int rc = lbm_bgk_stream_collide_xn(sv->lat[sv->cur ^ 1], sv->lat[sv->cur], &sv->g, &sv->bc,
                                         &sv->bgk, depth, 0, sv->g.R, sv->st);
      if (rc) return rc;
      sv->cur ^= 1;
      sv->steps += depth;
      i += depth;
      continue;
    }
    {
    // with an immersed boundary every step needs this step's rho, u (cylinder_test.cpp:110)
    const bool rec = (record_moments && i == n - 1) || sv->ibm;
    int rc = sv->post ? solver_fused(sv, rec ? sv->rho : nullptr, rec ? sv->u : nullptr)
                      : solver_collide_first(sv, rec ? sv->rho : nullptr, rec ? sv->u : nullptr);
    if (rc) return rc;
    if (sv->ibm) {  // :110-127: F = ib.eulerian_force_density(u, rho); f_coll[ROI] += S(u, F)
      rc = lbm_ibm_force(sv->ibm, sv->u, sv->rho, nullptr, sv->st);
      if (rc) return rc;
      rc = lbm_ibm_add_source(sv->ibm, sv->lat[sv->cur ^ 1], &sv->g, sv->u, sv->bgk.omega,
                              sv->guo_a, sv->guo_b, sv->st);
      if (rc) return rc;
    }
    sv->cur ^= 1;
    sv->post = true;
    if (rec) sv->have_moments = true;
    ++sv->steps;
    ++i;
    }
  }
  return LBM_OK;
}

int lbm_solver_get_moments_aos(lbm_solver* sv, double* rho_host, double* u_host) {
  LBM_REQUIRE(sv && rho_host && u_host, "lbm_solver_get_moments_aos: NULL argument");
  if (!sv->have_moments) {
    set_error("lbm_solver_get_moments_aos: no step(.., record_moments=1) since the last set_f");
    return LBM_ERR_STATE;
  }
  const size_t n = (size_t)sv->g.R * sv->g.C;
  LBM_CHECK_HIP(hipMemcpyAsync(rho_host, sv->rho, n * sizeof(double), hipMemcpyDeviceToHost, sv->st));
  int rc = lbm_soa_to_aos(sv->stage, sv->u, sv->g.R, sv->g.C, 2, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipMemcpyAsync(u_host, sv->stage, n * 2 * sizeof(double), hipMemcpyDeviceToHost, sv->st));
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  return LBM_OK;
}

int lbm_solver_attach_ibm(lbm_solver* sv, lbm_ibm* ib, double guo_a, double guo_b) {
  LBM_REQUIRE(sv, "lbm_solver_attach_ibm: NULL solver");
  LBM_REQUIRE(!ib || sv->model == LBM_MODEL_BGK, "lbm_solver_attach_ibm: BGK solvers only");
  if (ib) {
    int r0, r1, c0, c1;
    lbm_ibm_roi(ib, &r0, &r1, &c0, &c1);
    LBM_REQUIRE(r0 >= 1 && c0 >= 1 && r1 <= sv->g.R - 1 && c1 <= sv->g.C - 1,
                "lbm_solver_attach_ibm: ROI must lie strictly inside the lattice");
  }
  sv->ibm = ib;
  sv->guo_a = guo_a;
  sv->guo_b = guo_b;
  return LBM_OK;
}

int lbm_solver_sync(lbm_solver* sv) {
  LBM_REQUIRE(sv, "lbm_solver_sync: NULL solver");
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  return LBM_OK;
}

int lbm_solver_lattices(lbm_solver* sv, double** cur, double** other, lbm_geom* geom) {
  LBM_REQUIRE(sv && cur && other, "lbm_solver_lattices: NULL argument");
  *cur = sv->lat[sv->cur];
  *other = sv->lat[sv->cur ^ 1];
  if (geom) *geom = sv->g;
  return LBM_OK;
}

}  // extern "C"
