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

static int check_kbc(const char* fn, const lbm_kbc_params* prm) {
  LBM_REQUIRE(prm, "%s: NULL params", fn);
  LBM_REQUIRE(prm->s2 > 0.0 && prm->s2 <= 2.0, "%s: s2=%g outside (0, 2]", fn, prm->s2);
  return LBM_OK;
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
  return launch_collide_only("lbm_kbc_collide", p, f, g, bc, KbcModel{prm->s2}, rho, u, as_stream(s));
}

int lbm_kbc_stream_collide(double* p_new, const double* p_old, const lbm_geom* g,
                           const lbm_bc* bc, const lbm_kbc_params* prm, int row_begin,
                           int row_end, double* rho, double* u, lbm_stream_t s) {
  int rc = check_kbc("lbm_kbc_stream_collide", prm);
  if (rc) return rc;
  return launch_stream_collide("lbm_kbc_stream_collide", p_new, p_old, g, bc, KbcModel{prm->s2},
                               row_begin, row_end, rho, u, as_stream(s));
}

}  // extern "C"
