// C ABI, part 3: solver context = the driver loop of the reference's test/*.cpp mains
// (single block).  Model dispatch lives here; kernels are in d2q9.hpp / kbc.hpp.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

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
  long blocks = 0;  // multi-step blocks launched so far (lbm_solver_block_launches)
  lbm_ibm* ibm;  // optional immersed boundary (not owned)
  double guo_a, guo_b;
  bool given_moments = false;  // first iteration collides on sv->rho / sv->u as set by the caller
  hipStream_t side = nullptr;  // forcing chain of the immersed boundary, beside the lattice update
  hipEvent_t ev_roi = nullptr, ev_ibm = nullptr;
  double* band = nullptr;      // third lattice: odd / even steps of the forced band (solver_ibm_block)
  // the forced BOX (solver_ibm_block): ROI +- 2 D rows and columns as a small lattice pair of its own
  double* box[2] = {nullptr, nullptr};
  double *box_rho = nullptr, *box_u = nullptr;
  long long box_plane = 0;
  int box_rows_max = 0, box_cols_max = 0;
  hipStream_t far_st = nullptr;  // the D-step window over the whole lattice, beside the box chain
  hipEvent_t ev_far_fork = nullptr, ev_far_join = nullptr;
  // pressure-periodic rows at multi-step speed (solver_pressure_block): two small lattices of 4 D rows
  // holding the rows on both sides of the virtual rows, advanced in single steps on a helper stream
  double* seam[2] = {nullptr, nullptr};
  long long seam_plane = 0;
  hipStream_t seam_st = nullptr;
  hipEvent_t ev_seam_fork = nullptr, ev_seam_join = nullptr;
};
static constexpr int kSeamMaxDepth = 5;

using namespace lbm;

static int solver_collide_first(lbm_solver* sv, double* rho, double* u) {
  double* dst = sv->lat[sv->cur ^ 1];
  const double* src = sv->lat[sv->cur];
  if (sv->model == LBM_MODEL_BGK)
    return lbm_bgk_collide(dst, src, &sv->g, &sv->bc, &sv->bgk, rho, u, sv->st);
  if (sv->given_moments) {  // the driver's held m0 / m1 (ulbm_poiseuille.cpp:85-86); rho, u stay as given
    sv->given_moments = false;
    return lbm_kbc_collide_first(dst, src, sv->rho, sv->u, &sv->g, &sv->bc, &sv->kbc, sv->st);
  }
  return lbm_kbc_collide(dst, src, &sv->g, &sv->bc, &sv->kbc, rho, u, sv->st);
}
// with_ibm_overlap: the moments are wanted for the immersed boundary alone -> the ROI rows go first
// (rho, u written there only: saves the 24 B per node of a full store), the forcing + source
// (lbm_ibm_step, one latency-bound workgroup) runs on a side stream while the main stream updates
// the rows above and below the ROI.  Returns with the forcing joined back into sv->st.
static int solver_fused(lbm_solver* sv, double* rho, double* u, bool with_ibm_overlap = false) {
  double* dst = sv->lat[sv->cur ^ 1];
  const double* src = sv->lat[sv->cur];
  if (sv->model == LBM_MODEL_BGK && with_ibm_overlap && sv->ibm && sv->side) {
    int q0, q1, c0, c1;
    int rc = lbm_ibm_roi(sv->ibm, &q0, &q1, &c0, &c1);
    if (!rc) rc = lbm_bgk_stream_collide(dst, src, &sv->g, &sv->bc, &sv->bgk, q0, q1, rho, u, sv->st);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(sv->ev_roi, sv->st));
    LBM_CHECK_HIP(hipStreamWaitEvent(sv->side, sv->ev_roi, 0));
    rc = lbm_ibm_step(sv->ibm, dst, &sv->g, u, rho, sv->bgk.omega, sv->guo_a, sv->guo_b, sv->side);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(sv->ev_ibm, sv->side));
    rc = ibm_gate(sv->ibm, sv->st);  // the lattice launches below must not take the forcing workgroup's CU first
    if (rc) return rc;
    rc = lbm_bgk_stream_collide(dst, src, &sv->g, &sv->bc, &sv->bgk, q1, sv->g.R, nullptr, nullptr, sv->st);
    if (!rc) rc = lbm_bgk_stream_collide(dst, src, &sv->g, &sv->bc, &sv->bgk, 0, q0, nullptr, nullptr, sv->st);
    if (rc) return rc;
    LBM_CHECK_HIP(hipStreamWaitEvent(sv->st, sv->ev_ibm, 0));
    return LBM_OK;
  }
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
  LBM_REQUIRE(g->row_pitch == 0 || g->row_pitch == g->C, "lbm_solver_create: dense rows only (row_pitch = %d)", g->row_pitch);
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
  sv->given_moments = false;
  sv->have_moments = false;
  sv->steps = 0;
  sv->ibm = nullptr;
  sv->guo_a = sv->guo_b = 0.0;
  const size_t n = (size_t)g->R * g->C;
  // Population planes are padded off their natural (often power-of-two) stride: 18 streams at
  // the same offset modulo 2^k hit the same HBM channels (+9 % MLUPS at 8192^2).
  // KBC solvers pad their ROWS too (lbm_default_row_pitch: +3 % at 4096 columns; the BGK window gains nothing and the
  // BGK solver's immersed-boundary / checkpoint paths keep dense rows).  Everything that touches the lattices takes
  // &sv->g or goes through solver_copy_planes below.
  const int pitch = model == LBM_MODEL_KBC ? lbm_default_row_pitch(g->C) : g->C;
  sv->g.row_pitch = pitch > g->C ? pitch : 0;
  sv->g.plane_stride = (long long)g->R * pitch + lbm_default_plane_pad(g->R, pitch);
  const size_t lat_doubles = (size_t)sv->g.plane_stride * 9;
  sv->lat[0] = sv->lat[1] = sv->stage = sv->rho = sv->u = nullptr;
  hipError_t e = hipMalloc(&sv->lat[0], lat_doubles * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->lat[1], lat_doubles * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->stage, n * 9 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->rho, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sv->u, n * 2 * sizeof(double));
  if (e == hipSuccess && sv->bc.pressure_rows && g->C >= 64 && g->R >= 6 * 2 + 8) {  // solver_pressure_block
    const int rows = 4 * kSeamMaxDepth;
    sv->seam_plane = (long long)rows * g->C + 1088;  // off the power-of-two stride, 64-byte aligned
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
      e = hipMalloc(&sv->seam[k], (size_t)sv->seam_plane * 9 * sizeof(double));
      if (e == hipSuccess) e = hipMemsetAsync(sv->seam[k], 0, (size_t)sv->seam_plane * 9 * sizeof(double), sv->st);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&sv->seam_st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sv->ev_seam_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sv->ev_seam_join, hipEventDisableTiming);
  }
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
  if (sv->seam_st) {
    (void)hipStreamSynchronize(sv->seam_st);
    (void)hipStreamDestroy(sv->seam_st);
  }
  if (sv->ev_seam_fork) (void)hipEventDestroy(sv->ev_seam_fork);
  if (sv->ev_seam_join) (void)hipEventDestroy(sv->ev_seam_join);
  for (double* p : {sv->lat[0], sv->lat[1], sv->stage, sv->rho, sv->u, sv->band, sv->seam[0], sv->seam[1], sv->box[0], sv->box[1],
                    sv->box_rho, sv->box_u})
    if (p) (void)hipFree(p);
  if (sv->far_st) {
    (void)hipStreamSynchronize(sv->far_st);
    (void)hipStreamDestroy(sv->far_st);
  }
  if (sv->ev_far_fork) (void)hipEventDestroy(sv->ev_far_fork);
  if (sv->ev_far_join) (void)hipEventDestroy(sv->ev_far_join);
  if (sv->side) {
    (void)hipStreamSynchronize(sv->side);
    (void)hipStreamDestroy(sv->side);
  }
  if (sv->ev_roi) (void)hipEventDestroy(sv->ev_roi);
  if (sv->ev_ibm) (void)hipEventDestroy(sv->ev_ibm);
  delete sv;
  return LBM_OK;
}

int lbm_solver_set_moments_aos(lbm_solver* sv, const double* rho_host, const double* u_host) {
  LBM_REQUIRE(sv && rho_host && u_host, "lbm_solver_set_moments_aos: NULL argument");
  LBM_REQUIRE(sv->model == LBM_MODEL_KBC && !sv->post,
              "lbm_solver_set_moments_aos: KBC solvers only, after set_f and before the first step");
  const size_t n = (size_t)sv->g.R * sv->g.C;
  LBM_CHECK_HIP(hipMemcpyAsync(sv->rho, rho_host, n * sizeof(double), hipMemcpyHostToDevice, sv->st));
  LBM_CHECK_HIP(hipMemcpyAsync(sv->stage, u_host, 2 * n * sizeof(double), hipMemcpyHostToDevice, sv->st));
  int rc = lbm_aos_to_soa(sv->u, sv->stage, sv->g.R, sv->g.C, 2, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  sv->given_moments = true;
  return LBM_OK;
}

// rows [row0, row0 + n_rows) of all 9 planes between the solver's (padded) lattice and a DENSE array of planes [9][rows_dense][C]
// starting at its row `drow0` -- through the node-addressed copy kernel, so that neither side's strides matter
static int solver_copy_planes(lbm_solver* sv, double* lattice, bool to_lattice, double* dense, long long dense_plane, int rows_dense,
                              int drow0, int row0, int n_rows, hipStream_t st) {
  const lbm_geom dg{rows_dense, sv->g.C, 0, dense_plane, 0};
  if (to_lattice) return box_copy(lattice, sv->g, row0, 0, dense, dg, drow0, 0, n_rows, sv->g.C, st);
  return box_copy(dense, dg, drow0, 0, lattice, sv->g, row0, 0, n_rows, sv->g.C, st);
}

int lbm_solver_set_f_soa_dev(lbm_solver* sv, const double* f_dev) {
  LBM_REQUIRE(sv && f_dev, "lbm_solver_set_f_soa_dev: NULL argument");
  const size_t plane_bytes = (size_t)sv->g.R * sv->g.C * sizeof(double);
  if (sv->g.row_pitch) {
    int rc = solver_copy_planes(sv, sv->lat[sv->cur], true, const_cast<double*>(f_dev), 0, sv->g.R, 0, 0, sv->g.R, sv->st);
    if (rc) return rc;
  } else
  LBM_CHECK_HIP(hipMemcpy2DAsync(sv->lat[sv->cur], (size_t)sv->g.plane_stride * sizeof(double), f_dev,
                                 plane_bytes, plane_bytes, 9, hipMemcpyDeviceToDevice, sv->st));
  sv->post = false;
  sv->given_moments = false;
  sv->have_moments = false;
  return LBM_OK;
}

int lbm_solver_set_f_aos(lbm_solver* sv, const double* f_host) {
  LBM_REQUIRE(sv && f_host, "lbm_solver_set_f_aos: NULL argument");
  const size_t bytes = (size_t)sv->g.R * sv->g.C * 9 * sizeof(double);
  LBM_CHECK_HIP(hipMemcpyAsync(sv->stage, f_host, bytes, hipMemcpyHostToDevice, sv->st));
  int rc = lbm_aos_to_soa_pitched(sv->lat[sv->cur], sv->stage, sv->g.R, sv->g.C, 9, sv->g.plane_stride, sv->g.row_pitch, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));  // f_host may be reused by the caller
  sv->post = false;
  sv->given_moments = false;
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
  if (sv->g.row_pitch) return solver_copy_planes(sv, const_cast<double*>(src), false, f_dev, 0, sv->g.R, 0, 0, sv->g.R, sv->st);
  LBM_CHECK_HIP(hipMemcpy2DAsync(f_dev, plane_bytes, src, (size_t)sv->g.plane_stride * sizeof(double),
                                 plane_bytes, 9, hipMemcpyDeviceToDevice, sv->st));
  return LBM_OK;
}

int lbm_solver_get_f_aos(lbm_solver* sv, double* f_host) {
  LBM_REQUIRE(sv && f_host, "lbm_solver_get_f_aos: NULL argument");
  const double* src;
  int rc = solver_f_adve(sv, &src);
  if (rc) return rc;
  rc = lbm_soa_to_aos_pitched(sv->stage, src, sv->g.R, sv->g.C, 9, sv->g.plane_stride, sv->g.row_pitch, sv->st);
  if (rc) return rc;
  const size_t bytes = (size_t)sv->g.R * sv->g.C * 9 * sizeof(double);
  LBM_CHECK_HIP(hipMemcpyAsync(f_host, sv->stage, bytes, hipMemcpyDeviceToHost, sv->st));
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  return LBM_OK;
}

// D steps of a block with an immersed boundary.  The forcing changes every step, but only inside the
// ROI rows [q0, q1): rows at least D away from them see plain BGK for D steps and take the multi-step
// window (two launches, rows above / below), while a band around the ROI advances D single forced steps
// on a trapezoid that loses one row per side and step -- step k computes rows [q0 - 2D + k, q1 + 2D - k)
// from step k-1, so that after D steps rows [q0 - D, q1 + D) are valid and everything the band ever
// read outside itself came from the time-t lattice.  The band alternates between lat[other] and a third
// lattice, ending in lat[other]; the far launches come LAST (they read the time-t lattice only and
// overwrite the rows the wider early band steps left behind).  Same kernels per node as D single steps:
// same bits.  Returns 1 if the block does not qualify (caller falls back to single steps).
static int solver_ibm_block(lbm_solver* sv, int D) {
  int q0, q1, c0, c1;
  int rc = lbm_ibm_roi(sv->ibm, &q0, &q1, &c0, &c1);
  if (rc) return rc;
  const int R = sv->g.R;
  // the far rows need edges the multi-step window carries (periodic, bounce-back, specular, velocity)
  auto carried = [](int m) { return m == LBM_EDGE_PERIODIC || bc_is_wall(m); };
  const lbm_bc& b = sv->bc;
  if (b.pressure_rows || !carried(b.row_lo) || !carried(b.row_hi) || !carried(b.col_lo) || !carried(b.col_hi) ||
      sv->bgk.force_mode || bc_mixed_axis(make_bc(&b)))
    return 1;
  if (D < 2 || q0 - 2 * D < 2 || q1 + 2 * D > R - 2 || R < 4 * D + 8 || sv->g.C < 64) return 1;
  if (!sv->band) return 1;  // allocated by lbm_solver_attach_ibm (no allocation inside a step call)
  const double* src = sv->lat[sv->cur];
  double* dst = sv->lat[sv->cur ^ 1];
  // The forced BOX.  The forcing reaches a node only through the ROI: columns at least 2 D away from it need the single
  // steps as little as rows that far away do.  So the trapezoid is cut in both directions -- rows and columns ROI +- 2 D
  // (columns widened to multiples of 8) copied into a small periodic lattice pair, D forced single steps there (what its
  // wrap spoils is the frame that is dropped anyway), the box ROI +- D copied back -- and the D-step window runs over
  // ALL rows from the time-t lattice, on a stream of its own beside the chain of small launches on the caller's stream.
  // ("ibm_box" = 0: the full-width band below; also taken when the box would touch a wall column.)
  const int bc0 = (c0 - 2 * D) / 8 * 8, bc1 = (c1 + 2 * D + 7) / 8 * 8;
  if (sv->box[0] && sv->far_st && tuning("ibm_box", 1) && c0 - 2 * D >= 8 && bc1 <= sv->g.C - 1 && bc1 - bc0 <= sv->box_cols_max &&
      q1 - q0 + 4 * D <= sv->box_rows_max) {
    const int Rb = q1 - q0 + 4 * D, Cb = bc1 - bc0, br0 = q0 - 2 * D;
    const lbm_geom bg{Rb, Cb, 0, sv->box_plane};
    const lbm_bc pb{LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, 0, 1.0, 1.0, 0.0, 0.0};
    const bool beside = tuning("ibm_box_overlap", 1) != 0;
    hipStream_t far = beside ? sv->far_st : sv->st;
    if (beside) LBM_CHECK_HIP(hipEventRecord(sv->ev_far_fork, sv->st));
    rc = box_copy(sv->box[0], bg, 0, 0, src, sv->g, br0, bc0, Rb, Cb, sv->st);
    if (rc) return rc;
    int cur = 0;
    // "ibm_chain_kernel" = 1 (opt-in): the chain as ONE launch of a few workgroups on compute units of their own, the
    // window launch held back until they are resident.  Bit-identical; measured level with the 3 D small launches
    // (73 / 87 / 118 k against 77 / 98 / 121 k MLUPS at 2048 / 4096 / 16384 rows): what it gains in isolation it loses to
    // coherent (L2-bypassing) accesses and 3 D grid barriers -- DESIGN 5.3
    rc = tuning("ibm_chain_kernel", 0) ? ibm_box_chain(sv->ibm, br0, bc0, sv->box, &cur, &bg, &sv->bgk, bgk_uses_fast_model(&sv->bgk, &pb), D,
                                                       sv->box_rho, sv->box_u, sv->guo_a, sv->guo_b, sv->st)
                                       : 1;
    if (rc < 0) return rc;
    const bool one_launch = rc == 0;
    if (beside) {
      LBM_CHECK_HIP(hipStreamWaitEvent(far, sv->ev_far_fork, 0));
      if (one_launch) rc = ibm_gate(sv->ibm, far);
      else rc = LBM_OK;
      // (round 4, measured and not kept: the window as PERSISTENT waves that leave 32 .. 256 wave slots of the card free for
      // the chain from its first cycle to its last -- 143 - 150 k MLUPS against 148.5 / 149.3 k with every slot taken, and
      // 115 - 118 k against 124 - 128 k in the reference order: the chain is not waiting for slots.  profiles/r04_ibm_reserve.txt)
      if (!rc) rc = lbm_bgk_stream_collide_xn(dst, src, &sv->g, &sv->bc, &sv->bgk, D, 0, R, far);
      if (rc) return rc;
      LBM_CHECK_HIP(hipEventRecord(sv->ev_far_join, far));
    }
    for (int k = 1; k <= D && !one_launch; ++k) {
      rc = lbm_bgk_stream_collide(sv->box[cur ^ 1], sv->box[cur], &bg, &pb, &sv->bgk, k, Rb - k, sv->box_rho, sv->box_u, sv->st);
      if (!rc) rc = ibm_step_window(sv->ibm, br0, bc0, sv->box[cur ^ 1], &bg, sv->box_u, sv->box_rho, sv->bgk.omega, sv->guo_a, sv->guo_b, sv->st);
      if (rc) return rc;
      cur ^= 1;
    }
    if (beside) LBM_CHECK_HIP(hipStreamWaitEvent(sv->st, sv->ev_far_join, 0));
    else rc = lbm_bgk_stream_collide_xn(dst, src, &sv->g, &sv->bc, &sv->bgk, D, 0, R, sv->st);
    if (!rc) rc = box_copy(dst, sv->g, q0 - D, bc0 + D, sv->box[cur], bg, D, D, Rb - 2 * D, Cb - 2 * D, sv->st);
    return rc;
  }
  const double* in = src;
  for (int k = 1; k <= D; ++k) {
    double* out = ((D - k) % 2 == 0) ? dst : sv->band;
    const int lo = q0 - 2 * D + k, hi = q1 + 2 * D - k;
    // the whole band in ONE launch (rho, u are written for its rows outside the ROI too: harmless, and
    // four launches fewer per step), then the forcing on the same stream: beside its one workgroup
    // there is nothing left to run, and a cross-stream dependency costs more than it could hide
    rc = lbm_bgk_stream_collide(out, in, &sv->g, &sv->bc, &sv->bgk, lo, hi, sv->rho, sv->u, sv->st);
    if (!rc) rc = lbm_ibm_step(sv->ibm, out, &sv->g, sv->u, sv->rho, sv->bgk.omega, sv->guo_a, sv->guo_b, sv->st);
    if (rc) return rc;
    in = out;
  }
  rc = lbm_bgk_stream_collide_xn(dst, src, &sv->g, &sv->bc, &sv->bgk, D, 0, q0 - D, sv->st);
  if (!rc) rc = lbm_bgk_stream_collide_xn(dst, src, &sv->g, &sv->bc, &sv->bgk, D, q1 + D, R, sv->st);
  return rc;
}

// D steps of a block with PRESSURE-PERIODIC rows (horizontal_poiseuille_test.cpp:25-45, ulbm_poiseuille.cpp:
// 36-58).  The virtual rows 0 / R-1 are rewritten every step from the collision of rows R-2 / 1, so no
// multi-step window can carry them (row R-2 of a level would have to be known before row 1 of the level
// below).  But a forced value of row 0 at level l reaches row j only at level l + j: rows [D, R - D) see
// plain arithmetic for D steps and take the multi-step window from the time-t lattice, while the 2 D rows
// on either side of the seam are copied into a small periodic lattice of 4 D rows (rows [0, 2D) first,
// rows [R - 2D, R) behind them: its own wrap IS the seam, its virtual rows are ITS rows 0 / 4D - 1) and
// advance D ordinary single steps there -- the artificial seam in its middle spoils one more row per
// side and step, leaving rows [0, D) and [3D, 4D) valid, which are copied back.  That chain (3 small
// launches per step) runs on a helper stream beside the window launch.  Same kernels per node as D
// single steps: same bits.  Returns 1 if the block does not qualify (caller falls back to single steps).
static int solver_pressure_block(lbm_solver* sv, int D) {
  const lbm_bc& b = sv->bc;
  const int R = sv->g.R, C = sv->g.C;
  auto col_ok = [](int m) { return m == LBM_EDGE_PERIODIC || bc_is_wall(m); };
  if (!sv->seam[0] || !sv->seam_st || D < 2 || D > kSeamMaxDepth) return 1;
  if (b.row_lo != LBM_EDGE_PERIODIC || b.row_hi != LBM_EDGE_PERIODIC || !col_ok(b.col_lo) || !col_ok(b.col_hi)) return 1;
  if (bc_mixed_axis(make_bc(&b)) || C < 64 || R < 6 * D + 8) return 1;
  if (sv->model == LBM_MODEL_KBC && D > 2) return 1;
  const double* src = sv->lat[sv->cur];
  double* dst = sv->lat[sv->cur ^ 1];
  const int Rb = 4 * D;
  lbm_geom sg{Rb, C, 0, sv->seam_plane};
  const size_t spitch = (size_t)sv->g.plane_stride * sizeof(double), dpitch = (size_t)sv->seam_plane * sizeof(double);
  const size_t half = (size_t)2 * D * C * sizeof(double);
  hipStream_t ss = sv->seam_st;
  LBM_CHECK_HIP(hipEventRecord(sv->ev_seam_fork, sv->st));
  LBM_CHECK_HIP(hipStreamWaitEvent(ss, sv->ev_seam_fork, 0));
  // rows [0, 2D) -> small rows [0, 2D); rows [R - 2D, R) -> small rows [2D, 4D)
  if (sv->g.row_pitch) {  // padded rows: the node-addressed copy
    int rcp = solver_copy_planes(sv, const_cast<double*>(src), false, sv->seam[0], sv->seam_plane, Rb, 0, 0, 2 * D, ss);
    if (!rcp) rcp = solver_copy_planes(sv, const_cast<double*>(src), false, sv->seam[0], sv->seam_plane, Rb, 2 * D, R - 2 * D, 2 * D, ss);
    if (rcp) return rcp;
  } else {
  LBM_CHECK_HIP(hipMemcpy2DAsync(sv->seam[0], dpitch, src, spitch, half, 9, hipMemcpyDeviceToDevice, ss));
  LBM_CHECK_HIP(hipMemcpy2DAsync(sv->seam[0] + (size_t)2 * D * C, dpitch, src + (size_t)(R - 2 * D) * C, spitch, half, 9,
                                 hipMemcpyDeviceToDevice, ss));
  }
  int cur = 0, rc = LBM_OK;
  for (int k = 0; k < D && !rc; ++k, cur ^= 1)
    rc = sv->model == LBM_MODEL_BGK
             ? lbm_bgk_stream_collide(sv->seam[cur ^ 1], sv->seam[cur], &sg, &sv->bc, &sv->bgk, 0, Rb, nullptr, nullptr, ss)
             : lbm_kbc_stream_collide(sv->seam[cur ^ 1], sv->seam[cur], &sg, &sv->bc, &sv->kbc, 0, Rb, nullptr, nullptr, ss);
  if (rc) return rc;
  const size_t part = (size_t)D * C * sizeof(double);
  if (sv->g.row_pitch) {
    int rcp = solver_copy_planes(sv, dst, true, sv->seam[cur], sv->seam_plane, Rb, 0, 0, D, ss);
    if (!rcp) rcp = solver_copy_planes(sv, dst, true, sv->seam[cur], sv->seam_plane, Rb, 3 * D, R - D, D, ss);
    if (rcp) return rcp;
  } else {
  LBM_CHECK_HIP(hipMemcpy2DAsync(dst, spitch, sv->seam[cur], dpitch, part, 9, hipMemcpyDeviceToDevice, ss));
  LBM_CHECK_HIP(hipMemcpy2DAsync(dst + (size_t)(R - D) * C, spitch, sv->seam[cur] + (size_t)3 * D * C, dpitch, part, 9,
                                 hipMemcpyDeviceToDevice, ss));
  }
  LBM_CHECK_HIP(hipEventRecord(sv->ev_seam_join, ss));
  // the far rows: plain multi-step window (walls on the columns included), no pressure rows
  lbm_bc far = sv->bc;
  far.pressure_rows = 0;
  rc = sv->model == LBM_MODEL_BGK
           ? bgk_stream_collide_xn_ref(dst, src, &sv->g, &far, &sv->bgk, D, D, R - D, sv->st)
           : kbc_stream_collide_x2_ref(dst, src, &sv->g, &far, &sv->kbc, D, R - D, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipStreamWaitEvent(sv->st, sv->ev_seam_join, 0));
  return LBM_OK;
}

// periodic or wall-bounded block without per-step observers: steps can be fused several per launch
static bool solver_can_fuse_steps(const lbm_solver* sv) {
  const lbm_bc& b = sv->bc;
  // periodic edges, or the wall types the sliding window carries (bounce-back, specular, velocity)
  const bool bgk = sv->model == LBM_MODEL_BGK;
  auto plain = [](int m) { return m == LBM_EDGE_PERIODIC || bc_is_wall(m); };
  const bool model_ok = bgk || (sv->model == LBM_MODEL_KBC && kbc_uses_fast_model(&sv->kbc));
  return model_ok && !sv->ibm && !b.pressure_rows && !bc_mixed_axis(make_bc(&b)) && plain(b.row_lo) && plain(b.row_hi) &&
         plain(b.col_lo) && plain(b.col_hi) && sv->g.C >= 64;
}

int lbm_solver_step(lbm_solver* sv, int n, int record_moments) {
  LBM_REQUIRE(sv && n >= 0, "lbm_solver_step: bad argument (n=%d)", n);
  // steps fused per launch: 5 on periodic BGK blocks, 4 when the window carries walls (its register
  // ring plus the fix-ups: 365 VGPRs at 4, 442 at 5; measured 104-108 k vs 94-103 k MLUPS); KBC: 4 on periodic blocks (LDS ring,
  // round 4: 87.4 - 89.6 k against 86.0 k with 3 at 4096^2, the launch being power-bound the saved bytes count), 3 with walls
  const lbm_bc& bb = sv->bc;
  const bool walled = bc_is_wall(bb.row_lo) || bc_is_wall(bb.row_hi) || bc_is_wall(bb.col_lo) || bc_is_wall(bb.col_hi);
  // (the reference-order BGK collision on a PERIODIC block carries more live values per level: 4 steps per launch read 132.6 k MLUPS at
  // 8192^2 against 127.1 k with 5, 99.0 k with 3, 120.3 k with 6; the reassociated default 5: 177 / 171 / 143 k at 5 / 6 / 4)
  const bool ref_order_bgk = sv->model == LBM_MODEL_BGK && !bgk_uses_fast_model(&sv->bgk, &sv->bc);
  int max_depth = sv->model == LBM_MODEL_KBC ? tuning("kbc_depth", 4) : tuning("solver_depth", ref_order_bgk && !sv->ibm && !walled && !bb.pressure_rows ? 4 : 5);
  if (walled && sv->model == LBM_MODEL_BGK && max_depth > tuning("solver_depth_walls", 5)) max_depth = tuning("solver_depth_walls", 5);
  if (walled && sv->model == LBM_MODEL_KBC && max_depth > 3) max_depth = 3;
  for (int i = 0; i < n;) {
    // temporal blocking: D driver iterations in one launch (bit-identical); the iteration that
    // must record moments, and the first one on a pre-collision state, run singly
    const int fusable = n - i - (record_moments ? 1 : 0);
    int depth = fusable < max_depth ? fusable : max_depth;
    while (depth >= 2 && sv->g.R < 4 * depth + 8) --depth;
    if (sv->post && depth >= 2 && solver_can_fuse_steps(sv)) {
      int rc = sv->model == LBM_MODEL_KBC
                   ? lbm_kbc_stream_collide_xn(sv->lat[sv->cur ^ 1], sv->lat[sv->cur], &sv->g, &sv->bc,
                                               &sv->kbc, depth > 4 ? 4 : depth, 0, sv->g.R, sv->st)
                   : lbm_bgk_stream_collide_xn(sv->lat[sv->cur ^ 1], sv->lat[sv->cur], &sv->g, &sv->bc,
                                               &sv->bgk, depth, 0, sv->g.R, sv->st);
      if (rc) return rc;
      if (sv->model == LBM_MODEL_KBC && depth > 4) depth = 4;
      sv->cur ^= 1;
      ++sv->blocks;
      sv->steps += depth;
      i += depth;
      continue;
    }
    if (sv->post && sv->bc.pressure_rows && !sv->ibm && depth >= 2 && tuning("pressure_depth", sv->model == LBM_MODEL_KBC ? 2 : 5) >= 2) {
      int d = tuning("pressure_depth", sv->model == LBM_MODEL_KBC ? 2 : 5);
      if (d > depth) d = depth;
      if (d > kSeamMaxDepth) d = kSeamMaxDepth;
      if (sv->model == LBM_MODEL_KBC && d > 2) d = 2;
      while (d >= 2 && sv->g.R < 6 * d + 8) --d;
      const int rc = d >= 2 ? solver_pressure_block(sv, d) : 1;
      if (rc < 0) return rc;
      if (rc == 0) {
        sv->cur ^= 1;
        ++sv->blocks;
        sv->steps += d;
        sv->have_moments = false;
        i += d;
        continue;
      }
    }
    if (sv->post && sv->ibm && sv->side && sv->model == LBM_MODEL_BGK && depth >= 2 && tuning("ibm_depth", 5) >= 2) {
      const int d = depth < tuning("ibm_depth", 5) ? depth : tuning("ibm_depth", 5);
      const int rc = solver_ibm_block(sv, d);
      if (rc < 0) return rc;
      if (rc == 0) {
        sv->cur ^= 1;
        ++sv->blocks;
        sv->steps += d;
        sv->have_moments = false;  // rho, u hold the ROI rows of the last step only
        i += d;
        continue;
      }
    }
    {
    // with an immersed boundary every step needs this step's rho, u (cylinder_test.cpp:110)
    const bool want = record_moments && i == n - 1;  // the caller's full-field moments
    const bool first = !sv->post;                    // the collide-only step writes full fields
    const bool rec = want || sv->ibm;
    const bool overlap = sv->post && sv->ibm && sv->side && !want;  // forcing done inside solver_fused
    int rc = sv->post ? solver_fused(sv, rec ? sv->rho : nullptr, rec ? sv->u : nullptr, overlap)
                      : solver_collide_first(sv, rec ? sv->rho : nullptr, rec ? sv->u : nullptr);
    if (rc) return rc;
    if (sv->ibm && !overlap) {  // :110-127: F = ib.eulerian_force_density(u, rho); f_coll[ROI] += S(u, F)
      rc = lbm_ibm_step(sv->ibm, sv->lat[sv->cur ^ 1], &sv->g, sv->u, sv->rho, sv->bgk.omega,
                        sv->guo_a, sv->guo_b, sv->st);
      if (rc) return rc;
    }
    sv->cur ^= 1;
    sv->post = true;
    if (rec) sv->have_moments = want || first || !overlap;  // a ROI-windowed step leaves the full fields stale
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
  if (ib && !sv->side) {
    LBM_CHECK_HIP(hipStreamCreateWithFlags(&sv->side, hipStreamNonBlocking));
    LBM_CHECK_HIP(hipEventCreateWithFlags(&sv->ev_roi, hipEventDisableTiming));
    LBM_CHECK_HIP(hipEventCreateWithFlags(&sv->ev_ibm, hipEventDisableTiming));
  }
  if (ib && !sv->band && sv->model == LBM_MODEL_BGK) {  // third lattice of the 5-step blocks (solver_ibm_block)
    const size_t bytes = (size_t)sv->g.plane_stride * 9 * sizeof(double);
    LBM_CHECK_HIP(hipMalloc(&sv->band, bytes));
    LBM_CHECK_HIP(hipMemsetAsync(sv->band, 0, bytes, sv->st));
  }
  if (ib && sv->model == LBM_MODEL_BGK) {  // the forced box of the multi-step blocks, sized for the deepest block
    int r0, r1, c0, c1;
    lbm_ibm_roi(ib, &r0, &r1, &c0, &c1);
    const int Dm = 5;
    int rows = r1 - r0 + 4 * Dm, cols = (c1 - c0 + 4 * Dm + 16) / 8 * 8;
    if (rows > sv->box_rows_max || cols > sv->box_cols_max) {
      if (sv->box[0]) {  // a larger boundary than the one attached before: nothing may still be running on the old box
        LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
        if (sv->far_st) LBM_CHECK_HIP(hipStreamSynchronize(sv->far_st));
      }
      for (double** p : {&sv->box[0], &sv->box[1], &sv->box_rho, &sv->box_u}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
      }
      sv->box_rows_max = rows > sv->box_rows_max ? rows : sv->box_rows_max;
      sv->box_cols_max = cols > sv->box_cols_max ? cols : sv->box_cols_max;
      rows = sv->box_rows_max;
      cols = sv->box_cols_max;
      sv->box_plane = (long long)rows * cols + 136;
      const size_t n = (size_t)rows * cols;
      LBM_CHECK_HIP(hipMalloc(&sv->box[0], (size_t)sv->box_plane * 9 * sizeof(double)));
      LBM_CHECK_HIP(hipMalloc(&sv->box[1], (size_t)sv->box_plane * 9 * sizeof(double)));
      LBM_CHECK_HIP(hipMalloc(&sv->box_rho, n * sizeof(double)));
      LBM_CHECK_HIP(hipMalloc(&sv->box_u, 2 * n * sizeof(double)));
    }
    if (!sv->far_st) {
      int rc = make_background_stream(&sv->far_st);
      if (rc) return rc;
      LBM_CHECK_HIP(hipEventCreateWithFlags(&sv->ev_far_fork, hipEventDisableTiming));
      LBM_CHECK_HIP(hipEventCreateWithFlags(&sv->ev_far_join, hipEventDisableTiming));
    }
  }
  sv->guo_a = guo_a;
  sv->guo_b = guo_b;
  return LBM_OK;
}

long long lbm_solver_block_launches(const lbm_solver* sv) { return sv ? sv->blocks : -1; }

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

// ---- snapshot / checkpoint I/O ----------------------------------------------
// The reference keeps [R,C,n_snap] stacks on the host and torch::save()s them at the end
// (e.g. horizontal_poiseuille_test.cpp:85-88,157-160).  Here a snapshot is an asynchronous
// device->pinned-host copy of the recorded moments on a private stream (the solver keeps
// stepping meanwhile) written as NumPy .npy; a checkpoint is the raw resident lattice plus the
// solver's bookkeeping, restart is bitwise.
struct lbm_snapshot {
  lbm_solver* sv;
  hipStream_t copy;
  hipEvent_t ready, done;
  double *d_rho, *d_u;   // device staging (AoS u)
  double *h_rho, *h_u;   // pinned host
  long step;
};

namespace lbm {
int write_npy(const char* path, const double* data, const std::vector<long>& shape) {
  std::string dict = "{'descr': '<f8', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < shape.size(); ++i) dict += std::to_string(shape[i]) + (shape.size() == 1 || i + 1 < shape.size() ? ", " : "");
  dict += "), }";
  size_t total = 10 + dict.size() + 1;
  const size_t pad = (64 - total % 64) % 64;
  dict += std::string(pad, ' ') + "\n";
  FILE* f = std::fopen(path, "wb");
  LBM_REQUIRE(f, "cannot open %s for writing", path);
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  const uint16_t hl = (uint16_t)dict.size();
  size_t n = 1;
  for (long d : shape) n *= (size_t)d;
  bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(&hl, 2, 1, f) == 1 &&
            std::fwrite(dict.data(), 1, dict.size(), f) == dict.size() &&
            std::fwrite(data, sizeof(double), n, f) == n;
  ok = (std::fclose(f) == 0) && ok;
  LBM_REQUIRE(ok, "short write to %s", path);
  return LBM_OK;
}
}  // namespace lbm

struct CheckpointHeader {
  char magic[8];  // "LBMCKPT1"
  int32_t model, R, C, post;
  int64_t steps;
  lbm_bc bc;
  lbm_bgk_params bgk;
  lbm_kbc_params kbc;
};

extern "C" {

int lbm_snapshot_create(lbm_snapshot** out, lbm_solver* sv) {
  LBM_REQUIRE(out && sv, "lbm_snapshot_create: NULL argument");
  lbm_snapshot* sn = new (std::nothrow) lbm_snapshot();
  LBM_REQUIRE(sn, "lbm_snapshot_create: out of host memory");
  std::memset(sn, 0, sizeof *sn);
  sn->sv = sv;
  const size_t n = (size_t)sv->g.R * sv->g.C;
  hipError_t e = hipStreamCreateWithFlags(&sn->copy, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sn->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sn->done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(&sn->d_rho, n * 8);
  if (e == hipSuccess) e = hipMalloc(&sn->d_u, n * 16);
  if (e == hipSuccess) e = hipHostMalloc(&sn->h_rho, n * 8);
  if (e == hipSuccess) e = hipHostMalloc(&sn->h_u, n * 16);
  if (e != hipSuccess) {
    set_error("lbm_snapshot_create: %s", hipGetErrorString(e));
    lbm_snapshot_destroy(sn);
    return LBM_ERR_HIP;
  }
  *out = sn;
  return LBM_OK;
}

int lbm_snapshot_destroy(lbm_snapshot* sn) {
  if (!sn) return LBM_OK;
  if (sn->copy) (void)hipStreamSynchronize(sn->copy);
  if (sn->d_rho) (void)hipFree(sn->d_rho);
  if (sn->d_u) (void)hipFree(sn->d_u);
  if (sn->h_rho) (void)hipHostFree(sn->h_rho);
  if (sn->h_u) (void)hipHostFree(sn->h_u);
  if (sn->ready) (void)hipEventDestroy(sn->ready);
  if (sn->done) (void)hipEventDestroy(sn->done);
  if (sn->copy) (void)hipStreamDestroy(sn->copy);
  delete sn;
  return LBM_OK;
}

// Capture the moments recorded by the last lbm_solver_step(.., record_moments = 1).  The staging
// copy is ordered on the solver's stream (so later steps may overwrite rho/u at once); the
// device->host transfer runs on the snapshot's own stream.  Returns immediately.
int lbm_snapshot_record(lbm_snapshot* sn) {
  LBM_REQUIRE(sn, "lbm_snapshot_record: NULL snapshot");
  lbm_solver* sv = sn->sv;
  if (!sv->have_moments) {
    set_error("lbm_snapshot_record: no step(.., record_moments=1) since the last set_f");
    return LBM_ERR_STATE;
  }
  const int R = sv->g.R, C = sv->g.C;
  const size_t n = (size_t)R * C;
  LBM_CHECK_HIP(hipStreamWaitEvent(sv->st, sn->done, 0));  // previous D2H of this snapshot finished
  LBM_CHECK_HIP(hipMemcpyAsync(sn->d_rho, sv->rho, n * 8, hipMemcpyDeviceToDevice, sv->st));
  int rc = lbm_soa_to_aos(sn->d_u, sv->u, R, C, 2, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipEventRecord(sn->ready, sv->st));
  LBM_CHECK_HIP(hipStreamWaitEvent(sn->copy, sn->ready, 0));
  LBM_CHECK_HIP(hipMemcpyAsync(sn->h_rho, sn->d_rho, n * 8, hipMemcpyDeviceToHost, sn->copy));
  LBM_CHECK_HIP(hipMemcpyAsync(sn->h_u, sn->d_u, n * 16, hipMemcpyDeviceToHost, sn->copy));
  LBM_CHECK_HIP(hipEventRecord(sn->done, sn->copy));
  sn->step = sv->steps;
  return LBM_OK;
}

// Wait for the transfer and write rho [R,C] and u [R,C,2] as NumPy .npy files (either path may
// be NULL).  The host pointers stay valid until the next record: lbm_snapshot_host().
int lbm_snapshot_write_npy(lbm_snapshot* sn, const char* rho_path, const char* u_path) {
  LBM_REQUIRE(sn, "lbm_snapshot_write_npy: NULL snapshot");
  LBM_CHECK_HIP(hipStreamSynchronize(sn->copy));
  const long R = sn->sv->g.R, C = sn->sv->g.C;
  if (rho_path) {
    int rc = write_npy(rho_path, sn->h_rho, {R, C});
    if (rc) return rc;
  }
  if (u_path) return write_npy(u_path, sn->h_u, {R, C, 2});
  return LBM_OK;
}

int lbm_snapshot_host(lbm_snapshot* sn, const double** rho, const double** u, long long* step) {
  LBM_REQUIRE(sn, "lbm_snapshot_host: NULL snapshot");
  LBM_CHECK_HIP(hipStreamSynchronize(sn->copy));
  if (rho) *rho = sn->h_rho;
  if (u) *u = sn->h_u;
  if (step) *step = sn->step;
  return LBM_OK;
}

int lbm_solver_checkpoint_save(lbm_solver* sv, const char* path) {
  LBM_REQUIRE(sv && path, "lbm_solver_checkpoint_save: NULL argument");
  const size_t n = (size_t)sv->g.R * sv->g.C;
  std::vector<double> host(n * 9);
  const size_t plane_bytes = n * sizeof(double);
  if (sv->g.row_pitch) {  // padded rows: through the dense staging buffer (the file always holds dense planes)
    int rc = solver_copy_planes(sv, sv->lat[sv->cur], false, sv->stage, 0, sv->g.R, 0, 0, sv->g.R, sv->st);
    if (rc) return rc;
    LBM_CHECK_HIP(hipMemcpyAsync(host.data(), sv->stage, 9 * plane_bytes, hipMemcpyDeviceToHost, sv->st));
  } else
  LBM_CHECK_HIP(hipMemcpy2DAsync(host.data(), plane_bytes, sv->lat[sv->cur],
                                 (size_t)sv->g.plane_stride * sizeof(double), plane_bytes, 9,
                                 hipMemcpyDeviceToHost, sv->st));
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  CheckpointHeader h;
  std::memset(&h, 0, sizeof h);
  std::memcpy(h.magic, "LBMCKPT1", 8);
  h.model = sv->model; h.R = sv->g.R; h.C = sv->g.C; h.post = sv->post ? 1 : 0; h.steps = sv->steps;
  h.bc = sv->bc; h.bgk = sv->bgk; h.kbc = sv->kbc;
  FILE* f = std::fopen(path, "wb");
  LBM_REQUIRE(f, "lbm_solver_checkpoint_save: cannot open %s", path);
  bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(host.data(), 8, host.size(), f) == host.size();
  ok = (std::fclose(f) == 0) && ok;
  LBM_REQUIRE(ok, "lbm_solver_checkpoint_save: short write to %s", path);
  return LBM_OK;
}

// Restore lattice, state form and step counter into a solver created with the same model and size
// (boundary set and parameters are taken from the file).
int lbm_solver_checkpoint_load(lbm_solver* sv, const char* path) {
  LBM_REQUIRE(sv && path, "lbm_solver_checkpoint_load: NULL argument");
  FILE* f = std::fopen(path, "rb");
  LBM_REQUIRE(f, "lbm_solver_checkpoint_load: cannot open %s", path);
  CheckpointHeader h;
  const size_t n = (size_t)sv->g.R * sv->g.C;
  std::vector<double> host(n * 9);
  bool ok = std::fread(&h, sizeof h, 1, f) == 1 && std::memcmp(h.magic, "LBMCKPT1", 8) == 0;
  if (ok && (h.model != sv->model || h.R != sv->g.R || h.C != sv->g.C)) {
    std::fclose(f);
    set_error("lbm_solver_checkpoint_load: %s holds model %d %dx%d, solver is model %d %dx%d", path,
              h.model, h.R, h.C, sv->model, sv->g.R, sv->g.C);
    return LBM_ERR_INVALID;
  }
  ok = ok && std::fread(host.data(), 8, host.size(), f) == host.size();
  std::fclose(f);
  LBM_REQUIRE(ok, "lbm_solver_checkpoint_load: %s is not a complete checkpoint", path);
  const size_t plane_bytes = n * sizeof(double);
  if (sv->g.row_pitch) {
    LBM_CHECK_HIP(hipMemcpyAsync(sv->stage, host.data(), 9 * plane_bytes, hipMemcpyHostToDevice, sv->st));
    int rc = solver_copy_planes(sv, sv->lat[sv->cur], true, sv->stage, 0, sv->g.R, 0, 0, sv->g.R, sv->st);
    if (rc) return rc;
  } else
  LBM_CHECK_HIP(hipMemcpy2DAsync(sv->lat[sv->cur], (size_t)sv->g.plane_stride * sizeof(double),
                                 host.data(), plane_bytes, plane_bytes, 9, hipMemcpyHostToDevice, sv->st));
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  sv->bc = h.bc; sv->bgk = h.bgk; sv->kbc = h.kbc;
  sv->post = h.post != 0;
  sv->steps = h.steps;
  sv->have_moments = false;
  return LBM_OK;
}

}  // extern "C"
