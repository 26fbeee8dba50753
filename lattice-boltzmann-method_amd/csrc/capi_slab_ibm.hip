// C ABI, part 8: config 5 (test/cylinder_test.cpp:88-164) over row slabs at multi-step speed, with the
// immersed boundary anywhere -- also across a slab seam.
//
// The forcing (src/ibm.cpp:158-190) changes every step, but only inside the ROI rows [q0, q1).  A BAND
// of global rows [q0 - 2D, q1 + 2D) therefore advances D forced single steps on a shrinking trapezoid
// (step k computes rows [q0 - 2D + k, q1 + 2D - k)), leaving rows V = [q0 - D, q1 + D) valid after the
// block; every other row is at least D away from the ROI, sees plain BGK for D steps and takes the
// D-step window from the time-t lattice (the single-block form of this is solver_ibm_block,
// capi_solver.hip).  Here the band lives in a compact lattice of its own, REPLICATED on every slab that
// owns rows of V: each co-owner runs the whole band chain (same kernels on the same inputs: same bits),
// keeps its part of V, and needs from outside only the D outermost band rows on each side per block --
// far rows of this slab or, when V straddles a seam, of the co-owner, who sends them in place of the
// ordinary halo of that seam (same message size: 9 D rows of C doubles; the seam itself lies inside the
// band, nobody reads its ghost rows).  No forcing data ever crosses a seam.
//
// The transport is the caller's: lbm_slab_ibm_block_compute fills two send buffers, *_finish consumes
// two receive buffers (lbm_ring_bgk_block_ibm in capi_ring.hip moves them with RCCL; the emulated
// two-slab tests with plain device copies).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "d2q9.hpp"
#include "internal.hpp"
#include "slab_ibm.hpp"

using namespace lbm;

namespace {
inline long long plane_of(const lbm_geom& g) {
  return g.plane_stride > 0 ? g.plane_stride : (long long)(g.R + 2 * g.ghost) * g.C;
}
// a message buffer of n rows viewed as a dense lattice [9][n][C]
inline lbm_geom msg_geom(int n, int C) { return lbm_geom{n, C, 0, (long long)n * C}; }
}  // namespace

extern "C" {

// n_rows complete rows (all 9 populations) from one lattice into another; rows in owned-row indices,
// ghost rows (negative / >= R) allowed where the geometry has them
int lbm_rows_copy(double* dst, const lbm_geom* dg, int dst_row, const double* src, const lbm_geom* sg,
                  int src_row, int n_rows, lbm_stream_t s) {
  LBM_REQUIRE(dst && dg && src && sg, "lbm_rows_copy: NULL argument");
  LBM_REQUIRE(dg->C == sg->C && dg->C > 0, "lbm_rows_copy: column counts differ (%d vs %d)", dg->C, sg->C);
  LBM_REQUIRE(n_rows >= 0 && dst_row >= -dg->ghost && dst_row + n_rows <= dg->R + dg->ghost &&
                  src_row >= -sg->ghost && src_row + n_rows <= sg->R + sg->ghost,
              "lbm_rows_copy: rows [%d,+%d) -> [%d,+%d) outside the lattices", src_row, n_rows, dst_row, n_rows);
  if (n_rows == 0) return LBM_OK;
  const size_t width = (size_t)n_rows * dg->C * sizeof(double);
  LBM_CHECK_HIP(hipMemcpy2DAsync(dst + (size_t)(dst_row + dg->ghost) * dg->C, (size_t)plane_of(*dg) * sizeof(double),
                                 src + (size_t)(src_row + sg->ghost) * sg->C, (size_t)plane_of(*sg) * sizeof(double),
                                 width, 9, hipMemcpyDeviceToDevice, as_stream(s)));
  return LBM_OK;
}

int lbm_slab_ibm_destroy(lbm_slab_ibm* sl) {
  if (!sl) return LBM_OK;
  if (sl->aux) {
    (void)hipStreamSynchronize(sl->aux);
    (void)hipStreamDestroy(sl->aux);
  }
  if (sl->bgst) {
    (void)hipStreamSynchronize(sl->bgst);
    (void)hipStreamDestroy(sl->bgst);
  }
  if (sl->ev_fork) (void)hipEventDestroy(sl->ev_fork);
  if (sl->ev_join) (void)hipEventDestroy(sl->ev_join);
  if (sl->ib) (void)lbm_ibm_destroy(sl->ib);
  for (double* p : {sl->blat[0], sl->blat[1], sl->brho, sl->bu, sl->stash, sl->box[0], sl->box[1], sl->xrho, sl->xu})
    if (p) (void)hipFree(p);
  delete sl;
  return LBM_OK;
}

int lbm_slab_ibm_create(lbm_slab_ibm** out, const lbm_geom* slab, int slab_row0, int rows_global,
                        const lbm_bc* bc_global, const lbm_bgk_params* prm, int depth, const double* x,
                        const double* y, int n_markers, int m_max, double guo_a, double guo_b) {
  LBM_REQUIRE(out && slab && bc_global && prm && x && y && n_markers > 0, "lbm_slab_ibm_create: bad argument");
  LBM_REQUIRE(slab->row_pitch == 0 || slab->row_pitch == slab->C, "lbm_slab_ibm_create: dense rows only (row_pitch = %d)", slab->row_pitch);
  const int R = slab->R, C = slab->C, D = depth;
  LBM_REQUIRE(D >= 2 && D <= 5, "lbm_slab_ibm_create: depth=%d (supported: 2..5)", D);
  LBM_REQUIRE(slab->ghost >= D, "lbm_slab_ibm_create: slab has %d ghost rows, %d-step blocks need %d", slab->ghost, D, D);
  LBM_REQUIRE(R >= 4 * D + 8 && C >= 64, "lbm_slab_ibm_create: slab %dx%d too small for %d-step blocks", R, C, D);
  LBM_REQUIRE(slab_row0 >= 0 && slab_row0 + R <= rows_global, "lbm_slab_ibm_create: rows [%d,%d) outside the domain of %d rows", slab_row0, slab_row0 + R, rows_global);
  LBM_REQUIRE(!bc_global->pressure_rows && !prm->force_mode, "lbm_slab_ibm_create: no pressure rows / body force on this path");
  // ROI rows, ibm.cpp:124-153
  long q0 = 1L << 30, q1 = 0;
  for (int i = 0; i < n_markers; ++i) {
    const long fx = (long)std::floor(x[i]);
    q0 = std::min(q0, fx - 2);
    q1 = std::max(q1, fx + 3);
  }
  lbm_slab_ibm* sl = new (std::nothrow) lbm_slab_ibm();
  LBM_REQUIRE(sl, "lbm_slab_ibm_create: out of host memory");
  std::memset(sl, 0, sizeof *sl);
  sl->g = *slab;
  sl->row0 = slab_row0;
  sl->rows_global = rows_global;
  sl->bc_global = *bc_global;
  sl->bc = *bc_global;
  sl->has_prev = slab_row0 > 0;
  sl->has_next = slab_row0 + R < rows_global;
  if (sl->has_prev) sl->bc.row_lo = LBM_EDGE_HALO;
  if (sl->has_next) sl->bc.row_hi = LBM_EDGE_HALO;
  sl->prm = *prm;
  sl->D = D;
  sl->ga = guo_a;
  sl->gb = guo_b;
  sl->b0 = (int)q0 - 2 * D;
  sl->b1 = (int)q1 + 2 * D;
  const int v0 = sl->b0 + D, v1 = sl->b1 - D;
  sl->owner = v0 < slab_row0 + R && v1 > slab_row0;
  sl->straddle_prev = sl->owner && v0 < slab_row0;
  sl->straddle_next = sl->owner && v1 > slab_row0 + R;
  auto fail = [&](const char* why) {
    set_error("lbm_slab_ibm_create: band rows [%d,%d) vs slab rows [%d,%d): %s", sl->b0, sl->b1, slab_row0, slab_row0 + R, why);
    lbm_slab_ibm_destroy(sl);
    return LBM_ERR_INVALID;
  };
  if (sl->b0 < 2 || sl->b1 > rows_global - 2) return fail("the band must keep 2 rows from the domain's first / last row");
  if (sl->straddle_prev && sl->straddle_next) return fail("the band covers the whole slab (slabs must be taller than the band)");
  // a straddled band must end inside the two slabs that share it: its outer rows are far rows of one of them
  if (sl->straddle_prev && (sl->b1 > slab_row0 + R || sl->b0 < slab_row0 - R)) return fail("the band reaches a third slab");
  if (sl->straddle_next && (sl->b0 < slab_row0 || sl->b1 > slab_row0 + 2 * R)) return fail("the band reaches a third slab");
  if (!sl->owner) {
    *out = sl;
    return LBM_OK;
  }
  const int Rb = sl->b1 - sl->b0;
  sl->bg = lbm_geom{Rb, C, 0, (long long)Rb * C + 1088};
  sl->bbc = lbm_bc{LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, bc_global->col_lo, bc_global->col_hi, 0, 1.0, 1.0, 0.0, 0.0};
  int rc = lbm_ibm_create_slab(&sl->ib, x, y, n_markers, m_max, Rb, C, sl->b0);
  if (rc) {
    lbm_slab_ibm_destroy(sl);
    return rc;
  }
  const size_t lat_bytes = (size_t)sl->bg.plane_stride * 9 * sizeof(double), n = (size_t)Rb * C;
  hipError_t e = hipSuccess;
  for (double** p : {&sl->blat[0], &sl->blat[1]}) {
    if (e == hipSuccess) e = hipMalloc(p, lat_bytes);
    if (e == hipSuccess) e = hipMemset(*p, 0, lat_bytes);
  }
  if (e == hipSuccess) e = hipMalloc(&sl->brho, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sl->bu, 2 * n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&sl->stash, (size_t)9 * D * C * sizeof(double));
  if (e == hipSuccess) e = hipMemset(sl->stash, 0, (size_t)9 * D * C * sizeof(double));
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&sl->aux, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sl->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sl->ev_join, hipEventDisableTiming);
  // the forced box (lbm_slab_ibm_block_compute): only if it keeps clear of the wall columns
  int q0r, q1r, c0, c1;
  if (e == hipSuccess && lbm_ibm_roi(sl->ib, &q0r, &q1r, &c0, &c1) == LBM_OK && c0 - 2 * D >= 8 && (c1 + 2 * D + 7) / 8 * 8 <= C - 1) {
    sl->bc0 = (c0 - 2 * D) / 8 * 8;
    sl->bc1 = (c1 + 2 * D + 7) / 8 * 8;
    const int Cb = sl->bc1 - sl->bc0;
    sl->xg = lbm_geom{Rb, Cb, 0, (long long)Rb * Cb + 136};
    const size_t xb = (size_t)sl->xg.plane_stride * 9 * sizeof(double), xn = (size_t)Rb * Cb;
    for (double** p : {&sl->box[0], &sl->box[1]})
      if (e == hipSuccess) e = hipMalloc(p, xb);
    if (e == hipSuccess) e = hipMalloc(&sl->xrho, xn * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&sl->xu, 2 * xn * sizeof(double));
    // (ordinary priority: with the lowest one a co-owner's block took 1.15 instead of 0.70 ms in a running chain --
    // profiles/r02_cylinder_emulated_8_slabs_events.txt; one block on lbm_solver_step does not care, 75 / 96 / 119 k either way)
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&sl->bgst, hipStreamNonBlocking);
    sl->boxed = e == hipSuccess;
  }
  if (e != hipSuccess) {
    set_error("lbm_slab_ibm_create: %s", hipGetErrorString(e));
    lbm_slab_ibm_destroy(sl);
    return LBM_ERR_HIP;
  }
  *out = sl;
  return LBM_OK;
}

// Slab heights for a chain of n slabs: a chain runs at its slowest slab's pace, and the slab that
// carries the forced band pays its chain of D forced single steps (latency-bound, ~76 us per step beside the window launch)
// whatever its height -- so it gets the band and little else, and the band is kept INSIDE one slab (a seam through the band
// makes two co-owners run the whole chain).  Cost model per block, linear in the rows (MI355X, profiles/r02_cylinder_*,
// r03_cylinder_planned*): slabs without band rows  far_us_per_row x rows;  the owner  owner_us + owner_us_per_row x rows
// (its far rows run beside the chain).  costs = {far_us_per_row, owner_us, owner_us_per_row} or NULL for the built-in table
// (scaled with the columns and the depth).  Exhaustive search over the owner's slab index, first row and height (multiples of
// 4 rows); the slabs before / after the owner share their rows equally.  Falls back to equal heights when the band fits no
// single slab.  rows_out[n_slabs]; *predicted_us (may be NULL): block time of the slowest slab under the model.
int lbm_slab_ibm_plan_rows(int* rows_out, int n_slabs, int rows_global, int cols, int depth, const double* x, int n_markers,
                           const double* costs, double* predicted_us) {
  LBM_REQUIRE(rows_out && x && n_slabs >= 1 && n_markers > 0 && cols > 0 && depth >= 2 && depth <= 5,
              "lbm_slab_ibm_plan_rows: bad argument");
  const int N = n_slabs, Rg = rows_global, D = depth, hmin_any = 4 * D + 8;
  LBM_REQUIRE(Rg >= N * hmin_any, "lbm_slab_ibm_plan_rows: %d rows cannot be cut into %d slabs of at least %d", Rg, N, hmin_any);
  // built-in table, re-fitted at the end of round 3 on emulated 8-slab chains (profiles/r03_cylinder_emulated_8_slabs_forms.txt):
  // far slabs 0.180 us per row; owner 394.7 us at 316 rows, 472.8 us at 884 rows = 351 us + 0.1375 us per row
  const double far = costs ? costs[0] : 0.180 * (cols / 4096.0) * (D / 5.0);
  const double oc0 = costs ? costs[1] : 70.2 * D, oc1 = costs ? costs[2] : 0.764 * far;
  auto equal = [&](int total, int k, int* out) {  // k heights that differ by at most one row
    for (int i = 0; i < k; ++i) out[i] = total / k + (i < total % k ? 1 : 0);
  };
  long q0 = 1L << 30, q1 = 0;  // ROI rows, ibm.cpp:124-153
  for (int i = 0; i < n_markers; ++i) {
    const long fx = (long)std::floor(x[i]);
    q0 = std::min(q0, fx - 2);
    q1 = std::max(q1, fx + 3);
  }
  const int v0 = (int)q0 - D, v1 = (int)q1 + D;  // the band's valid rows [v0, v1): what one slab must hold
  double best = 1e300;
  int bk = -1, br0 = 0, bh = 0;
  if (N == 1) {
    rows_out[0] = Rg;
    if (predicted_us) *predicted_us = oc0 + oc1 * Rg;
    return LBM_OK;
  }
  const int h_lo = std::max((v1 - v0 + 3) / 4 * 4, (hmin_any + 3) / 4 * 4);
  for (int k = 0; k < N; ++k) {
    const int nb = k, na = N - 1 - k;
    for (int h = h_lo; h <= Rg - (N - 1) * hmin_any; h += 4) {
      const double t_own = oc0 + oc1 * h;
      if (t_own >= best) break;  // taller owners only get slower
      for (int r0 = std::max(0, v1 - h); r0 <= v0; ++r0) {
        if (r0 % 4 && r0 != v1 - h) continue;
        const int before = r0, after = Rg - r0 - h;
        if (after < 0) break;
        if ((nb == 0) != (before == 0) || (na == 0) != (after == 0)) continue;
        if ((nb && before < nb * hmin_any) || (na && after < na * hmin_any)) continue;
        const double tb = nb ? far * ((before + nb - 1) / nb) : 0.0, ta = na ? far * ((after + na - 1) / na) : 0.0;
        const double t = std::max(t_own, std::max(tb, ta));
        if (t < best) best = t, bk = k, br0 = r0, bh = h;
      }
    }
  }
  if (bk < 0) {  // the band fits no single slab under these constraints: equal heights (two slabs will share the band)
    equal(Rg, N, rows_out);
    if (predicted_us) *predicted_us = oc0 + far * ((Rg + N - 1) / N);
    return LBM_OK;
  }
  if (bk > 0) equal(br0, bk, rows_out);
  rows_out[bk] = bh;
  if (bk < N - 1) equal(Rg - br0 - bh, N - 1 - bk, rows_out + bk + 1);
  if (predicted_us) *predicted_us = best;
  return LBM_OK;
}

int lbm_slab_ibm_info(const lbm_slab_ibm* sl, int* owner, int* straddle_prev, int* straddle_next, int* b0, int* b1) {
  LBM_REQUIRE(sl, "lbm_slab_ibm_info: NULL argument");
  if (owner) *owner = sl->owner;
  if (straddle_prev) *straddle_prev = sl->straddle_prev;
  if (straddle_next) *straddle_next = sl->straddle_next;
  if (b0) *b0 = sl->b0;
  if (b1) *b1 = sl->b1;
  return LBM_OK;
}

// doubles per message per side per block: 9 D rows (ordinary complete halo and outer band rows alike)
long long lbm_slab_ibm_msg_doubles(const lbm_slab_ibm* sl) { return sl ? (long long)9 * sl->D * sl->g.C : -1; }

// Priming (once, on the initial post-collision state): across an ordinary seam the complete D-row halo;
// across a straddled seam each co-owner sends ALL its owned band rows, so that both hold the whole band.
int lbm_slab_ibm_prime_counts(const lbm_slab_ibm* sl, int side, long long* send, long long* recv) {
  LBM_REQUIRE(sl && send && recv && (side == 0 || side == 1), "lbm_slab_ibm_prime_counts: bad argument");
  const long long halo = (long long)9 * sl->D * sl->g.C;
  const bool has = side ? sl->has_next : sl->has_prev, strad = side ? sl->straddle_next : sl->straddle_prev;
  if (!has) {
    *send = *recv = 0;
    return LBM_OK;
  }
  if (!strad) {
    *send = *recv = halo;
    return LBM_OK;
  }
  const int seam = side ? sl->row0 + sl->g.R : sl->row0;
  const int mine = side ? seam - sl->b0 : sl->b1 - seam, theirs = (sl->b1 - sl->b0) - mine;
  *send = (long long)9 * mine * sl->g.C;
  *recv = (long long)9 * theirs * sl->g.C;
  return LBM_OK;
}

int lbm_slab_ibm_prime_pack(lbm_slab_ibm* sl, const double* lattice, double* send_prev, double* send_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && lattice, "lbm_slab_ibm_prime_pack: NULL argument");
  const int R = sl->g.R, C = sl->g.C, full = LBM_HALO_FULL(sl->D);
  int rc = LBM_OK;
  if (sl->has_prev && !rc) {
    LBM_REQUIRE(send_prev, "lbm_slab_ibm_prime_pack: NULL send_prev");
    if (sl->straddle_prev) {  // my band rows: global [row0, b1)
      const int n = sl->b1 - sl->row0;
      const lbm_geom mg = msg_geom(n, C);
      rc = lbm_rows_copy(send_prev, &mg, 0, lattice, &sl->g, 0, n, s);
    } else {
      rc = lbm_halo_pack(send_prev, lattice, &sl->g, full, 0, s);
    }
  }
  if (sl->has_next && !rc) {
    LBM_REQUIRE(send_next, "lbm_slab_ibm_prime_pack: NULL send_next");
    if (sl->straddle_next) {  // my band rows: global [b0, row0 + R)
      const int n = sl->row0 + R - sl->b0;
      const lbm_geom mg = msg_geom(n, C);
      rc = lbm_rows_copy(send_next, &mg, 0, lattice, &sl->g, sl->b0 - sl->row0, n, s);
    } else {
      rc = lbm_halo_pack(send_next, lattice, &sl->g, full, 1, s);
    }
  }
  return rc;
}

int lbm_slab_ibm_prime_finish(lbm_slab_ibm* sl, double* lattice, const double* recv_prev, const double* recv_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && lattice, "lbm_slab_ibm_prime_finish: NULL argument");
  const int R = sl->g.R, C = sl->g.C, full = LBM_HALO_FULL(sl->D);
  int rc = LBM_OK;
  if (sl->has_prev && !sl->straddle_prev) {
    LBM_REQUIRE(recv_prev, "lbm_slab_ibm_prime_finish: NULL recv_prev");
    rc = lbm_halo_unpack(lattice, recv_prev, &sl->g, full, 0, s);
  }
  if (!rc && sl->has_next && !sl->straddle_next) {
    LBM_REQUIRE(recv_next, "lbm_slab_ibm_prime_finish: NULL recv_next");
    rc = lbm_halo_unpack(lattice, recv_next, &sl->g, full, 1, s);
  }
  if (rc || !sl->owner) return rc;
  // the whole band at time t into the band lattice: my rows (owned, or ghost rows just filled) and the co-owner's
  double* bl = sl->blat[sl->bcur];
  int lo = sl->b0, hi = sl->b1;  // global rows I take from my own lattice
  if (sl->straddle_prev) {
    LBM_REQUIRE(recv_prev, "lbm_slab_ibm_prime_finish: NULL recv_prev");
    const int n = sl->row0 - sl->b0;
    const lbm_geom mg = msg_geom(n, C);
    rc = lbm_rows_copy(bl, &sl->bg, 0, recv_prev, &mg, 0, n, s);
    // ... whose first D rows are also the band's upper outer rows of the first block
    const lbm_geom sg = msg_geom(sl->D, C);
    if (!rc) rc = lbm_rows_copy(sl->stash, &sg, 0, recv_prev, &mg, 0, sl->D, s);
    lo = sl->row0;
  }
  if (!rc && sl->straddle_next) {
    LBM_REQUIRE(recv_next, "lbm_slab_ibm_prime_finish: NULL recv_next");
    const int n = sl->b1 - (sl->row0 + R);
    const lbm_geom mg = msg_geom(n, C);
    rc = lbm_rows_copy(bl, &sl->bg, sl->row0 + R - sl->b0, recv_next, &mg, 0, n, s);
    const lbm_geom sg = msg_geom(sl->D, C);  // its last D rows: the lower outer rows
    if (!rc) rc = lbm_rows_copy(sl->stash, &sg, 0, recv_next, &mg, n - sl->D, sl->D, s);
    hi = sl->row0 + R;
  }
  if (!rc) rc = lbm_rows_copy(bl, &sl->bg, lo - sl->b0, lattice, &sl->g, lo - sl->row0, hi - lo, s);
  return rc;
}

// The driver's FIRST iteration over slabs (cylinder_test.cpp:103-127 on the initial state: moments,
// collision, forcing, source -- no streaming yet).  The priming messages (lbm_slab_ibm_prime_pack) are
// taken from the PRE-collision lattice; here the halos land in its ghost rows, every row (ghost rows
// included: collision is node-local) is collided into `post`, and (co-)owners run collision + forcing +
// source on the whole band, whose rows are all valid afterwards, and take their share of it.
int lbm_slab_ibm_start_finish(lbm_slab_ibm* sl, double* post, double* pre, const double* recv_prev,
                              const double* recv_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && post && pre && post != pre, "lbm_slab_ibm_start_finish: bad argument");
  int rc = lbm_slab_ibm_prime_finish(sl, pre, recv_prev, recv_next, s);  // ghost rows of pre, band lattice = pre-collision band
  if (rc) return rc;
  const int R = sl->g.R, C = sl->g.C, G = sl->g.ghost, D = sl->D;
  const lbm_geom tall{R + 2 * G, C, 0, plane_of(sl->g)};  // the same planes seen as a ghost-less lattice: all rows collide
  rc = lbm_bgk_collide(post, pre, &tall, nullptr, &sl->prm, nullptr, nullptr, s);
  if (rc || !sl->owner) return rc;
  double* in = sl->blat[sl->bcur];
  double* outl = sl->blat[sl->bcur ^ 1];
  rc = lbm_bgk_collide(outl, in, &sl->bg, &sl->bbc, &sl->prm, sl->brho, sl->bu, s);
  if (!rc) rc = lbm_ibm_step(sl->ib, outl, &sl->bg, sl->bu, sl->brho, sl->prm.omega, sl->ga, sl->gb, s);
  if (rc) return rc;
  sl->bcur ^= 1;
  const int lo = std::max(sl->b0, sl->row0), hi = std::min(sl->b1, sl->row0 + R);
  rc = lbm_rows_copy(post, &sl->g, lo - sl->row0, outl, &sl->bg, lo - sl->b0, hi - lo, s);
  const lbm_geom sg = msg_geom(D, C);
  if (!rc && sl->straddle_prev) rc = lbm_rows_copy(sl->stash, &sg, 0, outl, &sl->bg, 0, D, s);
  if (!rc && sl->straddle_next) rc = lbm_rows_copy(sl->stash, &sg, 0, outl, &sl->bg, sl->bg.R - D, D, s);
  return rc;
}

// One block, phase A: dst = D steps from src on the owned rows (band chain on the helper stream beside
// the far rows), then both outgoing messages packed.  Ghost rows of src: complete and current.
int lbm_slab_ibm_block_compute(lbm_slab_ibm* sl, double* dst, const double* src, double* send_prev,
                               double* send_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && dst && src && dst != src, "lbm_slab_ibm_block_compute: bad argument");
  LBM_REQUIRE((!sl->has_prev || send_prev) && (!sl->has_next || send_next), "lbm_slab_ibm_block_compute: NULL send buffer");
  const int R = sl->g.R, C = sl->g.C, D = sl->D, full = LBM_HALO_FULL(D);
  hipStream_t st = as_stream(s);
  int rc = LBM_OK;
  int o0 = R, o1 = R;  // owned rows [o0, o1) come from the band; everything else is far
  if (sl->owner && sl->boxed && !sl->straddle_prev && !sl->straddle_next && tuning("ibm_box", 1) && tuning("ibm_box_sole", 1)) {
    // SOLE owner (the whole valid band inside this slab's rows; its outer D rows at most in the ghost rows): the band
    // lattice would only mirror rows this slab holds anyway.  So, as lbm_solver_step does on one block: the box straight
    // out of the slab lattice, the D-step window over ALL owned rows beside the chain, the box ROI +- D straight back --
    // no band window launch, no copy of 300-odd full-width rows per block (31 us), one far launch instead of three.
    const int Rb = sl->bg.R, Cb = sl->xg.C, br0 = sl->b0 - sl->row0;  // box row 0 in slab rows (>= -ghost)
    const lbm_bc pb{LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, 0, 1.0, 1.0, 0.0, 0.0};
    LBM_CHECK_HIP(hipEventRecord(sl->ev_fork, st));
    rc = box_copy(sl->box[0], sl->xg, 0, 0, src, sl->g, br0, sl->bc0, Rb, Cb, st);
    if (rc) return rc;
    int cur = 0;
    rc = tuning("ibm_chain_kernel", 0) ? ibm_box_chain(sl->ib, 0, sl->bc0, sl->box, &cur, &sl->xg, &sl->prm, bgk_uses_fast_model(&sl->prm, &pb), D,
                                                       sl->xrho, sl->xu, sl->ga, sl->gb, st)
                                       : 1;
    if (rc < 0) return rc;
    const bool one_launch = rc == 0;
    LBM_CHECK_HIP(hipStreamWaitEvent(sl->bgst, sl->ev_fork, 0));
    rc = one_launch ? ibm_gate(sl->ib, sl->bgst) : LBM_OK;
    if (!rc) rc = lbm_bgk_stream_collide_xn(dst, src, &sl->g, &sl->bc, &sl->prm, D, 0, R, sl->bgst);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(sl->ev_join, sl->bgst));
    for (int k = 1; k <= D && !one_launch && !rc; ++k) {  // cylinder_test.cpp:103-127 on the shrinking trapezoid
      rc = lbm_bgk_stream_collide(sl->box[cur ^ 1], sl->box[cur], &sl->xg, &pb, &sl->prm, k, Rb - k, sl->xrho, sl->xu, s);
      if (!rc) rc = ibm_step_window(sl->ib, 0, sl->bc0, sl->box[cur ^ 1], &sl->xg, sl->xu, sl->xrho, sl->prm.omega, sl->ga, sl->gb, st);
      cur ^= 1;
    }
    if (rc) return rc;
    LBM_CHECK_HIP(hipStreamWaitEvent(st, sl->ev_join, 0));
    rc = box_copy(dst, sl->g, br0 + D, sl->bc0 + D, sl->box[cur], sl->xg, D, D, Rb - 2 * D, Cb - 2 * D, st);
    if (rc) return rc;
    sl->blat_stale = true;
    o0 = 0, o1 = R;  // (nothing left for the far launches below)
  } else if (sl->owner && sl->boxed && tuning("ibm_box", 1)) {
    if (sl->blat_stale) {  // the band lattice fell behind while the slab lattice was worked on directly: all its rows are here
      rc = lbm_rows_copy(sl->blat[sl->bcur], &sl->bg, 0, src, &sl->g, sl->b0 - sl->row0, sl->bg.R, s);
      if (rc) return rc;
      sl->blat_stale = false;
    }
    // The forcing reaches a node only through the ROI, so the D forced single steps are cut to a BOX -- band rows x
    // columns ROI +- 2 D -- held as a small periodic lattice pair of its own (what its wrap spoils is the frame that is
    // dropped anyway), while the band as a whole takes the D-step window like any far row (unforced: right everywhere
    // outside the box ROI +- D, which is then overwritten with the forced result).  The band lattice stays the whole
    // band at time t / t + D on both co-owners, so nothing changes in what travels between them.  The chain of small
    // launches stays on the caller's stream; both window launches go to a stream of their own beside it.
    const int v0 = sl->b0 + D - sl->row0, v1 = sl->b1 - D - sl->row0;
    o0 = v0 < 0 ? 0 : v0;
    o1 = v1 > R ? R : v1;
    double* bl = sl->blat[sl->bcur];
    double* bn = sl->blat[sl->bcur ^ 1];
    const lbm_geom sg = msg_geom(D, C);
    const int Rb = sl->bg.R, Cb = sl->xg.C, hi = sl->b1 - D;  // hi: global
    if (sl->straddle_prev) rc = lbm_rows_copy(bl, &sl->bg, 0, sl->stash, &sg, 0, D, s);
    else rc = lbm_rows_copy(bl, &sl->bg, 0, src, &sl->g, sl->b0 - sl->row0, D, s);
    if (rc) return rc;
    if (sl->straddle_next) rc = lbm_rows_copy(bl, &sl->bg, hi - sl->b0, sl->stash, &sg, 0, D, s);
    else rc = lbm_rows_copy(bl, &sl->bg, hi - sl->b0, src, &sl->g, hi - sl->row0, D, s);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(sl->ev_fork, st));
    const lbm_bc pb{LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, LBM_EDGE_PERIODIC, 0, 1.0, 1.0, 0.0, 0.0};
    rc = box_copy(sl->box[0], sl->xg, 0, 0, bl, sl->bg, 0, sl->bc0, Rb, Cb, st);
    if (rc) return rc;
    int cur = 0;
    // "ibm_chain_kernel" = 1 (opt-in, level with the default): the chain as ONE launch on compute units of its own
    // (lbm::ibm_box_chain), the window launches held back until its workgroups are resident
    rc = tuning("ibm_chain_kernel", 0) ? ibm_box_chain(sl->ib, 0, sl->bc0, sl->box, &cur, &sl->xg, &sl->prm, bgk_uses_fast_model(&sl->prm, &pb), D,
                                                       sl->xrho, sl->xu, sl->ga, sl->gb, st)
                                       : 1;
    if (rc < 0) return rc;
    const bool one_launch = rc == 0;
    LBM_CHECK_HIP(hipStreamWaitEvent(sl->bgst, sl->ev_fork, 0));
    rc = one_launch ? ibm_gate(sl->ib, sl->bgst) : LBM_OK;
    if (!rc) rc = lbm_bgk_stream_collide_xn(bn, bl, &sl->bg, &sl->bbc, &sl->prm, D, D, Rb - D, sl->bgst);
    if (!rc && o0 > 0) rc = lbm_bgk_stream_collide_xn(dst, src, &sl->g, &sl->bc, &sl->prm, D, 0, o0 < R ? o0 : R, sl->bgst);
    if (!rc && o1 < R) rc = lbm_bgk_stream_collide_xn(dst, src, &sl->g, &sl->bc, &sl->prm, D, o1, R, sl->bgst);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(sl->ev_join, sl->bgst));
    for (int k = 1; k <= D && !one_launch && !rc; ++k) {  // cylinder_test.cpp:103-127 on the shrinking trapezoid
      rc = lbm_bgk_stream_collide(sl->box[cur ^ 1], sl->box[cur], &sl->xg, &pb, &sl->prm, k, Rb - k, sl->xrho, sl->xu, s);
      if (!rc) rc = ibm_step_window(sl->ib, 0, sl->bc0, sl->box[cur ^ 1], &sl->xg, sl->xu, sl->xrho, sl->prm.omega, sl->ga, sl->gb, st);
      cur ^= 1;
    }
    if (rc) return rc;
    LBM_CHECK_HIP(hipStreamWaitEvent(st, sl->ev_join, 0));
    rc = box_copy(bn, sl->bg, D, sl->bc0 + D, sl->box[cur], sl->xg, D, D, Rb - 2 * D, Cb - 2 * D, st);
    sl->bcur ^= 1;
    // my part of the valid rows into the slab lattice
    if (!rc) rc = lbm_rows_copy(dst, &sl->g, o0, bn, &sl->bg, o0 + sl->row0 - sl->b0, o1 - o0, s);
    if (rc) return rc;
  } else if (sl->owner) {
    if (sl->blat_stale) {
      rc = lbm_rows_copy(sl->blat[sl->bcur], &sl->bg, 0, src, &sl->g, sl->b0 - sl->row0, sl->bg.R, s);
      if (rc) return rc;
      sl->blat_stale = false;
    }
    const int v0 = sl->b0 + D - sl->row0, v1 = sl->b1 - D - sl->row0;
    o0 = v0 < 0 ? 0 : v0;
    o1 = v1 > R ? R : v1;
    LBM_CHECK_HIP(hipEventRecord(sl->ev_fork, st));
    LBM_CHECK_HIP(hipStreamWaitEvent(sl->aux, sl->ev_fork, 0));
    double* bl = sl->blat[sl->bcur];
    const lbm_geom sg = msg_geom(D, C);
    // the D outermost band rows of each side at time t: far rows of this slab (owned or ghost), or the co-owner's
    if (sl->straddle_prev) rc = lbm_rows_copy(bl, &sl->bg, 0, sl->stash, &sg, 0, D, sl->aux);
    else rc = lbm_rows_copy(bl, &sl->bg, 0, src, &sl->g, sl->b0 - sl->row0, D, sl->aux);
    if (rc) return rc;
    const int hi = sl->b1 - D;  // global
    if (sl->straddle_next) rc = lbm_rows_copy(bl, &sl->bg, hi - sl->b0, sl->stash, &sg, 0, D, sl->aux);
    else rc = lbm_rows_copy(bl, &sl->bg, hi - sl->b0, src, &sl->g, hi - sl->row0, D, sl->aux);
    if (rc) return rc;
    const int Rb = sl->bg.R;
    for (int k = 1; k <= D; ++k) {  // cylinder_test.cpp:103-127 on the shrinking trapezoid
      double* in = sl->blat[sl->bcur];
      double* outl = sl->blat[sl->bcur ^ 1];
      rc = lbm_bgk_stream_collide(outl, in, &sl->bg, &sl->bbc, &sl->prm, k, Rb - k, sl->brho, sl->bu, sl->aux);
      if (!rc) rc = lbm_ibm_step(sl->ib, outl, &sl->bg, sl->bu, sl->brho, sl->prm.omega, sl->ga, sl->gb, sl->aux);
      if (rc) return rc;
      sl->bcur ^= 1;
    }
    // my part of the valid rows into the slab lattice
    rc = lbm_rows_copy(dst, &sl->g, o0, sl->blat[sl->bcur], &sl->bg, o0 + sl->row0 - sl->b0, o1 - o0, sl->aux);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(sl->ev_join, sl->aux));
  }
  const bool boxed = sl->owner && sl->boxed && tuning("ibm_box", 1);  // (both boxed forms have launched their far rows)
  if (!boxed) {
    // far rows: plain D-step window from the time-t lattice
    if (o0 > 0) rc = lbm_bgk_stream_collide_xn(dst, src, &sl->g, &sl->bc, &sl->prm, D, 0, o0 < R ? o0 : R, st);
    if (!rc && o1 < R) rc = lbm_bgk_stream_collide_xn(dst, src, &sl->g, &sl->bc, &sl->prm, D, o1, R, st);
    if (rc) return rc;
    if (sl->owner) LBM_CHECK_HIP(hipStreamWaitEvent(st, sl->ev_join, 0));
  }
  // messages
  if (sl->has_prev) {
    if (sl->straddle_prev) {  // the co-owner above needs the band's lower outer rows: mine, far, now at t + D
      const lbm_geom sg = msg_geom(D, C);
      rc = lbm_rows_copy(send_prev, &sg, 0, dst, &sl->g, sl->b1 - D - sl->row0, D, s);
    } else {
      rc = lbm_halo_pack(send_prev, dst, &sl->g, full, 0, s);
    }
    if (rc) return rc;
  }
  if (sl->has_next) {
    if (sl->straddle_next) {  // the co-owner below needs the band's upper outer rows
      const lbm_geom sg = msg_geom(D, C);
      rc = lbm_rows_copy(send_next, &sg, 0, dst, &sl->g, sl->b0 - sl->row0, D, s);
    } else {
      rc = lbm_halo_pack(send_next, dst, &sl->g, full, 1, s);
    }
  }
  return rc;
}

// phase B: the neighbours' messages are in: ordinary halos into the ghost rows of dst, the co-owner's
// outer band rows into the stash the next block loads them from
int lbm_slab_ibm_block_finish(lbm_slab_ibm* sl, double* dst, const double* recv_prev, const double* recv_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && dst, "lbm_slab_ibm_block_finish: NULL argument");
  LBM_REQUIRE((!sl->has_prev || recv_prev) && (!sl->has_next || recv_next), "lbm_slab_ibm_block_finish: NULL receive buffer");
  const int full = LBM_HALO_FULL(sl->D);
  const size_t msg = (size_t)9 * sl->D * sl->g.C * sizeof(double);
  int rc = LBM_OK;
  if (sl->has_prev) {
    if (sl->straddle_prev) LBM_CHECK_HIP(hipMemcpyAsync(sl->stash, recv_prev, msg, hipMemcpyDeviceToDevice, as_stream(s)));
    else rc = lbm_halo_unpack(dst, recv_prev, &sl->g, full, 0, s);
  }
  if (!rc && sl->has_next) {
    if (sl->straddle_next) LBM_CHECK_HIP(hipMemcpyAsync(sl->stash, recv_next, msg, hipMemcpyDeviceToDevice, as_stream(s)));
    else rc = lbm_halo_unpack(dst, recv_next, &sl->g, full, 1, s);
  }
  return rc;
}

// F_s of the last step of the last block (cylinder_test.cpp:112); owners only
int lbm_slab_ibm_surface_force(lbm_slab_ibm* sl, double* out2, lbm_stream_t s) {
  LBM_REQUIRE(sl && out2, "lbm_slab_ibm_surface_force: NULL argument");
  LBM_REQUIRE(sl->owner && sl->ib, "lbm_slab_ibm_surface_force: this slab does not own the boundary");
  LBM_CHECK_HIP(hipStreamSynchronize(sl->aux));
  if (sl->bgst) LBM_CHECK_HIP(hipStreamSynchronize(sl->bgst));
  return lbm_ibm_surface_force(sl->ib, out2, s);
}

}  // extern "C"
