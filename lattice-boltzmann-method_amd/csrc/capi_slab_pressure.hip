// C ABI, part 9: pressure-periodic rows (test/horizontal_poiseuille_test.cpp:25-45; the two-block form of
// test/decompose_domain.cpp:50-73) over row slabs, in blocks of D steps.
//
// The virtual rows 0 / Rg-1 of the GLOBAL domain are rewritten every step from the collision of rows Rg-2 / 1
// -- rows that live on the OTHER end slab of the (periodic) ring.  As inside one block (capi_solver.hip
// solver_pressure_block) the 2 D rows on either side of that seam advance D ordinary single steps on a small
// periodic lattice of 4 D rows whose own wrap is the seam, and every row at least D away from the virtual rows
// takes the D-step window.  Over slabs the small lattice is REPLICATED on the two end slabs: both run the whole
// chain (same kernels, same inputs: same bits), each keeps its D rows next to the seam, and per block they swap
// the D rows at distance [D, 2D) from the seam -- far rows of their owner -- in place of that seam's halo (same
// size: 9 D rows of C doubles).  Middle slabs see an ordinary slab with wall columns.  Transport-free like
// capi_slab_ibm.hip: *_compute fills two send buffers, *_finish consumes two receive buffers.
#include <cstring>
#include <new>

#include "d2q9.hpp"
#include "internal.hpp"

using namespace lbm;

struct lbm_slab_pressure {
  int model;               // LBM_MODEL_BGK / LBM_MODEL_KBC
  lbm_kbc_params kprm;     // KBC: blocks of 2 steps, far rows through the reference-order 2-step window
  double *sm0, *sm1;       // KBC: the held moments of the small lattice's rows for the driver's first iteration
  lbm_geom g;
  int row0, rows_global, D;
  lbm_bc bc_seam, bc_far;  // the small lattice's edges (= the domain's, pressure rows on); this slab's far rows (HALO rows, no pressure rows)
  lbm_bgk_params prm;
  bool first, last;        // owns the virtual row 0 / Rg - 1
  lbm_geom sg;             // small lattice: rows [0, 2D) of the domain, then rows [Rg - 2D, Rg)
  double* slat[2];
  int scur;
  double* stash;           // [9][D][C]: the partner's rows at distance [D, 2D) from the seam
  hipStream_t aux;
  hipEvent_t ev_fork, ev_join;
};

namespace {
inline lbm_geom msg_geom(int n, int C) { return lbm_geom{n, C, 0, (long long)n * C}; }
inline long long plane_of(const lbm_geom& g) { return g.plane_stride > 0 ? g.plane_stride : (long long)(g.R + 2 * g.ghost) * g.C; }
}  // namespace

extern "C" {

int lbm_slab_pressure_info(const lbm_slab_pressure* sl, int* R, int* C, int* ghost, int* depth) {
  LBM_REQUIRE(sl, "lbm_slab_pressure_info: NULL slab");
  if (R) *R = sl->g.R;
  if (C) *C = sl->g.C;
  if (ghost) *ghost = sl->g.ghost;
  if (depth) *depth = sl->D;
  return LBM_OK;
}

int lbm_slab_pressure_destroy(lbm_slab_pressure* sl) {
  if (!sl) return LBM_OK;
  if (sl->aux) {
    (void)hipStreamSynchronize(sl->aux);
    (void)hipStreamDestroy(sl->aux);
  }
  if (sl->ev_fork) (void)hipEventDestroy(sl->ev_fork);
  if (sl->ev_join) (void)hipEventDestroy(sl->ev_join);
  for (double* p : {sl->slat[0], sl->slat[1], sl->stash, sl->sm0, sl->sm1})
    if (p) (void)hipFree(p);
  delete sl;
  return LBM_OK;
}

static int slab_pressure_create(lbm_slab_pressure** out, const lbm_geom* slab, int slab_row0, int rows_global,
                                const lbm_bc* bc_global, const lbm_bgk_params* prm, const lbm_kbc_params* kprm, int depth);

int lbm_slab_pressure_create(lbm_slab_pressure** out, const lbm_geom* slab, int slab_row0, int rows_global,
                             const lbm_bc* bc_global, const lbm_bgk_params* prm, int depth) {
  LBM_REQUIRE(prm, "lbm_slab_pressure_create: NULL argument");
  return slab_pressure_create(out, slab, slab_row0, rows_global, bc_global, prm, nullptr, depth);
}

// KBC + pressure-periodic rows (test/ulbm_poiseuille.cpp:36-58, :85-139) over slabs: blocks of 2 steps (the depth of the
// reference-order KBC window, as on one block); the driver's first iteration collides on HELD moments (:85-86), which the
// start-up calls below take per slab
int lbm_slab_pressure_create_kbc(lbm_slab_pressure** out, const lbm_geom* slab, int slab_row0, int rows_global,
                                 const lbm_bc* bc_global, const lbm_kbc_params* prm) {
  LBM_REQUIRE(prm && prm->s2 > 0.0 && prm->s2 <= 2.0, "lbm_slab_pressure_create_kbc: bad parameters");
  return slab_pressure_create(out, slab, slab_row0, rows_global, bc_global, nullptr, prm, 2);
}

static int slab_pressure_create(lbm_slab_pressure** out, const lbm_geom* slab, int slab_row0, int rows_global,
                                const lbm_bc* bc_global, const lbm_bgk_params* prm, const lbm_kbc_params* kprm, int depth) {
  LBM_REQUIRE(out && slab && bc_global, "lbm_slab_pressure_create: NULL argument");
  LBM_REQUIRE(slab->row_pitch == 0 || slab->row_pitch == slab->C, "lbm_slab_pressure_create: dense rows only (row_pitch = %d)", slab->row_pitch);
  const int R = slab->R, C = slab->C, D = depth;
  LBM_REQUIRE(D >= 2 && D <= 5, "lbm_slab_pressure_create: depth=%d (supported: 2..5)", D);
  LBM_REQUIRE(slab->ghost >= D && R >= 6 * D + 8 && C >= 64, "lbm_slab_pressure_create: slab %dx%d with %d ghost rows too small for %d-step blocks", R, C, slab->ghost, D);
  LBM_REQUIRE(slab_row0 >= 0 && slab_row0 + R <= rows_global && R < rows_global,
              "lbm_slab_pressure_create: rows [%d,%d) of %d (a single block runs lbm_solver_step)", slab_row0, slab_row0 + R, rows_global);
  auto col_ok = [](int m) { return m == LBM_EDGE_PERIODIC || bc_is_wall(m); };
  LBM_REQUIRE(bc_global->pressure_rows == 1 && bc_global->row_lo == LBM_EDGE_PERIODIC && bc_global->row_hi == LBM_EDGE_PERIODIC &&
                  col_ok(bc_global->col_lo) && col_ok(bc_global->col_hi) && !bc_mixed_axis(make_bc(bc_global)),
              "lbm_slab_pressure_create: needs pressure rows on periodic row edges and periodic / wall columns");
  lbm_slab_pressure* sl = new (std::nothrow) lbm_slab_pressure();
  LBM_REQUIRE(sl, "lbm_slab_pressure_create: out of host memory");
  std::memset(sl, 0, sizeof *sl);
  sl->g = *slab;
  sl->row0 = slab_row0;
  sl->rows_global = rows_global;
  sl->D = D;
  sl->model = kprm ? LBM_MODEL_KBC : LBM_MODEL_BGK;
  if (prm) sl->prm = *prm;
  if (kprm) {
    sl->kprm = *kprm;
    sl->kprm.form = LBM_FORM_REFERENCE_ORDER;  // lattices with pressure rows keep the reference order on every path
  }
  sl->bc_seam = *bc_global;
  sl->bc_far = *bc_global;
  sl->bc_far.pressure_rows = 0;
  sl->bc_far.row_lo = sl->bc_far.row_hi = LBM_EDGE_HALO;  // every seam of the ring, the periodic one included
  sl->first = slab_row0 == 0;
  sl->last = slab_row0 + R == rows_global;
  if (!sl->first && !sl->last) {
    *out = sl;
    return LBM_OK;
  }
  sl->sg = lbm_geom{4 * D, C, 0, (long long)4 * D * C + 1088};
  const size_t lat_bytes = (size_t)sl->sg.plane_stride * 9 * sizeof(double), stash_bytes = (size_t)9 * D * C * sizeof(double);
  hipError_t e = hipSuccess;
  for (double** p : {&sl->slat[0], &sl->slat[1]}) {
    if (e == hipSuccess) e = hipMalloc(p, lat_bytes);
    if (e == hipSuccess) e = hipMemset(*p, 0, lat_bytes);
  }
  if (e == hipSuccess) e = hipMalloc(&sl->stash, stash_bytes);
  if (e == hipSuccess) e = hipMemset(sl->stash, 0, stash_bytes);
  if (kprm) {
    if (e == hipSuccess) e = hipMalloc(&sl->sm0, (size_t)4 * D * C * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&sl->sm1, (size_t)8 * D * C * sizeof(double));
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&sl->aux, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sl->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sl->ev_join, hipEventDisableTiming);
  if (e != hipSuccess) {
    set_error("lbm_slab_pressure_create: %s", hipGetErrorString(e));
    lbm_slab_pressure_destroy(sl);
    return LBM_ERR_HIP;
  }
  *out = sl;
  return LBM_OK;
}

// doubles of the message towards `side` (0 = previous slab, 1 = next; the ring is periodic): per block every
// message is 9 D rows; the start-up message across the pressure seam carries the 2 D rows next to it
long long lbm_slab_pressure_msg_doubles(const lbm_slab_pressure* sl, int side, int start) {
  if (!sl) return -1;
  const bool seam = (side == 0 && sl->first) || (side == 1 && sl->last);
  // (KBC start-up across the pressure seam: the 2 D rows AND their held moments m0, m1: 9 + 3 planes)
  const int planes = (start && seam && sl->model == LBM_MODEL_KBC) ? 12 : 9;
  return (long long)planes * (start && seam ? 2 * sl->D : sl->D) * sl->g.C;
}

// start-up, on the driver's PRE-collision state: complete D-row halos across ordinary seams, the 2 D rows next to
// the pressure seam across that one
int lbm_slab_pressure_start_pack(lbm_slab_pressure* sl, const double* pre, double* send_prev, double* send_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && pre && send_prev && send_next, "lbm_slab_pressure_start_pack: NULL argument");
  LBM_REQUIRE(sl->model == LBM_MODEL_BGK, "lbm_slab_pressure_start_pack: a KBC slab starts with lbm_slab_pressure_start_pack_kbc (held moments)");
  const int R = sl->g.R, C = sl->g.C, D = sl->D, full = LBM_HALO_FULL(D);
  const lbm_geom mg = msg_geom(2 * D, C);
  int rc = sl->first ? lbm_rows_copy(send_prev, &mg, 0, pre, &sl->g, 0, 2 * D, s) : lbm_halo_pack(send_prev, pre, &sl->g, full, 0, s);
  if (!rc) rc = sl->last ? lbm_rows_copy(send_next, &mg, 0, pre, &sl->g, R - 2 * D, 2 * D, s) : lbm_halo_pack(send_next, pre, &sl->g, full, 1, s);
  return rc;
}

// ... then the driver's first iteration (collision of every row; the virtual rows and their neighbours from the
// small lattice, horizontal_poiseuille_test.cpp:130-140 on the initial state): `post` = post-collision state with
// current ghost rows on the ordinary seams, small lattice primed
int lbm_slab_pressure_start_finish(lbm_slab_pressure* sl, double* post, double* pre, const double* recv_prev,
                                   const double* recv_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && post && pre && post != pre && recv_prev && recv_next, "lbm_slab_pressure_start_finish: bad argument");
  LBM_REQUIRE(sl->model == LBM_MODEL_BGK, "lbm_slab_pressure_start_finish: a KBC slab starts with lbm_slab_pressure_start_finish_kbc");
  const int R = sl->g.R, C = sl->g.C, D = sl->D, G = sl->g.ghost, full = LBM_HALO_FULL(D);
  hipStream_t st = as_stream(s);
  int rc = LBM_OK;
  if (!sl->first) rc = lbm_halo_unpack(pre, recv_prev, &sl->g, full, 0, s);
  if (!rc && !sl->last) rc = lbm_halo_unpack(pre, recv_next, &sl->g, full, 1, s);
  if (rc) return rc;
  const lbm_geom tall{R + 2 * G, C, 0, plane_of(sl->g)};  // all rows, ghost rows included: collision is node-local
  rc = bgk_collide_ref(post, pre, &tall, &sl->prm, st);
  if (rc || (!sl->first && !sl->last)) return rc;
  // small lattice, pre-collision: rows [0, 2D) of the domain, then rows [Rg - 2D, Rg)
  const lbm_geom mg = msg_geom(2 * D, C);
  double* sp = sl->slat[sl->scur];
  if (sl->first) {
    rc = lbm_rows_copy(sp, &sl->sg, 0, pre, &sl->g, 0, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sp, &sl->sg, 2 * D, recv_prev, &mg, 0, 2 * D, s);
  } else {
    rc = lbm_rows_copy(sp, &sl->sg, 2 * D, pre, &sl->g, R - 2 * D, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sp, &sl->sg, 0, recv_next, &mg, 0, 2 * D, s);
  }
  double* so = sl->slat[sl->scur ^ 1];
  if (!rc) rc = lbm_bgk_collide(so, sp, &sl->sg, &sl->bc_seam, &sl->prm, nullptr, nullptr, s);  // incl. the pressure rows
  if (rc) return rc;
  sl->scur ^= 1;
  const lbm_geom dg = msg_geom(D, C);
  if (sl->first) {
    rc = lbm_rows_copy(post, &sl->g, 0, so, &sl->sg, 0, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sl->stash, &dg, 0, so, &sl->sg, 2 * D, D, s);  // the partner's rows [Rg - 2D, Rg - D)
  } else {
    rc = lbm_rows_copy(post, &sl->g, R - 2 * D, so, &sl->sg, 2 * D, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sl->stash, &dg, 0, so, &sl->sg, D, D, s);      // the partner's rows [D, 2D)
  }
  return rc;
}

// ---- KBC start-up (ulbm_poiseuille.cpp:85-139 on the initial state: adve_f as given, HELD moments m0 [R][C], m1 [2][R][C]
// of the slab's owned rows).  Across the pressure seam travel the 2 D rows next to it and their moments (12 planes);
// across ordinary seams nothing is needed (the message is the pre-collision halo, unused).  The collision with held
// moments is not node-local in what it needs (the moments of ghost rows live on the neighbour), so ghost rows of `post`
// are NOT current afterwards: exchange complete halos of `post` (LBM_HALO_FULL(2)) over the ordinary seams before the
// first block -- lbm_ring_pressure_start_kbc does.
static int rows_copy_plane(double* dst, const double* src, int n_rows, int C, hipStream_t st) {
  LBM_CHECK_HIP(hipMemcpyAsync(dst, src, (size_t)n_rows * C * sizeof(double), hipMemcpyDeviceToDevice, st));
  return LBM_OK;
}

int lbm_slab_pressure_start_pack_kbc(lbm_slab_pressure* sl, const double* pre, const double* m0, const double* m1,
                                     double* send_prev, double* send_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && pre && m0 && m1 && send_prev && send_next, "lbm_slab_pressure_start_pack_kbc: NULL argument");
  LBM_REQUIRE(sl->model == LBM_MODEL_KBC, "lbm_slab_pressure_start_pack_kbc: not a KBC slab");
  const int R = sl->g.R, C = sl->g.C, D = sl->D, full = LBM_HALO_FULL(D);
  hipStream_t st = as_stream(s);
  const lbm_geom mg = msg_geom(2 * D, C);
  const size_t n = (size_t)R * C, rows = (size_t)2 * D * C;
  int rc = LBM_OK;
  if (sl->first) {
    rc = lbm_rows_copy(send_prev, &mg, 0, pre, &sl->g, 0, 2 * D, s);
    if (!rc) rc = rows_copy_plane(send_prev + 9 * rows, m0, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(send_prev + 10 * rows, m1, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(send_prev + 11 * rows, m1 + n, 2 * D, C, st);
  } else {
    rc = lbm_halo_pack(send_prev, pre, &sl->g, full, 0, s);
  }
  if (rc) return rc;
  if (sl->last) {
    const size_t o = (size_t)(R - 2 * D) * C;
    rc = lbm_rows_copy(send_next, &mg, 0, pre, &sl->g, R - 2 * D, 2 * D, s);
    if (!rc) rc = rows_copy_plane(send_next + 9 * rows, m0 + o, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(send_next + 10 * rows, m1 + o, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(send_next + 11 * rows, m1 + n + o, 2 * D, C, st);
  } else {
    rc = lbm_halo_pack(send_next, pre, &sl->g, full, 1, s);
  }
  return rc;
}

int lbm_slab_pressure_start_finish_kbc(lbm_slab_pressure* sl, double* post, double* pre, const double* m0, const double* m1,
                                       const double* recv_prev, const double* recv_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && post && pre && post != pre && m0 && m1 && recv_prev && recv_next, "lbm_slab_pressure_start_finish_kbc: bad argument");
  LBM_REQUIRE(sl->model == LBM_MODEL_KBC, "lbm_slab_pressure_start_finish_kbc: not a KBC slab");
  const int R = sl->g.R, C = sl->g.C, D = sl->D, G = sl->g.ghost;
  hipStream_t st = as_stream(s);
  // the owned rows on their held moments (kbc::collide with the driver's m0, m1, ulbm.cpp:91-126)
  const lbm_geom own{R, C, 0, plane_of(sl->g)};
  int rc = lbm_kbc_collide_first(post + (size_t)G * C, pre + (size_t)G * C, m0, m1, &own, nullptr, &sl->kprm, s);
  if (rc || (!sl->first && !sl->last)) return rc;
  // small lattice, pre-collision: rows [0, 2D) of the domain, then rows [Rg - 2D, Rg); its held moments likewise
  const lbm_geom mg = msg_geom(2 * D, C);
  const size_t n = (size_t)R * C, rows = (size_t)2 * D * C, sn = (size_t)4 * D * C;
  double* sp = sl->slat[sl->scur];
  if (sl->first) {
    rc = lbm_rows_copy(sp, &sl->sg, 0, pre, &sl->g, 0, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sp, &sl->sg, 2 * D, recv_prev, &mg, 0, 2 * D, s);
    if (!rc) rc = rows_copy_plane(sl->sm0, m0, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1, m1, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1 + sn, m1 + n, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm0 + rows, recv_prev + 9 * rows, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1 + rows, recv_prev + 10 * rows, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1 + sn + rows, recv_prev + 11 * rows, 2 * D, C, st);
  } else {
    const size_t o = (size_t)(R - 2 * D) * C;
    rc = lbm_rows_copy(sp, &sl->sg, 2 * D, pre, &sl->g, R - 2 * D, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sp, &sl->sg, 0, recv_next, &mg, 0, 2 * D, s);
    if (!rc) rc = rows_copy_plane(sl->sm0 + rows, m0 + o, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1 + rows, m1 + o, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1 + sn + rows, m1 + n + o, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm0, recv_next + 9 * rows, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1, recv_next + 10 * rows, 2 * D, C, st);
    if (!rc) rc = rows_copy_plane(sl->sm1 + sn, recv_next + 11 * rows, 2 * D, C, st);
  }
  double* so = sl->slat[sl->scur ^ 1];
  if (!rc) rc = lbm_kbc_collide_first(so, sp, sl->sm0, sl->sm1, &sl->sg, &sl->bc_seam, &sl->kprm, s);  // incl. the pressure rows (:36-58)
  if (rc) return rc;
  sl->scur ^= 1;
  const lbm_geom dg = msg_geom(D, C);
  if (sl->first) {
    rc = lbm_rows_copy(post, &sl->g, 0, so, &sl->sg, 0, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sl->stash, &dg, 0, so, &sl->sg, 2 * D, D, s);
  } else {
    rc = lbm_rows_copy(post, &sl->g, R - 2 * D, so, &sl->sg, 2 * D, 2 * D, s);
    if (!rc) rc = lbm_rows_copy(sl->stash, &dg, 0, so, &sl->sg, D, D, s);
  }
  return rc;
}

// one block of D steps, phase A: dst from src on the owned rows, both outgoing messages packed
int lbm_slab_pressure_block_compute(lbm_slab_pressure* sl, double* dst, const double* src, double* send_prev,
                                    double* send_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && dst && src && dst != src && send_prev && send_next, "lbm_slab_pressure_block_compute: bad argument");
  const int R = sl->g.R, C = sl->g.C, D = sl->D, full = LBM_HALO_FULL(D);
  hipStream_t st = as_stream(s);
  const bool end = sl->first || sl->last;
  int rc = LBM_OK;
  if (end) {
    LBM_CHECK_HIP(hipEventRecord(sl->ev_fork, st));
    LBM_CHECK_HIP(hipStreamWaitEvent(sl->aux, sl->ev_fork, 0));
    double* sp = sl->slat[sl->scur];
    const lbm_geom dg = msg_geom(D, C);
    // the rows at distance [D, 2D) from the seam at time t: mine from the slab, the partner's from the stash
    if (sl->first) {
      rc = lbm_rows_copy(sp, &sl->sg, D, src, &sl->g, D, D, sl->aux);
      if (!rc) rc = lbm_rows_copy(sp, &sl->sg, 2 * D, sl->stash, &dg, 0, D, sl->aux);
    } else {
      rc = lbm_rows_copy(sp, &sl->sg, 2 * D, src, &sl->g, R - 2 * D, D, sl->aux);
      if (!rc) rc = lbm_rows_copy(sp, &sl->sg, D, sl->stash, &dg, 0, D, sl->aux);
    }
    for (int k = 0; k < D && !rc; ++k) {
      rc = sl->model == LBM_MODEL_KBC
               ? lbm_kbc_stream_collide(sl->slat[sl->scur ^ 1], sl->slat[sl->scur], &sl->sg, &sl->bc_seam, &sl->kprm, 0, 4 * D, nullptr, nullptr, sl->aux)
               : lbm_bgk_stream_collide(sl->slat[sl->scur ^ 1], sl->slat[sl->scur], &sl->sg, &sl->bc_seam, &sl->prm, 0, 4 * D, nullptr,
                                        nullptr, sl->aux);
      sl->scur ^= 1;
    }
    if (rc) return rc;
    // my D rows next to the seam (rows [0, D) and [3D, 4D) of the small lattice are valid)
    rc = sl->first ? lbm_rows_copy(dst, &sl->g, 0, sl->slat[sl->scur], &sl->sg, 0, D, sl->aux)
                   : lbm_rows_copy(dst, &sl->g, R - D, sl->slat[sl->scur], &sl->sg, 3 * D, D, sl->aux);
    if (rc) return rc;
    LBM_CHECK_HIP(hipEventRecord(sl->ev_join, sl->aux));
  }
  // far rows: the D-step window in the reference operation order (lattices with pressure rows keep it on every path)
  const int r0 = sl->first ? D : 0, r1 = sl->last ? R - D : R;
  rc = sl->model == LBM_MODEL_KBC ? kbc_stream_collide_x2_ref(dst, src, &sl->g, &sl->bc_far, &sl->kprm, r0, r1, st)
                                  : bgk_stream_collide_xn_ref(dst, src, &sl->g, &sl->bc_far, &sl->prm, D, r0, r1, st);
  if (rc) return rc;
  if (end) LBM_CHECK_HIP(hipStreamWaitEvent(st, sl->ev_join, 0));
  const lbm_geom dg = msg_geom(D, C);
  rc = sl->first ? lbm_rows_copy(send_prev, &dg, 0, dst, &sl->g, D, D, s) : lbm_halo_pack(send_prev, dst, &sl->g, full, 0, s);
  if (!rc) rc = sl->last ? lbm_rows_copy(send_next, &dg, 0, dst, &sl->g, R - 2 * D, D, s) : lbm_halo_pack(send_next, dst, &sl->g, full, 1, s);
  return rc;
}

// phase B: ordinary halos into the ghost rows of dst, the partner's rows into the stash
int lbm_slab_pressure_block_finish(lbm_slab_pressure* sl, double* dst, const double* recv_prev, const double* recv_next, lbm_stream_t s) {
  LBM_REQUIRE(sl && dst && recv_prev && recv_next, "lbm_slab_pressure_block_finish: NULL argument");
  const int full = LBM_HALO_FULL(sl->D);
  const size_t msg = (size_t)9 * sl->D * sl->g.C * sizeof(double);
  int rc = LBM_OK;
  if (sl->first) LBM_CHECK_HIP(hipMemcpyAsync(sl->stash, recv_prev, msg, hipMemcpyDeviceToDevice, as_stream(s)));
  else rc = lbm_halo_unpack(dst, recv_prev, &sl->g, full, 0, s);
  if (rc) return rc;
  if (sl->last) LBM_CHECK_HIP(hipMemcpyAsync(sl->stash, recv_next, msg, hipMemcpyDeviceToDevice, as_stream(s)));
  else rc = lbm_halo_unpack(dst, recv_next, &sl->g, full, 1, s);
  return rc;
}

}  // extern "C"
