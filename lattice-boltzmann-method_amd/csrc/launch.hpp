// Host-side launch logic shared by the single-phase models (BGK, KBC): choose the interior
// variant, launch it, then the edge pass that applies the boundary fix-ups.
#pragma once
#include <map>
#include <mutex>
#include <utility>
#include "d2q9.hpp"
#include "internal.hpp"

namespace lbm {

// reads_neighbours: the caller streams (pulls from neighbouring rows); collide-only entry points pass false
inline int validate_geom_bc(const char* fn, const lbm_geom* g, const lbm_bc* bc, bool reads_neighbours = true) {
  LBM_REQUIRE(g, "%s: NULL geometry", fn);
  LBM_REQUIRE(g->R >= 1 && g->C >= 1, "%s: R=%d C=%d must be positive", fn, g->R, g->C);
  LBM_REQUIRE(g->ghost >= 0 && g->ghost <= 15, "%s: ghost=%d must be 0..15", fn, g->ghost);
  LBM_REQUIRE(g->row_pitch == 0 || (g->row_pitch >= g->C && g->row_pitch % 2 == 0), "%s: row_pitch=%d must be even and >= C=%d (0 = dense)",
              fn, g->row_pitch, g->C);
  LBM_REQUIRE(g->plane_stride == 0 || g->plane_stride >= (long long)(g->R + 2 * g->ghost) * (g->row_pitch > 0 ? g->row_pitch : g->C),
              "%s: plane_stride=%lld smaller than a plane", fn, g->plane_stride);
  if (bc) {
    auto row_ok = [](int m) {
      return m == LBM_EDGE_PERIODIC || m == LBM_EDGE_HALO || m == LBM_EDGE_BOUNCE_BACK ||
             m == LBM_EDGE_ABB_VELOCITY;
    };
    auto col_ok = [](int m) {
      return m == LBM_EDGE_PERIODIC || m == LBM_EDGE_BOUNCE_BACK || m == LBM_EDGE_SPECULAR ||
             m == LBM_EDGE_WRAP_NOSHIFT;
    };
    LBM_REQUIRE(row_ok(bc->row_lo) && row_ok(bc->row_hi), "%s: unsupported row edge mode %d/%d", fn,
                bc->row_lo, bc->row_hi);
    LBM_REQUIRE(col_ok(bc->col_lo) && col_ok(bc->col_hi), "%s: unsupported column edge mode %d/%d",
                fn, bc->col_lo, bc->col_hi);
    LBM_REQUIRE(bc->pressure_rows == 0 || bc->pressure_rows == 1, "%s: pressure_rows=%d", fn,
                bc->pressure_rows);
    LBM_REQUIRE(!bc->pressure_rows || g->R >= 3, "%s: pressure rows need R >= 3", fn);
    if (g->ghost == 0)
      LBM_REQUIRE(bc->row_lo != LBM_EDGE_HALO && bc->row_hi != LBM_EDGE_HALO,
                  "%s: HALO rows need ghost rows (ghost >= 1)", fn);
  }
  // with ghost rows nothing wraps: a PERIODIC row edge (NULL bc = all periodic) would silently read
  // ghost rows nobody fills
  if (reads_neighbours && g->ghost > 0)
    LBM_REQUIRE(bc && bc->row_lo != LBM_EDGE_PERIODIC && bc->row_hi != LBM_EDGE_PERIODIC,
                "%s: a lattice with ghost rows needs HALO or wall row edges (PERIODIC given)", fn);
  return LBM_OK;
}

// p_new = collide(stream(p_old)) on rows [row_begin, row_end).
template <class Model>
int launch_stream_collide(const char* fn, double* pn, const double* po, const lbm_geom* lg,
                          const lbm_bc* lbc, const Model& m, int row_begin, int row_end,
                          double* rho, double* u, hipStream_t st) {
  int rc = validate_geom_bc(fn, lg, lbc);
  if (rc) return rc;
  LBM_REQUIRE(pn && po && pn != po, "%s: NULL or aliased lattices", fn);
  LBM_REQUIRE((rho == nullptr) == (u == nullptr), "%s: rho and u must both be given or both NULL", fn);
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= lg->R,
              "%s: row range [%d, %d) outside [0, %d)", fn, row_begin, row_end, lg->R);
  if (row_begin == row_end) return LBM_OK;
  const Geom g = make_geom(*lg);
  const Bc bc = make_bc(lbc);
  const bool mom = rho != nullptr;
  const int nrows = row_end - row_begin;
  const int variant = tuning("variant", 3);
  const bool fast = (g.C % 2 == 0) && g.C >= 64 && variant != 0;

  if (!fast) {  // generic path: boundary gather on every node
    const long n = (long)nrows * g.C;
    const int grid = capped_grid((n + 255) / 256);
    if (mom) LBM_KLAUNCH((k_generic_collide<Model, true, true>), dim3(grid), dim3(256), 0, st, pn, po, g, bc, m, row_begin, row_end, rho, u);
    else LBM_KLAUNCH((k_generic_collide<Model, true, false>), dim3(grid), dim3(256), 0, st, pn, po, g, bc, m, row_begin, row_end, rho, u);
    LBM_CHECK_LAUNCH();
    return LBM_OK;
  }

  const int nt = tuning("nt", 3);  // bit 0: non-temporal loads, bit 1: non-temporal stores
  const int cap = tuning("grid_cap", 0);
#define LBM_LAUNCH_VARIANT(KERNEL, NODES_PER_THREAD)                                            \
  do {                                                                                          \
    const int tiles = (g.C + 256 * NODES_PER_THREAD - 1) / (256 * NODES_PER_THREAD);            \
    const long items = (long)nrows * tiles;                                                     \
    const int grid = cap > 0 ? capped_grid(items, cap) : (int)(items < (1L << 30) ? items : (1L << 30)); \
    switch ((nt & 3) | (mom ? 4 : 0)) {                                                         \
      case 0: LBM_KLAUNCH((KERNEL<Model, false, false, false>), dim3(grid), dim3(256), 0, st, pn, po, g, m, row_begin, row_end, tiles, rho, u); break; \
      case 1: LBM_KLAUNCH((KERNEL<Model, true, false, false>), dim3(grid), dim3(256), 0, st, pn, po, g, m, row_begin, row_end, tiles, rho, u); break;  \
      case 2: LBM_KLAUNCH((KERNEL<Model, false, true, false>), dim3(grid), dim3(256), 0, st, pn, po, g, m, row_begin, row_end, tiles, rho, u); break;  \
      case 3: LBM_KLAUNCH((KERNEL<Model, true, true, false>), dim3(grid), dim3(256), 0, st, pn, po, g, m, row_begin, row_end, tiles, rho, u); break;   \
      default: LBM_KLAUNCH((KERNEL<Model, false, false, true>), dim3(grid), dim3(256), 0, st, pn, po, g, m, row_begin, row_end, tiles, rho, u); break; \
    }                                                                                           \
  } while (0)
  if (variant == 3 && !mom) {
    const int block = tuning("block", 256), rows = tuning("rows", 1);
    const int swz = tuning("xcd_swizzle", 0);  // measured slower at 8192^2
#define LBM_V3(B, RW)                                                                              \
  if (block == B && rows == RW) {                                                                  \
    const int tiles_x = (g.C + B - 1) / B;                                                         \
    const long items_l = (long)tiles_x * ((nrows + RW - 1) / RW);                                  \
    LBM_REQUIRE(items_l < (1L << 30), "%s: lattice too large for one launch", fn);                 \
    const int n_items = (int)items_l;                                                              \
    const dim3 grid(swz ? ((n_items + 7) / 8) * 8 : n_items);                                      \
    switch (nt & 3) {                                                                              \
      case 0: LBM_KLAUNCH((k_stream_collide_v3<Model, B, RW, false, false>), grid, dim3(B), 0, st, pn, po, g, m, row_begin, row_end, tiles_x, n_items, swz); break; \
      case 1: LBM_KLAUNCH((k_stream_collide_v3<Model, B, RW, true, false>), grid, dim3(B), 0, st, pn, po, g, m, row_begin, row_end, tiles_x, n_items, swz); break;  \
      case 2: LBM_KLAUNCH((k_stream_collide_v3<Model, B, RW, false, true>), grid, dim3(B), 0, st, pn, po, g, m, row_begin, row_end, tiles_x, n_items, swz); break;  \
      default: LBM_KLAUNCH((k_stream_collide_v3<Model, B, RW, true, true>), grid, dim3(B), 0, st, pn, po, g, m, row_begin, row_end, tiles_x, n_items, swz); break;  \
    }                                                                                              \
  } else
    LBM_V3(128, 1) LBM_V3(256, 1) LBM_V3(512, 1) LBM_V3(1024, 1) LBM_V3(128, 2) LBM_V3(256, 2)
    LBM_V3(512, 2) LBM_V3(256, 4) LBM_V3(128, 4) {
      set_error("%s: no v3 instantiation for block=%d rows=%d", fn, block, rows);
      return LBM_ERR_INVALID;
    }
#undef LBM_V3
  } else if (variant == 1) LBM_LAUNCH_VARIANT(k_stream_collide_v1, 1);
  else LBM_LAUNCH_VARIANT(k_stream_collide_v2, 2);
#undef LBM_LAUNCH_VARIANT
  LBM_CHECK_LAUNCH();

  if (bc_needs_edge_pass(bc)) {
    const int n_edge = 2 * g.C + 2 * (row_end - row_begin);
    if (mom) LBM_KLAUNCH((k_edge_stream_collide<Model, true>), dim3((n_edge + 255) / 256), dim3(256), 0, st, pn, po, g, bc, m, row_begin, row_end, rho, u);
    else LBM_KLAUNCH((k_edge_stream_collide<Model, false>), dim3((n_edge + 255) / 256), dim3(256), 0, st, pn, po, g, bc, m, row_begin, row_end, rho, u);
    LBM_CHECK_LAUNCH();
  }
  return LBM_OK;
}

// p_new = two steps from p_old (temporal blocking); rows [row_begin, row_end), periodic / ghost edges
template <class Model>
int launch_stream_collide_x2(const char* fn, double* pn, const double* po, const lbm_geom* lg,
                             const lbm_bc* lbc, const Model& m, int row_begin, int row_end,
                             hipStream_t st) {
  int rc = validate_geom_bc(fn, lg, lbc);
  if (rc) return rc;
  LBM_REQUIRE(pn && po && pn != po, "%s: NULL or aliased lattices", fn);
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= lg->R,
              "%s: row range [%d, %d) outside [0, %d)", fn, row_begin, row_end, lg->R);
  const Bc bc = make_bc(lbc);
  LBM_REQUIRE(!bc_needs_edge_pass(bc) && !bc.pressure_rows,
              "%s: two-step launches support periodic / halo edges only", fn);
  LBM_REQUIRE(lg->C % 64 == 0, "%s: C=%d must be a multiple of 64", fn, lg->C);
  LBM_REQUIRE(lg->ghost == 0 || lg->ghost == 2, "%s: ghost=%d (two-step launches need 0 or 2 ghost rows)", fn, lg->ghost);
  if (row_begin == row_end) return LBM_OK;
  const Geom g = make_geom(*lg);
  const int tiles_x = g.C / 64, nrows = row_end - row_begin;
  const int tr = tuning("tb_rows", 8), block = tuning("tb_block", 512), nt = tuning("nt", 3) & 2;
  const int order = (tiles_x % 8 == 0) ? tuning("tb_order", 0) : 0;
#define LBM_TB2(TRV, BV)                                                                          \
  if (tr == TRV && block == BV) {                                                                 \
    const int tiles_y = (nrows + TRV - 1) / TRV;                                                  \
    const long nblk = (long)tiles_x * tiles_y;                                                    \
    LBM_REQUIRE(nblk < (1L << 30), "%s: lattice too large for one launch", fn);                   \
    if (nt) LBM_KLAUNCH((k_stream_collide_tb2<Model, TRV, BV, true>), dim3((unsigned)nblk), dim3(BV), 0, st, pn, po, g, m, row_begin, row_end, tiles_x, tiles_y, order); \
    else LBM_KLAUNCH((k_stream_collide_tb2<Model, TRV, BV, false>), dim3((unsigned)nblk), dim3(BV), 0, st, pn, po, g, m, row_begin, row_end, tiles_x, tiles_y, order);   \
  } else
  LBM_TB2(4, 256) LBM_TB2(6, 256) LBM_TB2(8, 256) LBM_TB2(8, 512) LBM_TB2(12, 256) LBM_TB2(12, 512)
  LBM_TB2(14, 512) LBM_TB2(16, 512) LBM_TB2(16, 1024) LBM_TB2(30, 1024) {
    set_error("%s: no two-step instantiation for tb_rows=%d tb_block=%d", fn, tr, block);
    return LBM_ERR_INVALID;
  }
#undef LBM_TB2
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

// Rows per wave of a sliding-window launch.  A wave walks `rows` output rows after 2 (D - 1) warm-up
// rows; the launch runs in ceil(waves / resident wave slots) rounds and a partly filled last round
// costs a whole chunk.  With thousands of waves per slot (the 8192^2 headline: measured flat from 48
// to 160 rows) that tail is noise; at a few waves per slot (4096^2 KBC at one wave per SIMD: 4.3
// rounds at 64 rows) it is 20 % of the launch -- so pick the chunk height that fills the last round.
inline int sw_pick_rows(int nrows, int strips, int depth, long slots) {
  auto cost = [&](int rows) {
    const long waves = (long)strips * ((nrows + rows - 1) / rows);
    const long rounds = (waves + slots - 1) / slots;
    // few long rounds balance worse under dynamic dispatch than many short ones -- but only when they are FEW:
    // with the penalty at 0.25 / rounds the 8192^2 headline got 108-row chunks (10.9 rounds, 7.4 % warm-up rows)
    // where 172 rows (6.9 rounds, 4.7 %) measure 2-3 % faster: the launch is HBM-bound and warm-up rows are traffic
    return (double)rounds * (rows + 2 * (depth - 1)) * (1.0 + 0.25 / ((double)rounds * (double)rounds));
  };
  int best = nrows < 64 ? nrows : 64;
  double best_cost = cost(best) * 0.95;  // leave the default unless the gain is worth having
  for (int rows = 32; rows <= 256 && rows <= nrows; rows += 4) {
    const double c = cost(rows);
    if (c < best_cost) best = rows, best_cost = c;
  }
  return best;
}

// resident wavefronts of one kernel instance on the current device (cached per instance and device)
inline long sw_wave_slots(const void* kernel, int block_threads) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, long> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find({kernel, dev});
  if (it != cache.end()) return it->second;
  int blocks_per_cu = 0, cus = 0;
  long slots = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kernel, block_threads, 0) == hipSuccess &&
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && blocks_per_cu > 0 && cus > 0)
    slots = (long)blocks_per_cu * (block_threads / 64) * cus;
  else
    (void)hipGetLastError();
  cache[{kernel, dev}] = slots;
  return slots;
}

// one rectangle (rows [r0, r1) x strips [s0, s0 + ns)) of a sliding-window launch through the 2-wave
// instantiation with or without the wall fix-ups
template <class Model, int DV, bool HAS_BC>
void sw_launch_part(double* pn, const double* po, const Geom& g, const Model& m, const Bc& bc, int r0,
                    int r1, int s0, int ns, int rows_fixed, hipStream_t st) {
  if (r0 >= r1 || ns <= 0) return;
  const int nrows = r1 - r0;
  int rpc = rows_fixed > 0 ? rows_fixed : tuning("sw_rows", -1);
  if (rpc <= 0) {
    const long slots = sw_wave_slots((const void*)k_stream_collide_sw<Model, DV, 2, true, HAS_BC>, 128);
    rpc = slots > 0 ? sw_pick_rows(nrows, ns, DV, slots) : 64;
  }
  if (rpc > nrows) rpc = nrows;
  const int n_waves = ns * ((nrows + rpc - 1) / rpc);
  LBM_KLAUNCH((k_stream_collide_sw<Model, DV, 2, true, HAS_BC>), dim3((n_waves + 1) / 2), dim3(128), 0, st, pn, po, g, m,
              r0, r1, rpc, ns, n_waves, 0, bc, s0);
}

// Helper stream of the calling host thread on the current device: the few waves of a wall frame run
// there, beside the interior launch on the caller's stream (fork / join through two events).  Lives
// until the process ends; one per (thread, device).
struct SwSideStream {
  hipStream_t st = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int dev = -1;
};
inline SwSideStream* sw_side_stream() {
  static thread_local SwSideStream side[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SwSideStream& s = side[dev];
  if (s.dev != dev) {
    if (hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    s.dev = dev;
  }
  return &s;
}

// Wall-bounded multi-step launch.  Walls touch few waves: a strip whose 64-lane window (outputs + the
// D-1 halo lanes on each side) holds no wall column, on rows at least D away from a wall row, runs the
// PLAIN instantiation -- same arithmetic, no fix-up code, fewer registers -- and only the frame around
// that rectangle (the outermost strips, 16 rows next to a wall row) goes through the wall-carrying one,
// enqueued first and on the helper stream so that its waves hold their slots while the interior grid
// fills the rest.  "sw_split" = 0: everything through the wall-carrying instantiation.
template <class Model, int DV>
int sw_launch_walls(double* pn, const double* po, const Geom& g, const Model& m, const Bc& bc,
                    int row_begin, int row_end, hipStream_t st) {
  constexpr int W = sw_strip_width(DV, sw_full_strips<Model>::value);
  const int strips = (g.C + W - 1) / W;
  const bool col_walls = bc_is_wall(bc.col_lo) || bc_is_wall(bc.col_hi);
  int ra = row_begin, rb = row_end, s0 = 0, s1 = strips;
  if (bc_is_wall(bc.row_lo)) ra = row_begin > 16 ? row_begin : 16;
  if (bc_is_wall(bc.row_hi)) rb = row_end < g.R - 16 ? row_end : g.R - 16;
  if (col_walls) {
    s0 = 1;
    s1 = (g.C - W - DV) / W + 1;  // last strip with s W + W - 1 + (D - 1) <= C - 2
    if (s1 > strips) s1 = strips;
  }
  // frame / interior split: 1 = two launches (frame first, on a helper stream: it runs BESIDE the interior, at the price of
  // an event fork / join, ~0.1 ms), 2 = one dispatch holding both instantiations (no events; the interior compiled inside the
  // larger register budget runs ~3 % slower).  Measured (profiles/r02_walls_split.log): a whole wall-bounded 8192^2 block 158.0 k
  // MLUPS with 1, 153.7 k with 2; the far rows of an immersed-boundary block (two such launches per block beside the band
  // chain) 94.5 k / 55.7 k / 34.7 k with 1, 102.2 k / 62.9 k / 42.4 k with 2 (16384x4096 / 4096^2 / 2048x4096); the Poiseuille
  // channel's far rows 150.6 k with 1, 156.1 k with 2.  Default: 1 for a launch over a whole single block, 2 for row ranges.
  int split = tuning("sw_split", -1);
  if (split < 0) split = (row_begin == 0 && row_end == g.R && !g.ghost) ? 1 : 2;
  if (split == 0 || ra >= rb || s0 >= s1) {
    sw_launch_part<Model, DV, true>(pn, po, g, m, bc, row_begin, row_end, 0, strips, 0, st);
    return LBM_OK;
  }
  if (split == 2) {  // frame and interior in ONE dispatch (d2q9.hpp k_stream_collide_sw_walls)
    SwParts pp{};
    const int fr = 8;  // frame chunks: 8 rows after 2 (D - 1) warm-up rows, as in the two-launch form
    auto part = [&](int k, int r0, int r1, int s0f, int ns, int rpc, int& wave0) {
      if (r1 < r0) r1 = r0;
      if (ns < 0) ns = 0;
      if (rpc > r1 - r0) rpc = r1 - r0;
      pp.p[k] = SwPart{r0, r1, s0f, ns, rpc > 0 ? rpc : 1, wave0};
      if (r1 > r0 && ns > 0) wave0 += ns * ((r1 - r0 + pp.p[k].rpc - 1) / pp.p[k].rpc);
    };
    int w0 = 0;
    part(0, row_begin, row_end, 0, s0, fr, w0);
    part(1, row_begin, row_end, s1, strips - s1, fr, w0);
    part(2, row_begin, ra, s0, s1 - s0, fr, w0);
    part(3, rb, row_end, s0, s1 - s0, fr, w0);
    pp.n_frame_waves = w0;
    int rpc = tuning("sw_rows", -1);
    if (rpc <= 0) {
      const long slots = sw_wave_slots((const void*)k_stream_collide_sw_walls<Model, DV, true>, 128);
      rpc = slots > 0 ? sw_pick_rows(rb - ra, s1 - s0, DV, slots) : 64;
    }
    part(4, ra, rb, s0, s1 - s0, rpc, w0);
    pp.n_waves = w0;
    LBM_KLAUNCH((k_stream_collide_sw_walls<Model, DV, true>), dim3((pp.n_waves + 1) / 2), dim3(128), 0, st, pn, po, g, m, bc, pp);
    return LBM_OK;
  }
  SwSideStream* sd = sw_side_stream();
  hipStream_t fs = st;
  if (sd && hipEventRecord(sd->fork, st) == hipSuccess && hipStreamWaitEvent(sd->st, sd->fork, 0) == hipSuccess) fs = sd->st;
  // frame waves are few and latency-bound: short chunks (8 rows after 2 (D - 1) warm-up rows) keep each
  // of these launches to one brief round -- on a short row range they would otherwise outlast the interior
  sw_launch_part<Model, DV, true>(pn, po, g, m, bc, row_begin, row_end, 0, s0, 8, fs);
  sw_launch_part<Model, DV, true>(pn, po, g, m, bc, row_begin, row_end, s1, strips - s1, 8, fs);
  sw_launch_part<Model, DV, true>(pn, po, g, m, bc, row_begin, ra, s0, s1 - s0, 8, fs);
  sw_launch_part<Model, DV, true>(pn, po, g, m, bc, rb, row_end, s0, s1 - s0, 8, fs);
  sw_launch_part<Model, DV, false>(pn, po, g, m, bc, ra, rb, s0, s1 - s0, 0, st);
  if (fs != st) {
    LBM_CHECK_HIP(hipEventRecord(sd->join, fs));
    LBM_CHECK_HIP(hipStreamWaitEvent(st, sd->join, 0));
  }
  return LBM_OK;
}

// p_new = D steps from p_old with the register sliding-window kernel; rows [row_begin, row_end)
template <class Model>
int launch_stream_collide_sw(const char* fn, double* pn, const double* po, const lbm_geom* lg,
                             const lbm_bc* lbc, const Model& m, int depth, int row_begin,
                             int row_end, hipStream_t st, int default_waves = 4, int second_begin = -1) {
  // second_begin >= 0: ALSO rows [second_begin, second_begin + (row_end - row_begin)) in the same
  // launch (the two edge-row ranges of a slab: one dispatch instead of two serialised ones)
  if (second_begin >= 0) {
    LBM_REQUIRE(lg && second_begin >= row_end && second_begin + (row_end - row_begin) <= lg->R && row_end - row_begin <= 192,
                "%s: second row range [%d, +%d) must follow the first and fit the lattice", fn, second_begin, row_end - row_begin);
    const Bc b2 = make_bc(lbc);
    if (bc_needs_edge_pass(b2)) {  // wall-carrying launches keep their frame / interior split: two calls
      int rc2 = launch_stream_collide_sw(fn, pn, po, lg, lbc, m, depth, row_begin, row_end, st, default_waves);
      if (rc2) return rc2;
      return launch_stream_collide_sw(fn, pn, po, lg, lbc, m, depth, second_begin, second_begin + (row_end - row_begin), st, default_waves);
    }
  }
  int rc = validate_geom_bc(fn, lg, lbc);
  if (rc) return rc;
  LBM_REQUIRE(pn && po && pn != po, "%s: NULL or aliased lattices", fn);
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= lg->R,
              "%s: row range [%d, %d) outside [0, %d)", fn, row_begin, row_end, lg->R);
  const Bc bc = make_bc(lbc);
  auto carried = [](int m) { return m == LBM_EDGE_PERIODIC || m == LBM_EDGE_HALO || bc_is_wall(m); };
  LBM_REQUIRE(carried(bc.row_lo) && carried(bc.row_hi) && carried(bc.col_lo) && carried(bc.col_hi) && !bc.pressure_rows,
              "%s: multi-step launches carry periodic / halo / bounce-back / specular / velocity edges only", fn);
  LBM_REQUIRE(!bc_mixed_axis(bc), "%s: multi-step launches need both edges of an axis walled or neither (wall + PERIODIC on one axis: use single steps)", fn);
  const bool walls = bc_needs_edge_pass(bc);
  // (over slabs the ghost rows must then be COMPLETE: exchange with LBM_HALO_FULL(depth))
  LBM_REQUIRE(depth >= 2 && depth <= (walls ? 5 : 6), "%s: %d steps per launch (supported: 2..%d)", fn, depth, walls ? 5 : 6);
  LBM_REQUIRE(lg->ghost == 0 || lg->ghost >= depth, "%s: ghost=%d rows, need 0 or >= %d", fn, lg->ghost, depth);
  LBM_REQUIRE(lg->R >= 4 * depth + 8 && lg->C >= 64, "%s: lattice %dx%d too small for %d-step launches", fn, lg->R, lg->C, depth);
  if (row_begin == row_end) return LBM_OK;
  const Geom g = make_geom(*lg);
  const int nrows = row_end - row_begin;
  const int W = sw_strip_width(depth, sw_full_strips<Model>::value);
  const int strips = (g.C + W - 1) / W;
  LBM_REQUIRE((long)strips * ((nrows + 31) / 32) < (1L << 30), "%s: lattice too large for one launch", fn);
  const int nt = tuning("nt", 3) & 2, waves = tuning("sw_waves", default_waves);
  int rpc = 0, n_waves = 0;
  // sw_rows > 0: fixed chunk height; unset: chosen per kernel instance from its resident wave slots
  auto plan = [&](const void* kernel, int block_threads) {
    rpc = tuning("sw_rows", -1);
    if (rpc <= 0) {
      const long slots = sw_wave_slots(kernel, block_threads);
      rpc = slots > 0 ? sw_pick_rows(nrows, strips, depth, slots) : 64;
    }
    if (rpc > nrows) rpc = nrows;
    n_waves = strips * ((nrows + rpc - 1) / rpc);
    if (second_begin >= 0) rpc = nrows, n_waves = 2 * strips;  // one chunk per range
  };
  const int chunk_stride = second_begin >= 0 ? second_begin - row_begin : 0;
  const int row_end_k = second_begin >= 0 ? second_begin + nrows : row_end;  // the kernel clips chunks at this row
  if (walls) {  // wall-carrying variant: 2-wave blocks only (no register cap: 4-wave blocks, capped at
                // 168 / 256 VGPRs, spill up to 1300 registers with the fix-ups in)
#define LBM_SWBC(DV)                                                                              \
  if (depth == DV) {                                                                              \
    rc = sw_launch_walls<Model, DV>(pn, po, g, m, bc, row_begin, row_end, st);                    \
    if (rc) return rc;                                                                            \
  } else
    LBM_SWBC(2) LBM_SWBC(3) LBM_SWBC(4) LBM_SWBC(5) {
      set_error("%s: no wall-carrying sliding-window instantiation for depth=%d", fn, depth);
      return LBM_ERR_INVALID;
    }
#undef LBM_SWBC
    LBM_CHECK_LAUNCH();
    return LBM_OK;
  }
  bool launched = false;
#ifdef LBM_EXPERIMENTS
  if constexpr (std::is_same<Model, BgkFastModel>::value || std::is_same<Model, BgkModelT<0, 0>>::value) {
    // paired strips (d2q9.hpp k_stream_collide_swp): the 2 / 4 waves of a workgroup hand each other their edge columns
    const int pw = tuning("sw_pair", 0);
    if ((pw == 2 || pw == 4) && depth == 5 && nt && g.C >= 256) {
      const int GW = swp_group_width(5, pw), groups = (g.C + GW - 1) / GW;
      int rows = tuning("sw_rows", -1);
      if (second_begin >= 0) rows = nrows;
      else if (rows <= 0) {
        const long slots = pw == 2 ? sw_wave_slots((const void*)k_stream_collide_swp<Model, 5, 2, true>, 128)
                                   : sw_wave_slots((const void*)k_stream_collide_swp<Model, 5, 4, true>, 256);
        rows = slots > 0 ? sw_pick_rows(nrows, groups * pw, 7, slots) : 64;  // 12 pipeline rows per chunk ~ depth 7
      }
      if (rows > nrows) rows = nrows;
      const int chunks = second_begin >= 0 ? 2 : (nrows + rows - 1) / rows, total = groups * chunks;
      if (pw == 2) LBM_KLAUNCH((k_stream_collide_swp<Model, 5, 2, true>), dim3(total), dim3(128), 0, st, pn, po, g, m, row_begin, row_end_k, rows, groups, total, chunk_stride);
      else LBM_KLAUNCH((k_stream_collide_swp<Model, 5, 4, true>), dim3(total), dim3(256), 0, st, pn, po, g, m, row_begin, row_end_k, rows, groups, total, chunk_stride);
      launched = true;
    }
  }
  if constexpr (std::is_same<Model, BgkFastModel>::value) {
    // two columns per lane (d2q9.hpp / experiments/sw_two_columns.hpp): 16-byte accesses need even columns, pitch and plane stride
    if (!launched && tuning("sw_cols2", 0) && (depth == 5 || depth == 6) && nt && g.C % 2 == 0 && g.P % 2 == 0 && g.plane % 2 == 0 &&
        ((uintptr_t)pn % 16) == 0 && ((uintptr_t)po % 16) == 0) {
      const int W2 = sw2_strip_width(depth), strips2 = (g.C + W2 - 1) / W2;
      int rows = tuning("sw_rows", -1);
      const void* kp = depth == 5 ? (const void*)k_stream_collide_sw2<Model, 5, true> : (const void*)k_stream_collide_sw2<Model, 6, true>;
      if (second_begin >= 0) rows = nrows;
      else if (rows <= 0) {
        const long slots = sw_wave_slots(kp, 128);
        rows = slots > 0 ? sw_pick_rows(nrows, strips2, depth, slots) : 64;
      }
      if (rows > nrows) rows = nrows;
      const int nw = second_begin >= 0 ? 2 * strips2 : strips2 * ((nrows + rows - 1) / rows);
      if (depth == 5) LBM_KLAUNCH((k_stream_collide_sw2<Model, 5, true>), dim3((nw + 1) / 2), dim3(128), 0, st, pn, po, g, m, row_begin, row_end_k, rows, strips2, nw, chunk_stride);
      else LBM_KLAUNCH((k_stream_collide_sw2<Model, 6, true>), dim3((nw + 1) / 2), dim3(128), 0, st, pn, po, g, m, row_begin, row_end_k, rows, strips2, nw, chunk_stride);
      launched = true;
    }
  }
  if constexpr (std::is_same<Model, BgkFastModel>::value) {
    if (!launched && tuning("sw_pf2", 0) && depth == 5 && waves == 2 && nt) {  // level-1 rows prefetched two iterations ahead
      plan((const void*)k_stream_collide_sw<Model, 5, 2, true, false, true>, 128);
      LBM_KLAUNCH((k_stream_collide_sw<Model, 5, 2, true, false, true>), dim3((n_waves + 1) / 2), dim3(128), 0, st, pn, po, g, m,
                  row_begin, row_end_k, rpc, strips, n_waves, tuning("sw_xcd", 0), Bc{}, 0, chunk_stride);
      launched = true;
    }
  }
#endif  // LBM_EXPERIMENTS
  // (rounds 3 - 4, measured and removed: FOUR adjacent strips per workgroup with the register budget uncapped, so that the 128-byte lines
  // neighbouring strips share are fetched by waves of one compute unit -- 177.7 / 177.4 k against 177.0 / 176.9 k for the 2-wave blocks,
  // alternating on one box: within noise.  The capped 4-wave blocks spill at 5 steps: 85 k.)
  // (round 4, measured and not kept: the BGK window with its ring in wave-private LDS like the KBC window -- k_stream_collide_sw<BgkFastModel,
  // 5, 2, nt, LDSR>: 1119 instead of 1378 VALU instructions per three iterations (no lane shifts, no AGPR copies), 169 registers, six
  // waves per CU by LDS.  8192^2, alternating on one box at the 1400 W limit: 160.8 / 161.4 / 160.8 k with the register ring, 152.6 /
  // 152.8 / 152.3 k with the LDS ring at a 100 MHz lower clock.  profiles/r04_bgk_ldsring_ab.txt)
  if (!launched) {
#define LBM_SW(DV, WV)                                                                            \
  if (depth == DV && waves == WV) {                                                               \
    plan((const void*)k_stream_collide_sw<Model, DV, WV, true>, 64 * WV);                         \
    const dim3 grid((n_waves + WV - 1) / WV);                                                     \
    if (nt) LBM_KLAUNCH((k_stream_collide_sw<Model, DV, WV, true>), grid, dim3(64 * WV), 0, st, pn, po, g, m, row_begin, row_end_k, rpc, strips, n_waves, tuning("sw_xcd", 0), Bc{}, 0, chunk_stride); \
    else LBM_KLAUNCH((k_stream_collide_sw<Model, DV, WV, false>), grid, dim3(64 * WV), 0, st, pn, po, g, m, row_begin, row_end_k, rpc, strips, n_waves, tuning("sw_xcd", 0), Bc{}, 0, chunk_stride);   \
  } else
  LBM_SW(2, 4) LBM_SW(3, 4) LBM_SW(4, 4) LBM_SW(5, 4) LBM_SW(6, 4)
  LBM_SW(2, 1) LBM_SW(3, 1) LBM_SW(4, 1) LBM_SW(2, 2) LBM_SW(3, 2) LBM_SW(4, 2) LBM_SW(5, 2) LBM_SW(6, 2) {
    set_error("%s: no sliding-window instantiation for depth=%d sw_waves=%d", fn, depth, waves);
    return LBM_ERR_INVALID;
  }
  }
#undef LBM_SW
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

// P = collide(f) on a pre-collision lattice (first driver iteration).
template <class Model>
int launch_collide_only(const char* fn, double* p, const double* f, const lbm_geom* lg,
                        const lbm_bc* lbc, const Model& m, double* rho, double* u, hipStream_t st) {
  int rc = validate_geom_bc(fn, lg, lbc, false);
  if (rc) return rc;
  LBM_REQUIRE(p && f, "%s: NULL lattice", fn);
  LBM_REQUIRE((rho == nullptr) == (u == nullptr), "%s: rho and u must both be given or both NULL", fn);
  const Geom g = make_geom(*lg);
  const Bc bc = make_bc(lbc);
  const long n = (long)g.R * g.C;
  const int grid = capped_grid((n + 255) / 256);
  if (rho) LBM_KLAUNCH((k_generic_collide<Model, false, true>), dim3(grid), dim3(256), 0, st, p, f, g, bc, m, 0, g.R, rho, u);
  else LBM_KLAUNCH((k_generic_collide<Model, false, false>), dim3(grid), dim3(256), 0, st, p, f, g, bc, m, 0, g.R, rho, u);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

}  // namespace lbm
