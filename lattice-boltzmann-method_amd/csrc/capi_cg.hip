// C ABI, part 5: colour-gradient two-phase MRT step (test/mrtcg_rayleigh_taylor.cpp).
#include <new>

#include "cg.hpp"
#include "cg_fused.hpp"
#include "launch.hpp"

namespace lbm {

__global__ __launch_bounds__(256) void k_cg_equilibrium(double* __restrict__ f,
                                                        const double* __restrict__ rho_k,
                                                        const double* __restrict__ u, long n,
                                                        long plane, CgColour k) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double e[Q];
    cg_feq(e, rho_k[i], k, u[i], u[n + i]);
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q * plane + i] = e[q];
  }
}

// stand-alone differential::x / ::y (src/differential.cpp:23-33): 5x5 cross-correlation with
// replicate padding; dir 0 = d/d(row) ("x"), 1 = d/d(col) ("y").  Same tap order as the fused
// collide kernel.  Not on the hot path (the fused step reads its stencils from LDS).
__global__ __launch_bounds__(256) void k_diff5(double* __restrict__ out,
                                               const double* __restrict__ psi, int R, int C, int dir) {
  const long n = (long)R * C;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx / C), c = (int)(idx % C);
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const double w = ((1.0 / 5040.0) * cg_xi(i, j)) * (double)(dir == 0 ? i - 2 : j - 2);
        int rr = r + i - 2, cc = c + j - 2;
        rr = rr < 0 ? 0 : (rr > R - 1 ? R - 1 : rr);
        cc = cc < 0 ? 0 : (cc > C - 1 ? C - 1 : cc);
        s += w * psi[(long)rr * C + cc];
      }
    out[idx] = s;
  }
}

static int check_cg(const char* fn, const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* p) {
  int rc = validate_geom_bc(fn, g, bc);
  if (rc) return rc;
  LBM_REQUIRE(p, "%s: NULL params", fn);
  LBM_REQUIRE(g->ghost == 0 || g->ghost == 3, "%s: ghost=%d (the two-phase step needs 0 or 3 ghost rows)", fn, g->ghost);
  LBM_REQUIRE(p->red.rho_0 > 0 && p->blue.rho_0 > 0 && p->delta > 0, "%s: bad colour parameters", fn);
  LBM_REQUIRE(p->red.alpha < 1.0 && p->blue.alpha < 1.0, "%s: alpha must be < 1", fn);
  LBM_REQUIRE(p->form >= LBM_FORM_DEFAULT && p->form <= LBM_FORM_REASSOCIATED, "%s: form=%d (LBM_FORM_*)", fn, p->form);
  return LBM_OK;
}

static int launch_cg_collide(bool from_post, double* pn_r, double* pn_b, const double* in_r,
                             const double* in_b, const double* rho_r, const double* rho_b,
                             const double* u, const lbm_geom* lg, const lbm_bc* lbc,
                             const lbm_cg_params* prm, double* psi, double* snu, int row_begin,
                             int row_end, hipStream_t st) {
  const Geom g = make_geom(*lg);
  const Bc bc = make_bc(lbc);
  const CgConsts cc = make_cg_consts(*prm);
  const MacroIdx mi = make_macro_idx(g);
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= g.R, "lbm_cg: row range [%d, %d) outside [0, %d)", row_begin, row_end, g.R);
  if (row_begin == row_end) return LBM_OK;
  const int tiles = ((row_end - row_begin + CG_TR - 1) / CG_TR) * ((g.C + CG_TC - 1) / CG_TC);
  const bool fields = psi != nullptr;
  if (from_post) {
    if (fields) LBM_KLAUNCH((k_cg_collide<true, true>), dim3(tiles), dim3(256), 0, st, pn_r, pn_b, in_r, in_b, rho_r, rho_b, u, g, bc, cc, psi, snu, mi, row_begin, row_end);
    else LBM_KLAUNCH((k_cg_collide<true, false>), dim3(tiles), dim3(256), 0, st, pn_r, pn_b, in_r, in_b, rho_r, rho_b, u, g, bc, cc, psi, snu, mi, row_begin, row_end);
  } else {
    if (fields) LBM_KLAUNCH((k_cg_collide<false, true>), dim3(tiles), dim3(256), 0, st, pn_r, pn_b, in_r, in_b, rho_r, rho_b, u, g, bc, cc, psi, snu, mi, row_begin, row_end);
    else LBM_KLAUNCH((k_cg_collide<false, false>), dim3(tiles), dim3(256), 0, st, pn_r, pn_b, in_r, in_b, rho_r, rho_b, u, g, bc, cc, psi, snu, mi, row_begin, row_end);
  }
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

static thread_local int g_last_inner_form = -1;  // lbm_cg_last_inner_form

template <int TR, int TC, int WAVES>
static int launch_cg_fused_t(double* pn_r, double* pn_b, const double* in_r, const double* in_b,
                             const Geom& g, const Bc& bc, const CgFast& cf, double* rho_r,
                             double* rho_b, double* u, double* psi, double* snu, const MacroIdx& mi,
                             int row_begin, int row_end, hipStream_t st, int part = 0, int edge_rows = 0) {
  // part 0: the whole row range (frame beside the inner launch on a helper stream); 1 / 2: ONLY the frame -- widened to the
  // first and last `edge_rows` rows of the range, in whole tiles -- / ONLY the inner rectangle, on `st`: a slab runs the
  // two on two streams and sends its edge rows while the inner launch is still busy (lbm_cg_step_fused_part)
  const int tiles_r = (row_end - row_begin + TR - 1) / TR, tiles_c = (g.C + TC - 1) / TC;
  const int tiles = tiles_r * tiles_c;
  const int xs = tuning("cg_xcd", 2);  // pairs of column-neighbour tiles per XCD: +5 % at 4 waves per SIMD
  // inner rectangle of tiles: the tile's ring rows r_base-2 .. r_base+TR+1 are plain nodes (not the
  // wall rows of the global domain; across a seam the ghost rows count as plain), its ring columns
  // c_base-2 .. c_base+TC+1 lie in [1, C-2] (their gathers do not wrap), and the tile is complete
  const int lo_row = (g.ghost && bc.row_lo == LBM_EDGE_HALO) ? -2 : 1;
  const int hi_row = (g.ghost && bc.row_hi == LBM_EDGE_HALO) ? g.R + 1 : g.R - 2;
  CgTileRect rc{tiles_r, 0, 1, 0};
  for (int i = 0; i < tiles_r; ++i) {
    const int rb = row_begin + i * TR;
    if (rb - 2 >= lo_row && rb + TR + 1 <= hi_row && rb + TR <= row_end) {
      rc.ir0 = rc.ir0 < i ? rc.ir0 : i;
      rc.ir1 = i + 1;
    }
  }
  rc.ic1 = (g.C - 3) / TC;  // last tile column with c_base + TC + 1 <= C - 2
  if (rc.ic1 > tiles_c) rc.ic1 = tiles_c;
  if (part && edge_rows > 0) {
    const int et = (edge_rows + TR - 1) / TR;
    rc.ir0 = rc.ir0 > et ? rc.ir0 : et;
    rc.ir1 = rc.ir1 < tiles_r - et ? rc.ir1 : tiles_r - et;
  }
  bool split = (part || tuning("cg_split", 1) != 0) && rc.ir1 - rc.ir0 >= 1 && rc.ic1 - rc.ic0 >= 1;
  g_last_inner_form = 0;
  // several nodes per thread (k_cg_tile_mn): the inner rectangle is cut into BIG tiles from its top-left corner, what does
  // not fill a big tile joins the frame.  100 + shape; shapes: {rows, columns, threads, waves per SIMD the registers are budgeted for}
  // default (round 4): 16 x 64 tiles, two nodes per thread, the waiting one parked in LDS -- +4 .. +7 % over the 16 x 32 tile
  // kernel on every box measured (profiles/r04_cg_big_sweep.txt); "cg_big" = 0 restores k_cg_fused<16,32,4> on the inner rectangle
  const int big = TR == 16 && TC == 32 ? tuning("cg_big", 2) : 0;
  static const int big_shapes[][4] = {{32, 32, 512, 4}, {16, 64, 512, 4}, {16, 128, 1024, 4}, {32, 64, 1024, 4},
                                      {32, 64, 512, 2}, {8, 64, 512, 4}, {16, 64, 1024, 4}, {16, 128, 512, 2},
                                      {16, 32, 512, 4}};
  int n_btr = 0, n_btc = 0;
#ifndef LBM_EXPERIMENTS
  const bool walk_tile = false;
  const int shape = big > 0 ? 2 : 0;  // the default build ships shape 2 only (any non-zero "cg_big" selects it)
#else
  // 10: the WALKING tile (k_cg_walk_tile): the 16 x 64 tile advancing through chunks of "cg_walk_rows" rows
  const bool walk_tile = big == 10;
  const int shape = walk_tile ? 2 : (big >= 1 && big <= (int)(sizeof big_shapes / sizeof big_shapes[0]) ? big : 0);
#endif
  if (split && shape) {
    const int* s = big_shapes[shape - 1];
    const int rows16 = ((rc.ir1 - rc.ir0) * 16 / s[0]) * s[0] / 16 * 16;  // rows the big tiles cover: whole big tiles AND whole 16-row units
    n_btr = rows16 / s[0];
    n_btc = (rc.ic1 - rc.ic0) * 32 / s[1];
    if (n_btr >= 1 && n_btc >= 1 && n_btr * s[0] == rows16 && (n_btc * s[1]) % 32 == 0) {
      rc.ir1 = rc.ir0 + rows16 / 16;
      rc.ic1 = rc.ic0 + n_btc * s[1] / 32;
    } else n_btr = n_btc = 0;
  }
  if (!split) {
    if (part == 2) return LBM_OK;  // no inner rectangle: the frame part runs every tile
    if (psi) LBM_KLAUNCH((k_cg_fused<TR, TC, WAVES, true>), dim3(tiles), dim3(TR * TC), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, xs);
    else LBM_KLAUNCH((k_cg_fused<TR, TC, WAVES, false>), dim3(tiles), dim3(TR * TC), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, xs);
    LBM_CHECK_LAUNCH();
    return LBM_OK;
  }
  const int inner = (rc.ir1 - rc.ir0) * (rc.ic1 - rc.ic0), frame = tiles - inner;
#ifdef LBM_EXPERIMENTS
  if (!part && frame > 0 && !tuning("cg_strip2", 0) && tuning("cg_merge", 0)) {  // frame + inner tiles in one dispatch (opt-in: measured level with the two-launch form, 15.24 k either way)
    if (psi) LBM_KLAUNCH((k_cg_fused_merged<TR, TC, WAVES, true>), dim3(frame + inner), dim3(TR * TC), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, xs, rc, frame);
    else LBM_KLAUNCH((k_cg_fused_merged<TR, TC, WAVES, false>), dim3(frame + inner), dim3(TR * TC), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, xs, rc, frame);
    LBM_CHECK_LAUNCH();
    return LBM_OK;
  }
#endif
  // the frame (3-4 % of the tiles, latency-bound: 63 us on its own) goes FIRST and on the helper stream, so
  // that it runs beside the inner launch instead of behind it (fork / join through two events, launch.hpp)
  SwSideStream* sd = !part && frame > 0 && tuning("cg_frame_beside", 1) ? sw_side_stream() : nullptr;
  hipStream_t fs = st;
  if (sd && hipEventRecord(sd->fork, st) == hipSuccess && hipStreamWaitEvent(sd->st, sd->fork, 0) == hipSuccess) fs = sd->st;
  if (frame > 0 && part != 2) {
    if (psi) LBM_KLAUNCH((k_cg_fused<TR, TC, WAVES, true, 2>), dim3(frame), dim3(TR * TC), 0, fs, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, 0, rc);
    else LBM_KLAUNCH((k_cg_fused<TR, TC, WAVES, false, 2>), dim3(frame), dim3(TR * TC), 0, fs, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, 0, rc);
    LBM_CHECK_LAUNCH();
  }
  if (part == 1) return LBM_OK;
  const int sw4 = (n_btr || (g.P != g.C && tuning("cg_strip2", 0) < 41)) ? 0 : tuning("cg_strip2", 0);  // (the strip forms of the experiments build know dense rows only)
  if (n_btr) {  // k_cg_tile_mn: big tiles, several nodes per thread
    const int ra = row_begin + rc.ir0 * TR, ca = rc.ic0 * TC, nt = n_btr * n_btc;
    // patches of 4 x 2 tiles per XCD (100 PR + PC): ring rows and ring columns inside a patch are hits of one L2.  Larger
    // patches read less and less (4.09 GB per step with pairs of column neighbours, 3.99 with 4 x 2, 3.90 with 8 x 2, 3.73 with
    // 8 x 4) but the rate the memory system delivers falls with them on some boxes: 4 x 2 is the order that is never slower
    // than the pairs (+3.6 %, +0.4 %, +0.2 % on three boxes; 8 x 2: +4.7 %, -0.3 %, -2.9 %), profiles/r04_cg_order_pmc.txt
    const int bx = tuning("cg_big_xcd", 402);
    g_last_inner_form = 100 + (walk_tile ? 10 : shape);
#ifdef LBM_EXPERIMENTS
    if (walk_tile) {
      int rpc = tuning("cg_walk_rows", 128) / 16 * 16;
      if (rpc < 16) rpc = 16;
      const int rows_total = n_btr * 16, chunks = (rows_total + rpc - 1) / rpc;
      const int wx = tuning("cg_walk_tile_xcd", 2);
      if (psi) LBM_KLAUNCH((k_cg_walk_tile<true>), dim3(chunks * n_btc), dim3(512), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, ca, n_btc, rows_total, rpc, wx);
      else LBM_KLAUNCH((k_cg_walk_tile<false>), dim3(chunks * n_btc), dim3(512), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, ca, n_btc, rows_total, rpc, wx);
    } else
#endif
    {
#define LBM_CG_BIG(BR, BC, BT, BM, BP)                                                                               \
    if (psi) LBM_KLAUNCH((k_cg_tile_mn<BR, BC, BT, BM, BP, true>), dim3(nt), dim3(BT), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, ca, n_btc, bx); \
    else LBM_KLAUNCH((k_cg_tile_mn<BR, BC, BT, BM, BP, false>), dim3(nt), dim3(BT), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, ca, n_btc, bx);
    switch (shape) {
#ifdef LBM_EXPERIMENTS  // the shapes of the round-4 sweep that lost to 16 x 64 (profiles/r04_cg_big_sweep.txt)
      case 1: LBM_CG_BIG(32, 32, 512, 4, true) break;    // 2 nodes per thread, the second parked in LDS: 2 workgroups per CU
      case 3: LBM_CG_BIG(16, 128, 1024, 4, true) break;  // 1024 threads: one workgroup per CU
      case 4: LBM_CG_BIG(32, 64, 1024, 4, true) break;
      case 5: LBM_CG_BIG(32, 64, 512, 2, false) break;   // 4 nodes per thread, 2 waves per SIMD: one workgroup per CU
      case 6: LBM_CG_BIG(8, 64, 512, 4, false) break;    // one node per thread in the wide shape (what the width alone is worth)
      case 7: LBM_CG_BIG(16, 64, 1024, 4, false) break;
      case 8: LBM_CG_BIG(16, 128, 512, 2, false) break;
      case 9: LBM_CG_BIG(16, 32, 512, 4, false) break;   // the default tile's shape, one node per thread: what the patch orders alone are worth
#endif
      default: LBM_CG_BIG(16, 64, 512, 4, true) break;   // shape 2: 16 x 64, 2 nodes per thread, the second parked in LDS
    }
#undef LBM_CG_BIG
    }
  } else
  // 41 .. 47: k_cg_walk -- a workgroup of TR x WC waves walking down a strip of 64 WC - 4 columns, TR rows a step
  if (sw4 >= 41 && sw4 <= 47 && rc.ic0 * TC >= 4 && 9.0 * (double)g.plane * 8.0 < 4.0e9) {  // 32-bit plane offsets
    const int ra = row_begin + rc.ir0 * TR, rb = row_begin + rc.ir1 * TR, ca = rc.ic0 * TC, cb = rc.ic1 * TC;
    static const int shapes[7][2] = {{4, 1}, {6, 1}, {2, 2}, {3, 2}, {2, 3}, {2, 1}, {3, 1}};
    const int wtr = shapes[sw4 - 41][0], wc = shapes[sw4 - 41][1], outc = 64 * wc - 4;
    const int strips = (cb - ca + outc - 1) / outc;
    int rpc = tuning("cg_rows2", 0);
    // chunks of 128 rows: several rounds of workgroups, dispatched as slots free up.  One round of long chunks (every
    // workgroup resident at once) is 6-8 % slower on six boxes of nine: a static partition ends with its slowest workgroup
    if (rpc <= 0) rpc = 128;
    rpc = (rpc + wtr - 1) / wtr * wtr;
    if (rpc > rb - ra) rpc = rb - ra;
    const int chunks = (rb - ra + rpc - 1) / rpc, nb = strips * chunks, grid = (nb + 7) / 8 * 8;
    const int xo = tuning("cg_walk_xcd", 1);
    g_last_inner_form = sw4;
    if (sw4 == 41 && tuning("cg_walk_pf", 1) == 0) {  // the 4 x 1 block without prefetch, 4 waves per SIMD
      if (psi) LBM_KLAUNCH((k_cg_walk<4, 1, true, false>), dim3(grid), dim3(256), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, nb, xo);
      else LBM_KLAUNCH((k_cg_walk<4, 1, false, false>), dim3(grid), dim3(256), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, nb, xo);
    } else
#define LBM_CG_WALK(WTR, WWC)                                                                                          \
    if (psi) LBM_KLAUNCH((k_cg_walk<WTR, WWC, true>), dim3(grid), dim3(WTR * WWC * 64), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, nb, xo); \
    else LBM_KLAUNCH((k_cg_walk<WTR, WWC, false>), dim3(grid), dim3(WTR * WWC * 64), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, nb, xo);
    switch (sw4) {
      case 41: LBM_CG_WALK(4, 1) break;
      case 42: LBM_CG_WALK(6, 1) break;
      case 43: LBM_CG_WALK(2, 2) break;
      case 44: LBM_CG_WALK(3, 2) break;
      case 45: LBM_CG_WALK(2, 3) break;
      case 46: LBM_CG_WALK(2, 1) break;
      default: LBM_CG_WALK(3, 1) break;
    }
#undef LBM_CG_WALK
  } else
#ifdef LBM_EXPERIMENTS
  // 31 / 32: k_cg_strip5 -- private windows, 2 / 4 adjacent strips per workgroup, a barrier every "cg_sync" rows (0: none)
  if ((sw4 == 31 || sw4 == 32) && rc.ic0 * TC >= 8) {
    const int ra = row_begin + rc.ir0 * TR, rb = row_begin + rc.ir1 * TR, ca = rc.ic0 * TC, cb = rc.ic1 * TC;
    const int Wv = sw4 == 31 ? 2 : 4;
    const int strips = (cb - ca + CG_SW2 - 1) / CG_SW2, groups = (strips + Wv - 1) / Wv;
    const void* kfn = sw4 == 31 ? (psi ? (const void*)k_cg_strip5<2, true> : (const void*)k_cg_strip5<2, false>)
                                : (psi ? (const void*)k_cg_strip5<4, true> : (const void*)k_cg_strip5<4, false>);
    int rpc = tuning("cg_rows2", 0);
    if (rpc <= 0) {
      const long slots = sw_wave_slots(kfn, 64 * Wv);
      rpc = slots > 0 ? sw_pick_rows(rb - ra, groups * Wv, 3, slots) : 64;
    }
    if (rpc > rb - ra) rpc = rb - ra;
    const int chunks = (rb - ra + rpc - 1) / rpc, sync = tuning("cg_sync", 8);
    g_last_inner_form = sw4;
#define LBM_CG_S5(WV)                                                                                              \
    if (psi) LBM_KLAUNCH((k_cg_strip5<WV, true>), dim3(groups * chunks), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, groups, sync); \
    else LBM_KLAUNCH((k_cg_strip5<WV, false>), dim3(groups * chunks), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, groups, sync);
    if (sw4 == 31) { LBM_CG_S5(2) } else { LBM_CG_S5(4) }
#undef LBM_CG_S5
  } else
  // 21 / 22: the lockstep block kernel (k_cg_strip4, 4 / 8 waves per block); needs line-aligned rows and planes
  if ((sw4 == 21 || sw4 == 22) && g.C % 16 == 0 && g.plane % 16 == 0 && rc.ic0 * TC >= 2 * CG_S4_EDGE) {
    const int ra = row_begin + rc.ir0 * TR, rb = row_begin + rc.ir1 * TR, ca = rc.ic0 * TC, cb = rc.ic1 * TC;
    const int Wv = sw4 == 21 ? 4 : 8, S = 64 * Wv - 2 * CG_S4_EDGE;
    const int win0 = (ca - CG_S4_EDGE) / 16 * 16;  // line-aligned window start; lane CG_S4_EDGE = first possible output
    const int bstrips = (cb - (win0 + CG_S4_EDGE) + S - 1) / S;
    const void* kfn = sw4 == 21 ? (psi ? (const void*)k_cg_strip4<4, true> : (const void*)k_cg_strip4<4, false>)
                                : (psi ? (const void*)k_cg_strip4<8, true> : (const void*)k_cg_strip4<8, false>);
    int rpc = tuning("cg_rows2", 0);
    if (rpc <= 0) {
      const long slots = sw_wave_slots(kfn, 64 * Wv);
      rpc = slots > 0 ? sw_pick_rows(rb - ra, bstrips * Wv, 3, slots) : 64;
    }
    if (rpc > rb - ra) rpc = rb - ra;
    const int chunks = (rb - ra + rpc - 1) / rpc;
    g_last_inner_form = sw4;
#define LBM_CG_S4(WV)                                                                                              \
    if (psi) LBM_KLAUNCH((k_cg_strip4<WV, true>), dim3(bstrips * chunks), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, bstrips, win0); \
    else LBM_KLAUNCH((k_cg_strip4<WV, false>), dim3(bstrips * chunks), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, bstrips, win0);
    if (sw4 == 21) { LBM_CG_S4(4) } else { LBM_CG_S4(8) }
#undef LBM_CG_S4
  } else
  if (const int sw = (sw4 >= 21) ? 0 : sw4) {  // the inner rectangle through a register-ring strip kernel
    // 1, 2, 4: k_cg_strip2 (one wave per SIMD) with that many waves per workgroup; 11, 12: k_cg_strip3 (colour sums
    // of the ring rows in LDS, two waves per SIMD) with 1 / 2 waves per workgroup
    const int ra = row_begin + rc.ir0 * TR, rb = row_begin + rc.ir1 * TR, ca = rc.ic0 * TC, cb = rc.ic1 * TC;
    const int strips = (cb - ca + CG_SW2 - 1) / CG_SW2;
    const void* kfn = sw == 2 ? (psi ? (const void*)k_cg_strip2<2, true> : (const void*)k_cg_strip2<2, false>)
                    : sw == 1 ? (psi ? (const void*)k_cg_strip2<1, true> : (const void*)k_cg_strip2<1, false>)
                    : sw == 11 ? (psi ? (const void*)k_cg_strip3<1, true> : (const void*)k_cg_strip3<1, false>)
                    : sw == 12 ? (psi ? (const void*)k_cg_strip3<2, true> : (const void*)k_cg_strip3<2, false>)
                               : (psi ? (const void*)k_cg_strip2<4, true> : (const void*)k_cg_strip2<4, false>);
    const int wv = sw == 2 || sw == 12 ? 2 : (sw == 1 || sw == 11 ? 1 : 4);
    // rows per wave: at 16.8 M nodes the launch is only 1-3 rounds of resident waves deep -- a chunk height
    // that leaves the last round nearly empty costs up to a whole round; fit it to the resident wave slots
    int rpc = tuning("cg_rows2", 0);
    if (rpc <= 0) {
      const long slots = sw_wave_slots(kfn, 64 * wv);
      rpc = slots > 0 ? sw_pick_rows(rb - ra, strips, 3, slots) : 64;
    }
    if (rpc > rb - ra) rpc = rb - ra;
    const int chunks = (rb - ra + rpc - 1) / rpc, n_waves = strips * chunks;
    g_last_inner_form = sw;
#define LBM_CG_S2(KERNEL, WV)                                                                                      \
    if (psi) LBM_KLAUNCH((KERNEL<WV, true>), dim3((n_waves + WV - 1) / WV), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, n_waves); \
    else LBM_KLAUNCH((KERNEL<WV, false>), dim3((n_waves + WV - 1) / WV), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, n_waves);
    const int xo = tuning("cg_strip_xcd", 0);  // strip3: XCD k takes the k-th contiguous eighth of the strip sequence (measured: no effect)
#define LBM_CG_S3(WV)                                                                                              \
    {                                                                                                              \
      const int nblk = (n_waves + WV - 1) / WV, grid3 = xo ? ((nblk + 7) / 8) * 8 : nblk;                          \
      if (psi) LBM_KLAUNCH((k_cg_strip3<WV, true>), dim3(grid3), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, n_waves, xo); \
      else LBM_KLAUNCH((k_cg_strip3<WV, false>), dim3(grid3), dim3(64 * WV), 0, st, pn_r, pn_b, in_r, in_b, g, cf, rho_r, rho_b, u, psi, snu, mi, ra, rb, ca, cb, rpc, strips, n_waves, xo); \
    }
    if (sw == 2) { LBM_CG_S2(k_cg_strip2, 2) } else if (sw == 1) { LBM_CG_S2(k_cg_strip2, 1) }
    else if (sw == 11) LBM_CG_S3(1) else if (sw == 12) LBM_CG_S3(2)
    else { LBM_CG_S2(k_cg_strip2, 4) }
#undef LBM_CG_S2
#undef LBM_CG_S3
  } else
#endif  // LBM_EXPERIMENTS (strip kernels)
  if (psi) LBM_KLAUNCH((k_cg_fused<TR, TC, WAVES, true, 1>), dim3(inner), dim3(TR * TC), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, xs, rc);
  else LBM_KLAUNCH((k_cg_fused<TR, TC, WAVES, false, 1>), dim3(inner), dim3(TR * TC), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, xs, rc);
  LBM_CHECK_LAUNCH();
  if (fs != st) {
    LBM_CHECK_HIP(hipEventRecord(sd->join, fs));
    LBM_CHECK_HIP(hipStreamWaitEvent(st, sd->join, 0));
  }
  return LBM_OK;
}

#ifdef LBM_EXPERIMENTS
template <int WAVES>
static int launch_cg_strip_t(double* pn_r, double* pn_b, const double* in_r, const double* in_b,
                             const Geom& g, const Bc& bc, const CgFast& cf, double* rho_r,
                             double* rho_b, double* u, double* psi, double* snu, const MacroIdx& mi,
                             int row_begin, int row_end, hipStream_t st) {
  int rpc = tuning("cg_rows", 16);
  const int nrows = row_end - row_begin;
  if (rpc > nrows) rpc = nrows;
  const int strips = (g.C + CG_SW - 1) / CG_SW, chunks = (nrows + rpc - 1) / rpc;
  const int n_waves = strips * chunks;
  const dim3 grid((n_waves + WAVES - 1) / WAVES);
  if (psi) LBM_KLAUNCH((k_cg_strip<WAVES, true>), grid, dim3(64 * WAVES), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, rpc, strips, n_waves);
  else LBM_KLAUNCH((k_cg_strip<WAVES, false>), grid, dim3(64 * WAVES), 0, st, pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, rpc, strips, n_waves);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

#endif  // LBM_EXPERIMENTS

}  // namespace lbm

using namespace lbm;

extern "C" {

int lbm_diff5(double* out, const double* psi, int R, int C, int dir, lbm_stream_t s) {
  LBM_REQUIRE(out && psi && out != psi && R > 0 && C > 0 && (dir == 0 || dir == 1), "lbm_diff5: bad argument");
  const long n = (long)R * C;
  LBM_KLAUNCH(k_diff5, dim3(capped_grid((n + 255) / 256)), dim3(256), 0, as_stream(s), out, psi, R, C, dir);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

void lbm_cg_default_bc(lbm_bc* bc) {
  // apply_boundary_conditions, mrtcg_rayleigh_taylor.cpp:495-533
  if (!bc) return;
  *bc = lbm_bc{LBM_EDGE_BOUNCE_BACK, LBM_EDGE_BOUNCE_BACK, LBM_EDGE_WRAP_NOSHIFT,
               LBM_EDGE_WRAP_NOSHIFT, 0, 1.0, 1.0, 0.0, 0.0};
}

int lbm_cg_equilibrium(double* f, const double* rho_k, const double* u, const lbm_cg_colour* k,
                       int R, int C, long long plane_stride, lbm_stream_t s) {
  LBM_REQUIRE(f && rho_k && u && k && R > 0 && C > 0, "lbm_cg_equilibrium: bad argument");
  lbm_cg_params p{*k, *k, 0.0, 0.0, 0.0, 0, 0.1};
  const CgConsts cc = make_cg_consts(p);
  const long n = (long)R * C;
  LBM_REQUIRE(plane_stride == 0 || plane_stride >= n, "lbm_cg_equilibrium: plane_stride too small");
  LBM_KLAUNCH(k_cg_equilibrium, dim3(capped_grid((n + 255) / 256)), dim3(256), 0, as_stream(s), f,
              rho_k, u, n, plane_stride ? (long)plane_stride : n, cc.k[0]);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_cg_collide(double* p_r, double* p_b, const double* f_r, const double* f_b,
                   const double* rho_r, const double* rho_b, const double* u, const lbm_geom* g,
                   const lbm_bc* bc, const lbm_cg_params* prm, double* psi, double* snu,
                   lbm_stream_t s) {
  int rc = check_cg("lbm_cg_collide", g, bc, prm);
  if (rc) return rc;
  LBM_REQUIRE(p_r && p_b && f_r && f_b && rho_r && rho_b && u, "lbm_cg_collide: NULL pointer");
  LBM_REQUIRE((psi == nullptr) == (snu == nullptr), "lbm_cg_collide: psi and s_nu go together");
  return launch_cg_collide(false, p_r, p_b, f_r, f_b, rho_r, rho_b, u, g, bc, prm, psi, snu, 0, g->R, as_stream(s));
}

int lbm_cg_stream_moments(double* rho_r, double* rho_b, double* u, const double* p_r,
                          const double* p_b, const lbm_geom* g, const lbm_bc* bc,
                          const lbm_cg_params* prm, lbm_stream_t s) {
  int rc = check_cg("lbm_cg_stream_moments", g, bc, prm);
  if (rc) return rc;
  LBM_REQUIRE(rho_r && rho_b && u && p_r && p_b, "lbm_cg_stream_moments: NULL pointer");
  const Geom gg = make_geom(*g);
  const Bc bb = make_bc(bc);
  const int lo = cg_row_lo(gg, bb), hi = cg_row_hi(gg, bb) + 1;  // incl. the macro ghost rows of a slab
  const long n = (long)(hi - lo) * gg.C;
  LBM_KLAUNCH(k_cg_stream_moments, dim3(capped_grid((n + 255) / 256, 8192)), dim3(256), 0, as_stream(s),
              rho_r, rho_b, u, p_r, p_b, gg, bb, prm->gravity_r, prm->gravity_c, make_macro_idx(gg), lo, hi);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_cg_stream_collide(double* pn_r, double* pn_b, const double* p_r, const double* p_b,
                          const double* rho_r, const double* rho_b, const double* u,
                          const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* prm,
                          int row_begin, int row_end, double* psi, double* snu, lbm_stream_t s) {
  int rc = check_cg("lbm_cg_stream_collide", g, bc, prm);
  if (rc) return rc;
  LBM_REQUIRE(pn_r && pn_b && p_r && p_b && rho_r && rho_b && u, "lbm_cg_stream_collide: NULL pointer");
  LBM_REQUIRE(pn_r != p_r && pn_b != p_b, "lbm_cg_stream_collide: aliased lattices");
  LBM_REQUIRE((psi == nullptr) == (snu == nullptr), "lbm_cg_stream_collide: psi and s_nu go together");
  return launch_cg_collide(true, pn_r, pn_b, p_r, p_b, rho_r, rho_b, u, g, bc, prm, psi, snu, row_begin, row_end, as_stream(s));
}

static int cg_step_fused(double* pn_r, double* pn_b, const double* p_r, const double* p_b,
                         const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* prm, int row_begin,
                         int row_end, double* rho_r, double* rho_b, double* u, double* psi, double* snu,
                         lbm_stream_t s, int part, int edge_rows) {
  int rc = check_cg("lbm_cg_step_fused", g, bc, prm);
  if (rc) return rc;
  LBM_REQUIRE(pn_r && pn_b && p_r && p_b, "lbm_cg_step_fused: NULL lattice");
  LBM_REQUIRE(pn_r != p_r && pn_b != p_b, "lbm_cg_step_fused: aliased lattices");
  const bool any = rho_r || rho_b || u || psi || snu, all = rho_r && rho_b && u && psi && snu;
  LBM_REQUIRE(any == all, "lbm_cg_step_fused: the five field outputs go together (all or none)");
  const Geom gg = make_geom(*g);
  const Bc bb = make_bc(bc);
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= gg.R, "lbm_cg_step_fused: row range [%d, %d) outside [0, %d)", row_begin, row_end, gg.R);
  if (row_begin == row_end) return LBM_OK;
  const CgFast cf = make_cg_fast(make_cg_consts(*prm));
  const MacroIdx mi = make_macro_idx(gg);
  hipStream_t st = as_stream(s);
#ifdef LBM_EXPERIMENTS
  switch (gg.P != gg.C ? 0 : tuning("cg_strip", 0)) {  // column-strip sliding window (opt-in: slower as written, cg_fused.hpp), waves per workgroup
    case 0: break;
    case 2: return launch_cg_strip_t<2>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    case 4: return launch_cg_strip_t<4>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    default: return launch_cg_strip_t<1>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
  }
#endif
  switch (tuning("cg_tile", 4)) {  // default: 16x32 tiles budgeted for 4 waves per SIMD (128 VGPRs)
    case 0: return launch_cg_fused_t<8, 32, 1>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    case 2: return launch_cg_fused_t<8, 64, 1>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    case 3: return launch_cg_fused_t<16, 32, 3>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    case 1: return launch_cg_fused_t<16, 32, 1>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    case 5: return launch_cg_fused_t<32, 32, 4>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    case 6: return launch_cg_fused_t<16, 64, 4>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st);
    default: return launch_cg_fused_t<16, 32, 4>(pn_r, pn_b, p_r, p_b, gg, bb, cf, rho_r, rho_b, u, psi, snu, mi, row_begin, row_end, st, part, edge_rows);
  }
}

int lbm_cg_step_fused(double* pn_r, double* pn_b, const double* p_r, const double* p_b,
                      const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* prm, int row_begin,
                      int row_end, double* rho_r, double* rho_b, double* u, double* psi, double* snu,
                      lbm_stream_t s) {
  return cg_step_fused(pn_r, pn_b, p_r, p_b, g, bc, prm, row_begin, row_end, rho_r, rho_b, u, psi, snu, s, 0, 0);
}

int lbm_cg_step_fused_part(double* pn_r, double* pn_b, const double* p_r, const double* p_b,
                           const lbm_geom* g, const lbm_bc* bc, const lbm_cg_params* prm, int part,
                           int edge_rows, double* rho_r, double* rho_b, double* u, double* psi, double* snu,
                           lbm_stream_t s) {
  LBM_REQUIRE(part == LBM_CG_PART_FRAME || part == LBM_CG_PART_INNER, "lbm_cg_step_fused_part: part=%d (LBM_CG_PART_FRAME / _INNER)", part);
  LBM_REQUIRE(g && edge_rows >= 0 && 2 * edge_rows <= g->R, "lbm_cg_step_fused_part: edge_rows=%d", edge_rows);
  LBM_REQUIRE(tuning("cg_tile", 4) == 4, "lbm_cg_step_fused_part: needs the default tile kernel (\"cg_tile\" = 4)");
  return cg_step_fused(pn_r, pn_b, p_r, p_b, g, bc, prm, 0, g->R, rho_r, rho_b, u, psi, snu, s, part, edge_rows);
}

}  // extern "C"

// ---- solver context for the two-phase driver loop -----------------------------------------------
struct lbm_cg_solver {
  lbm_geom g;
  lbm_bc bc;
  lbm_cg_params prm;
  hipStream_t st;
  double* lat[2][2];  // [buffer][colour]
  double *rho_r, *rho_b, *u, *psi, *snu, *stage;
  int cur;
  bool post;
  long steps;
  // two steps per pass (cg_solver_step2): the frame of the lattice advances two single steps on two small lattices --
  // row band: rows [0, HB) and [R - HB, R) x all columns; column band: all rows x columns [0, WB) and [C - WB, C)
  double* rband[2][2];  // [buffer][colour]
  double* cband[2][2];
  lbm_geom rbg, cbg;
  hipStream_t band_st;
  hipEvent_t ev_band_fork, ev_band_join;
  long pair_launches;
};
#ifdef LBM_EXPERIMENTS
static constexpr int kCgX2RowBand = 16, kCgX2ColBand = 32;  // the frame the two-step kernel leaves out (tile-aligned)
static constexpr int kCgX2HB = 32, kCgX2WB = 48;            // rows / columns per side the band lattices hold (valid after 2 steps: HB - 6, WB - 6)
#endif

extern "C" {

int lbm_cg_solver_create(lbm_cg_solver** out, const lbm_geom* g, const lbm_bc* bc,
                         const lbm_cg_params* prm, lbm_stream_t s) {
  LBM_REQUIRE(out, "lbm_cg_solver_create: NULL out pointer");
  lbm_bc dflt;
  lbm_cg_default_bc(&dflt);
  int rc = check_cg("lbm_cg_solver_create", g, bc ? bc : &dflt, prm);
  if (rc) return rc;
  lbm_cg_solver* sv = new (std::nothrow) lbm_cg_solver();
  LBM_REQUIRE(sv, "lbm_cg_solver_create: out of host memory");
  sv->g = *g;
  sv->bc = bc ? *bc : dflt;
  sv->prm = *prm;
  sv->st = as_stream(s);
  sv->cur = 0;
  sv->post = false;
  sv->steps = 0;
  sv->pair_launches = 0;
  sv->band_st = nullptr;
  sv->ev_band_fork = sv->ev_band_join = nullptr;
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 2; ++k) sv->rband[b][k] = sv->cband[b][k] = nullptr;
  const size_t n = (size_t)g->R * g->C;
  // the solver's own lattices: rows padded off a power-of-two stride (lbm_default_row_pitch), planes likewise.  Everything
  // that reads or writes them takes &sv->g; the macroscopic fields and the host-side AoS arrays stay dense.
  const int pitch = lbm_default_row_pitch(g->C);
  sv->g.row_pitch = pitch > g->C ? pitch : 0;
  sv->g.plane_stride = (long long)g->R * pitch + lbm_default_plane_pad(g->R, pitch);
  const size_t lat_bytes = (size_t)sv->g.plane_stride * 9 * sizeof(double);
  double** all[] = {&sv->lat[0][0], &sv->lat[0][1], &sv->lat[1][0], &sv->lat[1][1], &sv->rho_r,
                    &sv->rho_b, &sv->u, &sv->psi, &sv->snu, &sv->stage};
  const size_t bytes[] = {lat_bytes, lat_bytes, lat_bytes, lat_bytes, n * 8, n * 8, n * 16, n * 8, n * 8, n * 72};
  for (auto p : all) *p = nullptr;
  for (int i = 0; i < 10; ++i) {
    hipError_t e = hipMalloc(all[i], bytes[i]);
    if (e != hipSuccess) {
      set_error("lbm_cg_solver_create: hipMalloc failed: %s", hipGetErrorString(e));
      lbm_cg_solver_destroy(sv);
      return LBM_ERR_HIP;
    }
  }
  LBM_CHECK_HIP(hipMemsetAsync(sv->psi, 0, n * 8, sv->st));  // phase_field / s_nu start as zeros (:380,:382)
  LBM_CHECK_HIP(hipMemsetAsync(sv->snu, 0, n * 8, sv->st));
  *out = sv;
  return LBM_OK;
}

int lbm_cg_solver_destroy(lbm_cg_solver* sv) {
  if (!sv) return LBM_OK;
  for (double* p : {sv->lat[0][0], sv->lat[0][1], sv->lat[1][0], sv->lat[1][1], sv->rho_r, sv->rho_b,
                    sv->u, sv->psi, sv->snu, sv->stage, sv->rband[0][0], sv->rband[0][1], sv->rband[1][0], sv->rband[1][1],
                    sv->cband[0][0], sv->cband[0][1], sv->cband[1][0], sv->cband[1][1]})
    if (p) (void)hipFree(p);
  if (sv->band_st) {
    (void)hipStreamSynchronize(sv->band_st);
    (void)hipStreamDestroy(sv->band_st);
  }
  if (sv->ev_band_fork) (void)hipEventDestroy(sv->ev_band_fork);
  if (sv->ev_band_join) (void)hipEventDestroy(sv->ev_band_join);
  delete sv;
  return LBM_OK;
}

// host AoS state in the reference's shapes: f_r, f_b [R][C][9] (adv_f), rho_r, rho_b [R][C], u [R][C][2]
int lbm_cg_solver_set_state(lbm_cg_solver* sv, const double* f_r, const double* f_b,
                            const double* rho_r, const double* rho_b, const double* u) {
  LBM_REQUIRE(sv && f_r && f_b && rho_r && rho_b && u, "lbm_cg_solver_set_state: NULL argument");
  const int R = sv->g.R, C = sv->g.C;
  const size_t n = (size_t)R * C;
  const double* fs[2] = {f_r, f_b};
  for (int k = 0; k < 2; ++k) {
    LBM_CHECK_HIP(hipMemcpyAsync(sv->stage, fs[k], n * 72, hipMemcpyHostToDevice, sv->st));
    int rc = lbm_aos_to_soa_pitched(sv->lat[sv->cur][k], sv->stage, R, C, 9, sv->g.plane_stride, sv->g.row_pitch, sv->st);
    if (rc) return rc;
  }
  LBM_CHECK_HIP(hipMemcpyAsync(sv->rho_r, rho_r, n * 8, hipMemcpyHostToDevice, sv->st));
  LBM_CHECK_HIP(hipMemcpyAsync(sv->rho_b, rho_b, n * 8, hipMemcpyHostToDevice, sv->st));
  LBM_CHECK_HIP(hipMemcpyAsync(sv->stage, u, n * 16, hipMemcpyHostToDevice, sv->st));
  int rc = lbm_aos_to_soa(sv->u, sv->stage, R, C, 2, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  sv->post = false;
  return LBM_OK;
}

#ifdef LBM_EXPERIMENTS
// ---- two steps per pass ---------------------------------------------------------------------------------------------------
// k_cg_two_step (cg_fused.hpp) on the nodes whose two-step dependency cone holds plain nodes only -- rows [16, R - 16) x
// columns [32, C - 32) --, and the frame around them through TWO single steps of the ordinary one-launch kernel on two small
// lattices: a row band (rows [0, 32) then [R - 32, R): its first / last rows ARE the walls, the artificial seam in its middle
// spoils 3 rows per side and step, rows [0, 16) and [48, 64) are copied back) and a column band (columns [0, 48) then
// [C - 48, C): its first / last columns are the pair the driver's same-row column copy couples, :517-523).  The band chain
// runs on a helper stream beside the big launch.  Same kernels per node as two single steps: same bits.
static bool cg_two_step_applies(const lbm_cg_solver* sv) {
  lbm_bc d;
  lbm_cg_default_bc(&d);
  const lbm_bc& b = sv->bc;
  const bool walls = b.row_lo == d.row_lo && b.row_hi == d.row_hi && b.col_lo == d.col_lo && b.col_hi == d.col_hi && !b.pressure_rows;
  const long long plane = sv->g.plane_stride;
  return walls && sv->g.row_pitch == 0 && sv->g.ghost == 0 && sv->g.C % 16 == 0 && plane % 16 == 0 && sv->g.R >= 2 * kCgX2HB + 64 && sv->g.C >= 2 * kCgX2WB + 256;
}

static int cg_two_step_prepare(lbm_cg_solver* sv) {
  if (sv->band_st) return LBM_OK;
  const int R = sv->g.R, C = sv->g.C;
  sv->rbg = lbm_geom{2 * kCgX2HB, C, 0, (long long)2 * kCgX2HB * C + 1088};
  sv->cbg = lbm_geom{R, 2 * kCgX2WB, 0, (long long)R * 2 * kCgX2WB + 1088};
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 2; ++k) {
      LBM_CHECK_HIP(hipMalloc(&sv->rband[b][k], (size_t)sv->rbg.plane_stride * 9 * sizeof(double)));
      LBM_CHECK_HIP(hipMalloc(&sv->cband[b][k], (size_t)sv->cbg.plane_stride * 9 * sizeof(double)));
    }
  LBM_CHECK_HIP(hipEventCreateWithFlags(&sv->ev_band_fork, hipEventDisableTiming));
  LBM_CHECK_HIP(hipEventCreateWithFlags(&sv->ev_band_join, hipEventDisableTiming));
  LBM_CHECK_HIP(hipStreamCreateWithFlags(&sv->band_st, hipStreamNonBlocking));
  return LBM_OK;
}

static int cg_solver_step2(lbm_cg_solver* sv) {
  int rc = cg_two_step_prepare(sv);
  if (rc) return rc;
  const int R = sv->g.R, C = sv->g.C, HB = kCgX2HB, WB = kCgX2WB;
  double** src = sv->lat[sv->cur];
  double** dst = sv->lat[sv->cur ^ 1];
  hipStream_t st = sv->st, bs = sv->band_st;
  LBM_CHECK_HIP(hipEventRecord(sv->ev_band_fork, st));
  LBM_CHECK_HIP(hipStreamWaitEvent(bs, sv->ev_band_fork, 0));
  // ---- the frame: copy in, two single steps, on the helper stream ----
  for (int k = 0; k < 2 && !rc; ++k) {
    rc = box_copy(sv->rband[0][k], sv->rbg, 0, 0, src[k], sv->g, 0, 0, HB, C, bs);
    if (!rc) rc = box_copy(sv->rband[0][k], sv->rbg, HB, 0, src[k], sv->g, R - HB, 0, HB, C, bs);
    if (!rc) rc = box_copy(sv->cband[0][k], sv->cbg, 0, 0, src[k], sv->g, 0, 0, R, WB, bs);
    if (!rc) rc = box_copy(sv->cband[0][k], sv->cbg, 0, WB, src[k], sv->g, 0, C - WB, R, WB, bs);
  }
  for (int t = 0; t < 2 && !rc; ++t) {
    rc = lbm_cg_step_fused(sv->rband[t ^ 1][0], sv->rband[t ^ 1][1], sv->rband[t][0], sv->rband[t][1], &sv->rbg, &sv->bc, &sv->prm, 0,
                           2 * HB, nullptr, nullptr, nullptr, nullptr, nullptr, bs);
    if (!rc) rc = lbm_cg_step_fused(sv->cband[t ^ 1][0], sv->cband[t ^ 1][1], sv->cband[t][0], sv->cband[t][1], &sv->cbg, &sv->bc, &sv->prm,
                                    0, R, nullptr, nullptr, nullptr, nullptr, nullptr, bs);
  }
  if (rc) return rc;
  // ---- the inner rectangle: two steps in one pass, on the caller's stream ----
  {
    constexpr int Wv = 4, S = 64 * Wv - 2 * CG_X2_EDGE;
    const Geom g = make_geom(sv->g);
    const CgFast cf = make_cg_fast(make_cg_consts(sv->prm));
    const int ra = kCgX2RowBand, rb = R - kCgX2RowBand, ca = kCgX2ColBand, cb = C - kCgX2ColBand;
    const int win0 = (ca - CG_X2_EDGE) / 16 * 16;
    const int bstrips = (cb - (win0 + CG_X2_EDGE) + S - 1) / S;
    const int mode = tuning("cg_x2_unroll", 0) & 3;
    const void* kfn = mode == 0 ? (const void*)k_cg_two_step<Wv, 0> : mode == 1 ? (const void*)k_cg_two_step<Wv, 1>
                    : mode == 2 ? (const void*)k_cg_two_step<Wv, 2> : (const void*)k_cg_two_step<Wv, 3>;
    int rpc = tuning("cg_rows2", 0);
    if (rpc <= 0) {
      const long slots = sw_wave_slots(kfn, 64 * Wv);
      rpc = slots > 0 ? sw_pick_rows(rb - ra, bstrips * Wv, 8, slots) : 256;  // 14 warm-up rows ~ a depth-8 window's
    }
    if (rpc > rb - ra) rpc = rb - ra;
    const int chunks = (rb - ra + rpc - 1) / rpc;
#define LBM_CG_X2(M) LBM_KLAUNCH((k_cg_two_step<Wv, M>), dim3(bstrips * chunks), dim3(64 * Wv), 0, st, dst[0], dst[1], src[0], src[1], g, cf, ra, rb, ca, cb, rpc, bstrips, win0)
    if (mode == 0) LBM_CG_X2(0); else if (mode == 1) LBM_CG_X2(1); else if (mode == 2) LBM_CG_X2(2); else LBM_CG_X2(3);
#undef LBM_CG_X2
    LBM_CHECK_LAUNCH();
  }
  // ---- the frame's valid part into the new lattice (behind the big launch: the regions are disjoint, but one stream writes) ----
  LBM_CHECK_HIP(hipEventRecord(sv->ev_band_join, bs));
  LBM_CHECK_HIP(hipStreamWaitEvent(st, sv->ev_band_join, 0));
  for (int k = 0; k < 2 && !rc; ++k) {
    rc = box_copy(dst[k], sv->g, 0, 0, sv->rband[0][k], sv->rbg, 0, 0, kCgX2RowBand, C, st);
    if (!rc) rc = box_copy(dst[k], sv->g, R - kCgX2RowBand, 0, sv->rband[0][k], sv->rbg, 2 * HB - kCgX2RowBand, 0, kCgX2RowBand, C, st);
    if (!rc) rc = box_copy(dst[k], sv->g, 0, 0, sv->cband[0][k], sv->cbg, 0, 0, R, kCgX2ColBand, st);
    if (!rc) rc = box_copy(dst[k], sv->g, 0, C - kCgX2ColBand, sv->cband[0][k], sv->cbg, 0, 2 * WB - kCgX2ColBand, R, kCgX2ColBand, st);
  }
  if (rc) return rc;
  sv->cur ^= 1;
  sv->steps += 2;
  ++sv->pair_launches;
  return LBM_OK;
}

#endif  // LBM_EXPERIMENTS

long long lbm_cg_solver_pair_launches(const lbm_cg_solver* sv) { return sv ? sv->pair_launches : -1; }
int lbm_cg_last_inner_form(void) { return lbm::g_last_inner_form; }

int lbm_cg_solver_step(lbm_cg_solver* sv, int n_steps) {
  LBM_REQUIRE(sv && n_steps >= 0, "lbm_cg_solver_step: bad argument");
  const bool fused = sv->prm.form == LBM_FORM_DEFAULT ? tuning("cg_fused", 1) != 0 : sv->prm.form == LBM_FORM_REASSOCIATED;
  (void)fused;
  for (int i = 0; i < n_steps; ++i) {
#ifdef LBM_EXPERIMENTS
    // "cg_depth" = 2 (opt-in): two steps per pass while at least three remain (the LAST step of a call writes the observable
    // fields: a single step).  Bit-identical and slower: 13.6 k against 15.7-16.1 k MLUPS at 8192 x 2048.
    if (fused && tuning("cg_depth", 1) >= 2 && cg_two_step_applies(sv) && sv->post && n_steps - i >= 3) {
      int rc = cg_solver_step2(sv);
      if (rc) return rc;
      ++i;
      continue;
    }
#endif
    double** src = sv->lat[sv->cur];
    double** dst = sv->lat[sv->cur ^ 1];
    int rc;
    if (!sv->post) {  // iteration on the given (rho, u): the driver's first pass through :431-464
      rc = lbm_cg_collide(dst[0], dst[1], src[0], src[1], sv->rho_r, sv->rho_b, sv->u, &sv->g,
                          &sv->bc, &sv->prm, sv->psi, sv->snu, sv->st);
    } else if (sv->prm.form == LBM_FORM_DEFAULT ? tuning("cg_fused", 1) != 0 : sv->prm.form == LBM_FORM_REASSOCIATED) {
      // one launch per step; the observable fields are written by the last step of the call
      const bool last = (i == n_steps - 1);
      rc = lbm_cg_step_fused(dst[0], dst[1], src[0], src[1], &sv->g, &sv->bc, &sv->prm, 0, sv->g.R,
                             last ? sv->rho_r : nullptr, last ? sv->rho_b : nullptr,
                             last ? sv->u : nullptr, last ? sv->psi : nullptr,
                             last ? sv->snu : nullptr, sv->st);
    } else {
      rc = lbm_cg_stream_moments(sv->rho_r, sv->rho_b, sv->u, src[0], src[1], &sv->g, &sv->bc,
                                 &sv->prm, sv->st);
      if (rc) return rc;
      rc = lbm_cg_stream_collide(dst[0], dst[1], src[0], src[1], sv->rho_r, sv->rho_b, sv->u,
                                 &sv->g, &sv->bc, &sv->prm, 0, sv->g.R, sv->psi, sv->snu, sv->st);
    }
    if (rc) return rc;
    sv->cur ^= 1;
    sv->post = true;
    ++sv->steps;
  }
  return LBM_OK;
}

// State as the reference holds it after its loop ran: adv_f of both colours, rho_r, rho_b, u
// (all refreshed from the streamed populations, :466-477) and the last psi / s_nu.
// Any output pointer may be NULL.
int lbm_cg_solver_get_state(lbm_cg_solver* sv, double* f_r, double* f_b, double* rho_r,
                            double* rho_b, double* u, double* psi, double* snu) {
  LBM_REQUIRE(sv, "lbm_cg_solver_get_state: NULL solver");
  const int R = sv->g.R, C = sv->g.C;
  const size_t n = (size_t)R * C;
  double* fo[2] = {f_r, f_b};
  double* scratch = sv->lat[sv->cur ^ 1][0];
  for (int k = 0; k < 2; ++k) {
    if (!fo[k]) continue;
    const double* src = sv->lat[sv->cur][k];
    if (sv->post) {
      int rc = lbm_stream(scratch, src, &sv->g, &sv->bc, sv->st);
      if (rc) return rc;
      src = scratch;
    }
    int rc = lbm_soa_to_aos_pitched(sv->stage, src, R, C, 9, sv->g.plane_stride, sv->g.row_pitch, sv->st);
    if (rc) return rc;
    LBM_CHECK_HIP(hipMemcpyAsync(fo[k], sv->stage, n * 72, hipMemcpyDeviceToHost, sv->st));
    LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  }
  if (rho_r || rho_b || u) {
    if (sv->post) {
      int rc = lbm_cg_stream_moments(sv->rho_r, sv->rho_b, sv->u, sv->lat[sv->cur][0],
                                     sv->lat[sv->cur][1], &sv->g, &sv->bc, &sv->prm, sv->st);
      if (rc) return rc;
    }
    if (rho_r) LBM_CHECK_HIP(hipMemcpyAsync(rho_r, sv->rho_r, n * 8, hipMemcpyDeviceToHost, sv->st));
    if (rho_b) LBM_CHECK_HIP(hipMemcpyAsync(rho_b, sv->rho_b, n * 8, hipMemcpyDeviceToHost, sv->st));
    if (u) {
      int rc = lbm_soa_to_aos(sv->stage, sv->u, R, C, 2, sv->st);
      if (rc) return rc;
      LBM_CHECK_HIP(hipMemcpyAsync(u, sv->stage, n * 16, hipMemcpyDeviceToHost, sv->st));
    }
  }
  if (psi) LBM_CHECK_HIP(hipMemcpyAsync(psi, sv->psi, n * 8, hipMemcpyDeviceToHost, sv->st));
  if (snu) LBM_CHECK_HIP(hipMemcpyAsync(snu, sv->snu, n * 8, hipMemcpyDeviceToHost, sv->st));
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  return LBM_OK;
}

int lbm_cg_solver_sync(lbm_cg_solver* sv) {
  LBM_REQUIRE(sv, "lbm_cg_solver_sync: NULL solver");
  LBM_CHECK_HIP(hipStreamSynchronize(sv->st));
  return LBM_OK;
}

}  // extern "C"

// ---- asynchronous snapshots of the two-phase solver (reference: the [R,C,n_snap] stacks written by
// torch::save at the end of a run, mrtcg_rayleigh_taylor.cpp:481-485) ---------------------------------
// record(): rho_r, rho_b, u of the CURRENT state (pass A on the resident post-collision lattices =
// what the driver holds after the iterations run so far) -> device staging on the solver's stream
// -> pinned host on a private stream; returns at once, the solver may keep stepping.
struct lbm_cg_snapshot {
  lbm_cg_solver* sv;
  hipStream_t copy;
  hipEvent_t ready, done;
  double* d_stage;  // [rho_r | rho_b | u (AoS)] = 4 n doubles
  double* h_stage;  // pinned host, same layout
  long step;
};

extern "C" {

int lbm_cg_snapshot_destroy(lbm_cg_snapshot* sn) {
  if (!sn) return LBM_OK;
  if (sn->copy) (void)hipStreamSynchronize(sn->copy);
  if (sn->d_stage) (void)hipFree(sn->d_stage);
  if (sn->h_stage) (void)hipHostFree(sn->h_stage);
  if (sn->ready) (void)hipEventDestroy(sn->ready);
  if (sn->done) (void)hipEventDestroy(sn->done);
  if (sn->copy) (void)hipStreamDestroy(sn->copy);
  delete sn;
  return LBM_OK;
}

int lbm_cg_snapshot_create(lbm_cg_snapshot** out, lbm_cg_solver* sv) {
  LBM_REQUIRE(out && sv, "lbm_cg_snapshot_create: NULL argument");
  lbm_cg_snapshot* sn = new (std::nothrow) lbm_cg_snapshot();
  LBM_REQUIRE(sn, "lbm_cg_snapshot_create: out of host memory");
  *sn = lbm_cg_snapshot{sv, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  const size_t n = (size_t)sv->g.R * sv->g.C;
  hipError_t e = hipStreamCreateWithFlags(&sn->copy, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sn->ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&sn->done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(&sn->d_stage, n * 32);
  if (e == hipSuccess) e = hipHostMalloc(&sn->h_stage, n * 32);
  if (e != hipSuccess) {
    set_error("lbm_cg_snapshot_create: %s", hipGetErrorString(e));
    lbm_cg_snapshot_destroy(sn);
    return LBM_ERR_HIP;
  }
  *out = sn;
  return LBM_OK;
}

int lbm_cg_snapshot_record(lbm_cg_snapshot* sn) {
  LBM_REQUIRE(sn, "lbm_cg_snapshot_record: NULL snapshot");
  lbm_cg_solver* sv = sn->sv;
  const int R = sv->g.R, C = sv->g.C;
  const size_t n = (size_t)R * C;
  if (sv->post) {  // moments of the streamed state, :466-477
    int rc = lbm_cg_stream_moments(sv->rho_r, sv->rho_b, sv->u, sv->lat[sv->cur][0], sv->lat[sv->cur][1],
                                   &sv->g, &sv->bc, &sv->prm, sv->st);
    if (rc) return rc;
  }
  LBM_CHECK_HIP(hipStreamWaitEvent(sv->st, sn->done, 0));  // previous D2H of this snapshot finished
  LBM_CHECK_HIP(hipMemcpyAsync(sn->d_stage, sv->rho_r, n * 8, hipMemcpyDeviceToDevice, sv->st));
  LBM_CHECK_HIP(hipMemcpyAsync(sn->d_stage + n, sv->rho_b, n * 8, hipMemcpyDeviceToDevice, sv->st));
  int rc = lbm_soa_to_aos(sn->d_stage + 2 * n, sv->u, R, C, 2, sv->st);
  if (rc) return rc;
  LBM_CHECK_HIP(hipEventRecord(sn->ready, sv->st));
  LBM_CHECK_HIP(hipStreamWaitEvent(sn->copy, sn->ready, 0));
  LBM_CHECK_HIP(hipMemcpyAsync(sn->h_stage, sn->d_stage, n * 32, hipMemcpyDeviceToHost, sn->copy));
  LBM_CHECK_HIP(hipEventRecord(sn->done, sn->copy));
  sn->step = sv->steps;
  return LBM_OK;
}

int lbm_cg_snapshot_host(lbm_cg_snapshot* sn, const double** rho_r, const double** rho_b,
                         const double** u, long long* step) {
  LBM_REQUIRE(sn, "lbm_cg_snapshot_host: NULL snapshot");
  LBM_CHECK_HIP(hipStreamSynchronize(sn->copy));
  const size_t n = (size_t)sn->sv->g.R * sn->sv->g.C;
  if (rho_r) *rho_r = sn->h_stage;
  if (rho_b) *rho_b = sn->h_stage + n;
  if (u) *u = sn->h_stage + 2 * n;
  if (step) *step = sn->step;
  return LBM_OK;
}

int lbm_cg_snapshot_write_npy(lbm_cg_snapshot* sn, const char* rho_r_path, const char* rho_b_path,
                              const char* u_path) {
  LBM_REQUIRE(sn, "lbm_cg_snapshot_write_npy: NULL snapshot");
  LBM_CHECK_HIP(hipStreamSynchronize(sn->copy));
  const long R = sn->sv->g.R, C = sn->sv->g.C;
  const size_t n = (size_t)R * C;
  int rc = LBM_OK;
  if (rho_r_path) rc = write_npy(rho_r_path, sn->h_stage, {R, C});
  if (!rc && rho_b_path) rc = write_npy(rho_b_path, sn->h_stage + n, {R, C});
  if (!rc && u_path) rc = write_npy(u_path, sn->h_stage + 2 * n, {R, C, 2});
  return rc;
}

}  // extern "C"
