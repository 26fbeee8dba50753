// EXPERIMENT (round 4, measured and not kept): the walking 16 x 64 tile of the two-phase step.  Included by cg_fused.hpp in
// the LBM_EXPERIMENTS build only (tuning "cg_big" = 10).  Bit-identical to k_cg_tile_mn; reads 3.74 GB per step instead of
// 3.99 (traffic 1.35x the algorithmic bytes instead of 1.39x) and still runs at 15.0 - 15.8 k MLUPS where the independent
// tiles run at 16.7 k on the same box: two barriers per 16 rows keep a workgroup's eight waves in lock step, the tile kernel's
// workgroups drift apart and overlap their phases (profiles/r04_cg_walk_tile.txt).
#pragma once
// ---- the same tile, WALKING down a column of tiles (round 4) --------------------------------------------------------------
// k_cg_tile_mn pays for its ring rows: 22 rows fetched per 16 computed.  Here a workgroup keeps its 16 x 64 tile shape, its two
// nodes per thread, its one parked node and its two workgroups per CU, but advances through a chunk of rows in steps of 16 and
// carries what the next step needs instead of fetching it again:
//   * psi, Q_x, Q_y live in a RING of 20 rows (the rows a step's stencils read: 2 above its 16, 2 below); a step evaluates only
//     the 16 NEW rows -- its own rows 2 .. 15 and the two rows below it, 16 and 17 -- 18 source rows for 16 rows of nodes;
//   * rows 16 and 17 are the next step's rows 0 and 1: the threads that evaluated them (waves 6 and 7, second pass) keep their
//     colour sums in a private LDS slot (two buffers, alternating) and collide them one step later -- nothing is gathered twice;
//   * a chunk starts with a warm-up that evaluates rows -2 .. 1.
// Per step: gather + reduce | barrier | two collisions per thread | barrier (the ring slots the next step overwrites are the ones
// this step's stencils read).  Same per-node expressions as k_cg_tile_mn / k_cg_fused: identical bits.  Inner rectangle only.
constexpr int CG_WT_RB = 20;  // rows in the field ring
// 5x5 derivatives on the ring: rows rs, rs + 1, ... (mod CG_WT_RB) in place of tr .. tr + 4; same operations, same order
template <int LDC>
__device__ __forceinline__ double cg_ddrow_ring(const double (*s)[LDC], int rs, int tc) {
#pragma clang fp contract(on)
  const int r1 = rs + 1 >= CG_WT_RB ? rs + 1 - CG_WT_RB : rs + 1, r3 = rs + 3 >= CG_WT_RB ? rs + 3 - CG_WT_RB : rs + 3;
  const int r4 = rs + 4 >= CG_WT_RB ? rs + 4 - CG_WT_RB : rs + 4;
  double acc = 0.0;
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    const double a0 = CG_STENCIL_TAPS[j][0], a1 = CG_STENCIL_TAPS[j][1];
    acc += a0 * (s[r4][tc + j] - s[rs][tc + j]);
    acc += a1 * (s[r3][tc + j] - s[r1][tc + j]);
  }
  return acc;
}
template <int LDC>
__device__ __forceinline__ double cg_ddcol_ring(const double (*s)[LDC], int rs, int tc) {
#pragma clang fp contract(on)
  double acc = 0.0;
  int r = rs;
#pragma unroll 1
  for (int i = 0; i < 5; ++i) {
    const double a0 = CG_STENCIL_TAPS[i][0], a1 = CG_STENCIL_TAPS[i][1];
    acc += a0 * (s[r][tc + 4] - s[r][tc]);
    acc += a1 * (s[r][tc + 3] - s[r][tc + 1]);
    r = r + 1 >= CG_WT_RB ? 0 : r + 1;
  }
  return acc;
}

template <bool WITH_FIELDS>
__global__ __launch_bounds__(512, 4) void k_cg_walk_tile(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int ra, int ca, int tiles_c, int rows_total, int rows_per_chunk,
    int xcd_swizzle) {
#pragma clang fp contract(on)
  constexpr int TC = 64, NT = 512, LC = TC + 4, LDC = LC + 1, RB = CG_WT_RB;
  __shared__ double s_psi[RB][LDC], s_qx[RB][LDC], s_qy[RB][LDC];
  __shared__ double s_park[Q][384];      // waves 0 .. 5: the node of the second pass (rows 10 .. 15 of the step)
  __shared__ double s_carry[2][Q][128];  // waves 6, 7: rows 16, 17 of step s = rows 0, 1 of step s + 1
  int blk = blockIdx.x;
  if (xcd_swizzle > 1) {  // groups of G column-neighbour strips per XCD inside a common window (as k_cg_tile_mn)
    const int G = xcd_swizzle, x = blk % 8, m = blk / 8, win = 8 * G;
    const int t2 = (m / G) * win + x * G + (m % G);
    if ((m / G + 1) * win <= (int)gridDim.x) blk = t2;
  }
  const int chunk = blk / tiles_c, c_base = ca + (blk % tiles_c) * TC;
  const int R0 = ra + chunk * rows_per_chunk;
  const int rows = rows_total - chunk * rows_per_chunk < rows_per_chunk ? rows_total - chunk * rows_per_chunk : rows_per_chunk;
  const int n_steps = rows / 16;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool carrier = w >= 6;  // wave-uniform

  // evaluate node (rel row rho, lane's column): fields into the ring; returns its colour sums
  auto eval = [&](int rho, int tc, double (&fk)[Q], double& rr, double& rb) {
    const CgNode me = cg_node<true>(fk, in_r, in_b, g, Bc{}, cf, R0 + rho, c_base + tc);
    rr = me.rr;
    rb = me.rb;
    const int sl = (rho + 2) % RB;
    s_psi[sl][tc + 2] = me.psi;
    s_qx[sl][tc + 2] = me.qx;
    s_qy[sl][tc + 2] = me.qy;
  };
  auto side = [&](int rho0, int n_rows, int i) {  // the 2 + 2 ring columns beside rows rho0 .. rho0 + n_rows - 1: threads i < 4 n_rows
    if (i < 4 * n_rows) {
      const int rho = rho0 + (i >> 2), lc = (i & 3) < 2 ? (i & 3) : TC + (i & 3);
      double tmp[Q];
      const CgNode nb = cg_node<true>(tmp, in_r, in_b, g, Bc{}, cf, R0 + rho, c_base + lc - 2);
      const int sl = (rho + 2) % RB;
      s_psi[sl][lc] = nb.psi;
      s_qx[sl][lc] = nb.qx;
      s_qy[sl][lc] = nb.qy;
    }
  };
  // collide node (rel row rho) from its colour sums
  auto collide = [&](int rho_in, int tc, const double (&fk)[Q], double rr, double rb) {
    int rho = rho_in;
    asm volatile("" : "+v"(rho));  // (as in k_cg_tile_mn: store addresses and the uniform source-term products are not carried between nodes)
    CgFast cfk = cf;
    asm volatile("" : "+s"(cfk.Gr), "+s"(cfk.Gc));
    CgNode me;
    me.rr = rr;
    me.rb = rb;
    const double jx = ((fk[1] - fk[3]) + (fk[5] - fk[6])) + (fk[8] - fk[7]);
    const double jy = ((fk[2] - fk[4]) + (fk[5] - fk[8])) + (fk[6] - fk[7]);
    me.irt = 1.0 / (me.rr + me.rb);
    me.ux = (jx + 0.5 * cfk.Gr) * me.irt;
    me.uy = (jy + 0.5 * cfk.Gc) * me.irt;
    const int rs = rho % RB;  // slot of row rho - 2
    const int rc = rs + 2 >= RB ? rs + 2 - RB : rs + 2;
    me.psi = s_psi[rc][tc + 2];
    me.qx = me.qy = 0.0;
    const double gx = cg_ddrow_ring<LDC>(s_psi, rs, tc), gy = cg_ddcol_ring<LDC>(s_psi, rs, tc);
    const double dxqx = cg_ddrow_ring<LDC>(s_qx, rs, tc), dyqy = cg_ddcol_ring<LDC>(s_qy, rs, tc);
    cg_collide_store<WITH_FIELDS>(fk, me, gx, gy, dxqx, dyqy, cfk, g, mi, R0 + rho, c_base + tc, pn_r, pn_b, rho_r_out,
                                  rho_b_out, u_out, psi_out, snu_out);
  };

  // ---- warm-up: rows -2, -1 (fields), rows 0, 1 (fields + carry) ------------------------------------------------------------
  double crr = 0.0, crb = 0.0;  // densities of the carried node (waves 6, 7)
  {
    const int tid = threadIdx.x, tc = tid & 63;
    double fk[Q], rr, rb;
    if (w < 2) eval(-2 + w, tc, fk, rr, rb);
    if (carrier) {
      eval(w - 6, tc, fk, crr, crb);
#pragma unroll
      for (int q = 0; q < Q; ++q) s_carry[0][q][tid - 384] = fk[q];
    }
    if (w == 0) side(-2, 4, tid);
  }
  __syncthreads();
  for (int s = 0; s < n_steps; ++s) {
    const int rho0 = 16 * s;
    // the thread index is made opaque per step: everything derived from it (column, LDS addresses, byte offsets) is then
    // recomputed in the step instead of being hoisted out of the loop into twenty spilled registers
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int tc = tid & 63, cslot = tid - 384;
    // ---- phase 1: the 16 new rows rho0 + 2 .. rho0 + 17 ---------------------------------------------------------------------
    // The loads of the second-pass node are ISSUED AHEAD of the barrier that ends the previous step: the compiler puts an
    // s_waitcnt vmcnt(0) in front of every s_barrier on this target, i.e. that barrier also waits for the acknowledgement of
    // the 36 stores each thread has just issued -- with loads already in flight the wait costs the longer of the two
    // latencies instead of their sum.  Nothing is written to LDS before the barrier.
    double fa[Q], rra, rba, nrr, nrb;
    {
      double fr[Q], fb[Q];
      cg_node_gather(fr, fb, in_r, in_b, g, R0 + rho0 + 10 + w, c_base + tc);
      __syncthreads();  // the ring slots / park slots written below are the ones the previous step's collisions read
      const CgNode nb = cg_node_reduce(fr, fb, cf);
      const int sl = (rho0 + 10 + w + 2) % RB;
      s_psi[sl][tc + 2] = nb.psi;
      s_qx[sl][tc + 2] = nb.qx;
      s_qy[sl][tc + 2] = nb.qy;
      nrr = nb.rr;  // (the densities wait in registers)
      nrb = nb.rb;
      if (!carrier) {
#pragma unroll
        for (int q = 0; q < Q; ++q) s_park[q][tid] = fb[q];
      } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) s_carry[(s + 1) & 1][q][cslot] = fb[q];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    eval(rho0 + 2 + w, tc, fa, rra, rba);
    if (w == 0) side(rho0 + 2, 16, tid);
    __syncthreads();
    // ---- phase 2: rows rho0 .. rho0 + 15 ----------------------------------------------------------------------------------------
    collide(rho0 + 2 + w, tc, fa, rra, rba);
    __builtin_amdgcn_sched_barrier(0);
    {
      double fb[Q];
      if (!carrier) {
#pragma unroll
        for (int q = 0; q < Q; ++q) fb[q] = s_park[q][tid];
        collide(rho0 + 10 + w, tc, fb, nrr, nrb);
      } else {
#pragma unroll
        for (int q = 0; q < Q; ++q) fb[q] = s_carry[s & 1][q][cslot];
        collide(rho0 + (w - 6), tc, fb, crr, crb);
        crr = nrr;
        crb = nrb;
      }
    }
  }
}
