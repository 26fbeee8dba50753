// EXPERIMENTS build only (make -C lattice-boltzmann-method_amd/csrc EXPERIMENTS=1): two-phase step, two steps per pass (k_cg_two_step)
// Measured and not kept -- DESIGN.md 4.2 / 9 hold the numbers.  Included from cg_fused.hpp at the place the code used to stand;
// not a stand-alone header (it uses what that file has declared above the include).
// bit-identical to two single steps and SLOWER than them: make EXPERIMENTS=1
// ---- TWO time steps per pass (round 3) -----------------------------------------------------------------
// The k_cg_strip4 structure with a second level on top: a workgroup of W waves walks down a line-aligned 64 W-column window
// in lockstep; level 1 is the single step of k_cg_strip4 (rows read from HBM, psi / Q in the block's ring 1), but its
// post-collision populations (18 per node) go into LDS instead of HBM; level 2 pull-streams them -- the +-1-lane offset of
// the ds_read IS the column shift, rows r-1, r, r+1 come from a compact ring (of a row published in iteration i the next
// level reads the c_x = -1 populations in iteration i, the c_x = 0 ones in i + 1, the c_x = +1 ones in i + 2: 1 + 2 + 3
// slots of 3 populations per colour) --, reduces them to the step-(t+1) macroscopic fields (ring 2), and collides the row
// three behind, which is stored.  Per iteration: A level-1 reduce | barrier | C level-1 collide -> LDS | barrier | B level-2
// pull + reduce, D level-2 collide -> HBM.  Every lattice row is read once and written once per TWO steps: 144 B per update
// instead of 288.  One wave per SIMD (the colour sums and macroscopic values of 3 + 4 rows wait in registers), 141 KB of LDS
// per 4-wave block.  Only for nodes whose two-step dependency cone holds plain nodes (launch_cg_two_steps: the frame of the
// lattice advances two single steps on small band lattices).  Per-node arithmetic = two applications of the tile kernel's:
// identical bits.
constexpr int CG_X2_EDGE = 8;   // ring-only lanes at each end of a block's window (the two levels need 6)
constexpr int CG_X2_WARM = 14;  // pipeline iterations before the first stored row

template <int W>
struct CgX2Lds {
  double ring1[3][5][64 * W];    // [psi, Qx, Qy][slot][block lane]: step t+0 fields of the last 5 level-1 rows
  double ring2[3][6][64 * W];    // the same for level 2 (6 slots: rows are consumed one iteration later than in level 1)
  double p1[2][6][3][64 * W];    // [colour][slot: A | B0 B1 | C0 C1 C2][population of the group][block lane]
};

// the four 5x5 stencil results of the node at block lane l0 + 2 from a ring whose rows r-2 .. r+2 sit in slots sl[0..4]
// (one wave per SIMD and registers to spare: the 80 ring reads are issued together; accumulation order of cg_ddrow / cg_ddcol)
template <int LN, bool UNROLL>
__device__ __forceinline__ void cg_stencils(const double (*s_psi)[LN], const double (*s_qx)[LN], const double (*s_qy)[LN],
                                            const int (&sl)[5], int l0, double& gx, double& gy, double& dxqx, double& dyqy) {
#pragma clang fp contract(on)
  constexpr double k = 1.0 / 5040.0;
  constexpr double a0[5] = {2 * k * 1, 2 * k * 32, 2 * k * 84, 2 * k * 32, 2 * k * 1};
  constexpr double a1[5] = {k * 32, k * 448, k * 960, k * 448, k * 32};
  gx = 0.0, dxqx = 0.0, gy = 0.0, dyqy = 0.0;
  if constexpr (UNROLL) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {  // == cg_ddrow
      gx += a0[j] * (s_psi[sl[4]][l0 + j] - s_psi[sl[0]][l0 + j]);
      gx += a1[j] * (s_psi[sl[3]][l0 + j] - s_psi[sl[1]][l0 + j]);
      dxqx += a0[j] * (s_qx[sl[4]][l0 + j] - s_qx[sl[0]][l0 + j]);
      dxqx += a1[j] * (s_qx[sl[3]][l0 + j] - s_qx[sl[1]][l0 + j]);
    }
#pragma unroll
    for (int ii = 0; ii < 5; ++ii) {  // == cg_ddcol
      gy += a0[ii] * (s_psi[sl[ii]][l0 + 4] - s_psi[sl[ii]][l0]);
      gy += a1[ii] * (s_psi[sl[ii]][l0 + 3] - s_psi[sl[ii]][l0 + 1]);
      dyqy += a0[ii] * (s_qy[sl[ii]][l0 + 4] - s_qy[sl[ii]][l0]);
      dyqy += a1[ii] * (s_qy[sl[ii]][l0 + 3] - s_qy[sl[ii]][l0 + 1]);
    }
  } else {
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
      gx += a0[j] * (s_psi[sl[4]][l0 + j] - s_psi[sl[0]][l0 + j]);
      gx += a1[j] * (s_psi[sl[3]][l0 + j] - s_psi[sl[1]][l0 + j]);
      dxqx += a0[j] * (s_qx[sl[4]][l0 + j] - s_qx[sl[0]][l0 + j]);
      dxqx += a1[j] * (s_qx[sl[3]][l0 + j] - s_qx[sl[1]][l0 + j]);
    }
#pragma unroll 1
    for (int ii = 0; ii < 5; ++ii) {
      gy += a0[ii] * (s_psi[sl[ii]][l0 + 4] - s_psi[sl[ii]][l0]);
      gy += a1[ii] * (s_psi[sl[ii]][l0 + 3] - s_psi[sl[ii]][l0 + 1]);
      dyqy += a0[ii] * (s_qy[sl[ii]][l0 + 4] - s_qy[sl[ii]][l0]);
      dyqy += a1[ii] * (s_qy[sl[ii]][l0 + 3] - s_qy[sl[ii]][l0 + 1]);
    }
  }
}

// one pipeline iteration; STEADY: i >= CG_X2_WARM, so both collisions run unconditionally -- in ONE barrier interval, as
// straight-line code the scheduler can interleave (at one wave per SIMD nothing else hides their dependent chains)
template <int W, bool STEADY, int MODE>
__device__ __forceinline__ void cg_two_step_iter(CgX2Lds<W>& L, double (&raw_r)[Q], double (&raw_b)[Q], double (&s1)[3][Q],
                                                 double (&n1)[3][6], double (&s2)[3][Q], double (&n2)[3][6],
                                                 double* __restrict__ pn_r, double* __restrict__ pn_b,
                                                 const double* __restrict__ in_r, const double* __restrict__ in_b, const Geom& g,
                                                 const CgFast& cf, int i, int n_iter, int R0, int gl, int l0, int cl, int c,
                                                 bool lane_out) {
#pragma clang fp contract(on)
  constexpr int LN = 64 * W;
  // ---- A: level-1 reduce of the arrived row m1 = R0 - 8 + i ----
#pragma unroll
  for (int q = 0; q < Q; ++q) s1[2][q] = s1[1][q], s1[1][q] = s1[0][q];
#pragma unroll
  for (int q = 0; q < 6; ++q) n1[2][q] = n1[1][q], n1[1][q] = n1[0][q];
  {
#pragma unroll
    for (int q = 0; q < Q; ++q) s1[0][q] = raw_b[q];
    double psi, qx, qy;
    cg_reduce_row(raw_r, s1[0], cf, n1[0], psi, qx, qy);
    const int slot = i % 5;
    L.ring1[0][slot][gl] = psi;
    L.ring1[1][slot][gl] = qx;
    L.ring1[2][slot][gl] = qy;
  }
  if (i + 1 < n_iter) {  // the next row into the registers just freed; in flight during the collisions below
    const long o = g.at(R0 - 7 + i, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[q] = in_r[off];
      raw_b[q] = in_b[off];
    }
  }
  __syncthreads();
  // ---- C: level-1 collision of row m1 - 2 (reduced two iterations ago); its populations of step t+1 into LDS ----
  // ---- D: level-2 collision of row m2 - 3 = R0 - 14 + i (its ring rows were all published in earlier iterations) ----
  double o_r[Q], o_b[Q], o2_r[Q], o2_b[Q], snu1, snu2;
  if (STEADY || i >= 4) {
    const int sl[5] = {(i - 4) % 5, (i - 3) % 5, (i - 2) % 5, (i - 1) % 5, i % 5};
    double gx, gy, dxqx, dyqy;
    cg_stencils<LN, (MODE & 1) != 0>(L.ring1[0], L.ring1[1], L.ring1[2], sl, l0, gx, gy, dxqx, dyqy);
    CgNode me;
    me.rr = n1[2][0]; me.rb = n1[2][1]; me.ux = n1[2][2]; me.uy = n1[2][3]; me.irt = n1[2][4]; me.psi = n1[2][5];
    me.qx = 0.0; me.qy = 0.0;
    cg_collide_values(s1[2], me, gx, gy, dxqx, dyqy, cf, o_r, o_b, snu1);
  }
  if (STEADY) {
    const int sl[5] = {(i - 5) % 6, (i - 4) % 6, (i - 3) % 6, (i - 2) % 6, (i - 1) % 6};
    double gx, gy, dxqx, dyqy;
    cg_stencils<LN, (MODE & 2) != 0>(L.ring2[0], L.ring2[1], L.ring2[2], sl, l0, gx, gy, dxqx, dyqy);
    CgNode me;
    // (before this iteration's rotation of the level-2 rows: [2] = the row reduced three iterations ago)
    me.rr = n2[2][0]; me.rb = n2[2][1]; me.ux = n2[2][2]; me.uy = n2[2][3]; me.irt = n2[2][4]; me.psi = n2[2][5];
    me.qx = 0.0; me.qy = 0.0;
    cg_collide_values(s2[2], me, gx, gy, dxqx, dyqy, cf, o2_r, o2_b, snu2);
  }
  if (STEADY || i >= 4) {
    const int sb = 1 + (i & 1), sc = 3 + i % 3;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int slot = icx(q) == -1 ? 0 : (icx(q) == 0 ? sb : sc);
      L.p1[0][slot][sw_grp_pos(q)][gl] = o_r[q];
      L.p1[1][slot][sw_grp_pos(q)][gl] = o_b[q];
    }
  }
  if (STEADY && lane_out) {
    const long lo = g.at(R0 - CG_X2_WARM + i, c);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      __builtin_nontemporal_store(o2_r[q], &pn_r[q * g.plane + lo]);
      __builtin_nontemporal_store(o2_b[q], &pn_b[q * g.plane + lo]);
    }
  }
  __syncthreads();
  // ---- B: level-2 pull of row m2 = m1 - 3 from the level-1 rows m2 + 1 (this iteration), m2, m2 - 1; reduce ----
#pragma unroll
  for (int q = 0; q < Q; ++q) s2[2][q] = s2[1][q], s2[1][q] = s2[0][q];
#pragma unroll
  for (int q = 0; q < 6; ++q) n2[2][q] = n2[1][q], n2[1][q] = n2[0][q];
  if (STEADY || i >= 6) {
    const int sb = 1 + ((i - 1) & 1), sc = 3 + (i - 2) % 3;
    double fr[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int slot = icx(q) == -1 ? 0 : (icx(q) == 0 ? sb : sc);
      int lsrc = gl - icy(q);
      lsrc = lsrc < 0 ? 0 : (lsrc > LN - 1 ? LN - 1 : lsrc);
      fr[q] = L.p1[0][slot][sw_grp_pos(q)][lsrc];
      s2[0][q] = L.p1[1][slot][sw_grp_pos(q)][lsrc];
    }
    double psi, qx, qy;
    cg_reduce_row(fr, s2[0], cf, n2[0], psi, qx, qy);
    const int slot = i % 6;
    L.ring2[0][slot][gl] = psi;
    L.ring2[1][slot][gl] = qx;
    L.ring2[2][slot][gl] = qy;
  }
}

// MODE: bit 0 / bit 1 = the stencil loops of level 1 / level 2 unrolled (all 80 ring reads of a level in flight at once)
template <int W, int MODE>
__global__ __launch_bounds__(64 * W, 1) void k_cg_two_step(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, int row_begin, int row_end, int col_begin, int col_end,
    int rows_per_chunk, int bstrips, int win0) {
  constexpr int LN = 64 * W;
  __shared__ CgX2Lds<W> L;
  const int gl = threadIdx.x;
  constexpr int S = LN - 2 * CG_X2_EDGE;  // output columns per block (a multiple of 16: windows stay line-aligned)
  const int bs = blockIdx.x % bstrips, chunk = blockIdx.x / bstrips;
  const int R0 = row_begin + chunk * rows_per_chunk;
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int c = win0 + bs * S + gl;
  const bool lane_out = gl >= CG_X2_EDGE && gl < LN - CG_X2_EDGE && c >= col_begin && c < col_end;
  const int cl = c < 1 ? 1 : (c > g.C - 2 ? g.C - 2 : c);
  const int l0 = gl < 2 ? 0 : (gl > LN - 3 ? LN - 5 : gl - 2);  // stencil window of this lane, kept inside the ring
  double raw_r[Q], raw_b[Q];
  double s1[3][Q], n1[3][6];  // level 1: colour sums / (rho_r, rho_b, ux, uy, 1/rho, psi) of its last 3 rows, [0] = newest
  double s2[3][Q], n2[3][6];  // level 2: of its last 3 rows (collided before the iteration's rotation)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int q = 0; q < Q; ++q) s1[a][q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) n1[a][q] = 1.0;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int q = 0; q < Q; ++q) s2[a][q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) n2[a][q] = 1.0;
  }
  {
    const long o = g.at(R0 - 8, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[q] = in_r[off];
      raw_b[q] = in_b[off];
    }
  }
  const int n_iter = (R1 - R0) + CG_X2_WARM;
  for (int i = 0; i < CG_X2_WARM; ++i)
    cg_two_step_iter<W, false, MODE>(L, raw_r, raw_b, s1, n1, s2, n2, pn_r, pn_b, in_r, in_b, g, cf, i, n_iter, R0, gl, l0, cl, c, lane_out);
  for (int i = CG_X2_WARM; i < n_iter; ++i)
    cg_two_step_iter<W, true, MODE>(L, raw_r, raw_b, s1, n1, s2, n2, pn_r, pn_b, in_r, in_b, g, cf, i, n_iter, R0, gl, l0, cl, c, lane_out);
}
