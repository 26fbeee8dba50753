// EXPERIMENTS build only (make -C lattice-boltzmann-method_amd/csrc EXPERIMENTS=1): two-phase step, merged frame + inner dispatch and the strip kernels of generations 1 - 4
// Measured and not kept -- DESIGN.md 4.2 / 9 hold the numbers.  Included from cg_fused.hpp at the place the code used to stand;
// not a stand-alone header (it uses what that file has declared above the include).
// the launch forms below (merged dispatch, strip kernels 1 - 4) were measured and not kept: make EXPERIMENTS=1
// frame tiles and inner tiles in ONE dispatch: workgroups [0, n_frame) run the frame instantiation (general boundary
// gather), the rest the inner one (plain offsets).  The two-launch form either runs the frame behind the inner launch
// (63 us) or beside it on a helper stream, whose event fork / join costs as much as it hides (profiles/r02_ring_dissect.txt).
template <int TR, int TC, int WAVES, bool WITH_FIELDS>
__global__ __launch_bounds__(TR* TC, WAVES) void k_cg_fused_merged(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, Bc bc, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int xcd_swizzle, CgTileRect rect,
    int n_frame) {
  if ((int)blockIdx.x < n_frame)
    cg_fused_body<TR, TC, WITH_FIELDS, 2>(pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r_out, rho_b_out, u_out, psi_out, snu_out, mi,
                                          row_begin, row_end, 0, rect, (int)blockIdx.x, n_frame);
  else
    cg_fused_body<TR, TC, WITH_FIELDS, 1>(pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r_out, rho_b_out, u_out, psi_out, snu_out, mi,
                                          row_begin, row_end, xcd_swizzle, rect, (int)blockIdx.x - n_frame, (int)gridDim.x - n_frame);
}

// ---- column-strip sliding window --------------------------------------------------------------
// One WAVEFRONT owns a strip of 64 columns (60 outputs + the +-2 stencil ring) and walks down a
// chunk of rows.  Every iteration it (1) streams one new row of both colours and reduces it to
// psi, Qx, Qy, which go into a wave-private LDS ring of the last 5 rows; (2) collides the row two
// behind: its own 18 populations are gathered again (L2 / L1 hits: the wave read them two
// iterations ago) and the 5x5 stencils read the ring.  Rows are wave-uniform, so the gathers are
// scalar-base + lane-offset loads; nothing is recomputed along r (4 warm-up rows per chunk), only
// 4 of 64 columns along c; no workgroup barrier exists (LDS visibility inside a wave needs only
// program order).  Against the tile kernel above: ~2x less HBM read traffic (its +-3 column ring
// costs whole 128-B lines on both sides of a 32-column tile, 299 B read per node measured).
// Same per-node arithmetic as the tile kernel: identical bits.
// MEASURED (8192 x 2048): 11.5 k MLUPS (4 waves per block, 16 rows per chunk) against the tile
// kernel's 14.1 k -- opt-in (tuning "cg_strip" = 1 / 2 / 4 waves per block, "cg_rows").  History: 6.3 k
// with two inlined copies of the gather and the stencils unrolled (362 VGPRs, 1 wave per SIMD);
// one inlined copy (a 2-pass loop) and rolled stencil loops: 166 VGPRs, 3 waves per SIMD.  What
// still separates it from the tile kernel: the row to collide is gathered a second time (its
// populations are not kept across the two iterations) and nothing is prefetched.
constexpr int CG_SW = 60;  // output columns per wavefront

template <int WAVES, bool WITH_FIELDS>
__global__ __launch_bounds__(64 * WAVES) void k_cg_strip(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, Bc bc, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int rows_per_chunk,
    int strips, int n_waves) {
#pragma clang fp contract(on)
  __shared__ double ring[WAVES][3][5][64 + 4];  // [wave][field][slot][2 pad + lane + 2 pad]
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wave = blockIdx.x * WAVES + wib;
  if (wave >= n_waves) return;
  const int strip = wave % strips, chunk = wave / strips;
  const int R0 = row_begin + chunk * rows_per_chunk;
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int rlo = cg_row_lo(g, bc), rhi = cg_row_hi(g, bc);
  const int c = strip * CG_SW - 2 + lane;                   // this lane's column (may be outside)
  const int cm = c < 0 ? 0 : (c > g.C - 1 ? g.C - 1 : c);   // replicate padding along c
  const bool lane_out = lane >= 2 && lane < 2 + CG_SW && c < g.C;
  // strips that touch column 0 / C-1 need the boundary gather (the same-row column copy, Q5)
  const bool edge_strip = strip * CG_SW - 3 <= 0 || strip * CG_SW + CG_SW + 2 >= g.C - 1;
  double(*s_psi)[68] = ring[wib][0];
  double(*s_qx)[68] = ring[wib][1];
  double(*s_qy)[68] = ring[wib][2];

  auto node = [&](double (&ft)[Q], int row, int col) -> CgNode {
    // rows 0 / R-1 of the block carry wall fix-ups; (single block) their neighbours wrap
    const bool plain = !edge_strip && row >= 1 && row <= g.R - 2;
    if (!plain) return cg_node<false>(ft, in_r, in_b, g, bc, cf, row, col);
    double fr[Q];
    const long ro[3] = {g.at(row + 1, 0), g.at(row, 0), g.at(row - 1, 0)};  // source rows of cx = -1, 0, +1
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long o = q * g.plane + ro[icx(q) + 1] + (col - icy(q));
      fr[q] = in_r[o];
      ft[q] = in_b[o];
    }
    CgNode n;
    n.rr = (((fr[0] + fr[1]) + (fr[2] + fr[3])) + ((fr[4] + fr[5]) + (fr[6] + fr[7]))) + fr[8];
    n.rb = (((ft[0] + ft[1]) + (ft[2] + ft[3])) + ((ft[4] + ft[5]) + (ft[6] + ft[7]))) + ft[8];
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] += fr[q];
    const double jx = ((ft[1] - ft[3]) + (ft[5] - ft[6])) + (ft[8] - ft[7]);
    const double jy = ((ft[2] - ft[4]) + (ft[5] - ft[8])) + (ft[6] - ft[7]);
    n.irt = 1.0 / (n.rr + n.rb);
    n.ux = (jx + 0.5 * cf.Gr) * n.irt;
    n.uy = (jy + 0.5 * cf.Gc) * n.irt;
    const double a = n.rr * cf.inv_rho0[0], b = n.rb * cf.inv_rho0[1];
    n.psi = (a - b) / (a + b);
    const double qcs = cf.qc[0] * n.rr + cf.qc[1] * n.rb;
    n.qx = qcs * n.ux;
    n.qy = qcs * n.uy;
    return n;
  };

  const int n_iter = (R1 - R0) + 4;
  for (int i = 0; i < n_iter; ++i) {
    // two gathers per iteration through ONE inlined copy of `node`: pass 0 = the new macroscopic
    // row R0 - 2 + i (-> ring slot i % 5), pass 1 = the row to collide, r = R0 + i - 4
    double ft[Q];
    CgNode me;
    const int r = R0 + i - 4;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      int row, col;
      if (pass == 0) {
        row = R0 - 2 + i;
        row = row < rlo ? rlo : (row > rhi ? rhi : row);  // replicate padding along r (global edges only)
        col = cm;
      } else {
        if (i < 4) break;
        row = r;
        col = lane_out ? c : cm;
      }
      me = node(ft, row, col);
      if (pass == 0) {
        const int slot = i % 5;
        s_psi[slot][lane + 2] = me.psi;
        s_qx[slot][lane + 2] = me.qx;
        s_qy[slot][lane + 2] = me.qy;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (i < 4) continue;
    if (lane_out) {
      // the 5 ring rows in stencil order; columns lane-2 .. lane+2 sit at [lane .. lane+4]
      const int s0 = (i - 4) % 5, s1 = (i - 3) % 5, s3 = (i - 1) % 5, s4 = i % 5;
      constexpr double k = 1.0 / 5040.0;
      constexpr double a0[5] = {2 * k * 1, 2 * k * 32, 2 * k * 84, 2 * k * 32, 2 * k * 1};
      constexpr double a1[5] = {k * 32, k * 448, k * 960, k * 448, k * 32};
      double gx = 0.0, dxqx = 0.0;
#pragma unroll 1
      for (int j = 0; j < 5; ++j) {  // == cg_ddrow
        gx += a0[j] * (s_psi[s4][lane + j] - s_psi[s0][lane + j]);
        gx += a1[j] * (s_psi[s3][lane + j] - s_psi[s1][lane + j]);
        dxqx += a0[j] * (s_qx[s4][lane + j] - s_qx[s0][lane + j]);
        dxqx += a1[j] * (s_qx[s3][lane + j] - s_qx[s1][lane + j]);
      }
      double gy = 0.0, dyqy = 0.0;
#pragma unroll 1
      for (int ii = 0; ii < 5; ++ii) {  // == cg_ddcol
        const int sl = (i - 4 + ii) % 5;
        gy += a0[ii] * (s_psi[sl][lane + 4] - s_psi[sl][lane]);
        gy += a1[ii] * (s_psi[sl][lane + 3] - s_psi[sl][lane + 1]);
        dyqy += a0[ii] * (s_qy[sl][lane + 4] - s_qy[sl][lane]);
        dyqy += a1[ii] * (s_qy[sl][lane + 3] - s_qy[sl][lane + 1]);
      }
      cg_collide_store<WITH_FIELDS>(ft, me, gx, gy, dxqx, dyqy, cf, g, mi, r, c, pn_r, pn_b, rho_r_out,
                                    rho_b_out, u_out, psi_out, snu_out);
    }
    // the slot written next iteration is (i + 1) % 5 = the oldest row, no longer read: no hazard
  }
}


// ---- column-strip sliding window, second generation: the INNER rectangle of a launch ------------------
// As k_cg_strip, with what separated it from the tile kernel removed:
//   * the colour-summed populations and macroscopic fields of a row are KEPT in registers (a ring of 3
//     rows, the row loop unrolled by 3 so that every index is static) from the iteration that streams
//     the row to the one, two later, that collides it -- nothing is gathered twice;
//   * the 18 loads of the next row are issued before the current row is reduced and collided;
//   * it only ever runs on nodes whose +-2 ring and +-1 gathers are plain (the inner rectangle the tile
//     launch already separates from its frame): no clamps, no wraps, no boundary gather;
//   * 56 output columns per wave (lanes 4..59; lanes 2, 3, 60, 61 carry the stencil ring): every row a
//     wave stores starts on a 64-byte boundary.
// Every lattice row of the rectangle is read once per strip (64 of 56 columns) plus 4 warm-up rows per
// chunk; the frame keeps the tile kernel.  Per-node arithmetic = the tile kernel's: identical bits.
constexpr int CG_SW2 = 56;

template <int K, bool WITH_FIELDS>
__device__ __forceinline__ void cg_strip2_iter(
    double (&rf)[3][Q], double (&rn)[3][6], double (&raw_r)[3][Q], double (&raw_b)[3][Q], double (*s_psi)[68],
    double (*s_qx)[68], double (*s_qy)[68], double* __restrict__ pn_r, double* __restrict__ pn_b,
    const double* __restrict__ in_r, const double* __restrict__ in_b, const Geom& g, const CgFast& cf,
    const MacroIdx& mi, int i, int n_iter, int R0, int lane, int cl, int c, bool lane_out,
    double* __restrict__ rho_r_out, double* __restrict__ rho_b_out, double* __restrict__ u_out,
    double* __restrict__ psi_out, double* __restrict__ snu_out) {
#pragma clang fp contract(on)
  if (i >= n_iter) return;  // wave-uniform
  // raw populations of this iteration's macroscopic row R0 - 2 + i arrived in buffer K; the buffer the
  // previous iteration consumed ((K + 2) % 3) takes the row TWO ahead: two rows of loads stay in flight
  constexpr int KN = (K + 2) % 3;
  if (i + 2 < n_iter) {
    const long o = g.at(R0 + i, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[KN][q] = in_r[off];
      raw_b[KN][q] = in_b[off];
    }
  }
  double ft[Q];
  const double (&fr)[Q] = raw_r[K];
#pragma unroll
  for (int q = 0; q < Q; ++q) ft[q] = raw_b[K][q];
  // reduce the arrived row (== cg_node<true>)
  const double rr = (((fr[0] + fr[1]) + (fr[2] + fr[3])) + ((fr[4] + fr[5]) + (fr[6] + fr[7]))) + fr[8];
  const double rb = (((ft[0] + ft[1]) + (ft[2] + ft[3])) + ((ft[4] + ft[5]) + (ft[6] + ft[7]))) + ft[8];
#pragma unroll
  for (int q = 0; q < Q; ++q) ft[q] += fr[q];
  const double jx = ((ft[1] - ft[3]) + (ft[5] - ft[6])) + (ft[8] - ft[7]);
  const double jy = ((ft[2] - ft[4]) + (ft[5] - ft[8])) + (ft[6] - ft[7]);
  const double irt = 1.0 / (rr + rb);
  const double ux = (jx + 0.5 * cf.Gr) * irt, uy = (jy + 0.5 * cf.Gc) * irt;
  const double a = rr * cf.inv_rho0[0], b = rb * cf.inv_rho0[1];
  const double psi = (a - b) / (a + b);
  const double qcs = cf.qc[0] * rr + cf.qc[1] * rb;
  const int slot = i % 5;
  s_psi[slot][lane + 2] = psi;
  s_qx[slot][lane + 2] = qcs * ux;
  s_qy[slot][lane + 2] = qcs * uy;
#pragma unroll
  for (int q = 0; q < Q; ++q) rf[K][q] = ft[q];
  rn[K][0] = rr; rn[K][1] = rb; rn[K][2] = ux; rn[K][3] = uy; rn[K][4] = irt; rn[K][5] = psi;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (i < 4 || !lane_out) return;
  // collide row r = R0 + i - 4: reduced two iterations ago (ring slot K + 1 mod 3)
  constexpr int KC = (K + 1) % 3;
  const int r = R0 + i - 4;
  const int s0 = (i - 4) % 5, s1 = (i - 3) % 5, s3 = (i - 1) % 5, s4 = i % 5;
  constexpr double k = 1.0 / 5040.0;
  constexpr double a0[5] = {2 * k * 1, 2 * k * 32, 2 * k * 84, 2 * k * 32, 2 * k * 1};
  constexpr double a1[5] = {k * 32, k * 448, k * 960, k * 448, k * 32};
  // one wave per SIMD and registers to spare: the 80 ring reads of the four stencils are issued together
  // (unrolled), the accumulation order stays that of cg_ddrow / cg_ddcol
  double gx = 0.0, dxqx = 0.0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {  // == cg_ddrow
    gx += a0[j] * (s_psi[s4][lane + j] - s_psi[s0][lane + j]);
    gx += a1[j] * (s_psi[s3][lane + j] - s_psi[s1][lane + j]);
    dxqx += a0[j] * (s_qx[s4][lane + j] - s_qx[s0][lane + j]);
    dxqx += a1[j] * (s_qx[s3][lane + j] - s_qx[s1][lane + j]);
  }
  double gy = 0.0, dyqy = 0.0;
#pragma unroll
  for (int ii = 0; ii < 5; ++ii) {  // == cg_ddcol
    const int sl = (i - 4 + ii) % 5;
    gy += a0[ii] * (s_psi[sl][lane + 4] - s_psi[sl][lane]);
    gy += a1[ii] * (s_psi[sl][lane + 3] - s_psi[sl][lane + 1]);
    dyqy += a0[ii] * (s_qy[sl][lane + 4] - s_qy[sl][lane]);
    dyqy += a1[ii] * (s_qy[sl][lane + 3] - s_qy[sl][lane + 1]);
  }
  double fc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) fc[q] = rf[KC][q];
  CgNode me;
  me.rr = rn[KC][0]; me.rb = rn[KC][1]; me.ux = rn[KC][2]; me.uy = rn[KC][3]; me.irt = rn[KC][4]; me.psi = rn[KC][5];
  me.qx = 0.0; me.qy = 0.0;
  cg_collide_store<WITH_FIELDS>(fc, me, gx, gy, dxqx, dyqy, cf, g, mi, r, c, pn_r, pn_b, rho_r_out, rho_b_out,
                                u_out, psi_out, snu_out);
}

// one wave per SIMD (~350 VGPRs): budgeted for two, the kernel spilled 16 registers into scratch memory and
// lost more to their reloads (12.7 k MLUPS) than the second wave hid
template <int WAVES, bool WITH_FIELDS>
__global__ __launch_bounds__(64 * WAVES, 1) void k_cg_strip2(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int col_begin, int col_end,
    int rows_per_chunk, int strips, int n_waves) {
  __shared__ double ring[WAVES][3][5][68];  // [wave][field][slot][2 pad + lane + 2 pad]
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wave = blockIdx.x * WAVES + wib;
  if (wave >= n_waves) return;
  const int strip = wave % strips, chunk = wave / strips;
  const int R0 = row_begin + chunk * rows_per_chunk;
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int c = col_begin + strip * CG_SW2 - 4 + lane;  // this lane's column
  const bool lane_out = lane >= 4 && lane < 4 + CG_SW2 && c < col_end;
  // loads stay inside the rectangle's ring (its +-1 gathers are in bounds by construction)
  const int cl = c < col_begin - 2 ? col_begin - 2 : (c > col_end + 1 ? col_end + 1 : c);
  double(*s_psi)[68] = ring[wib][0];
  double(*s_qx)[68] = ring[wib][1];
  double(*s_qy)[68] = ring[wib][2];
  double rf[3][Q], rn[3][6], raw_r[3][Q], raw_b[3][Q];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int q = 0; q < Q; ++q) rf[a][q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) rn[a][q] = 1.0;
  }
  const int n_iter = (R1 - R0) + 4;
#pragma unroll
  for (int a = 0; a < 2; ++a) {  // the first two macroscopic rows: R0 - 2, R0 - 1 (n_iter >= 5)
    const long o = g.at(R0 - 2 + a, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[a][q] = in_r[off];
      raw_b[a][q] = in_b[off];
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) raw_r[2][q] = raw_b[2][q] = 0.0;
  for (int i = 0; i < n_iter; i += 3) {
    cg_strip2_iter<0, WITH_FIELDS>(rf, rn, raw_r, raw_b, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i, n_iter, R0, lane, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
    cg_strip2_iter<1, WITH_FIELDS>(rf, rn, raw_r, raw_b, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i + 1, n_iter, R0, lane, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
    cg_strip2_iter<2, WITH_FIELDS>(rf, rn, raw_r, raw_b, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i + 2, n_iter, R0, lane, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
  }
}


// ---- third generation: two waves per SIMD -----------------------------------------------------------------
// k_cg_strip2 needs ~390 registers (one wave per SIMD: every dependent chain of the collision is exposed).
// Here the colour-summed populations of the 3 ring rows wait in wave-private LDS instead (13.8 KB per wave;
// the six macroscopic values per row stay in registers), and the raw populations of the next row are loaded
// into the registers the reduction has just freed: ~210 VGPRs, two waves per SIMD, 22 KB of LDS per wave.
template <int K, bool WITH_FIELDS>
__device__ __forceinline__ void cg_strip3_iter(
    double (&rn)[3][6], double (&raw_r)[Q], double (&raw_b)[Q], double (*s_ft)[Q][64], double (*s_psi)[68],
    double (*s_qx)[68], double (*s_qy)[68], double* __restrict__ pn_r, double* __restrict__ pn_b,
    const double* __restrict__ in_r, const double* __restrict__ in_b, const Geom& g, const CgFast& cf,
    const MacroIdx& mi, int i, int n_iter, int R0, int lane, int cl, int c, bool lane_out,
    double* __restrict__ rho_r_out, double* __restrict__ rho_b_out, double* __restrict__ u_out,
    double* __restrict__ psi_out, double* __restrict__ snu_out) {
#pragma clang fp contract(on)
  if (i >= n_iter) return;  // wave-uniform
  {
    // reduce the arrived row R0 - 2 + i (== cg_node<true>)
    const double (&fr)[Q] = raw_r;
    double ft[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] = raw_b[q];
    const double rr = (((fr[0] + fr[1]) + (fr[2] + fr[3])) + ((fr[4] + fr[5]) + (fr[6] + fr[7]))) + fr[8];
    const double rb = (((ft[0] + ft[1]) + (ft[2] + ft[3])) + ((ft[4] + ft[5]) + (ft[6] + ft[7]))) + ft[8];
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] += fr[q];
#pragma unroll
    for (int q = 0; q < Q; ++q) s_ft[K][q][lane] = ft[q];
    const double jx = ((ft[1] - ft[3]) + (ft[5] - ft[6])) + (ft[8] - ft[7]);
    const double jy = ((ft[2] - ft[4]) + (ft[5] - ft[8])) + (ft[6] - ft[7]);
    const double irt = 1.0 / (rr + rb);
    const double ux = (jx + 0.5 * cf.Gr) * irt, uy = (jy + 0.5 * cf.Gc) * irt;
    const double a = rr * cf.inv_rho0[0], b = rb * cf.inv_rho0[1];
    const double psi = (a - b) / (a + b);
    const double qcs = cf.qc[0] * rr + cf.qc[1] * rb;
    const int slot = i % 5;
    s_psi[slot][lane + 2] = psi;
    s_qx[slot][lane + 2] = qcs * ux;
    s_qy[slot][lane + 2] = qcs * uy;
    rn[K][0] = rr; rn[K][1] = rb; rn[K][2] = ux; rn[K][3] = uy; rn[K][4] = irt; rn[K][5] = psi;
  }
  if (i + 1 < n_iter) {  // the next row into the registers just freed; in flight during the collision below
    const long o = g.at(R0 - 1 + i, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[q] = in_r[off];
      raw_b[q] = in_b[off];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (i < 4 || !lane_out) return;
  constexpr int KC = (K + 1) % 3;
  const int r = R0 + i - 4;
  const int s0 = (i - 4) % 5, s1 = (i - 3) % 5, s3 = (i - 1) % 5, s4 = i % 5;
  constexpr double k = 1.0 / 5040.0;
  constexpr double a0[5] = {2 * k * 1, 2 * k * 32, 2 * k * 84, 2 * k * 32, 2 * k * 1};
  constexpr double a1[5] = {k * 32, k * 448, k * 960, k * 448, k * 32};
  double gx = 0.0, dxqx = 0.0;
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {  // == cg_ddrow
    gx += a0[j] * (s_psi[s4][lane + j] - s_psi[s0][lane + j]);
    gx += a1[j] * (s_psi[s3][lane + j] - s_psi[s1][lane + j]);
    dxqx += a0[j] * (s_qx[s4][lane + j] - s_qx[s0][lane + j]);
    dxqx += a1[j] * (s_qx[s3][lane + j] - s_qx[s1][lane + j]);
  }
  double gy = 0.0, dyqy = 0.0;
#pragma unroll 1
  for (int ii = 0; ii < 5; ++ii) {  // == cg_ddcol
    const int sl = (i - 4 + ii) % 5;
    gy += a0[ii] * (s_psi[sl][lane + 4] - s_psi[sl][lane]);
    gy += a1[ii] * (s_psi[sl][lane + 3] - s_psi[sl][lane + 1]);
    dyqy += a0[ii] * (s_qy[sl][lane + 4] - s_qy[sl][lane]);
    dyqy += a1[ii] * (s_qy[sl][lane + 3] - s_qy[sl][lane + 1]);
  }
  double fc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) fc[q] = s_ft[KC][q][lane];
  CgNode me;
  me.rr = rn[KC][0]; me.rb = rn[KC][1]; me.ux = rn[KC][2]; me.uy = rn[KC][3]; me.irt = rn[KC][4]; me.psi = rn[KC][5];
  me.qx = 0.0; me.qy = 0.0;
  cg_collide_store<WITH_FIELDS>(fc, me, gx, gy, dxqx, dyqy, cf, g, mi, r, c, pn_r, pn_b, rho_r_out, rho_b_out,
                                u_out, psi_out, snu_out);
}

template <int WAVES, bool WITH_FIELDS>
__global__ __launch_bounds__(64 * WAVES, 2) void k_cg_strip3(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int col_begin, int col_end,
    int rows_per_chunk, int strips, int n_waves, int xcd_order) {
  __shared__ double ring[WAVES][3][5][68];  // [wave][field][slot][2 pad + lane + 2 pad]
  __shared__ double ftr[WAVES][3][Q][64];   // [wave][ring row][population][lane]
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // workgroups are dealt round-robin over the 8 XCDs: with xcd_order, XCD k takes the k-th contiguous eighth of
  // the (chunk-major, strip-minor) sequence, so that neighbouring strips -- which share the 128-byte lines at
  // their window edges -- run back to back on one L2 instead of on eight different ones
  int blk = blockIdx.x;
  if (xcd_order) blk = (blk % 8) * ((int)gridDim.x / 8) + blk / 8;  // the launch pads the grid to a multiple of 8
  const int wave = blk * WAVES + wib;
  if (wave >= n_waves) return;
  const int strip = wave % strips, chunk = wave / strips;
  const int R0 = row_begin + chunk * rows_per_chunk;
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int c = col_begin + strip * CG_SW2 - 4 + lane;
  const bool lane_out = lane >= 4 && lane < 4 + CG_SW2 && c < col_end;
  const int cl = c < col_begin - 2 ? col_begin - 2 : (c > col_end + 1 ? col_end + 1 : c);
  double(*s_psi)[68] = ring[wib][0];
  double(*s_qx)[68] = ring[wib][1];
  double(*s_qy)[68] = ring[wib][2];
  double(*s_ft)[Q][64] = ftr[wib];
  double rn[3][6], raw_r[Q], raw_b[Q];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int q = 0; q < 6; ++q) rn[a][q] = 1.0;
  {
    const long o = g.at(R0 - 2, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[q] = in_r[off];
      raw_b[q] = in_b[off];
    }
  }
  const int n_iter = (R1 - R0) + 4;
  for (int i = 0; i < n_iter; i += 3) {
    cg_strip3_iter<0, WITH_FIELDS>(rn, raw_r, raw_b, s_ft, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i, n_iter, R0, lane, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
    cg_strip3_iter<1, WITH_FIELDS>(rn, raw_r, raw_b, s_ft, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i + 1, n_iter, R0, lane, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
    cg_strip3_iter<2, WITH_FIELDS>(rn, raw_r, raw_b, s_ft, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i + 2, n_iter, R0, lane, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
  }
}

// ---- fourth generation: a WORKGROUP of W waves walks down a 64 W-column window in lockstep ------------------------
// What the round-3 calibration (scripts/calib/fetch_calib.hip, profiles/r03_fetch_calib.txt) showed: the L2 fetches whole
// 128-BYTE LINES, one request per line a wave's load touches, and neighbouring waves share a line only when they sit in
// one workgroup at the same time.  A private 64-column window with 56 outputs that starts 32 bytes off a line boundary
// (k_cg_strip2 / 3) therefore pays 5 lines for 3.5 lines of output on every one of its 18 streams -- the 1.38x read
// amplification PMC measured.  Here
//   * the block's window starts on a line boundary (a multiple of 16 columns) and is 64 W columns wide; only its first and
//     last 8 lanes are ring-only, so a block reads 4 W lines per row and stream for 4 W - 1 lines of output (W = 4: 1.067x),
//     and the +-1-column pulls of a wave land in lines its neighbours in the block load in the same iteration;
//   * psi, Qx, Qy of a row go into ONE ring shared by the block (6 slots: the slot a fast wave writes next is never one a
//     slow wave still reads), so all lanes but the 16 at the block's edges produce output -- one workgroup barrier per row;
//   * the colour-summed populations wait in a wave-private ring of TWO rows (the row just reduced stays in registers until
//     the row two behind it has been collided out of the slot it takes): 18.4 KB of LDS per wave, 8 waves per CU.
// Per-node arithmetic = the tile kernel's: identical bits.
constexpr int CG_S4_EDGE = 8;

template <int W, int K, bool WITH_FIELDS>
__device__ __forceinline__ void cg_strip4_iter(
    double (&rn)[3][6], double (&raw_r)[Q], double (&raw_b)[Q], double (*s_ft)[Q][64], double (*s_psi)[64 * W],
    double (*s_qx)[64 * W], double (*s_qy)[64 * W], double* __restrict__ pn_r, double* __restrict__ pn_b,
    const double* __restrict__ in_r, const double* __restrict__ in_b, const Geom& g, const CgFast& cf,
    const MacroIdx& mi, int i, int n_iter, int R0, int lane, int gl, int cl, int c, bool lane_out,
    double* __restrict__ rho_r_out, double* __restrict__ rho_b_out, double* __restrict__ u_out,
    double* __restrict__ psi_out, double* __restrict__ snu_out) {
#pragma clang fp contract(on)
  if (i >= n_iter) return;  // uniform over the block: all its waves walk the same chunk
  double ft[Q];
  {
    // reduce the arrived row R0 - 2 + i (== cg_node<true>)
    const double (&fr)[Q] = raw_r;
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] = raw_b[q];
    const double rr = (((fr[0] + fr[1]) + (fr[2] + fr[3])) + ((fr[4] + fr[5]) + (fr[6] + fr[7]))) + fr[8];
    const double rb = (((ft[0] + ft[1]) + (ft[2] + ft[3])) + ((ft[4] + ft[5]) + (ft[6] + ft[7]))) + ft[8];
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] += fr[q];
    const double jx = ((ft[1] - ft[3]) + (ft[5] - ft[6])) + (ft[8] - ft[7]);
    const double jy = ((ft[2] - ft[4]) + (ft[5] - ft[8])) + (ft[6] - ft[7]);
    const double irt = 1.0 / (rr + rb);
    const double ux = (jx + 0.5 * cf.Gr) * irt, uy = (jy + 0.5 * cf.Gc) * irt;
    const double a = rr * cf.inv_rho0[0], b = rb * cf.inv_rho0[1];
    const double psi = (a - b) / (a + b);
    const double qcs = cf.qc[0] * rr + cf.qc[1] * rb;
    const int slot = i % 6;
    s_psi[slot][gl] = psi;
    s_qx[slot][gl] = qcs * ux;
    s_qy[slot][gl] = qcs * uy;
    rn[K][0] = rr; rn[K][1] = rb; rn[K][2] = ux; rn[K][3] = uy; rn[K][4] = irt; rn[K][5] = psi;
  }
  if (i + 1 < n_iter) {  // the next row into the registers just freed; in flight during the collision below
    const long o = g.at(R0 - 1 + i, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[q] = in_r[off];
      raw_b[q] = in_b[off];
    }
  }
  __syncthreads();  // the row's psi, Qx, Qy of every wave of the block are in the ring
  if (i >= 4 && lane_out) {
    constexpr int KC = (K + 1) % 3;
    const int r = R0 + i - 4;
    const int s0 = (i - 4) % 6, s1 = (i - 3) % 6, s3 = (i - 1) % 6, s4 = i % 6;
    constexpr double k = 1.0 / 5040.0;
    constexpr double a0[5] = {2 * k * 1, 2 * k * 32, 2 * k * 84, 2 * k * 32, 2 * k * 1};
    constexpr double a1[5] = {k * 32, k * 448, k * 960, k * 448, k * 32};
    const int l0 = gl - 2;  // columns c - 2 .. c + 2 sit at [l0 .. l0 + 4]
    double gx = 0.0, dxqx = 0.0;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {  // == cg_ddrow
      gx += a0[j] * (s_psi[s4][l0 + j] - s_psi[s0][l0 + j]);
      gx += a1[j] * (s_psi[s3][l0 + j] - s_psi[s1][l0 + j]);
      dxqx += a0[j] * (s_qx[s4][l0 + j] - s_qx[s0][l0 + j]);
      dxqx += a1[j] * (s_qx[s3][l0 + j] - s_qx[s1][l0 + j]);
    }
    double gy = 0.0, dyqy = 0.0;
#pragma unroll 1
    for (int ii = 0; ii < 5; ++ii) {  // == cg_ddcol
      const int sl = (i - 4 + ii) % 6;
      gy += a0[ii] * (s_psi[sl][l0 + 4] - s_psi[sl][l0]);
      gy += a1[ii] * (s_psi[sl][l0 + 3] - s_psi[sl][l0 + 1]);
      dyqy += a0[ii] * (s_qy[sl][l0 + 4] - s_qy[sl][l0]);
      dyqy += a1[ii] * (s_qy[sl][l0 + 3] - s_qy[sl][l0 + 1]);
    }
    double fc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) fc[q] = s_ft[i & 1][q][lane];  // reduced two iterations ago
    CgNode me;
    me.rr = rn[KC][0]; me.rb = rn[KC][1]; me.ux = rn[KC][2]; me.uy = rn[KC][3]; me.irt = rn[KC][4]; me.psi = rn[KC][5];
    me.qx = 0.0; me.qy = 0.0;
    cg_collide_store<WITH_FIELDS>(fc, me, gx, gy, dxqx, dyqy, cf, g, mi, r, c, pn_r, pn_b, rho_r_out, rho_b_out,
                                  u_out, psi_out, snu_out);
  }
  // ... and only now does this iteration's row take that slot (same wave, same lanes: program order suffices)
#pragma unroll
  for (int q = 0; q < Q; ++q) s_ft[i & 1][q][lane] = ft[q];
}

template <int W, bool WITH_FIELDS>
__global__ __launch_bounds__(64 * W, 2) void k_cg_strip4(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int col_begin, int col_end,
    int rows_per_chunk, int bstrips, int win0) {
  __shared__ double ring[3][6][64 * W];  // [field][slot][block lane]
  __shared__ double ftr[W][2][Q][64];    // [wave][ring row][population][lane]
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, gl = threadIdx.x;
  constexpr int S = 64 * W - 2 * CG_S4_EDGE;  // output columns per block (a multiple of 16: windows stay line-aligned)
  const int bs = blockIdx.x % bstrips, chunk = blockIdx.x / bstrips;
  const int R0 = row_begin + chunk * rows_per_chunk;
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int c = win0 + bs * S + gl;  // this lane's column
  const bool lane_out = gl >= CG_S4_EDGE && gl < 64 * W - CG_S4_EDGE && c >= col_begin && c < col_end;
  // loads stay inside the lattice (lanes beyond the rectangle's ring feed nothing that is stored)
  const int cl = c < 1 ? 1 : (c > g.C - 2 ? g.C - 2 : c);
  double(*s_psi)[64 * W] = ring[0];
  double(*s_qx)[64 * W] = ring[1];
  double(*s_qy)[64 * W] = ring[2];
  double(*s_ft)[Q][64] = ftr[wib];
  double rn[3][6], raw_r[Q], raw_b[Q];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int q = 0; q < 6; ++q) rn[a][q] = 1.0;
  {
    const long o = g.at(R0 - 2, cl);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const long off = q * g.plane + (o - icx(q) * g.C - icy(q));
      raw_r[q] = in_r[off];
      raw_b[q] = in_b[off];
    }
  }
  const int n_iter = (R1 - R0) + 4;
  for (int i = 0; i < n_iter; i += 3) {
    cg_strip4_iter<W, 0, WITH_FIELDS>(rn, raw_r, raw_b, s_ft, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i, n_iter, R0, lane, gl, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
    cg_strip4_iter<W, 1, WITH_FIELDS>(rn, raw_r, raw_b, s_ft, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i + 1, n_iter, R0, lane, gl, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
    cg_strip4_iter<W, 2, WITH_FIELDS>(rn, raw_r, raw_b, s_ft, s_psi, s_qx, s_qy, pn_r, pn_b, in_r, in_b, g, cf, mi, i + 2, n_iter, R0, lane, gl, cl, c, lane_out, rho_r_out, rho_b_out, u_out, psi_out, snu_out);
  }
}

