// EXPERIMENTS build only (make -C lattice-boltzmann-method_amd/csrc EXPERIMENTS=1): two-phase step, strip kernel of generation 5 (adjacent private windows)
// Measured and not kept -- DESIGN.md 4.2 / 9 hold the numbers.  Included from cg_fused.hpp at the place the code used to stand;
// not a stand-alone header (it uses what that file has declared above the include).
// 14.0-14.3 k against the tile kernel's 15.3 k
// ---- fifth form: k_cg_strip3's private windows, FOUR adjacent strips per workgroup kept loosely together -------------------
// The calibration of round 3 says neighbouring strips share a 128-byte line only inside one workgroup at about the same
// time; k_cg_strip4 buys that with a shared ring and a barrier per row and loses more to the lockstep than it gains.  Here
// every wave keeps its own 64-column window, rings and pace (no data passes between waves), the W waves of a workgroup own
// ADJACENT strips of one chunk, and a workgroup barrier every `sync_every` rows only bounds how far they drift apart -- the
// lines at the window edges are then mostly L2 hits.  Colour sums in a two-row ring (as k_cg_strip4): 17.4 KB of LDS per
// wave, 8 waves per CU.  Per-node arithmetic = the tile kernel's: identical bits.
template <int W, bool WITH_FIELDS>
__global__ __launch_bounds__(64 * W, 2) void k_cg_strip5(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int col_begin, int col_end,
    int rows_per_chunk, int groups, int sync_every) {
#pragma clang fp contract(on)
  __shared__ double ring[W][3][5][68];  // [wave][field][slot][2 pad + lane + 2 pad]
  __shared__ double ftr[W][2][Q][64];   // [wave][ring row][population][lane]
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int grp = blockIdx.x % groups, chunk = blockIdx.x / groups;
  const int strip = grp * W + wib;
  const int R0 = row_begin + chunk * rows_per_chunk;
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int c = col_begin + strip * CG_SW2 - 4 + lane;
  const bool lane_out = lane >= 4 && lane < 4 + CG_SW2 && c < col_end;
  const int cl = c < 1 ? 1 : (c > g.C - 2 ? g.C - 2 : c);
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
  for (int i = 0; i < n_iter; ++i) {
    if (sync_every > 0 && i % sync_every == 0) __syncthreads();  // uniform: all waves of a block walk the same chunk
#pragma unroll
    for (int q = 0; q < 6; ++q) rn[2][q] = rn[1][q], rn[1][q] = rn[0][q];
    double ft[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] = raw_b[q];
    double psi, qx, qy;
    cg_reduce_row(raw_r, ft, cf, rn[0], psi, qx, qy);
    const int slot = i % 5;
    s_psi[slot][lane + 2] = psi;
    s_qx[slot][lane + 2] = qx;
    s_qy[slot][lane + 2] = qy;
    if (i + 1 < n_iter) {
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
    if (i >= 4 && lane_out) {
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
      for (int q = 0; q < Q; ++q) fc[q] = s_ft[i & 1][q][lane];  // reduced two iterations ago
      CgNode me;
      me.rr = rn[2][0]; me.rb = rn[2][1]; me.ux = rn[2][2]; me.uy = rn[2][3]; me.irt = rn[2][4]; me.psi = rn[2][5];
      me.qx = 0.0; me.qy = 0.0;
      cg_collide_store<WITH_FIELDS>(fc, me, gx, gy, dxqx, dyqy, cf, g, mi, r, c, pn_r, pn_b, rho_r_out, rho_b_out,
                                    u_out, psi_out, snu_out);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) s_ft[i & 1][q][lane] = ft[q];
  }
}

