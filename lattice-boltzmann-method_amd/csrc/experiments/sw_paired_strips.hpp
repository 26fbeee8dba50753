// EXPERIMENTS build only (make -C lattice-boltzmann-method_amd/csrc EXPERIMENTS=1): sliding window on paired strips (k_stream_collide_swp)
// Measured and not kept -- DESIGN.md 4.2 / 9 hold the numbers.  Included from d2q9.hpp at the place the code used to stand;
// not a stand-alone header (it uses what that file has declared above the include).
// measured and not kept (DESIGN.md "experiments"): compiled only with make EXPERIMENTS=1
// ---- paired strips: the waves of a workgroup own ADJACENT 64-column windows and hand each other the edge
// columns of every level through LDS, so only the outer 2 (D - 1) columns of the GROUP are redundant: 120 of
// 128 lanes (2 waves) or 248 of 256 (4 waves) produce output instead of 56 of 64, and a group's rows are read
// as one 1 - 2 KB run instead of 512-byte pieces whose 128-byte lines neighbouring strips fetch again.
// Schedule: level l lags level l-1 by TWO rows (level l at iteration i computes row rbase + i - 2 (l - 1)), so
// everything a level pulls from the level below -- rows r-1, r, r+1 -- was computed in EARLIER iterations: the
// edge values published in iteration i are first read in iteration i + 1 and ONE workgroup barrier per
// iteration orders them (with a one-row lag the c_x = -1 populations would come from the row computed in the
// same iteration: a barrier per level).  Price: a 4-row register ring per level (288 VGPRs at D = 5; one wave
// per SIMD has them) and D - 1 more pipeline iterations per chunk.  A lane without a source lane in a DPP
// wave shift keeps the `old` operand: that operand is the neighbour wave's edge value read from LDS, so the
// hand-off costs no select.  Same arithmetic per node as every other path: identical bits.
template <int D>
struct SwpExch {
  double v[D - 1][4][2][3];  // [level produced][ring slot][0: a wave's lane 0 (q = 4,7,8) / 1: its lane 63 (q = 2,5,6)][j]
};
__device__ __forceinline__ double lane_from_prev_fill(double v, double fill) {  // lane i <- lane i-1; lane 0 <- fill
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_next_fill(double v, double fill) {  // lane i <- lane i+1; lane 63 <- fill
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__host__ __device__ constexpr int swp_q_side0(int j) { return j == 0 ? 4 : (j == 1 ? 7 : 8); }  // icy = -1: wanted by the wave on the left
__host__ __device__ constexpr int swp_q_side1(int j) { return j == 0 ? 2 : (j == 1 ? 5 : 6); }  // icy = +1: wanted by the wave on the right
__host__ __device__ constexpr int swp_j_of(int q) { return (q == 4 || q == 2) ? 0 : ((q == 7 || q == 5) ? 1 : 2); }

template <class Model, int D, int WAVES, int K4, bool NT_STORE>
__device__ __forceinline__ void swp_iteration(double (&ring)[D - 1][4][Q], double (&cur)[Q], SwpExch<D>* ex,
                                              double* __restrict__ pn, const double* __restrict__ po, const Geom& g,
                                              const Model& m, int i, int n_iter, int rbase, int R0, int R1,
                                              const int (&cols)[3], bool lane_ok, int c_out, int w, int lane,
                                              int last_row_needed) {
  if (i >= n_iter) return;  // workgroup-uniform (all waves of a group share the chunk)
  __syncthreads();          // the edge values published in iteration i - 1 are visible; those read then are consumed
  // ---- the neighbours' edge values every level of this iteration pulls: all from EARLIER iterations (ring slots
  // K4+1 .. K4+3), read in one batch before this iteration publishes anything (slot K4) ---------------------------
  double e[D - 1][6];  // [level l - 2][0..2: q = 2,5,6 from the left wave's lane 63 | 3..5: q = 4,7,8 from the right wave's lane 0]
#pragma unroll
  for (int l = 2; l <= D; ++l) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q1 = swp_q_side1(j), q0 = swp_q_side0(j);
      const int s1 = icx(q1) == -1 ? (K4 + 3) % 4 : (icx(q1) == 0 ? (K4 + 2) % 4 : (K4 + 1) % 4);
      const int s0 = icx(q0) == -1 ? (K4 + 3) % 4 : (icx(q0) == 0 ? (K4 + 2) % 4 : (K4 + 1) % 4);
      e[l - 2][j] = ex[w > 0 ? w - 1 : 0].v[l - 2][s1][1][j];
      e[l - 2][3 + j] = ex[w < WAVES - 1 ? w + 1 : WAVES - 1].v[l - 2][s0][0][j];
    }
  }
  // ---- prefetch level-1 inputs of the NEXT iteration ------------------------------------------------------------
  double nxt[Q];
  {
    const int r1n = rbase + i + 1;
    int rr[3] = {r1n + 1, r1n, r1n - 1};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (g.ghost) rr[k] = rr[k] < -g.ghost ? -g.ghost : (rr[k] > g.R + g.ghost - 1 ? g.R + g.ghost - 1 : rr[k]);
      else {
        rr[k] = rr[k] < 0 ? rr[k] + g.R : (rr[k] >= g.R ? rr[k] - g.R : rr[k]);
        rr[k] = rr[k] >= g.R ? rr[k] - g.R : rr[k];
      }
    }
    if (r1n <= last_row_needed) {
#pragma unroll
      for (int q = 0; q < Q; ++q) nxt[q] = po[q * g.plane + g.at(rr[icx(q) + 1], 0) + cols[icy(q) + 1]];
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q) nxt[q] = 1.0;
    }
  }
  double f[Q], rho, ux, uy;
#pragma unroll
  for (int q = 0; q < Q; ++q) f[q] = cur[q];
  m.collide(f, rho, ux, uy);
#pragma unroll
  for (int l = 2; l <= D; ++l) {
    // publish level l-1's row of this iteration: ring slot K4; its edge lanes go to the neighbours
#pragma unroll
    for (int q = 0; q < Q; ++q) ring[l - 2][K4][q] = f[q];
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) ex[w].v[l - 2][K4][0][j] = f[swp_q_side0(j)];
    }
    if (lane == 63) {
#pragma unroll
      for (int j = 0; j < 3; ++j) ex[w].v[l - 2][K4][1][j] = f[swp_q_side1(j)];
    }
    // gather level l's row r = rbase + i - 2 (l - 1) from level l-1's rows r+1, r, r-1 = ring slots K4+3, K4+2, K4+1
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int slot = icx(q) == -1 ? (K4 + 3) % 4 : (icx(q) == 0 ? (K4 + 2) % 4 : (K4 + 1) % 4);
      double v = ring[l - 2][slot][q];
      if (icy(q) == 1) v = lane_from_prev_fill(v, w > 0 ? e[l - 2][swp_j_of(q)] : v);             // from column c-1
      else if (icy(q) == -1) v = lane_from_next_fill(v, w < WAVES - 1 ? e[l - 2][3 + swp_j_of(q)] : v);  // from column c+1
      f[q] = v;
    }
    m.collide(f, rho, ux, uy);
  }
  const int rD = rbase + i - 2 * (D - 1);
  if (lane_ok && rD >= R0 && rD < R1) {
    const long o = g.at(rD, c_out);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double* dst = pn + q * g.plane + o;
      if (NT_STORE) __builtin_nontemporal_store(f[q], dst);
      else *dst = f[q];
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) cur[q] = nxt[q];
}

__host__ __device__ constexpr int swp_group_width(int D, int WAVES) { return 64 * WAVES - 2 * (D - 1); }

template <class Model, int D, int WAVES, bool NT_STORE>
__global__ __launch_bounds__(64 * WAVES, 1) void k_stream_collide_swp(
    double* __restrict__ pn, const double* __restrict__ po, Geom g, Model m, int row_begin, int row_end,
    int rows_per_chunk, int groups, int n_groups_total, int chunk_stride) {
  constexpr int GW = swp_group_width(D, WAVES);
  __shared__ SwpExch<D> ex[WAVES];
  const int grp = blockIdx.x;  // one workgroup = one group of adjacent windows on one chunk of rows
  if (grp >= n_groups_total) return;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int gs = grp % groups, chunk = grp / groups;
  const int R0 = row_begin + chunk * (chunk_stride > 0 ? chunk_stride : rows_per_chunk);
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int gl = 64 * w + lane;                      // lane index inside the group
  int c = gs * GW - (D - 1) + gl;                    // this lane's column at every level
  const bool lane_ok = gl >= D - 1 && gl < D - 1 + GW && c < g.C;
  c = c < 0 ? c + g.C : (c >= g.C ? c - g.C : c);
  c = c >= g.C ? c - g.C : c;                        // the last group may run more than one period past the edge
  const int cols[3] = {wrap_col(g, c + 1), c, wrap_col(g, c - 1)};
  const int rbase = R0 - (D - 1);
  const int n_iter = (R1 - R0) + 3 * (D - 1);
  const int last_row_needed = R1 - 1 + (D - 1);      // level-1 rows beyond it feed nothing that is stored
  double ring[D - 1][4][Q];
#pragma unroll
  for (int a = 0; a < D - 1; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < Q; ++q) ring[a][b][q] = 1.0;  // warm-up garbage, never stored
  if (lane < 3) {  // the exchange slots the first iterations read before anything was published
#pragma unroll
    for (int a = 0; a < D - 1; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        ex[w].v[a][b][0][lane] = 1.0;
        ex[w].v[a][b][1][lane] = 1.0;
      }
  }
  double cur[Q];
  {  // level-1 inputs of iteration 0
    int rr[3] = {rbase + 1, rbase, rbase - 1};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (g.ghost) rr[k] = rr[k] < -g.ghost ? -g.ghost : (rr[k] > g.R + g.ghost - 1 ? g.R + g.ghost - 1 : rr[k]);
      else rr[k] = rr[k] < 0 ? rr[k] + g.R : (rr[k] >= g.R ? rr[k] - g.R : rr[k]);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) cur[q] = po[q * g.plane + g.at(rr[icx(q) + 1], 0) + cols[icy(q) + 1]];
  }
  for (int i = 0; i < n_iter; i += 4) {  // unrolled by the 4 ring slots: every register index is static
    swp_iteration<Model, D, WAVES, 0, NT_STORE>(ring, cur, ex, pn, po, g, m, i, n_iter, rbase, R0, R1, cols, lane_ok, c, w, lane, last_row_needed);
    swp_iteration<Model, D, WAVES, 1, NT_STORE>(ring, cur, ex, pn, po, g, m, i + 1, n_iter, rbase, R0, R1, cols, lane_ok, c, w, lane, last_row_needed);
    swp_iteration<Model, D, WAVES, 2, NT_STORE>(ring, cur, ex, pn, po, g, m, i + 2, n_iter, rbase, R0, R1, cols, lane_ok, c, w, lane, last_row_needed);
    swp_iteration<Model, D, WAVES, 3, NT_STORE>(ring, cur, ex, pn, po, g, m, i + 3, n_iter, rbase, R0, R1, cols, lane_ok, c, w, lane, last_row_needed);
  }
}

