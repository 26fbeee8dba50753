// EXPERIMENT (round 4): the register sliding window with TWO columns per lane.  Included by d2q9.hpp in the LBM_EXPERIMENTS build only
// (tuning "sw_cols2" = 1, BGK reassociated collision, periodic / halo edges, 5 steps per launch).
// A wave covers 128 columns instead of 64, so the 2 (D - 1) halo columns it recomputes are 6 % of its work instead of 12.5 % (fewer
// redundant f64 operations, fewer halo bytes), a lane's two nodes hand each other one of the two +-1-column neighbours without a lane
// shift (half the DPP moves per node), c_y = 0 populations are loaded and everything is stored 16 bytes per lane.  Same arithmetic
// per node as sw_iteration: identical bits.
// MEASURED, NOT KEPT (profiles/r04_sw_two_columns.txt): bit-identical on five shapes, and 84 k MLUPS against 158 k at 8192^2.  Two nodes per
// lane need ~350 live registers at 5 steps (ring 144, prefetched + current rows 72, two nodes' populations 36, one collision ~50, lane
// pointers ~48): the kernel takes all 512 registers, moves 700 values per three iterations through v_accvgpr_read / write (101 VALU
// instructions per collision instead of 92) and spills 164 - 196 bytes -- and scratch reloads share vmcnt with the prefetch, so every
// reload waits for the row in flight.  One wave per SIMD cannot hide that.
#pragma once

__host__ __device__ constexpr int sw2_halo(int D) { return ((D - 1) + 1) & ~1; }            // halo columns per side, rounded up to even
__host__ __device__ constexpr int sw2_strip_width(int D) { return (128 - 2 * sw2_halo(D)) / 8 * 8; }

template <class Model, int D, int K, bool NT_STORE>
__device__ __forceinline__ void sw2_iteration(double (&ringA)[D - 1][3][Q], double (&ringB)[D - 1][3][Q], double (&curA)[Q],
                                              double (&curB)[Q], double* __restrict__ pn, const double* __restrict__ po, const Geom& g,
                                              const Model& m, int i, int rbase, int R0, int R1, const int (&colsA)[3],
                                              const int (&colsB)[3], bool lane_ok) {
  // ---- prefetch level-1 inputs of the NEXT iteration ----------------------------------------------------------------------
  double nxtA[Q], nxtB[Q];
  {
    const int r1n = rbase + i + 1;
    int rr[3] = {r1n + 1, r1n, r1n - 1};  // rows supplying cx = -1, 0, +1
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (g.ghost) rr[k] = rr[k] < -g.ghost ? -g.ghost : (rr[k] > g.R + g.ghost - 1 ? g.R + g.ghost - 1 : rr[k]);
      else rr[k] = rr[k] < 0 ? rr[k] + g.R : (rr[k] >= g.R ? rr[k] - g.R : rr[k]);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double* rowp = po + q * g.plane + g.at(rr[icx(q) + 1], 0);
      if (icy(q) == 0) {  // the lane's own pair of columns: one aligned 16-byte load
        const dbl2 v = *reinterpret_cast<const dbl2*>(rowp + colsA[1]);
        nxtA[q] = v.x;
        nxtB[q] = v.y;
      } else {
        nxtA[q] = rowp[colsA[icy(q) + 1]];
        nxtB[q] = rowp[colsB[icy(q) + 1]];
      }
    }
  }
  // ---- level 1 ----------------------------------------------------------------------------------------------------------------
  double fA[Q], fB[Q], rho, ux, uy;
#pragma unroll
  for (int q = 0; q < Q; ++q) fA[q] = curA[q], fB[q] = curB[q];
  m.collide(fA, rho, ux, uy);
  __builtin_amdgcn_sched_barrier(0);  // one collision's working set at a time
  m.collide(fB, rho, ux, uy);
  __builtin_amdgcn_sched_barrier(0);
  // ---- levels 2..D ---------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int l = 2; l <= D; ++l) {
#pragma unroll
    for (int q = 0; q < Q; ++q) ringA[l - 2][K][q] = fA[q], ringB[l - 2][K][q] = fB[q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int slot = icx(q) == -1 ? K : (icx(q) == 0 ? (K + 2) % 3 : (K + 1) % 3);
      const double vA = ringA[l - 2][slot][q], vB = ringB[l - 2][slot][q];
      if (icy(q) == 1) {  // from column c - 1: A's is the previous lane's B, B's is this lane's A
        fA[q] = lane_from_prev(vB);
        fB[q] = vA;
      } else if (icy(q) == -1) {  // from column c + 1: A's is this lane's B, B's is the next lane's A
        fA[q] = vB;
        fB[q] = lane_from_next(vA);
      } else {
        fA[q] = vA;
        fB[q] = vB;
      }
    }
    m.collide(fA, rho, ux, uy);
    __builtin_amdgcn_sched_barrier(0);
    m.collide(fB, rho, ux, uy);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- store level D's row: 16 bytes per lane -----------------------------------------------------------------------------------
  const int rD = rbase + i - (D - 1);
  if (lane_ok && rD >= R0 && rD < R1) {
    const long o = g.at(rD, 0) + colsA[1];
#pragma unroll
    for (int q = 0; q < Q; ++q) store2<NT_STORE>(pn + q * g.plane + o, fA[q], fB[q]);
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) curA[q] = nxtA[q], curB[q] = nxtB[q];
}

template <class Model, int D, bool NT_STORE>
__global__ __launch_bounds__(128, 1) void k_stream_collide_sw2(double* __restrict__ pn, const double* __restrict__ po, Geom g, Model m,
                                                               int row_begin, int row_end, int rows_per_chunk, int strips,
                                                               int n_waves, int chunk_stride) {
  constexpr int H = sw2_halo(D), W = sw2_strip_width(D);
  const int wave = blockIdx.x * 2 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= n_waves) return;
  const int strip = wave % strips, chunk = wave / strips;
  const int R0 = row_begin + chunk * (chunk_stride > 0 ? chunk_stride : rows_per_chunk);
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  int cA = strip * W - H + 2 * lane;  // even: a pair never straddles the periodic wrap (C is even)
  const bool lane_ok = 2 * lane >= H && 2 * lane < H + W && cA >= 0 && cA + 1 < g.C;
  cA = cA < 0 ? cA + g.C : (cA >= g.C ? cA - g.C : cA);
  cA = cA >= g.C ? cA - g.C : cA;
  const int cB = cA + 1;
  const int colsA[3] = {cB, cA, wrap_col(g, cA - 1)};
  const int colsB[3] = {wrap_col(g, cB + 1), cB, cA};
  const int rbase = R0 - (D - 1);
  const int n_iter = (R1 - R0) + 2 * (D - 1);
  double ringA[D - 1][3][Q], ringB[D - 1][3][Q];
#pragma unroll
  for (int a = 0; a < D - 1; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int q = 0; q < Q; ++q) ringA[a][b][q] = 1.0, ringB[a][b][q] = 1.0;  // warm-up garbage, never stored
  double curA[Q], curB[Q];
  {
    int rr[3] = {rbase + 1, rbase, rbase - 1};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (g.ghost) rr[k] = rr[k] < -g.ghost ? -g.ghost : (rr[k] > g.R + g.ghost - 1 ? g.R + g.ghost - 1 : rr[k]);
      else rr[k] = rr[k] < 0 ? rr[k] + g.R : (rr[k] >= g.R ? rr[k] - g.R : rr[k]);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double* rowp = po + q * g.plane + g.at(rr[icx(q) + 1], 0);
      curA[q] = rowp[colsA[icy(q) + 1]];
      curB[q] = rowp[colsB[icy(q) + 1]];
    }
  }
  for (int i = 0; i < n_iter; i += 3) {
    sw2_iteration<Model, D, 0, NT_STORE>(ringA, ringB, curA, curB, pn, po, g, m, i, rbase, R0, R1, colsA, colsB, lane_ok);
    sw2_iteration<Model, D, 1, NT_STORE>(ringA, ringB, curA, curB, pn, po, g, m, i + 1, rbase, R0, R1, colsA, colsB, lane_ok);
    sw2_iteration<Model, D, 2, NT_STORE>(ringA, ringB, curA, curB, pn, po, g, m, i + 2, rbase, R0, R1, colsA, colsB, lane_ok);
  }
}
