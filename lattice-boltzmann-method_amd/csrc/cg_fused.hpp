// Colour-gradient MRT two-phase step, ONE kernel per time step ("fused" path; cg.hpp keeps the
// two-pass kernels that follow the reference operation by operation and are bit-identical to the
// oracle).  Same mathematics (test/mrtcg_rayleigh_taylor.cpp:431-477), reassociated:
//
//  * only the SUM over the two colours of the relaxed populations enters recolouring (:455), and
//    the MRT operator is linear, so ONE moment transform of d = (feq_r + feq_b) - (f_r + f_b)
//    replaces the two 9x9 x 2 products per colour; rows 0, 3, 5 of S are zero and are skipped;
//    the remaining 6 moments come from a butterfly over opposite pairs;
//  * the correction terms C_k (update_C :320-336) enter only summed, so the stencil runs on
//    Q_r + Q_b: 3 staged fields (psi, Qx, Qy) instead of 5, antisymmetric taps paired;
//  * 36 of the 39 f64 divisions per node become multiplications by 1/rho and 1/(1e-20 + |grad psi|);
//  * rho_r, rho_b, u are recomputed in the tile (+-2 halo ring) from the streamed populations
//    instead of round-tripping through HBM: 288 B/LUP (read 18 + write 18 doubles) instead of
//    the 496 B/LUP of the two-pass form; halo re-reads hit L2.
//
// Results differ from the reference order by rounding only (tolerance stated in
// tests/test_gpu_cg.py); FMA contraction is enabled here per source expression (contract(on)), so the
// two kernels below -- tile and column-strip -- produce identical bits.
#pragma once
#include "cg.hpp"

namespace lbm {

struct CgFast {
  double inv_rho0[2], qc[2], beta[2];
  double phi0[2], phi1[2], phi5[2];  // phi by |c|^2 class (colour.cpp:49-64)
  double eta1[2], eta5[2];
  double sigma, Gr, Gc;
  int add_source;
  double delta, r_omega, b_omega, s1, s2, s3, t2, t3;
};

inline CgFast make_cg_fast(const CgConsts& c) {
  CgFast f;
  for (int k = 0; k < 2; ++k) {
    f.inv_rho0[k] = 1.0 / c.k[k].rho_0;
    f.qc[k] = c.k[k].qcoef;
    f.beta[k] = c.k[k].beta;
    f.phi0[k] = c.k[k].phi[0];
    f.phi1[k] = c.k[k].phi[1];
    f.phi5[k] = c.k[k].phi[5];
    f.eta1[k] = c.k[k].eta[1];
    f.eta5[k] = c.k[k].eta[5];
  }
  f.sigma = c.sigma;
  f.Gr = c.gr;
  f.Gc = c.gc;
  f.add_source = c.add_source;
  f.delta = c.delta;
  f.r_omega = c.r_omega;
  f.b_omega = c.b_omega;
  f.s1 = c.s1;
  f.s2 = c.s2;
  f.s3 = c.s3;
  f.t2 = c.t2;
  f.t3 = c.t3;
  return f;
}

__device__ __forceinline__ double cg_snu_fast(const CgFast& c, double psi) {  // :84-100
  double v = 0.0;
  if (psi > c.delta) v = c.r_omega;
  if (c.delta >= psi && psi > 0.0) v = c.s1 + c.s2 * psi + c.s3 * psi * psi;
  if (0.0 >= psi && psi >= -c.delta) v = c.s1 + c.t2 * psi + c.t3 * psi * psi;
  if (psi < -c.delta) v = c.b_omega;
  return v;
}

// streamed populations of node (gr, gc) -> colour-summed populations and macroscopic fields
// (:466-477), phase field (:212-225) and Q = sum_k (1.8 alpha_k - 0.8) rho_k u (:326-327)
struct CgNode {
  double rr, rb, ux, uy, irt, psi, qx, qy;
};
// INTERIOR: the node and its 8 neighbours are inside the block and no boundary fix-up applies
template <bool INTERIOR>
__device__ __forceinline__ CgNode cg_node(double (&ft)[Q], const double* __restrict__ in_r,
                                          const double* __restrict__ in_b, const Geom& g,
                                          const Bc& bc, const CgFast& cf, int gr, int gc) {
#pragma clang fp contract(on)
  double fr[Q];
  if (INTERIOR) {
    const long o = g.at(gr, gc);
#pragma unroll
    for (int q = 0; q < Q; ++q) fr[q] = in_r[q * g.plane + (o - icx(q) * g.C - icy(q))];
#pragma unroll
    for (int q = 0; q < Q; ++q) ft[q] = in_b[q * g.plane + (o - icx(q) * g.C - icy(q))];
  } else {
    gather_bc(fr, in_r, g, bc, gr, gc);
    gather_bc(ft, in_b, g, bc, gr, gc);
  }
  CgNode n;
  n.rr = (((fr[0] + fr[1]) + (fr[2] + fr[3])) + ((fr[4] + fr[5]) + (fr[6] + fr[7]))) + fr[8];
  n.rb = (((ft[0] + ft[1]) + (ft[2] + ft[3])) + ((ft[4] + ft[5]) + (ft[6] + ft[7]))) + ft[8];
#pragma unroll
  for (int q = 0; q < Q; ++q) ft[q] += fr[q];
  const double jx = ((ft[1] - ft[3]) + (ft[5] - ft[6])) + (ft[8] - ft[7]);
  const double jy = ((ft[2] - ft[4]) + (ft[5] - ft[8])) + (ft[6] - ft[7]);
  n.irt = 1.0 / (n.rr + n.rb);
  n.ux = (jx + 0.5 * cf.Gr) * n.irt;  // u + Fg / (2 rho), :477
  n.uy = (jy + 0.5 * cf.Gc) * n.irt;
  const double a = n.rr * cf.inv_rho0[0], b = n.rb * cf.inv_rho0[1];
  n.psi = (a - b) / (a + b);
  const double qcs = cf.qc[0] * n.rr + cf.qc[1] * n.rb;
  n.qx = qcs * n.ux;
  n.qy = qcs * n.uy;
  return n;
}

// {a0[j], a1[j]}: the taps of the stencil, in constant memory.  The loops below stay rolled (unrolled, the scheduler hoists
// all LDS reads of a stencil to the top: ~100 extra live VGPRs) and index the taps with their -- wave-uniform -- counter:
// from here that is one s_load_dwordx4 per step; from a constexpr array it was a chain of 15 v_cndmask per step, more
// VALU instructions than the 8 f64 operations they fed (found on k_cg_walk's ISA, round 3).
static __constant__ double CG_STENCIL_TAPS[5][2] = {
    {2 * (1.0 / 5040.0) * 1, (1.0 / 5040.0) * 32},  {2 * (1.0 / 5040.0) * 32, (1.0 / 5040.0) * 448},
    {2 * (1.0 / 5040.0) * 84, (1.0 / 5040.0) * 960}, {2 * (1.0 / 5040.0) * 32, (1.0 / 5040.0) * 448},
    {2 * (1.0 / 5040.0) * 1, (1.0 / 5040.0) * 32}};
// 5x5 isotropic derivative (differential.hpp:9-16) with the antisymmetric taps paired.
// d/d(row): sum_j [ 2 a0_j (P[4][j] - P[0][j]) + a1_j (P[3][j] - P[1][j]) ]; d/d(col) transposed.
template <int LDC>
__device__ __forceinline__ double cg_ddrow(const double (*s)[LDC], int tr, int tc) {
#pragma clang fp contract(on)
  double acc = 0.0;
  // NOT unrolled: unrolled, the scheduler hoists the 20 LDS reads of each of the four stencils to the
  // top of the collision (~100 extra live VGPRs: 196 instead of 96 for stencil + collision)
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    const double a0 = CG_STENCIL_TAPS[j][0], a1 = CG_STENCIL_TAPS[j][1];
    acc += a0 * (s[tr + 4][tc + j] - s[tr][tc + j]);
    acc += a1 * (s[tr + 3][tc + j] - s[tr + 1][tc + j]);
  }
  return acc;
}
template <int LDC>
__device__ __forceinline__ double cg_ddcol(const double (*s)[LDC], int tr, int tc) {
#pragma clang fp contract(on)
  double acc = 0.0;
#pragma unroll 1
  for (int i = 0; i < 5; ++i) {
    const double a0 = CG_STENCIL_TAPS[i][0], a1 = CG_STENCIL_TAPS[i][1];
    acc += a0 * (s[tr + i][tc + 4] - s[tr + i][tc]);
    acc += a1 * (s[tr + i][tc + 3] - s[tr + i][tc + 1]);
  }
  return acc;
}

// collision of one node from its colour-summed streamed populations, its macroscopic fields and
// the four stencil results; writes both colours (and the observable fields on request)
// cg_collide_values: the post-collision populations of both colours as values (the two-step kernel keeps level 1's in LDS);
// cg_collide_store: the same, stored.
__device__ __forceinline__ void cg_collide_values(
    const double (&ft)[Q], const CgNode& me, double gx, double gy, double dxqx, double dyqy,
    const CgFast& cf, double (&out_r)[Q], double (&out_b)[Q], double& s_nu_out) {
#pragma clang fp contract(on)
  const double rr = me.rr, rb = me.rb, ux = me.ux, uy = me.uy, irt = me.irt;
  const double rt = rr + rb;
  const double s_nu = cg_snu_fast(cf, me.psi);
  constexpr double W0 = 4.0 / 9.0, W1 = 1.0 / 9.0, W5 = 1.0 / 36.0;
  constexpr double B0 = -4.0 / 27.0, B1 = 2.0 / 27.0, B5 = 5.0 / 108.0;  // :158-163

  // perturbation operator (:263-273, :290-300), identical for both colours: 5 distinct values
  const double gn = sqrt(gx * gx + gy * gy);
  const double ig = 1.0 / (1e-20 + gn);
  const double hA = 0.5 * (4.5 * cf.sigma * s_nu) * gn;
  const double ig2 = ig * ig, gs = gx + gy, gd = gx - gy;
  const double p0 = -2.0 * hA * B0;  // 2 Omega2 (once per colour)
  const double p1 = 2.0 * hA * (W1 * (gx * gx * ig2) - B1), p2 = 2.0 * hA * (W1 * (gy * gy * ig2) - B1);
  const double p5 = 2.0 * hA * (W5 * (gs * gs * ig2) - B5), p6 = 2.0 * hA * (W5 * (gd * gd * ig2) - B5);

  // feq_r + feq_b (:233-247) split into the parts even and odd under c -> -c
  const double P0 = rr * cf.phi0[0] + rb * cf.phi0[1], P1 = rr * cf.phi1[0] + rb * cf.phi1[1],
               P5 = rr * cf.phi5[0] + rb * cf.phi5[1];
  const double H1 = rr * cf.eta1[0] + rb * cf.eta1[1], H5 = rr * cf.eta5[0] + rb * cf.eta5[1];
  const double c3 = -3.0 * (ux * ux + uy * uy), us = ux + uy, ud = ux - uy;
  const double E0 = P0 + W0 * rt * c3;
  const double E1 = P1 + W1 * rt * (9.0 * ux * ux + c3), E2 = P1 + W1 * rt * (9.0 * uy * uy + c3);
  const double E5 = P5 + W5 * rt * (9.0 * us * us + c3), E6 = P5 + W5 * rt * (9.0 * ud * ud + c3);
  const double O1 = (3.0 * W1) * H1 * ux, O2 = (3.0 * W1) * H1 * uy;
  const double O5 = (3.0 * W5) * H5 * us, O8 = (3.0 * W5) * H5 * ud;
  const double d0 = E0 - ft[0];
  const double d1 = (E1 + O1) - ft[1], d3 = (E1 - O1) - ft[3];
  const double d2 = (E2 + O2) - ft[2], d4 = (E2 - O2) - ft[4];
  const double d5 = (E5 + O5) - ft[5], d7 = (E5 - O5) - ft[7];
  const double d8 = (E6 + O8) - ft[8], d6 = (E6 - O8) - ft[6];

  // M d for the 6 relaxed moments (:130-140, rows 1, 2, 4, 6, 7, 8)
  const double a = d1 + d3, b = d2 + d4, p = d5 + d7, qd = d6 + d8, cd = p + qd, ab = a + b;
  const double e13 = d1 - d3, e24 = d2 - d4, e57 = d5 - d7, e68 = d6 - d8;
  const double m1 = 2.0 * cd - 4.0 * d0 - ab;
  const double m2 = 4.0 * d0 - 2.0 * ab + cd;
  const double m4 = (e57 - e68) - 2.0 * e13;
  const double m6 = (e57 + e68) - 2.0 * e24;
  const double m7 = a - b, m8 = p - qd;
  // S m + C_r + C_b (:249-261, :320-336)
  const double n1 = 1.25 * m1 + 1.125 * (dxqx + dyqy);
  const double n2 = 1.14 * m2;
  const double n4 = 1.6 * m4, n6 = 1.6 * m6;
  const double n7 = s_nu * m7 + (1.0 - 0.5 * s_nu) * (dxqx - dyqy);
  const double n8 = s_nu * m8;
  // M^-1 (:146-156)
  const double A = (-1.0 / 36.0) * (n1 + 2.0 * n2), Bq = (1.0 / 36.0) * (2.0 * n1 + n2);
  const double k4 = (1.0 / 6.0) * n4, k6 = (1.0 / 6.0) * n6, k7 = 0.25 * n7, k8 = 0.25 * n8;
  const double h4 = 0.5 * k4, h6 = 0.5 * k6;
  double tot[Q];  // total_f, :455
  tot[0] = ft[0] + (1.0 / 9.0) * (n2 - n1) + p0;
  tot[1] = ft[1] + ((A - k4) + k7) + p1;
  tot[3] = ft[3] + ((A + k4) + k7) + p1;
  tot[2] = ft[2] + ((A - k6) - k7) + p2;
  tot[4] = ft[4] + ((A + k6) - k7) + p2;
  tot[5] = ft[5] + ((Bq + h4) + (h6 + k8)) + p5;
  tot[6] = ft[6] + ((Bq - h4) + (h6 - k8)) + p6;
  tot[7] = ft[7] + ((Bq - h4) - (h6 - k8)) + p5;
  tot[8] = ft[8] + ((Bq + h4) - (h6 + k8)) + p6;

  // recolouring (:275-288, :302-318): kappa_q = rho_r rho_b (grad psi . c_q/|c_q|) P_q / (rho^2 |grad psi|)
  const double xr = rr * irt, xb = rb * irt;
  const double kk = (rr * rb) * (ig * (irt * irt));
  const double K1 = kk * P1, K5 = (kk * P5) * 0.70710678118654752440;
  double kap[Q];
  kap[0] = 0.0;
  kap[1] = K1 * gx;
  kap[3] = -kap[1];
  kap[2] = K1 * gy;
  kap[4] = -kap[2];
  kap[5] = K5 * gs;
  kap[7] = -kap[5];
  kap[8] = K5 * gd;
  kap[6] = -kap[8];
  double src[Q];
  if (cf.add_source) {  // :460-464 (unweighted Guo-type term)
    const double sf = 1.0 - 0.5 * s_nu, uFg3 = 3.0 * (ux * cf.Gr + uy * cf.Gc);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double cu = ux * (double)icx(q) + uy * (double)icy(q);
      const double FgE = cf.Gr * (double)icx(q) + cf.Gc * (double)icy(q);
      src[q] = (sf * wq(q)) * ((3.0 + 9.0 * cu) * FgE - uFg3);
    }
  } else {
#pragma unroll
    for (int q = 0; q < Q; ++q) src[q] = 0.0;
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    out_r[q] = (xr * tot[q] + cf.beta[0] * kap[q]) + src[q];
    out_b[q] = (xb * tot[q] + cf.beta[1] * kap[q]) + src[q];
  }
  s_nu_out = s_nu;
}

template <bool WITH_FIELDS>
__device__ __forceinline__ void cg_collide_store(
    const double (&ft)[Q], const CgNode& me, double gx, double gy, double dxqx, double dyqy,
    const CgFast& cf, const Geom& g, const MacroIdx& mi, int r, int c, double* __restrict__ pn_r,
    double* __restrict__ pn_b, double* __restrict__ rho_r_out, double* __restrict__ rho_b_out,
    double* __restrict__ u_out, double* __restrict__ psi_out, double* __restrict__ snu_out) {
  double out_r[Q], out_b[Q], s_nu;
  cg_collide_values(ft, me, gx, gy, dxqx, dyqy, cf, out_r, out_b, s_nu);
  const long lo = g.at(r, c);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    // streaming stores: the new lattices are not read again in this launch, and the L2 they would
    // occupy is what serves the ring re-reads of the neighbouring tiles
    __builtin_nontemporal_store(out_r[q], &pn_r[q * g.plane + lo]);
    __builtin_nontemporal_store(out_b[q], &pn_b[q * g.plane + lo]);
  }
  if (WITH_FIELDS) {
    const long o = mi.at(r, c), oo = (long)r * g.C + c;  // diagnostics carry no ghost rows
    rho_r_out[o] = me.rr;
    rho_b_out[o] = me.rb;
    u_out[o] = me.ux;
    u_out[mi.n + o] = me.uy;
    psi_out[oo] = me.psi;
    snu_out[oo] = s_nu;
  }
}

// The tiles of a launch split into an INNER rectangle [ir0, ir1) x [ic0, ic1) (tile coordinates) --
// every node of the tile, of its +-2 ring and of their +-1 gathers lies inside the block or its
// ghost rows and carries no boundary fix-up: plain offsets, no clamps, no wraps -- and the FRAME
// around it, which keeps the general boundary gather.  MODE 0: all tiles through the general path
// (small lattices), 1: the inner tiles, 2: the frame.  Same arithmetic per node in every mode.
struct CgTileRect {
  int ir0, ir1, ic0, ic1;
};
template <int TR, int TC, bool WITH_FIELDS, int MODE>
__device__ __forceinline__ void cg_fused_body(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, const Geom& g, const Bc& bc, const CgFast& cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, const MacroIdx& mi, int row_begin, int row_end, int xcd_swizzle,
    const CgTileRect& rect, int block_index, int grid_size) {
#pragma clang fp contract(on)
  constexpr int NT = TR * TC, LR = TR + 4, LC = TC + 4, LDC = LC + 1;
  constexpr bool INNER = MODE == 1;
  __shared__ double s_psi[LR][LDC], s_qx[LR][LDC], s_qy[LR][LDC];
  const int tiles_c = MODE == 1 ? rect.ic1 - rect.ic0 : (g.C + TC - 1) / TC;
  // XCD-aware tile order: the hardware deals consecutive workgroups round-robin over the 8 XCDs,
  // each with its own L2.  Neighbouring tiles share their +-3 ring (and, at 128-B lines, whole
  // cache lines on both sides of a 32-column tile: 2x read amplification when every XCD fetches
  // them on its own), so XCD k gets the k-th contiguous eighth of the tile sequence.
  int tile = block_index;
  if (xcd_swizzle == 1) {  // XCD k gets the k-th contiguous eighth of the tile sequence
    const int per = grid_size / 8;
    if (tile < per * 8) tile = (tile % 8) * per + tile / 8;
  } else if (xcd_swizzle > 1) {
    // groups of G column-neighbour tiles per XCD, all XCDs inside the same window of 8 G tiles:
    // workgroup b runs on XCD b % 8 as its (b / 8)-th block; consecutive blocks of an XCD get
    // consecutive tiles of a group, so the shared ring lines are L2 hits, while the eight XCDs
    // keep streaming through the same region of the lattice
    const int G = xcd_swizzle, x = tile % 8, m = tile / 8, win = 8 * G;
    const int t2 = (m / G) * win + x * G + (m % G);
    if ((m / G + 1) * win <= (int)grid_size) tile = t2;
  }
  int tile_r = tile / tiles_c, tile_c = tile % tiles_c;
  if (MODE == 1) {
    tile_r += rect.ir0;
    tile_c += rect.ic0;
  } else if (MODE == 2) {  // frame: tile rows above the rectangle, below it, then its left / right margins
    const int top = rect.ir0 * tiles_c, tiles_r = (row_end - row_begin + TR - 1) / TR;
    const int bottom = (tiles_r - rect.ir1) * tiles_c, margin = rect.ic0 + (tiles_c - rect.ic1);
    if (tile < top) {
      tile_r = tile / tiles_c;
      tile_c = tile % tiles_c;
    } else if (tile < top + bottom) {
      tile_r = rect.ir1 + (tile - top) / tiles_c;
      tile_c = (tile - top) % tiles_c;
    } else {
      const int t = tile - top - bottom, k = t % margin;
      tile_r = rect.ir0 + t / margin;
      tile_c = k < rect.ic0 ? k : rect.ic1 + (k - rect.ic0);
    }
  }
  const int r_base = row_begin + tile_r * TR, c_base = tile_c * TC;
  const int rlo = cg_row_lo(g, bc), rhi = cg_row_hi(g, bc);
  const int tr = threadIdx.x / TC, tc = threadIdx.x % TC;
  const int r = r_base + tr, c = c_base + tc;

  // own node first (keeps the colour-summed populations in registers), then the +-2 ring.
  // Replicate padding of the stencils = clamping the node the fields are evaluated at
  // (differential.cpp:5-9) -- at the edges of the GLOBAL domain only; across a slab seam the
  // ring continues into the neighbour's (ghost) rows.
  double ft[Q];
  CgNode me;
  constexpr int NH = LR * LC - TR * TC;
  auto ring_slot = [&](int i, int& lr, int& lc) {
    if (i < 2 * LC) {
      lr = i / LC;
      lc = i % LC;
    } else if (i < 4 * LC) {
      lr = TR + 2 + (i - 2 * LC) / LC;
      lc = (i - 2 * LC) % LC;
    } else {
      const int j = i - 4 * LC;
      lr = 2 + (j >> 2);
      lc = (j & 3) < 2 ? (j & 3) : TC + (j & 3);
    }
  };
  {
    me = INNER ? cg_node<true>(ft, in_r, in_b, g, bc, cf, r, c)
               : cg_node<false>(ft, in_r, in_b, g, bc, cf, r > rhi ? rhi : r, c > g.C - 1 ? g.C - 1 : c);
    for (int i = threadIdx.x; i < NH; i += NT) {
      int lr, lc;
      ring_slot(i, lr, lc);
      int gr = r_base + lr - 2, gc = c_base + lc - 2;
      if (!INNER) {
        gr = gr < rlo ? rlo : (gr > rhi ? rhi : gr);
        gc = gc < 0 ? 0 : (gc > g.C - 1 ? g.C - 1 : gc);
      }
      double tmp[Q];
      const CgNode nb = INNER ? cg_node<true>(tmp, in_r, in_b, g, bc, cf, gr, gc)
                              : cg_node<false>(tmp, in_r, in_b, g, bc, cf, gr, gc);
      s_psi[lr][lc] = nb.psi;
      s_qx[lr][lc] = nb.qx;
      s_qy[lr][lc] = nb.qy;
    }
  }
  s_psi[tr + 2][tc + 2] = me.psi;
  s_qx[tr + 2][tc + 2] = me.qx;
  s_qy[tr + 2][tc + 2] = me.qy;
  __syncthreads();
  if (!INNER && (r >= row_end || c >= g.C)) return;

  const double gx = cg_ddrow<LDC>(s_psi, tr, tc), gy = cg_ddcol<LDC>(s_psi, tr, tc);
  const double dxqx = cg_ddrow<LDC>(s_qx, tr, tc), dyqy = cg_ddcol<LDC>(s_qy, tr, tc);

  cg_collide_store<WITH_FIELDS>(ft, me, gx, gy, dxqx, dyqy, cf, g, mi, r, c, pn_r, pn_b, rho_r_out,
                                rho_b_out, u_out, psi_out, snu_out);
}

template <int TR, int TC, int WAVES, bool WITH_FIELDS, int MODE = 0>
__global__ __launch_bounds__(TR* TC, WAVES) void k_cg_fused(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, Bc bc, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int xcd_swizzle,
    CgTileRect rect = CgTileRect{0, 0, 0, 0}) {
  cg_fused_body<TR, TC, WITH_FIELDS, MODE>(pn_r, pn_b, in_r, in_b, g, bc, cf, rho_r_out, rho_b_out, u_out, psi_out, snu_out, mi,
                                           row_begin, row_end, xcd_swizzle, rect, (int)blockIdx.x, (int)gridDim.x);
}

// ---- the tile kernel with SEVERAL NODES PER THREAD (round 4) ----------------------------------------------------------
// Same structure as k_cg_fused MODE 1 (one barrier, independent workgroups), larger tile: a workgroup of NT threads covers
// TR x TC nodes, NPT = TR TC / NT per thread (node n = thread + k NT, so a wave still reads whole rows), which cuts what
// the 16 x 32 tile pays for its ring: rows fetched (TR + 6) / TR, 128-byte lines per row (TC / 16 + 2) / (TC / 16), ring
// nodes reduced (TR + 4)(TC + 4) / (TR TC).  Between the barrier and its collision a node waits as 11 doubles (colour sums
// + the two densities); u, 1 / rho are recomputed by the expressions of cg_node, psi comes back from the LDS tile: identical
// bits.  The thread's nodes collide one after another (a scheduling barrier between them keeps the register count that of
// one collision + 22 per waiting node).  PARK: the waiting nodes' colour sums wait in LDS instead (9 doubles per node, lane-major:
// no conflicts, no barrier -- a thread reads back what it wrote), which is what lets two nodes per thread fit the 128
// registers of four waves per SIMD without scratch.  Inner rectangle only: plain offsets, no clamps, no wraps (the frame
// keeps k_cg_fused).
template <int TR, int TC, int NT, int MINB, bool PARK, bool WITH_FIELDS>
__global__ __launch_bounds__(NT, MINB) void k_cg_tile_mn(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int ra, int ca, int tiles_c, int xcd_swizzle) {
#pragma clang fp contract(on)
  static_assert((TR * TC) % NT == 0 && NT % TC == 0, "whole rows per pass");
  constexpr int NPT = TR * TC / NT, RPP = NT / TC, LR = TR + 4, LC = TC + 4, LDC = LC + 1;
  __shared__ double s_psi[LR][LDC], s_qx[LR][LDC], s_qy[LR][LDC];
  __shared__ double s_park[PARK ? NPT - 1 : 1][PARK ? Q : 1][PARK ? NT : 1];
  int tile = blockIdx.x;
  if (xcd_swizzle > 1) {  // groups of G column-neighbour tiles per XCD inside a common window (as k_cg_fused)
    const int G = xcd_swizzle, x = tile % 8, m = tile / 8, win = 8 * G;
    const int t2 = (m / G) * win + x * G + (m % G);
    if ((m / G + 1) * win <= (int)gridDim.x) tile = t2;
  } else if (xcd_swizzle == 1) {
    const int per = gridDim.x / 8;
    if (tile < per * 8) tile = (tile % 8) * per + tile / 8;
  }
  const int r_base = ra + (tile / tiles_c) * TR, c_base = ca + (tile % tiles_c) * TC;
  const int tr0 = threadIdx.x / TC, tc = threadIdx.x % TC;

  double ft[PARK ? 1 : NPT][Q], rr[NPT], rb[NPT];
#pragma unroll
  for (int k = NPT - 1; k >= 0; --k) {  // node 0 last: it stays in registers
    const int tr = tr0 + k * RPP;
    double fk[Q];
    const CgNode me = cg_node<true>(fk, in_r, in_b, g, Bc{}, cf, r_base + tr, c_base + tc);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if (PARK && k > 0) s_park[k - 1][q][threadIdx.x] = fk[q];
      else ft[PARK ? 0 : k][q] = fk[q];
    }
    rr[k] = me.rr;
    rb[k] = me.rb;
    s_psi[tr + 2][tc + 2] = me.psi;
    s_qx[tr + 2][tc + 2] = me.qx;
    s_qy[tr + 2][tc + 2] = me.qy;
    __builtin_amdgcn_sched_barrier(0);
  }
  constexpr int NH = LR * LC - TR * TC;
  for (int i = threadIdx.x; i < NH; i += NT) {  // the +-2 ring: two rows above, two below, then 2 + 2 columns beside each row
    int lr, lc;
    if (i < 2 * LC) {
      lr = i / LC;
      lc = i % LC;
    } else if (i < 4 * LC) {
      lr = TR + 2 + (i - 2 * LC) / LC;
      lc = (i - 2 * LC) % LC;
    } else {
      const int j = i - 4 * LC;
      lr = 2 + (j >> 2);
      lc = (j & 3) < 2 ? (j & 3) : TC + (j & 3);
    }
    double tmp[Q];
    const CgNode nb = cg_node<true>(tmp, in_r, in_b, g, Bc{}, cf, r_base + lr - 2, c_base + lc - 2);
    s_psi[lr][lc] = nb.psi;
    s_qx[lr][lc] = nb.qx;
    s_qy[lr][lc] = nb.qy;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NPT; ++k) {
    int tr = tr0 + k * RPP;
    // opaque per node: otherwise the 18 store addresses of the later nodes are derived from the first node's and wait
    // in registers (scratch at 128) through its collision
    asm volatile("" : "+v"(tr));
    // likewise the wave-uniform f64 products of the source term (c_q . Fg: VALU work, there is no scalar f64 unit) are
    // recomputed per node instead of waiting in ten register pairs
    CgFast cfk = cf;
    asm volatile("" : "+s"(cfk.Gr), "+s"(cfk.Gc));
    double fk[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) fk[q] = (PARK && k > 0) ? s_park[k > 0 ? k - 1 : 0][q][threadIdx.x] : ft[PARK ? 0 : k][q];
    CgNode me;  // == cg_node's expressions on the held sums
    me.rr = rr[k];
    me.rb = rb[k];
    const double jx = ((fk[1] - fk[3]) + (fk[5] - fk[6])) + (fk[8] - fk[7]);
    const double jy = ((fk[2] - fk[4]) + (fk[5] - fk[8])) + (fk[6] - fk[7]);
    me.irt = 1.0 / (me.rr + me.rb);
    me.ux = (jx + 0.5 * cfk.Gr) * me.irt;
    me.uy = (jy + 0.5 * cfk.Gc) * me.irt;
    me.psi = s_psi[tr + 2][tc + 2];
    me.qx = me.qy = 0.0;  // not used by the collision
    const double gx = cg_ddrow<LDC>(s_psi, tr, tc), gy = cg_ddcol<LDC>(s_psi, tr, tc);
    const double dxqx = cg_ddrow<LDC>(s_qx, tr, tc), dyqy = cg_ddcol<LDC>(s_qy, tr, tc);
    cg_collide_store<WITH_FIELDS>(fk, me, gx, gy, dxqx, dyqy, cfk, g, mi, r_base + tr, c_base + tc, pn_r, pn_b,
                                  rho_r_out, rho_b_out, u_out, psi_out, snu_out);
    if (k + 1 < NPT) __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef LBM_EXPERIMENTS  // the launch forms below (merged dispatch, strip kernels 1 - 4) were measured and not kept: make EXPERIMENTS=1
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

#endif  // LBM_EXPERIMENTS (strip kernels, generations 1 - 4)

__device__ __forceinline__ void cg_reduce_row(const double (&fr)[Q], double (&ft)[Q], const CgFast& cf, double (&n6)[6],
                                              double& psi, double& qx, double& qy) {
#pragma clang fp contract(on)
  // == cg_node<true> on already gathered populations; ft: blue in, colour sum out
  const double rr = (((fr[0] + fr[1]) + (fr[2] + fr[3])) + ((fr[4] + fr[5]) + (fr[6] + fr[7]))) + fr[8];
  const double rb = (((ft[0] + ft[1]) + (ft[2] + ft[3])) + ((ft[4] + ft[5]) + (ft[6] + ft[7]))) + ft[8];
#pragma unroll
  for (int q = 0; q < Q; ++q) ft[q] += fr[q];
  const double jx = ((ft[1] - ft[3]) + (ft[5] - ft[6])) + (ft[8] - ft[7]);
  const double jy = ((ft[2] - ft[4]) + (ft[5] - ft[8])) + (ft[6] - ft[7]);
  const double irt = 1.0 / (rr + rb);
  const double ux = (jx + 0.5 * cf.Gr) * irt, uy = (jy + 0.5 * cf.Gc) * irt;
  const double a = rr * cf.inv_rho0[0], b = rb * cf.inv_rho0[1];
  psi = (a - b) / (a + b);
  const double qcs = cf.qc[0] * rr + cf.qc[1] * rb;
  qx = qcs * ux;
  qy = qcs * uy;
  n6[0] = rr; n6[1] = rb; n6[2] = ux; n6[3] = uy; n6[4] = irt; n6[5] = psi;
}

#ifdef LBM_EXPERIMENTS  // 14.0-14.3 k against the tile kernel's 15.3 k
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

#endif  // LBM_EXPERIMENTS (fifth form)

// ---- sixth form: a workgroup of TR x WC waves walks down a strip, TR rows a step ---------------------------------------
// Between the tile kernel (16 waves per CU, moves its actual traffic at 6.4-6.7 TB/s, but reduces 22 x 36 nodes for 16 x 32
// outputs) and the strip kernels (one row per wave and iteration, 2 waves per SIMD, a barrier or a private ring per row).
// The block covers a window of 64 WC columns (all but 2 + 2 of them outputs) and advances TR rows per step: every thread
// reduces ONE node of the new rows R_k + 2 .. R_k + TR + 1 into a ring of 2 TR + 4 field rows shared by the block, ONE
// barrier, then collides one node: the one it just reduced, or -- the last two rows of threads, whose stencils reach rows
// of the next step -- the one it reduced a step ago and parked in LDS (15 doubles).  No ring rows above or below, no extra
// nodes: each (plane, row) is read once per strip.  The loads of step k + 1 are issued right after the barrier and arrive
// behind the collision of step k (3 waves per SIMD: 168 VGPRs hold both).  Per-node arithmetic = the tile kernel's:
// identical bits.
// Addresses: one uniform 64-bit base per lattice and chunk + (32-bit scalar plane offset + this thread's 32-bit byte
// offset), the saddr form.  The empty asm statements make the step's row and the lane index opaque per step: left alone,
// the compiler hoists 36 + 18 per-thread plane addresses (and every LDS address) out of the walk loop and spills them,
// and a spill reload waits behind the prefetched loads (vmcnt counts in order).
__device__ __forceinline__ double cg_ld(const double* __restrict__ base, unsigned voff) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + voff);
}
__device__ __forceinline__ void cg_st_nt(double v, double* __restrict__ base, unsigned voff) {
  __builtin_nontemporal_store(v, reinterpret_cast<double*>(reinterpret_cast<char*>(base) + voff));
}
__device__ __forceinline__ void cg_st(double v, double* __restrict__ base, unsigned voff) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + voff) = v;
}

// PF (default): the next step's rows prefetched behind the collision, results stored a step late, 3 waves per SIMD.
// !PF: rows loaded at the head of their step and results stored at its end (the wait for the rows then covers the
// stores issued before them -- both in flight together), nothing carried over the collision: 4 waves per SIMD.
template <int TR, int WC, bool WITH_FIELDS, bool PF = true>
__global__ __launch_bounds__(TR* WC * 64, PF ? 3 : 4) void k_cg_walk(
    double* __restrict__ pn_r, double* __restrict__ pn_b, const double* __restrict__ in_r,
    const double* __restrict__ in_b, Geom g, CgFast cf, double* __restrict__ rho_r_out,
    double* __restrict__ rho_b_out, double* __restrict__ u_out, double* __restrict__ psi_out,
    double* __restrict__ snu_out, MacroIdx mi, int row_begin, int row_end, int col_begin, int col_end,
    int rows_per_chunk, int strips, int n_blocks, int xcd_order) {
#pragma clang fp contract(on)
  static_assert(TR >= 2, "two rows of threads park their node for a step");
  constexpr int LW = 64 * WC, OUTC = LW - 4, NR = 2 * TR + 4, NP = 2 * LW, KW = (4 + TR - 1) / TR;
  __shared__ double s_psi[NR][LW], s_qx[NR][LW], s_qy[NR][LW];
  __shared__ double s_park[15][NP];  // [9 colour sums, rho_r, rho_b, u_x, u_y, 1 / rho, psi][thread]: each thread's own column
  int blk = blockIdx.x;
  // XCD k takes the k-th contiguous eighth of the (chunk-major, strip-minor) sequence: the column neighbours of a
  // workgroup -- which read the same 128-byte lines at the window edges -- walk beside it behind the same L2
  if (xcd_order) blk = (blk % 8) * ((int)gridDim.x / 8) + blk / 8;  // the launch pads the grid to a multiple of 8
  if (blk >= n_blocks) return;
  const int strip = blk % strips, chunk = blk / strips;
  const int R0 = row_begin + chunk * rows_per_chunk;
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  const int n_steps = (R1 - R0 + TR - 1) / TR;
  const int c_base = col_begin + strip * OUTC;
  const int tr = __builtin_amdgcn_readfirstlane(threadIdx.x / LW), l_ = threadIdx.x % LW;
  const bool parks = tr >= TR - 2;  // uniform over a wave
  // byte offsets of the 9 gather sources and of the 9 planes from a lattice's chunk origin: 32-bit scalars for the whole
  // walk (the launcher keeps 9 planes under 4 GB), one v_add_u32 per access -- against 36 + 18 base addresses of 64 bits
  // rebuilt on the scalar unit every step, which spilled SGPRs into VGPR lanes
  unsigned goff[Q], poff[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    goff[q] = (unsigned)((q * g.plane - icx(q) * g.C - icy(q)) * 8);
    poff[q] = (unsigned)(q * g.plane * 8);
  }
  double raw_r[Q], raw_b[Q];
  auto issue = [&](int k) {  // the populations of the node this thread reduces in step k
    int Rk = R0 + k * TR;
    asm volatile("" : "+s"(Rk));
    int rn = Rk + 2 + tr;  // its row, kept inside [R0 - 2, R1 + 1]
    rn = rn < R0 - 2 ? R0 - 2 : (rn > R1 + 1 ? R1 + 1 : rn);
    const long o = g.at(R0 - 3, c_base - 3);  // one row and one column further out: goff[0 .. 8] + v >= 0
    int l = l_;
    asm volatile("" : "+v"(l));
    const int c = c_base - 2 + l;
    const int dl = (c > g.C - 2 ? g.C - 2 : c) - (c_base - 3);  // loads stay inside the lattice (such lanes feed nothing stored)
    const unsigned v = (unsigned)((rn - (R0 - 3)) * g.C + dl) * 8u;
#pragma unroll
    for (int q = 0; q < Q; ++q) raw_r[q] = cg_ld(in_r + o, goff[q] + v);
#pragma unroll
    for (int q = 0; q < Q; ++q) raw_b[q] = cg_ld(in_b + o, goff[q] + v);
  };
  // The results of step k are STORED in step k + 1, behind its reduction: vmcnt counts in order, so stores issued between
  // the prefetch and its use would make the wait for the prefetched rows a wait for the write acknowledgements as well.
  double out_r[Q], out_b[Q], of[6];
  bool stored = true;
  auto flush = [&](int Rp, int l) {
    const int m = parks ? tr - (TR - 2) : tr + 2;
    const unsigned v_out = (unsigned)(m * g.C + l) * 8u;  // from (Rp, c_base - 2)
    const long o_out = g.at(Rp, c_base - 2);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      cg_st_nt(out_r[q], pn_r + o_out, poff[q] + v_out);
      cg_st_nt(out_b[q], pn_b + o_out, poff[q] + v_out);
    }
    if (WITH_FIELDS) {
      const long o = mi.at(Rp, c_base - 2), oo = (long)Rp * g.C + (c_base - 2);  // diagnostics carry no ghost rows
      cg_st(of[0], rho_r_out + o, v_out);
      cg_st(of[1], rho_b_out + o, v_out);
      cg_st(of[2], u_out + o, v_out);
      cg_st(of[3], u_out + (mi.n + o), v_out);
      cg_st(of[4], psi_out + oo, v_out);
      cg_st(of[5], snu_out + oo, v_out);
    }
  };
  if (PF) issue(-KW);
  for (int k = -KW; k < n_steps; ++k) {  // k < 0: rows R0 - 2 .. R0 + 1 only, nothing collided
    if (!PF) issue(k);
    int Rk = R0 + k * TR;
    asm volatile("" : "+s"(Rk));
    CgFast cfl = cf;
    asm volatile("" : "+v"(cfl.Gr), "+v"(cfl.Gc));
    // ... and the lane index: everything derived from it (LDS addresses, byte offsets) is recomputed per step, not carried --
    // a carried value that spills is reloaded from scratch BEHIND the prefetched loads (vmcnt counts in order)
    int l = l_;
    asm volatile("" : "+v"(l));
    const int pl = tr * LW + l - (TR - 2) * LW;
    const bool lane_out = l >= 2 && l < LW - 2 && c_base - 2 + l < col_end;
    const int b = ((k + KW) * TR) % NR;  // ring slot of row Rk - 2
    double ft[Q], n6[6];
    {
      double psi, qx, qy;
#pragma unroll
      for (int q = 0; q < Q; ++q) ft[q] = raw_b[q];
      cg_reduce_row(raw_r, ft, cfl, n6, psi, qx, qy);
      int sl = b + tr + 4;
      sl -= sl >= NR ? NR : 0;
      s_psi[sl][l] = psi;
      s_qx[sl][l] = qx;
      s_qy[sl][l] = qy;
    }
    if (!stored) flush(Rk - TR, l);
    stored = true;
    // a parking thread swaps the node it has just reduced for the one it parked a step ago (its own column of s_park: read,
    // then write -- one buffer); here, behind the flush, neither the prefetched rows nor the deferred results are live
    if (parks) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const double t = s_park[q][pl];
        s_park[q][pl] = ft[q];
        ft[q] = t;
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const double t = s_park[9 + q][pl];
        s_park[9 + q][pl] = n6[q];
        n6[q] = t;
      }
    }
    __syncthreads();  // the fields of rows Rk - 2 .. Rk + TR + 1 are in the ring
    // (no second barrier: the next step's rows take the slots of rows Rk - TR - 2 .. Rk - 3, which nobody reads any more,
    // and no wave gets two steps ahead -- it would have to pass the next barrier first)
    if (PF && k + 1 < n_steps) issue(k + 1);
    const int m = parks ? tr - (TR - 2) : tr + 2;  // this thread collides row Rk + m
    if (k >= 0 && lane_out && Rk + m < R1) {
      int rs[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        rs[i] = b + m + i;
        rs[i] -= rs[i] >= NR ? NR : 0;
      }
      const int l0 = l - 2;  // columns c - 2 .. c + 2 sit at [l0 .. l0 + 4]
      // rolled (unrolled, the 80 LDS reads are hoisted on top of each other and spill -- scheduling barriers do not stop
      // it), with the coefficients of step j read from constant memory by the scalar unit: indexing the constexpr arrays
      // with the loop counter cost 15 v_cndmask per step for 8 f64 operations.  Same operations, same order, same values
      // as cg_ddrow / cg_ddcol.
      double gx = 0.0, dxqx = 0.0, gy = 0.0, dyqy = 0.0;
#pragma unroll 1
      for (int j = 0; j < 5; ++j) {  // == cg_ddrow
        const double c0 = CG_STENCIL_TAPS[j][0], c1 = CG_STENCIL_TAPS[j][1];
        gx += c0 * (s_psi[rs[4]][l0 + j] - s_psi[rs[0]][l0 + j]);
        gx += c1 * (s_psi[rs[3]][l0 + j] - s_psi[rs[1]][l0 + j]);
        dxqx += c0 * (s_qx[rs[4]][l0 + j] - s_qx[rs[0]][l0 + j]);
        dxqx += c1 * (s_qx[rs[3]][l0 + j] - s_qx[rs[1]][l0 + j]);
      }
      int sl = rs[0];
#pragma unroll 1
      for (int i = 0; i < 5; ++i) {  // == cg_ddcol
        const double c0 = CG_STENCIL_TAPS[i][0], c1 = CG_STENCIL_TAPS[i][1];
        gy += c0 * (s_psi[sl][l0 + 4] - s_psi[sl][l0]);
        gy += c1 * (s_psi[sl][l0 + 3] - s_psi[sl][l0 + 1]);
        dyqy += c0 * (s_qy[sl][l0 + 4] - s_qy[sl][l0]);
        dyqy += c1 * (s_qy[sl][l0 + 3] - s_qy[sl][l0 + 1]);
        sl = sl + 1 >= NR ? 0 : sl + 1;
      }
      CgNode me;
      me.rr = n6[0]; me.rb = n6[1]; me.ux = n6[2]; me.uy = n6[3]; me.irt = n6[4]; me.psi = n6[5];
      me.qx = 0.0;
      me.qy = 0.0;
      double s_nu;
      cg_collide_values(ft, me, gx, gy, dxqx, dyqy, cfl, out_r, out_b, s_nu);
      if (WITH_FIELDS) {
        of[0] = me.rr; of[1] = me.rb; of[2] = me.ux; of[3] = me.uy; of[4] = me.psi; of[5] = s_nu;
      }
      stored = false;
      if (!PF) {
        flush(Rk, l);
        stored = true;
      }
    }
  }
  if (!stored) flush(R0 + (n_steps - 1) * TR, l_);
}

#ifdef LBM_EXPERIMENTS  // bit-identical to two single steps and SLOWER than them: make EXPERIMENTS=1
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
#endif  // LBM_EXPERIMENTS

}  // namespace lbm
