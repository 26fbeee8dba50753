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
// the reduction of cg_node on already gathered populations (fr: red, ft: blue in / colour sum out) -- ONE definition, so
// that every kernel that gathers on its own (the walking tile issues its loads ahead of a barrier) leaves the same bits
__device__ __forceinline__ CgNode cg_node_reduce(const double (&fr)[Q], double (&ft)[Q], const CgFast& cf) {
#pragma clang fp contract(on)
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
// plain gather of an interior node (no clamps, no wraps): the 9 + 9 pulled populations
__device__ __forceinline__ void cg_node_gather(double (&fr)[Q], double (&ft)[Q], const double* __restrict__ in_r,
                                               const double* __restrict__ in_b, const Geom& g, int gr, int gc) {
  const long o = g.at(gr, gc);
#pragma unroll
  for (int q = 0; q < Q; ++q) fr[q] = in_r[q * g.plane + (o - icx(q) * g.P - icy(q))];
#pragma unroll
  for (int q = 0; q < Q; ++q) ft[q] = in_b[q * g.plane + (o - icx(q) * g.P - icy(q))];
}
// INTERIOR: the node and its 8 neighbours are inside the block and no boundary fix-up applies
template <bool INTERIOR>
__device__ __forceinline__ CgNode cg_node(double (&ft)[Q], const double* __restrict__ in_r,
                                          const double* __restrict__ in_b, const Geom& g,
                                          const Bc& bc, const CgFast& cf, int gr, int gc) {
#pragma clang fp contract(on)
  double fr[Q];
  if (INTERIOR) {
    cg_node_gather(fr, ft, in_r, in_b, g, gr, gc);
  } else {
    gather_bc(fr, in_r, g, bc, gr, gc);
    gather_bc(ft, in_b, g, bc, gr, gc);
  }
  return cg_node_reduce(fr, ft, cf);
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

// f(integral_constant<0>) ... f(integral_constant<N - 1>), in that order (C++17: no templated lambdas)
template <int N, class F>
__device__ __forceinline__ void cg_static_for(F&& f) {
  if constexpr (N > 0) {
    cg_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
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
  int tile_r, tile_c;
  if (xcd_swizzle >= 100) {
    // PATCHES: consecutive workgroups of one XCD (b % 8 = the XCD, b / 8 = its m-th workgroup) take the PR x PC tiles of one
    // patch, so that both the ring rows and the ring columns inside a patch are hits of that XCD's L2; the eight XCDs work on
    // eight neighbouring patches.  What does not fill a group of eight whole patches follows in plain order.
    const int PR = xcd_swizzle / 100, PC = xcd_swizzle % 100, P = PR * PC;
    const int tiles_r = gridDim.x / tiles_c, pr_n = tiles_r / PR, pc_n = tiles_c / PC;
    const int covered = (pr_n * pc_n / 8) * 8;  // whole patches in groups of eight
    if (tile < covered * P) {
      const int x = tile % 8, m = tile / 8, gp = (m / P) * 8 + x, j = m % P;
      tile_r = (gp / pc_n) * PR + j / PC;
      tile_c = (gp % pc_n) * PC + j % PC;
    } else {
      int k = tile - covered * P;
      const int left = pr_n * pc_n - covered;  // whole patches beyond the last group of eight
      if (k < left * P) {
        const int gp = covered + k / P, j = k % P;
        tile_r = (gp / pc_n) * PR + j / PC;
        tile_c = (gp % pc_n) * PC + j % PC;
      } else {
        k -= left * P;
        const int wr = tiles_c - pc_n * PC;  // tile columns right of the patches
        if (k < pr_n * PR * wr) {
          tile_r = k / wr;
          tile_c = pc_n * PC + k % wr;
        } else {
          k -= pr_n * PR * wr;
          tile_r = pr_n * PR + k / tiles_c;
          tile_c = k % tiles_c;
        }
      }
    }
  } else {
    if (xcd_swizzle > 1) {  // groups of G column-neighbour tiles per XCD inside a common window (as k_cg_fused)
      const int G = xcd_swizzle, x = tile % 8, m = tile / 8, win = 8 * G;
      const int t2 = (m / G) * win + x * G + (m % G);
      if ((m / G + 1) * win <= (int)gridDim.x) tile = t2;
    } else if (xcd_swizzle == 1) {
      const int per = gridDim.x / 8;
      if (tile < per * 8) tile = (tile % 8) * per + tile / 8;
    }
    tile_r = tile / tiles_c;
    tile_c = tile % tiles_c;
  }
  const int r_base = ra + tile_r * TR, c_base = ca + tile_c * TC;
  const int tr0 = threadIdx.x / TC, tc = threadIdx.x % TC;

  // Phase 2 collides the thread's nodes in the order j = 0 .. NPT - 1; node j is node kj[j] of the tile pass order.  PARK: j = 0
  // waits in registers (ft[0]), j >= 1 in LDS (s_park[j - 1]); otherwise all wait in registers (ft[j]).
  double ft[PARK ? 1 : NPT][Q], rr[NPT], rb[NPT];
  int kj[NPT];
  auto own = [&](int k, auto jc) {  // gather + reduce node k; it becomes phase 2's node j
    constexpr int j = decltype(jc)::value;
    const int tr = tr0 + k * RPP;
    double fk[Q];
    const CgNode me = cg_node<true>(fk, in_r, in_b, g, Bc{}, cf, r_base + tr, c_base + tc);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      if constexpr (PARK && j > 0) s_park[j - 1][q][threadIdx.x] = fk[q];
      else ft[PARK ? 0 : j][q] = fk[q];
    }
    rr[j] = me.rr;
    rb[j] = me.rb;
    kj[j] = k;
    s_psi[tr + 2][tc + 2] = me.psi;
    s_qx[tr + 2][tc + 2] = me.qx;
    s_qy[tr + 2][tc + 2] = me.qy;
  };
  auto ring_node = [&](int lr, int lc) {
    double tmp[Q];
    const CgNode nb = cg_node<true>(tmp, in_r, in_b, g, Bc{}, cf, r_base + lr - 2, c_base + lc - 2);
    s_psi[lr][lc] = nb.psi;
    s_qx[lr][lc] = nb.qx;
    s_qy[lr][lc] = nb.qy;
  };
  // (round 4, measured and not kept: tile rows of even index read top to bottom, odd ones bottom to top, so that vertically
  // adjacent tiles request the six rows they share at the same phase of their lives -- 16.44 - 16.59 k MLUPS against 16.61 - 16.72 k
  // for the plain order over six patch orders on one box, profiles/r04_cg_alt_sweep_ab.txt)
  {
    // plain order: the last pass first, pass 0 last (it is the one that stays in registers), then the ring
    cg_static_for<NPT>([&](auto kc) {
      constexpr int k = NPT - 1 - decltype(kc)::value;
      own(k, std::integral_constant<int, k>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    // the +-2 ring, ONE slot per thread (336 of the 512 threads at 16 x 64): two rows above, two below, then 2 + 2 columns beside each row
    constexpr int NH = LR * LC - TR * TC;
    for (int i = threadIdx.x; i < NH; i += NT) {
      if (i < 2 * LC) ring_node(i / LC, i % LC);
      else if (i < 4 * LC) ring_node(TR + 2 + (i - 2 * LC) / LC, (i - 2 * LC) % LC);
      else ring_node(2 + ((i - 4 * LC) >> 2), ((i - 4 * LC) & 3) < 2 ? ((i - 4 * LC) & 3) : TC + ((i - 4 * LC) & 3));
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    int tr = tr0 + kj[j] * RPP;
    // opaque per node: otherwise the 18 store addresses of the later nodes are derived from the first node's and wait
    // in registers (scratch at 128) through its collision
    asm volatile("" : "+v"(tr));
    // likewise the wave-uniform f64 products of the source term (c_q . Fg: VALU work, there is no scalar f64 unit) are
    // recomputed per node instead of waiting in ten register pairs
    CgFast cfk = cf;
    asm volatile("" : "+s"(cfk.Gr), "+s"(cfk.Gc));
    double fk[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) fk[q] = (PARK && j > 0) ? s_park[j > 0 ? j - 1 : 0][q][threadIdx.x] : ft[PARK ? 0 : j][q];
    CgNode me;  // == cg_node's expressions on the held sums
    me.rr = rr[j];
    me.rb = rb[j];
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
    if (j + 1 < NPT) __builtin_amdgcn_sched_barrier(0);
  }
}

#ifdef LBM_EXPERIMENTS  // two-phase step, the 16 x 64 tile walking down a chunk of rows (k_cg_walk_tile): csrc/experiments/cg_walk_tile.hpp
#include "experiments/cg_walk_tile.hpp"
#endif

#ifdef LBM_EXPERIMENTS  // two-phase step, merged frame + inner dispatch and the strip kernels of generations 1 - 4: csrc/experiments/cg_strips_1_4.hpp
#include "experiments/cg_strips_1_4.hpp"
#endif

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

#ifdef LBM_EXPERIMENTS  // two-phase step, strip kernel of generation 5 (adjacent private windows): csrc/experiments/cg_strip5.hpp
#include "experiments/cg_strip5.hpp"
#endif

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
    goff[q] = (unsigned)((q * g.plane - icx(q) * g.P - icy(q)) * 8);
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
    const unsigned v = (unsigned)((rn - (R0 - 3)) * g.P + dl) * 8u;
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
    const unsigned v_out = (unsigned)(m * g.P + l) * 8u;  // from (Rp, c_base - 2)
    const long o_out = g.at(Rp, c_base - 2);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      cg_st_nt(out_r[q], pn_r + o_out, poff[q] + v_out);
      cg_st_nt(out_b[q], pn_b + o_out, poff[q] + v_out);
    }
    if (WITH_FIELDS) {
      const long o = mi.at(Rp, c_base - 2), oo = (long)Rp * g.C + (c_base - 2);  // diagnostics carry no ghost rows
      const unsigned v_f = (unsigned)(m * g.C + l) * 8u;  // the fields are dense: rows C apart, whatever the lattice's pitch
      cg_st(of[0], rho_r_out + o, v_f);
      cg_st(of[1], rho_b_out + o, v_f);
      cg_st(of[2], u_out + o, v_f);
      cg_st(of[3], u_out + (mi.n + o), v_f);
      cg_st(of[4], psi_out + oo, v_f);
      cg_st(of[5], snu_out + oo, v_f);
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

#ifdef LBM_EXPERIMENTS  // two-phase step, two steps per pass (k_cg_two_step): csrc/experiments/cg_two_step.hpp
#include "experiments/cg_two_step.hpp"
#endif

}  // namespace lbm
