// Entropic KBC collision in central-moment space, one node per thread, all nine moments in
// registers (reference: ulbm::d2q9::kbc, src/ulbm.cpp:91-320, ~250 ATen launches and 19
// [R,C,9] work tensors per step there).  Polynomials are restated term by term INCLUDING the
// reference's "ux2+uy" terms in eval_delta_h rows 5-8.
#pragma once
#include "d2q9.hpp"

namespace lbm {

struct KbcModel {
  double s2;

  // ulbm.cpp:234-242 (== :252-260): equilibrium polynomials, to be scaled by m0
  __device__ __forceinline__ static void feq_poly(double (&e)[Q], double ux, double uy, double ux2,
                                                  double uy2) {
    const double cs2 = 1.0 / 3.0, cs4 = 1.0 / 9.0;  // ulbm.hpp:26-27
    e[0] = 2.0 * cs2 * (0.5 * ux2 + 0.5 * uy2 - 1.0) + cs4 + ux2 * uy2 - ux2 - uy2 + 1.0;
    e[1] = 0.5 * (-cs2 * (ux2 + uy2 + ux - 1.0) - cs4 - ux2 * uy2 + ux2 - uy2 * ux + ux);
    e[2] = 0.5 * (-cs2 * (ux2 + uy2 + uy - 1.0) - cs4 - ux2 * uy2 - ux2 * uy + uy2 + uy);
    e[3] = 0.5 * (-cs2 * (ux2 + uy2 - ux - 1.0) - cs4 - ux2 * uy2 + ux2 + uy2 * ux - ux);
    e[4] = 0.5 * (-cs2 * (ux2 + uy2 - uy - 1.0) - cs4 - ux2 * uy2 + ux2 * uy + uy2 - uy);
    e[5] = 0.25 * (cs2 * (ux2 + uy2 + ux + uy) + cs4 + ux2 * uy2 + ux2 * uy + uy2 * ux + ux * uy);
    e[6] = 0.25 * (cs2 * (ux2 + uy2 - ux + uy) + cs4 + ux2 * uy2 + ux2 * uy - uy2 * ux - ux * uy);
    e[7] = 0.25 * (cs2 * (ux2 + uy2 - ux - uy) + cs4 + ux2 * uy2 - ux2 * uy - uy2 * ux + ux * uy);
    e[8] = 0.25 * (cs2 * (ux2 + uy2 + ux - uy) + cs4 + ux2 * uy2 - ux2 * uy + uy2 * ux - ux * uy);
  }

  // kbc::collide() for one node given the moments the driver holds (m0, m1 = (ux, uy)).
  __device__ __forceinline__ void collide_with(double (&f)[Q], double m0, double ux, double uy) const {
    const double cs2 = 1.0 / 3.0, cs4 = 1.0 / 9.0;
    const double is2 = 1.0 / s2;
    const double ux2 = ux * ux, uy2 = uy * uy;  // eval_m1_components :150-155
    // eval_central_momenta :265-320
    double T[Q] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double cmx = (double)icx(q) - ux, cmy = (double)icy(q) - uy;
      const double cmx2 = cmx * cmx, cmy2 = cmy * cmy;
      T[0] += f[q];
      T[1] += f[q] * cmx;
      T[2] += f[q] * cmy;
      T[3] += f[q] * (cmx2 + cmy2);
      T[4] += f[q] * (cmx2 - cmy2);
      T[5] += f[q] * cmx * cmy;
      T[6] += f[q] * cmx2 * cmy;
      T[7] += f[q] * cmx * cmy2;
      T[8] += f[q] * cmx2 * cmy2;
    }
    const double C3 = T[3], C4 = T[4], C5 = T[5], C6 = T[6], C7 = T[7], C8 = T[8];
    const double D3 = C3 - 2.0 * cs2 * m0;
    double ds[Q], dh[Q], ie[Q];
    // eval_delta_s :157-192
    ds[0] = -0.5 * C4 * (ux2 - uy2) + 4.0 * C5 * ux * uy - cs4 * m0 - m0 * (ux2 * uy2 - ux2 - uy2 + 1) + D3 * (0.5 * ux2 + 0.5 * uy2 - 1.0);
    ds[1] = 0.25 * C4 * (ux2 - uy2 + ux + 1) - C5 * uy * (2.0 * ux + 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 + uy2 * ux - ux) - 0.25 * D3 * (ux2 + uy2 + ux - 1.0);
    ds[2] = -0.25 * C4 * (-ux2 + uy2 + uy + 1) - C5 * ux * (2.0 * uy + 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - uy2 + ux2 * uy - uy) - 0.25 * D3 * (ux2 + uy2 + uy - 1.0);
    ds[3] = 0.25 * C4 * (ux2 - uy2 - ux + 1) - C5 * uy * (2.0 * ux - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 - uy2 * ux + ux) - 0.25 * D3 * (ux2 + uy2 - ux - 1.0);
    ds[4] = 0.25 * C4 * (ux2 - uy2 + uy - 1) - C5 * ux * (2.0 * uy - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - uy2 - ux2 * uy + uy) - 0.25 * D3 * (ux2 + uy2 - uy - 1.0);
    ds[5] = -0.125 * C4 * (ux2 - uy2 + ux - uy) + C5 * (ux * uy + 0.5 * ux + 0.5 * uy + 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 * uy + uy2 * ux + ux * uy) + 0.125 * D3 * (ux2 + uy2 + ux + uy);
    ds[6] = 0.125 * C4 * (-ux2 + uy2 + ux + uy) + C5 * (ux * uy + 0.5 * ux - 0.5 * uy - 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 * uy - uy2 * ux - ux * uy) + 0.125 * D3 * (ux2 + uy2 - ux + uy);
    ds[7] = -0.125 * C4 * (ux2 - uy2 - ux + uy) + C5 * (ux * uy - 0.5 * ux - 0.5 * uy + 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 * uy - uy2 * ux + ux * uy) + 0.125 * D3 * (ux2 + uy2 - ux - uy);
    ds[8] = -0.125 * C4 * (ux2 - uy2 + ux + uy) + C5 * (ux * uy - 0.5 * ux + 0.5 * uy - 0.25) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 * uy + uy2 * ux - ux * uy) + 0.125 * D3 * (ux2 + uy2 + ux - uy);
    // eval_delta_h :194-228 (rows 5-8: "ux2 + uy" as written in the reference)
    dh[0] = 2.0 * C6 * uy + 2.0 * C7 * ux + C8 - 2.0 * cs2 * m0 * (0.5 * ux2 + 0.5 * uy2 - 1.0) - cs4 * m0 - m0 * (ux2 * uy2 - ux2 - uy2 + 1.0);
    dh[1] = -C6 * uy - C7 * (ux + 0.5) - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 + ux - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 + uy2 * ux - ux);
    dh[2] = -C6 * (uy + 0.5) - C7 * ux - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 + uy - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 + ux2 * uy - uy2 - uy);
    dh[3] = -C6 * uy - C7 * (ux - 0.5) - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 - ux - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 - uy2 * ux + ux);
    dh[4] = -C6 * (uy - 0.5) - C7 * ux - 0.5 * C8 + 0.5 * cs2 * m0 * (ux2 + uy2 - uy - 1.0) + 0.5 * cs4 * m0 + 0.5 * m0 * (ux2 * uy2 - ux2 * uy - uy2 + uy);
    dh[5] = C6 * (0.5 * uy + 0.25) + C7 * (0.5 * ux + 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 + ux + uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 + uy + uy2 * ux + ux * uy);
    dh[6] = C6 * (0.5 * uy + 0.25) + C7 * (0.5 * ux - 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 - ux + uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 + ux2 + uy - uy2 * ux - ux * uy);
    dh[7] = C6 * (0.5 * uy - 0.25) + C7 * (0.5 * ux - 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 - ux - uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 + uy - uy2 * ux + ux * uy);
    dh[8] = C6 * (0.5 * uy - 0.25) + C7 * (0.5 * ux + 0.25) + 0.25 * C8 - 0.25 * cs2 * m0 * (ux2 + uy2 + ux - uy) - 0.25 * cs4 * m0 - 0.25 * m0 * (ux2 * uy2 - ux2 + uy + uy2 * ux - ux * uy);
    // eval_iequilibrium :230-246
    feq_poly(ie, ux, uy, ux2, uy2);
#pragma unroll
    for (int q = 0; q < Q; ++q) ie[q] = 1.0 / (ie[q] * m0);
    // eval_gamma :138-148
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      num += ds[q] * dh[q] * ie[q];
      den += dh[q] * dh[q] * ie[q];
    }
    const double gamma = is2 - (1.0 - is2) * num / den;
    // collide :98-125
    T[0] += -m0;
    T[3] += -2.0 * cs2 * m0;
    T[8] += -cs4 * m0;
    const double gs = gamma * s2;
    T[0] *= 1.0; T[1] *= 1.0; T[2] *= 1.0;
    T[3] *= s2; T[4] *= s2; T[5] *= s2;
    T[6] *= gs; T[7] *= gs; T[8] *= gs;
    const double T0 = T[0], T1 = T[1], T2 = T[2], T3 = T[3], T4 = T[4], T5 = T[5], T6 = T[6],
                 T7 = T[7], T8 = T[8];
    const double i0 = T0;
    const double i1 = T0 * ux + T1;
    const double i2 = T0 * uy + T2;
    const double i3 = T0 * (ux2 + uy2) + 2.0 * T1 * ux + 2.0 * T2 * uy + T3;
    const double i4 = T0 * (ux2 - uy2) + 2.0 * T1 * ux - 2.0 * T2 * uy + T4;
    const double i5 = T0 * ux * uy + T1 * uy + T2 * ux + T5;
    const double i6 = T0 * ux2 * uy + 2.0 * T1 * ux * uy + T2 * ux2 + 0.5 * T3 * uy + 0.5 * T4 * uy + 2.0 * T5 * ux + T6;
    const double i7 = T0 * ux * uy2 + T1 * uy2 + 2.0 * T2 * ux * uy + 0.5 * T3 * ux - 0.5 * T4 * ux + 2.0 * T5 * uy + T7;
    const double i8 = T0 * ux2 * uy2 + 2.0 * T1 * ux * uy2 + 2.0 * T2 * ux2 * uy + 0.5 * T3 * (ux2 + uy2) - 0.5 * T4 * (ux2 - uy2) + 4.0 * T5 * ux * uy + 2.0 * T6 * uy + 2.0 * T7 * ux + T8;
    double o[Q];
    o[0] = i0 - i3 + i8;
    o[1] = 0.5 * i1 + 0.25 * i3 + 0.25 * i4 - 0.5 * i7 - 0.5 * i8;
    o[2] = 0.5 * i2 + 0.25 * i3 - 0.25 * i4 - 0.5 * i6 - 0.5 * i8;
    o[3] = -0.5 * i1 + 0.25 * i3 + 0.25 * i4 + 0.5 * i7 - 0.5 * i8;
    o[4] = -0.5 * i2 + 0.25 * i3 - 0.25 * i4 + 0.5 * i6 - 0.5 * i8;
    o[5] = 0.25 * (i5 + i6 + i7 + i8);
    o[6] = 0.25 * (-i5 + i6 - i7 + i8);
    o[7] = 0.25 * (i5 - i6 - i7 + i8);
    o[8] = 0.25 * (-i5 - i6 + i7 + i8);
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q] = o[q] * -1.0 + f[q];  // :123-125
  }

  // Model interface used by the stream+collide kernels: the moments the driver would hold
  // are recomputed from the streamed populations (ulbm_double_shear_flow.cpp:141-142).
  __device__ __forceinline__ void collide(double (&f)[Q], double& rho, double& ux, double& uy) const {
    double jx, jy;
    BgkModel::moments(f, rho, jx, jy);
    ux = jx / rho;
    uy = jy / rho;
    collide_with(f, rho, ux, uy);
  }
};

// The same collision, reassociated ("fast" path, default of the stream+collide entry points; the
// model above follows the reference operation by operation and is bit-identical to the oracle).
// Same mathematics as kbc::collide() with the populations' own moments:
//  * central moments through the raw moments (butterfly over opposite pairs) and the binomial
//    shift, instead of nine 9-term sums of products of (c - u) powers;
//  * eval_delta_s / eval_delta_h (ulbm.cpp:157-228) are, term by term,
//        delta_s = M^-1 N^-1 (0,0,0, C3, C4, C5, 0,0,0) - feq
//        delta_h = M^-1 N^-1 (0,...,0, C6, C7, C8)      - feq  (+ the "ux2 + uy" slip of rows 5-8, Q8)
//    so both come from ONE back-transform each and the product-form equilibrium
//    feq_q = rho psi_cx(ux) psi_cy(uy);  the relaxed populations reuse the same two vectors:
//        f' = f - s2 (S - cs2 rho G) - gamma s2 (H - cs4 rho V8);
//  * the conserved central moments T0, T1, T2 (zero up to rounding when the moments are the
//    populations' own) are not carried through the back-transform.
// Round 3 (the kernel is VALU-bound at one wave per SIMD, so only the instruction count moves it:
// 276 f64 operations + 8 v_rcp_f64 per collision before, profiles/r03_kbc_*):
//  * every vector is carried WITHOUT its lattice weight w_q = 1, 1/2, 1/4 (S = w S~, H = w H~, G = w G~,
//    feq = w rho D_i D_j): the 25 scalings by 1/2 and 1/4 disappear into three constants of the final update;
//  * delta_s~ = S~ - (rho D_i) D_j and delta_h~ likewise are single FMAs: feq is never formed;
//  * gamma's two sums (eval_gamma :138-148) are sums over the 3 x 3 velocity grid of  X_i Y_j ds~ dh~  with
//    X_i = w_i / D_i: 1/rho cancels between numerator and denominator, and (round 4) so does the product of the six
//    1 / D: X_i ~ w_i x the other two D of its axis, no reciprocal -- 2 reciprocals per node instead of 8.
// Round 4: 189 f64 operations + 2 v_rcp_f64 per collision (235 + 3 before): the k22 central moment is folded into H~_0, the
// weights of gamma's sums need no reciprocal, the final update forms S~ + gamma H~ first.
// Agreement with the reference-order model to rounding (tests/test_gpu_kbc.py states the tolerance).
struct KbcFastModel {
  double s2, is2, hs2, qs2;  // is2 = 1 / s2 (one IEEE division on the host instead of one per node); s2 / 2, s2 / 4
  __host__ __device__ explicit KbcFastModel(double s) : s2(s), is2(1.0 / s), hs2(0.5 * s), qs2(0.25 * s) {}
  // Strips of 56 columns, not of all 64 - 2 (D - 1) valid ones (round 4): with 60-column strips every stored row segment starts
  // 16 bytes off a 32-byte sector and the launch sits on a memory-side floor of 38 ps per node whatever its arithmetic (2 or 3
  // steps, 235 or 196 operations per collision: 0.65 - 0.68 ms per 4096^2 launch); 448-byte segments run at the BGK windows'
  // 31 ps per node (2 steps: 0.53 ms, 3 steps: 0.59 ms = 85 k instead of 77.5 k MLUPS; profiles/r04_kbc_strip_width.txt)
  static constexpr bool kFullStrips = false;  // d2q9.hpp sw_strip_width

  __device__ __forceinline__ static double rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  }

  __device__ __forceinline__ void collide(double (&f)[Q], double& rho, double& ux, double& uy) const {
    // contract(on): products feeding a sum IN THE SAME EXPRESSION become FMAs, decided by the front
    // end -- identical in every kernel this model is inlined into (single-step, sliding-window,
    // edge pass), which contract(fast) (back-end, context dependent) does not guarantee
#pragma clang fp contract(on)
    constexpr double cs2 = 1.0 / 3.0;
    // raw moments (butterfly)
    const double a = f[1] + f[3], b = f[2] + f[4], d57 = f[5] + f[7], d68 = f[6] + f[8];
    const double e57 = f[5] - f[7], e68 = f[6] - f[8];
    const double m22 = d57 + d68, m11 = d57 - d68, m21 = e57 + e68, m12 = e57 - e68;
    const double m20 = a + m22, m02 = b + m22;
    rho = (f[0] + b) + m20;
    const double jx = (f[1] - f[3]) + m12, jy = (f[2] - f[4]) + m21;
    const double irho = rcp(rho);
    ux = jx * irho;
    uy = jy * irho;
    const double ux2 = ux * ux, uy2 = uy * uy;
    // central moments (binomial shift)
    const double k20 = m20 - jx * ux, k02 = m02 - jy * uy, C5 = m11 - jx * uy;
    const double z6 = (2.0 * C5) * ux, z7 = (2.0 * C5) * uy;  // shared by C6 / C7 and the S~ block
    const double C6 = (m21 - uy * m20) - z6;
    const double C7 = (m12 - ux * m02) - z7;
    // (C8, the central moment k22 = m22 - 2 (ux C7 + uy C6) - i8s - rho ux^2 uy^2 with the i8s of the S~ block below, is
    // never formed: it enters only through H~_0 = 2 (C6 uy + C7 ux) + C8, where the first bracket cancels -- round 4)
    // S~ = M^-1 N^-1 (0,0,0,C3,C4,C5,0,0,0) / w, C3 = k20 + k02, C4 = k20 - k02
    const double i6s = k20 * uy + z6, i7s = k02 * ux + z7;
    const double i8s = ux * i7s + uy * i6s;  // = k02 ux^2 + k20 uy^2 + 4 C5 ux uy
    const double c5p = C5 + i8s, c5m = i8s - C5, s67 = i6s + i7s, d67 = i6s - i7s;
    const double A8 = k20 - i8s, B8 = k02 - i8s;
    const double S0 = -(A8 + k02);
    const double S1 = A8 - i7s, S3 = A8 + i7s;
    const double S2 = B8 - i6s, S4 = B8 + i6s;
    const double S5 = c5p + s67, S6 = c5m + d67, S7 = c5p - s67, S8 = c5m - d67;
    // H~ = M^-1 N^-1 (0,...,0,C6,C7,C8) / w
    const double w20 = jx * ux;  // rho ux^2
    const double H0 = (m22 - i8s) - w20 * uy2;
    const double csum = C6 + C7, cdif = C6 - C7;
    const double H1 = -(C7 + H0), H3 = C7 - H0, H2 = -(C6 + H0), H4 = C6 - H0;
    const double H5 = H0 + csum, H6 = H0 + cdif, H7 = H0 - csum, H8 = H0 - cdif;
    // product-form equilibrium feq_q = w_q rho D_i D_j: D_0 = 2/3 - u^2, D_+- = (u^2 + 1/3) +- u
    const double bx = ux2 + cs2, by = uy2 + cs2;
    const double Dx0 = (1.0 - cs2) - ux2, Dxp = bx + ux, Dxm = bx - ux;
    const double Dy0 = (1.0 - cs2) - uy2, Dyp = by + uy, Dym = by - uy;
    const double rx0 = rho * Dx0, rxp = rho * Dxp, rxm = rho * Dxm;
    // delta_s~, delta_h~ (rows 5-8 of delta_h as written in the reference: "ux2 + uy" where the algebra has ux2 * uy, Q8)
    // qa = rho ((ux2 + uy) - ux2 uy), qb = rho ((uy - ux2) + ux2 uy): rho uy = jy, rho ux2 = jx ux
    const double qd = w20 - w20 * uy, qa = jy + qd, qb = jy - qd;
    const double ds0 = S0 - rx0 * Dy0, ds1 = S1 - rxp * Dy0, ds2 = S2 - rx0 * Dyp, ds3 = S3 - rxm * Dy0, ds4 = S4 - rx0 * Dym;
    const double ds5 = S5 - rxp * Dyp, ds6 = S6 - rxm * Dyp, ds7 = S7 - rxm * Dym, ds8 = S8 - rxp * Dym;
    const double dh0 = H0 - rx0 * Dy0, dh1 = H1 - rxp * Dy0, dh2 = H2 - rx0 * Dyp, dh3 = H3 - rxm * Dy0, dh4 = H4 - rx0 * Dym;
    const double dh5 = (H5 - rxp * Dyp) - qa, dh6 = (H6 - rxm * Dyp) - qa, dh7 = (H7 - rxm * Dym) - qb, dh8 = (H8 - rxp * Dym) - qb;
    // X_i = w_i / D_i, Y_j = w_j / D_j enter only through the RATIO of two sums weighted by X_i Y_j: the common factor
    // 1 / (Dx0 Dxp Dxm Dy0 Dyp Dym) cancels like 1 / rho does, X_i ~ w_i prod_{k != i} D_k needs no reciprocal (round 4)
    const double X0 = Dxp * Dxm, hx = 0.5 * Dx0, Xp = hx * Dxm, Xm = hx * Dxp;
    const double Y0 = Dyp * Dym, hy = 0.5 * Dy0, Yp = hy * Dym, Ym = hy * Dyp;
    // eval_gamma :138-148: num / den = sum X_i Y_j ds~ dh~ / sum X_i Y_j dh~^2
    const double t0 = dh0 * Y0, t1 = dh1 * Y0, t3 = dh3 * Y0;
    const double t2 = dh2 * Yp, t5 = dh5 * Yp, t6 = dh6 * Yp;
    const double t4 = dh4 * Ym, t8 = dh8 * Ym, t7 = dh7 * Ym;
    const double n0 = (ds0 * t0 + ds2 * t2) + ds4 * t4, e0 = (dh0 * t0 + dh2 * t2) + dh4 * t4;
    const double np = (ds1 * t1 + ds5 * t5) + ds8 * t8, ep = (dh1 * t1 + dh5 * t5) + dh8 * t8;
    const double nm = (ds3 * t3 + ds6 * t6) + ds7 * t7, em = (dh3 * t3 + dh6 * t6) + dh7 * t7;
    const double num = (X0 * n0 + Xp * np) + Xm * nm, den = (X0 * e0 + Xp * ep) + Xm * em;
    const double gamma = is2 - (1.0 - is2) * (num * rcp(den));
    // relaxed populations: f - s2 w (S~ - cs2 rho G~) - gamma s2 w (H~ - cs4 rho V8~), V8~ = (1, -1 x 4, 1 x 4)
    //                    = f - s2 w ((S~ + gamma H~) - cs2 rho (G~ + (gamma / 3) V8~))          (cs4 = cs2 / 3)
    // (round 4: four operations per population instead of five, gamma s2 w and cs4 rho are never formed)
    const double g2 = ux2 + uy2, us = ux + uy, ud = ux - uy;
    const double cr = cs2 * rho, g3 = cs2 * gamma;
    const double k5 = g2 + g3, k1 = k5 - 1.0, k0 = k5 - 2.0;
    f[0] = f[0] - s2 * ((S0 + gamma * H0) - cr * k0);
    f[1] = f[1] - hs2 * ((S1 + gamma * H1) + cr * (k1 + ux));
    f[2] = f[2] - hs2 * ((S2 + gamma * H2) + cr * (k1 + uy));
    f[3] = f[3] - hs2 * ((S3 + gamma * H3) + cr * (k1 - ux));
    f[4] = f[4] - hs2 * ((S4 + gamma * H4) + cr * (k1 - uy));
    f[5] = f[5] - qs2 * ((S5 + gamma * H5) - cr * (k5 + us));
    f[6] = f[6] - qs2 * ((S6 + gamma * H6) - cr * (k5 - ud));
    f[7] = f[7] - qs2 * ((S7 + gamma * H7) - cr * (k5 - us));
    f[8] = f[8] - qs2 * ((S8 + gamma * H8) - cr * (k5 + ud));
  }
};

}  // namespace lbm
