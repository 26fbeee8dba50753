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
//    populations' own) are not carried through the back-transform;
//  * 1 / feq_q = (1/rho)(1/psi_cx)(1/psi_cy): 7 reciprocals (v_rcp_f64 + one Newton step) and one
//    division instead of 12 IEEE divisions.  FMA contraction is on.
// ~330 f64 operations per node instead of ~940; agreement with the reference-order model to
// rounding (tests/test_gpu_kbc.py states the tolerance).
struct KbcFastModel {
  double s2, is2;  // is2 = 1 / s2, one IEEE division on the host instead of one per node
  __host__ __device__ explicit KbcFastModel(double s) : s2(s), is2(1.0 / s) {}
  static constexpr bool kFullStrips = true;  // VALU-bound in the sliding window: d2q9.hpp sw_strip_width

  __device__ __forceinline__ static double rcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  }

  __device__ __forceinline__ void collide(double (&f)[Q], double& rho, double& ux, double& uy) const {
    // contract(on): products feeding a sum IN THE SAME EXPRESSION become FMAs, decided by the front
    // end -- identical in every kernel this model is inlined into (single-step, sliding-window,
    // edge pass), which contract(fast) (back-end, context dependent) does not guarantee
#pragma clang fp contract(on)
    constexpr double cs2 = 1.0 / 3.0, cs4 = 1.0 / 9.0;
    // raw moments (butterfly)
    const double a = f[1] + f[3], b = f[2] + f[4], d57 = f[5] + f[7], d68 = f[6] + f[8];
    const double e57 = f[5] - f[7], e68 = f[6] - f[8];
    const double m22 = d57 + d68, m11 = d57 - d68, m21 = e57 + e68, m12 = e57 - e68;
    const double m20 = a + m22, m02 = b + m22;
    rho = (f[0] + a) + (b + m22);
    const double jx = (f[1] - f[3]) + m12, jy = (f[2] - f[4]) + m21;
    const double irho = rcp(rho);
    ux = jx * irho;
    uy = jy * irho;
    const double ux2 = ux * ux, uy2 = uy * uy, uxy = ux * uy;
    // central moments (binomial shift)
    const double k20 = m20 - jx * ux, k02 = m02 - jy * uy, C5 = m11 - jx * uy;
    const double C6 = (m21 - uy * m20) - 2.0 * ux * C5;
    const double C7 = (m12 - ux * m02) - 2.0 * uy * C5;
    const double C8 = ((m22 - 2.0 * (uy * m21 + ux * m12)) + (uy2 * m20 + ux2 * m02)) +
                      (4.0 * uxy * m11 - 3.0 * (jx * ux) * uy2);
    // S = M^-1 N^-1 (0,0,0,C3,C4,C5,0,0,0), C3 = k20 + k02, C4 = k20 - k02
    const double i6s = k20 * uy + 2.0 * C5 * ux, i7s = k02 * ux + 2.0 * C5 * uy;
    const double i8s = (k02 * ux2 + k20 * uy2) + 4.0 * C5 * uxy;
    double S[Q], H[Q];
    S[0] = i8s - (k20 + k02);
    S[1] = 0.5 * (k20 - (i7s + i8s));
    S[3] = 0.5 * (k20 + (i7s - i8s));
    S[2] = 0.5 * (k02 - (i6s + i8s));
    S[4] = 0.5 * (k02 + (i6s - i8s));
    S[5] = 0.25 * ((C5 + i8s) + (i6s + i7s));
    S[6] = 0.25 * ((i8s - C5) + (i6s - i7s));
    S[7] = 0.25 * ((C5 + i8s) - (i6s + i7s));
    S[8] = 0.25 * ((i8s - C5) - (i6s - i7s));
    // H = M^-1 N^-1 (0,...,0,C6,C7,C8)
    const double i8h = 2.0 * (C6 * uy + C7 * ux) + C8;
    H[0] = i8h;
    H[1] = -0.5 * (C7 + i8h);
    H[3] = 0.5 * (C7 - i8h);
    H[2] = -0.5 * (C6 + i8h);
    H[4] = 0.5 * (C6 - i8h);
    H[5] = 0.25 * (i8h + (C6 + C7));
    H[6] = 0.25 * (i8h + (C6 - C7));
    H[7] = 0.25 * (i8h - (C6 + C7));
    H[8] = 0.25 * (i8h - (C6 - C7));
    // product-form equilibrium and its reciprocal
    const double px0 = (1.0 - cs2) - ux2, pxp = 0.5 * ((ux2 + cs2) + ux), pxm = 0.5 * ((ux2 + cs2) - ux);
    const double py0 = (1.0 - cs2) - uy2, pyp = 0.5 * ((uy2 + cs2) + uy), pym = 0.5 * ((uy2 + cs2) - uy);
    const double rx0 = rho * px0, rxp = rho * pxp, rxm = rho * pxm;       // rho folded into the x factor
    const double ix0 = irho * rcp(px0), ixp = irho * rcp(pxp), ixm = irho * rcp(pxm);
    const double iy0 = rcp(py0), iyp = rcp(pyp), iym = rcp(pym);
    const double fe[Q] = {rx0 * py0, rxp * py0, rx0 * pyp, rxm * py0, rx0 * pym,
                          rxp * pyp, rxm * pyp, rxm * pym, rxp * pym};
    const double ie[Q] = {ix0 * iy0, ixp * iy0, ix0 * iyp, ixm * iy0, ix0 * iym,
                          ixp * iyp, ixm * iyp, ixm * iym, ixp * iym};
    // delta_h rows 5-8 as written in the reference: "ux2 + uy" where the algebra has ux2 * uy (Q8)
    const double qa = -0.25 * rho * ((ux2 + uy) - ux2 * uy), qb = -0.25 * rho * ((uy - ux2) + ux2 * uy);
    const double quirk[Q] = {0.0, 0.0, 0.0, 0.0, 0.0, qa, qa, qb, qb};
    double num = 0.0, den = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double ds = S[q] - fe[q], dh = (H[q] - fe[q]) + quirk[q];
      const double t = dh * ie[q];
      num += ds * t;
      den += dh * t;
    }
    const double gamma = is2 - (1.0 - is2) * (num * rcp(den));  // eval_gamma :138-148
    // relaxed populations: f - s2 (S - cs2 rho G) - gamma s2 (H - cs4 rho V8)
    const double g2 = ux2 + uy2, cr = cs2 * rho, gs = gamma * s2, hr = cs4 * rho;
    const double G[Q] = {g2 - 2.0,
                         -0.5 * ((g2 - 1.0) + ux), -0.5 * ((g2 - 1.0) + uy),
                         -0.5 * ((g2 - 1.0) - ux), -0.5 * ((g2 - 1.0) - uy),
                         0.25 * (g2 + (ux + uy)), 0.25 * (g2 - (ux - uy)),
                         0.25 * (g2 - (ux + uy)), 0.25 * (g2 + (ux - uy))};
    constexpr double V8[Q] = {1.0, -0.5, -0.5, -0.5, -0.5, 0.25, 0.25, 0.25, 0.25};
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q] = (f[q] - s2 * (S[q] - cr * G[q])) - gs * (H[q] - hr * V8[q]);
  }
};

}  // namespace lbm
