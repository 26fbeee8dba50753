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

}  // namespace lbm
