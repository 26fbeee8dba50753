// D2Q9 device core for gfx950: lattice constants, SoA indexing, the pull-streaming gather
// (fast interior path + boundary path) and the model-templated stream+collide kernels.
//
// Conventions (SURVEY.md naming trap): W = the reference's solver::E (weights,
// src/solver.cpp:12-16); (CX, CY) = solver::c rows 0/1 (:18-21).  CX pairs with the row
// index r, CY with the column index c.  All arithmetic is f64 and the translation unit is
// built with -ffp-contract=off: every expression below is evaluated in the order the
// reference writes it, so results agree with the CPU oracle to the last bit wherever the
// oracle itself is order-deterministic.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include "../../include/lbm_hip.h"

namespace lbm {

constexpr int Q = 9;

#define LBM_W0 (4.0 / 9.0)
#define LBM_WS (1.0 / 9.0)
#define LBM_WD (1.0 / 36.0)

__device__ __forceinline__ constexpr double wq(int q) {
  return q == 0 ? LBM_W0 : (q < 5 ? LBM_WS : LBM_WD);
}
__host__ __device__ __forceinline__ constexpr int icx(int q) {
  return (q == 1 || q == 5 || q == 8) ? 1 : ((q == 3 || q == 6 || q == 7) ? -1 : 0);
}
__host__ __device__ __forceinline__ constexpr int icy(int q) {
  return (q == 2 || q == 5 || q == 6) ? 1 : ((q == 4 || q == 7 || q == 8) ? -1 : 0);
}
__host__ __device__ __forceinline__ constexpr int opp(int q) {
  return q == 0 ? 0 : (q == 1 ? 3 : (q == 2 ? 4 : (q == 3 ? 1 : (q == 4 ? 2 : (q == 5 ? 7 : (q == 6 ? 8 : (q == 7 ? 5 : 6)))))));
}

// Geometry of one SoA lattice (device copy of lbm_geom + derived strides).
struct Geom {
  int R, C, ghost;
  long plane;  // doubles per population plane >= (R + 2*ghost) * P
  int P;       // doubles per row of a plane (row pitch) >= C: lbm_geom.row_pitch; C itself stays the WIDTH of the lattice
  __host__ __device__ long at(int r, int c) const { return (long)(r + ghost) * P + c; }
};
inline Geom make_geom(const lbm_geom& g) {
  const int P = g.row_pitch > 0 ? g.row_pitch : g.C;
  const long dense = (long)(g.R + 2 * g.ghost) * P;
  return Geom{g.R, g.C, g.ghost, g.plane_stride > 0 ? (long)g.plane_stride : dense, P};
}

struct Bc {
  int row_lo, row_hi, col_lo, col_hi, pressure_rows;
  double rho_inlet, rho_outlet, uw_r, uw_c;
};
inline Bc make_bc(const lbm_bc* b) {
  if (!b) return Bc{0, 0, 0, 0, 0, 1.0, 1.0, 0.0, 0.0};
  return Bc{b->row_lo, b->row_hi, b->col_lo, b->col_hi, b->pressure_rows,
            b->rho_inlet, b->rho_outlet, b->uw_r, b->uw_c};
}
inline bool bc_needs_edge_pass(const Bc& b) {
  auto active = [](int m) { return m != LBM_EDGE_PERIODIC && m != LBM_EDGE_HALO; };
  return active(b.row_lo) || active(b.row_hi) || active(b.col_lo) || active(b.col_hi);
}

// ---------------------------------------------------------------------------------------
// Streaming (pull form of solver::advect, src/solver.cpp:76-131):
//   g_q(r, c) = f_q((r - cx_q) mod R, (c - cy_q) mod C)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int wrap_row(const Geom& g, int r) {
  if (g.ghost) return r;  // ghost rows -1 and R exist
  return r < 0 ? r + g.R : (r >= g.R ? r - g.R : r);
}
__device__ __forceinline__ int wrap_col(const Geom& g, int c) {
  return c < 0 ? c + g.C : (c >= g.C ? c - g.C : c);
}

// Generic gather of the 9 incoming populations of node (r, c), including every boundary
// fix-up the reference drivers apply after advect().  Slow path: used for
// boundary nodes only and by the small-lattice kernels.
__device__ inline void gather_bc(double (&f)[Q], const double* __restrict__ p, const Geom& g,
                                 const Bc& bc, int r, int c) {
#pragma unroll
  for (int q = 0; q < Q; ++q)
    f[q] = p[q * g.plane + g.at(wrap_row(g, r - icx(q)), wrap_col(g, c - icy(q)))];
  const long own = g.at(r, c);
#define OWN(q) p[(q) * g.plane + own]
  // rows first (cylinder_test.cpp applies its inlet/outlet rows before the side walls) ...
  const bool lo = (r == 0), hi = (r == g.R - 1);
  if (lo && bc.row_lo == LBM_EDGE_BOUNCE_BACK) {  // mrtcg_rayleigh_taylor.cpp:529-531
    f[1] = OWN(3); f[5] = OWN(7); f[8] = OWN(6);
  }
  if (hi && bc.row_hi == LBM_EDGE_BOUNCE_BACK) {  // :525-527
    f[3] = OWN(1); f[7] = OWN(5); f[6] = OWN(8);
  }
  if ((lo && bc.row_lo == LBM_EDGE_ABB_VELOCITY) || (hi && bc.row_hi == LBM_EDGE_ABB_VELOCITY)) {
    // cylinder_test.cpp:135-154: f[opp(q)] = -f_coll[q] + (2 + 9 (c_q.u_w)^2 - 3 u_w.u_w) w_q
    const double uu = bc.uw_r * bc.uw_r + bc.uw_c * bc.uw_c;
#pragma unroll
    for (int q = 1; q < Q; ++q) {
      const double cu = bc.uw_r * (double)icx(q) + bc.uw_c * (double)icy(q);
      const double abb = (2.0 + 9.0 * (cu * cu) - 3.0 * uu) * wq(q);
      f[opp(q)] = -OWN(q) + abb;
    }
  }
  // ... then columns (they win at the corners, as in the drivers)
  if (c == g.C - 1) {
    if (bc.col_hi == LBM_EDGE_BOUNCE_BACK) {  // horizontal_poiseuille_test.cpp:146-148
      f[4] = OWN(2); f[7] = OWN(5); f[8] = OWN(6);
    } else if (bc.col_hi == LBM_EDGE_SPECULAR) {  // cylinder_test.cpp:157-159
      f[4] = OWN(2); f[7] = OWN(6); f[8] = OWN(5);
    } else if (bc.col_hi == LBM_EDGE_WRAP_NOSHIFT) {  // mrtcg_rayleigh_taylor.cpp:521-523
      const bool skip = (lo && bc.row_lo != LBM_EDGE_HALO) || (hi && bc.row_hi != LBM_EDGE_HALO);
      if (!skip) {
        const long o = g.at(r, 0);
        f[4] = p[4 * g.plane + o]; f[8] = p[8 * g.plane + o]; f[7] = p[7 * g.plane + o];
      }
    }
  }
  if (c == 0) {
    if (bc.col_lo == LBM_EDGE_BOUNCE_BACK) {  // :150-152
      f[2] = OWN(4); f[5] = OWN(7); f[6] = OWN(8);
    } else if (bc.col_lo == LBM_EDGE_SPECULAR) {  // cylinder_test.cpp:161-163
      f[2] = OWN(4); f[5] = OWN(8); f[6] = OWN(7);
    } else if (bc.col_lo == LBM_EDGE_WRAP_NOSHIFT) {  // mrtcg_rayleigh_taylor.cpp:517-519
      const bool skip = (lo && bc.row_lo != LBM_EDGE_HALO) || (hi && bc.row_hi != LBM_EDGE_HALO);
      if (!skip) {
        const long o = g.at(r, g.C - 1);
        f[2] = p[2 * g.plane + o]; f[5] = p[5 * g.plane + o]; f[6] = p[6 * g.plane + o];
      }
    }
  }
#undef OWN
}

// The boundary fix-ups of gather_bc that only need the node's OWN post-collision populations
// (bounce-back, specular, anti-bounce-back velocity): same order -- rows first, columns win at the
// corners.  Used by the multi-step kernels, where levels 2..D take `own` from the register ring.
__device__ __forceinline__ void bc_fixups_own(double (&f)[Q], const double (&own)[Q], const Geom& g,
                                              const Bc& bc, int r, int c) {
  const bool lo = (r == 0), hi = (r == g.R - 1);
  if (lo && bc.row_lo == LBM_EDGE_BOUNCE_BACK) {
    f[1] = own[3]; f[5] = own[7]; f[8] = own[6];
  }
  if (hi && bc.row_hi == LBM_EDGE_BOUNCE_BACK) {
    f[3] = own[1]; f[7] = own[5]; f[6] = own[8];
  }
  if ((lo && bc.row_lo == LBM_EDGE_ABB_VELOCITY) || (hi && bc.row_hi == LBM_EDGE_ABB_VELOCITY)) {
    const double uu = bc.uw_r * bc.uw_r + bc.uw_c * bc.uw_c;
#pragma unroll
    for (int q = 1; q < Q; ++q) {
      const double cu = bc.uw_r * (double)icx(q) + bc.uw_c * (double)icy(q);
      const double abb = (2.0 + 9.0 * (cu * cu) - 3.0 * uu) * wq(q);
      f[opp(q)] = -own[q] + abb;
    }
  }
  if (c == g.C - 1) {
    if (bc.col_hi == LBM_EDGE_BOUNCE_BACK) {
      f[4] = own[2]; f[7] = own[5]; f[8] = own[6];
    } else if (bc.col_hi == LBM_EDGE_SPECULAR) {
      f[4] = own[2]; f[7] = own[6]; f[8] = own[5];
    }
  }
  if (c == 0) {
    if (bc.col_lo == LBM_EDGE_BOUNCE_BACK) {
      f[2] = own[4]; f[5] = own[7]; f[6] = own[8];
    } else if (bc.col_lo == LBM_EDGE_SPECULAR) {
      f[2] = own[4]; f[5] = own[8]; f[6] = own[7];
    }
  }
}
// gather_bc for the wall modes above, fully inlined (the out-of-line gather_bc would force the
// caller's population arrays -- and with them the register ring -- into scratch memory)
__device__ __forceinline__ void gather_walls(double (&f)[Q], const double* __restrict__ p, const Geom& g,
                                             const Bc& bc, int r, int c) {
  double own[Q];
  const long o = g.at(r, c);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    f[q] = p[q * g.plane + g.at(wrap_row(g, r - icx(q)), wrap_col(g, c - icy(q)))];
    own[q] = p[q * g.plane + o];
  }
  bc_fixups_own(f, own, g, bc, r, c);
}
// boundary modes the multi-step kernels can carry (no pressure rows, no same-row column copy)
__host__ __device__ inline bool bc_is_wall(int m) {
  return m == LBM_EDGE_BOUNCE_BACK || m == LBM_EDGE_SPECULAR || m == LBM_EDGE_ABB_VELOCITY;
}

// An axis with a wall on one side and PERIODIC on the other: the single-step gather handles it, the
// multi-step window does not (its column clamp / row wrap are per axis, not per side) -- launchers
// reject it and the solver falls back to single steps.  Wall + HALO (a chain-end slab) is fine.
inline bool bc_mixed_axis(const Bc& b) {
  auto mixed = [](int lo, int hi) {
    return (bc_is_wall(lo) && hi == LBM_EDGE_PERIODIC) || (bc_is_wall(hi) && lo == LBM_EDGE_PERIODIC);
  };
  return mixed(b.row_lo, b.row_hi) || mixed(b.col_lo, b.col_hi);
}

__device__ __forceinline__ bool is_edge_node(const Geom& g, int r, int c) {
  return r == 0 || r == g.R - 1 || c == 0 || c == g.C - 1;
}

// ---------------------------------------------------------------------------------------
// BGK collision model (solver.cpp:23-74 per node)
// ---------------------------------------------------------------------------------------
struct BgkModel {
  double omega;
  int incompressible;
  int delta_form;  // see lbm_bgk_params
  int force_mode;  // 1: gravity_test.cpp body force
  double Fr, Fc, ga, gb;

  __device__ __forceinline__ static void moments(const double (&f)[Q], double& rho, double& jx,
                                                 double& jy) {
    // calc_rho: sum over q in index order.  matmul(f, c^T): the zero products add +-0.
    rho = ((((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7]) + f[8]);
    jx = ((((f[1] - f[3]) + f[5]) - f[6]) - f[7]) + f[8];
    jy = ((((f[2] - f[4]) + f[5]) + f[6]) - f[7]) - f[8];
  }
  __device__ __forceinline__ static void feq_comp(double (&e)[Q], double rho, double ux, double uy) {
    const double u_u = ux * ux + uy * uy;  // solver.cpp:57
    const double cu[Q] = {0.0, ux, uy, -ux, -uy, ux + uy, -ux + uy, -ux - uy, ux - uy};
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double A = 1.0 + 3.0 * cu[q] + 4.5 * (cu[q] * cu[q]) - 1.5 * u_u;  // :60
      e[q] = (rho * A) * wq(q);                                                // :61
    }
  }
  __device__ __forceinline__ static void feq_incomp(double (&e)[Q], double rho, double ux, double uy) {
    const double cu[Q] = {0.0, ux, uy, -ux, -uy, ux + uy, -ux + uy, -ux - uy, ux - uy};
#pragma unroll
    for (int q = 0; q < Q; ++q) e[q] = (rho + 3.0 * cu[q]) * wq(q);  // solver.cpp:47-48
  }
  __device__ __forceinline__ void feq(double (&e)[Q], double rho, double ux, double uy) const {
    if (incompressible) feq_incomp(e, rho, ux, uy);
    else feq_comp(e, rho, ux, uy);
  }
  // in: f = pre-collision populations; out: f = post-collision, moments, equilibrium
  __device__ __forceinline__ void collide(double (&f)[Q], double& rho, double& ux, double& uy,
                                          double (&e)[Q]) const {
    double jx, jy;
    moments(f, rho, jx, jy);
    if (incompressible) {  // calc_incomp_u, solver.cpp:28-31
      ux = jx;
      uy = jy;
    } else {  // calc_u, :34-37
      ux = jx / rho;
      uy = jy / rho;
    }
    if (force_mode) {  // gravity_test.cpp:146  u += Fg.t()
      ux += Fr;
      uy += Fc;
    }
    feq(e, rho, ux, uy);
    if (force_mode) {  // gravity_test.cpp:151-160: f + (-omega (f - feq)) + S
      const double uF = ux * Fr + uy * Fc;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const double cu = ux * (double)icx(q) + uy * (double)icy(q);
        const double cF = Fr * (double)icx(q) + Fc * (double)icy(q);
        const double S = ((1 - 0.5 * omega) * ((ga + gb * cu) * cF - ga * uF) * wq(q));
        f[q] = f[q] + (-omega * (f[q] - e[q])) + S;
      }
    } else if (delta_form) {
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = f[q] + (-omega * (f[q] - e[q]));  // cylinder_test.cpp:108,123
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = (1.0 - omega) * f[q] + omega * e[q];  // :73
    }
  }
  __device__ __forceinline__ void collide(double (&f)[Q], double& rho, double& ux, double& uy) const {
    double e[Q];
    collide(f, rho, ux, uy, e);
  }
};

// Compile-time specialisation of the force-free BGK model (same arithmetic, no runtime mode
// branches): shrinks the deeply unrolled multi-step kernels.
template <int INCOMP, int DELTA>
struct BgkModelT {
  double omega;
  __device__ __forceinline__ void collide(double (&f)[Q], double& rho, double& ux, double& uy) const {
    double jx, jy, e[Q];
    BgkModel::moments(f, rho, jx, jy);
    if (INCOMP) {
      ux = jx;
      uy = jy;
      BgkModel::feq_incomp(e, rho, ux, uy);
    } else {
      ux = jx / rho;
      uy = jy / rho;
      BgkModel::feq_comp(e, rho, ux, uy);
    }
    if (DELTA) {
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = f[q] + (-omega * (f[q] - e[q]));
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = (1.0 - omega) * f[q] + omega * e[q];
    }
  }
};

// The compressible BGK collision reassociated (opt-in, tuning "bgk_fast" = 1; the models above
// follow solver.cpp operation by operation and are bit-identical to the oracle): pairwise moment
// sums, u = j * (1/rho) with v_rcp_f64 + one Newton step instead of two IEEE divisions, the
// equilibrium split into its parts even / odd under c -> -c, relaxation as f + omega (feq - f),
// FMA contraction per expression (identical in every kernel it is inlined into).  ~75 f64
// operations per node instead of ~130; agreement with the reference order to rounding.
struct BgkFastModel {
  double omega, keep, ow0, ow1, ow5;  // 1 - omega and omega * w_q, formed once on the host
  __host__ __device__ explicit BgkFastModel(double om)
      : omega(om), keep(1.0 - om), ow0(om * (4.0 / 9.0)), ow1(om * (1.0 / 9.0)), ow5(om * (1.0 / 36.0)) {}
  __device__ __forceinline__ void collide(double (&f)[Q], double& rho, double& ux, double& uy) const {
#pragma clang fp contract(on)
    const double a = f[1] + f[3], b = f[2] + f[4], d57 = f[5] + f[7], d68 = f[6] + f[8];
    const double e57 = f[5] - f[7], e68 = f[6] - f[8];
    rho = (f[0] + a) + (b + (d57 + d68));
    const double jx = (f[1] - f[3]) + (e57 - e68), jy = (f[2] - f[4]) + (e57 + e68);
    double ir = __builtin_amdgcn_rcp(rho);
    ir = __builtin_fma(ir, __builtin_fma(-rho, ir, 1.0), ir);
    ux = jx * ir;
    uy = jy * ir;
    const double us = ux + uy, ud = ux - uy;
    const double base = 1.0 - 1.5 * (ux * ux + uy * uy);
    // relaxation as (1 - omega) f + omega feq with omega folded into the weights: one add + one FMA
    // per population instead of two adds + one FMA for f + omega (feq - f)
    const double r1 = ow1 * rho, r5 = ow5 * rho;
    const double E1 = r1 * (base + 4.5 * ux * ux), E2 = r1 * (base + 4.5 * uy * uy);
    const double E5 = r5 * (base + 4.5 * us * us), E6 = r5 * (base + 4.5 * ud * ud);
    const double O1 = 3.0 * r1 * ux, O2 = 3.0 * r1 * uy, O5 = 3.0 * r5 * us, O8 = 3.0 * r5 * ud;
    f[0] = keep * f[0] + (ow0 * rho) * base;
    f[1] = keep * f[1] + (E1 + O1);
    f[3] = keep * f[3] + (E1 - O1);
    f[2] = keep * f[2] + (E2 + O2);
    f[4] = keep * f[4] + (E2 - O2);
    f[5] = keep * f[5] + (E5 + O5);
    f[7] = keep * f[7] + (E5 - O5);
    f[8] = keep * f[8] + (E6 + O8);
    f[6] = keep * f[6] + (E6 - O8);
  }
};

// ---------------------------------------------------------------------------------------
// Kernels
// ---------------------------------------------------------------------------------------
typedef double dbl2 __attribute__((ext_vector_type(2)));
// 16-byte vector with only 8-byte alignment: the +-1-column shifted reads of the pull step
typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));

template <bool NT>
__device__ __forceinline__ void store2(double* p, double a, double b) {
  dbl2 v = {a, b};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(p));
  else *reinterpret_cast<dbl2*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ dbl2 load2a(const double* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(p));
  return *reinterpret_cast<const dbl2*>(p);
}
template <bool NT>
__device__ __forceinline__ dbl2 load2u(const double* p) {
  if (NT) {
    dbl2u v = __builtin_nontemporal_load(reinterpret_cast<const dbl2u*>(p));
    return dbl2{v.x, v.y};
  }
  dbl2u v = *reinterpret_cast<const dbl2u*>(p);
  return dbl2{v.x, v.y};
}

// Fused pull step, interior fast path.  One thread updates two column-adjacent nodes
// (16 B per lane per population); a block walks (row, column-tile) work items with a
// grid-stride loop.  Column periodicity is resolved per thread (only the first and last
// thread of a row leave the vector path); row neighbours come from wrap_row (single block
// of rows, periodic) or from the ghost rows (slab).  Boundary fix-ups are NOT applied
// here: k_edge_stream_collide recomputes the edge nodes afterwards.
// Requires C % 2 == 0.
template <class Model, bool NT_LOAD, bool NT_STORE, bool WITH_MOMENTS>
__global__ __launch_bounds__(256) void k_stream_collide_v2(
    double* __restrict__ pn, const double* __restrict__ po, Geom g, Model m, int row_begin,
    int row_end, int tiles_per_row, double* __restrict__ rho_out, double* __restrict__ u_out) {
  const long items = (long)(row_end - row_begin) * tiles_per_row;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int r = row_begin + (int)(it / tiles_per_row);
    const int c = ((int)(it % tiles_per_row) * 256 + threadIdx.x) * 2;
    if (c >= g.C) continue;
    const long rm = g.at(wrap_row(g, r - 1), 0);  // source row of cx = +1 populations
    const long r0 = g.at(r, 0);
    const long rp = g.at(wrap_row(g, r + 1), 0);  // source row of cx = -1 populations
    double a[Q], b[Q];                            // node (r, c) and node (r, c + 1)
    dbl2 v;
    v = load2a<NT_LOAD>(po + 0 * g.plane + r0 + c); a[0] = v.x; b[0] = v.y;
    v = load2a<NT_LOAD>(po + 1 * g.plane + rm + c); a[1] = v.x; b[1] = v.y;
    v = load2a<NT_LOAD>(po + 3 * g.plane + rp + c); a[3] = v.x; b[3] = v.y;
    if (c > 0 && c + 2 < g.C) {
      v = load2u<NT_LOAD>(po + 2 * g.plane + r0 + c - 1); a[2] = v.x; b[2] = v.y;
      v = load2u<NT_LOAD>(po + 5 * g.plane + rm + c - 1); a[5] = v.x; b[5] = v.y;
      v = load2u<NT_LOAD>(po + 6 * g.plane + rp + c - 1); a[6] = v.x; b[6] = v.y;
      v = load2u<NT_LOAD>(po + 4 * g.plane + r0 + c + 1); a[4] = v.x; b[4] = v.y;
      v = load2u<NT_LOAD>(po + 7 * g.plane + rp + c + 1); a[7] = v.x; b[7] = v.y;
      v = load2u<NT_LOAD>(po + 8 * g.plane + rm + c + 1); a[8] = v.x; b[8] = v.y;
    } else {
      const int cm = wrap_col(g, c - 1), cp = wrap_col(g, c + 2);
      a[2] = po[2 * g.plane + r0 + cm]; b[2] = po[2 * g.plane + r0 + c];
      a[5] = po[5 * g.plane + rm + cm]; b[5] = po[5 * g.plane + rm + c];
      a[6] = po[6 * g.plane + rp + cm]; b[6] = po[6 * g.plane + rp + c];
      a[4] = po[4 * g.plane + r0 + c + 1]; b[4] = po[4 * g.plane + r0 + cp];
      a[7] = po[7 * g.plane + rp + c + 1]; b[7] = po[7 * g.plane + rp + cp];
      a[8] = po[8 * g.plane + rm + c + 1]; b[8] = po[8 * g.plane + rm + cp];
    }
    double rho_a, ux_a, uy_a, rho_b, ux_b, uy_b;
    m.collide(a, rho_a, ux_a, uy_a);
    m.collide(b, rho_b, ux_b, uy_b);
#pragma unroll
    for (int q = 0; q < Q; ++q) store2<NT_STORE>(pn + q * g.plane + r0 + c, a[q], b[q]);
    if (WITH_MOMENTS) {
      const long o = (long)r * g.C + c;  // moment fields carry no ghost rows
      const long n = (long)g.R * g.C;
      store2<false>(rho_out + o, rho_a, rho_b);
      store2<false>(u_out + o, ux_a, ux_b);
      store2<false>(u_out + n + o, uy_a, uy_b);
    }
  }
}

// One node per thread, every access a naturally aligned 8-byte load/store.
template <class Model, bool NT_LOAD, bool NT_STORE, bool WITH_MOMENTS>
__global__ __launch_bounds__(256) void k_stream_collide_v1(
    double* __restrict__ pn, const double* __restrict__ po, Geom g, Model m, int row_begin,
    int row_end, int tiles_per_row, double* __restrict__ rho_out, double* __restrict__ u_out) {
  const long items = (long)(row_end - row_begin) * tiles_per_row;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int r = row_begin + (int)(it / tiles_per_row);
    const int c = (int)(it % tiles_per_row) * 256 + threadIdx.x;
    if (c >= g.C) continue;
    const long rows[3] = {g.at(wrap_row(g, r + 1), 0), g.at(r, 0), g.at(wrap_row(g, r - 1), 0)};
    const int cols[3] = {wrap_col(g, c + 1), c, wrap_col(g, c - 1)};
    double f[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double* src = po + q * g.plane + rows[icx(q) + 1] + cols[icy(q) + 1];
      f[q] = NT_LOAD ? __builtin_nontemporal_load(src) : *src;
    }
    double rho, ux, uy;
    m.collide(f, rho, ux, uy);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double* dst = pn + q * g.plane + rows[1] + c;
      if (NT_STORE) __builtin_nontemporal_store(f[q], dst);
      else *dst = f[q];
    }
    if (WITH_MOMENTS) {
      const long o = (long)r * g.C + c, n = (long)g.R * g.C;
      rho_out[o] = rho;
      u_out[o] = ux;
      u_out[n + o] = uy;
    }
  }
}

// Variant 3: one node per thread and row, ROWS rows per thread with all 9*ROWS loads issued
// before the first collision, naturally aligned 8-byte accesses only.  Work item = (column
// tile, group of ROWS rows), numbered row-major.  XCD-aware mapping (swizzle != 0): workgroups
// are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md "Workgroup dispatch"), so block b
// is given item (b % 8) * chunk + b / 8: every XCD walks its own contiguous band of rows and
// the 64-byte sectors that neighbouring tiles both touch (the +-1 column reads) hit in that
// XCD's L2 instead of being fetched from HBM twice.  Placement affects speed only.
template <class Model, int BLOCK, int ROWS, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(BLOCK) void k_stream_collide_v3(
    double* __restrict__ pn, const double* __restrict__ po, Geom g, Model m, int row_begin,
    int row_end, int tiles_x, int n_items, int swizzle) {
  int item = blockIdx.x;
  if (swizzle) {
    const int chunk = (n_items + 7) >> 3;
    item = (item & 7) * chunk + (item >> 3);
    if (item >= n_items) return;
  }
  const int c = (item % tiles_x) * BLOCK + threadIdx.x;
  const int rbase = row_begin + (item / tiles_x) * ROWS;
  if (c >= g.C) return;
  const int cols[3] = {wrap_col(g, c + 1), c, wrap_col(g, c - 1)};
  double f[ROWS][Q];
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const int r = rbase + k;
    if (r < row_end) {
      const long rows[3] = {g.at(wrap_row(g, r + 1), 0), g.at(r, 0), g.at(wrap_row(g, r - 1), 0)};
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const double* src = po + q * g.plane + rows[icx(q) + 1] + cols[icy(q) + 1];
        f[k][q] = NT_LOAD ? __builtin_nontemporal_load(src) : *src;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const int r = rbase + k;
    if (r < row_end) {
      double rho, ux, uy;
      m.collide(f[k], rho, ux, uy);
      const long o = g.at(r, c);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        double* dst = pn + q * g.plane + o;
        if (NT_STORE) __builtin_nontemporal_store(f[k][q], dst);
        else *dst = f[k][q];
      }
    }
  }
}

// Temporal blocking: TWO time steps per launch.  A workgroup owns a TR x 64 tile of the t+2
// lattice.  Phase 1 pull-streams from global memory and collides the (TR+2) x 66 nodes the
// tile's second step depends on, leaving their post-collision populations in LDS; phase 2
// pull-streams from LDS, collides again and stores the tile.  HBM traffic per launch is one
// read (+ halo, mostly served by L2 / Infinity Cache where tiles overlap) and one write of the
// lattice for two updates of every node: 72 B/LUP instead of 144 B/LUP.  Each node still
// undergoes exactly the single-step arithmetic, so results are bit-identical; the price is the
// halo's redundant first-step collisions ((TR+2)*66 / (TR*64): 1.29x at TR = 8).
// Periodic or ghost-row (ghost >= 2 rows) edges only: boundary fix-ups are not fused here.
// Requires C % 64 == 0.
template <class Model, int TR, int BLOCK, bool NT_STORE>
__global__ __launch_bounds__(BLOCK) void k_stream_collide_tb2(
    double* __restrict__ pn, const double* __restrict__ po, Geom g, Model m, int row_begin,
    int row_end, int tiles_x, int tiles_y, int order) {
  constexpr int TC = 64, LR = TR + 2, LC = TC + 2;
  __shared__ double s[Q][LR][LC];
  int tx, ty;
  if (order == 1) {
    // XCD-aware: workgroups are dealt round-robin over the 8 XCDs; XCD k owns tile columns
    // tx = k (mod 8) and walks DOWN each of them, so vertically adjacent tiles -- which share
    // two halo rows of every population -- run back to back on one L2.  (tiles_x % 8 == 0)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tx = (j / tiles_y) * 8 + xcd;
    ty = j % tiles_y;
  } else {
    tx = blockIdx.x % tiles_x;
    ty = blockIdx.x / tiles_x;
  }
  const int r0 = row_begin + ty * TR, c0 = tx * TC;
  for (int i = threadIdx.x; i < LR * LC; i += BLOCK) {
    const int lr = i / LC, lc = i - lr * LC;
    const int r = wrap_row(g, r0 + lr - 1), c = wrap_col(g, c0 + lc - 1);
    const long rows[3] = {g.at(wrap_row(g, r + 1), 0), g.at(r, 0), g.at(wrap_row(g, r - 1), 0)};
    const int cols[3] = {wrap_col(g, c + 1), c, wrap_col(g, c - 1)};
    double f[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q] = po[q * g.plane + rows[icx(q) + 1] + cols[icy(q) + 1]];
    double rho, ux, uy;
    m.collide(f, rho, ux, uy);
#pragma unroll
    for (int q = 0; q < Q; ++q) s[q][lr][lc] = f[q];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TR * TC; i += BLOCK) {
    const int tr = i / TC, tc = i % TC;
    const int r = r0 + tr;
    if (r >= row_end) break;
    double f[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q] = s[q][tr + 1 - icx(q)][tc + 1 - icy(q)];
    double rho, ux, uy;
    m.collide(f, rho, ux, uy);
    const long o = g.at(r, c0 + tc);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double* dst = pn + q * g.plane + o;
      if (NT_STORE) __builtin_nontemporal_store(f[q], dst);
      else *dst = f[q];
    }
  }
}

// Temporal blocking, register sliding window: D time steps per launch, no LDS, no barriers.
// One WAVEFRONT owns a strip of 64 columns and walks down a chunk of rows.  Per iteration it
//   level 1 : pull-streams ONE new row from global memory (9 coalesced 512-B loads, prefetched
//             one iteration ahead) and collides it,
//   level l : pull-streams row (r - l + 1) of step t+l from the last three rows of step t+l-1,
//             which it keeps in REGISTERS (ring[l-2][3][9]); the +-1-column neighbours come from
//             the adjacent lanes by wave shuffles,
//   level D : stores its row.
// Every lattice row is read once and written once per D steps; the only redundancy is the
// (D-1)-column overlap on each side of a strip (valid output columns per wave: 64 - 2(D-1)) and
// 2(D-1) warm-up rows per row chunk.  Arithmetic per node and step is the single-step arithmetic:
// bit-identical results.  The row loop is unrolled by 3 so that every ring index is static.
// Periodic or ghost-row (ghost >= D) edges only.
// value of the neighbouring lane: DPP wavefront shifts (one v_mov_b32_dpp per dword, no LDS
// crossbar traffic) -- gfx9-family wave_shr:1 / wave_shl:1.  Edge lanes keep their own value;
// they are outside the valid column range of the level that consumes them.
#ifndef LBM_SW_DPP
#define LBM_SW_DPP 1
#endif
__device__ __forceinline__ double lane_from_prev(double v) {  // lane i <- lane i-1
#if LBM_SW_DPP
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);  // wave_shr:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#else
  return __shfl_up(v, 1);
#endif
}
__device__ __forceinline__ double lane_from_next(double v) {  // lane i <- lane i+1
#if LBM_SW_DPP
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);  // wave_shl:1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
#else
  return __shfl_down(v, 1);
#endif
}

// LDSR: the ring lives in wave-private LDS instead of registers -- a level's row is written once and the next level
// pulls its 9 populations with ds_read_b64 whose +-1-lane offsets ARE the column shift (no DPP moves), and nothing of
// the ring competes for the 256 architectural VGPRs (the register ring of a VALU-heavy model lives partly in AGPRs:
// two v_accvgpr moves per double and use).  Of a row computed in iteration i the next level reads the c_x = -1
// populations (q = 3, 6, 7) in iteration i, the c_x = 0 ones (0, 2, 4) in i + 1 and the c_x = +1 ones (1, 5, 8) in
// i + 2; pulling the old rows BEFORE publishing the new one, that is 1 + 1 + 2 slots of 3 populations per level --
// 12 doubles per lane instead of 27: 6.3 KB per wave and level, so that a 3-step window runs TWO waves per SIMD.
// Same values, same bits.
constexpr int SW_LDS_LANES = 66;  // 1 + lane + 1: the +-1-lane reads of lanes 0 / 63 stay inside (they feed invalid lanes only)
__host__ __device__ constexpr int sw_lds_level_doubles() { return 12 * SW_LDS_LANES; }
__host__ __device__ constexpr int sw_lds_ring_doubles(int D) { return (D > 1 ? D - 1 : 1) * sw_lds_level_doubles(); }
__host__ __device__ constexpr int sw_grp_pos(int q) { return (q == 3 || q == 0 || q == 1) ? 0 : ((q == 6 || q == 2 || q == 5) ? 1 : 2); }

template <class Model, int D, int K, bool NT_STORE, bool HAS_BC = false, bool LDSR = false>
__device__ __forceinline__ void sw_iteration(double (&ring)[D > 1 ? D - 1 : 1][3][Q], double (&cur)[Q],
                                             double* __restrict__ pn, const double* __restrict__ po,
                                             const Geom& g, const Model& m, int i, int rbase, int R0,
                                             int R1, int c_load, const int (&cols)[3], bool lane_ok,
                                             int c_out, const Bc& bc = Bc{}, int c_raw = 0, double* lds = nullptr, int lane = 0) {
  // ---- prefetch level-1 inputs of the NEXT iteration -----------------------------------------
  double nxt[Q];
  {
    const int r1n = rbase + i + 1;
    int rr[3] = {r1n + 1, r1n, r1n - 1};  // rows supplying cx = -1, 0, +1
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (g.ghost) rr[k] = rr[k] < -g.ghost ? -g.ghost : (rr[k] > g.R + g.ghost - 1 ? g.R + g.ghost - 1 : rr[k]);
      else rr[k] = rr[k] < 0 ? rr[k] + g.R : (rr[k] >= g.R ? rr[k] - g.R : rr[k]);
    }
    // uniform row base (scalar registers) + this lane's 32-bit byte offset: the loads take the saddr form and no
    // 64-bit vector address arithmetic (one v_lshl_add_u64 per access before; the KBC window is VALU-bound)
    // (round 4, measured and NOT kept: for the LDS-ring window the 18 + 9 accesses of an iteration were forced onto the saddr
    // form -- row base as an opaque scalar pair, 32-bit lane offset -- which takes 55 v_lshl_add_u64 out of the 3-iteration
    // body (2288 -> 2233 VALU instructions, 281 instead of 287 lane-ops per update) and puts 250 scalar instructions in;
    // KBC 4096^2 on one box, alternating: 70.0 / 70.1 / 70.2 / 71.9 k before, 69.0 / 69.1 / 69.1 / 69.0 k after.
    // profiles/r04_kbc_saddr_ab.txt)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const char* rowp = reinterpret_cast<const char*>(po + q * g.plane + g.at(rr[icx(q) + 1], 0));
      nxt[q] = *reinterpret_cast<const double*>(rowp + (unsigned)cols[icy(q) + 1] * 8u);
    }
  }
  // ---- level 1 ---------------------------------------------------------------------------------
  double f[Q], rho, ux, uy;
#pragma unroll
  for (int q = 0; q < Q; ++q) f[q] = cur[q];
  // HAS_BC: nodes on a wall row / wall column take the full boundary gather instead of the
  // prefetched plain one (a few lanes per row, two rows per lattice)
  const bool col_in = c_raw >= 0 && c_raw < g.C;
  const bool wall_col = HAS_BC && col_in && ((c_raw == 0 && bc_is_wall(bc.col_lo)) || (c_raw == g.C - 1 && bc_is_wall(bc.col_hi)));
  // rows: periodic ones wrap (row -1 IS row R-1 and needs its wall columns treated); beyond a wall
  // row there is only garbage that the wall row never reads
  // (slabs: the ghost rows behind a HALO edge are real nodes of the neighbour, wall columns included)
  const bool rows_wrap = HAS_BC && !g.ghost && !bc_is_wall(bc.row_lo) && !bc_is_wall(bc.row_hi);
  const int row_min = (HAS_BC && g.ghost && bc.row_lo == LBM_EDGE_HALO) ? -g.ghost : 0;
  const int row_max = (HAS_BC && g.ghost && bc.row_hi == LBM_EDGE_HALO) ? g.R + g.ghost - 1 : g.R - 1;
  if (HAS_BC) {
    int r1 = rbase + i;
    if (rows_wrap) r1 = r1 < 0 ? r1 + g.R : (r1 >= g.R ? r1 - g.R : r1);
    const bool wall_row = (r1 == 0 && bc_is_wall(bc.row_lo)) || (r1 == g.R - 1 && bc_is_wall(bc.row_hi));
    if (col_in && r1 >= row_min && r1 <= row_max && (wall_row || wall_col)) gather_walls(f, po, g, bc, r1, c_raw);
  }
  m.collide(f, rho, ux, uy);
  // ---- levels 2..D -------------------------------------------------------------------------------
#pragma unroll
  for (int l = 2; l <= D; ++l) {
    // publish level l-1's row of this iteration, then gather level l's row from the ring:
    // cx = -1 pops from the row just computed (slot K), cx = 0 from the previous iteration's row
    // (slot K+2), cx = +1 from the one before (slot K+1); cy = +-1 via the neighbouring lanes.
    if constexpr (LDSR) {
      double* lv = lds + (l - 2) * sw_lds_level_doubles();
      double* slotA = lv;                                               // c_x = -1 populations of the newest row
      double* slotB = lv + 3 * SW_LDS_LANES;                            // c_x = 0 populations of the row before
      double* slotC = lv + (6 + 3 * (i & 1)) * SW_LDS_LANES;            // c_x = +1 populations of the row two before (same parity as i)
      double fn[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {  // 1. the old rows, before this iteration's row takes their slots
        if (icx(q) == 0) fn[q] = slotB[sw_grp_pos(q) * SW_LDS_LANES + lane + 1 - icy(q)];
        else if (icx(q) == 1) fn[q] = slotC[sw_grp_pos(q) * SW_LDS_LANES + lane + 1 - icy(q)];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < Q; ++q) {  // 2. publish level l-1's row of this iteration
        double* dst = icx(q) == -1 ? slotA : (icx(q) == 0 ? slotB : slotC);
        dst[sw_grp_pos(q) * SW_LDS_LANES + lane + 1] = f[q];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int q = 0; q < Q; ++q)    // 3. the c_x = -1 populations come from the row just published
        if (icx(q) == -1) fn[q] = slotA[sw_grp_pos(q) * SW_LDS_LANES + lane + 1 - icy(q)];
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = fn[q];
    } else {
#pragma unroll
    for (int q = 0; q < Q; ++q) ring[l - 2][K][q] = f[q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int slot = icx(q) == -1 ? K : (icx(q) == 0 ? (K + 2) % 3 : (K + 1) % 3);
      double v = ring[l - 2][slot][q];
      if (icy(q) == 1) v = lane_from_prev(v);        // from column c-1
      else if (icy(q) == -1) v = lane_from_next(v);  // from column c+1
      f[q] = v;
    }
    }
    if (HAS_BC && !LDSR) {  // level l's row; its own level-(l-1) populations sit in the ring's c_x = 0 slot
      int rl = rbase + i - (l - 1);
      if (rows_wrap) rl = rl < 0 ? rl + g.R : (rl >= g.R ? rl - g.R : rl);
      const bool wall_row = (rl == 0 && bc_is_wall(bc.row_lo)) || (rl == g.R - 1 && bc_is_wall(bc.row_hi));
      if (col_in && rl >= row_min && rl <= row_max && (wall_row || wall_col)) {
        double own[Q];  // element-wise copy: binding the ring row by reference keeps the ring in scratch
#pragma unroll
        for (int q = 0; q < Q; ++q) own[q] = ring[l - 2][(K + 2) % 3][q];
        bc_fixups_own(f, own, g, bc, rl, c_raw);
      }
    }
    m.collide(f, rho, ux, uy);
  }
  // ---- store level D's row -------------------------------------------------------------------------
  // (round 4, measured and NOT kept: the stores sit behind a branch -- warm-up rows, halo lanes -- and vmcnt counts loads and
  // stores in issue order, so the compiler's wait for the prefetched row at the top of the next iteration is vmcnt(8): it
  // also covers eight of the nine stores issued just before.  Taking the prefetched row over HERE, ahead of the stores (an
  // empty asm consuming nxt[] and f[]), removes that wait from the code -- and measures 0.6 - 1.3 % SLOWER on the headline,
  // the KBC and the cylinder windows, alternating on one box: profiles/r04_kbc_strip_width.txt)
  const int rD = rbase + i - (D - 1);  // level D's row = level 1's row - (D-1)
  if (lane_ok && rD >= R0 && rD < R1) {
    const long o = g.at(rD, 0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double* dst = reinterpret_cast<double*>(reinterpret_cast<char*>(pn + q * g.plane + o) + (unsigned)c_out * 8u);
      if (NT_STORE) __builtin_nontemporal_store(f[q], dst);
      else *dst = f[q];
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) cur[q] = nxt[q];
  (void)c_load;
}

#ifdef LBM_EXPERIMENTS  // sliding window with the level-1 rows prefetched two iterations ahead (sw_iteration_pf2): csrc/experiments/sw_prefetch2.hpp
#include "experiments/sw_prefetch2.hpp"
#endif

// Register budget (ring 54*(D-1) VGPRs + prefetch 18 + working set) -> waves per SIMD the kernel
// is compiled for: D = 2: 4, D = 3: 3, D = 4, 5: 2, deeper: 1 (only enforced for 4-wave blocks).
// Valid output columns per wave: 64 - 2 (D - 1) lanes hold valid level-D values; the strip keeps the
// largest multiple of 8 of them, so that every strip's stores start on a 64-byte boundary.  Measured
// on the 8192^2 box: D = 4 at 58 columns 106 k MLUPS (every store row straddles two extra sectors:
// partial writes), D = 5 at 56 columns 167 k.
// A model may declare `static constexpr bool kFullStrips = true` and keep all 64 - 2 (D - 1) columns.  KBC did in rounds 2 - 3
// (400 lane-ops per update then: 7 % fewer strips beat the aligned stores, 63.9 k against 60.3 k) and no longer does: with
// the collision at 196 operations the misaligned segments were what bounded the launch (kbc.hpp, 85 k against 77.5 k).
__host__ __device__ constexpr int sw_strip_width(int D, bool full = false) {
  return full ? 64 - 2 * (D - 1) : (64 - 2 * (D - 1)) / 8 * 8;
}
template <class M, class = void>
struct sw_full_strips : std::false_type {};
template <class M>
struct sw_full_strips<M, std::void_t<decltype(M::kFullStrips)>> : std::bool_constant<M::kFullStrips> {};
__host__ __device__ constexpr int sw_waves_per_simd(int D) { return D <= 2 ? 4 : (D == 3 ? 3 : (D <= 5 ? 2 : 1)); }
// one wavefront's walk down its strip chunk: `wave` = its index among the strips x chunks of this part of the launch
template <class Model, int D, bool NT_STORE, bool HAS_BC, bool PF2, bool LDSR = false>
__device__ __forceinline__ void sw_wave_body(double* __restrict__ pn, const double* __restrict__ po, const Geom& g,
                                             const Model& m, int row_begin, int row_end, int rows_per_chunk, int strips,
                                             int wave, int lane, const Bc& bc, int strip0, int chunk_stride,
                                             double* lds = nullptr) {
  static_assert(!(LDSR && (HAS_BC || PF2)), "the LDS ring exists for the plain window only");
  constexpr int W = sw_strip_width(D, sw_full_strips<Model>::value);  // output columns per wave
  const int strip = strip0 + wave % strips, chunk = wave / strips;  // strips = those of this launch, from strip0 on
  // chunk_stride > rows_per_chunk: the chunks are row ranges apart from each other (both edge-row
  // ranges of a slab in one launch: chunk 0 = [row_begin, +rows), chunk 1 = [row_begin + stride, +rows))
  const int R0 = row_begin + chunk * (chunk_stride > 0 ? chunk_stride : rows_per_chunk);
  const int R1 = R0 + rows_per_chunk < row_end ? R0 + rows_per_chunk : row_end;
  // this lane's column at every level; lanes D-1 .. 63-(D-1) hold valid level-D values
  int c = strip * W - (D - 1) + lane;
  const int c_raw = c;
  const bool lane_ok = lane >= D - 1 && lane < D - 1 + W && c < g.C;
  // walls on the columns: no wrap -- lanes left of column 0 / right of column C-1 compute garbage
  // that the wall nodes never read (their fix-ups replace exactly the populations coming from there)
  const bool walled_cols = HAS_BC && (bc_is_wall(bc.col_lo) || bc_is_wall(bc.col_hi));
  if (walled_cols) {
    c = c < 0 ? 0 : (c > g.C - 1 ? g.C - 1 : c);
  } else {
    c = c < 0 ? c + g.C : (c >= g.C ? c - g.C : c);
    c = c >= g.C ? c - g.C : c;  // last strip may run more than one period past the edge
  }
  const int cols[3] = {walled_cols ? (c + 1 > g.C - 1 ? g.C - 1 : c + 1) : wrap_col(g, c + 1), c,
                       walled_cols ? (c - 1 < 0 ? 0 : c - 1) : wrap_col(g, c - 1)};
  // level 1 at iteration i computes row rbase + i; level D's row = rbase + i - (D-1); the first
  // valid level-D row (all inputs warmed up) appears at i = 2(D-1) and must be R0
  const int rbase = R0 - (D - 1);
  const int n_iter = (R1 - R0) + 2 * (D - 1);
  double ring[D > 1 ? D - 1 : 1][3][Q];
#pragma unroll
  for (int a = 0; a < (D > 1 ? D - 1 : 1); ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int q = 0; q < Q; ++q) ring[a][b][q] = 1.0;  // warm-up garbage, never stored
#ifdef LBM_EXPERIMENTS
  if constexpr (PF2 && !HAS_BC) {
    double raw[3][Q];
#pragma unroll
    for (int a = 0; a < 2; ++a) {  // level-1 inputs of iterations 0 and 1
      int rr[3] = {rbase + a + 1, rbase + a, rbase + a - 1};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (g.ghost) rr[k] = rr[k] < -g.ghost ? -g.ghost : (rr[k] > g.R + g.ghost - 1 ? g.R + g.ghost - 1 : rr[k]);
        else rr[k] = rr[k] < 0 ? rr[k] + g.R : (rr[k] >= g.R ? rr[k] - g.R : rr[k]);
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) raw[a][q] = po[q * g.plane + g.at(rr[icx(q) + 1], 0) + cols[icy(q) + 1]];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) raw[2][q] = 0.0;
    for (int i = 0; i < n_iter; i += 3) {
      sw_iteration_pf2<Model, D, 0, NT_STORE>(ring, raw, pn, po, g, m, i, rbase, R0, R1, cols, lane_ok, c);
      sw_iteration_pf2<Model, D, 1, NT_STORE>(ring, raw, pn, po, g, m, i + 1, rbase, R0, R1, cols, lane_ok, c);
      sw_iteration_pf2<Model, D, 2, NT_STORE>(ring, raw, pn, po, g, m, i + 2, rbase, R0, R1, cols, lane_ok, c);
    }
    return;
  }
#else
  static_assert(!PF2, "the two-row prefetch is an experiment (make EXPERIMENTS=1)");
#endif
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
  // the column the wall fix-ups see: with wall columns the raw one (lanes beyond the lattice hold no node), with
  // periodic columns the WRAPPED one -- a lane left of column 0 holds a real node of a wall ROW and needs its fix-ups
  // like any other (row walls + periodic columns: the corner lanes fed garbage into the valid columns before)
  const int c_bc = walled_cols ? c_raw : c;
  if constexpr (LDSR) {  // warm-up garbage must be finite numbers here too
    for (int k = lane; k < sw_lds_ring_doubles(D); k += 64) lds[k] = 1.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  for (int i = 0; i < n_iter; i += 3) {
    sw_iteration<Model, D, 0, NT_STORE, HAS_BC, LDSR>(ring, cur, pn, po, g, m, i, rbase, R0, R1, c, cols, lane_ok, c, bc, c_bc, lds, lane);
    sw_iteration<Model, D, 1, NT_STORE, HAS_BC, LDSR>(ring, cur, pn, po, g, m, i + 1, rbase, R0, R1, c, cols, lane_ok, c, bc, c_bc, lds, lane);
    sw_iteration<Model, D, 2, NT_STORE, HAS_BC, LDSR>(ring, cur, pn, po, g, m, i + 2, rbase, R0, R1, c, cols, lane_ok, c, bc, c_bc, lds, lane);
  }
}

template <class Model, int D, int WAVES, bool NT_STORE, bool HAS_BC = false, bool PF2 = false, bool LDSR = false>
__global__ __launch_bounds__(64 * WAVES, (LDSR ? 2 : (WAVES == 4 ? sw_waves_per_simd(D) : 1))) void k_stream_collide_sw(
    double* __restrict__ pn, const double* __restrict__ po, Geom g, Model m, int row_begin,
    int row_end, int rows_per_chunk, int strips, int n_waves, int xcd_group, Bc bc = Bc{}, int strip0 = 0,
    int chunk_stride = 0) {
  // workgroup b runs on XCD b % 8 as that XCD's (b / 8)-th block.  xcd_group = G > 0: consecutive
  // blocks of one XCD take G consecutive strip groups, so the 128-B lines that neighbouring strips
  // share (a strip's 64 columns start 8 doubles before a line boundary) are fetched once per L2
  int blk = blockIdx.x;
  if (xcd_group > 0) {
    const int x = blk % 8, mth = blk / 8, win = 8 * xcd_group;
    if ((mth / xcd_group + 1) * win <= (int)gridDim.x) blk = (mth / xcd_group) * win + x * xcd_group + mth % xcd_group;
  }
  // readfirstlane: the wave index is uniform, and only then do the row indices and the 18 row base addresses of an
  // iteration live in scalar registers (KBC window: ~6 % fewer VALU instructions; the kernel is VALU-bound)
  const int wave = blk * WAVES + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= n_waves) return;
  if constexpr (LDSR) {
    __shared__ double lring[WAVES * sw_lds_ring_doubles(D)];
    sw_wave_body<Model, D, NT_STORE, HAS_BC, PF2, true>(pn, po, g, m, row_begin, row_end, rows_per_chunk, strips, wave, lane, bc, strip0, chunk_stride,
                                                        lring + (threadIdx.x >> 6) * sw_lds_ring_doubles(D));
  } else {
    sw_wave_body<Model, D, NT_STORE, HAS_BC, PF2>(pn, po, g, m, row_begin, row_end, rows_per_chunk, strips, wave, lane, bc, strip0, chunk_stride);
  }
}

// Wall-bounded launch in ONE dispatch: the few waves of the frame (outermost strips, rows next to a wall row) run the
// wall-carrying body, all others the plain one.  Both instantiations live in one kernel, whose register budget is the
// larger one's -- no loss: at D >= 4 either runs one wave per SIMD anyway.  Frame waves come first in the grid.  (The
// two-launch form forks the frame onto a helper stream and joins it: that event pair alone costs ~0.1 ms per launch,
// profiles/r02_ring_dissect.txt.)
struct SwPart {
  int r0, r1, s0, ns, rpc, wave0;  // rows [r0, r1) x strips [s0, s0 + ns) in chunks of rpc rows; first wave of the part
};
struct SwParts {
  SwPart p[5];  // 0..3: frame (left strips, right strips, top rows, bottom rows), 4: interior
  int n_frame_waves, n_waves;
};
template <class Model, int D, bool NT_STORE>
__global__ __launch_bounds__(128, 1) void k_stream_collide_sw_walls(double* __restrict__ pn, const double* __restrict__ po,
                                                                    Geom g, Model m, Bc bc, SwParts parts) {
  const int wave = blockIdx.x * 2 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= parts.n_waves) return;
  if (wave < parts.n_frame_waves) {
    int k = 0;
#pragma unroll
    for (int a = 1; a < 4; ++a)
      if (parts.p[a].ns > 0 && parts.p[a].r1 > parts.p[a].r0 && wave >= parts.p[a].wave0) k = a;
    const SwPart pt = parts.p[k];
    sw_wave_body<Model, D, NT_STORE, true, false>(pn, po, g, m, pt.r0, pt.r1, pt.rpc, pt.ns, wave - pt.wave0, lane, bc, pt.s0, 0);
  } else {
    const SwPart pt = parts.p[4];
    sw_wave_body<Model, D, NT_STORE, false, false>(pn, po, g, m, pt.r0, pt.r1, pt.rpc, pt.ns, wave - pt.wave0, lane, bc, pt.s0, 0);
  }
}

#ifdef LBM_EXPERIMENTS  // sliding window on paired strips (k_stream_collide_swp): csrc/experiments/sw_paired_strips.hpp
#include "experiments/sw_paired_strips.hpp"
#endif
#ifdef LBM_EXPERIMENTS  // sliding window with two columns per lane (k_stream_collide_sw2): csrc/experiments/sw_two_columns.hpp
#include "experiments/sw_two_columns.hpp"
#endif

// Edge pass: recompute the boundary nodes (rows 0 / R-1 where they carry a fix-up, columns
// 0 / C-1 where they do) with the full boundary gather and overwrite what the interior
// kernel stored for them.  O(R + C) nodes.
template <class Model, bool WITH_MOMENTS>
__global__ __launch_bounds__(256) void k_edge_stream_collide(
    double* __restrict__ pn, const double* __restrict__ po, Geom g, Bc bc, Model m, int row_begin,
    int row_end, double* __restrict__ rho_out, double* __restrict__ u_out) {
  // edge list of the row range (n = row_end - row_begin):
  // [0, C) row 0 | [C, 2C) row R-1 | [2C, 2C+n) col 0 | [2C+n, 2C+2n) col C-1
  const int i = blockIdx.x * blockDim.x + threadIdx.x, n = row_end - row_begin;
  int r, c;
  if (i < g.C) { r = 0; c = i; }
  else if (i < 2 * g.C) { r = g.R - 1; c = i - g.C; }
  else if (i < 2 * g.C + n) { r = row_begin + i - 2 * g.C; c = 0; }
  else if (i < 2 * g.C + 2 * n) { r = row_begin + i - 2 * g.C - n; c = g.C - 1; }
  else return;
  if (r < row_begin || r >= row_end) return;
  if (i >= 2 * g.C && (r == 0 || r == g.R - 1)) return;  // corners belong to the row lists
  double f[Q], rho, ux, uy;
  gather_bc(f, po, g, bc, r, c);
  m.collide(f, rho, ux, uy);
  const long o = g.at(r, c);
#pragma unroll
  for (int q = 0; q < Q; ++q) pn[q * g.plane + o] = f[q];
  if (WITH_MOMENTS) {
    const long n = (long)g.R * g.C, oo = (long)r * g.C + c;
    rho_out[oo] = rho;
    u_out[oo] = ux;
    u_out[n + oo] = uy;
  }
}

// Whole-lattice generic kernel (any C, boundary gather on every node): small lattices and
// odd column counts.  FROM_POST = true: input holds post-collision populations and is
// streamed at read time; false: input is the pre-collision state (first iteration).
template <class Model, bool FROM_POST, bool WITH_MOMENTS>
__global__ __launch_bounds__(256) void k_generic_collide(
    double* __restrict__ pn, const double* __restrict__ in, Geom g, Bc bc, Model m, int row_begin,
    int row_end, double* __restrict__ rho_out, double* __restrict__ u_out) {
  const long n_nodes = (long)(row_end - row_begin) * g.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes;
       i += (long)gridDim.x * blockDim.x) {
    const int r = row_begin + (int)(i / g.C), c = (int)(i % g.C);
    double f[Q], rho, ux, uy;
    if (FROM_POST) {
      gather_bc(f, in, g, bc, r, c);
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = in[q * g.plane + g.at(r, c)];
    }
    m.collide(f, rho, ux, uy);
    const long o = g.at(r, c);
#pragma unroll
    for (int q = 0; q < Q; ++q) pn[q * g.plane + o] = f[q];
    if (WITH_MOMENTS) {
      const long n = (long)g.R * g.C, oo = (long)r * g.C + c;
      rho_out[oo] = rho;
      u_out[oo] = ux;
      u_out[n + oo] = uy;
    }
  }
}

}  // namespace lbm
