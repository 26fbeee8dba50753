// EXPERIMENTS build only (make -C lattice-boltzmann-method_amd/csrc EXPERIMENTS=1): sliding window with the level-1 rows prefetched two iterations ahead (sw_iteration_pf2)
// Measured and not kept -- DESIGN.md 4.2 / 9 hold the numbers.  Included from d2q9.hpp at the place the code used to stand;
// not a stand-alone header (it uses what that file has declared above the include).
// The same iteration with the level-1 inputs prefetched TWO rows ahead (plain edges only): three raw-row
// buffers rotate with the unroll index K -- buffer K holds this iteration's row (loaded two iterations ago),
// buffer (K + 2) % 3, consumed by the previous iteration, takes row i + 2.  One wave per SIMD leaves the
// registers for it (18 more), and nothing else hides a late row.
template <class Model, int D, int K, bool NT_STORE>
__device__ __forceinline__ void sw_iteration_pf2(double (&ring)[D > 1 ? D - 1 : 1][3][Q], double (&raw)[3][Q],
                                                 double* __restrict__ pn, const double* __restrict__ po,
                                                 const Geom& g, const Model& m, int i, int rbase, int R0,
                                                 int R1, const int (&cols)[3], bool lane_ok, int c_out) {
  {
    const int r1n = rbase + i + 2;
    int rr[3] = {r1n + 1, r1n, r1n - 1};  // rows supplying cx = -1, 0, +1
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (g.ghost) rr[k] = rr[k] < -g.ghost ? -g.ghost : (rr[k] > g.R + g.ghost - 1 ? g.R + g.ghost - 1 : rr[k]);
      else rr[k] = rr[k] < 0 ? rr[k] + g.R : (rr[k] >= g.R ? rr[k] - g.R : rr[k]);
      if (!g.ghost) rr[k] = rr[k] >= g.R ? rr[k] - g.R : rr[k];  // two rows ahead may wrap twice on tiny lattices
    }
    constexpr int KN = (K + 2) % 3;
#pragma unroll
    for (int q = 0; q < Q; ++q) raw[KN][q] = po[q * g.plane + g.at(rr[icx(q) + 1], 0) + cols[icy(q) + 1]];
  }
  double f[Q], rho, ux, uy;
#pragma unroll
  for (int q = 0; q < Q; ++q) f[q] = raw[K][q];
  m.collide(f, rho, ux, uy);
#pragma unroll
  for (int l = 2; l <= D; ++l) {
#pragma unroll
    for (int q = 0; q < Q; ++q) ring[l - 2][K][q] = f[q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int slot = icx(q) == -1 ? K : (icx(q) == 0 ? (K + 2) % 3 : (K + 1) % 3);
      double v = ring[l - 2][slot][q];
      if (icy(q) == 1) v = lane_from_prev(v);
      else if (icy(q) == -1) v = lane_from_next(v);
      f[q] = v;
    }
    m.collide(f, rho, ux, uy);
  }
  const int rD = rbase + i - (D - 1);
  if (lane_ok && rD >= R0 && rD < R1) {
    const long o = g.at(rD, c_out);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double* dst = pn + q * g.plane + o;
      if (NT_STORE) __builtin_nontemporal_store(f[q], dst);
      else *dst = f[q];
    }
  }
}
