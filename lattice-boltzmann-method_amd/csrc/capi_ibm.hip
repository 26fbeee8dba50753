// C ABI, part 6: immersed boundary, multi-direct forcing (src/ibm.cpp) and its coupling to the
// BGK step (test/cylinder_test.cpp:110-127).  See include/lbm_hip.h for the design note.
#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "d2q9.hpp"
#include "internal.hpp"

namespace lbm {

// 4-point Peskin kernel, ibm.cpp:39-45
static double peskin4(double r_) {
  const double r = std::abs(r_);
  if (r <= 1) return 0.125 * (3.0 - 2.0 * r + std::sqrt(1.0 + 4.0 * r - 4.0 * r * r));
  else if (r <= 2) return 0.125 * (5.0 - 2.0 * r - std::sqrt(-7.0 + 12.0 * r - 4.0 * r * r));
  return 0.0;
}

struct IbmDev {
  int n_markers, RR, RC, r0, c0, X, Y;
  const int* box0;     // [n_markers] ROI-flat index of the box's first node (row0 * RC + col0)
  const double* phi;   // [n_markers][16], k = i*4 + j over the box flattened [row i][col j]
  const int* csr_ptr;  // [RR*RC + 1]
  const int* csr_mk;   // [nnz] marker of each (node, tap) pair, ascending per node
  const double* csr_w; // [nnz] its weight phi
  const int* touched;  // [n_touched] ROI-flat indices of the nodes with at least one tap
  int n_touched;
  const int* tap_t;    // [n_markers][16] index into `touched` of each marker tap
  const int* tptr;     // [n_touched + 1] csr_ptr restricted to the touched nodes
  double Ubx, Uby;     // marker velocity U_b (0 in the reference: stationary boundary)
  int moving;          // U_b != 0
  // the same tables laid out for k_ibm_step, where lane = marker (taps) and lane = touched node (pairs): consecutive
  // lanes read consecutive words.  ([marker][16] and CSR make every lane of a load hit its own cache line: the one
  // workgroup then spends its time in the texture addresser, 16 x more line requests than lines.)
  const int* tap_k;    // [16][n_markers]
  const double* phi_k; // [16][n_markers]
  const int* ell_cnt;  // [n_touched] pairs of the node
  const int* ell_mk;   // [ell_deg][n_touched] marker of pair s of the node (ascending), 0 beyond its count
  const double* ell_w; // [ell_deg][n_touched]
  int ell_deg;
};

// u, rho of the ROI window copied out of the full fields (ibm.cpp:163-164); F_sum = 0
__global__ __launch_bounds__(256) void k_ibm_begin(IbmDev d, const double* __restrict__ u,
                                                   const double* __restrict__ rho,
                                                   double* __restrict__ u_roi,
                                                   double* __restrict__ rho_roi,
                                                   double* __restrict__ F_sum) {
  const int n = d.RR * d.RC;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = i / d.RC, c = i % d.RC;
  const long s = (long)(d.r0 + r) * d.Y + (d.c0 + c), N = (long)d.X * d.Y;
  u_roi[i] = u[s];
  u_roi[n + i] = u[N + s];
  rho_roi[i] = rho[s];
  F_sum[i] = 0.0;
  F_sum[n + i] = 0.0;
}

// per marker: interpolate u_j, rho_j over its 4x4 box (taps in k order, as matmul(phi, box)),
// f_j = -2 rho_j u_j   (ibm.cpp:171-177)
__global__ __launch_bounds__(64) void k_ibm_interp(IbmDev d, const double* __restrict__ u_roi,
                                                   const double* __restrict__ rho_roi,
                                                   double* __restrict__ fj) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= d.n_markers) return;
  const int n = d.RR * d.RC, b0 = d.box0[j];
  double ujx = 0.0, ujy = 0.0, rhoj = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int node = b0 + (k / 4) * d.RC + (k % 4);
    const double w = d.phi[j * 16 + k];
    ujx += w * u_roi[node];
    ujy += w * u_roi[n + node];
    rhoj += w * rho_roi[node];
  }
  if (d.moving) {  // f_j = 2 rho_j (U_b - u_j): extension, no reference counterpart
    fj[j] = 2.0 * rhoj * (d.Ubx - ujx);
    fj[d.n_markers + j] = 2.0 * rhoj * (d.Uby - ujy);
  } else {
    fj[j] = -2.0 * rhoj * ujx;
    fj[d.n_markers + j] = -2.0 * rhoj * ujy;
  }
}

// per ROI node: F_n = sum over its (marker, tap) pairs in marker order (== the reference's
// sequential "F[box] += phi f_j", ibm.cpp:180-182); u += F_n / (2 rho); F_sum += F_n (:186,:189)
__global__ __launch_bounds__(256) void k_ibm_spread(IbmDev d, const double* __restrict__ fj,
                                                    double* __restrict__ u_roi,
                                                    const double* __restrict__ rho_roi,
                                                    double* __restrict__ F_sum) {
  const int n = d.RR * d.RC;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double fx = 0.0, fy = 0.0;
  for (int e = d.csr_ptr[i]; e < d.csr_ptr[i + 1]; ++e) {
    const int j = d.csr_mk[e];
    const double w = d.csr_w[e];
    fx += w * fj[j];
    fy += w * fj[d.n_markers + j];
  }
  const double rh = rho_roi[i];
  u_roi[i] += 0.5 * fx / rh;
  u_roi[n + i] += 0.5 * fy / rh;
  F_sum[i] += fx;
  F_sum[n + i] += fy;
}

// cylinder_test.cpp:116-127: S = ((1-0.5w)((a + b u.c)(F.c) - a (u.F)) E); f_coll[ROI] += S
__global__ __launch_bounds__(256) void k_ibm_add_source(IbmDev d, double* __restrict__ p, Geom g,
                                                        const double* __restrict__ u,
                                                        const double* __restrict__ F_sum,
                                                        double omega, double a, double b) {
  const int n = d.RR * d.RC;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int r = d.r0 + i / d.RC, c = d.c0 + i % d.RC;
  const long s = (long)r * d.Y + c, N = (long)d.X * d.Y;
  const double ux = u[s], uy = u[N + s], Fx = F_sum[i], Fy = F_sum[n + i];
  const double uF = ux * Fx + uy * Fy;
  const long o = g.at(r, c);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double cu = ux * (double)icx(q) + uy * (double)icy(q);
    const double cF = Fx * (double)icx(q) + Fy * (double)icy(q);
    p[q * g.plane + o] += ((1 - 0.5 * omega) * ((a + b * cu) * cF - a * uF) * wq(q));
  }
}

// the same on the touched nodes only (F = 0 elsewhere: the source there is exactly 0), many workgroups
__global__ __launch_bounds__(256) void k_ibm_add_source_touched(IbmDev d, double* __restrict__ p, Geom g,
                                                                const double* __restrict__ u,
                                                                const double* __restrict__ F_sum,
                                                                double omega, double a, double b) {
  const int n = d.RR * d.RC;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= d.n_touched) return;
  const int i = d.touched[t];
  const int r = d.r0 + i / d.RC, c = d.c0 + i % d.RC;
  const long s = (long)r * d.Y + c, N = (long)d.X * d.Y;
  const double ux = u[s], uy = u[N + s], Fx = F_sum[i], Fy = F_sum[n + i];
  const double uF = ux * Fx + uy * Fy;
  const long o = g.at(r, c);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double cu = ux * (double)icx(q) + uy * (double)icy(q);
    const double cF = Fx * (double)icx(q) + Fy * (double)icy(q);
    p[q * g.plane + o] += ((1 - 0.5 * omega) * ((a + b * cu) * cF - a * uF) * wq(q));
  }
}

// The whole forcing of one time step in ONE launch of ONE workgroup: eulerian_force_density
// (ibm.cpp:158-190, all m_max - 1 iterations) followed by the driver's source term
// (cylinder_test.cpp:116-127).  Only the nodes under a marker's 4x4 box ("touched", a band of a
// few thousand nodes) take part: elsewhere F = 0, u is never read again and the source is exactly
// 0.  The separate kernels above cost 2 (m_max - 1) + 2 dependent launches of microseconds of work
// each -- a ~0.1 ms latency chain per step; here the phases are separated by workgroup barriers
// and the working set (u, rho of the touched nodes, f_j of the markers) lives in LDS.
// Arithmetic per marker / per node is that of k_ibm_interp / k_ibm_spread / k_ibm_add_source in
// the same order, so results are bit-identical.  Dynamic LDS: (3 n_touched + 2 n_markers) doubles.
// OPT = 1 (default; "ibm_step_opt" = 0 selects the form of round 1, same bits): taps and (marker, weight) pairs are read
// from the lane-major tables (IbmDev::tap_k ... ell_w), pairs 4 at a time; all sums keep their order.  The launch is one workgroup deep: its time IS its chain of
// latencies (82 us -> see profiles/r02_ibm_force.txt).
// COH: accesses that are coherent across the compute units of the device without any cache-wide maintenance (agent-scope
// relaxed atomics = sc1 loads / stores): for data that workgroups on different XCDs hand to each other inside one launch
template <bool COH>
__device__ __forceinline__ double ld_d(const double* p) {
  return COH ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool COH>
__device__ __forceinline__ void st_d(double* p, double v) {
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

template <int OPT, int MAXOWN, bool COH = false>
__device__ __forceinline__ void ibm_force_wg(const IbmDev& d, int m_max, const double* u,
                                             const double* rho, double* F_sum,
                                             double* __restrict__ p, const Geom& g, double omega, double a, double b,
                                             int with_source, double* lds) {
  const int nt = d.n_touched, nm = d.n_markers, n = d.RR * d.RC;
  double* s_ux = lds;            // [nt]
  double* s_uy = lds + nt;       // [nt]
  double* s_rho = lds + 2 * nt;  // [nt]
  double* s_fj = lds + 3 * nt;   // [2 nm]
  const long N = (long)d.X * d.Y;
  // each thread owns the touched nodes t = tid, tid + 1024, ...: F accumulates in registers
  // (MAXOWN: host guarantees nt <= MAXOWN * 1024)
  double Fx[MAXOWN], Fy[MAXOWN];
  int cnt[MAXOWN];
  constexpr int NB = 4;  // pairs fetched together (8: no faster)
#pragma unroll
  for (int k = 0; k < MAXOWN; ++k) {
    Fx[k] = 0.0;
    Fy[k] = 0.0;
    cnt[k] = 0;
    const int t = threadIdx.x + k * 1024;
    if (t < nt) {
      const int i = d.touched[t];
      const long s = (long)(d.r0 + i / d.RC) * d.Y + (d.c0 + i % d.RC);
      s_ux[t] = ld_d<COH>(u + s);
      s_uy[t] = ld_d<COH>(u + N + s);
      s_rho[t] = ld_d<COH>(rho + s);
      if (OPT) cnt[k] = d.ell_cnt[t];
    }
  }
  __syncthreads();
  for (int it = 1; it < m_max; ++it) {
    for (int j = threadIdx.x; j < nm; j += 1024) {
      double ujx = 0.0, ujy = 0.0, rhoj = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int t = OPT ? d.tap_k[k * nm + j] : d.tap_t[j * 16 + k];
        const double w = OPT ? d.phi_k[k * nm + j] : d.phi[j * 16 + k];
        ujx += w * s_ux[t];
        ujy += w * s_uy[t];
        rhoj += w * s_rho[t];
      }
      if (d.moving) {
        s_fj[j] = 2.0 * rhoj * (d.Ubx - ujx);
        s_fj[nm + j] = 2.0 * rhoj * (d.Uby - ujy);
      } else {
        s_fj[j] = -2.0 * rhoj * ujx;
        s_fj[nm + j] = -2.0 * rhoj * ujy;
      }
    }
    __syncthreads();
    if (OPT) {
#pragma unroll
      for (int k = 0; k < MAXOWN; ++k) {
        const int t = threadIdx.x + k * 1024;
        double fx = 0.0, fy = 0.0;
        for (int s0 = 0; s0 < cnt[k]; s0 += NB) {  // (cnt = 0 for the slots this thread does not own)
          const int c4 = cnt[k] - s0;
          int jj[NB];
          double ww[NB];
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            jj[i] = i < c4 ? d.ell_mk[(s0 + i) * nt + t] : 0;
            ww[i] = i < c4 ? d.ell_w[(s0 + i) * nt + t] : 0.0;
          }
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            if (i < c4) {
              fx += ww[i] * s_fj[jj[i]];
              fy += ww[i] * s_fj[nm + jj[i]];
            }
          }
        }
        if (t < nt) {
          const double rh = s_rho[t];
          s_ux[t] += 0.5 * fx / rh;
          s_uy[t] += 0.5 * fy / rh;
          Fx[k] += fx;
          Fy[k] += fy;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < MAXOWN; ++k) {
        const int t = threadIdx.x + k * 1024;
        if (t < nt) {
          double fx = 0.0, fy = 0.0;
          for (int e = d.tptr[t]; e < d.tptr[t + 1]; ++e) {
            const int j = d.csr_mk[e];
            const double w = d.csr_w[e];
            fx += w * s_fj[j];
            fy += w * s_fj[nm + j];
          }
          const double rh = s_rho[t];
          s_ux[t] += 0.5 * fx / rh;
          s_uy[t] += 0.5 * fy / rh;
          Fx[k] += fx;
          Fy[k] += fy;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < MAXOWN; ++k) {
    const int t = threadIdx.x + k * 1024;
    if (t >= nt) continue;
    const int i = d.touched[t];
    st_d<COH>(F_sum + i, Fx[k]);
    st_d<COH>(F_sum + n + i, Fy[k]);
    if (!with_source) continue;
    const int r = d.r0 + i / d.RC, c = d.c0 + i % d.RC;
    const long s = (long)r * d.Y + c;
    const double ux = u[s], uy = u[N + s];
    const double uF = ux * Fx[k] + uy * Fy[k];
    const long o = g.at(r, c);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double cu = ux * (double)icx(q) + uy * (double)icy(q);
      const double cF = Fx[k] * (double)icx(q) + Fy[k] * (double)icy(q);
      p[q * g.plane + o] += ((1 - 0.5 * omega) * ((a + b * cu) * cF - a * uF) * wq(q));
    }
  }
}

template <int OPT, int MAXOWN>
__global__ __launch_bounds__(1024) void k_ibm_step(IbmDev d, int m_max, const double* __restrict__ u,
                                                   const double* __restrict__ rho,
                                                   double* __restrict__ F_sum, double* __restrict__ p,
                                                   Geom g, double omega, double a, double b,
                                                   int with_source, int* flag = nullptr, int seq = 0) {
  // resident: tell the gate on the lattice stream (k_ibm_gate) that the rest of the step may start
  if (flag && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  extern __shared__ double lds[];
  ibm_force_wg<OPT, MAXOWN>(d, m_max, u, rho, F_sum, p, g, omega, a, b, with_source, lds);
}

#ifdef LBM_EXPERIMENTS  // measured level with the launch chain and not kept (DESIGN.md "experiments"): make EXPERIMENTS=1
// ---- the whole chain of a forced box in ONE launch ---------------------------------------------------------------
// D x (single BGK step on the box lattice, forcing, source term) for a block of an immersed-boundary lattice, as one
// kernel of a few workgroups that hold their compute units for the whole block: beside a grid-filling window launch
// the 3 D small dependent launches of the chain each wait for wave slots and then share their SIMDs with its waves
// (2-3 x slower, profiles/r02_ibm_block_trace_*.txt).  Every workgroup asks for the forcing's LDS, so each sits
// alone on a compute unit.  Phases are separated by a barrier over the grid: arrival counter + generation word,
// agent-scope release before / acquire after (the workgroups sit on different XCDs, whose L2s are not coherent for
// plain stores), and a BOUNDED spin -- if the grid is ever not co-resident the kernel gives up, flags it
// (lbm_ibm_surface_force and lbm_ibm_chain_status report it) and ends; nothing can hang.
// Per node the arithmetic is that of k_stream_collide_v2 / k_ibm_step / k_ibm_add_source_touched: same bits.
struct ChainSync {
  unsigned* cnt;   // arrivals at the current barrier
  unsigned* gen;   // barriers completed so far (monotonic over launches; the host passes the value at launch)
  int* abort;      // set when a spin ran out of budget
};
__device__ __forceinline__ bool chain_barrier(const ChainSync& sy, unsigned nwg, unsigned& my_gen) {
  // everything the workgroups hand to each other goes through coherent accesses (ld_d / st_d<true>): the barrier only
  // has to order them -- this wave's memory operations complete (workgroup-scope fence = s_waitcnt), the workgroup
  // meets, one lane announces it.  No agent-scope fence: that would write back and invalidate the XCD's whole L2 under
  // the window launch running beside (measured: chain 155 us per step, window launch 520 instead of 380 us).
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __syncthreads();
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned prev = __hip_atomic_fetch_add(sy.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == nwg - 1) {
      __hip_atomic_store(sy.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the reset is performed before the release of the others
      __hip_atomic_store(sy.gen, my_gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long budget = 400000;  // x (sleep + one memory round trip): some 0.1-0.4 s
      while (__hip_atomic_load(sy.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen && --budget > 0 &&
             __hip_atomic_load(sy.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __builtin_amdgcn_s_sleep(1);
      if (__hip_atomic_load(sy.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) {
        __hip_atomic_store(sy.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
      }
    }
    s_ok = ok;
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  ++my_gen;
  return ok;
}

template <class Model, int MAXOWN>
__global__ __launch_bounds__(1024) void k_ibm_box_chain(IbmDev d, int m_max, double* bx0, double* bx1, Geom g, Model m,
                                                        int D, double* xrho, double* xu,
                                                        double* F_sum, double omega, double a, double b,
                                                        ChainSync sy, unsigned gen0, int* flag, int seq) {
  extern __shared__ double lds[];
  const unsigned nwg = gridDim.x;
  unsigned my_gen = gen0;
  if (!chain_barrier(sy, nwg, my_gen)) return;  // every workgroup holds its compute unit from here on
  if (blockIdx.x == 0 && threadIdx.x == 0 && flag) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int Rb = g.R, Cb = g.C;
  const long n = (long)Rb * Cb;
  double* in = bx0;
  double* out = bx1;
  for (int k = 1; k <= D; ++k) {
    // single step on rows [k, Rb - k): cylinder_test.cpp:103-109 on the shrinking trapezoid
    const long items = (long)(Rb - 2 * k) * Cb;
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < items; i += (long)nwg * 1024) {
      const int r = k + (int)(i / Cb), c = (int)(i % Cb);
      const long rows[3] = {g.at(wrap_row(g, r + 1), 0), g.at(r, 0), g.at(wrap_row(g, r - 1), 0)};
      const int cols[3] = {wrap_col(g, c + 1), c, wrap_col(g, c - 1)};
      double f[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) f[q] = ld_d<true>(in + q * g.plane + rows[icx(q) + 1] + cols[icy(q) + 1]);
      double rho, ux, uy;
      m.collide(f, rho, ux, uy);
#pragma unroll
      for (int q = 0; q < Q; ++q) st_d<true>(out + q * g.plane + rows[1] + c, f[q]);
      const long o = (long)r * Cb + c;
      st_d<true>(xrho + o, rho);
      st_d<true>(xu + o, ux);
      st_d<true>(xu + n + o, uy);
    }
    if (!chain_barrier(sy, nwg, my_gen)) return;
    if (blockIdx.x == 0) ibm_force_wg<1, MAXOWN, true>(d, m_max, xu, xrho, F_sum, out, g, omega, a, b, 0, lds);
    if (!chain_barrier(sy, nwg, my_gen)) return;
    {  // source term on the touched nodes (== k_ibm_add_source_touched)
      const int nroi = d.RR * d.RC;
      for (int t = blockIdx.x * 1024 + threadIdx.x; t < d.n_touched; t += nwg * 1024) {
        const int i = d.touched[t];
        const int r = d.r0 + i / d.RC, c = d.c0 + i % d.RC;
        const long s = (long)r * d.Y + c, N = (long)d.X * d.Y;
        const double ux = ld_d<true>(xu + s), uy = ld_d<true>(xu + N + s), Fx = ld_d<true>(F_sum + i), Fy = ld_d<true>(F_sum + nroi + i);
        const double uF = ux * Fx + uy * Fy;
        const long o = g.at(r, c);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const double cu = ux * (double)icx(q) + uy * (double)icy(q);
          const double cF = Fx * (double)icx(q) + Fy * (double)icy(q);
          double* pq = out + q * g.plane + o;
          st_d<true>(pq, ld_d<true>(pq) + ((1 - 0.5 * omega) * ((a + b * cu) * cF - a * uF) * wq(q)));
        }
      }
    }
    if (k < D && !chain_barrier(sy, nwg, my_gen)) return;
    double* t_ = in;
    in = out;
    out = t_;
  }
}
#endif  // LBM_EXPERIMENTS

// F_s = F.reshape(-1, 2).sum(0): one block, fixed-order tree -> reproducible
__global__ __launch_bounds__(256) void k_ibm_sum(int n, const double* __restrict__ F_sum,
                                                 double* __restrict__ out2) {
  __shared__ double sx[256], sy[256];
  double ax = 0.0, ay = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    ax += F_sum[i];
    ay += F_sum[n + i];
  }
  sx[threadIdx.x] = ax;
  sy[threadIdx.x] = ay;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sx[threadIdx.x] += sx[threadIdx.x + s];
      sy[threadIdx.x] += sy[threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = sx[0];
    out2[1] = sy[0];
  }
}

}  // namespace lbm

namespace lbm {
// One wave on the lattice stream that ends as soon as the forcing workgroup of step `seq` is resident
// (or after a bounded wait).  Without it the lattice launches enqueued right behind the ROI rows fill
// every CU first, and the forcing workgroup -- 16 waves x 128 VGPRs + 135 KB LDS: a whole CU -- starts
// only when their grid drains (kernel trace: 204 us beside them against 82 us alone).
__global__ __launch_bounds__(64) void k_ibm_gate(const int* flag, int seq, int budget) {
  if (threadIdx.x != 0) return;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq && --budget > 0)
    __builtin_amdgcn_s_sleep(32);
}
}  // namespace lbm

struct lbm_ibm {
  int* flag = nullptr;  // device word: sequence number of the last forcing workgroup that became resident
  int seq = 0;          // host count of lbm_ibm_step launches of the one-workgroup kernel
  bool gate_ok = false; // the last lbm_ibm_step went through that kernel
  lbm::IbmDev d;
  int m_max, r1, c1;
  unsigned lds_opt_in = 0;  // bit per k_ibm_step instantiation that has its dynamic-LDS limit raised
  void* dev_blob;                  // all constant device arrays in one allocation
  void* dev_blob2;                 // their lane-major copies (k_ibm_step)
  unsigned* chain_sync = nullptr;  // device: {arrivals, generation, abort} of k_ibm_box_chain's grid barrier
  unsigned chain_gen = 0;          // host: barriers passed by all launches so far
  double *u_roi, *rho_roi, *F_sum, *fj, *out2;  // device work arrays
};

using namespace lbm;

extern "C" {

int lbm_ibm_create(lbm_ibm** out, const double* x, const double* y, int n_markers, int m_max,
                   int X, int Y) {
  return lbm_ibm_create_slab(out, x, y, n_markers, m_max, X, Y, 0);
}

int lbm_ibm_create_slab(lbm_ibm** out, const double* x, const double* y, int n_markers, int m_max,
                        int X, int Y, int row_offset) {
  LBM_REQUIRE(out && x && y && n_markers > 0, "lbm_ibm_create: bad argument");
  LBM_REQUIRE(m_max >= 2, "lbm_ibm_create: m_max=%d (need >= 2 for one forcing iteration)", m_max);
  // ROI, ibm.cpp:124-153
  long r_min = 1000000, r_max = 0, c_min = 1000000, c_max = 0;
  for (int i = 0; i < n_markers; ++i) {
    const int fx = (int)std::floor(x[i]), fy = (int)std::floor(y[i]);
    r_min = std::min<long>(r_min, fx - 2);
    r_max = std::max<long>(r_max, fx + 2);
    c_min = std::min<long>(c_min, fy - 2);
    c_max = std::max<long>(c_max, fy + 2);
  }
  // X rows of this block / slab start at global row `row_offset`; weights are computed from the
  // GLOBAL coordinates (as the reference does), only the stored ROI origin is slab-local
  LBM_REQUIRE(r_min - row_offset >= 1 && c_min >= 1 && r_max + 1 - row_offset <= X - 1 && c_max + 1 <= Y - 1,
              "lbm_ibm_create: ROI rows [%ld,%ld] cols [%ld,%ld] must lie strictly inside rows [%d,%d) x %d columns",
              r_min, r_max, c_min, c_max, row_offset, row_offset + X, Y);
  const int RR = (int)(r_max - r_min + 1), RC = (int)(c_max - c_min + 1), n = RR * RC;
  std::vector<int> box0(n_markers);
  std::vector<double> phi((size_t)n_markers * 16);
  std::vector<std::vector<std::pair<int, double>>> per_node(n);
  for (int j = 0; j < n_markers; ++j) {
    // marker::set_box with coordinates relative to the ROI origin (ibm.cpp:20-37, :116)
    const double xr = x[j] - (double)r_min, yr = y[j] - (double)c_min;
    const double fx = std::floor(xr), fy = std::floor(yr);
    const int row0 = (int)fx - 1, col0 = (int)fy - 1;
    box0[j] = row0 * RC + col0;
    for (int k = 0; k < 16; ++k) {
      // stencil row 0 (k%4) pairs with x, row 1 (k/4) with y, while the box is flattened
      // [row = k/4][col = k%4]: the reference's transposed kernel, kept as is
      const double sx = xr - ((double)(k % 4) + fx - 1.0);
      const double sy = yr - ((double)(k / 4) + fy - 1.0);
      const double w = peskin4(sx) * peskin4(sy);
      phi[(size_t)j * 16 + k] = w;
      per_node[(row0 + k / 4) * RC + (col0 + k % 4)].push_back({j, w});
    }
  }
  std::vector<int> csr_ptr(n + 1, 0), csr_mk;
  std::vector<double> csr_w;
  for (int i = 0; i < n; ++i) {
    for (auto& e : per_node[i]) {  // markers were visited in ascending order
      csr_mk.push_back(e.first);
      csr_w.push_back(e.second);
    }
    csr_ptr[i + 1] = (int)csr_mk.size();
  }
  const size_t nnz = csr_mk.size();
  // the nodes under at least one box, in ROI order; per-tap index into that list; CSR offsets of
  // those nodes (entries of consecutive touched nodes are NOT contiguous in csr_mk: keep begin/end
  // by storing ptr of the node and relying on csr_ptr[i + 1] of the same node)
  std::vector<int> touched, t_of(n, -1);
  for (int i = 0; i < n; ++i)
    if (csr_ptr[i + 1] > csr_ptr[i]) {
      t_of[i] = (int)touched.size();
      touched.push_back(i);
    }
  std::vector<int> tap_t((size_t)n_markers * 16), tptr(touched.size() + 1);
  for (int j = 0; j < n_markers; ++j)
    for (int k = 0; k < 16; ++k) tap_t[(size_t)j * 16 + k] = t_of[box0[j] + (k / 4) * RC + (k % 4)];
  // untouched nodes have empty CSR rows, so the touched nodes' entries ARE contiguous and in order
  for (size_t t = 0; t < touched.size(); ++t) tptr[t] = csr_ptr[touched[t]];
  tptr[touched.size()] = (int)nnz;

  // lane-major copies for k_ibm_step
  const size_t nt_ = touched.size();
  int ell_deg = 0;
  std::vector<int> ell_cnt(nt_);
  for (size_t t = 0; t < nt_; ++t) {
    ell_cnt[t] = tptr[t + 1] - tptr[t];
    ell_deg = std::max(ell_deg, ell_cnt[t]);
  }
  std::vector<int> tap_k((size_t)16 * n_markers), ell_mk((size_t)ell_deg * nt_, 0);
  std::vector<double> phi_k((size_t)16 * n_markers), ell_w((size_t)ell_deg * nt_, 0.0);
  for (int j = 0; j < n_markers; ++j)
    for (int k = 0; k < 16; ++k) {
      tap_k[(size_t)k * n_markers + j] = tap_t[(size_t)j * 16 + k];
      phi_k[(size_t)k * n_markers + j] = phi[(size_t)j * 16 + k];
    }
  for (size_t t = 0; t < nt_; ++t)
    for (int s = 0; s < ell_cnt[t]; ++s) {
      ell_mk[(size_t)s * nt_ + t] = csr_mk[tptr[t] + s];
      ell_w[(size_t)s * nt_ + t] = csr_w[tptr[t] + s];
    }

  lbm_ibm* ib = new (std::nothrow) lbm_ibm();
  LBM_REQUIRE(ib, "lbm_ibm_create: out of host memory");
  ib->m_max = m_max;
  ib->r1 = (int)r_max + 1 - row_offset;
  ib->c1 = (int)c_max + 1;
  ib->dev_blob = ib->dev_blob2 = nullptr;
  ib->u_roi = ib->rho_roi = ib->F_sum = ib->fj = ib->out2 = nullptr;
  // one blob: doubles first (8-byte aligned), then ints
  const size_t n_dbl = phi.size() + nnz, n_int = box0.size() + csr_ptr.size() + nnz + touched.size() + tap_t.size() + tptr.size();
  const size_t blob_bytes = n_dbl * 8 + n_int * 4;
  std::vector<char> host(blob_bytes);
  double* hd = reinterpret_cast<double*>(host.data());
  std::copy(phi.begin(), phi.end(), hd);
  std::copy(csr_w.begin(), csr_w.end(), hd + phi.size());
  int* hi = reinterpret_cast<int*>(host.data() + n_dbl * 8);
  std::copy(box0.begin(), box0.end(), hi);
  std::copy(csr_ptr.begin(), csr_ptr.end(), hi + box0.size());
  std::copy(csr_mk.begin(), csr_mk.end(), hi + box0.size() + csr_ptr.size());
  std::copy(touched.begin(), touched.end(), hi + box0.size() + csr_ptr.size() + nnz);
  std::copy(tap_t.begin(), tap_t.end(), hi + box0.size() + csr_ptr.size() + nnz + touched.size());
  std::copy(tptr.begin(), tptr.end(), hi + box0.size() + csr_ptr.size() + nnz + touched.size() + tap_t.size());
  hipError_t e = hipMalloc(&ib->dev_blob, blob_bytes);
  if (e == hipSuccess) e = hipMemcpy(ib->dev_blob, host.data(), blob_bytes, hipMemcpyHostToDevice);
  // second blob: doubles (phi_k, ell_w), then ints (tap_k, ell_cnt, ell_mk)
  const size_t n_dbl2 = phi_k.size() + ell_w.size(), n_int2 = tap_k.size() + ell_cnt.size() + ell_mk.size();
  std::vector<char> host2(n_dbl2 * 8 + n_int2 * 4);
  {
    double* h2 = reinterpret_cast<double*>(host2.data());
    std::copy(phi_k.begin(), phi_k.end(), h2);
    std::copy(ell_w.begin(), ell_w.end(), h2 + phi_k.size());
    int* i2 = reinterpret_cast<int*>(host2.data() + n_dbl2 * 8);
    std::copy(tap_k.begin(), tap_k.end(), i2);
    std::copy(ell_cnt.begin(), ell_cnt.end(), i2 + tap_k.size());
    std::copy(ell_mk.begin(), ell_mk.end(), i2 + tap_k.size() + ell_cnt.size());
  }
  if (e == hipSuccess) e = hipMalloc(&ib->dev_blob2, host2.size());
  if (e == hipSuccess) e = hipMemcpy(ib->dev_blob2, host2.data(), host2.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&ib->u_roi, (size_t)n * 16);
  if (e == hipSuccess) e = hipMalloc(&ib->rho_roi, (size_t)n * 8);
  if (e == hipSuccess) e = hipMalloc(&ib->F_sum, (size_t)n * 16);
  if (e == hipSuccess) e = hipMemset(ib->F_sum, 0, (size_t)n * 16);  // k_ibm_step writes touched nodes only
  if (e == hipSuccess) e = hipMalloc(&ib->fj, (size_t)n_markers * 16);
  if (e == hipSuccess) e = hipMalloc(&ib->out2, 16);
  if (e == hipSuccess) e = hipMalloc(&ib->flag, sizeof(int));
  if (e == hipSuccess) e = hipMemset(ib->flag, 0, sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&ib->chain_sync, 4 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(ib->chain_sync, 0, 4 * sizeof(unsigned));
  if (e != hipSuccess) {
    set_error("lbm_ibm_create: HIP allocation/copy failed: %s", hipGetErrorString(e));
    lbm_ibm_destroy(ib);
    return LBM_ERR_HIP;
  }
  char* base = static_cast<char*>(ib->dev_blob);
  const double* dd = reinterpret_cast<const double*>(base);
  const int* di = reinterpret_cast<const int*>(base + n_dbl * 8);
  ib->d = IbmDev{n_markers, RR, RC, (int)r_min - row_offset, (int)c_min, X, Y, di, dd,
                 di + box0.size(), di + box0.size() + csr_ptr.size(), dd + phi.size(),
                 di + box0.size() + csr_ptr.size() + nnz, (int)touched.size(),
                 di + box0.size() + csr_ptr.size() + nnz + touched.size(),
                 di + box0.size() + csr_ptr.size() + nnz + touched.size() + tap_t.size(), 0.0, 0.0, 0,
                 nullptr, nullptr, nullptr, nullptr, nullptr, ell_deg};
  {
    const double* d2 = reinterpret_cast<const double*>(ib->dev_blob2);
    const int* i2 = reinterpret_cast<const int*>(static_cast<char*>(ib->dev_blob2) + n_dbl2 * 8);
    ib->d.phi_k = d2;
    ib->d.ell_w = d2 + phi_k.size();
    ib->d.tap_k = i2;
    ib->d.ell_cnt = i2 + tap_k.size();
    ib->d.ell_mk = i2 + tap_k.size() + ell_cnt.size();
  }
  *out = ib;
  return LBM_OK;
}

int lbm_ibm_destroy(lbm_ibm* ib) {
  if (!ib) return LBM_OK;
  for (void* p : {ib->dev_blob, ib->dev_blob2, (void*)ib->u_roi, (void*)ib->rho_roi, (void*)ib->F_sum,
                  (void*)ib->fj, (void*)ib->out2, (void*)ib->flag, (void*)ib->chain_sync})
    if (p) (void)hipFree(p);
  delete ib;
  return LBM_OK;
}

int lbm_ibm_set_velocity(lbm_ibm* ib, double Ur, double Uc) {
  LBM_REQUIRE(ib, "lbm_ibm_set_velocity: NULL boundary");
  ib->d.Ubx = Ur;
  ib->d.Uby = Uc;
  ib->d.moving = (Ur != 0.0 || Uc != 0.0) ? 1 : 0;
  return LBM_OK;
}

int lbm_ibm_roi(const lbm_ibm* ib, int* r0, int* r1, int* c0, int* c1) {
  LBM_REQUIRE(ib && r0 && r1 && c0 && c1, "lbm_ibm_roi: NULL argument");
  *r0 = ib->d.r0;
  *r1 = ib->r1;
  *c0 = ib->d.c0;
  *c1 = ib->c1;
  return LBM_OK;
}

int lbm_ibm_force(lbm_ibm* ib, const double* u, const double* rho, double* F_out, lbm_stream_t s) {
  LBM_REQUIRE(ib && u && rho, "lbm_ibm_force: NULL argument");
  hipStream_t st = as_stream(s);
  const int n = ib->d.RR * ib->d.RC, nb = (n + 255) / 256;
  LBM_KLAUNCH(k_ibm_begin, dim3(nb), dim3(256), 0, st, ib->d, u, rho, ib->u_roi, ib->rho_roi, ib->F_sum);
  LBM_CHECK_LAUNCH();
  for (int it = 1; it < ib->m_max; ++it) {  // ibm.cpp:166
    LBM_KLAUNCH(k_ibm_interp, dim3((ib->d.n_markers + 63) / 64), dim3(64), 0, st, ib->d, ib->u_roi,
                ib->rho_roi, ib->fj);
    LBM_CHECK_LAUNCH();
    LBM_KLAUNCH(k_ibm_spread, dim3(nb), dim3(256), 0, st, ib->d, ib->fj, ib->u_roi, ib->rho_roi, ib->F_sum);
    LBM_CHECK_LAUNCH();
  }
  if (F_out)
    LBM_CHECK_HIP(hipMemcpyAsync(F_out, ib->F_sum, (size_t)n * 16, hipMemcpyDeviceToDevice, st));
  return LBM_OK;
}

int lbm_ibm_add_source(lbm_ibm* ib, double* p, const lbm_geom* g, const double* u, double omega,
                       double a, double b, lbm_stream_t s) {
  LBM_REQUIRE(ib && p && g && u, "lbm_ibm_add_source: NULL argument");
  // a slab passes its own geometry (ghost rows allowed) and created the boundary in slab-local
  // row coordinates; u is the slab's [2][R][C] moment field (no ghost rows)
  LBM_REQUIRE(g->R == ib->d.X && g->C == ib->d.Y,
              "lbm_ibm_add_source: lattice %dx%d does not match the boundary's %dx%d", g->R, g->C,
              ib->d.X, ib->d.Y);
  const int n = ib->d.RR * ib->d.RC;
  LBM_KLAUNCH(k_ibm_add_source, dim3((n + 255) / 256), dim3(256), 0, as_stream(s), ib->d, p,
              make_geom(*g), u, ib->F_sum, omega, a, b);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_ibm_step(lbm_ibm* ib, double* p, const lbm_geom* g, const double* u, const double* rho,
                 double omega, double a, double b, lbm_stream_t s) {
  LBM_REQUIRE(ib && p && g && u && rho, "lbm_ibm_step: NULL argument");
  LBM_REQUIRE(g->R == ib->d.X && g->C == ib->d.Y,
              "lbm_ibm_step: lattice %dx%d does not match the boundary's %dx%d", g->R, g->C, ib->d.X, ib->d.Y);
  const size_t lds = ((size_t)3 * ib->d.n_touched + 2 * (size_t)ib->d.n_markers) * sizeof(double);
  if (ib->d.n_touched > 8 * 1024 || lds > 150 * 1024 || tuning("ibm_step_chain", 0)) {  // large boundaries: the launch chain
    ib->gate_ok = false;
    int rc = lbm_ibm_force(ib, u, rho, nullptr, s);
    if (!rc) rc = lbm_ibm_add_source(ib, p, g, u, omega, a, b, s);
    return rc;
  }
  ++ib->seq;
  ib->gate_ok = true;
  const int own = ib->d.n_touched <= 4096 ? 4 : (ib->d.n_touched <= 6144 ? 6 : 8);  // touched nodes per thread
  const int variant = (tuning("ibm_step_opt", 1) ? 10 : 0) + own;
  // the source term as a second, many-workgroup launch: 9 scattered read-modify-writes per touched node from ONE
  // compute unit cost 16 of the 60 us (profiles/r02_ibm_force.txt)
  const bool split = tuning("ibm_step_split", 1) != 0;
  auto go = [&](auto kern) -> int {
    if (!(ib->lds_opt_in & (1u << (variant % 10 / 2 + (variant >= 10 ? 4 : 0))))) {
      LBM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      ib->lds_opt_in |= 1u << (variant % 10 / 2 + (variant >= 10 ? 4 : 0));
    }
    LBM_KLAUNCH(kern, dim3(1), dim3(1024), lds, as_stream(s), ib->d, ib->m_max, u, rho, ib->F_sum, p, make_geom(*g), omega,
                a, b, split ? 0 : 1, ib->flag, ib->seq);
    if (split)
      LBM_KLAUNCH(k_ibm_add_source_touched, dim3((ib->d.n_touched + 255) / 256), dim3(256), 0, as_stream(s), ib->d, p,
                  make_geom(*g), u, ib->F_sum, omega, a, b);
    return LBM_OK;
  };
  int rc = LBM_OK;
  switch (variant) {
    case 4: rc = go(&k_ibm_step<0, 4>); break;
    case 6: rc = go(&k_ibm_step<0, 6>); break;
    case 8: rc = go(&k_ibm_step<0, 8>); break;
    case 14: rc = go(&k_ibm_step<1, 4>); break;
    case 16: rc = go(&k_ibm_step<1, 6>); break;
    default: rc = go(&k_ibm_step<1, 8>); break;
  }
  if (rc) return rc;
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

}  // extern "C"
namespace lbm {
int ibm_step_window(lbm_ibm* ib, int row_off, int col_off, double* p, const lbm_geom* g, const double* u,
                    const double* rho, double omega, double a, double b, hipStream_t st) {
  LBM_REQUIRE(ib && g, "ibm_step_window: NULL argument");
  LBM_REQUIRE(ib->d.r0 - row_off >= 1 && ib->d.c0 - col_off >= 1 && ib->r1 - row_off <= g->R - 1 && ib->c1 - col_off <= g->C - 1,
              "ibm_step_window: ROI outside the window");
  lbm_ibm w = *ib;  // shares every device array; never destroyed
  w.d.r0 -= row_off;
  w.d.c0 -= col_off;
  w.r1 -= row_off;
  w.c1 -= col_off;
  w.d.X = g->R;
  w.d.Y = g->C;
  const int rc = lbm_ibm_step(&w, p, g, u, rho, omega, a, b, reinterpret_cast<lbm_stream_t>(st));
  ib->seq = w.seq;
  ib->gate_ok = w.gate_ok;
  ib->lds_opt_in = w.lds_opt_in;
  return rc;
}
// D forced single steps of a box lattice pair in one launch (k_ibm_box_chain); *cur: index of the lattice holding the
// state (in: time t, out: time t + D).  Returns 1 when the boundary does not qualify (too many touched nodes for the
// one-workgroup forcing, or switched off): the caller runs the launch chain instead.
int ibm_box_chain(lbm_ibm* ib, int row_off, int col_off, double* const box[2], int* cur, const lbm_geom* g,
                  const lbm_bgk_params* prm, bool fast_model, int D, double* xrho, double* xu, double a, double b,
                  hipStream_t st) {
#ifndef LBM_EXPERIMENTS
  (void)ib; (void)row_off; (void)col_off; (void)box; (void)cur; (void)g; (void)prm; (void)fast_model; (void)D; (void)xrho; (void)xu; (void)a; (void)b; (void)st;
  return 1;  // the one-launch chain is an experiment (make EXPERIMENTS=1): callers run the launch chain
#else
  LBM_REQUIRE(ib && box && cur && g && prm, "ibm_box_chain: NULL argument");
  const size_t lds = ((size_t)3 * ib->d.n_touched + 2 * (size_t)ib->d.n_markers) * sizeof(double);
  const int nwg = tuning("ibm_chain_wgs", 16);
  if (nwg < 2 || nwg > 64 || ib->d.n_touched > 6144 || lds > 150 * 1024 || g->ghost != 0) return 1;
  LBM_REQUIRE(ib->d.r0 - row_off >= 1 && ib->d.c0 - col_off >= 1 && ib->r1 - row_off <= g->R - 1 && ib->c1 - col_off <= g->C - 1,
              "ibm_box_chain: ROI outside the window");
  IbmDev d = ib->d;
  d.r0 -= row_off;
  d.c0 -= col_off;
  d.X = g->R;
  d.Y = g->C;
  ChainSync sy{ib->chain_sync, ib->chain_sync + 1, reinterpret_cast<int*>(ib->chain_sync + 2)};
  const unsigned gen0 = ib->chain_gen;
  ib->chain_gen += 3u * (unsigned)D;  // 1 at entry + 3 per step - 1 (none behind the last source term)
  ++ib->seq;
  ib->gate_ok = true;
  const Geom gg = make_geom(*g);
  const int own = ib->d.n_touched <= 4096 ? 4 : 6;
  auto go = [&](auto kern, auto model) -> int {
    const unsigned bit = 1u << (16 + (fast_model ? 2 : 0) + (own == 4 ? 0 : 1));
    if (!(ib->lds_opt_in & bit)) {
      LBM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      ib->lds_opt_in |= bit;
    }
    // (at least 100 KB each: one workgroup per compute unit, whatever the forcing itself needs)
    const size_t want = lds > 100 * 1024 ? lds : 100 * 1024;
    LBM_KLAUNCH(kern, dim3(nwg), dim3(1024), want, st, d, ib->m_max, box[*cur], box[*cur ^ 1], gg, model, D, xrho, xu, ib->F_sum,
                prm->omega, a, b, sy, gen0, ib->flag, ib->seq);
    LBM_CHECK_LAUNCH();
    return LBM_OK;
  };
  int rc;
  if (fast_model) {
    const BgkFastModel m(prm->omega);
    rc = own == 4 ? go(&k_ibm_box_chain<BgkFastModel, 4>, m) : go(&k_ibm_box_chain<BgkFastModel, 6>, m);
  } else {
    const BgkModel m{prm->omega, prm->incompressible, prm->delta_form, prm->force_mode, prm->force_r, prm->force_c, prm->guo_a, prm->guo_b};
    rc = own == 4 ? go(&k_ibm_box_chain<BgkModel, 4>, m) : go(&k_ibm_box_chain<BgkModel, 6>, m);
  }
  if (rc) return rc;
  if (D % 2) *cur ^= 1;
  return LBM_OK;
#endif  // LBM_EXPERIMENTS
}
// 0: no launch of k_ibm_box_chain ever gave up at its grid barrier (synchronises the device word read)
int ibm_chain_status(lbm_ibm* ib, hipStream_t st) {
  int flag = 0;
  LBM_CHECK_HIP(hipMemcpyAsync(&flag, ib->chain_sync + 2, sizeof flag, hipMemcpyDeviceToHost, st));
  LBM_CHECK_HIP(hipStreamSynchronize(st));
  LBM_REQUIRE(flag == 0, "immersed boundary: a forced-box launch gave up at its grid barrier (its workgroups were not co-resident); results since then are invalid");
  return LBM_OK;
}
int ibm_gate(lbm_ibm* ib, hipStream_t st) {
  if (!ib || !ib->gate_ok || tuning("ibm_gate", 1) == 0) return LBM_OK;
  LBM_KLAUNCH(k_ibm_gate, dim3(1), dim3(64), 0, st, ib->flag, ib->seq, 4000);  // <= ~4 ms, then gives up
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}
}  // namespace lbm
extern "C" {

int lbm_ibm_surface_force(lbm_ibm* ib, double* out2, lbm_stream_t s) {
  LBM_REQUIRE(ib && out2, "lbm_ibm_surface_force: NULL argument");
  if (ib->chain_gen) {
    const int rc = ibm_chain_status(ib, as_stream(s));
    if (rc) return rc;
  }
  LBM_KLAUNCH(k_ibm_sum, dim3(1), dim3(256), 0, as_stream(s), ib->d.RR * ib->d.RC, ib->F_sum, ib->out2);
  LBM_CHECK_LAUNCH();
  LBM_CHECK_HIP(hipMemcpyAsync(out2, ib->out2, 16, hipMemcpyDeviceToHost, as_stream(s)));
  LBM_CHECK_HIP(hipStreamSynchronize(as_stream(s)));
  return LBM_OK;
}

}  // extern "C"
