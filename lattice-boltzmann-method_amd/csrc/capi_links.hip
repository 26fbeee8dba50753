// C ABI, part 8: population links between lattices on one GPU -- the reference's multi-block
// "binding" (test/decompose_domain.cpp:181-187, test/decompose_domain_loop.cpp:235-261) and its
// slice-assignment walls (:173-231) as ONE gather launch per time step instead of ~70 thin
// index_put_ launches.  A table is built once from slice descriptions in the order the driver
// executes them; where two slices write the same destination element the later one wins, which is
// resolved on the host, so the device pass is a race-free element-wise gather
//     dst[dst_lattice][dst_index] = src[src_lattice][src_index].
// Also: the uniform momentum source of decompose_domain_loop.cpp:152-160 on a row window.
#include <new>
#include <unordered_map>
#include <vector>

#include "d2q9.hpp"
#include "internal.hpp"

namespace lbm {

constexpr int kMaxLinkLattices = 8;
struct LinkPtrs {
  double* dst[kMaxLinkLattices];
  const double* src[kMaxLinkLattices];
};

__global__ __launch_bounds__(256) void k_links_apply(LinkPtrs p, int n, const int* __restrict__ lat,
                                                     const long* __restrict__ dst_off,
                                                     const long* __restrict__ src_off) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int l = lat[k];  // low 4 bits: destination lattice, next 4: source lattice
  p.dst[l & 15][dst_off[k]] = p.src[(l >> 4) & 15][src_off[k]];
}

// affine links: dst = scale * src (+ addends[add_idx]); scale = -1 is the anti-bounce-back
// "-f_coll + term" of rectangle_sedimentation_test.cpp:152-170, 210-230 (exact: -1 * x = -x)
__global__ __launch_bounds__(256) void k_links_apply_affine(LinkPtrs p, int n, const int* __restrict__ lat,
                                                            const long* __restrict__ dst_off,
                                                            const long* __restrict__ src_off,
                                                            const double* __restrict__ scale,
                                                            const long* __restrict__ add_idx,
                                                            const double* __restrict__ addends) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int l = lat[k];
  double v = scale[k] * p.src[(l >> 4) & 15][src_off[k]];
  if (add_idx[k] >= 0) v = v + addends[add_idx[k]];
  p.dst[l & 15][dst_off[k]] = v;
}

// wall terms of the sedimentation driver, one row of 9 per lattice row r, from the wall velocity
//   uw = wa * u[:, r, col_a] + wb * u[:, r, col_b] + shift      (u: moment field [2][X][Y])
// mode 0: factor * ((2 + 9 (uw.c_q)^2) - 3 uw.uw) w_q                              (:135, :149)
// mode 1: factor * ((((1 + 3 uw.c_q) + 4.5 (uw.c_q)^2) - 1.5 uw.uw) w_q) * field[r]  (:202-208)
__global__ __launch_bounds__(256) void k_wall_terms(double* __restrict__ out, const double* __restrict__ u,
                                                    int X, int Y, int col_a, double wa, int col_b,
                                                    double wb, double shift, int mode,
                                                    const double* __restrict__ field, double factor) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= X) return;
  const long N = (long)X * Y, a = (long)r * Y + col_a, b = (long)r * Y + col_b;
  double w0, w1;
  if (wb != 0.0) {
    w0 = wa * u[a] - (-wb) * u[b];  // the driver writes 1.5 u[-1] - 0.5 u[-2]
    w1 = wa * u[N + a] - (-wb) * u[N + b];
  } else {
    w0 = wa * u[a];
    w1 = wa * u[N + a];
  }
  w0 = w0 + shift;
  w1 = w1 + shift;
  const double uu = w0 * w0 + w1 * w1;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double uc = w0 * (double)icx(q) + w1 * (double)icy(q);
    double t;
    if (mode == 0) t = ((2.0 + 9.0 * (uc * uc)) - 3.0 * uu) * wq(q);
    else t = ((((1.0 + 3.0 * uc) + 4.5 * (uc * uc)) - 1.5 * uu) * wq(q)) * field[r];
    out[(long)r * Q + q] = factor * t;
  }
}

// one pressure-periodic virtual row between two lattices (test/decompose_domain.cpp:50-73; with
// dst == src lattice: horizontal_poiseuille_test.cpp:25-45 at the operator level):
//   coll_dst[dst_row] = (feq(rho_bc, u_src[src_row]) + coll_src[src_row]) - equi_src[src_row]
__global__ __launch_bounds__(256) void k_pressure_row(double* __restrict__ coll_dst, Geom gd, int dst_row,
                                                      const double* __restrict__ coll_src,
                                                      const double* __restrict__ equi_src,
                                                      const double* __restrict__ u_src, Geom gs,
                                                      int src_row, double rho_bc, int incompressible) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= gd.C) return;
  const long Ns = (long)gs.R * gs.C, su = (long)src_row * gs.C + c;
  const BgkModel m{1.0, incompressible, 0, 0, 0.0, 0.0, 0.0, 0.0};
  double te[Q];
  m.feq(te, rho_bc * 1.0, u_src[su], u_src[Ns + su]);
  const long od = gd.at(dst_row, c), os = gs.at(src_row, c);
#pragma unroll
  for (int q = 0; q < Q; ++q)
    coll_dst[q * gd.plane + od] = (te[q] + coll_src[q * gs.plane + os]) - equi_src[q * gs.plane + os];
}

__global__ __launch_bounds__(256) void k_axpb(double* __restrict__ out, const double* __restrict__ in,
                                              double a, double b, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a * in[i] + b;
}

// p[q][r][c] += ((1 - omega/2) ((a + b u.c_q)(F.c_q) - a u.F)) w_q on rows [row_begin, row_end)
__global__ __launch_bounds__(256) void k_bgk_add_force_rows(double* __restrict__ p, Geom g,
                                                            const double* __restrict__ u, double omega,
                                                            double Fr, double Fc, double a, double b,
                                                            int row_begin, int row_end) {
  const long n = (long)(row_end - row_begin) * g.C, N = (long)g.R * g.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = row_begin + (int)(i / g.C), c = (int)(i % g.C);
    const long s = (long)r * g.C + c, o = g.at(r, c);
    const double ux = u[s], uy = u[N + s];
    const double uF = ux * Fr + uy * Fc;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double cu = ux * (double)icx(q) + uy * (double)icy(q);
      const double cF = Fr * (double)icx(q) + Fc * (double)icy(q);
      p[q * g.plane + o] = p[q * g.plane + o] + ((1 - 0.5 * omega) * ((a + b * cu) * cF - a * uF) * wq(q));
    }
  }
}

}  // namespace lbm

struct lbm_links {
  std::vector<lbm_geom> geoms;
  // build phase: destination element -> source element, in insertion order of first appearance
  std::unordered_map<unsigned long long, size_t> slot;
  std::vector<int> lat;
  std::vector<long> dst_off, src_off, add_idx;
  std::vector<double> scale;
  bool affine = false;  // any link with scale != 1 or an addend
  int n = 0;
  int* d_lat = nullptr;
  long *d_dst = nullptr, *d_src = nullptr, *d_add = nullptr;
  double* d_scale = nullptr;
};

using namespace lbm;

extern "C" {

int lbm_links_create(lbm_links** out, int n_lattices, const lbm_geom* geoms) {
  LBM_REQUIRE(out && geoms && n_lattices >= 1 && n_lattices <= kMaxLinkLattices,
              "lbm_links_create: 1..%d lattices", kMaxLinkLattices);
  for (int i = 0; i < n_lattices; ++i) {
    LBM_REQUIRE(geoms[i].R > 0 && geoms[i].C > 0 && geoms[i].ghost >= 0, "lbm_links_create: bad geometry %d", i);
    LBM_REQUIRE(geoms[i].row_pitch == 0 || geoms[i].row_pitch == geoms[i].C, "lbm_links_create: dense rows only (lattice %d)", i);
  }
  lbm_links* t = new (std::nothrow) lbm_links();
  LBM_REQUIRE(t, "lbm_links_create: out of host memory");
  t->geoms.assign(geoms, geoms + n_lattices);
  *out = t;
  return LBM_OK;
}

int lbm_links_add(lbm_links* t, int dst_lat, int q_dst, int r0, int c0, int dr, int dc, int src_lat,
                  int q_src, int sr0, int sc0, int sdr, int sdc, int count) {
  return lbm_links_add_affine(t, dst_lat, q_dst, r0, c0, dr, dc, src_lat, q_src, sr0, sc0, sdr, sdc, count,
                              1.0, -1, 0);
}

int lbm_links_add_affine(lbm_links* t, int dst_lat, int q_dst, int r0, int c0, int dr, int dc,
                         int src_lat, int q_src, int sr0, int sc0, int sdr, int sdc, int count,
                         double scale, long long add0, long long add_stride) {
  LBM_REQUIRE(t && !t->d_lat, "lbm_links_add: NULL or already finalized table");
  if (scale != 1.0 || add0 >= 0) t->affine = true;
  const int nl = (int)t->geoms.size();
  LBM_REQUIRE(dst_lat >= 0 && dst_lat < nl && src_lat >= 0 && src_lat < nl, "lbm_links_add: lattice index");
  LBM_REQUIRE(q_dst >= 0 && q_dst < 9 && q_src >= 0 && q_src < 9 && count >= 0, "lbm_links_add: bad population / count");
  const Geom gd = make_geom(t->geoms[dst_lat]), gs = make_geom(t->geoms[src_lat]);
  for (int k : {0, count - 1}) {  // the slice is linear: its two ends decide; nothing is added on failure
    if (count == 0) break;
    const int r = r0 + k * dr, c = c0 + k * dc, sr = sr0 + k * sdr, sc = sc0 + k * sdc;
    LBM_REQUIRE(r >= 0 && r < gd.R && c >= 0 && c < gd.C && sr >= 0 && sr < gs.R && sc >= 0 && sc < gs.C,
                "lbm_links_add: slice leaves the lattice (dst %d,%d of %dx%d; src %d,%d of %dx%d)", r, c, gd.R, gd.C, sr, sc, gs.R, gs.C);
  }
  for (int k = 0; k < count; ++k) {
    const int r = r0 + k * dr, c = c0 + k * dc, sr = sr0 + k * sdr, sc = sc0 + k * sdc;
    const long d = q_dst * gd.plane + gd.at(r, c), s = q_src * gs.plane + gs.at(sr, sc);
    const unsigned long long key = ((unsigned long long)dst_lat << 56) | (unsigned long long)d;
    auto it = t->slot.find(key);
    const long ai = add0 >= 0 ? (long)(add0 + (long long)k * add_stride) : -1;
    if (it == t->slot.end()) {
      t->slot.emplace(key, t->lat.size());
      t->lat.push_back(dst_lat | (src_lat << 4));
      t->dst_off.push_back(d);
      t->src_off.push_back(s);
      t->scale.push_back(scale);
      t->add_idx.push_back(ai);
    } else {  // a later slice assignment overrides an earlier one at the same element
      t->lat[it->second] = dst_lat | (src_lat << 4);
      t->src_off[it->second] = s;
      t->scale[it->second] = scale;
      t->add_idx[it->second] = ai;
    }
  }
  return LBM_OK;
}

int lbm_links_finalize(lbm_links* t) {
  LBM_REQUIRE(t && !t->d_lat, "lbm_links_finalize: NULL or already finalized table");
  t->n = (int)t->lat.size();
  const size_t n = t->n ? t->n : 1;
  LBM_CHECK_HIP(hipMalloc(&t->d_lat, n * sizeof(int)));
  LBM_CHECK_HIP(hipMalloc(&t->d_dst, n * sizeof(long)));
  LBM_CHECK_HIP(hipMalloc(&t->d_src, n * sizeof(long)));
  if (t->affine) {
    LBM_CHECK_HIP(hipMalloc(&t->d_add, n * sizeof(long)));
    LBM_CHECK_HIP(hipMalloc(&t->d_scale, n * sizeof(double)));
    if (t->n) {
      LBM_CHECK_HIP(hipMemcpy(t->d_add, t->add_idx.data(), t->n * sizeof(long), hipMemcpyHostToDevice));
      LBM_CHECK_HIP(hipMemcpy(t->d_scale, t->scale.data(), t->n * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  if (t->n) {
    LBM_CHECK_HIP(hipMemcpy(t->d_lat, t->lat.data(), t->n * sizeof(int), hipMemcpyHostToDevice));
    LBM_CHECK_HIP(hipMemcpy(t->d_dst, t->dst_off.data(), t->n * sizeof(long), hipMemcpyHostToDevice));
    LBM_CHECK_HIP(hipMemcpy(t->d_src, t->src_off.data(), t->n * sizeof(long), hipMemcpyHostToDevice));
  }
  t->slot.clear();
  return LBM_OK;
}

int lbm_links_count(const lbm_links* t) { return t ? (t->d_lat ? t->n : (int)t->lat.size()) : 0; }

int lbm_links_apply(lbm_links* t, double* const* dst, const double* const* src, lbm_stream_t s) {
  return lbm_links_apply_affine(t, dst, src, nullptr, s);
}

int lbm_links_apply_affine(lbm_links* t, double* const* dst, const double* const* src,
                           const double* addends, lbm_stream_t s) {
  LBM_REQUIRE(t && t->d_lat && dst && src, "lbm_links_apply: NULL argument or table not finalized");
  if (!t->n) return LBM_OK;
  if (t->affine) {
    bool needs = false;
    for (long a : t->add_idx) needs = needs || a >= 0;
    LBM_REQUIRE(!needs || addends, "lbm_links_apply_affine: the table has addend links but addends is NULL");
  }
  LinkPtrs p{};
  for (size_t i = 0; i < t->geoms.size(); ++i) {
    LBM_REQUIRE(dst[i] && src[i], "lbm_links_apply: NULL lattice %d", (int)i);
    p.dst[i] = dst[i];
    p.src[i] = src[i];
  }
  if (t->affine)
    LBM_KLAUNCH(k_links_apply_affine, dim3((t->n + 255) / 256), dim3(256), 0, as_stream(s), p, t->n, t->d_lat, t->d_dst, t->d_src, t->d_scale, t->d_add, addends);
  else
    LBM_KLAUNCH(k_links_apply, dim3((t->n + 255) / 256), dim3(256), 0, as_stream(s), p, t->n, t->d_lat, t->d_dst, t->d_src);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_links_destroy(lbm_links* t) {
  if (!t) return LBM_OK;
  for (void* p : {(void*)t->d_lat, (void*)t->d_dst, (void*)t->d_src, (void*)t->d_add, (void*)t->d_scale})
    if (p) (void)hipFree(p);
  delete t;
  return LBM_OK;
}

int lbm_wall_terms(double* out, const double* u, int X, int Y, int col_a, double wa, int col_b, double wb,
                   double shift, int mode, const double* field, double factor, lbm_stream_t s) {
  LBM_REQUIRE(out && u && X > 0 && Y > 0, "lbm_wall_terms: bad argument");
  LBM_REQUIRE(col_a >= 0 && col_a < Y && col_b >= 0 && col_b < Y, "lbm_wall_terms: column outside the lattice");
  LBM_REQUIRE(mode == 0 || (mode == 1 && field), "lbm_wall_terms: mode %d (1 needs the per-row field)", mode);
  LBM_KLAUNCH(k_wall_terms, dim3((X + 255) / 256), dim3(256), 0, as_stream(s), out, u, X, Y, col_a, wa, col_b,
              wb, shift, mode, field, factor);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_pressure_row(double* coll_dst, const lbm_geom* g_dst, int dst_row, const double* coll_src,
                     const double* equi_src, const double* u_src, const lbm_geom* g_src, int src_row,
                     double rho_bc, int incompressible, lbm_stream_t s) {
  LBM_REQUIRE(coll_dst && g_dst && coll_src && equi_src && u_src && g_src, "lbm_pressure_row: NULL argument");
  LBM_REQUIRE(g_dst->C == g_src->C && g_dst->ghost == 0 && g_src->ghost == 0, "lbm_pressure_row: blocks of equal width without ghost rows");
  LBM_REQUIRE(dst_row >= 0 && dst_row < g_dst->R && src_row >= 0 && src_row < g_src->R, "lbm_pressure_row: row outside the block");
  LBM_KLAUNCH(k_pressure_row, dim3((g_dst->C + 255) / 256), dim3(256), 0, as_stream(s), coll_dst, make_geom(*g_dst),
              dst_row, coll_src, equi_src, u_src, make_geom(*g_src), src_row, rho_bc, incompressible);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_axpb(double* out, const double* in, double a, double b, long long n, lbm_stream_t s) {
  LBM_REQUIRE(out && in && n >= 0, "lbm_axpb: bad argument");
  if (!n) return LBM_OK;
  LBM_KLAUNCH(k_axpb, dim3(capped_grid((n + 255) / 256)), dim3(256), 0, as_stream(s), out, in, a, b, (long)n);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_bgk_add_force_rows(double* p, const lbm_geom* g, const double* u, double omega, double Fr,
                           double Fc, double a, double b, int row_begin, int row_end, lbm_stream_t s) {
  LBM_REQUIRE(p && g && u, "lbm_bgk_add_force_rows: NULL argument");
  LBM_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= g->R, "lbm_bgk_add_force_rows: row range [%d, %d) outside [0, %d)", row_begin, row_end, g->R);
  if (row_begin == row_end) return LBM_OK;
  const long n = (long)(row_end - row_begin) * g->C;
  LBM_KLAUNCH(k_bgk_add_force_rows, dim3(capped_grid((n + 255) / 256)), dim3(256), 0, as_stream(s), p,
              make_geom(*g), u, omega, Fr, Fc, a, b, row_begin, row_end);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

}  // extern "C"
