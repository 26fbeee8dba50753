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
  std::vector<long> dst_off, src_off;
  int n = 0;
  int* d_lat = nullptr;
  long *d_dst = nullptr, *d_src = nullptr;
};

using namespace lbm;

extern "C" {

int lbm_links_create(lbm_links** out, int n_lattices, const lbm_geom* geoms) {
  LBM_REQUIRE(out && geoms && n_lattices >= 1 && n_lattices <= kMaxLinkLattices,
              "lbm_links_create: 1..%d lattices", kMaxLinkLattices);
  for (int i = 0; i < n_lattices; ++i)
    LBM_REQUIRE(geoms[i].R > 0 && geoms[i].C > 0 && geoms[i].ghost >= 0, "lbm_links_create: bad geometry %d", i);
  lbm_links* t = new (std::nothrow) lbm_links();
  LBM_REQUIRE(t, "lbm_links_create: out of host memory");
  t->geoms.assign(geoms, geoms + n_lattices);
  *out = t;
  return LBM_OK;
}

int lbm_links_add(lbm_links* t, int dst_lat, int q_dst, int r0, int c0, int dr, int dc, int src_lat,
                  int q_src, int sr0, int sc0, int sdr, int sdc, int count) {
  LBM_REQUIRE(t && !t->d_lat, "lbm_links_add: NULL or already finalized table");
  const int nl = (int)t->geoms.size();
  LBM_REQUIRE(dst_lat >= 0 && dst_lat < nl && src_lat >= 0 && src_lat < nl, "lbm_links_add: lattice index");
  LBM_REQUIRE(q_dst >= 0 && q_dst < 9 && q_src >= 0 && q_src < 9 && count >= 0, "lbm_links_add: bad population / count");
  const Geom gd = make_geom(t->geoms[dst_lat]), gs = make_geom(t->geoms[src_lat]);
  for (int k = 0; k < count; ++k) {
    const int r = r0 + k * dr, c = c0 + k * dc, sr = sr0 + k * sdr, sc = sc0 + k * sdc;
    LBM_REQUIRE(r >= 0 && r < gd.R && c >= 0 && c < gd.C && sr >= 0 && sr < gs.R && sc >= 0 && sc < gs.C,
                "lbm_links_add: slice leaves the lattice (dst %d,%d of %dx%d; src %d,%d of %dx%d)", r, c, gd.R, gd.C, sr, sc, gs.R, gs.C);
    const long d = q_dst * gd.plane + gd.at(r, c), s = q_src * gs.plane + gs.at(sr, sc);
    const unsigned long long key = ((unsigned long long)dst_lat << 56) | (unsigned long long)d;
    auto it = t->slot.find(key);
    if (it == t->slot.end()) {
      t->slot.emplace(key, t->lat.size());
      t->lat.push_back(dst_lat | (src_lat << 4));
      t->dst_off.push_back(d);
      t->src_off.push_back(s);
    } else {  // a later slice assignment overrides an earlier one at the same element
      t->lat[it->second] = dst_lat | (src_lat << 4);
      t->src_off[it->second] = s;
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
  LBM_REQUIRE(t && t->d_lat && dst && src, "lbm_links_apply: NULL argument or table not finalized");
  if (!t->n) return LBM_OK;
  LinkPtrs p{};
  for (size_t i = 0; i < t->geoms.size(); ++i) {
    LBM_REQUIRE(dst[i] && src[i], "lbm_links_apply: NULL lattice %d", (int)i);
    p.dst[i] = dst[i];
    p.src[i] = src[i];
  }
  LBM_KLAUNCH(k_links_apply, dim3((t->n + 255) / 256), dim3(256), 0, as_stream(s), p, t->n, t->d_lat, t->d_dst, t->d_src);
  LBM_CHECK_LAUNCH();
  return LBM_OK;
}

int lbm_links_destroy(lbm_links* t) {
  if (!t) return LBM_OK;
  for (void* p : {(void*)t->d_lat, (void*)t->d_dst, (void*)t->d_src})
    if (p) (void)hipFree(p);
  delete t;
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
