// Known-bytes micro-kernels to calibrate rocprofv3's FETCH_SIZE on gfx950 for the access shapes of this engine
// ("is FETCH_SIZE x 2 right for 8-byte-shifted 512-B row reads?").  Build: hipcc --offload-arch=gfx950
// -O3 fetch_calib.hip -o fetch_calib; run under `rocprofv3 --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum).
// Each kernel reads a region ONCE (the buffer is far larger than L2 + Infinity Cache between re-touches); the host prints,
// per kernel, the bytes it asked for, the bytes of the 64-B sectors and of the 128-B lines those loads touch, with and
// without sharing between neighbouring waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

// one wave per 512-B segment, 8 B per lane, perfectly aligned stream
__global__ __launch_bounds__(256) void k_calib_aligned8(const double* __restrict__ p, double* out, long n) {
  double acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += p[i];
  if (acc == 1.2345e300) out[0] = acc;
}
// 16 B per lane (the shape the guide calibrated on)
__global__ __launch_bounds__(256) void k_calib_aligned16(const double2* __restrict__ p, double* out, long n2) {
  double acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) { double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 1.2345e300) out[0] = acc;
}
// the strip kernels' shape: a wave owns a 64-column window starting at column c0 + s * step - back and walks down the
// rows of a [R][C] plane, one load per lane per row at column shift `sh` (-1, 0, +1: the pull of cy = +1, 0, -1)
template <int NSH>
__global__ __launch_bounds__(64) void k_calib_strip(const double* __restrict__ p, double* out, int R, int C, int c0, int step, int back,
                                                    int strips, int rows_per_chunk) {
  const int wave = blockIdx.x, lane = threadIdx.x;
  const int strip = wave % strips, chunk = wave / strips;
  const int r0 = chunk * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  int c = c0 + strip * step - back + lane;
  c = c < 1 ? 1 : (c > C - 2 ? C - 2 : c);
  double acc = 0;
  for (int r = r0; r < r1; ++r) {
    const double* row = p + (long)r * C + c;
    if (NSH == 1) acc += row[0];
    else acc += row[-1] + row[0] + row[1];
  }
  if (acc == 1.2345e300) out[0] = acc;
}

int main() {
  const int R = 8192, C = 2048, planes = 24;   // 24 planes of 128 MiB: every kernel below gets fresh ones
  const long plane = (long)R * C;
  double *buf, *out;
  CK(hipMalloc(&buf, plane * planes * sizeof(double)));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 0, plane * planes * sizeof(double)));
  CK(hipDeviceSynchronize());
  int pl = 0;
  auto fresh = [&]() { return buf + plane * (pl++ % planes); };
  std::printf("plane = %ld bytes\n", plane * 8);
  hipLaunchKernelGGL(k_calib_aligned8, dim3(2048), dim3(256), 0, 0, fresh(), out, plane);
  std::printf("k_calib_aligned8 : asked %ld B\n", plane * 8);
  hipLaunchKernelGGL(k_calib_aligned16, dim3(2048), dim3(256), 0, 0, (const double2*)fresh(), out, plane / 2);
  std::printf("k_calib_aligned16: asked %ld B\n", plane * 8);
  // strip shapes: (step, back): 56 / 4 = the strip2 / strip3 window (starts 32 B off a sector), 48 / 8 = sector-aligned
  // window, 64 / 0 = disjoint aligned windows
  const int shapes[3][2] = {{56, 4}, {48, 8}, {64, 0}};
  for (int k = 0; k < 3; ++k)
    for (int nsh = 1; nsh <= 3; nsh += 2) {
      const int step = shapes[k][0], back = shapes[k][1], c0 = 32, c1 = C - 32;
      const int strips = (c1 - c0 + step - 1) / step, rpc = 256, chunks = R / rpc;
      const double* p = fresh();
      if (nsh == 1) hipLaunchKernelGGL(k_calib_strip<1>, dim3(strips * chunks), dim3(64), 0, 0, p, out, R, C, c0, step, back, strips, rpc);
      else hipLaunchKernelGGL(k_calib_strip<3>, dim3(strips * chunks), dim3(64), 0, 0, p, out, R, C, c0, step, back, strips, rpc);
      // host model, per row: bytes asked, sectors / lines touched per load (no sharing) and unique over the row
      long asked = 0, sec_each = 0, line_each = 0;
      std::set<long> sec_u, line_u;
      for (int s = 0; s < strips; ++s)
        for (int sh = (nsh == 1 ? 0 : -1); sh <= (nsh == 1 ? 0 : 1); ++sh) {
          std::set<long> sec, line;
          for (int lane = 0; lane < 64; ++lane) {
            int c = c0 + s * step - back + lane;
            c = c < 1 ? 1 : (c > C - 2 ? C - 2 : c);
            const long b = (long)(c + sh) * 8;
            sec.insert(b / 64), line.insert(b / 128), sec_u.insert(b / 64), line_u.insert(b / 128);
          }
          asked += 512, sec_each += 64 * (long)sec.size(), line_each += 128 * (long)line.size();
        }
      std::printf("k_calib_strip<%d> step %d back %d: per plane asked %ld B; sectors per load %ld B, lines per load %ld B; unique sectors %ld B, unique lines %ld B\n",
                  nsh, step, back, asked * R, sec_each * R, line_each * R, 64 * (long)sec_u.size() * R, 128 * (long)line_u.size() * R);
    }
  CK(hipDeviceSynchronize());
  std::printf("done\n");
  return 0;
}
