// sell_prototype.hip -- feasibility check for a row-per-lane (sliced-ELL) value-indexed SpMV.
// Synthetic stand-in for the N = 74^3 Taylor-Hood A: NB row blocks of 256 rows (4 slices of 64
// rows, one wave each), every row LEN entries, an x window of W doubles per block staged in
// LDS, a 256-entry value dictionary per block, columns as 16-bit LDS byte offsets, value codes
// 8 bit.  Stream layout: per slice, per group g of 4 entries, per lane: one dword of codes and
// one dwordx2 of columns ([slice][g][lane]) -- coalesced, aligned, 3 B per entry.
// Lane r sums ITS row: acc[s] += v * x for entry 4 g + s (4 independent chains), then
// (acc0 + acc1) + (acc2 + acc3).  No cross-lane reduction, no per-row descriptors.
//   hipcc --offload-arch=gfx950 -O3 -o sell_prototype sell_prototype.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kDict = 512;   // LDS doubles in front of the window

template <int XMODE>
__global__ __launch_bounds__(256) void sell_vi_kernel(int ngroups, int W, const uint32_t *__restrict__ codes,
                                                      const uint2 *__restrict__ cols, const double *__restrict__ dict,
                                                      const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  for (int t = threadIdx.x; t < 256; t += 256) lds[t] = dict[b * 256 + t];
  const double *xb = x + b * 768;   // window start of this block (overlapping windows)
  for (int t = threadIdx.x; t < W; t += 256) lds[kDict + t] = xb[t];
  __syncthreads();
  const char *l8 = reinterpret_cast<const char *>(lds);
  const int64_t slice = b * 4 + wave;
  const uint32_t *cp = codes + slice * (int64_t)ngroups * 64 + lane;
  const uint2 *xp = cols + slice * (int64_t)ngroups * 64 + lane;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  uint32_t cw = cp[0];
  uint2 xw = xp[0];
  for (int g = 0; g < ngroups; ++g) {
    const uint32_t c = cw;
    const uint2 xo = xw;
    if (g + 1 < ngroups) {
      cw = cp[(int64_t)(g + 1) * 64];
      xw = xp[(int64_t)(g + 1) * 64];
    }
    const double v0 = *reinterpret_cast<const double *>(l8 + ((c & 0xffu) << 3));
    const double v1 = *reinterpret_cast<const double *>(l8 + ((c >> 5) & 0x7f8u));
    const double v2 = *reinterpret_cast<const double *>(l8 + ((c >> 13) & 0x7f8u));
    const double v3 = *reinterpret_cast<const double *>(l8 + ((c >> 21) & 0x7f8u));
    const double x0 = *reinterpret_cast<const double *>(l8 + (xo.x & 0xffffu));
    const double x1 = *reinterpret_cast<const double *>(l8 + (xo.x >> 16));
    const double x2 = *reinterpret_cast<const double *>(l8 + (xo.y & 0xffffu));
    const double x3 = *reinterpret_cast<const double *>(l8 + (xo.y >> 16));
    a0 = fma(v0, x0, a0);
    a1 = fma(v1, x1, a1);
    a2 = fma(v2, x2, a2);
    a3 = fma(v3, x3, a3);
  }
  y[slice * 64 + lane] = (a0 + a1) + (a2 + a3);
}

int main(int argc, char **argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 38800, LEN = argc > 2 ? atoi(argv[2]) : 180, W = argc > 3 ? atoi(argv[3]) : 3000;
  const int ngroups = (LEN + 3) / 4;
  const int64_t nslices = (int64_t)NB * 4, nwords = nslices * ngroups * 64;
  std::vector<uint32_t> codes(nwords);
  std::vector<uint2> cols(nwords);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
  for (int64_t sl = 0; sl < nslices; ++sl)
    for (int g = 0; g < ngroups; ++g)
      for (int l = 0; l < 64; ++l) {
        const int64_t i = (sl * ngroups + g) * 64 + l;
        // rows of a slice at the same stencil position share value codes (FE structure); columns are a
        // row-dependent base plus a stencil offset, as in a node-major FE numbering
        uint32_t c = 0;
        uint32_t xo[4];
        for (int e = 0; e < 4; ++e) {
          c |= ((uint32_t)((g * 4 + e) * 37 % 200 + (l % 3)) & 0xff) << (8 * e);
          const int col = (l * 3 + (g * 4 + e) * 13 + (int)(rnd() % 3)) % W;
          xo[e] = kDict * 8 + col * 8;
        }
        codes[i] = c;
        cols[i] = make_uint2(xo[0] | (xo[1] << 16), xo[2] | (xo[3] << 16));
      }
  uint32_t *dc;
  uint2 *dx;
  double *dd, *x, *y;
  hipMalloc(&dc, nwords * 4);
  hipMalloc(&dx, nwords * 8);
  hipMalloc(&dd, (size_t)NB * 256 * 8);
  hipMalloc(&x, ((size_t)NB * 768 + W + 64) * 8);
  hipMalloc(&y, nslices * 64 * 8);
  hipMemcpy(dc, codes.data(), nwords * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, cols.data(), nwords * 8, hipMemcpyHostToDevice);
  hipMemset(dd, 0, (size_t)NB * 256 * 8);
  hipMemset(x, 0, ((size_t)NB * 768 + W + 64) * 8);
  const size_t ldsb = (size_t)(kDict + W) * 8;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL((sell_vi_kernel<0>), dim3(NB), dim3(256), ldsb, 0, ngroups, W, dc, dx, dd, x, y);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double nnz = (double)nslices * 64 * ngroups * 4;
    std::printf("NB=%d LEN=%d W=%d: %.3f ms per launch, %.2f G entries, stream %.2f GB -> %.0f GB/s stream, %.2f T entries/s\n", NB,
                LEN, W, ms / 10, nnz / 1e9, nnz * 3 / 1e9, nnz * 3 / (ms / 10) / 1e6, nnz / (ms / 10) / 1e9);
  }
  return 0;
}
