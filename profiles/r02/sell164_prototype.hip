// sell164_prototype.hip -- second feasibility check: "SELL-16-4" value-indexed SpMV with the block
// structure the real N = 74^3 operator would have.  Row blocks of RB = 128 rows = 8 slices of 16 rows;
// 4 lanes per row (canonical order with L = 4: lane l sums entries l, l+4, ... then (l0+l2)+(l1+l3));
// per slice and group of 16 entries per row one dword of 4 value codes + one dwordx2 of 4 LDS
// column offsets per lane; 4 waves per workgroup, wave w takes slices w and 7-w (long + short);
// x window of W doubles staged from 25 segments of the x vector (5 x 5 neighbouring grid lines);
// block dictionary of 256 doubles.  Slice lengths alternate 375 / 225 entries per row.
//   hipcc --offload-arch=gfx950 -O3 -o sell164_prototype sell164_prototype.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kDict = 512;

__device__ __forceinline__ double quad_tree(double v) {   // (l0 + l2) + (l1 + l3) in every quad
  int lo = __double2loint(v), hi = __double2hiint(v);
  double o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false),
                              __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v = v + o;
  lo = __double2loint(v), hi = __double2hiint(v);
  o = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false),
                       __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false));          // quad_perm [1,0,3,2]
  return v + o;
}

__global__ __launch_bounds__(256) void sell164_kernel(int W, int seglen, int64_t line_stride, const int64_t *__restrict__ sl_off,
                                                      const int32_t *__restrict__ sl_ng, const uint32_t *__restrict__ codes,
                                                      const uint2 *__restrict__ cols, const double *__restrict__ dict,
                                                      const double *__restrict__ x, double *__restrict__ y, int xcd) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t b = blockIdx.x;
  if (xcd) {
    const int64_t nwg = gridDim.x, q = nwg / 8, rm = nwg % 8, xc = b % 8, idx = b / 8;
    b = (xc < rm ? xc * (q + 1) : rm * (q + 1) + (xc - rm) * q) + idx;
  }
  lds[threadIdx.x] = dict[b * 256 + threadIdx.x];
  // window: 25 segments of seglen doubles, segment s starts at x + b * 128 + s * line_stride (5 x 5 lines)
  for (int t = threadIdx.x; t < W; t += 256) {
    const int s = t / seglen, o = t - s * seglen;
    lds[kDict + t] = x[b * 128 + (int64_t)s * line_stride + o];
  }
  if (threadIdx.x == 0) lds[kDict + W] = 0.0;
  __syncthreads();
  const char *l8 = reinterpret_cast<const char *>(lds);
  for (int pass = 0; pass < 2; ++pass) {
    const int64_t slice = b * 8 + (pass == 0 ? wave : 7 - wave);
    const int ng = sl_ng[slice];
    const uint32_t *cp = codes + sl_off[slice] * 64 + lane;
    const uint2 *xp = cols + sl_off[slice] * 64 + lane;
    double acc = 0.0;
    uint32_t cw = cp[0];
    uint2 xw = xp[0];
    for (int g = 0; g < ng; ++g) {
      const uint32_t c = cw;
      const uint2 xo = xw;
      if (g + 1 < ng) {
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
      acc = fma(v0, x0, acc);
      acc = fma(v1, x1, acc);
      acc = fma(v2, x2, acc);
      acc = fma(v3, x3, acc);
    }
    acc = quad_tree(acc);
    if ((lane & 3) == 0) y[slice * 16 + (lane >> 2)] = acc;
  }
}

int main(int argc, char **argv) {
  const int NB = argc > 1 ? atoi(argv[1]) : 77500;          // 9.92 M rows / 128
  const int W = argc > 2 ? atoi(argv[2]) : 3500;
  const int xcd = argc > 3 ? atoi(argv[3]) : 0;
  const int seglen = W / 25;
  const int64_t nslices = (int64_t)NB * 8;
  std::vector<int64_t> off(nslices + 1, 0);
  std::vector<int32_t> ng(nslices);
  for (int64_t s = 0; s < nslices; ++s) {
    const int len = (s % 8) < 4 ? 375 : 225;      // 4 long + 4 short slices per block
    ng[s] = (len + 15) / 16;
    off[s + 1] = off[s] + ng[s];
  }
  const int64_t nwords = off[nslices] * 64;
  std::vector<uint32_t> codes(nwords);
  std::vector<uint2> cols(nwords);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 11); };
  for (int64_t s = 0; s < nslices; ++s)
    for (int g = 0; g < ng[s]; ++g)
      for (int l = 0; l < 64; ++l) {
        const int64_t i = (off[s] + g) * 64 + l;
        const int row = l >> 2, q = l & 3;
        uint32_t c = 0, xo[4];
        for (int e = 0; e < 4; ++e) {
          const int k = g * 16 + q + 4 * e;                        // entry index within the row
          c |= ((uint32_t)(k * 37 % 200 + row % 3) & 0xff) << (8 * e);
          // entry k sits in grid line k / 15 (25 lines of 15 consecutive columns around the row's node)
          const int seg = (k / 15) % 25, within = (row * 2 + k % 15) % seglen;
          xo[e] = kDict * 8 + (seg * seglen + within) * 8;
        }
        codes[i] = c;
        cols[i] = make_uint2(xo[0] | (xo[1] << 16), xo[2] | (xo[3] << 16));
      }
  const int64_t line_stride = 447;     // 149 nodes x 3 components per grid line at N = 74
  const size_t nx = (size_t)NB * 128 + 25 * line_stride + W + 1024;
  uint32_t *dc; uint2 *dx; double *dd, *x, *y; int64_t *doff; int32_t *dng;
  hipMalloc(&dc, nwords * 4); hipMalloc(&dx, nwords * 8); hipMalloc(&dd, (size_t)NB * 256 * 8);
  hipMalloc(&x, nx * 8); hipMalloc(&y, nslices * 16 * 8); hipMalloc(&doff, (nslices + 1) * 8); hipMalloc(&dng, nslices * 4);
  hipMemcpy(dc, codes.data(), nwords * 4, hipMemcpyHostToDevice);
  hipMemcpy(dx, cols.data(), nwords * 8, hipMemcpyHostToDevice);
  hipMemcpy(doff, off.data(), (nslices + 1) * 8, hipMemcpyHostToDevice);
  hipMemcpy(dng, ng.data(), nslices * 4, hipMemcpyHostToDevice);
  hipMemset(dd, 0, (size_t)NB * 256 * 8);
  hipMemset(x, 0, nx * 8);
  const size_t ldsb = (size_t)(kDict + W + 1) * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL(sell164_kernel, dim3(NB), dim3(256), ldsb, 0, W, seglen, line_stride, doff, dng, dc, dx, dd, x, y, xcd);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double slots = (double)nwords * 4, nnz = (double)NB * 64 * (375 + 225);
    std::printf("NB=%d W=%d xcd=%d: %.3f ms per launch; %.2f G slots for %.2f G entries; stream %.2f GB + windows %.2f GB\n", NB, W, xcd,
                ms / 10, slots / 1e9, nnz / 1e9, slots * 3 / 1e9, (double)NB * W * 8 / 1e9);
  }
  return 0;
}
