// FETCH_SIZE calibration for narrow per-lane loads (run under rocprofv3 --pmc FETCH_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class T>
__global__ void read_stream(const T *__restrict__ p, int64_t n, unsigned long long *out) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    acc += (unsigned long long)p[i];
  if (acc == 0x123456789abcdefull) out[0] = acc;
}
// the SpMV kernel's pattern: per wave, 8 independent loads of consecutive 64-element chunks
template <class T>
__global__ void read_chunks(const T *__restrict__ p, int64_t n, unsigned long long *out) {
  unsigned long long acc = 0;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = wave * 8; c * 64 < n; c += nw * 8) {
    T v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c + j) * 64 + lane < n ? p[(c + j) * 64 + lane] : (T)0;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += (unsigned long long)v[j];
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;
}
int main() {
  const int64_t bytes = 1ll << 30;
  void *d; unsigned long long *o;
  hipMalloc(&d, bytes); hipMalloc((void **)&o, 8);
  hipMemset(d, 1, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_stream<uint8_t>, dim3(256 * 8), dim3(256), 0, 0, (const uint8_t *)d, bytes, o);
    hipLaunchKernelGGL(read_stream<uint16_t>, dim3(256 * 8), dim3(256), 0, 0, (const uint16_t *)d, bytes / 2, o);
    hipLaunchKernelGGL(read_stream<uint32_t>, dim3(256 * 8), dim3(256), 0, 0, (const uint32_t *)d, bytes / 4, o);
    hipLaunchKernelGGL(read_stream<unsigned long long>, dim3(256 * 8), dim3(256), 0, 0, (const unsigned long long *)d, bytes / 8, o);
    hipLaunchKernelGGL(read_chunks<uint8_t>, dim3(256 * 8), dim3(256), 0, 0, (const uint8_t *)d, bytes, o);
    hipLaunchKernelGGL(read_chunks<uint16_t>, dim3(256 * 8), dim3(256), 0, 0, (const uint16_t *)d, bytes / 2, o);
  }
  hipDeviceSynchronize();
  printf("done, %lld bytes per kernel\n", (long long)bytes);
  return 0;
}
