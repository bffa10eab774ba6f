// vmem_issue_rate.hip -- how fast can one CU issue coalesced global loads of each width?
// Round-2 question behind the A-SpMV: the value-indexed kernel issues
// one 1-byte and one 2-byte load per lane and 64-entry chunk.  If the texture addresser needs a
// fixed number of cycles per wave-level load instruction regardless of its width, that
// instruction count -- not HBM -- bounds the kernel, and packing 4 codes / 4 columns per lane
// into one dword / dwordx2 load is the fix.  Every variant streams the same 1 GiB once.
//   build: hipcc --offload-arch=gfx950 -O3 -o vmem_issue_rate vmem_issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

// window = 0: stream nbytes once from HBM.  window > 0 (power of two): the same number of loads, all
// falling into the first `window` bytes (per-workgroup slices), i.e. served by L1 / L2: what is left
// is the cost of issuing the load itself (address processing + cache return path).
template <class T, int U, int STRIDE_BYTES>
__global__ __launch_bounds__(256) void stream_kernel(const uint8_t *__restrict__ p, size_t nbytes, uint64_t *out,
                                                     size_t window) {
  // each lane reads sizeof(T) bytes at byte offset STRIDE_BYTES * (global lane index)
  const size_t lanes_total = (size_t)gridDim.x * blockDim.x;
  const size_t n = nbytes / STRIDE_BYTES - 8;
  const size_t wmask = window ? window / STRIDE_BYTES - 1 : ~(size_t)0;
  uint64_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * lanes_total < n; i += U * lanes_total) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint8_t *q = p + ((i + u * lanes_total) & wmask) * STRIDE_BYTES;
      __builtin_memcpy(&v[u], q, sizeof(T));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      uint64_t w[2] = {0, 0};
      __builtin_memcpy(w, &v[u], sizeof(T));
      acc += w[0] ^ w[1];
    }
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;
}

struct V3 { uint8_t b[3]; };
struct V6 { uint16_t h[3]; };
struct alignas(4) A4 { uint32_t w; };
struct alignas(8) A8 { uint32_t w[2]; };
struct alignas(16) A16 { uint32_t w[4]; };
struct __attribute__((packed)) P4 { uint32_t w; };     // 4-byte load at any byte address
struct __attribute__((packed, aligned(2))) P8 { uint32_t w[2]; };  // 8-byte load at 2-byte alignment

template <class T, int U, int S>
static void run(const char *name, const uint8_t *d, size_t nbytes, uint64_t *out, double clk_ghz, int ncu,
                size_t window = 0) {
  const int grid = ncu * 8;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((stream_kernel<T, U, S>), dim3(grid), dim3(256), 0, 0, d, nbytes, out, window);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double instr = (double)nbytes / S / 64.0;  // wave-level load instructions
  const double cyc = ms * 1e-3 * clk_ghz * 1e9;
  std::printf("%-34s %s U=%d  %8.3f ms  %8.1f GB/s useful  %6.2f cycles per wave-load per CU\n", name, window ? "L2-resident" : "HBM        ", U, ms,
              (double)nbytes / S * sizeof(T) / ms / 1e6, cyc / (instr / ncu));
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int ncu = pr.multiProcessorCount;
  const double ghz = pr.clockRate * 1e-6;
  std::printf("%s: %d CUs, %.2f GHz\n", pr.name, ncu, ghz);
  const size_t nbytes = (size_t)1 << 30;
  uint8_t *d;
  uint64_t *out;
  hipMalloc(&d, nbytes + 4096);
  hipMalloc(&out, 8);
  hipMemset(d, 1, nbytes + 4096);
  run<uint8_t, 8, 1>("u8  per lane (stride 1)", d, nbytes, out, ghz, ncu);
  run<uint8_t, 16, 1>("u8  per lane (stride 1)", d, nbytes, out, ghz, ncu);
  run<uint16_t, 8, 2>("u16 per lane (stride 2)", d, nbytes, out, ghz, ncu);
  run<A4, 8, 4>("dword per lane (stride 4)", d, nbytes, out, ghz, ncu);
  run<A8, 8, 8>("dwordx2 per lane (stride 8)", d, nbytes, out, ghz, ncu);
  run<A16, 4, 16>("dwordx4 per lane (stride 16)", d, nbytes, out, ghz, ncu);
  run<A16, 8, 16>("dwordx4 per lane (stride 16)", d, nbytes, out, ghz, ncu);
  run<P4, 8, 3>("unaligned dword (stride 3)", d, nbytes, out, ghz, ncu);
  run<P4, 8, 2>("overlapping dword (stride 2)", d, nbytes, out, ghz, ncu);
  run<P8, 8, 6>("2-aligned dwordx2 (stride 6)", d, nbytes, out, ghz, ncu);
  run<V3, 8, 3>("3 x u8 (stride 3)", d, nbytes, out, ghz, ncu);
  run<V6, 8, 6>("3 x u16 (stride 6)", d, nbytes, out, ghz, ncu);
  // the same load counts out of a 3 MiB (stride-dependent, < one XCD's 4 MiB L2) window
  const size_t w = (size_t)1 << 21;
  run<uint8_t, 8, 1>("u8  per lane (stride 1)", d, nbytes, out, ghz, ncu, w);
  run<uint16_t, 8, 2>("u16 per lane (stride 2)", d, nbytes, out, ghz, ncu, w);
  run<A4, 8, 4>("dword per lane (stride 4)", d, nbytes, out, ghz, ncu, w);
  run<A8, 8, 8>("dwordx2 per lane (stride 8)", d, nbytes, out, ghz, ncu, w);
  run<A16, 8, 16>("dwordx4 per lane (stride 16)", d, nbytes, out, ghz, ncu, w);
  run<P4, 8, 2>("overlapping dword (stride 2)", d, nbytes, out, ghz, ncu, w);
  run<P8, 8, 8>("8-aligned packed dwordx2", d, nbytes, out, ghz, ncu, w);
  return 0;
}
