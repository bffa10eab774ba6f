// Batch-major value-indexed SpMV ("vs" format) for gfx950 -- the kernel the A-SpMV runs on.
//
// Same arithmetic as spmv_window_vib_kernel (canonical lane assignment, fma order and 64-lane tree of
// ALFD-arith v1 -- bit-identical results); what differs is the storage (planned on the host by plan_vs
// in alfd.hip, decoded back by alfd_host_stream_plan for the CPU tests):
//
//  * a row block is a LIST of rows, not a run of the numbering.  With blocks that are bricks of the mesh
//    (alfd_set_row_blocks / alfd_host_row_blocks_from_points) the x window a block stages in LDS is
//    800..1 200 slots instead of 2 700 for 96 consecutive rows of a lexicographic numbering: a third of
//    the L2 -> LDS staging and of the HBM re-fetch of x, 8 resident waves per SIMD instead of 4.9;
//  * an entry is a 24-bit field, (9-bit dictionary code << 15) | (12-bit window column << 3): both LDS
//    byte offsets ready-shifted.  Dictionaries hold up to 512 values per block (blocks beyond are halved);
//  * the stream is BATCH-MAJOR and LANE-MAJOR: the rows of a block are grouped by chunk count
//    ceil(len / 64) into batches; for chunk j, lane l finds what it needs at one address:
//      - plain batch (4 rows): a 12-byte cell with the fields of the four rows at 768 j + 12 l -- one
//        global_load_dwordx3 instead of eight 1- and 2-byte loads (a wave-level load costs ~8.5 issue
//        cycles whether it carries 1 or 4 bytes per lane; with brick windows the stream loads were what
//        bounded the kernel: ablation without them 0.61 ms, without the LDS gathers 1.29 of 1.30 ms);
//      - shared batch (2..16 rows that are translates of one another: same length, same values entry by
//        entry, window columns differing by one constant per row -- the rows of one node type inside a
//        brick of a uniform mesh): ONE stored row, a dword at 256 j + 4 l, plus a window shift per row;
//        one dictionary gather serves all rows.  96 % of the entries of the Stokes velocity block;
//    of the last, partial chunk only the lanes below the batch's longest remainder are stored;
//  * the 128-byte batch descriptor names the GLOBAL row of each row, so the sums go straight to y.
//  * per block an 8-dword header {batches, window slots, window pieces, dictionary size, dictionary offset, stream offset
//    lo / hi, 0} and a segment table with a fixed stride per block ((column, slot) pairs): both addressable from the block
//    index alone, so the x window is requested after one dependent round trip; a launch covers the blocks
//    block_base .. block_base + gridDim.x - 1 (partitioned contexts launch the blocks without halo columns first and the
//    rest when the halo has arrived).
//
// At N = 74: 0.445 ms per launch (round 2: 0.52 ms with 8-row batches; round-1 kernel: 6.08 GB, 1.5 ms).  The time is
// t = 2.6 ms / (waves per SIMD) + 0.19 ms (measured by padding the LDS request): what counts is the serial length of
// one wave's instruction stream per row -- hence 16 rows per batch, descriptors that never touch the scalar memory
// path, row numbers by ds_bpermute.  Experiments and history: DESIGN.md section 5, profiles/r03/, profiles/r02/.
#pragma once

namespace alfd {

constexpr int kVsMaxDict = 512;   // distinct values per block (9-bit codes)
constexpr int kVsDictOff = 0;     // the block's dictionary at LDS offset 0
constexpr int kVsWinOff = kVsMaxDict * 8;   // the x window behind it (at most 4096 slots: 12-bit columns)
constexpr int kVsMaxRows = 250;   // rows per block
// Field layout of an entry.  WD = 0: 9-bit dictionary code, 12-bit window column (512 values, 4096 slots per block);
// WD = 1 ("wide codes"): 10-bit code, 11-bit column (1024 values, 2048 slots) -- for operators whose blocks hold more
// than 512 distinct values, e.g. cell-wise assembled matrices whose mathematically equal entries differ in their last
// bits with the visiting order of the cells (720 instead of 285 distinct values in the Stokes velocity block).
template <int WD>
struct VsFmt {
  static constexpr uint32_t kColMask = WD ? 0x3ff8u : 0x7ff8u;
  static constexpr int kDictShift = WD ? 11 : 12;
  static constexpr uint32_t kDictMask = WD ? 0x1ff8u : 0xff8u;
  static constexpr int kWinOff = WD ? 8192 : 4096;
  static constexpr int kCodeShift = WD ? 14 : 15;    // host side: field = (code << kCodeShift) | (slot << 3)
  static constexpr int kMaxDict = WD ? 1024 : 512;
  static constexpr int kMaxSlots = WD ? 2048 : 4096;
};
constexpr int kVsMaxLen = 384;    // longest row the format takes (class 6)

// batch descriptor: 16 x uint64 (128 bytes), one per row: eb (20 bits) | entry count (9) | class = ceil(count / 64)
// (3) | global row, 0xffffffff = filler (32).  eb: the batch's offset from the block's first byte in 16-byte
// units.  Plain batches use rows 0..3 (fillers repeat row 0).  Bit 63 of row 0 marks a SHARED batch
// (vs_shared) of up to 16 rows; there the low dword of rows 1.. is the row's window shift in bytes (signed).
__device__ __forceinline__ uint32_t vs_off(uint32_t d) { return d & 0xfffffu; }
__device__ __forceinline__ int32_t vs_len(uint32_t d) { return (int32_t)((d >> 20) & 0x1ffu); }
__device__ __forceinline__ int vs_cls(uint32_t d) { return (int)((d >> 29) & 7u); }

// LDS read at a byte offset from the start of the workgroup's LDS (the kernel has no static
// __shared__ data, so its dynamic array starts at 0): no symbol, so nothing is added to the offset
__device__ __forceinline__ double vs_lds_f64(uint32_t byte_off) {
  return *(const __attribute__((address_space(3))) double *)(uintptr_t)byte_off;
}

// The kernels of this file read LDS at ABSOLUTE byte offsets (vs_lds_f64), which is right only while their dynamic
// __shared__ array starts at LDS address 0, i.e. while neither they nor anything they call owns static LDS.  This
// one-workgroup probe has the same property (dynamic LDS only) and reports where its array starts; the library runs it
// once per context and refuses the batch-major formats if the answer is not 0.
__global__ void vs_lds_base_probe_kernel(uint32_t *out) {
  extern __shared__ double xs[];
  if (threadIdx.x == 0) out[0] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(char *)xs;
}

struct VsWord3 {
  uint32_t x, y, z;
};

// One batch of 4 rows with NCH chunks each, stored lane-major with the four rows interleaved: for
// chunk j, lane l holds at fb + 768 j + 12 l four 24-bit fields, one per row:
//     f = (value code << 15) | (window column << 3)          (9-bit code, 12-bit column)
// i.e. both LDS byte offsets ready-shifted: column offset = f & 0x7ff8, dictionary offset =
// (f >> 12) & 0xff8.  ONE global_load_dwordx3 per chunk serves all four rows.  The last chunk is
// partial: only lanes below the batch's longest remainder are stored (rows of a class are sorted by
// length, so the remainders of a batch are equal or close); rem[i] masks row i's fma.
template <int NCH, int WD>
__device__ __forceinline__ void vs_batch(const int32_t (&rem)[4], int32_t maxrem, int lane,
                                         const uint8_t *__restrict__ fb, const char *sm, double (&acc)[4]) {
  VsWord3 w[NCH];
#pragma unroll
  for (int j = 0; j < NCH - 1; ++j) w[j] = *(const VsWord3 *)(fb + 768 * j + 12u * (uint32_t)lane);
  w[NCH - 1] = *(const VsWord3 *)(fb + 768 * (NCH - 1) + 12u * (uint32_t)(lane < maxrem ? lane : maxrem - 1));
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    double xv[4], v[4];
    const uint32_t f[4] = {w[j].x, __builtin_amdgcn_alignbit(w[j].y, w[j].x, 24),
                           __builtin_amdgcn_alignbit(w[j].z, w[j].y, 16), w[j].z >> 8};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[i] = vs_lds_f64(VsFmt<WD>::kWinOff + (f[i] & VsFmt<WD>::kColMask));
      v[i] = vs_lds_f64(kVsDictOff + ((f[i] >> VsFmt<WD>::kDictShift) & VsFmt<WD>::kDictMask));
    }
    // keep the eight gathers of the chunk ahead of the (masked) fmas
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(xv[i]), "+v"(v[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (j == NCH - 1) {
        if (lane < rem[i]) acc[i] = fma(v[i], xv[i], acc[i]);
      } else {
        acc[i] = fma(v[i], xv[i], acc[i]);
      }
    }
  }
}

// A SHARED batch: its R = 4 or 8 rows are translates of one another (same length, same values entry by
// entry, window columns differing by one constant per row) -- the rows of one node type inside a mesh
// brick.  One row is stored (a dword per lane and chunk, lane-major), the others add their window shift
// sh[i] (bytes): 1/R of the stream, one dictionary gather instead of R.
// Its stream words are fetched ONE BATCH AHEAD (vs_shared_fetch, issued before the tree and the store of the batch
// in front, or before the block frame for a wave's first batch): a wave spent most of a batch's ~7 000 cycles waiting
// for exactly this round trip.  w[0] = the last, partial chunk; w[1 + j] = full chunk j (static register indices for
// every chunk count).
__device__ __forceinline__ void vs_shared_fetch(uint32_t d0, const uint8_t *__restrict__ sbase, int lane, uint32_t (&w)[6]) {
  const int cls = vs_cls(d0);
  const uint8_t *fb = sbase + 16u * (size_t)vs_off(d0);
  const int32_t rem = vs_len(d0) - 64 * (cls - 1);
  // unsigned 32-bit lane offsets made opaque here: a scalar base + one VGPR per load (global_load ... saddr), instead of
  // loop-invariant 64-bit per-lane addresses that the register file of the 16-row path has no room for
  uint32_t l4 = 4u * (uint32_t)lane, r4 = 4u * (uint32_t)(lane < rem ? lane : rem - 1);
  asm volatile("" : "+v"(l4), "+v"(r4));
  w[0] = *(const uint32_t *)(fb + 256 * (cls - 1) + r4);
#pragma unroll
  for (int j = 0; j < 5; ++j)
    if (j < cls - 1) w[1 + j] = *(const uint32_t *)(fb + 256 * j + l4);
}

template <int NCH, int R, int WD>
__device__ __forceinline__ void vs_shared(int32_t rem, const int32_t (&sh)[R], int lane, const uint32_t (&w)[6],
                                          double (&acc)[R]) {
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const uint32_t wj = j == NCH - 1 ? w[0] : w[1 + (j < 5 ? j : 0)];
    const uint32_t lc = wj & VsFmt<WD>::kColMask;
    double v = vs_lds_f64(kVsDictOff + ((wj >> VsFmt<WD>::kDictShift) & VsFmt<WD>::kDictMask));
    constexpr int G = R > 8 ? 4 : R;   // gathers in flight (a 16-row batch takes four rounds: with 8 the kernel spills, 0.64 instead of 0.44 ms)
#pragma unroll
    for (int g = 0; g < R; g += G) {
      double xv[G];
#pragma unroll
      for (int i = 0; i < G; ++i) xv[i] = vs_lds_f64(VsFmt<WD>::kWinOff + (uint32_t)((int32_t)lc + sh[g + i]));
#pragma unroll
      for (int i = 0; i < G; ++i) asm volatile("" : "+v"(xv[i]));
      if (g == 0) asm volatile("" : "+v"(v));
      if (j == NCH - 1) {
        if (lane < rem) {   // one exec mask around the fmas (the empty asm keeps it a branch, not 2 G selects)
          asm volatile("");
#pragma unroll
          for (int i = 0; i < G; ++i) acc[g + i] = fma(v, xv[i], acc[g + i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < G; ++i) acc[g + i] = fma(v, xv[i], acc[g + i]);
      }
      if (R > 8) {   // round by round: G gathers, G fmas (left alone the fmas sink below ALL gathers of the batch: spills)
#pragma unroll
        for (int i = 0; i < G; ++i) asm volatile("" : "+v"(acc[g + i]));
      }
    }
  }
}

template <int EPI, int TAG, int NW, int WD = 0>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmv_vs_kernel(
    const uint8_t *__restrict__ stream, const uint64_t *__restrict__ tab, int32_t stride,
    const int32_t *__restrict__ hdrb, const int32_t *__restrict__ segx, int32_t seg_stride,
    const double *__restrict__ dict, const double *__restrict__ x, const double *__restrict__ x_halo,
    int32_t n_local, double *__restrict__ y, double alpha, const double *__restrict__ d, double *__restrict__ y2,
    int xcd_remap, int32_t block_base) {
  extern __shared__ double xs[];
  char *sm = (char *)xs;
  double *ds = (double *)(sm + kVsDictOff);
  double *xw = (double *)(sm + VsFmt<WD>::kWinOff);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t b = blockIdx.x;
  if (xcd_remap) {  // workgroups with equal blockIdx % 8 share an XCD: give each XCD a contiguous run of blocks
    const int64_t nwg = gridDim.x, q = nwg / 8, rm = nwg % 8, xcd = b % 8, idx = b / 8;
    b = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + idx;
  }
  b += block_base;   // a launch covers the blocks block_base .. block_base + gridDim.x - 1 (interior / boundary halves)
  // Everything the block needs first hangs off its index: the 8-dword header, the first pieces of its segment table
  // (fixed stride per block) and the wave's first two batch descriptors are requested together; the x window, the
  // dictionary and the first stream words follow after that ONE round trip.
  const int32_t *hb = hdrb + 8 * b;
  const int32_t nbatch = hb[0], W = hb[1], nseg = hb[2], nd = hb[3], doff = hb[4];
  const uint8_t *sbase = stream + (int64_t)(((uint64_t)(uint32_t)hb[6] << 32) | (uint64_t)(uint32_t)hb[5]);
  const int32_t *sx = segx + 2 * b * (int64_t)seg_stride;
  // Batch descriptors travel through the VECTOR memory path, two batches ahead: lane k of either half-wave holds dword k
  // of a descriptor, the scalars come out of it with v_readlane.  (An s_load one batch ahead, as before, sat in the same
  // lgkmcnt counter as the LDS gathers, and scalar loads return out of order: the first gather wait of every batch also
  // waited for the descriptor of the NEXT one -- a full HBM round trip per batch, the kernel ran at the speed of that.)
  const uint32_t *tb = (const uint32_t *)(tab + (int64_t)b * stride * 16);
  const int hl = lane & 31;
  auto hdr_at = [&](int32_t q) {   // batch q of the block (the table is padded: q may run past the block's last batch)
    uint32_t h4 = 4u * (uint32_t)hl;
    asm volatile("" : "+v"(h4));   // as in vs_shared_fetch: scalar base + 32-bit lane offset
    return *(const uint32_t *)((const char *)(tb + 32 * (int64_t)q) + h4);
  };
  auto hdr_load = [&](int32_t q) {   // the same, clamped to the block's last batch
    return hdr_at(q < nbatch ? q : (nbatch > 0 ? nbatch - 1 : 0));
  };
  auto hdr = [](uint32_t hv, int k) { return (uint32_t)__builtin_amdgcn_readlane((int)hv, k); };
  uint32_t hv1 = hdr_at(wave);
  uint32_t hv2 = hdr_at(wave + NW);
  uint32_t w[6];
  {  // block frame: dictionary and x window (segments of at most 64 slots, 4 in flight per wave)
    constexpr int U = 4;
    auto stage = [&](int32_t s) {   // pieces s .. s + U - 1 of the block's window; those past the last are masked
      int32_t c0[U], o0[U], sl[U];
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t q = s + u;   // wave-uniform
        c0[u] = sx[2 * q];
        o0[u] = sx[2 * q + 1];
        const int32_t on = sx[2 * q + 3];
        sl[u] = q < nseg ? ((q + 1 < nseg ? on : W) - o0[u]) : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t c = c0[u] + (lane < sl[u] ? lane : 0);
        v[u] = (c < n_local) ? x[c] : x_halo[c - n_local];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (lane < sl[u]) xw[o0[u] + lane] = v[u];
    };
    for (int t = threadIdx.x; t < nd; t += 64 * NW) ds[t] = dict[doff + t];
    if (wave < nbatch && (int32_t)hdr(hv1, 1) < 0) vs_shared_fetch(hdr(hv1, 0), sbase, lane, w);   // in flight during the frame
    stage(wave * U);   // the first round unconditionally: its table reads do not wait for the header
    for (int32_t s = (wave + NW) * U; s < nseg; s += NW * U) stage(s);
  }
  __syncthreads();
  // The lane that ends up with a row's tree value stores it.  Which row that is depends on the lane alone --
  // batch row {0, 2, 1, 3}[lane / 16] + 4 * (bit 3 of the lane) + 8 * (bit 2), see reduce_rows16 / 8 / 4 -- so every
  // lane reads the global row number of "its" row out of the descriptor register with one ds_bpermute (the rows a
  // smaller batch does not have are fillers, -1) instead of selecting it out of the descriptor's scalars.
  const int q16 = lane >> 4;
  const int rsel = 4 * (2 * ((((q16 & 1) << 1) | (q16 >> 1)) + 4 * ((lane >> 3) & 1) + 8 * ((lane >> 2) & 1)) + 1);   // ds_bpermute address of that dword
  auto store = [&](double s, int32_t r, bool mine) {
    if (mine && r >= 0) {
      if (EPI == 0)
        y[r] = s;
      else if (EPI == 1)
        y[r] = fma(alpha, s, y[r]);
      else if (EPI == 2)
        y[r] = d[r] * s;
      else {
        y[r] = s;
        y2[r] = d[r] * s;
      }
    }
  };
  for (int32_t bi = wave; bi < nbatch; bi += NW) {
    uint32_t lo[16];   // low dwords of the rows: offset | count | class (row 0, plain rows), window shifts (shared rows 1..)
#pragma unroll
    for (int i = 0; i < 16; ++i) lo[i] = hdr(hv1, 2 * i);
    const bool shared = (int32_t)hdr(hv1, 1) < 0;                       // wave-uniform
    const bool sixteen = shared && (int32_t)hdr(hv1, 17) >= 0;          // a ninth row
    const bool eight = shared && !sixteen && (int32_t)hdr(hv1, 9) >= 0;  // a fifth row
    const uint32_t eb = vs_off(lo[0]);
    const int cls = vs_cls(lo[0]);
    const int32_t full = cls > 0 ? 64 * (cls - 1) : 0;
    const uint8_t *fb = sbase + 16u * (size_t)eb;
    double sres;    // the tree value of "its" row, in the lanes whose number is a multiple of mstep
    int mstep;      // wave-uniform
    uint32_t n0, n1;   // row 0 of the NEXT batch's descriptor (offset | count | class, row | shared flag)
    if (sixteen) {                                       // 9..16 translates
      const int32_t rem = vs_len(lo[0]) - full;
      int32_t sh[16];
      sh[0] = 0;
#pragma unroll
      for (int i = 1; i < 16; ++i) sh[i] = (int32_t)lo[i];
      double acc[16] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      switch (cls) {
        case 1: vs_shared<1, 16, WD>(rem, sh, lane, w, acc); break;
        case 2: vs_shared<2, 16, WD>(rem, sh, lane, w, acc); break;
        case 3: vs_shared<3, 16, WD>(rem, sh, lane, w, acc); break;
        case 4: vs_shared<4, 16, WD>(rem, sh, lane, w, acc); break;
        case 5: vs_shared<5, 16, WD>(rem, sh, lane, w, acc); break;
        default: vs_shared<6, 16, WD>(rem, sh, lane, w, acc); break;
      }
      // lane 16 q + 8 h + 4 g holds the tree of batch row {0, 2, 1, 3}[q] + 4 h + 8 g
      sres = reduce_rows16(acc, lane);
      mstep = 4;
    } else if (eight) {                                  // 5..8 translates
      const int32_t rem = vs_len(lo[0]) - full;
      int32_t sh[8];
      sh[0] = 0;
#pragma unroll
      for (int i = 1; i < 8; ++i) sh[i] = (int32_t)lo[i];
      double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      switch (cls) {
        case 1: vs_shared<1, 8, WD>(rem, sh, lane, w, acc); break;
        case 2: vs_shared<2, 8, WD>(rem, sh, lane, w, acc); break;
        case 3: vs_shared<3, 8, WD>(rem, sh, lane, w, acc); break;
        case 4: vs_shared<4, 8, WD>(rem, sh, lane, w, acc); break;
        case 5: vs_shared<5, 8, WD>(rem, sh, lane, w, acc); break;
        default: vs_shared<6, 8, WD>(rem, sh, lane, w, acc); break;
      }
      // lane 16 q + 8 h (q = 0..3, h = 0..1) holds the tree of batch row {0, 2, 1, 3}[q] + 4 h
      sres = reduce_rows8(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], lane);
      mstep = 8;
    } else if (shared) {   // 2..4 translates
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      const int32_t rem = vs_len(lo[0]) - full;
      const int32_t sh[4] = {0, (int32_t)lo[1], (int32_t)lo[2], (int32_t)lo[3]};
      switch (cls) {
        case 1: vs_shared<1, 4, WD>(rem, sh, lane, w, acc); break;
        case 2: vs_shared<2, 4, WD>(rem, sh, lane, w, acc); break;
        case 3: vs_shared<3, 4, WD>(rem, sh, lane, w, acc); break;
        case 4: vs_shared<4, 4, WD>(rem, sh, lane, w, acc); break;
        case 5: vs_shared<5, 4, WD>(rem, sh, lane, w, acc); break;
        default: vs_shared<6, 4, WD>(rem, sh, lane, w, acc); break;
      }
      sres = reduce_rows4(acc[0], acc[1], acc[2], acc[3]);   // 16-lane row q: batch row {0, 2, 1, 3}[q]
      mstep = 16;
    } else {   // plain batch: stores at once (its registers are the kernel's tightest spot, nothing is carried further)
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      int32_t rem[4], maxrem = 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rem[i] = vs_len(lo[i]) - full;
        maxrem = rem[i] > maxrem ? rem[i] : maxrem;
      }
      switch (cls) {
        case 1: vs_batch<1, WD>(rem, maxrem, lane, fb, sm, acc); break;
        case 2: vs_batch<2, WD>(rem, maxrem, lane, fb, sm, acc); break;
        case 3: vs_batch<3, WD>(rem, maxrem, lane, fb, sm, acc); break;
        case 4: vs_batch<4, WD>(rem, maxrem, lane, fb, sm, acc); break;
        case 5: vs_batch<5, WD>(rem, maxrem, lane, fb, sm, acc); break;
        case 6: vs_batch<6, WD>(rem, maxrem, lane, fb, sm, acc); break;
        default: break;  // class 0: empty rows
      }
      // the next descriptor is read BEFORE the store: after it, the wait that makes hv2 readable would also wait for
      // the store's acknowledgement, and only then would the next batch's loads go out (two round trips in a row)
      n0 = hdr(hv2, 0);
      n1 = hdr(hv2, 1);
      store(reduce_rows4(acc[0], acc[1], acc[2], acc[3]), (int32_t)__builtin_amdgcn_ds_bpermute(rsel, (int)hv1), (lane & 15) == 0);
      mstep = 0;
    }
    // the stream words of the wave's next shared batch (the ONE place of the loop that writes w: no copies of loaded
    // values, which would have to wait for them), and behind them the descriptor of the batch after that.  The store
    // comes LAST: vector memory operations complete in order, so a wait for anything requested after a store would
    // also wait for the store's acknowledgement (as the round-2 kernel did, once per batch).
    int32_t rid = -1;
    if (mstep) {
      rid = (int32_t)((uint32_t)__builtin_amdgcn_ds_bpermute(rsel, (int)hv1) &
                      (lane < 8 ? 0x7fffffffu : 0xffffffffu));   // bit 63 of row 0 is the shared-batch flag
      n0 = hdr(hv2, 0);
      n1 = hdr(hv2, 1);
    }
    hv1 = hv2;
    if (bi + NW < nbatch && (int32_t)n1 < 0) vs_shared_fetch(n0, sbase, lane, w);
    hv2 = hdr_load(bi + 2 * NW);
    if (mstep) store(sres, rid, (lane & (mstep - 1)) == 0);
  }
}

// --------------------------------------------------------------------------
// The same storage idea for SHORT rows (canonical L = 8, 16 or 32 lanes per row).  A wave holds G = 64 / L rows
// side by side; a batch is one stored template row (a dword per entry) shared by up to 4 G rows that are
// translates of one another, each with its own window shift (plan_vss in alfd.hip).  Lane (g, l) -- group g,
// lane l of the group -- accumulates entry l + L j of rows g, G + g, 2 G + g, 3 G + g in four registers and
// reduces each with the canonical L-lane tree: the result is the canonical one bit for bit.
template <int L, int EPI, int TAG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmv_vss_kernel(
    const uint8_t *__restrict__ stream, const int64_t *__restrict__ sb, const uint64_t *__restrict__ tab,
    const int32_t *__restrict__ cnt, int32_t stride, const int32_t *__restrict__ blk_seg_begin,
    const int32_t *__restrict__ blk_W, const int32_t *__restrict__ seg_col, const int32_t *__restrict__ seg_off,
    const int32_t *__restrict__ blk_dict_off, const int32_t *__restrict__ blk_dict_n,
    const double *__restrict__ dict, const double *__restrict__ x, const double *__restrict__ x_halo,
    int32_t n_local, double *__restrict__ y, double alpha, const double *__restrict__ d, double *__restrict__ y2) {
  extern __shared__ double xs[];
  constexpr int G = 64 / L, W64 = 1 + 4 * G, NW = 4;
  char *sm = (char *)xs;
  double *ds = (double *)(sm + kVsDictOff);
  double *xw = (double *)(sm + kVsWinOff);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l = lane & (L - 1), g = lane / L;
  const int64_t b = blockIdx.x;
  const int32_t nbatch = cnt[b];
  const uint8_t *sbase = stream + sb[b];
  {  // block frame: dictionary and x window (as in spmv_vs_kernel)
    const int32_t nd = blk_dict_n[b];
    for (int t = threadIdx.x; t < nd; t += 64 * NW) ds[t] = dict[blk_dict_off[b] + t];
    const int32_t W = blk_W[b];
    const int32_t s0 = blk_seg_begin[b], s1 = blk_seg_begin[b + 1];
    constexpr int U = 4;
    for (int32_t s = s0 + wave * U; s < s1; s += NW * U) {
      int32_t c0[U], o0[U], sl[U];
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t q = s + u < s1 ? s + u : s1 - 1;
        c0[u] = seg_col[q];
        o0[u] = seg_off[q];
        sl[u] = ((q + 1 < s1) ? seg_off[q + 1] : W) - o0[u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t c = c0[u] + (lane < sl[u] ? lane : 0);
        v[u] = (c < n_local) ? x[c] : x_halo[c - n_local];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (lane < sl[u]) xw[o0[u] + lane] = v[u];
    }
  }
  __syncthreads();
  // software pipeline over the wave's batches: header two batches ahead (scalar), rows / shifts / template
  // words one batch ahead (a fixed four predicated loads each, so the compiler's wait counts stay exact)
  const uint64_t *dsc = tab + ((int64_t)b * stride + wave) * W64;
  auto load_batch = [&](const uint64_t *dp, uint64_t hh, uint64_t (&e)[4], uint32_t (&f)[4]) {
    const uint8_t *fbp = sbase + 16u * (size_t)vs_off(hh);
    const int32_t ln = vs_len(hh);
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = dp[1 + q * G + g];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int32_t k = L * j + l;
      f[j] = k < ln ? *(const uint32_t *)(fbp + 4 * k) : 0u;   // k < len implies j < class
    }
  };
  uint64_t h1 = 0, h2 = 0;
  uint64_t e_cur[4] = {0, 0, 0, 0}, e_nxt[4] = {0, 0, 0, 0};
  uint32_t f_cur[4] = {0, 0, 0, 0}, f_nxt[4] = {0, 0, 0, 0};
  if (wave < nbatch) {
    h1 = dsc[0];
    load_batch(dsc, h1, e_cur, f_cur);
  }
  if (wave + NW < nbatch) h2 = dsc[(size_t)NW * W64];
  for (int32_t bi = wave; bi < nbatch; bi += NW) {
    uint64_t h3 = 0;
    if (bi + 2 * NW < nbatch) h3 = dsc[(size_t)2 * NW * W64];
    if (bi + NW < nbatch) load_batch(dsc + (size_t)NW * W64, h2, e_nxt, f_nxt);
    const int32_t len = vs_len(h1);
    const int cls = vs_cls(h1);
    int32_t row[4], sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      row[q] = (int32_t)(uint32_t)e_cur[q];
      sh[q] = (int32_t)(e_cur[q] >> 32);
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < cls) {   // wave-uniform
        const uint32_t f = f_cur[j];
        const uint32_t lc = f & 0x7ff8u;
        const double v = vs_lds_f64(kVsDictOff + ((f >> 12) & 0xff8u));
        double xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = vs_lds_f64(kVsWinOff + (uint32_t)((int32_t)lc + sh[q]));
        if (L * j + l < len) {
          asm volatile("");
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = fma(v, xv[q], acc[q]);
        }
      }
    }
    auto store = [&](double s, int32_t r, bool writer) {
      if (writer && r >= 0) {
        if (EPI == 0)
          y[r] = s;
        else if (EPI == 1)
          y[r] = fma(alpha, s, y[r]);
        else if (EPI == 2)
          y[r] = d[r] * s;
        else {
          y[r] = s;
          y2[r] = d[r] * s;
        }
      }
    };
    // canonical L-lane trees of the 4 x G rows, packed where the hardware exchanges allow and one-directional
    // towards the lane that stores (same pairs as group_reduce<L>; a + b is b + a bit for bit)
    const bool w8 = (lane & 7) == 0, h8 = (lane & 8) != 0;
    if (L == 32) {
      // acc[q]: lanes 0..31 row (q, g = 0), lanes 32..63 row (q, g = 1)
      swap16(acc[0], acc[1]);
      double u = acc[0] + acc[1];   // 16-lane rows: (q0, g0), (q1, g0), (q0, g1), (q1, g1)
      swap16(acc[2], acc[3]);
      double w = acc[2] + acc[3];   // (q2, g0), (q3, g0), (q2, g1), (q3, g1)
      u = u + dpp_xor_mov(u, 8);
      w = w + dpp_xor_mov(w, 8);
      double t = h8 ? w : u;
      t = t + dpp_shl<4>(t);
      t = t + dpp_shl<2>(t);
      t = t + dpp_shl<1>(t);
      const bool odd = (lane & 16) != 0;   // lane 16 rr + 8 h holds row q = (rr & 1) + 2 h of this lane's group
      const int32_t r = h8 ? (odd ? row[3] : row[2]) : (odd ? row[1] : row[0]);
      store(t, r, w8);
    } else if (L == 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = acc[q] + dpp_xor_mov(acc[q], 8);
#pragma unroll
      for (int p = 0; p < 2; ++p) {   // lanes 0..7 of a 16-lane row: row 2 p, lanes 8..15: row 2 p + 1 (of group g)
        double t = h8 ? acc[2 * p + 1] : acc[2 * p];
        t = t + dpp_shl<4>(t);
        t = t + dpp_shl<2>(t);
        t = t + dpp_shl<1>(t);
        store(t, h8 ? row[2 * p + 1] : row[2 * p], w8);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double t = acc[q];
        t = t + dpp_shl<4>(t);
        t = t + dpp_shl<2>(t);
        t = t + dpp_shl<1>(t);
        store(t, row[q], w8);
      }
    }
    dsc += (size_t)NW * W64;
    h1 = h2;
    h2 = h3;
#pragma unroll
    for (int q = 0; q < 4; ++q) e_cur[q] = e_nxt[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) f_cur[j] = f_nxt[j];
  }
}

}  // namespace alfd
