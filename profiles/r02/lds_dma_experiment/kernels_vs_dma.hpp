// LDS-streamed value-indexed SpMV ("vs" format) for gfx950.
//
// Same arithmetic as spmv_window_vib_kernel (canonical lane assignment, fma order and
// 64-lane tree of ALFD-arith v1 -- bit-identical results), different data path:
//
//  * the 3 B/nnz stream (16-bit window column + 8-bit dictionary code) is stored
//    BATCH-MAJOR: the rows of a row block are grouped by chunk count into batches of
//    4 rows (2 rows for rows of 193..384 entries); a batch's window columns, then its
//    codes, lie contiguously, 16-byte aligned.  A wave copies a whole batch into a private
//    LDS buffer with at most three 1 KiB `global_load_lds_dwordx4` (LDS-DMA: no VGPR
//    destination, ~20 issue cycles per KiB instead of ~8.5 per 64..128 B for the narrow
//    register loads of the vib kernel), the next batch in flight while the current one
//    is consumed with ds_read_u16 / ds_read_u8 at immediate offsets;
//  * a row block is an arbitrary LIST of rows (rowmap), not a run of the numbering: with
//    blocks that are bricks of the mesh graph (alfd_set_row_blocks) the x window a block
//    stages shrinks by half or more;
//  * row sums are collected in LDS and written (through rowmap) after the block's last
//    batch, so that the only vector-memory operations in the batch loop are the LDS-DMA
//    loads and `s_waitcnt vmcnt(0)` at the top of an iteration names exactly the batch
//    about to be read.
#pragma once

namespace alfd {

constexpr int kVsBuf = 3072;      // bytes of one stream buffer = largest batch (768 entries + padding) x 3 B
constexpr int kVsDictOff = 0;     // 256 dictionary doubles
constexpr int kVsYOff = 2048;     // row sums of the block (<= 256 rows)
constexpr int kVsBufOff = 4096;   // NW x 2 stream buffers
constexpr int kVsMaxRows = 250;   // rows per block (8-bit ids, 0xff = filler)
constexpr int kVsMaxLen = 384;    // longest row the format takes (class 6)
__host__ __device__ constexpr int vs_win_off(int NW, bool dma = true) { return kVsBufOff + (dma ? NW * 2 * kVsBuf : 0); }

// one LDS-DMA load: lane l moves 16 bytes from its own gsrc to lds_dst + 16 l (lds_dst wave-uniform)
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// copy `bytes` (multiple of 16, <= kVsBuf) from src (wave-uniform, 16-byte aligned) to LDS offset dst
__device__ __forceinline__ void vs_issue(const uint8_t *src, uint32_t bytes, uint32_t dst, int lane) {
  const uint32_t o = (uint32_t)lane * 16u;
  const uint8_t *p = src + o;
  if (o < bytes) glds16(p, dst);
  if (bytes > 1024u) {
    if (o + 1024u < bytes) glds16(p + 1024, dst + 1024u);
    if (bytes > 2048u) {
      if (o + 2048u < bytes) glds16(p + 2048, dst + 2048u);
    }
  }
}

// chunks J0 .. J0+JN-1 of the R rows of a batch; la / va: LDS byte address of this lane's
// first window column / first code of row i
template <int J0, int JN, int NCH, int R, int WINOFF>
__device__ __forceinline__ void vs_pass(const uint32_t (&la)[R], const uint32_t (&va)[R], const int32_t (&len)[R],
                                        int lane, const char *sm, double (&acc)[R]) {
  int32_t c[R][JN], iv[R][JN];
  bool ok[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      c[i][j] = *(const uint16_t *)(sm + la[i] + 128 * (J0 + j));
      iv[i][j] = *(const uint8_t *)(sm + va[i] + 64 * (J0 + j));
      if (J0 + j == NCH - 1) {
        ok[i] = 64 * (J0 + j) + lane < len[i];
        c[i][j] = ok[i] ? c[i][j] : 0;     // lanes past the row end read whatever follows in the buffer
        iv[i][j] = ok[i] ? iv[i][j] : 0;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    double xv[JN], v[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      xv[j] = *(const double *)(sm + WINOFF + 8 * c[i][j]);
      v[j] = *(const double *)(sm + kVsDictOff + 8 * iv[i][j]);
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) asm volatile("" : "+v"(xv[j]), "+v"(v[j]));
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      if (J0 + j == NCH - 1) {
        if (ok[i]) acc[i] = fma(v[j], xv[j], acc[i]);
      } else {
        acc[i] = fma(v[j], xv[j], acc[i]);
      }
    }
  }
}

// The same pass with the stream read by register loads (DMA = false): lc / vc point at the batch's
// window columns / codes in global memory, o[i] is row i's entry offset inside the batch.
template <int J0, int JN, int NCH, int R, int WINOFF>
__device__ __forceinline__ void vsg_pass(const uint32_t (&o)[R], const int32_t (&len)[R], int lane,
                                         const uint16_t *__restrict__ lc, const uint8_t *__restrict__ vc,
                                         const char *sm, double (&acc)[R]) {
  int32_t c[R][JN], iv[R][JN];
  bool ok[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int32_t e = 64 * (J0 + j) + lane;
      uint32_t k = o[i] + (uint32_t)e;
      if (J0 + j == NCH - 1) {
        ok[i] = e < len[i];
        k = o[i] + (uint32_t)(ok[i] ? e : 0);
      }
      c[i][j] = lc[k];
      iv[i][j] = vc[k];
    }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) {
    double xv[JN], v[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      xv[j] = *(const double *)(sm + WINOFF + 8 * c[i][j]);
      v[j] = *(const double *)(sm + kVsDictOff + 8 * iv[i][j]);
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) asm volatile("" : "+v"(xv[j]), "+v"(v[j]));
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      if (J0 + j == NCH - 1) {
        if (ok[i]) acc[i] = fma(v[j], xv[j], acc[i]);
      } else {
        acc[i] = fma(v[j], xv[j], acc[i]);
      }
    }
  }
}

template <int NCH, int R, int WINOFF>
__device__ __forceinline__ void vsg_rows(const uint32_t (&o)[R], const int32_t (&len)[R], int lane,
                                         const uint16_t *__restrict__ lc, const uint8_t *__restrict__ vc,
                                         const char *sm, double (&acc)[R]) {
  constexpr int A = NCH <= 3 ? NCH : (NCH + 1) / 2;
  vsg_pass<0, A, NCH, R, WINOFF>(o, len, lane, lc, vc, sm, acc);
  if (NCH > A) vsg_pass<A, (NCH > A ? NCH - A : 1), NCH, R, WINOFF>(o, len, lane, lc, vc, sm, acc);
}

template <int NCH, int R, int WINOFF>
__device__ __forceinline__ void vs_rows(const uint32_t (&la)[R], const uint32_t (&va)[R], const int32_t (&len)[R],
                                        int lane, const char *sm, double (&acc)[R]) {
  constexpr int A = NCH <= 3 ? NCH : (NCH + 1) / 2;
  vs_pass<0, A, NCH, R, WINOFF>(la, va, len, lane, sm, acc);
  if (NCH > A) vs_pass<A, (NCH > A ? NCH - A : 1), NCH, R, WINOFF>(la, va, len, lane, sm, acc);
}

// batch descriptor (4 x uint64, one per row): entry offset from the block's first entry (32) |
// entry count (16) | block-local row id, 0xff = filler (8) | class = ceil(count / 64) (8).
// Row 0 starts the batch (its offset is a multiple of 16); classes 1..3 hold 4 rows, 4..6 hold 2.
__device__ __forceinline__ void vs_extent(const uint64_t (&desc)[4], uint32_t &eb, uint32_t &Lb) {
  eb = (uint32_t)desc[0];
  uint32_t end = eb;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t e = (uint32_t)desc[i] + (uint32_t)((desc[i] >> 32) & 0xffff);
    end = e > end ? e : end;
  }
  Lb = (end - eb + 15u) & ~15u;
}

template <int EPI, int TAG, int NW, bool DMA>
__global__ __launch_bounds__(64 * NW) void spmv_vs_kernel(
    const uint8_t *__restrict__ stream, const int64_t *__restrict__ sb, const uint64_t *__restrict__ tab,
    const int32_t *__restrict__ cnt, int32_t stride, const int32_t *__restrict__ rowmap, int32_t rbs,
    const int32_t *__restrict__ blk_seg_begin, const int32_t *__restrict__ blk_W,
    const int32_t *__restrict__ seg_col, const int32_t *__restrict__ seg_off,
    const int32_t *__restrict__ blk_dict_off, const int32_t *__restrict__ blk_dict_n,
    const double *__restrict__ dict, const double *__restrict__ x, const double *__restrict__ x_halo,
    int32_t n_local, double *__restrict__ y, double alpha, const double *__restrict__ d, double *__restrict__ y2,
    int xcd_remap) {
  extern __shared__ double xs[];
  constexpr int WINOFF = vs_win_off(NW, DMA);
  char *sm = (char *)xs;
  double *ys = (double *)(sm + kVsYOff);
  double *ds = (double *)(sm + kVsDictOff);
  double *xw = (double *)(sm + WINOFF);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t b = blockIdx.x;
  if (xcd_remap) {
    const int64_t nwg = gridDim.x, q = nwg / 8, rm = nwg % 8, xcd = b % 8, idx = b / 8;
    b = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + idx;
  }
  const int32_t cn = cnt[b];
  const int32_t nbatch = cn & 0xffff, nrows_b = cn >> 16;
  const uint8_t *sbase = stream + sb[b];
  const uint64_t *bt = tab + ((int64_t)b * stride + wave) * 4;
  const uint32_t mybuf = (uint32_t)(kVsBufOff + wave * 2 * kVsBuf);
  uint64_t cur[4] = {0, 0, 0, 0}, nx[4] = {0, 0, 0, 0};
  if (wave < nbatch) {
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = bt[i];
    uint32_t eb, Lb;
    vs_extent(cur, eb, Lb);
    if (DMA) vs_issue(sbase + 3u * (size_t)eb, 3u * Lb, mybuf, lane);
  }
  bt += 4 * NW;
  if (wave + NW < nbatch) {
#pragma unroll
    for (int i = 0; i < 4; ++i) nx[i] = bt[i];
  }
  {  // block frame: dictionary and x window
    const int32_t nd = blk_dict_n[b];
    for (int t = threadIdx.x; t < nd; t += 64 * NW) ds[t] = dict[blk_dict_off[b] + t];
    const int32_t W = blk_W[b];
    const int32_t s0 = blk_seg_begin[b], s1 = blk_seg_begin[b + 1];
    for (int32_t s = s0 + wave; s < s1; s += NW) {
      const int32_t c0 = seg_col[s], o0 = seg_off[s];
      const int32_t slen = ((s + 1 < s1) ? seg_off[s + 1] : W) - o0;
      for (int32_t i = lane; i < slen; i += 64) {
        const int32_t c = c0 + i;
        xw[o0 + i] = (c < n_local) ? x[c] : x_halo[c - n_local];
      }
    }
  }
  __syncthreads();
  uint32_t par = 0;
  for (int32_t bi = wave; bi < nbatch; bi += NW) {
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this batch's stream has landed
    uint64_t desc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) desc[i] = cur[i];
    if (bi + NW < nbatch) {
#pragma unroll
      for (int i = 0; i < 4; ++i) cur[i] = nx[i];
      uint32_t eb2, Lb2;
      vs_extent(cur, eb2, Lb2);
      if (DMA) vs_issue(sbase + 3u * (size_t)eb2, 3u * Lb2, mybuf + (par ^ 1u) * kVsBuf, lane);
      bt += 4 * NW;
      if (bi + 2 * NW < nbatch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) nx[i] = bt[i];
      }
    }
    uint32_t eb, Lb;
    vs_extent(desc, eb, Lb);
    const int cls = (int)(desc[0] >> 56);
    const uint32_t buf = mybuf + par * kVsBuf;
    int32_t len[4];
    uint32_t la[4], va[4], ro[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t o = (uint32_t)desc[i] - eb;
      ro[i] = o;
      len[i] = (int32_t)((desc[i] >> 32) & 0xffff);
      la[i] = buf + 2u * o + 2u * (uint32_t)lane;
      va[i] = buf + 2u * Lb + o + (uint32_t)lane;
    }
    const uint16_t *glc = (const uint16_t *)(sbase + 3u * (size_t)eb);
    const uint8_t *gvc = sbase + 3u * (size_t)eb + 2u * (size_t)Lb;
    double s;
    uint64_t dq;
    bool writer;
    if (cls <= 3) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      if (DMA) {
        switch (cls) {
          case 1: vs_rows<1, 4, WINOFF>(la, va, len, lane, sm, acc); break;
          case 2: vs_rows<2, 4, WINOFF>(la, va, len, lane, sm, acc); break;
          case 3: vs_rows<3, 4, WINOFF>(la, va, len, lane, sm, acc); break;
          default: break;
        }
      } else {
        switch (cls) {
          case 1: vsg_rows<1, 4, WINOFF>(ro, len, lane, glc, gvc, sm, acc); break;
          case 2: vsg_rows<2, 4, WINOFF>(ro, len, lane, glc, gvc, sm, acc); break;
          case 3: vsg_rows<3, 4, WINOFF>(ro, len, lane, glc, gvc, sm, acc); break;
          default: break;
        }
      }
      s = reduce_rows4(acc[0], acc[1], acc[2], acc[3]);
      const int q = lane >> 4;  // 16-lane row q holds the tree of batch row {0, 2, 1, 3}[q]
      dq = q == 0 ? desc[0] : (q == 1 ? desc[2] : (q == 2 ? desc[1] : desc[3]));
      writer = (lane & 15) == 0;
    } else {
      const uint32_t la2[2] = {la[0], la[1]}, va2[2] = {va[0], va[1]};
      const int32_t len2[2] = {len[0], len[1]};
      double acc[2] = {0.0, 0.0};
      if (DMA) {
        switch (cls) {
          case 4: vs_rows<4, 2, WINOFF>(la2, va2, len2, lane, sm, acc); break;
          case 5: vs_rows<5, 2, WINOFF>(la2, va2, len2, lane, sm, acc); break;
          default: vs_rows<6, 2, WINOFF>(la2, va2, len2, lane, sm, acc); break;
        }
      } else {
        const uint32_t ro2[2] = {ro[0], ro[1]};
        switch (cls) {
          case 4: vsg_rows<4, 2, WINOFF>(ro2, len2, lane, glc, gvc, sm, acc); break;
          case 5: vsg_rows<5, 2, WINOFF>(ro2, len2, lane, glc, gvc, sm, acc); break;
          default: vsg_rows<6, 2, WINOFF>(ro2, len2, lane, glc, gvc, sm, acc); break;
        }
      }
      s = reduce_rows2(acc[0], acc[1]);
      dq = lane < 32 ? desc[0] : desc[1];
      writer = (lane & 31) == 0;
    }
    const int id = (int)((dq >> 48) & 0xff);
    if (writer && id != 0xff) ys[id] = s;
    par ^= 1u;
  }
  __syncthreads();
  for (int32_t i = threadIdx.x; i < nrows_b; i += 64 * NW) {
    const int64_t r = rowmap[(int64_t)b * rbs + i];
    const double s = ys[i];
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
}

}  // namespace alfd
