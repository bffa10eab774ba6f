// kernels.hpp -- hand-written HIP kernels for gfx950 (CDNA4, wave64).
//
// All kernels are HBM-bandwidth-bound fp64 sparse / streaming work (no MFMA by
// design: SURVEY.md 8(d)).  Every reduction follows the fixed evaluation order
// of "ALFD-arith v1"  so that results are bit-identical to
// the CPU oracle:
//   SpMV row : L lanes per row; lane l does fma over entries k0+l, k0+l+L, ..;
//              xor-butterfly over the L lanes (lane 0 == binary tree).
//   dot      : 4096-element chunk per 256-thread block; thread t takes the
//              element pairs base + e*512 + 2t (e=0..7) with fma; 64-lane
//              butterfly; (w0+w1)+(w2+w3); second stage = one block doing
//              thread-strided adds and the same tree.
// Replaces: dealii::SparseMatrix<double>::vmult /
// vmult_add / Tvmult behind linear_operator() (stokes_immersed_boundary.cc:923-929)
// and the Vector dot / add / sadd / equ primitives inside SolverCG / SolverFGMRES.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alfd {

constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr int64_t kChunk = 4096;     // dot chunk == vector padding granule
constexpr int kMaxBasis = 64;        // max FGMRES basis vectors handled by the batched kernels

// --------------------------------------------------------------------------
// Several 64-lane trees at once, on the vector ALU only (no LDS-pipe shuffles).
// The canonical tree adds partner lanes l^32, l^16, l^8, l^4, l^2, l^1 in that
// order.  v_permlane32_swap / v_permlane16_swap exchange half-waves / 16-lane rows
// between two registers, so the first steps of two (four) rows cost one add and
// leave the rows packed side by side; the last four steps are DPP moves inside
// 16-lane rows and serve all packed rows together.  Same pairs, same order =>
// same bits as group_reduce<64>.
__device__ __forceinline__ double dpp_xor_mov(double v, int which) {
  int lo = __double2loint(v), hi = __double2hiint(v), rl, rh;
  if (which == 8) {
    rl = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, true);  // row_ror:8 (bound_ctrl: no "old" operand to set up)
    rh = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, true);
  } else if (which == 4) {
    rl = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0x5, false);   // row_shl:4 -> banks 0,2
    rl = __builtin_amdgcn_update_dpp(rl, lo, 0x114, 0xf, 0xa, false);  // row_shr:4 -> banks 1,3
    rh = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0x5, false);
    rh = __builtin_amdgcn_update_dpp(rh, hi, 0x114, 0xf, 0xa, false);
  } else if (which == 2) {
    rl = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    rh = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true);
  } else {
    rl = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    rh = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true);
  }
  return __hiloint2double(rh, rl);
}
// a -> [a_lo | b_lo], b -> [a_hi | b_hi]  (32-lane halves)
__device__ __forceinline__ void swap32(double &a, double &b) {
  auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)h[0], (int)l[0]);
  b = __hiloint2double((int)h[1], (int)l[1]);
}
// odd 16-lane rows of a <-> even rows of b
__device__ __forceinline__ void swap16(double &a, double &b) {
  auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)h[0], (int)l[0]);
  b = __hiloint2double((int)h[1], (int)l[1]);
}
// --------------------------------------------------------------------------
// wave / group reductions: the canonical tree adds partner lanes l^(L/2), ..., l^2, l^1 in that
// order (all lanes end with the tree value).  Evaluated on the vector ALU: half-wave and
// 16-lane-row exchanges by v_permlane32_swap / v_permlane16_swap on two copies of the value,
// the last four steps by DPP moves -- no ds_bpermute, same pairs in the same order.
template <int L>
__device__ __forceinline__ double group_reduce(double v) {
  if (L >= 64) {
    double a = v, b = v;
    swap32(a, b);  // a = [lo | lo], b = [hi | hi]
    v = a + b;
  }
  if (L >= 32) {
    double p = v, q = v;
    swap16(p, q);  // p = rows [0,0,2,2], q = rows [1,1,3,3]
    v = p + q;
  }
  if (L >= 16) v = v + dpp_xor_mov(v, 8);
  if (L >= 8) v = v + dpp_xor_mov(v, 4);
  if (L >= 4) v = v + dpp_xor_mov(v, 2);
  if (L >= 2) v = v + dpp_xor_mov(v, 1);
  return v;
}

// lane l <- lane l + N of its 16-lane row (row_shl:N; the top N lanes of a row keep 0)
template <int N>
__device__ __forceinline__ double dpp_shl(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int rl = __builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xf, 0xf, true);   // bound_ctrl: the lanes shifted in read 0
  const int rh = __builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xf, 0xf, true);
  return __hiloint2double(rh, rl);
}
// Last four tree steps inside 16-lane rows, valid in lane 0 of every row (the lane that stores): after the
// symmetric l^8 step lane l only takes its partner l + k -- the pairs of the canonical tree, and a + b is
// b + a bit for bit -- which needs one DPP move per word instead of the two masked ones of an exchange.
__device__ __forceinline__ double row16_tree(double v) {
  v = v + dpp_xor_mov(v, 8);
  v = v + dpp_shl<4>(v);
  v = v + dpp_shl<2>(v);
  v = v + dpp_shl<1>(v);
  return v;
}
// Trees of 4 rows; the result of row q sits in the 16-lane row kRow4Pos(q).
__device__ __forceinline__ double reduce_rows4(double a0, double a1, double a2, double a3) {
  swap32(a0, a1);
  double p = a0 + a1;  // [row0 | row1], step l^32 done
  swap32(a2, a3);
  double q = a2 + a3;  // [row2 | row3]
  swap16(p, q);
  return row16_tree(p + q);  // 16-lane rows: row0, row2, row1, row3
}
// Trees of 8 rows, packed as far as the hardware swaps allow: after the l^32 and l^16 steps two registers
// hold 4 rows x 16 lanes each; the l^8 step folds each to 8 lanes per row and the two are merged into ONE
// register (lanes 0..7 of every 16-lane row: rows 0..3, lanes 8..15: rows 4..7); the last three steps only
// feed the lane that stores (lane l takes its partner l + k: same pairs as the canonical tree, and a + b is
// b + a bit for bit).  Result of row {0, 2, 1, 3}[lane >> 4] + 4 * ((lane >> 3) & 1) in lanes with lane % 8 == 0.
__device__ __forceinline__ double reduce_rows8(double a0, double a1, double a2, double a3, double a4, double a5,
                                               double a6, double a7, int lane) {
  swap32(a0, a1);
  double p01 = a0 + a1;
  swap32(a2, a3);
  double p23 = a2 + a3;
  swap32(a4, a5);
  double p45 = a4 + a5;
  swap32(a6, a7);
  double p67 = a6 + a7;
  swap16(p01, p23);
  double u = p01 + p23;  // 16-lane rows: row0, row2, row1, row3
  swap16(p45, p67);
  double w = p45 + p67;  // row4, row6, row5, row7
  u = u + dpp_xor_mov(u, 8);
  w = w + dpp_xor_mov(w, 8);
  double x = (lane & 8) ? w : u;
  x = x + dpp_shl<4>(x);
  x = x + dpp_shl<2>(x);
  x = x + dpp_shl<1>(x);
  return x;
}

// lane l <- lane l - N of its 16-lane row (row_shr:N; the bottom N lanes of a row read 0)
template <int N>
__device__ __forceinline__ double dpp_shr(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int rl = __builtin_amdgcn_update_dpp(0, lo, 0x110 + N, 0xf, 0xf, true);
  const int rh = __builtin_amdgcn_update_dpp(0, hi, 0x110 + N, 0xf, 0xf, true);
  return __hiloint2double(rh, rl);
}
// Trees of 16 rows in 66 vector instructions: the l^32 and l^16 steps leave four registers of 4 rows x 16 lanes, the l^8
// step folds each to 8 lanes per row and pairs of them merge (lanes 0..7 / 8..15 of a 16-lane row), the l^4 step folds
// to 4 lanes per row -- the first register towards the lanes with bit 2 clear (partner l + 4), the second towards those
// with bit 2 set (partner l - 4; a + b is b + a bit for bit) -- and the two merge into ONE register; the last two steps
// feed the lane that stores.  Result of row {0, 2, 1, 3}[lane >> 4] + 4 * ((lane >> 3) & 1) + 8 * ((lane >> 2) & 1) in
// the lanes with lane % 4 == 0.  Same pairs in the same order as group_reduce<64> for every row.
__device__ __forceinline__ double reduce_rows16(double (&a)[16], int lane) {
  double p[8], u[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    swap32(a[2 * j], a[2 * j + 1]);
    p[j] = a[2 * j] + a[2 * j + 1];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    swap16(p[2 * j], p[2 * j + 1]);
    u[j] = p[2 * j] + p[2 * j + 1];   // 16-lane rows: rows 4 j + {0, 2, 1, 3}
    u[j] = u[j] + dpp_xor_mov(u[j], 8);
  }
  const bool h = (lane & 8) != 0;
  double x0 = h ? u[1] : u[0], x1 = h ? u[3] : u[2];
  x0 = x0 + dpp_shl<4>(x0);
  x1 = x1 + dpp_shr<4>(x1);
  double z = (lane & 4) ? x1 : x0;
  z = z + dpp_shl<2>(z);
  z = z + dpp_shl<1>(z);
  return z;
}

// Trees of 2 rows: row 0 in lanes 0..31, row 1 in lanes 32..63.
__device__ __forceinline__ double reduce_rows2(double a0, double a1) {
  swap32(a0, a1);
  double p = a0 + a1, q = p;
  swap16(p, q);  // p = rows [0,0,2,2], q = rows [1,1,3,3] of the old p
  return row16_tree(p + q);
}

// block of 256 threads -> (w0+w1)+(w2+w3), valid in thread 0
__device__ __forceinline__ double block_reduce_256(double v, double *lds4) {
  v = group_reduce<64>(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds4[wave] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// --------------------------------------------------------------------------
// CSR SpMV, L lanes per row, 256/L rows per workgroup, grid-stride over row
// groups so that the resident workgroups sweep a compact moving window of rows
// (x re-use out of L2 / Infinity Cache).
//   EPI 0: y[r]  = s
//   EPI 1: y[r]  = fma(alpha, s, y[r])          (vmult_add with a scalar)
//   EPI 2: y[r]  = d[r] * s                      (diag-scaled: t = invW .* (C x))
//   EPI 3: y[r]  = s and y2[r] = d[r] * s        (C x kept for the constraint row too)
// SPARSE: rows[] lists the non-empty rows, rp[] is indexed by list position.
// Columns >= n_local read the halo buffer (multi-GPU row partition).
template <int L, int EPI, bool SPARSE>
__global__ __launch_bounds__(kBlock) void spmv_kernel(int64_t nrows, const int64_t *__restrict__ rp,
                                                      const int32_t *__restrict__ col,
                                                      const double *__restrict__ val,
                                                      const int32_t *__restrict__ rows,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ x_halo, int32_t n_local,
                                                      double *__restrict__ y, double alpha,
                                                      const double *__restrict__ d,
                                                      double *__restrict__ y2) {
  constexpr int RPB = kBlock / L;
  const int lane = threadIdx.x % L;
  const int sub = threadIdx.x / L;
  const int64_t ngroups = (nrows + RPB - 1) / RPB;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t r = g * RPB + sub;
    if (r < nrows) {  // uniform within the L-lane group
      const int64_t k0 = rp[r], k1 = rp[r + 1];
      double acc = 0.0;
      for (int64_t k = k0 + lane; k < k1; k += L) {
        const int32_t c = col[k];
        const double xv = (c < n_local) ? x[c] : x_halo[c - n_local];
        acc = fma(val[k], xv, acc);
      }
      acc = group_reduce<L>(acc);
      if (lane == 0) {
        const int64_t ro = SPARSE ? (int64_t)rows[r] : r;
        if (EPI == 0)
          y[ro] = acc;
        else if (EPI == 1)
          y[ro] = fma(alpha, acc, y[ro]);
        else if (EPI == 2)
          y[ro] = d[ro] * acc;
        else {
          y[ro] = acc;
          y2[ro] = d[ro] * acc;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------
// Streaming CSR SpMV for long rows (canonical L = 64).  One wave owns a batch
// of R consecutive rows and streams their CONTIGUOUS nnz range with perfectly
// coalesced, U-way unrolled col/val loads: U*768 B (+ the x gathers) in flight
// per wave instead of one row's worth, and no idle lanes at row tails.
// Bit-compatible with the canonical order: entry k of row i belongs to
// canonical lane (k - k0_i) mod 64; the stream lane that sees it is
// (k - k_begin) mod 64 and stays the same for all later entries of that
// canonical lane (k advances by 64), so each stream lane's per-row accumulator
// IS one canonical lane partial, rotated by d_i = (k0_i - k_begin) mod 64.  One
// ds_bpermute un-rotates before the butterfly.
template <int R, int U, int EPI, bool NT>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(
    int64_t nrows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, const double *__restrict__ x_halo,
    int32_t n_local, double *__restrict__ y, double alpha, const double *__restrict__ d,
    double *__restrict__ y2) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nbatches = (nrows + R - 1) / R;
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < nbatches; b += (int64_t)gridDim.x * 4) {
    const int64_t r0 = b * R;
    int64_t kb[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) kb[i] = rp[r0 + i < nrows ? r0 + i : nrows];  // wave-uniform
    const int64_t k_begin = kb[0];
    const int32_t n_batch = (int32_t)(kb[R] - k_begin);
    int32_t rel[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) rel[i] = (int32_t)(kb[i] - k_begin);
    double acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.0;
    const int32_t *cb = col + k_begin;
    const double *vb = val + k_begin;
    for (int32_t base = 0; base < n_batch; base += 64 * U) {
      int32_t c[U];
      double v[U], xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t o = base + 64 * u + lane;
        if (o < n_batch) {
          c[u] = NT ? __builtin_nontemporal_load(cb + o) : cb[o];
          v[u] = NT ? __builtin_nontemporal_load(vb + o) : vb[o];
        } else {
          c[u] = -1;
          v[u] = 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xv[u] = 0.0;
        if (c[u] >= 0) xv[u] = (c[u] < n_local) ? x[c[u]] : x_halo[c[u] - n_local];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t o = base + 64 * u + lane;
        if (c[u] >= 0) {
#pragma unroll
          for (int i = 0; i < R; ++i)
            if (o >= rel[i] && o < rel[i + 1]) acc[i] = fma(v[u], xv[u], acc[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      if (r < nrows) {  // wave-uniform
        const int dsh = rel[i] & 63;
        double s = __shfl(acc[i], (lane + dsh) & 63, 64);
        s = group_reduce<64>(s);
        if (lane == 0) {
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
    }
  }
}

// --------------------------------------------------------------------------
// LDS-windowed streaming SpMV (the roofline kernel for long-row matrices).
// Rows are grouped in blocks of RB; at upload the host finds, per block, the
// union of column intervals its rows touch (the "x window"), lays the
// intervals out back to back and rewrites every column index as a 16-bit
// offset into that window.  A workgroup then
//   1. stages its window from x into LDS with coalesced loads (each x entry is
//      fetched once per block instead of once per use),
//   2. streams val (8 B) + local col (2 B) = 10 B/nnz instead of 12 B/nnz,
//   3. gathers from LDS.
// Same fma/tree order as spmv_stream_kernel => bit-identical results.
// Blocks whose window exceeds the LDS budget (blk_W < 0) gather from global
// memory with the original 32-bit columns.
// TAG only distinguishes instantiations (fine operator = 0, multigrid level matrices = 1)
// so that profiler summaries report them separately.
template <int R, int U, int EPI, int TAG = 0>
__global__ __launch_bounds__(kBlock) void spmv_window_kernel(
    int64_t nrows, int32_t RB, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
    const uint16_t *__restrict__ lcol, const double *__restrict__ val,
    const int32_t *__restrict__ blk_seg_begin, const int32_t *__restrict__ blk_W,
    const int32_t *__restrict__ seg_col, const int32_t *__restrict__ seg_off,
    const double *__restrict__ x, const double *__restrict__ x_halo, int32_t n_local,
    double *__restrict__ y, double alpha, const double *__restrict__ d, double *__restrict__ y2,
    int xcd_remap) {
  extern __shared__ double xs[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware block order: workgroups with equal blockIdx % 8 share an XCD (and its L2),
  // so each XCD is given a CONTIGUOUS run of row blocks -- neighbouring blocks share
  // most of their x window, which then hits in that XCD's L2 (bijective remap for any
  // grid size; affects speed only).
  int64_t b = blockIdx.x;
  if (xcd_remap) {
    const int64_t nwg = gridDim.x, q = nwg / 8, rm = nwg % 8, xcd = b % 8, idx = b / 8;
    b = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + idx;
  }
  const int32_t W = blk_W[b];
  if (W >= 0) {
    const int32_t s0 = blk_seg_begin[b], s1 = blk_seg_begin[b + 1];
    for (int32_t s = s0 + wave; s < s1; s += 4) {
      const int32_t c0 = seg_col[s], o0 = seg_off[s];
      const int32_t len = ((s + 1 < s1) ? seg_off[s + 1] : W) - o0;
      for (int32_t i = lane; i < len; i += 64) {
        const int32_t c = c0 + i;
        xs[o0 + i] = (c < n_local) ? x[c] : x_halo[c - n_local];
      }
    }
    __syncthreads();
  }
  const int64_t row_begin = b * RB;
  const int64_t row_end = (row_begin + RB < nrows) ? row_begin + RB : nrows;
  for (int64_t r0 = row_begin + (int64_t)wave * R; r0 < row_end; r0 += 4 * R) {
    int64_t kb[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) kb[i] = rp[r0 + i < row_end ? r0 + i : row_end];  // wave-uniform
    const int64_t k_begin = kb[0];
    const int32_t n_batch = (int32_t)(kb[R] - k_begin);
    int32_t rel[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) rel[i] = (int32_t)(kb[i] - k_begin);
    double acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.0;
    const double *vb = val + k_begin;
    if (W >= 0) {
      const uint16_t *cb = lcol + k_begin;
      for (int32_t base = 0; base < n_batch; base += 64 * U) {
        int32_t c[U];
        double v[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t o = base + 64 * u + lane;
          if (o < n_batch) {
            c[u] = cb[o];
            v[u] = vb[o];
          } else {
            c[u] = -1;
            v[u] = 0.0;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = c[u] >= 0 ? xs[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t o = base + 64 * u + lane;
          if (c[u] >= 0) {
#pragma unroll
            for (int i = 0; i < R; ++i)
              if (o >= rel[i] && o < rel[i + 1]) acc[i] = fma(v[u], xv[u], acc[i]);
          }
        }
      }
    } else {
      const int32_t *cb = col + k_begin;
      for (int32_t base = 0; base < n_batch; base += 64 * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int32_t o = base + 64 * u + lane;
          if (o < n_batch) {
            const int32_t c = cb[o];
            const double xv = (c < n_local) ? x[c] : x_halo[c - n_local];
            const double v = vb[o];
#pragma unroll
            for (int i = 0; i < R; ++i)
              if (o >= rel[i] && o < rel[i + 1]) acc[i] = fma(v, xv, acc[i]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      if (r < row_end) {  // wave-uniform
        const int dsh = rel[i] & 63;
        double s = __shfl(acc[i], (lane + dsh) & 63, 64);
        s = group_reduce<64>(s);
        if (lane == 0) {
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
    }
  }
}

// --------------------------------------------------------------------------
// Value-indexed blocks: finite-element matrices on the reference's uniformly refined
// hyper_cube grids repeat a few hundred distinct entry values.  At upload each row block
// whose entries take <= 256 distinct bit patterns gets a private dictionary (stored once,
// staged into LDS next to the x window) and an 8-bit code per entry, so the stream is
// 2 B (window column) + 1 B (value code) = 3 B/nnz instead of 10.  The looked-up double
// is the stored one bit for bit, so results do not change.  Blocks with 257..512 distinct
// values use 16-bit codes (4 B/nnz); blocks beyond that keep streaming the 8-byte values.
//
// Row-batched window kernel for value-indexed matrices.  With 3 B/nnz the stream
// is no longer HBM-bound but latency-bound, so the kernel is organised to keep
// many independent loads in flight and to spend few instructions per entry:
// a wave takes R consecutive rows; lane l reads entries k0_i + l + 64 j of row i
// -- exactly the canonical lane assignment, so there is no rotation and no
// per-row predication -- and issues the loads of all R rows (J chunks each)
// before the first use.  Idle lanes at row tails cost issue slots, not traffic.
// MODE 0: dictionary values + LDS window; 1: 8-byte values + LDS window;
// MODE 2: 8-byte values, x gathered from global memory (block without a window);
// MODE 3: as 0 with 16-bit value codes (blocks with 257..512 distinct values).
constexpr int32_t kDictWide = 1 << 16;    // blk_dict_n flag: 16-bit codes
constexpr int32_t kDictMaxEntries = 512;  // LDS slots reserved for a block dictionary

// ks[i]: offset of row i's first entry from the block's first entry (the *_b pointers are
// pre-offset to it); len[i]: its entry count.
template <int R, int J, int MODE>
__device__ __forceinline__ void vi_rows(const uint32_t (&ks)[R], const int32_t (&len)[R], int lane,
                                        const uint16_t *__restrict__ lcol_b, const uint8_t *__restrict__ vidx_b,
                                        const uint16_t *__restrict__ vidw_b,
                                        const int32_t *__restrict__ col_b, const double *__restrict__ val_b,
                                        const double *xs, const double *ds, const double *__restrict__ x,
                                        const double *__restrict__ x_halo, int32_t n_local, double (&acc)[R]) {
  int32_t maxlen = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) maxlen = len[i] > maxlen ? len[i] : maxlen;
  for (int32_t base = 0; base < maxlen; base += 64 * J) {  // one pass unless a row has > 64 J entries
    // One wave-uniform guard per row and pass; inside it the J chunks are straight-line code:
    // lanes past the row end re-read the row's first entry (a cache hit) and are masked at
    // the fma.  Everything loaded under a guard is only used under the same guard, so the
    // arrays need no defaults, and all loads of the R rows are in flight before the first use.
    int32_t c[R][J], iv[R][J];
    double v[R][J];
    bool ok[R][J];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (base < len[i]) {  // wave-uniform
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const int32_t o = base + 64 * j + lane;
          ok[i][j] = o < len[i];
          const uint32_t k = ks[i] + (uint32_t)(ok[i][j] ? o : 0);
          if (MODE == 2) c[i][j] = col_b[k];
          else c[i][j] = lcol_b[k];
          if (MODE == 0) iv[i][j] = vidx_b[k];
          else if (MODE == 3) iv[i][j] = vidw_b[k];
          else v[i][j] = val_b[k];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      if (base < len[i]) {  // wave-uniform
        double xv[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
          if (MODE == 2) xv[j] = (c[i][j] < n_local) ? x[c[i][j]] : x_halo[c[i][j] - n_local];
          else xv[j] = xs[c[i][j]];
          if (MODE == 0 || MODE == 3) v[i][j] = ds[iv[i][j]];
        }
        // keep the gathers of all J chunks ahead of the masked fmas (the compiler would
        // otherwise sink each ds_read into its exec-masked block and wait on it there)
#pragma unroll
        for (int j = 0; j < J; ++j) asm volatile("" : "+v"(xv[j]), "+v"(v[i][j]));
#pragma unroll
        for (int j = 0; j < J; ++j)
          if (ok[i][j]) acc[i] = fma(v[i][j], xv[j], acc[i]);
      }
    }
  }
}

template <int J0, int JN, int NCH, int MODE>
__device__ __forceinline__ void vib_pass(const uint32_t (&ks)[4], const int32_t (&len)[4], int lane,
                                         const uint16_t *__restrict__ lcol_b,
                                         const uint8_t *__restrict__ vidx_b, const uint16_t *__restrict__ vidw_b,
                                         const double *xs, const double *ds, double (&acc)[4]) {
  // chunks J0 .. J0+JN-1 of every row of the batch; chunk NCH-1 is the only partial one
  int32_t c[4][JN], iv[4][JN];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int32_t o = 64 * (J0 + j) + lane;
      uint32_t k = ks[i] + (uint32_t)o;
      if (J0 + j == NCH - 1) {
        ok[i] = o < len[i];
        k = ks[i] + (uint32_t)(ok[i] ? o : 0);
      }
      c[i][j] = lcol_b[k];
      iv[i][j] = MODE == 3 ? (int32_t)vidw_b[k] : (int32_t)vidx_b[k];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double xv[JN], v[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      xv[j] = xs[c[i][j]];
      v[j] = ds[iv[i][j]];
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

// Batch of 4 rows that all have exactly NCH 64-entry chunks (the last one possibly partial):
// no guards at all, and only the last chunk of each row carries a lane mask.  At most 3
// chunks per row are in flight at a time (12 chunk loads, ~56 VGPRs: 8 waves per SIMD).
// MODE 0 (8-bit codes) or 3 (16-bit codes).
template <int NCH, int MODE>
__device__ __forceinline__ void vib_rows(const uint32_t (&ks)[4], const int32_t (&len)[4], int lane,
                                         const uint16_t *__restrict__ lcol_b,
                                         const uint8_t *__restrict__ vidx_b, const uint16_t *__restrict__ vidw_b,
                                         const double *xs, const double *ds, double (&acc)[4]) {
  constexpr int A = NCH <= 3 ? NCH : (NCH + 1) / 2;
  vib_pass<0, A, NCH, MODE>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc);
  if (NCH > A) vib_pass<A, (NCH > A ? NCH - A : 1), NCH, MODE>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc);
}

template <int R, int J, int EPI>
__global__ __launch_bounds__(kBlock) void spmv_window_vi_kernel(
    int64_t nrows, int32_t RB, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
    const uint16_t *__restrict__ lcol, const double *__restrict__ val,
    const int32_t *__restrict__ blk_seg_begin, const int32_t *__restrict__ blk_W,
    const int32_t *__restrict__ seg_col, const int32_t *__restrict__ seg_off,
    const double *__restrict__ x, const double *__restrict__ x_halo, int32_t n_local,
    double *__restrict__ y, double alpha, const double *__restrict__ d, double *__restrict__ y2,
    const uint8_t *__restrict__ vidx, const uint16_t *__restrict__ vidw,
    const int32_t *__restrict__ blk_dict_off, const int32_t *__restrict__ blk_dict_n,
    const double *__restrict__ dict, int32_t dict_lds_off) {
  extern __shared__ double xs[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b = blockIdx.x;
  const int32_t W = blk_W[b];
  const double *ds = xs + dict_lds_off;
  int mode = 2;  // block-uniform
  if (W >= 0) {
    const int32_t nd = blk_dict_n[b];
    mode = nd < 0 ? 1 : ((nd & kDictWide) ? 3 : 0);
    const int32_t ndv = nd & 0xffff;
    if (nd >= 0)
      for (int t = threadIdx.x; t < ndv; t += kBlock) xs[dict_lds_off + t] = dict[blk_dict_off[b] + t];
    const int32_t s0 = blk_seg_begin[b], s1 = blk_seg_begin[b + 1];
    for (int32_t s = s0 + wave; s < s1; s += 4) {
      const int32_t c0 = seg_col[s], o0 = seg_off[s];
      const int32_t len = ((s + 1 < s1) ? seg_off[s + 1] : W) - o0;
      for (int32_t i = lane; i < len; i += 64) {
        const int32_t c = c0 + i;
        xs[o0 + i] = (c < n_local) ? x[c] : x_halo[c - n_local];
      }
    }
    __syncthreads();
  }
  const int64_t row_begin = b * RB;
  const int64_t row_end = (row_begin + RB < nrows) ? row_begin + RB : nrows;
  for (int64_t r0 = row_begin + (int64_t)wave * R; r0 < row_end; r0 += 4 * R) {
    int64_t kb[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) kb[i] = rp[r0 + i < row_end ? r0 + i : row_end];  // wave-uniform
    uint32_t ks[R];
    int32_t len[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      ks[i] = (uint32_t)(kb[i] - kb[0]);
      len[i] = (int32_t)(kb[i + 1] - kb[i]);
    }
    const uint16_t *lcol_b = lcol + kb[0];
    const uint8_t *vidx_b = vidx + kb[0];
    const uint16_t *vidw_b = vidw + kb[0];
    const int32_t *col_b = col + kb[0];
    const double *val_b = val + kb[0];
    double acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.0;
    if (mode == 3) vi_rows<R, J, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
    else if (mode == 0) vi_rows<R, J, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
    else if (mode == 1) vi_rows<R, J, 1>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
    else vi_rows<R, J, 2>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
    // rows past row_end have no entries (acc = 0) and are not written
    static_assert(R == 2 || R == 4, "row batches of 2 or 4");
    double s;
    int64_t r;
    bool writer;
    if (R == 4) {
      s = reduce_rows4(acc[0], acc[1], acc[R - 2], acc[R - 1]);
      const int q = lane >> 4;                         // 16-lane row: holds row {0, 2, 1, 3}[q]
      r = r0 + (((q & 1) << 1) | (q >> 1));
      writer = (lane & 15) == 0;
    } else {
      s = reduce_rows2(acc[0], acc[1]);
      r = r0 + (lane >> 5);
      writer = (lane & 31) == 0;
    }
    if (writer && r < row_end) {
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
}

// Class-batched variant (the default for value-indexed matrices).  At upload the rows of
// every block are grouped by their chunk count ceil(len / 64) into batches of 4 rows (a
// 32-byte descriptor per batch: per row its entry offset, length and block-local row id).
// A wave runs the batch through code specialised for that class -- straight-line, no
// guards -- and writes the four sums to the rows the descriptor names.  Rows keep their
// canonical lane assignment and fma order, so the result is unchanged; only the order in
// which a block's rows are visited differs.
constexpr int kVibMaxClass = 6;   // classes 0..6 are specialised, longer rows take the generic path
// The register budget is capped at 64 VGPRs so that 8 waves fit a SIMD (only the rare
// generic paths spill).  TAG only separates the fine operator (0) from multigrid level
// matrices (1) in profiler summaries.
template <int EPI, int TAG = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(8, 8))) void spmv_window_vib_kernel(
    int64_t nrows, int32_t RB, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
    const uint16_t *__restrict__ lcol, const double *__restrict__ val,
    const int32_t *__restrict__ blk_seg_begin, const int32_t *__restrict__ blk_W,
    const int32_t *__restrict__ seg_col, const int32_t *__restrict__ seg_off,
    const double *__restrict__ x, const double *__restrict__ x_halo, int32_t n_local,
    double *__restrict__ y, double alpha, const double *__restrict__ d, double *__restrict__ y2,
    const uint8_t *__restrict__ vidx, const uint16_t *__restrict__ vidw,
    const int32_t *__restrict__ blk_dict_off, const int32_t *__restrict__ blk_dict_n,
    const double *__restrict__ dict, int32_t dict_lds_off,
    const uint64_t *__restrict__ btab, const int32_t *__restrict__ bcnt, int32_t bstride, int xcd_remap) {
  extern __shared__ double xs[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Workgroups with equal blockIdx % 8 share an XCD and its L2.  With 3 B/nnz the x windows are
  // a third of the traffic, and neighbouring row blocks share most of their window: give each
  // XCD a contiguous run of row blocks so that the overlap hits in that XCD's L2.
  int64_t b = blockIdx.x;
  if (xcd_remap) {
    const int64_t nwg = gridDim.x, q = nwg / 8, rm = nwg % 8, xcd = b % 8, idx = b / 8;
    b = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + idx;
  }
  const int32_t W = blk_W[b];
  const double *ds = xs + dict_lds_off;
  int mode = 2;  // block-uniform
  if (W >= 0) {
    const int32_t nd = blk_dict_n[b];
    mode = nd < 0 ? 1 : ((nd & kDictWide) ? 3 : 0);
    const int32_t ndv = nd & 0xffff;
    if (nd >= 0)
      for (int t = threadIdx.x; t < ndv; t += 64 * NW) xs[dict_lds_off + t] = dict[blk_dict_off[b] + t];
    const int32_t s0 = blk_seg_begin[b], s1 = blk_seg_begin[b + 1];
    for (int32_t s = s0 + wave; s < s1; s += NW) {
      const int32_t c0 = seg_col[s], o0 = seg_off[s];
      const int32_t len = ((s + 1 < s1) ? seg_off[s + 1] : W) - o0;
      for (int32_t i = lane; i < len; i += 64) {
        const int32_t c = c0 + i;
        xs[o0 + i] = (c < n_local) ? x[c] : x_halo[c - n_local];
      }
    }
    __syncthreads();
  }
  const int64_t row_begin = b * RB;
  const int64_t k_blk = rp[row_begin];
  const uint16_t *lcol_b = lcol + k_blk;
  const uint8_t *vidx_b = vidx + k_blk;
  const uint16_t *vidw_b = vidw + k_blk;
  const int32_t *col_b = col + k_blk;
  const double *val_b = val + k_blk;
  const int32_t nbatch = bcnt[b];
  // batch descriptor: 4 x uint64 = {entry offset from the block start (32) | entry count (16) |
  // block-local row id, 0xFF = filler (8) | class (8)}; the next batch's descriptor is
  // fetched while the current one is processed (one s_load_dwordx8 each)
  const uint64_t *bt = btab + (b * bstride + wave) * 4;
  uint64_t nx[4] = {0, 0, 0, 0};
  if (wave < nbatch) {
#pragma unroll
    for (int i = 0; i < 4; ++i) nx[i] = bt[i];
  }
  for (int32_t bi = wave; bi < nbatch; bi += NW) {
    uint64_t desc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) desc[i] = nx[i];
    bt += 4 * NW;
    if (bi + NW < nbatch) {
#pragma unroll
      for (int i = 0; i < 4; ++i) nx[i] = bt[i];
    }
    const int cls = (int)(desc[0] >> 56);
    uint32_t ks[4];
    int32_t len[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ks[i] = (uint32_t)desc[i];
      len[i] = (int32_t)((desc[i] >> 32) & 0xffff);
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (mode == 0) {
      switch (cls) {
        case 0: break;
        case 1: vib_rows<1, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 2: vib_rows<2, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 3: vib_rows<3, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 4: vib_rows<4, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 5: vib_rows<5, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 6: vib_rows<6, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        default: vi_rows<4, 2, 0>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
      }
    } else if (mode == 3) {
      switch (cls) {
        case 0: break;
        case 1: vib_rows<1, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 2: vib_rows<2, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 3: vib_rows<3, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 4: vib_rows<4, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 5: vib_rows<5, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        case 6: vib_rows<6, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, xs, ds, acc); break;
        default: vi_rows<4, 2, 3>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
      }
    } else if (mode == 1) {
      vi_rows<4, 2, 1>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
    } else {
      vi_rows<4, 2, 2>(ks, len, lane, lcol_b, vidx_b, vidw_b, col_b, val_b, xs, ds, x, x_halo, n_local, acc);
    }
    const double s = reduce_rows4(acc[0], acc[1], acc[2], acc[3]);
    const int q = lane >> 4;  // 16-lane row q holds the tree of batch row {0, 2, 1, 3}[q]
    const uint64_t dq = q == 0 ? desc[0] : (q == 1 ? desc[2] : (q == 2 ? desc[1] : desc[3]));
    const int id = (int)((dq >> 48) & 0xff);
    if ((lane & 15) == 0 && id != 0xff) {
      const int64_t r = row_begin + id;
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
}

// --------------------------------------------------------------------------
// The same LDS-window scheme for short-row matrices (canonical L = 8, 16 or 32:
// Q1 stencils, pressure mass, multigrid level operators).  An L-lane group plays
// the role of the wave: it owns a batch of R consecutive rows of the block and
// streams their contiguous entries L*U at a time; the un-rotation and the tree
// stay inside the group (width-L shuffles), so the result is again the canonical
// one bit for bit.  Row blocks are larger (RB scales with 64/L) so that a window
// is amortised over about as many entries as in the long-row kernel.
template <int L, int R, int U, int EPI, int TAG = 0>
__global__ __launch_bounds__(kBlock) void spmv_window_group_kernel(
    int64_t nrows, int32_t RB, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
    const uint16_t *__restrict__ lcol, const double *__restrict__ val,
    const int32_t *__restrict__ blk_seg_begin, const int32_t *__restrict__ blk_W,
    const int32_t *__restrict__ seg_col, const int32_t *__restrict__ seg_off,
    const double *__restrict__ x, const double *__restrict__ x_halo, int32_t n_local,
    double *__restrict__ y, double alpha, const double *__restrict__ d, double *__restrict__ y2) {
  extern __shared__ double xs[];
  constexpr int G = kBlock / L;  // groups per workgroup
  const int lane = threadIdx.x % L;
  const int grp = threadIdx.x / L;
  const int wlane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b = blockIdx.x;
  const int32_t W = blk_W[b];
  if (W >= 0) {
    const int32_t s0 = blk_seg_begin[b], s1 = blk_seg_begin[b + 1];
    for (int32_t s = s0 + wave; s < s1; s += 4) {
      const int32_t c0 = seg_col[s], o0 = seg_off[s];
      const int32_t len = ((s + 1 < s1) ? seg_off[s + 1] : W) - o0;
      for (int32_t i = wlane; i < len; i += 64) {
        const int32_t c = c0 + i;
        xs[o0 + i] = (c < n_local) ? x[c] : x_halo[c - n_local];
      }
    }
    __syncthreads();
  }
  const int64_t row_begin = b * RB;
  const int64_t row_end = (row_begin + RB < nrows) ? row_begin + RB : nrows;
  for (int64_t r0 = row_begin + (int64_t)grp * R; r0 < row_end; r0 += G * R) {
    int64_t kb[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) kb[i] = rp[r0 + i < row_end ? r0 + i : row_end];  // group-uniform
    const int64_t k_begin = kb[0];
    const int32_t n_batch = (int32_t)(kb[R] - k_begin);
    int32_t rel[R + 1];
#pragma unroll
    for (int i = 0; i <= R; ++i) rel[i] = (int32_t)(kb[i] - k_begin);
    double acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.0;
    const double *vb = val + k_begin;
    const uint16_t *cl = lcol + k_begin;
    const int32_t *cg = col + k_begin;
    for (int32_t base = 0; base < n_batch; base += L * U) {
      int32_t c[U];
      double v[U], xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t o = base + L * u + lane;
        if (o < n_batch) {
          c[u] = (W >= 0) ? (int32_t)cl[o] : cg[o];
          v[u] = vb[o];
        } else {
          c[u] = -1;
          v[u] = 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xv[u] = 0.0;
        if (c[u] >= 0) xv[u] = (W >= 0) ? xs[c[u]] : ((c[u] < n_local) ? x[c[u]] : x_halo[c[u] - n_local]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int32_t o = base + L * u + lane;
        if (c[u] >= 0) {
#pragma unroll
          for (int i = 0; i < R; ++i)
            if (o >= rel[i] && o < rel[i + 1]) acc[i] = fma(v[u], xv[u], acc[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int64_t r = r0 + i;
      if (r < row_end) {  // group-uniform
        const int dsh = rel[i] & (L - 1);
        double s = __shfl(acc[i], (lane + dsh) & (L - 1), L);
        s = group_reduce<L>(s);
        if (lane == 0) {
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
    }
  }
}

// device scalars -> mapped pinned host mirror (read by the host after a stream sync)
__global__ void mirror_scalars_kernel(const double *__restrict__ sc, double *__restrict__ host, int count) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) host[i] = sc[i];
  __threadfence_system();
}

// --------------------------------------------------------------------------
// Streaming vector kernels. All vectors are padded to a multiple of kChunk with
// zeros, so no bounds checks: one workgroup per chunk, thread t owns the pairs
// base + e*512 + 2t -- the SAME mapping as the canonical dot, which lets the
// fused "update + dot" kernels emit canonical partials.
#define ALFD_FOR_PAIRS(i)                                   \
  const int64_t base__ = (int64_t)blockIdx.x * kChunk;      \
  _Pragma("unroll") for (int e__ = 0; e__ < 8; ++e__)       \
      for (int64_t i = base__ + e__ * 512 + 2 * threadIdx.x, once__ = 1; once__; once__ = 0)

__device__ __forceinline__ double2 ld2(const double *p, int64_t i) {
  return *reinterpret_cast<const double2 *>(p + i);
}
__device__ __forceinline__ void st2(double *p, int64_t i, double2 v) {
  *reinterpret_cast<double2 *>(p + i) = v;
}

// partial[b] = canonical chunk sum of x.y
__global__ __launch_bounds__(kBlock) void dot_partial_kernel(const double *__restrict__ x,
                                                             const double *__restrict__ y,
                                                             double *__restrict__ partial) {
  __shared__ double lds4[4];
  double acc = 0.0;
  ALFD_FOR_PAIRS(i) {
    const double2 a = ld2(x, i), b = ld2(y, i);
    acc = fma(a.x, b.x, acc);
    acc = fma(a.y, b.y, acc);
  }
  const double s = block_reduce_256(acc, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Second stage for `count` dots at once: block j reduces partial[j*stride ..+nb).
// post-ops on the device scalar table `sc` (so PCG needs no host round trip for
// alpha / beta):
//   FIN_STORE  : sc[out+j] = sum
//   FIN_ALPHA  : sc[out] = sum (p.Ap); sc[S_ALPHA] = sc[S_RZ]/sum; sc[S_NALPHA] = -alpha
//   FIN_RZ     : sc[S_RZ_OLD] = sc[S_RZ]; sc[S_RZ] = sum; sc[S_BETA] = sum / old
enum { FIN_STORE = 0, FIN_ALPHA = 1, FIN_RZ = 2 };
enum { S_RZ = 0, S_RZ_OLD = 1, S_PAP = 2, S_ALPHA = 3, S_NALPHA = 4, S_BETA = 5, S_RR = 6, S_TMP = 7, S_H = 8 };
constexpr int S_STAGE = S_H + 2 * kMaxBasis;  // multi-rank local sums before the all-gather
constexpr int kNumScalars = S_H + 3 * kMaxBasis + 8;

__global__ __launch_bounds__(kBlock) void dot_final_kernel(const double *__restrict__ partial, int64_t nb,
                                                           int64_t stride, double *__restrict__ sc,
                                                           int out, int fin) {
  __shared__ double lds4[4];
  const double *p = partial + (int64_t)blockIdx.x * stride;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < nb; i += kBlock) acc = acc + p[i];
  const double s = block_reduce_256(acc, lds4);
  if (threadIdx.x == 0) {
    if (fin == FIN_STORE) {
      sc[out + blockIdx.x] = s;
    } else if (fin == FIN_ALPHA) {
      sc[S_PAP] = s;
      const double a = sc[S_RZ] / s;
      sc[S_ALPHA] = a;
      sc[S_NALPHA] = -a;
    } else {
      const double old = sc[S_RZ];
      sc[S_RZ_OLD] = old;
      sc[S_RZ] = s;
      sc[S_BETA] = s / old;
    }
  }
}

// Multi-rank second stage: sums nranks gathered local results in rank order.
// in[r*count + j] -> sc[out + j]
__global__ void rank_sum_kernel(const double *__restrict__ in, int nranks, int count,
                                double *__restrict__ sc, int out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) {
    double s = in[j];
    for (int r = 1; r < nranks; ++r) s = s + in[(int64_t)r * count + j];
    sc[out + j] = s;
  }
}

// h_j partials for j < count: partial[j*stride + b] = chunk sum of V_j . w
__global__ __launch_bounds__(kBlock) void multi_dot_partial_kernel(const double *__restrict__ V,
                                                                   int64_t vstride, int count,
                                                                   const double *__restrict__ w,
                                                                   double *__restrict__ partial,
                                                                   int64_t pstride) {
  __shared__ double lds[kMaxBasis + 2][4];
  double2 wr[8];
  const int64_t base = (int64_t)blockIdx.x * kChunk;
#pragma unroll
  for (int e = 0; e < 8; ++e) wr[e] = ld2(w, base + e * 512 + 2 * threadIdx.x);
  const int wave = threadIdx.x >> 6;
  for (int j = 0; j < count; ++j) {
    const double *v = V + (int64_t)j * vstride;
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double2 a = ld2(v, base + e * 512 + 2 * threadIdx.x);
      acc = fma(a.x, wr[e].x, acc);
      acc = fma(a.y, wr[e].y, acc);
    }
    acc = group_reduce<64>(acc);
    if ((threadIdx.x & 63) == 0) lds[j][wave] = acc;
  }
  __syncthreads();
  if (threadIdx.x < count)
    partial[(int64_t)threadIdx.x * pstride + blockIdx.x] =
        (lds[threadIdx.x][0] + lds[threadIdx.x][1]) + (lds[threadIdx.x][2] + lds[threadIdx.x][3]);
}

// w = fma(-h_j, V_j, w) for j = 0..count-1 in order (h read from sc[hoff+j])
__global__ __launch_bounds__(kBlock) void multi_axpy_neg_kernel(const double *__restrict__ V,
                                                                int64_t vstride, int count,
                                                                const double *__restrict__ sc, int hoff,
                                                                double *__restrict__ w) {
  ALFD_FOR_PAIRS(i) {
    double2 a = ld2(w, i);
    for (int j = 0; j < count; ++j) {
      const double h = -sc[hoff + j];
      const double2 v = ld2(V + (int64_t)j * vstride, i);
      a.x = fma(h, v.x, a.x);
      a.y = fma(h, v.y, a.y);
    }
    st2(w, i, a);
  }
}

// x = fma(y_j, Z_j, x) for j in order, y on the host side -> passed via sc[hoff+j]
__global__ __launch_bounds__(kBlock) void multi_axpy_kernel(const double *__restrict__ Z, int64_t zstride,
                                                            int count, const double *__restrict__ sc,
                                                            int hoff, double *__restrict__ x) {
  ALFD_FOR_PAIRS(i) {
    double2 a = ld2(x, i);
    for (int j = 0; j < count; ++j) {
      const double h = sc[hoff + j];
      const double2 v = ld2(Z + (int64_t)j * zstride, i);
      a.x = fma(h, v.x, a.x);
      a.y = fma(h, v.y, a.y);
    }
    st2(x, i, a);
  }
}

// y = fma(a, x, y); a = sc[ai] when sc != nullptr else aval
__global__ __launch_bounds__(kBlock) void axpy_kernel(const double *__restrict__ sc, int ai, double aval,
                                                      const double *__restrict__ x, double *__restrict__ y) {
  const double a = sc ? sc[ai] : aval;
  ALFD_FOR_PAIRS(i) {
    const double2 xv = ld2(x, i);
    double2 yv = ld2(y, i);
    yv.x = fma(a, xv.x, yv.x);
    yv.y = fma(a, xv.y, yv.y);
    st2(y, i, yv);
  }
}

// x *= a  (a = 1/sc[ai] when inv, for the Arnoldi normalisation v /= ||v||)
__global__ __launch_bounds__(kBlock) void scale_kernel(const double *__restrict__ sc, int ai, int inv_sqrt,
                                                       double aval, double *__restrict__ x) {
  double a = aval;
  if (sc) a = inv_sqrt ? 1.0 / sqrt(sc[ai]) : sc[ai];
  ALFD_FOR_PAIRS(i) {
    double2 v = ld2(x, i);
    v.x = a * v.x;
    v.y = a * v.y;
    st2(x, i, v);
  }
}

// y = a * x
__global__ __launch_bounds__(kBlock) void scale_copy_kernel(double a, const double *__restrict__ x,
                                                            double *__restrict__ y) {
  ALFD_FOR_PAIRS(i) {
    double2 v = ld2(x, i);
    v.x = a * v.x;
    v.y = a * v.y;
    st2(y, i, v);
  }
}

// v = b - v
__global__ __launch_bounds__(kBlock) void sub_from_kernel(const double *__restrict__ b,
                                                          double *__restrict__ v) {
  ALFD_FOR_PAIRS(i) {
    const double2 bv = ld2(b, i);
    double2 vv = ld2(v, i);
    vv.x = bv.x - vv.x;
    vv.y = bv.y - vv.y;
    st2(v, i, vv);
  }
}

// y = a * (d .* x)          (v2 = -gamma invW u2, ...preconditioner.h:32,66)
__global__ __launch_bounds__(kBlock) void pmul_scale_kernel(double a, const double *__restrict__ d,
                                                            const double *__restrict__ x,
                                                            double *__restrict__ y) {
  ALFD_FOR_PAIRS(i) {
    const double2 dv = ld2(d, i), xv = ld2(x, i);
    double2 yv;
    yv.x = a * (dv.x * xv.x);
    yv.y = a * (dv.y * xv.y);
    st2(y, i, yv);
  }
}

// ---- PCG fused kernels (canonical dot partials come out of the same pass)
// z = d .* r ; partial = chunk sum of r.z            (Jacobi + r.z)
__global__ __launch_bounds__(kBlock) void jacobi_dot_kernel(const double *__restrict__ d,
                                                            const double *__restrict__ r,
                                                            double *__restrict__ z,
                                                            double *__restrict__ partial) {
  __shared__ double lds4[4];
  double acc = 0.0;
  ALFD_FOR_PAIRS(i) {
    const double2 dv = ld2(d, i), rv = ld2(r, i);
    double2 zv;
    zv.x = dv.x * rv.x;
    zv.y = dv.y * rv.y;
    st2(z, i, zv);
    acc = fma(rv.x, zv.x, acc);
    acc = fma(rv.y, zv.y, acc);
  }
  const double s = block_reduce_256(acc, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// p = fma(beta, p, z)  (first: p = z)
__global__ __launch_bounds__(kBlock) void p_update_kernel(const double *__restrict__ sc, int first,
                                                          const double *__restrict__ z,
                                                          double *__restrict__ p) {
  const double beta = first ? 0.0 : sc[S_BETA];
  ALFD_FOR_PAIRS(i) {
    const double2 zv = ld2(z, i);
    if (first) {
      st2(p, i, zv);
    } else {
      double2 pv = ld2(p, i);
      pv.x = fma(beta, pv.x, zv.x);
      pv.y = fma(beta, pv.y, zv.y);
      st2(p, i, pv);
    }
  }
}

// x = fma(alpha, p, x); r = fma(-alpha, Ap, r); partial = chunk sum of r.r
__global__ __launch_bounds__(kBlock) void xr_update_dot_kernel(const double *__restrict__ sc,
                                                               const double *__restrict__ p,
                                                               const double *__restrict__ Ap,
                                                               double *__restrict__ x,
                                                               double *__restrict__ r,
                                                               double *__restrict__ partial) {
  __shared__ double lds4[4];
  const double a = sc[S_ALPHA], na = sc[S_NALPHA];
  double acc = 0.0;
  ALFD_FOR_PAIRS(i) {
    const double2 pv = ld2(p, i), av = ld2(Ap, i);
    double2 xv = ld2(x, i), rv = ld2(r, i);
    xv.x = fma(a, pv.x, xv.x);
    xv.y = fma(a, pv.y, xv.y);
    rv.x = fma(na, av.x, rv.x);
    rv.y = fma(na, av.y, rv.y);
    st2(x, i, xv);
    st2(r, i, rv);
    acc = fma(rv.x, rv.x, acc);
    acc = fma(rv.y, rv.y, acc);
  }
  const double s = block_reduce_256(acc, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ---- Chebyshev (Saad Alg. 12.1, zero start) on D^-1 Aug
// d = inv_theta * (dinv .* r); z = d; res = r
__global__ __launch_bounds__(kBlock) void cheb_init_kernel(double inv_theta, const double *__restrict__ dinv,
                                                           const double *__restrict__ r,
                                                           double *__restrict__ d, double *__restrict__ z,
                                                           double *__restrict__ res, int keep_res) {
  ALFD_FOR_PAIRS(i) {
    const double2 dv = ld2(dinv, i), rv = ld2(r, i);
    double2 o;
    o.x = inv_theta * (dv.x * rv.x);
    o.y = inv_theta * (dv.y * rv.y);
    st2(d, i, o);
    st2(z, i, o);
    if (keep_res) st2(res, i, rv);
  }
}
// res -= tmp; d = fma(c1, d, c2 * (dinv .* res)); z += d
__global__ __launch_bounds__(kBlock) void cheb_step_kernel(double c1, double c2,
                                                           const double *__restrict__ dinv,
                                                           const double *__restrict__ tmp,
                                                           double *__restrict__ res, double *__restrict__ d,
                                                           double *__restrict__ z) {
  ALFD_FOR_PAIRS(i) {
    const double2 dv = ld2(dinv, i), tv = ld2(tmp, i);
    double2 rv = ld2(res, i), dd = ld2(d, i), zv = ld2(z, i);
    rv.x = rv.x - tv.x;
    rv.y = rv.y - tv.y;
    dd.x = fma(c1, dd.x, c2 * (dv.x * rv.x));
    dd.y = fma(c1, dd.y, c2 * (dv.y * rv.y));
    zv.x = zv.x + dd.x;
    zv.y = zv.y + dd.y;
    st2(res, i, rv);
    st2(d, i, dd);
    st2(z, i, zv);
  }
}

// ---- batched lock-step CG (RationalPreconditioner: 21 independent SPD solves on
// the immersed matrices, rational_preconditioner.h:41-56).  All systems advance
// together, one workgroup-chunk per 4096 entries; segment s owns chunks
// [s*cps, (s+1)*cps) and its own scalar row scb[s*kBS ..].  A system that has
// met its stop rule is frozen (active flag 0): its x, r, p are never touched
// again, so every system performs exactly the arithmetic of a stand-alone CG.
constexpr int kBS = 8;  // scalars per system: RZ, RZ_OLD, PAP, ALPHA, NALPHA, BETA, RR, ACTIVE
enum { B_RZ = 0, B_RZ_OLD = 1, B_PAP = 2, B_ALPHA = 3, B_NALPHA = 4, B_BETA = 5, B_RR = 6, B_ACTIVE = 7 };

__global__ __launch_bounds__(kBlock) void b_jacobi_dot_kernel(const double *__restrict__ scb, int cps,
                                                              const double *__restrict__ d,
                                                              const double *__restrict__ r,
                                                              double *__restrict__ z,
                                                              double *__restrict__ partial) {
  __shared__ double lds4[4];
  const int seg = blockIdx.x / cps;
  if (scb[seg * kBS + B_ACTIVE] == 0.0) return;
  double acc = 0.0;
  ALFD_FOR_PAIRS(i) {
    const double2 dv = ld2(d, i), rv = ld2(r, i);
    double2 zv;
    zv.x = dv.x * rv.x;
    zv.y = dv.y * rv.y;
    st2(z, i, zv);
    acc = fma(rv.x, zv.x, acc);
    acc = fma(rv.y, zv.y, acc);
  }
  const double sum = block_reduce_256(acc, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = sum;
}

// one workgroup per system: reduce its cps chunk partials, then the CG scalar update
__global__ __launch_bounds__(kBlock) void b_final_kernel(const double *__restrict__ partial, int cps,
                                                         double *__restrict__ scb, int fin) {
  __shared__ double lds4[4];
  const int seg = blockIdx.x;
  double *sc = scb + seg * kBS;
  if (sc[B_ACTIVE] == 0.0) return;
  const double *p = partial + (int64_t)seg * cps;
  double acc = 0.0;
  for (int i = threadIdx.x; i < cps; i += kBlock) acc = acc + p[i];
  const double s = block_reduce_256(acc, lds4);
  if (threadIdx.x == 0) {
    if (fin == FIN_STORE) {
      sc[B_RR] = s;
    } else if (fin == FIN_ALPHA) {
      sc[B_PAP] = s;
      const double a = sc[B_RZ] / s;
      sc[B_ALPHA] = a;
      sc[B_NALPHA] = -a;
    } else {
      const double old = sc[B_RZ];
      sc[B_RZ_OLD] = old;
      sc[B_RZ] = s;
      sc[B_BETA] = s / old;
    }
  }
}

__global__ __launch_bounds__(kBlock) void b_p_update_kernel(const double *__restrict__ scb, int cps, int first,
                                                            const double *__restrict__ z,
                                                            double *__restrict__ p) {
  const int seg = blockIdx.x / cps;
  if (scb[seg * kBS + B_ACTIVE] == 0.0) return;
  const double beta = first ? 0.0 : scb[seg * kBS + B_BETA];
  ALFD_FOR_PAIRS(i) {
    const double2 zv = ld2(z, i);
    if (first) {
      st2(p, i, zv);
    } else {
      double2 pv = ld2(p, i);
      pv.x = fma(beta, pv.x, zv.x);
      pv.y = fma(beta, pv.y, zv.y);
      st2(p, i, pv);
    }
  }
}

__global__ __launch_bounds__(kBlock) void b_xr_update_dot_kernel(const double *__restrict__ scb, int cps,
                                                                 const double *__restrict__ p,
                                                                 const double *__restrict__ Ap,
                                                                 double *__restrict__ x,
                                                                 double *__restrict__ r,
                                                                 double *__restrict__ partial) {
  __shared__ double lds4[4];
  const int seg = blockIdx.x / cps;
  if (scb[seg * kBS + B_ACTIVE] == 0.0) return;
  const double a = scb[seg * kBS + B_ALPHA], na = scb[seg * kBS + B_NALPHA];
  double acc = 0.0;
  ALFD_FOR_PAIRS(i) {
    const double2 pv = ld2(p, i), av = ld2(Ap, i);
    double2 xv = ld2(x, i), rv = ld2(r, i);
    xv.x = fma(a, pv.x, xv.x);
    xv.y = fma(a, pv.y, xv.y);
    rv.x = fma(na, av.x, rv.x);
    rv.y = fma(na, av.y, rv.y);
    st2(x, i, xv);
    st2(r, i, rv);
    acc = fma(rv.x, rv.x, acc);
    acc = fma(rv.y, rv.y, acc);
  }
  const double s = block_reduce_256(acc, lds4);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// r_s = b for every system s (one padded copy of the rhs per segment)
__global__ __launch_bounds__(kBlock) void b_replicate_kernel(int cps, const double *__restrict__ b,
                                                             double *__restrict__ r) {
  const int64_t lbase = (int64_t)(blockIdx.x % cps) * kChunk;
  const int64_t gbase = (int64_t)blockIdx.x * kChunk;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t o = e * 512 + 2 * threadIdx.x;
    st2(r, gbase + o, ld2(b, lbase + o));
  }
}

// out = sum_s coef[s] * X_s, accumulated s = 0..nseg-1 as out = out + coef*x
// (rational_preconditioner.h:59-62)
__global__ __launch_bounds__(kBlock) void b_combine_kernel(int nseg, int64_t seglen,
                                                           const double *__restrict__ coef,
                                                           const double *__restrict__ X,
                                                           double *__restrict__ out) {
  ALFD_FOR_PAIRS(i) {
    double2 a;
    a.x = 0.0;
    a.y = 0.0;
    for (int s = 0; s < nseg; ++s) {
      const double c = coef[s];
      const double2 v = ld2(X + (int64_t)s * seglen, i);
      a.x = a.x + c * v.x;
      a.y = a.y + c * v.y;
    }
    st2(out, i, a);
  }
}

// dinv[i] = 1 / dA[i] (rows < n)
__global__ void inv_diag_kernel(int64_t n, const double *__restrict__ dA, double *__restrict__ dinv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dinv[i] = 1.0 / dA[i];
}

// ---- setup kernels
// dA[r] = A_rr (exact copy, no arithmetic); diag column = r + row_offset, and
// locally that is column (r) when the matrix's local columns come first.
template <int L>
__global__ __launch_bounds__(kBlock) void extract_diag_kernel(int64_t nrows, const int64_t *__restrict__ rp,
                                                              const int32_t *__restrict__ col,
                                                              const double *__restrict__ val,
                                                              double *__restrict__ dA) {
  constexpr int RPB = kBlock / L;
  const int lane = threadIdx.x % L, sub = threadIdx.x / L;
  const int64_t ngroups = (nrows + RPB - 1) / RPB;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t r = g * RPB + sub;
    if (r < nrows)
      for (int64_t k = rp[r] + lane; k < rp[r + 1]; k += L)
        if (col[k] == (int32_t)r) dA[r] = val[k];
  }
}
// dinv[i] = 1 / fma(gamma, s_i, dA[i]), s_i = sequential fma over row i of Ct of
// (w[c] * v) * v ; one thread per non-empty row of Ct; other rows: 1/dA.
__global__ void aug_diag_rows_kernel(int64_t n_sr, const int64_t *__restrict__ rp,
                                     const int32_t *__restrict__ col, const double *__restrict__ val,
                                     const int32_t *__restrict__ rows, const double *__restrict__ w,
                                     const double *__restrict__ w_halo, int32_t n_local,
                                     double *__restrict__ s_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_sr) {
    double s = 0.0;
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
      const int32_t c = col[k];
      const double wv = c < n_local ? w[c] : w_halo[c - n_local];
      s = fma(wv * val[k], val[k], s);
    }
    s_out[rows ? rows[r] : r] = s;
  }
}
__global__ void aug_diag_finish_kernel(int64_t n, double gamma, const double *__restrict__ dA,
                                       const double *__restrict__ s, double *__restrict__ dinv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dinv[i] = 1.0 / fma(gamma, s[i], dA[i]);
}
// power-iteration start vector: v_i = 1 + ((g*2654435761) & 1023)/1024, g = global index
__global__ void hash_vector_kernel(int64_t n, int64_t goff, double *__restrict__ v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    v[i] = 1.0 + (double)(((uint64_t)(i + goff) * 2654435761ull) & 1023ull) * (1.0 / 1024.0);
}

// gather x[idx[i]] -> out[i]   (halo send packing)
__global__ void gather_kernel(int64_t n, const int32_t *__restrict__ idx, const double *__restrict__ x,
                              double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[idx[i]];
}

// out[idx[i]] += x[i]   (interface-patch corrections: idx lists distinct rows)
__global__ void scatter_add_kernel(int64_t n, const int32_t *__restrict__ idx, const double *__restrict__ x,
                                   double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[idx[i]] = out[idx[i]] + x[i];
}

// ---------------------------------------------------------------------------------------------------
// Sparse product C = A * P on the device, one 64-lane workgroup per output row (the Galerkin products of the multigrid
// setup: A P, then P^T (A P), and C P).  Deterministic and equal bit for bit to the host / oracle definition: every
// output entry (i, J) is ONE sequential fma chain over the entries k of row i of A in CSR order and the entries of row
// col_k of P in CSR order, acc = fma(a_ik, p_kJ, acc) from 0; the row comes out sorted by column.
//   pass 0 (symbolic): counts[i] = number of distinct columns of row i (LDS hash set of the candidates)
//   pass 1 (numeric):  the same set, compacted and sorted in LDS (bitonic), then lane t owns output column t and walks
//                      the row of A once, looking its column up in every P row it meets (wave-uniform control flow,
//                      broadcast loads).
// kSgTable slots per hash set, at most kSgMaxOut distinct columns per row (overflow[0] is set otherwise: the host falls
// back to its own product).
constexpr int kSgTable = 4096;
constexpr int kSgMaxOut = 1024;
__global__ __launch_bounds__(64) void spgemm_rows_kernel(int64_t nrows, const int64_t *__restrict__ arp,
                                                         const int32_t *__restrict__ acol, const double *__restrict__ aval,
                                                         const int64_t *__restrict__ prp, const int32_t *__restrict__ pcol,
                                                         const double *__restrict__ pval, int pass,
                                                         int32_t *__restrict__ counts, const int64_t *__restrict__ crp,
                                                         int32_t *__restrict__ ccol, double *__restrict__ cval,
                                                         int32_t *__restrict__ overflow) {
  __shared__ int32_t table[kSgTable];
  __shared__ int32_t keys[kSgMaxOut];
  __shared__ int32_t cnt;
  const int lane = threadIdx.x;
  for (int64_t i = blockIdx.x; i < nrows; i += gridDim.x) {
    const int64_t k0 = arp[i], k1 = arp[i + 1];
    if (k0 == k1) {
      if (pass == 0 && lane == 0) counts[i] = 0;
      continue;
    }
    for (int s = lane; s < kSgTable; s += 64) table[s] = -1;
    if (lane == 0) cnt = 0;
    __syncthreads();
    for (int64_t k = k0 + lane; k < k1; k += 64) {
      const int64_t j = acol[k];
      for (int64_t e = prp[j]; e < prp[j + 1]; ++e) {
        const int32_t J = pcol[e];
        uint32_t h = ((uint32_t)J * 2654435761u) >> 20;   // 12 bits
        for (;;) {
          const int32_t old = atomicCAS(&table[h], -1, J);
          if (old == -1) {
            atomicAdd(&cnt, 1);
            break;
          }
          if (old == J) break;
          h = (h + 1) & (kSgTable - 1);
          if (*(volatile int32_t *)&cnt > kSgMaxOut) break;   // overflowing row: stop probing (reported below)
        }
      }
    }
    __syncthreads();
    const int n = cnt;
    if (n > kSgMaxOut) {
      if (lane == 0) overflow[0] = 1;
      if (pass == 0 && lane == 0) counts[i] = 0;
      __syncthreads();
      continue;
    }
    if (pass == 0) {
      if (lane == 0) counts[i] = n;
      __syncthreads();
      continue;
    }
    // compact, pad to a power of two with INT_MAX, bitonic sort
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    __syncthreads();
    if (lane == 0) cnt = 0;
    __syncthreads();
    for (int s = lane; s < kSgTable; s += 64) {
      const int32_t J = table[s];
      if (J != -1) keys[atomicAdd(&cnt, 1)] = J;
    }
    __syncthreads();
    for (int s = n + lane; s < n2; s += 64) keys[s] = 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = lane; t < n2 / 2; t += 64) {
          const int lo = 2 * t - (t & (stride - 1));
          const int hi = lo + stride;
          const bool up = (lo & size) == 0;
          const int32_t a = keys[lo], b = keys[hi];
          if ((a > b) == up) {
            keys[lo] = b;
            keys[hi] = a;
          }
        }
        __syncthreads();
      }
    // numeric: lane t owns output column keys[t]
    const int64_t c0 = crp[i];
    for (int t0 = 0; t0 < n; t0 += 64) {
      const int t = t0 + lane;
      const int32_t J = t < n ? keys[t] : -2;
      double acc = 0.0;
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t j = acol[k];
        const double a = aval[k];
        const int64_t e1 = prp[j + 1];
        for (int64_t e = prp[j]; e < e1; ++e)
          if (pcol[e] == J) acc = fma(a, pval[e], acc);
      }
      if (t < n) {
        ccol[c0 + t] = J;
        cval[c0 + t] = acc;
      }
    }
    __syncthreads();
  }
}

// flag[i] = 1 if row i of the CSR matrix has a column j with mark[j] >= 0 (rows of A that reach the interface patch)
__global__ void rows_touching_kernel(int64_t nrows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                     const int32_t *__restrict__ mark, uint8_t *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (i >= nrows) return;
  const int lane = threadIdx.x & 63;
  int hit = 0;
  for (int64_t k = rp[i] + lane; k < rp[i + 1]; k += 64) hit |= mark[col[k]] >= 0;
  hit = __any(hit);
  if (lane == 0) flag[i] = (uint8_t)(hit ? 1 : 0);
}

// copies the listed rows of a CSR matrix into a compact CSR whose row pointer orp the host computed
__global__ void gather_rows_kernel(int64_t nlist, const int32_t *__restrict__ rows, const int64_t *__restrict__ rp,
                                   const int32_t *__restrict__ col, const double *__restrict__ val,
                                   const int64_t *__restrict__ orp, int32_t *__restrict__ ocol, double *__restrict__ oval) {
  const int64_t q = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  if (q >= nlist) return;
  const int lane = threadIdx.x & 63;
  const int64_t k0 = rp[rows[q]], len = rp[rows[q] + 1] - k0, o0 = orp[q];
  for (int64_t k = lane; k < len; k += 64) {
    ocol[o0 + k] = col[k0 + k];
    oval[o0 + k] = val[k0 + k];
  }
}

}  // namespace alfd
