// alfd.hip -- host side of libalfd.so: context, HBM-resident operators, the
// inner PCG, the AL preconditioner vmult()s, FGMRES, and the C ABI of
// include/alfd/alfd.h.  All arithmetic runs in the kernels of kernels.hpp; the
// host only sequences launches and does the O(m^2) Hessenberg/Givens algebra.
//
// Reference objects replaced (file:line in /root/reference):
//   * Aug = A + gamma Ct invW C                stokes_immersed_boundary.cc:991-993,
//                                              immersed_laplace.cc:880-884
//   * Aug_inv = inverse_operator(Aug, CG, prec) stokes...:1020-1045, immersed_laplace.cc:907-916
//   * Mp_inv  = inverse_operator(Mp, CG(100,1e-6), lumped diag)   stokes...:931-957
//   * BlockPreconditionerAugmentedLagrangian{,Stokes,Diagonal}::vmult
//                                              augmented_lagrangian_preconditioner.h:28-34,62-70,95-103
//   * AA = block_operator<..>                  stokes...:1000-1003, immersed_laplace.cc:891-892
//   * SolverFGMRES<BlockVector<double>>::solve stokes...:1067-1074 ([EXT] deal.II 9.6 flavour)
//   * SolverControl / ReductionControl / IterationNumberControl stop rules [EXT]
// There is NO CPU fallback: without a HIP device every entry point fails.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <climits>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "alfd/alfd.h"
#include "kernels.hpp"
#include "kernels_vs.hpp"

namespace alfd {


// ------------------------------------------------------------------ helpers
#define HIPC(call)                                                                          \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if (e__ != hipSuccess) {                                                                \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + ":" + \
                 std::to_string(__LINE__) + ")";                                            \
      return ALFD_E_HIP;                                                                    \
    }                                                                                       \
  } while (0)
#define RC(call)                  \
  do {                            \
    int rc__ = (call);            \
    if (rc__ != ALFD_OK) return rc__; \
  } while (0)

static inline int64_t pad_chunk(int64_t n) { return (n + kChunk - 1) / kChunk * kChunk; }

struct DevCsr {
  bool present = false;
  int64_t nrows = 0, ncols = 0, nnz = 0;  // local rows, global cols
  int64_t n_list = 0;                     // rows the kernel iterates over (== nrows unless sparse)
  bool sparse = false;
  bool rep = false;  // replicated on every rank (coarse multigrid levels): never exchanges a halo
  bool derived = false;  // C / B built by the library as the transpose of an uploaded CT / BT
  int L = 64;
  int32_t n_local_cols = 0;               // columns < n_local_cols read x, others the halo
  int64_t *rp = nullptr;
  int32_t *col = nullptr;
  double *val = nullptr;
  int32_t *rows = nullptr;
  // multi-rank halo plan
  std::vector<int64_t> send_off, recv_off;  // per peer prefix (size nranks+1)
  std::vector<int32_t> halo_globals;        // global column id of every halo entry (host)
  int32_t *send_idx = nullptr;              // local indices to pack
  double *send_buf = nullptr, *halo = nullptr;
  int64_t n_halo = 0;
  std::vector<void *> owned;  // device arrays of this matrix (released on re-upload / destroy)
  // LDS-window format (long-row matrices): see spmv_window_kernel
  bool win = false;
  bool win_deferred = false, win_user = true;   // window plan not built yet (the batch-major form holds the matrix)
  int tag = 0;  // 1 = multigrid level matrix (separate kernel instantiation for profiling)
  int32_t win_RB = 0, win_maxW = 0;
  int64_t win_nblocks = 0, win_fallback_blocks = 0, win_nseg = 0;
  uint16_t *lcol = nullptr;
  // value-indexed blocks (see spmv_window_kernel): 8-bit dictionary index per entry
  bool vi = false;
  uint8_t *vidx = nullptr;
  int32_t *blk_dict_off = nullptr, *blk_dict_n = nullptr;
  double *dict = nullptr;
  int64_t vi_blocks = 0, vi_nnz = 0, vi_dict_total = 0, vi_wide_nnz = 0;
  uint16_t *vidw = nullptr;  // 16-bit value codes of the blocks with 257..512 distinct values
  uint64_t *vib_tab = nullptr;  // class-sorted row batches per block (spmv_window_vib_kernel)
  int32_t *vib_cnt = nullptr;
  int32_t vib_stride = 0;
  int32_t *blk_seg_begin = nullptr, *blk_W = nullptr, *seg_col = nullptr, *seg_off = nullptr;
  // batch-major format (spmv_vs_kernel, kernels_vs.hpp)
  struct Vs {
    bool on = false, bricks = false;
    int64_t nb = 0, nseg = 0, stream_bytes = 0, nbatch = 0, dict_total = 0, shared_nnz = 0;
    int32_t stride = 0, maxW = 0;
    int32_t L = 64, tab_u64 = 16;  // lanes per row of the format; descriptor words per batch
    int32_t wide = 0;              // 10-bit codes / 11-bit window columns (VsFmt<1>)
    uint8_t *stream = nullptr;
    int64_t *sb = nullptr;
    uint64_t *tab = nullptr;
    int32_t *cnt = nullptr, *seg_begin = nullptr, *blkW = nullptr, *seg_col = nullptr,
            *seg_off = nullptr, *doff = nullptr, *dn = nullptr;
    // long rows (spmv_vs_kernel): everything a block needs first is addressable from its index alone -- an 8-dword
    // header {batches, window slots, segments, dictionary size, dictionary offset, stream offset lo / hi, 0} and the
    // segment table with a fixed stride per block ((column, slot) pairs) -- so that the x window is requested after ONE
    // dependent round trip instead of two (header -> segment table -> x)
    int32_t *hdrb = nullptr, *segx = nullptr;
    int32_t seg_stride = 0;
    int64_t nb_interior = 0;     // partitioned contexts: the first nb_interior blocks read no halo column
    double *dict = nullptr;
  } vs;
  // bytes the kernel in use moves per launch (format bytes, x read once)
  double streamed_bytes(bool use_vi, bool use_vs = false) const {
    const double vec = (double)(n_list + 1) * 8.0 + (double)n_list * 8.0 + (double)(sparse ? n_list : ncols) * 8.0;
    if (vs.on && use_vs && use_vi)
      return (double)vs.stream_bytes + 8.0 * vs.tab_u64 * (double)vs.nbatch + 28.0 * (double)vs.nb +
             8.0 * (double)vs.nseg + 8.0 * (double)vs.dict_total + 8.0 * (double)nrows +
             8.0 * (double)(sparse ? n_list : ncols);
    if (!win) return (double)nnz * 12.0 + vec;
    const double meta = (double)win_nblocks * 8.0 + (double)win_nseg * 8.0;
    if (!(vi && use_vi)) return (double)nnz * 10.0 + vec + meta;
    return (double)(vi_nnz - vi_wide_nnz) * 3.0 + (double)vi_wide_nnz * 4.0 + (double)(nnz - vi_nnz) * 10.0 + vec + meta + (double)vi_dict_total * 8.0 +
           (double)win_nblocks * (8.0 + (vib_tab ? 32.0 * vib_stride + 4.0 : 0.0));
  }
  double algorithmic_bytes() const {
    // nnz*(8+4) + (nrows+1)*8 + nrows*8 + ncols*8  (x read once)
    return (double)nnz * 12.0 + (double)(n_list + 1) * 8.0 + (double)n_list * 8.0 +
           (double)(sparse ? n_list : ncols) * 8.0;
  }
};

struct HostCsr {
  int64_t nrows = 0, ncols = 0;
  std::vector<int64_t> rp;
  std::vector<int32_t> col;
  std::vector<double> val;
  int64_t nnz() const { return rp.empty() ? 0 : rp.back(); }
};

constexpr int kScratchSlot = ALFD_NSLOTS;  // internal upload target (level matrices, batched systems)

// One level of the aggregation multigrid hierarchy (level 0 = the augmented block itself).
struct MlLevel {
  DevCsr A, C, Ct;   // operator pieces of this level (levels >= 1; level 0 uses the slots)
  DevCsr P, R;       // prolongation to this level from the next coarser one, and R = P^T
  int64_t n = 0, npad = 0;
  double *dinv = nullptr;
  double lmax = 0;
  double *r = nullptr, *z = nullptr, *t = nullptr, *cd = nullptr, *cres = nullptr, *ctmp = nullptr;
  // multi-rank: levels from alfd_ctx::ml_rep_level on are REPLICATED on every rank (global operators and
  // vectors, no halo exchanges); gP / gR connect two replicated levels, P / R above stay rank-local
  DevCsr gA, gC, gCt, gP, gR;
  bool P_global_cols = false;   // P of this level addresses the replicated coarse vector by GLOBAL ids (CSR prolongators)
  int64_t gn = 0, gnpad = 0, g_maxpiece = 0;
  std::vector<int64_t> g_offs;  // rank offsets of this level's unknowns (size nranks + 1)
  double *gdinv = nullptr, *gr = nullptr, *gz = nullptr, *gt = nullptr, *gcd = nullptr, *gcres = nullptr,
         *gctmp = nullptr, *g_send = nullptr, *g_stage = nullptr;
};

enum State { ITERATE = 0, SUCCESS = 1, FAILURE = 2 };
struct Control {
  alfd_control c;
  double initial = 0, reduced_tol = 0, last_value = 0;
  int last_step = 0;
  State check(int step, double v) {
    last_step = step;
    last_value = v;
    if (step == 0) {
      initial = v;
      reduced_tol = v * c.reduce;
    }
    if (c.kind == ALFD_CTRL_REDUCTION && v < reduced_tol) return SUCCESS;
    if (c.kind == ALFD_CTRL_FIXED_ITERS && step >= c.max_steps) return SUCCESS;
    if (v <= c.tol) return SUCCESS;
    if (step >= c.max_steps || std::isnan(v)) return FAILURE;
    return ITERATE;
  }
};

struct TimedLaunch {
  int cls;
  hipEvent_t a, b;
};

}  // namespace alfd

using namespace alfd;

struct alfd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t xstream = nullptr;                  // halo exchanges that run beside the interior row blocks of an SpMV
  hipEvent_t ev_x = nullptr, ev_halo = nullptr;
  int overlap_halo = -1;                          // ALFD_SPMV_OVERLAP_HALO: 1 on, 0 off, -1 (default): on for the in-process and
                                                  // host transports, off over RCCL until that path has run on hardware once
                                                  // (one communicator driven from two streams)
  std::string err;
  // partition / comm
  int rank = 0, nranks = 1;
  ncclComm_t nccl = nullptr;
  struct alfd_local_group *local = nullptr;  // in-process rank group (single-GPU emulation of N ranks)
  alfd_host_allgather_fn host_allgather = nullptr;   // host-transport rank group (MPI / gloo launchers)
  alfd_host_alltoallv_fn host_alltoallv = nullptr;
  void *host_user = nullptr;
  std::vector<char> host_send, host_recv;
  std::vector<std::vector<int64_t>> part;  // [block][nranks+1] global offsets
  // layout
  int nblocks = 0;
  int64_t n[ALFD_MAX_BLOCKS] = {0, 0, 0};        // local block lengths
  int64_t off[ALFD_MAX_BLOCKS + 1] = {0, 0, 0, 0};  // padded local offsets
  int64_t nmax = 0;                               // max padded block length
  int64_t diag_cap = 0;                           // length of the dA / s_aug scratch vectors
  DevCsr mat[ALFD_NSLOTS + 1];
  double *diag[ALFD_NDIAGS] = {nullptr, nullptr};
  int64_t diag_n[ALFD_NDIAGS] = {0, 0};
  alfd_config cfg;
  bool configured = false, is_setup = false;
  // device workspace
  double *sc = nullptr;        // device scalar table
  double *sc_host = nullptr;   // pinned mirror
  double *partial = nullptr;   // dot partials [(kMaxBasis+2) * pstride]
  int64_t pstride = 0;
  double *gather = nullptr;    // multi-rank scalar all-gather buffer
  double *dinv_aug = nullptr, *dA = nullptr, *s_aug = nullptr;
  double *dinv_a22 = nullptr, *dinv_aug2 = nullptr;   // elliptic: 1/diag(A22_aug), [dinv_aug | dinv_a22]
  double lam_max[7] = {0, 0, 0, 0, 0, 0, 0};              // per inner operator kind
  double *dinv_k = nullptr;                            // rational: 1/diag(K)
  // exact W^-1 (alfd_config::w_inverse != 0): Jacobi CG on M nested inside the operator the
  // outer inner-CG runs on, so it owns a second set of CG vectors / scalar tables
  double *dinv_m = nullptr, *m_tmp = nullptr, *m_tmp2 = nullptr;
  double *n_r = nullptr, *n_z = nullptr, *n_p = nullptr, *n_Ap = nullptr;
  double *n_sc = nullptr, *n_sc_host = nullptr, *n_partial = nullptr, *n_gather = nullptr;
  int64_t mass_its = 0;
  // RationalPreconditioner state (batched CG over the 21 immersed systems)
  HostCsr h_M, h_K;                                    // host copies of the (tiny) immersed matrices
  // multilevel inner preconditioner
  std::vector<int32_t> ml_agg[ALFD_MAX_LEVELS];
  std::vector<double> ml_wgt[ALFD_MAX_LEVELS];
  int64_t ml_ncoarse[ALFD_MAX_LEVELS] = {0, 0, 0, 0, 0, 0, 0, 0};
  HostCsr ml_P[ALFD_MAX_LEVELS];               // CSR prolongator of a level (alfd_set_prolongator), replaces its aggregates
  DevCsr ml_inv;                               // explicit inverse of the coarsest operator (alfd_config::ml_coarse_direct)
  // interface patch (alfd_config::ml_patch_degree > 0): S = non-empty rows of Ct, vectors of length |S|
  struct Patch {
    bool on = false;
    int64_t m = 0, mpad = 0;
    DevCsr Ass, As, Ats, Cs, Cts;              // A[S,S], A[S,:], A[:,S] (sparse rows), C[:,S], Ct[S,:]
    int32_t *S = nullptr;
    double *dinv = nullptr, *rS = nullptr, *zS = nullptr, *uS = nullptr, *eS = nullptr, *cd = nullptr,
           *cres = nullptr, *ctmp = nullptr, *rr = nullptr;
    double lmax = 0;
    // partitioned context: the patch is REPLICATED (Ass, Cs, Cts whole on every rank); S[] = this rank's rows
    // (m_loc of them, patch ids soff[rank] ..), Cts_loc / Ctg = Ct[S_loc, :] / Ct[owned rows, :] over GLOBAL
    // multiplier ids, As / Ats = this rank's rows of A[S, :] (halo on the fine vector) / A[:, S]
    bool rep = false;
    int64_t m_loc = 0, piece = 1, lam_piece = 1;
    std::vector<int64_t> soff;
    DevCsr Cts_loc, Ctg;
    double *send = nullptr, *stage = nullptr, *lam_send = nullptr, *lam_stage = nullptr, *loc = nullptr;
  } patch;
  std::vector<MlLevel> ml;
  int ml_rep_level = -1;                      // first replicated level (multi-rank), -1: none
  bool dots_replicated = false;               // reductions over REPLICATED vectors (every rank holds the whole vector): no exchange
  int64_t ml_rep_threshold = 300000;          // replicate levels with at most this many unknowns (ALFD_ML_REPLICATE)
  int ml_gpu_galerkin = 1;                    // Galerkin products of CSR-prolongator levels on the device (ALFD_ML_GPU_GALERKIN)
  double *g_w = nullptr, *g_tlam = nullptr;   // global W^-1 diagonal and multiplier work vector of the replicated levels
  std::vector<int64_t> ml_coff[ALFD_MAX_LEVELS];       // rank offsets of the coarse dofs of each level
  const int64_t *up_col_offsets = nullptr;             // upload_matrix overrides (level matrices)
  bool up_local_only = false;
  DevCsr rat_mat;                                      // block-diagonal [S_1 .. S_20, M]
  double *rt_r = nullptr, *rt_z = nullptr, *rt_p = nullptr, *rt_Ap = nullptr, *rt_x = nullptr;
  double *rt_dinv = nullptr, *rt_partial = nullptr, *rt_scb = nullptr, *rt_coef = nullptr;
  double *rt_scb_host = nullptr;
  int64_t rational_its = 0;
  int64_t wmax = 0;                                    // length of the inner-solve work vectors
  std::vector<void *> ws_allocs;                       // workspace of the current setup()
  double *w_r = nullptr, *w_z = nullptr, *w_p = nullptr, *w_Ap = nullptr;  // PCG
  double *c_d = nullptr, *c_res = nullptr, *c_tmp = nullptr;               // Chebyshev
  double *t_lam = nullptr;                                                 // invW .* (C x)
  double *q_tmp = nullptr, *rhs_tmp = nullptr;                             // precond scratch
  double *V = nullptr, *Z = nullptr, *xb = nullptr, *bb = nullptr, *io = nullptr;
  double *st_in = nullptr, *st_out = nullptr;   // staging of the depth-1 calls (the resident rhs / guess stay intact)
  double lambda_max = 0, lambda_min = 0;
  // stats of the current solve
  int64_t inner_its = 0, mp_its = 0;
  int inner_failures = 0, precond_applications = 0;
  std::vector<double> history;
  // timing
  int timing = 0;  // 0 off, 1 = A-SpMV launches only (cheap), 2 = every kernel class
  std::vector<TimedLaunch> timed;
  double t_ms[ALFD_T_NCLASSES] = {0, 0, 0, 0};
  int64_t t_launches[ALFD_T_NCLASSES] = {0, 0, 0, 0};
  double t_bytes[ALFD_T_NCLASSES] = {0, 0, 0, 0};
  double t_fbytes[ALFD_T_NCLASSES] = {0, 0, 0, 0};   // the same launches priced by format bytes
  double setup_s[ALFD_SETUP_NPHASES] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool upload_since_setup = false;
  std::vector<void *> allocs;
  int spmv_stream_R = 2, spmv_stream_U = 8, spmv_nt = 0, spmv_grid_mult = 8;  // tunables (env ALFD_SPMV_*)
  int spmv_group_R = 4, spmv_group_U = 4;  // batch shape of the short-row window kernel
  bool vi_off = false;                      // alfd_bench_spmv_format: time the plain 10 B/nnz kernel on a value-indexed matrix
  int vi_rows_R = 4, vi_rows_J = 2;         // row-batched VI kernel shape (ALFD_SPMV_VI_R=0: stream-ordered VI kernel)
  int vi_xcd = 0;                           // XCD-contiguous row-block order in the class-batched kernel
  int vi_levels = 1;                        // also dictionary-code multigrid level matrices
                        // waves per workgroup of the class-batched kernel (4 or 8)
  int vi_batched = 1;                       // class-batched VI kernel (ALFD_SPMV_VI_BATCHED=0: in-order batches)
  int win_RB_vi = 96;                       // row block of value-indexed matrices (ALFD_SPMV_WINDOW_RB_VI)
  int win_vi = 1;                           // dictionary-coded values in window blocks (ALFD_SPMV_VALUE_INDEX)
  int win_short_scale = 2;                  // short-row block = min(512, win_RB * scale * 64 / L) rows; 0 = off
  int vs_enable = 1, vs_NW = 4, vs_RB = 96, vs_xcd = 0, vs_share = 1, vs_wide = 1;
  int vs_lds_base_ok = -1;                  // -1 unknown; the absolute LDS addressing of kernels_vs.hpp needs base 0   // batch-major format (alfd_set_tunable "batch_major")
  std::vector<int64_t> rb_ptr[ALFD_NSLOTS + 1];   // row-block hint per slot (alfd_set_row_blocks)
  std::vector<int32_t> rb_rows[ALFD_NSLOTS + 1];
  int win_short_min_blocks = 256;           // the same for short-row matrices (ALFD_SPMV_WINDOW_SHORT_MIN_BLOCKS; 64 costs cfg 3 30 %, 1024 leaves its 262 k-row level operator out)
  int win_min_blocks = 256;                 // long-row window formats need this many row blocks (ALFD_SPMV_WINDOW_MIN_BLOCKS; 1 per CU: the level-2 operator of the bench, 152 k rows, gains 2 % of the solve, the 36.7 k-row patch matrix another 1 %; 128: slower)
  int win_enable = 1, win_RB = 96, win_maxW = 4096, win_gap = 8, win_xcd = 0;  // win_xcd: XCD-contiguous block order (measured neutral on MI355X)
  int64_t ntot() const { return off[nblocks]; }
};

// ---------------------------------------------------------------------------
// In-process rank group: N contexts in ONE process (one host thread per rank)
// exchange through device-to-device copies and a host barrier instead of RCCL.
// It exists so that everything of the multi-rank path except the literal RCCL
// calls (halo plans, pack kernels, halo SpMV, ordered reductions, setup
// protocol) can be executed on a single-GPU box; one process per GPU over
// RCCL remains the production configuration.
struct alfd_local_group {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::vector<const void *> buf;            // per rank: published buffer
  std::vector<const int64_t *> off;         // per rank: published prefix (alltoallv)
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long g = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != g; });
    }
  }
};

namespace alfd {

// all-gather of `bytes` per rank, device buffers, result ordered by rank
static int comm_allgather(alfd_ctx *ctx, const void *send, void *recv, size_t bytes) {
  if (ctx->local) {
    alfd_local_group *g = ctx->local;
    HIPC(hipStreamSynchronize(ctx->stream));
    g->buf[ctx->rank] = send;
    g->barrier();
    for (int p = 0; p < g->n; ++p)
      HIPC(hipMemcpyAsync((char *)recv + (size_t)p * bytes, g->buf[p], bytes, hipMemcpyDeviceToDevice,
                          ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    g->barrier();
    return ALFD_OK;
  }
  if (ctx->host_allgather) {
    ctx->host_send.resize(bytes);
    ctx->host_recv.resize(bytes * (size_t)ctx->nranks);
    HIPC(hipMemcpyAsync(ctx->host_send.data(), send, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    if (ctx->host_allgather(ctx->host_user, ctx->host_send.data(), ctx->host_recv.data(), bytes) != 0)
      return ctx->err = "host all-gather callback failed", ALFD_E_COMM;
    HIPC(hipMemcpyAsync(recv, ctx->host_recv.data(), ctx->host_recv.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));   // the staging buffer is reused by the next collective
    return ALFD_OK;
  }
  if (ncclAllGather(send, recv, bytes, ncclChar, ctx->nccl, ctx->stream) != ncclSuccess)
    return ctx->err = "ncclAllGather failed", ALFD_E_COMM;
  return ALFD_OK;
}

// personalised exchange: rank r sends sendbuf[send_off[p] .. send_off[p+1]) to p and
// receives recvbuf[recv_off[p] .. recv_off[p+1]) from p (element size `es` bytes)
static int comm_alltoallv(alfd_ctx *ctx, const void *sendbuf, const int64_t *send_off, void *recvbuf,
                          const int64_t *recv_off, size_t es, hipStream_t st = nullptr) {
  if (!st) st = ctx->stream;
  if (ctx->local) {
    alfd_local_group *g = ctx->local;
    HIPC(hipStreamSynchronize(st));
    g->buf[ctx->rank] = sendbuf;
    g->off[ctx->rank] = send_off;
    g->barrier();
    for (int p = 0; p < g->n; ++p) {
      const int64_t nr = recv_off[p + 1] - recv_off[p];
      if (nr > 0)  // what p sends to me starts at p's send_off[my rank]
        HIPC(hipMemcpyAsync((char *)recvbuf + (size_t)recv_off[p] * es,
                            (const char *)g->buf[p] + (size_t)g->off[p][ctx->rank] * es, (size_t)nr * es,
                            hipMemcpyDeviceToDevice, st));
    }
    HIPC(hipStreamSynchronize(st));
    g->barrier();
    return ALFD_OK;
  }
  if (ctx->host_alltoallv) {
    const size_t ns = (size_t)send_off[ctx->nranks] * es, nr = (size_t)recv_off[ctx->nranks] * es;
    ctx->host_send.resize(std::max<size_t>(ns, 1));
    ctx->host_recv.resize(std::max<size_t>(nr, 1));
    if (ns) HIPC(hipMemcpyAsync(ctx->host_send.data(), sendbuf, ns, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (ctx->host_alltoallv(ctx->host_user, ctx->host_send.data(), send_off, ctx->host_recv.data(), recv_off, es) != 0)
      return ctx->err = "host all-to-all callback failed", ALFD_E_COMM;
    if (nr) HIPC(hipMemcpyAsync(recvbuf, ctx->host_recv.data(), nr, hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));
    return ALFD_OK;
  }
  if (ncclGroupStart() != ncclSuccess) return ctx->err = "ncclGroupStart", ALFD_E_COMM;
  const char *failed = nullptr;  // the group is closed on every path: an open group would strand the peers
  for (int p = 0; p < ctx->nranks && !failed; ++p) {
    const int64_t ns = send_off[p + 1] - send_off[p], nr = recv_off[p + 1] - recv_off[p];
    if (ns > 0 && ncclSend((const char *)sendbuf + (size_t)send_off[p] * es, (size_t)ns * es, ncclChar, p,
                           ctx->nccl, st) != ncclSuccess)
      failed = "ncclSend";
    if (!failed && nr > 0 &&
        ncclRecv((char *)recvbuf + (size_t)recv_off[p] * es, (size_t)nr * es, ncclChar, p, ctx->nccl,
                 st) != ncclSuccess)
      failed = "ncclRecv";
  }
  if (ncclGroupEnd() != ncclSuccess && !failed) failed = "ncclGroupEnd";
  if (failed) return ctx->err = failed, ALFD_E_COMM;
  return ALFD_OK;
}

}  // namespace alfd

namespace alfd {

template <class T>
static int dev_alloc(alfd_ctx *ctx, T **p, int64_t count) {
  void *q = nullptr;
  HIPC(hipMalloc(&q, std::max<int64_t>(count, 1) * sizeof(T)));
  ctx->allocs.push_back(q);
  *p = static_cast<T *>(q);
  return ALFD_OK;
}
static int dev_alloc_zero(alfd_ctx *ctx, double **p, int64_t count) {
  RC(dev_alloc(ctx, p, count));
  HIPC(hipMemsetAsync(*p, 0, std::max<int64_t>(count, 1) * sizeof(double), ctx->stream));
  return ALFD_OK;
}

template <class T>
static int csr_alloc(alfd_ctx *ctx, DevCsr &m, T **p, int64_t count) {
  void *q = nullptr;
  HIPC(hipMalloc(&q, std::max<int64_t>(count, 1) * sizeof(T)));
  m.owned.push_back(q);
  *p = static_cast<T *>(q);
  return ALFD_OK;
}
static void csr_free(DevCsr &m) {
  for (void *q : m.owned) hipFree(q);
  m = DevCsr();
}

static inline int grid_for_rows(int64_t nrows, int L) {
  const int64_t groups = (nrows + (kBlock / L) - 1) / (kBlock / L);
  // 256 CUs x 8 resident 256-thread workgroups; grid-stride beyond that
  return (int)std::max<int64_t>(1, std::min<int64_t>(groups, 256 * 8));
}

struct Timer {
  alfd_ctx *ctx;
  int cls;
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  Timer(alfd_ctx *c, int cl, double bytes, double fbytes = -1.0) : ctx(c), cls(cl) {
    on = ctx->timing >= 2 || (ctx->timing == 1 && cl == ALFD_T_SPMV_A);
    if (on) {
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a, ctx->stream);
      ctx->t_bytes[cls] += bytes;
      ctx->t_fbytes[cls] += fbytes < 0.0 ? bytes : fbytes;
      ctx->t_launches[cls]++;
    }
  }
  ~Timer() {
    if (on) {
      hipEventRecord(b, ctx->stream);
      ctx->timed.push_back({cls, a, b});
    }
  }
};

// accumulates the wall time of a setup phase (device-synchronised at its end)
struct PhaseClock {
  alfd_ctx *ctx;
  int phase;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  PhaseClock(alfd_ctx *c, int ph) : ctx(c), phase(ph) {}
  ~PhaseClock() {
    hipStreamSynchronize(ctx->stream);
    ctx->setup_s[phase] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
};

static void flush_timers(alfd_ctx *ctx) {
  for (auto &t : ctx->timed) {
    hipEventSynchronize(t.b);
    float ms = 0;
    hipEventElapsedTime(&ms, t.a, t.b);
    ctx->t_ms[t.cls] += ms;
    hipEventDestroy(t.a);
    hipEventDestroy(t.b);
  }
  ctx->timed.clear();
}

// ------------------------------------------------------------------- SpMV
template <int L>
static void launch_spmv_L(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi, double alpha,
                          const double *d, double *y2) {
  const int grid = grid_for_rows(m.n_list, L);
  const double *xh = m.halo;
#define ALFD_SPMV(EPI, SP)                                                                          \
  hipLaunchKernelGGL((spmv_kernel<L, EPI, SP>), dim3(grid), dim3(kBlock), 0, ctx->stream, m.n_list, \
                     m.rp, m.col, m.val, m.rows, x, xh, m.n_local_cols, y, alpha, d, y2)
  if (m.sparse) {
    if (epi == 0) ALFD_SPMV(0, true);
    else if (epi == 1) ALFD_SPMV(1, true);
    else if (epi == 2) ALFD_SPMV(2, true);
    else ALFD_SPMV(3, true);
  } else {
    if (epi == 0) ALFD_SPMV(0, false);
    else if (epi == 1) ALFD_SPMV(1, false);
    else if (epi == 2) ALFD_SPMV(2, false);
    else ALFD_SPMV(3, false);
  }
#undef ALFD_SPMV
}

template <int R, int U, bool NT>
static void launch_stream(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi, double alpha,
                          const double *d, double *y2) {
  const int64_t nbatches = (m.nrows + R - 1) / R;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((nbatches + 3) / 4, 256 * ctx->spmv_grid_mult));
#define ALFD_STREAM(EPI)                                                                              \
  hipLaunchKernelGGL((spmv_stream_kernel<R, U, EPI, NT>), dim3(grid), dim3(kBlock), 0, ctx->stream,   \
                     m.nrows, m.rp, m.col, m.val, x, m.halo, m.n_local_cols, y, alpha, d, y2)
  if (epi == 0) ALFD_STREAM(0);
  else if (epi == 1) ALFD_STREAM(1);
  else if (epi == 2) ALFD_STREAM(2);
  else ALFD_STREAM(3);
#undef ALFD_STREAM
}

template <bool NT>
static bool launch_stream_RU(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi,
                             double alpha, const double *d, double *y2) {
  const int R = ctx->spmv_stream_R, U = ctx->spmv_stream_U;
#define ALFD_RU(RR, UU) \
  if (R == RR && U == UU) return launch_stream<RR, UU, NT>(ctx, m, x, y, epi, alpha, d, y2), true
  ALFD_RU(2, 2); ALFD_RU(2, 4); ALFD_RU(4, 2); ALFD_RU(4, 4); ALFD_RU(4, 8); ALFD_RU(8, 4); ALFD_RU(8, 8);
  ALFD_RU(1, 4); ALFD_RU(2, 8); ALFD_RU(8, 2);
#undef ALFD_RU
  return false;
}

template <int R, int U>
static void launch_window(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi, double alpha,
                          const double *d, double *y2) {
  const size_t lds = (size_t)(m.win_maxW + (m.vi && !ctx->vi_off ? kDictMaxEntries : 0)) * sizeof(double);
#define ALFD_WIN(EPI, TAG)                                                                              \
  hipLaunchKernelGGL((spmv_window_kernel<R, U, EPI, TAG>), dim3((unsigned)m.win_nblocks), dim3(kBlock), \
                     lds, ctx->stream, m.nrows, m.win_RB, m.rp, m.col, m.lcol, m.val, m.blk_seg_begin,   \
                     m.blk_W, m.seg_col, m.seg_off, x, m.halo, m.n_local_cols, y, alpha, d, y2, ctx->win_xcd)
  const bool vi = m.vi && !ctx->vi_off;
  if (vi && ctx->vi_batched && m.vib_tab) {
#define ALFD_VIB(EPI, TAG)                                                                                    \
  hipLaunchKernelGGL((spmv_window_vib_kernel<EPI, TAG>), dim3((unsigned)m.win_nblocks), dim3(kBlock), lds,   \
                     ctx->stream, m.nrows, m.win_RB, m.rp, m.col, m.lcol, m.val, m.blk_seg_begin, m.blk_W,    \
                     m.seg_col, m.seg_off, x, m.halo, m.n_local_cols, y, alpha, d, y2, m.vidx, m.vidw,        \
                     m.blk_dict_off, m.blk_dict_n, m.dict, m.win_maxW, m.vib_tab, m.vib_cnt, m.vib_stride,    \
                     ctx->vi_xcd)
    if (m.tag == 0) {
      if (epi == 0) ALFD_VIB(0, 0);
      else if (epi == 1) ALFD_VIB(1, 0);
      else if (epi == 2) ALFD_VIB(2, 0);
      else ALFD_VIB(3, 0);
    } else {
      if (epi == 0) ALFD_VIB(0, 1);
      else if (epi == 1) ALFD_VIB(1, 1);
      else if (epi == 2) ALFD_VIB(2, 1);
      else ALFD_VIB(3, 1);
    }
#undef ALFD_VIB
    return;
  }
  if (vi && ctx->vi_rows_R > 0) {
    const int RR = ctx->vi_rows_R, JJ = ctx->vi_rows_J;
#define ALFD_VI(RV, JV, EPI)                                                                                  \
  hipLaunchKernelGGL((spmv_window_vi_kernel<RV, JV, EPI>), dim3((unsigned)m.win_nblocks), dim3(kBlock), lds,  \
                     ctx->stream, m.nrows, m.win_RB, m.rp, m.col, m.lcol, m.val, m.blk_seg_begin, m.blk_W,     \
                     m.seg_col, m.seg_off, x, m.halo, m.n_local_cols, y, alpha, d, y2, m.vidx, m.vidw, \
                     m.blk_dict_off, m.blk_dict_n, m.dict, m.win_maxW)
#define ALFD_VI_E(RV, JV)              \
  if (RR == RV && JJ == JV) {          \
    if (epi == 0) ALFD_VI(RV, JV, 0);  \
    else if (epi == 1) ALFD_VI(RV, JV, 1); \
    else if (epi == 2) ALFD_VI(RV, JV, 2); \
    else ALFD_VI(RV, JV, 3);           \
    return;                            \
  }
    ALFD_VI_E(4, 3) ALFD_VI_E(2, 4) ALFD_VI_E(4, 4)
    if (epi == 0) ALFD_VI(4, 2, 0);  // default shape
    else if (epi == 1) ALFD_VI(4, 2, 1);
    else if (epi == 2) ALFD_VI(4, 2, 2);
    else ALFD_VI(4, 2, 3);
    return;
#undef ALFD_VI_E
#undef ALFD_VI
  }
  if (m.tag == 0) {
    if (epi == 0) ALFD_WIN(0, 0);
    else if (epi == 1) ALFD_WIN(1, 0);
    else if (epi == 2) ALFD_WIN(2, 0);
    else ALFD_WIN(3, 0);
  } else {
    if (epi == 0) ALFD_WIN(0, 1);
    else if (epi == 1) ALFD_WIN(1, 1);
    else if (epi == 2) ALFD_WIN(2, 1);
    else ALFD_WIN(3, 1);
  }
#undef ALFD_WIN
}

template <int L, int R, int U>
static void launch_window_group(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi,
                                double alpha, const double *d, double *y2) {
  const size_t lds = (size_t)m.win_maxW * sizeof(double);
#define ALFD_WING(EPI, TAG)                                                                               \
  hipLaunchKernelGGL((spmv_window_group_kernel<L, R, U, EPI, TAG>), dim3((unsigned)m.win_nblocks),       \
                     dim3(kBlock), lds, ctx->stream, m.nrows, m.win_RB, m.rp, m.col, m.lcol, m.val,        \
                     m.blk_seg_begin, m.blk_W, m.seg_col, m.seg_off, x, m.halo, m.n_local_cols, y, alpha, \
                     d, y2)
  if (m.tag == 0) {
    if (epi == 0) ALFD_WING(0, 0);
    else if (epi == 1) ALFD_WING(1, 0);
    else if (epi == 2) ALFD_WING(2, 0);
    else ALFD_WING(3, 0);
  } else {
    if (epi == 0) ALFD_WING(0, 1);
    else if (epi == 1) ALFD_WING(1, 1);
    else if (epi == 2) ALFD_WING(2, 1);
    else ALFD_WING(3, 1);
  }
#undef ALFD_WING
}

template <int L>
static bool launch_window_group_RU(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi,
                                   double alpha, const double *d, double *y2) {
  const int R = ctx->spmv_group_R, U = ctx->spmv_group_U;
#define ALFD_RU(RR, UU) \
  if (R == RR && U == UU) return launch_window_group<L, RR, UU>(ctx, m, x, y, epi, alpha, d, y2), true
  ALFD_RU(4, 4); ALFD_RU(2, 4); ALFD_RU(4, 2); ALFD_RU(8, 4); ALFD_RU(8, 2);
#undef ALFD_RU
  return false;
}

static bool launch_window_RU(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi,
                             double alpha, const double *d, double *y2) {
  if (m.L == 32) return launch_window_group_RU<32>(ctx, m, x, y, epi, alpha, d, y2);
  if (m.L == 16) return launch_window_group_RU<16>(ctx, m, x, y, epi, alpha, d, y2);
  if (m.L == 8) return launch_window_group_RU<8>(ctx, m, x, y, epi, alpha, d, y2);
  if (m.L != 64) return false;
  const int R = ctx->spmv_stream_R, U = ctx->spmv_stream_U;
#define ALFD_RU(RR, UU) \
  if (R == RR && U == UU) return launch_window<RR, UU>(ctx, m, x, y, epi, alpha, d, y2), true
  ALFD_RU(1, 4); ALFD_RU(2, 4); ALFD_RU(2, 8); ALFD_RU(4, 2); ALFD_RU(4, 4); ALFD_RU(4, 8); ALFD_RU(8, 4);
  ALFD_RU(1, 8); ALFD_RU(1, 16); ALFD_RU(2, 16);
#undef ALFD_RU
  return false;
}


template <int L>
static void launch_vss(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi, double alpha,
                       const double *d, double *y2) {
  const DevCsr::Vs &v = m.vs;
  const size_t lds = (size_t)kVsWinOff + (size_t)v.maxW * sizeof(double);
#define ALFD_VSS(EPI, TAG)                                                                                          \
  hipLaunchKernelGGL((spmv_vss_kernel<L, EPI, TAG>), dim3((unsigned)v.nb), dim3(256), lds, ctx->stream, v.stream, \
                     v.sb, v.tab, v.cnt, v.stride, v.seg_begin, v.blkW, v.seg_col, v.seg_off, v.doff, v.dn, v.dict, \
                     x, m.halo, m.n_local_cols, y, alpha, d, y2)
  if (m.tag == 0) {
    if (epi == 0) ALFD_VSS(0, 0);
    else if (epi == 1) ALFD_VSS(1, 0);
    else if (epi == 2) ALFD_VSS(2, 0);
    else ALFD_VSS(3, 0);
  } else {
    if (epi == 0) ALFD_VSS(0, 1);
    else if (epi == 1) ALFD_VSS(1, 1);
    else if (epi == 2) ALFD_VSS(2, 1);
    else ALFD_VSS(3, 1);
  }
#undef ALFD_VSS
}

static bool launch_vs(alfd_ctx *ctx, const DevCsr &m, const double *x, double *y, int epi, double alpha,
                      const double *d, double *y2, int64_t first_block = 0, int64_t n_blocks = -1) {
  const DevCsr::Vs &v = m.vs;
  if (ctx->vs_lds_base_ok < 0) {   // first batch-major launch of this context: does dynamic LDS start at offset 0?
    uint32_t *probe = nullptr, base = 1;
    if (hipMalloc((void **)&probe, sizeof(uint32_t)) == hipSuccess) {
      hipLaunchKernelGGL(vs_lds_base_probe_kernel, dim3(1), dim3(64), 4096, ctx->stream, probe);
      hipMemcpyAsync(&base, probe, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
      hipStreamSynchronize(ctx->stream);
      hipFree(probe);
    }
    ctx->vs_lds_base_ok = base == 0 ? 1 : 0;
    if (!ctx->vs_lds_base_ok)
      std::fprintf(stderr, "[alfd] dynamic LDS does not start at offset 0 (%u): batch-major SpMV formats disabled\n", base);
  }
  if (!ctx->vs_lds_base_ok) return false;
  if (v.L == 32) return launch_vss<32>(ctx, m, x, y, epi, alpha, d, y2), true;
  if (v.L == 16) return launch_vss<16>(ctx, m, x, y, epi, alpha, d, y2), true;
  if (v.L == 8) return launch_vss<8>(ctx, m, x, y, epi, alpha, d, y2), true;
  const int NW = ctx->vs_NW;
  const size_t lds = (size_t)(v.wide ? VsFmt<1>::kWinOff : VsFmt<0>::kWinOff) + (size_t)v.maxW * sizeof(double);
  if (n_blocks < 0) n_blocks = v.nb - first_block;
  if (n_blocks <= 0) return true;
  const unsigned grid = (unsigned)n_blocks;
  const int32_t base = (int32_t)first_block;
#define ALFD_VS_ARGS \
  v.stream, v.tab, v.stride, v.hdrb, v.segx, v.seg_stride, v.dict, x, m.halo, m.n_local_cols, y, alpha, d, y2, ctx->vs_xcd, base
#define ALFD_VS(EPI, NWV)                                                                                        \
  do {                                                                                                           \
    if (v.wide) /* 10-bit codes: one instantiation per epilogue (4 waves) */                                    \
      hipLaunchKernelGGL((spmv_vs_kernel<EPI, 0, 4, 1>), dim3(grid), dim3(256), lds, ctx->stream,     \
                         ALFD_VS_ARGS);                                                                          \
    else if (m.tag == 0)                                                                                         \
      hipLaunchKernelGGL((spmv_vs_kernel<EPI, 0, NWV>), dim3(grid), dim3(64 * NWV), lds, ctx->stream, \
                         ALFD_VS_ARGS);                                                                          \
    else /* multigrid level matrix: its own instantiation, so that profiles keep the two apart */               \
      hipLaunchKernelGGL((spmv_vs_kernel<EPI, 1, 4>), dim3(grid), dim3(256), lds, ctx->stream,        \
                         ALFD_VS_ARGS);                                                                          \
  } while (0)
#define ALFD_VS_E(NWV)                  \
  do {                                  \
    if (epi == 0) ALFD_VS(0, NWV);      \
    else if (epi == 1) ALFD_VS(1, NWV); \
    else if (epi == 2) ALFD_VS(2, NWV); \
    else ALFD_VS(3, NWV);               \
  } while (0)
  if (NW == 8) ALFD_VS_E(8);
  else if (NW == 2) ALFD_VS_E(2);
  else ALFD_VS_E(4);
#undef ALFD_VS_E
#undef ALFD_VS
#undef ALFD_VS_ARGS
  return true;
}

static int halo_exchange(alfd_ctx *ctx, DevCsr &m, const double *x, hipStream_t st = nullptr);
static int spmv_launch_local(alfd_ctx *ctx, DevCsr &m, const double *x, double *y, int epi, double alpha, const double *d,
                             double *y2);

// epi 0: y = A x; 1: y = fma(alpha, A x, y); 2: y = d .* (A x); 3: y = A x, y2 = d .* y
static int spmv_m(alfd_ctx *ctx, DevCsr &m, int cls, const double *x, double *y, int epi, double alpha = 0.0,
                  const double *d = nullptr, double *y2 = nullptr) {
  if (!m.present) return ctx->err = "matrix not set", ALFD_E_NOT_SETUP;
  // RCCL send/recv pairs can be skipped by ranks with nothing to exchange; the
  // barrier-based in-process group needs every rank in every exchange.
  const bool exchange = ctx->nranks > 1 && !m.rep && (ctx->local || ctx->host_alltoallv || m.n_halo > 0 || m.send_off.back() > 0);
  // Long-row batch-major operators on a partitioned context: the row blocks that read no halo column come first in the
  // plan, so they are launched BEFORE the exchange is started (on a stream of its own: pack, send / receive); the
  // blocks along the partition boundary follow when the halo has arrived.  Row sums do not depend on the order of the
  // blocks: same bits as the one-launch form.
  const bool overlap = exchange && (ctx->overlap_halo > 0 || (ctx->overlap_halo < 0 && (ctx->local || ctx->host_alltoallv))) && ctx->xstream && m.vs.on && m.vs.L == 64 && ctx->vs_enable && !ctx->vi_off &&
                       m.vs.nb_interior > 0 && !m.sparse && m.n_list > 0;
  if (overlap) {
    Timer tm(ctx, cls, m.algorithmic_bytes(), m.streamed_bytes(true, true));
    HIPC(hipEventRecord(ctx->ev_x, ctx->stream));          // x is complete here
    if (!launch_vs(ctx, m, x, y, epi, alpha, d, y2, 0, m.vs.nb_interior)) return ctx->err = "batch-major launch failed", ALFD_E_HIP;
    HIPC(hipStreamWaitEvent(ctx->xstream, ctx->ev_x, 0));
    RC(halo_exchange(ctx, m, x, ctx->xstream));
    HIPC(hipEventRecord(ctx->ev_halo, ctx->xstream));
    HIPC(hipStreamWaitEvent(ctx->stream, ctx->ev_halo, 0));
    launch_vs(ctx, m, x, y, epi, alpha, d, y2, m.vs.nb_interior, m.vs.nb - m.vs.nb_interior);
    HIPC(hipGetLastError());
    return ALFD_OK;
  }
  if (exchange) RC(halo_exchange(ctx, m, x));
  if (m.sparse && epi != 1) {
    // rows outside the list are structurally empty: their result is 0
    HIPC(hipMemsetAsync(y, 0, m.nrows * sizeof(double), ctx->stream));
    if (epi == 3) HIPC(hipMemsetAsync(y2, 0, m.nrows * sizeof(double), ctx->stream));
  }
  if (m.n_list == 0) return ALFD_OK;
  Timer tm(ctx, cls, m.algorithmic_bytes(), m.streamed_bytes(!ctx->vi_off, ctx->vs_enable != 0 && !ctx->vi_off));
  return spmv_launch_local(ctx, m, x, y, epi, alpha, d, y2);
}

// the kernel of the storage form in use, on this rank's rows (halo already in place)
static int spmv_launch_local(alfd_ctx *ctx, DevCsr &m, const double *x, double *y, int epi, double alpha, const double *d,
                             double *y2) {
  if (m.vs.on && ctx->vs_enable && !ctx->vi_off && launch_vs(ctx, m, x, y, epi, alpha, d, y2)) {   // vi_off: alfd_bench_spmv_format(…, 0)
    HIPC(hipGetLastError());
    return ALFD_OK;
  }
  if (m.win && launch_window_RU(ctx, m, x, y, epi, alpha, d, y2)) {
    HIPC(hipGetLastError());
    return ALFD_OK;
  }
  if (m.L == 64 && !m.sparse && ctx->spmv_stream_R > 0) {
    const bool ok = ctx->spmv_nt ? launch_stream_RU<true>(ctx, m, x, y, epi, alpha, d, y2)
                                 : launch_stream_RU<false>(ctx, m, x, y, epi, alpha, d, y2);
    if (ok) {
      HIPC(hipGetLastError());
      return ALFD_OK;
    }
  }
  switch (m.L) {
    case 4: launch_spmv_L<4>(ctx, m, x, y, epi, alpha, d, y2); break;
    case 8: launch_spmv_L<8>(ctx, m, x, y, epi, alpha, d, y2); break;
    case 16: launch_spmv_L<16>(ctx, m, x, y, epi, alpha, d, y2); break;
    case 32: launch_spmv_L<32>(ctx, m, x, y, epi, alpha, d, y2); break;
    default: launch_spmv_L<64>(ctx, m, x, y, epi, alpha, d, y2); break;
  }
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// A short-row operator that took the batch-major form (spmv_vss_kernel) also has the windowed group kernel to fall back
// on, and which of the two is faster depends on the operator: the 27-point stencils of cfg 2 run 3x faster batch-major,
// the 9.9 M x 0.42 M gradient block Bt of the Stokes system (15 entries per row, few translates) 1.5x SLOWER (0.66 against
// 0.44 ms).  Large operators are therefore timed once at upload, five launches of each on a zero vector, and keep the
// faster form; results do not depend on the choice (same canonical sums).
static int pick_short_row_format(alfd_ctx *ctx, DevCsr &m) {
  if (!m.vs.on || m.vs.L == 64 || m.nnz < 20000000 || ctx->vi_off || !ctx->vs_enable || m.sparse) return ALFD_OK;
  double *x = nullptr, *y = nullptr;
  const int64_t nx = std::max<int64_t>(ctx->nranks > 1 && !m.rep ? m.n_local_cols : m.ncols, 1);
  HIPC(hipMalloc((void **)&x, nx * sizeof(double)));
  HIPC(hipMalloc((void **)&y, std::max<int64_t>(m.nrows, 1) * sizeof(double)));
  HIPC(hipMemsetAsync(x, 0, nx * sizeof(double), ctx->stream));
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  float t[2] = {0.f, 0.f};
  bool ok = true;
  for (int f = 0; f < 2 && ok; ++f) {
    for (int it = 0; it < 6 && ok; ++it) {   // the first launch of each form is a warm-up
      if (it == 1) hipEventRecord(e0, ctx->stream);
      m.vs.on = f == 0;   // f = 1: whatever the operator runs on without the batch-major form
      ok = spmv_launch_local(ctx, m, x, y, 0, 0.0, nullptr, nullptr) == ALFD_OK;
    }
    m.vs.on = true;
    hipEventRecord(e1, ctx->stream);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[f], e0, e1);
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(x);
  hipFree(y);
  HIPC(hipGetLastError());
  if (ok && t[1] < 0.9f * t[0]) {
    m.vs.on = false;
    if (ctx->cfg.log_level > 0 || std::getenv("ALFD_LOG_UPLOAD"))
      std::fprintf(stderr, "[alfd] short-row operator (%lld rows, %lld nnz): without the batch-major form %.3f ms, with it %.3f ms per launch "
                   "-> batch-major form dropped\n", (long long)m.nrows, (long long)m.nnz, t[1] / 5.0, t[0] / 5.0);
  }
  return ALFD_OK;
}

static int spmv(alfd_ctx *ctx, int slot, const double *x, double *y, int epi, double alpha = 0.0,
                const double *d = nullptr, double *y2 = nullptr) {
  DevCsr &m = ctx->mat[slot];
  if (!m.present) return ctx->err = "matrix slot " + std::to_string(slot) + " not set", ALFD_E_NOT_SETUP;
  return spmv_m(ctx, m, slot == ALFD_A ? ALFD_T_SPMV_A : ALFD_T_SPMV_OTHER, x, y, epi, alpha, d, y2);
}

// ------------------------------------------------------------ reductions
// count dots whose chunk partials sit at partial[j*pstride ..]; results land in
// sc[out+j] (single rank) with optional PCG post-op.
static int finish_dots(alfd_ctx *ctx, int64_t nb, int count, int out, int fin) {
  if (ctx->nranks == 1 || ctx->dots_replicated) {
    hipLaunchKernelGGL(dot_final_kernel, dim3(count), dim3(kBlock), 0, ctx->stream, ctx->partial, nb,
                       ctx->pstride, ctx->sc, out, fin);
    HIPC(hipGetLastError());
    return ALFD_OK;
  }
  // multi-rank: local sums -> all-gather -> ordered sum (rank 0 first); the
  // PCG post-ops are applied by a tiny follow-up kernel on every rank.
  hipLaunchKernelGGL(dot_final_kernel, dim3(count), dim3(kBlock), 0, ctx->stream, ctx->partial, nb,
                     ctx->pstride, ctx->sc, (int)S_STAGE, (int)FIN_STORE);
  HIPC(hipGetLastError());
  RC(comm_allgather(ctx, ctx->sc + S_STAGE, ctx->gather, (size_t)count * sizeof(double)));
  const int tgt = fin == FIN_STORE ? out : (int)S_TMP;
  hipLaunchKernelGGL(rank_sum_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->gather, ctx->nranks, count,
                     ctx->sc, tgt);
  if (fin != FIN_STORE) {
    // re-run the post-op on the summed value: a 1-chunk "partial" == the value
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->sc + S_TMP,
                       (int64_t)1, (int64_t)0, ctx->sc, out, fin);
  }
  HIPC(hipGetLastError());
  return ALFD_OK;
}

static int dot_async(alfd_ctx *ctx, int64_t npad, const double *x, const double *y, int out,
                     int fin = FIN_STORE) {
  const int64_t nb = npad / kChunk;
  if (nb > 0) {
    Timer tm(ctx, ALFD_T_DOT, 16.0 * npad);
    hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream, x, y,
                       ctx->partial);
  }
  HIPC(hipGetLastError());
  return finish_dots(ctx, nb, 1, out, fin);
}

static int read_scalars(alfd_ctx *ctx, int first, int count) {
  // The mirror is mapped pinned host memory: a one-workgroup kernel stores the scalars
  // there directly (a D2H hipMemcpyAsync goes through the runtime's blit kernels, which
  // cost far more than this launch at one readback per CG iteration).
  hipLaunchKernelGGL(mirror_scalars_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, ctx->sc + first,
                     ctx->sc_host + first, count);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(ctx->stream));
  return ALFD_OK;
}

// A rank may own zero rows of a block (e.g. the multiplier rows of a localised immersed body):
// its vectors are empty and the launch is skipped; collectives are still entered by every rank.
#define VEC_LAUNCH(kernel, npad, bytes_per_elem, ...)                                              \
  do {                                                                                             \
    if ((npad) > 0) {                                                                              \
      Timer tm__(ctx, ALFD_T_VEC, (double)(bytes_per_elem) * (double)(npad));                      \
      hipLaunchKernelGGL(kernel, dim3((unsigned)((npad) / kChunk)), dim3(kBlock), 0, ctx->stream,  \
                         __VA_ARGS__);                                                             \
    }                                                                                              \
  } while (0)

// ------------------------------------------------------------- operators
// Inner operators (SPD) the CG runs on:
//   OP_AUG   y = A x + gamma Ct (invW .* (C x))         stokes...:991-993; A11_aug elliptic...:807
//   OP_MP    y = Mp x                                   stokes...:929
//   OP_A22   y = A2 x + gamma2 M (invW .* (M x))        A22_aug, elliptic_interface.cc:810
//   OP_AUG2  2x2 [[A11_aug, A12_aug],[A21_aug, A22_aug]] on [x0 | pad | x1] (elliptic...:927-929):
//            s = C x0 - M x1, t = invW .* s, y0 = A x0 + gamma Ct t, y1 = A2 x1 - gamma2 M t
//   OP_K     y = A x  (K_inv of the rational branch: UMFPACK in the reference,
//            immersed_laplace.cc:617-620; here CG to alfd_config::inner)
//   OP_MASS  y = M x  (the immersed mass matrix of the exact W^-1 = (M^-1)^2, stokes...:979-985)
enum OpKind { OP_AUG = 0, OP_MP = 1, OP_A22 = 2, OP_AUG2 = 3, OP_K = 4, OP_MASS = 5 };

static inline int64_t op_npad(const alfd_ctx *ctx, int op) {
  if (op == OP_MASS) return pad_chunk(ctx->n[ctx->nblocks - 1]);
  return (op == OP_AUG || op == OP_K) ? pad_chunk(ctx->n[0]) : op == OP_AUG2 ? ctx->off[2] : pad_chunk(ctx->n[1]);
}

static int winv_scale(alfd_ctx *ctx, double alpha, const double *src, double *dst);

// exact_w: apply the configured W^-1 (the operator the inner CG and the outer system see);
// false: the diagonal weight (everything inside the inner preconditioner).
static int op_apply(alfd_ctx *ctx, int op, const double *x, double *y, bool exact_w = false) {
  const double *w = ctx->diag[ALFD_INVW];
  switch (op) {
    case OP_AUG:
      RC(spmv(ctx, ALFD_A, x, y, 0));
      if (ctx->cfg.aug_assembled) return ALFD_OK;  // operator form: A already holds the AL term
      if (exact_w && ctx->cfg.w_inverse != ALFD_W_DIAGONAL) {
        RC(spmv(ctx, ALFD_C, x, ctx->t_lam, 0));
        RC(winv_scale(ctx, 1.0, ctx->t_lam, ctx->t_lam));
      } else {
        RC(spmv(ctx, ALFD_C, x, ctx->t_lam, 2, 0.0, w));
      }
      return spmv(ctx, ALFD_CT, ctx->t_lam, y, 1, ctx->cfg.gamma);
    case OP_MASS:
      return spmv(ctx, ALFD_M, x, y, 0);
    case OP_MP:
      return spmv(ctx, ALFD_MP, x, y, 0);
    case OP_K:
      return spmv(ctx, ALFD_A, x, y, 0);
    case OP_A22:
      RC(spmv(ctx, ALFD_A2, x, y, 0));
      if (exact_w && ctx->cfg.w_inverse != ALFD_W_DIAGONAL) {
        RC(spmv(ctx, ALFD_M, x, ctx->t_lam, 0));
        RC(winv_scale(ctx, 1.0, ctx->t_lam, ctx->t_lam));
      } else {
        RC(spmv(ctx, ALFD_M, x, ctx->t_lam, 2, 0.0, w));
      }
      return spmv(ctx, ALFD_M, ctx->t_lam, y, 1, ctx->cfg.gamma2);
    default: {
      const double *x1 = x + ctx->off[1];
      double *y1 = y + ctx->off[1];
      const int64_t nlp = pad_chunk(ctx->n[2]);
      RC(spmv(ctx, ALFD_C, x, ctx->t_lam, 0));
      RC(spmv(ctx, ALFD_M, x1, ctx->t_lam, 1, -1.0));
      if (exact_w) RC(winv_scale(ctx, 1.0, ctx->t_lam, ctx->t_lam));  // diagonal weight unless w_inverse says otherwise
      else VEC_LAUNCH(pmul_scale_kernel, nlp, 24, 1.0, w, ctx->t_lam, ctx->t_lam);
      RC(spmv(ctx, ALFD_A, x, y, 0));
      RC(spmv(ctx, ALFD_CT, ctx->t_lam, y, 1, ctx->cfg.gamma));
      RC(spmv(ctx, ALFD_A2, x1, y1, 0));
      return spmv(ctx, ALFD_M, ctx->t_lam, y1, 1, -ctx->cfg.gamma2);
    }
  }
}

static const double *op_dinv(const alfd_ctx *ctx, int op) {
  return op == OP_AUG ? ctx->dinv_aug : op == OP_A22 ? ctx->dinv_a22 : op == OP_AUG2 ? ctx->dinv_aug2
         : op == OP_K ? ctx->dinv_k : op == OP_MASS ? ctx->dinv_m : ctx->diag[ALFD_MP_LUMPED_INV];
}

// Chebyshev sweep z = p_k(D^-1 Op) D^-1 r
static int cheb_apply(alfd_ctx *ctx, int op, const double *r, double *z, int64_t npad) {
  const double lmax = ctx->lam_max[op], lmin = lmax / ctx->cfg.cheb_eig_ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  const double sigma = theta / delta;
  double rho = 1.0 / sigma;
  const int k = ctx->cfg.cheb_degree;
  const double *dinv = op_dinv(ctx, op);
  VEC_LAUNCH(cheb_init_kernel, npad, k > 1 ? 40 : 32, 1.0 / theta, dinv, r, ctx->c_d, z, ctx->c_res,
             k > 1 ? 1 : 0);
  // SpMV writes rows only: padding of the product vector must be zero
  if (k > 1) HIPC(hipMemsetAsync(ctx->c_tmp, 0, npad * sizeof(double), ctx->stream));
  for (int j = 1; j < k; ++j) {
    RC(op_apply(ctx, op, ctx->c_d, ctx->c_tmp));
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
    VEC_LAUNCH(cheb_step_kernel, npad, 64, c1, c2, dinv, ctx->c_tmp, ctx->c_res, ctx->c_d, z);
    rho = rho_new;
  }
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// deal.II SolverCG via inverse_operator (zero initial guess) [EXT]; b and x are
// padded device vectors of the operator's span.
static int ml_cycle(alfd_ctx *ctx, int l, const double *r, double *z);
static int ml_apply(alfd_ctx *ctx, const double *r, double *z);
static int pcg(alfd_ctx *ctx, int op, int prec, const alfd_control &ctrl, const double *b, double *x,
               int *its_out, State *st_out, double *res_out) {
  const int64_t npad = op_npad(ctx, op);
  const int64_t nb = npad / kChunk;
  const double *dinv = op_dinv(ctx, op);
  double *r = ctx->w_r, *z = ctx->w_z, *p = ctx->w_p, *Ap = ctx->w_Ap;
  VEC_LAUNCH(scale_copy_kernel, npad, 16, 1.0, b, r);   // r = b
  HIPC(hipMemsetAsync(x, 0, npad * sizeof(double), ctx->stream));
  // SpMV writes rows only: the padding of Ap may hold data of a previous, longer
  // solve and must be zero for the fused r-update / dot kernels.
  HIPC(hipMemsetAsync(Ap, 0, npad * sizeof(double), ctx->stream));
  Control sc{ctrl};
  RC(dot_async(ctx, npad, r, r, S_RR));
  RC(read_scalars(ctx, S_RR, 1));
  double res = std::sqrt(ctx->sc_host[S_RR]);
  State st = sc.check(0, res);
  int its = 0;
  while (st == ITERATE) {
    ++its;
    const double *zz = z;
    if (prec == ALFD_PREC_IDENTITY) {
      zz = r;
      RC(dot_async(ctx, npad, r, r, 0, FIN_RZ));
    } else if (prec == ALFD_PREC_JACOBI) {
      VEC_LAUNCH(jacobi_dot_kernel, npad, 24, dinv, r, z, ctx->partial);
      RC(finish_dots(ctx, nb, 1, 0, FIN_RZ));
    } else if (prec == ALFD_PREC_MULTILEVEL && op == OP_AUG) {
      RC(ml_apply(ctx, r, z));
      RC(dot_async(ctx, npad, r, z, 0, FIN_RZ));
    } else {
      RC(cheb_apply(ctx, op, r, z, npad));
      RC(dot_async(ctx, npad, r, z, 0, FIN_RZ));
    }
    VEC_LAUNCH(p_update_kernel, npad, its == 1 ? 16 : 24, ctx->sc, its == 1 ? 1 : 0, zz, p);
    RC(op_apply(ctx, op, p, Ap, true));
    RC(dot_async(ctx, npad, p, Ap, 0, FIN_ALPHA));
    VEC_LAUNCH(xr_update_dot_kernel, npad, 48, ctx->sc, p, Ap, x, r, ctx->partial);
    RC(finish_dots(ctx, nb, 1, S_RR, FIN_STORE));
    RC(read_scalars(ctx, S_RR, 1));
    res = std::sqrt(ctx->sc_host[S_RR]);
    st = sc.check(its, res);
  }
  *its_out = its;
  *st_out = st;
  *res_out = res;
  return ALFD_OK;
}

static int inner_solve(alfd_ctx *ctx, int op, const double *b, double *x) {
  int its = 0;
  State st;
  double res;
  const bool mp = op == OP_MP;
  RC(pcg(ctx, op, mp ? (int)ALFD_PREC_JACOBI : ctx->cfg.inner_prec, mp ? ctx->cfg.mp_inner : ctx->cfg.inner, b,
         x, &its, &st, &res));
  (mp ? ctx->mp_its : ctx->inner_its) += its;
  if (ctx->cfg.log_level >= 3 && ctx->rank == 0)
    std::printf("DEAL:%s:cg::%s step %d value %.17g\n", mp ? "mp" : "aug",
                st == SUCCESS ? "Convergence" : "Failure", its, res);
  if (st == FAILURE) {
    if (std::isnan(res)) return ctx->err = "inner CG breakdown (NaN)", ALFD_E_BREAKDOWN;
    if (ctx->cfg.on_inner_failure == ALFD_INNER_THROW)
      return ctx->err = "inner CG did not converge (SolverControl::NoConvergence)",
             ALFD_E_NO_CONVERGENCE_INNER;
    ctx->inner_failures++;
  }
  return ALFD_OK;
}

static inline bool is_elliptic(int v) { return v == ALFD_AL_ELL_IDEAL || v == ALFD_AL_ELL_MODIFIED; }

// x = M^-1 b by Jacobi-preconditioned CG to alfd_config::mass (UMFPACK in the reference,
// stokes...:966-968).  It may run in the middle of an outer inner-CG iteration (inside
// Aug p), so it works on its own CG vectors, scalar table and reduction buffers.
static int mass_solve(alfd_ctx *ctx, const double *b, double *x) {
  auto swap_ws = [&]() {
    std::swap(ctx->w_r, ctx->n_r);
    std::swap(ctx->w_z, ctx->n_z);
    std::swap(ctx->w_p, ctx->n_p);
    std::swap(ctx->w_Ap, ctx->n_Ap);
    std::swap(ctx->sc, ctx->n_sc);
    std::swap(ctx->sc_host, ctx->n_sc_host);
    std::swap(ctx->partial, ctx->n_partial);
    std::swap(ctx->gather, ctx->n_gather);
  };
  int its = 0;
  State st = FAILURE;
  double res = 0;
  swap_ws();
  const int rc = pcg(ctx, OP_MASS, ALFD_PREC_JACOBI, ctx->cfg.mass, b, x, &its, &st, &res);
  swap_ws();
  if (rc != ALFD_OK) return rc;
  ctx->mass_its += its;
  if (st == FAILURE) {
    if (std::isnan(res)) return ctx->err = "mass-matrix CG breakdown (NaN)", ALFD_E_BREAKDOWN;
    return ctx->err = "mass-matrix CG (exact W^-1) did not converge", ALFD_E_NO_CONVERGENCE_INNER;
  }
  return ALFD_OK;
}

// dst = alpha * W^-1 src on the multiplier block (padded vectors; dst may alias src)
static int winv_scale(alfd_ctx *ctx, double alpha, const double *src, double *dst) {
  const int64_t nlp = pad_chunk(ctx->n[ctx->nblocks - 1]);
  if (ctx->cfg.w_inverse == ALFD_W_DIAGONAL) {
    VEC_LAUNCH(pmul_scale_kernel, nlp, 24, alpha, ctx->diag[ALFD_INVW], src, dst);
    return ALFD_OK;
  }
  RC(mass_solve(ctx, src, ctx->m_tmp));
  const double *z = ctx->m_tmp;
  if (ctx->cfg.w_inverse == ALFD_W_MASS_INV_SQUARED) {
    RC(mass_solve(ctx, ctx->m_tmp, ctx->m_tmp2));
    z = ctx->m_tmp2;
  }
  VEC_LAUNCH(scale_copy_kernel, nlp, 16, alpha, z, dst);
  return ALFD_OK;
}

// ---- RationalPreconditioner (rational_preconditioner.h:29-63) ----------------
static const double kRatRes[21] = {
    1.1133752551375149e+01,  -4.5192561264009555e+02, -5.4280235488093114e+00, -6.6119823627983498e-01,
    -1.5483255874020074e-01, -4.8435293477731435e-02, -1.7569986796633446e-02, -6.9011933591631392e-03,
    -2.8275585395562131e-03, -1.1823861060446343e-03, -4.9806992558149195e-04, -2.0975776516702764e-04,
    -8.7959042415258930e-05, -3.6650480089224726e-05, -1.5149104182285630e-05, -6.1866179967421625e-06,
    -2.4691626461139533e-06, -9.3898594542244485e-07, -3.2099152020952601e-07, -8.4169497470931466e-08,
    -7.7616172944516437e-09};
static const double kRatPoles[20] = {
    -4.9917060842594275e+01, -5.2698715191349796e+00, -1.7156755741861143e+00, -7.5569620064292298e-01,
    -3.7811376547012854e-01, -2.0130525955937850e-01, -1.1058502730933521e-01, -6.1664070123493613e-02,
    -3.4578652087400880e-02, -1.9394206381182760e-02, -1.0845568864180035e-02, -6.0343457447149737e-03,
    -3.3328397814762593e-03, -1.8198589302273998e-03, -9.7434812604726647e-04, -5.0332017175529794e-04,
    -2.4317839761161207e-04, -1.0297057301403903e-04, -3.2227929557637293e-05, -3.3293811779427837e-06};
constexpr int kRatSystems = 21;  // 20 shifted systems + the mass system

// v1 = sum_i rho res_i (A_Gamma - rho p_i M)^-1 u1 + res_0 M^-1 u1 : the 21 CG solves
// (Jacobi-preconditioned; the mass solve unpreconditioned, :54-56) run in lock step.
static int rational_apply(alfd_ctx *ctx, const double *u1, double *v1) {
  const int64_t npl = pad_chunk(ctx->n[1]);
  const int cps = (int)(npl / kChunk);
  const int64_t ntot = npl * kRatSystems;
  const unsigned grid = (unsigned)(ntot / kChunk);
  const alfd_control &ctl = ctx->cfg.rational;
  double *sh = ctx->rt_scb_host;
  hipLaunchKernelGGL(b_replicate_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, cps, u1, ctx->rt_r);
  HIPC(hipMemsetAsync(ctx->rt_x, 0, ntot * sizeof(double), ctx->stream));
  HIPC(hipMemsetAsync(ctx->rt_Ap, 0, ntot * sizeof(double), ctx->stream));
  for (int s = 0; s < kRatSystems; ++s) {
    for (int k = 0; k < kBS; ++k) sh[s * kBS + k] = 0.0;
    sh[s * kBS + B_ACTIVE] = 1.0;
  }
  HIPC(hipMemcpyAsync(ctx->rt_scb, sh, kRatSystems * kBS * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  // initial residuals
  hipLaunchKernelGGL(dot_partial_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->rt_r, ctx->rt_r,
                     ctx->rt_partial);
  hipLaunchKernelGGL(b_final_kernel, dim3(kRatSystems), dim3(kBlock), 0, ctx->stream, ctx->rt_partial, cps,
                     ctx->rt_scb, (int)FIN_STORE);
  HIPC(hipMemcpyAsync(sh, ctx->rt_scb, kRatSystems * kBS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  Control sc[kRatSystems];
  State st[kRatSystems];
  int its[kRatSystems];
  int nactive = 0;
  for (int s = 0; s < kRatSystems; ++s) {
    sc[s] = Control{ctl};
    its[s] = 0;
    st[s] = sc[s].check(0, std::sqrt(sh[s * kBS + B_RR]));
    nactive += st[s] == ITERATE;
  }
  int it = 0;
  while (nactive > 0) {
    ++it;
    // freeze the systems that have stopped
    for (int s = 0; s < kRatSystems; ++s) sh[s * kBS + B_ACTIVE] = st[s] == ITERATE ? 1.0 : 0.0;
    for (int s = 0; s < kRatSystems; ++s)
      HIPC(hipMemcpyAsync(ctx->rt_scb + s * kBS + B_ACTIVE, sh + s * kBS + B_ACTIVE, sizeof(double),
                          hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(b_jacobi_dot_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->rt_scb, cps,
                       ctx->rt_dinv, ctx->rt_r, ctx->rt_z, ctx->rt_partial);
    hipLaunchKernelGGL(b_final_kernel, dim3(kRatSystems), dim3(kBlock), 0, ctx->stream, ctx->rt_partial, cps,
                       ctx->rt_scb, (int)FIN_RZ);
    hipLaunchKernelGGL(b_p_update_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->rt_scb, cps,
                       it == 1 ? 1 : 0, ctx->rt_z, ctx->rt_p);
    {
      const DevCsr &m = ctx->rat_mat;
      switch (m.L) {
        case 4: launch_spmv_L<4>(ctx, m, ctx->rt_p, ctx->rt_Ap, 0, 0.0, nullptr, nullptr); break;
        case 8: launch_spmv_L<8>(ctx, m, ctx->rt_p, ctx->rt_Ap, 0, 0.0, nullptr, nullptr); break;
        case 16: launch_spmv_L<16>(ctx, m, ctx->rt_p, ctx->rt_Ap, 0, 0.0, nullptr, nullptr); break;
        case 32: launch_spmv_L<32>(ctx, m, ctx->rt_p, ctx->rt_Ap, 0, 0.0, nullptr, nullptr); break;
        default: launch_spmv_L<64>(ctx, m, ctx->rt_p, ctx->rt_Ap, 0, 0.0, nullptr, nullptr); break;
      }
    }
    hipLaunchKernelGGL(dot_partial_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->rt_p, ctx->rt_Ap,
                       ctx->rt_partial);
    hipLaunchKernelGGL(b_final_kernel, dim3(kRatSystems), dim3(kBlock), 0, ctx->stream, ctx->rt_partial, cps,
                       ctx->rt_scb, (int)FIN_ALPHA);
    hipLaunchKernelGGL(b_xr_update_dot_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, ctx->rt_scb, cps,
                       ctx->rt_p, ctx->rt_Ap, ctx->rt_x, ctx->rt_r, ctx->rt_partial);
    hipLaunchKernelGGL(b_final_kernel, dim3(kRatSystems), dim3(kBlock), 0, ctx->stream, ctx->rt_partial, cps,
                       ctx->rt_scb, (int)FIN_STORE);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(sh, ctx->rt_scb, kRatSystems * kBS * sizeof(double), hipMemcpyDeviceToHost,
                        ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    nactive = 0;
    for (int s = 0; s < kRatSystems; ++s)
      if (st[s] == ITERATE) {
        its[s] = it;
        st[s] = sc[s].check(it, std::sqrt(sh[s * kBS + B_RR]));
        nactive += st[s] == ITERATE;
      }
  }
  for (int s = 0; s < kRatSystems; ++s) {
    ctx->rational_its += its[s];
    if (st[s] == FAILURE) {
      if (std::isnan(sc[s].last_value)) return ctx->err = "rational CG breakdown (NaN)", ALFD_E_BREAKDOWN;
      if (ctx->cfg.on_inner_failure == ALFD_INNER_THROW)
        return ctx->err = "rational-preconditioner CG did not converge", ALFD_E_NO_CONVERGENCE_INNER;
      ctx->inner_failures++;
    }
  }
  hipLaunchKernelGGL(b_combine_kernel, dim3((unsigned)cps), dim3(kBlock), 0, ctx->stream, kRatSystems, npl,
                     ctx->rt_coef, ctx->rt_x, v1);
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// Preconditioner vmult on padded device block vectors.
static int precond_apply(alfd_ctx *ctx, const double *u, double *v) {
  ctx->precond_applications++;
  const alfd_config &c = ctx->cfg;
  const double *w = ctx->diag[ALFD_INVW];
  const int64_t *off = ctx->off;
  const int64_t n0p = pad_chunk(ctx->n[0]), n1p = pad_chunk(ctx->n[1]), n2p = pad_chunk(ctx->n[2]);
  if (c.variant == ALFD_AL2) {
    // augmented_lagrangian_preconditioner.h:28-34
    RC(winv_scale(ctx, -c.gamma, u + off[1], v + off[1]));
    VEC_LAUNCH(scale_copy_kernel, n0p, 16, 1.0, u + off[0], ctx->rhs_tmp);
    RC(spmv(ctx, ALFD_CT, v + off[1], ctx->rhs_tmp, 1, -1.0));
    return inner_solve(ctx, OP_AUG, ctx->rhs_tmp, v + off[0]);
  }
  if (c.variant == ALFD_AL_STOKES || c.variant == ALFD_AL_STOKES_DIAG) {
    // :62-70 (block triangular) / :95-103 (block diagonal SPD)
    const bool tri = c.variant == ALFD_AL_STOKES;
    const double sgn = tri ? -1.0 : 1.0;
    RC(winv_scale(ctx, sgn * c.gamma, u + off[2], v + off[2]));
    RC(inner_solve(ctx, OP_MP, u + off[1], ctx->q_tmp));
    VEC_LAUNCH(scale_copy_kernel, n1p, 16, sgn * c.gamma_grad_div, ctx->q_tmp, v + off[1]);
    VEC_LAUNCH(scale_copy_kernel, n0p, 16, 1.0, u + off[0], ctx->rhs_tmp);
    if (tri) {
      RC(spmv(ctx, ALFD_BT, v + off[1], ctx->rhs_tmp, 1, -1.0));
      RC(spmv(ctx, ALFD_CT, v + off[2], ctx->rhs_tmp, 1, -1.0));
    }
    return inner_solve(ctx, OP_AUG, ctx->rhs_tmp, v + off[0]);
  }
  if (c.variant == ALFD_AL_ELL_MODIFIED) {
    // BlockTriangularALPreconditionerModified::vmult, ...preconditioner.h:225-228
    double *d0 = v + off[0], *d1 = v + off[1], *d2 = v + off[2];
    RC(winv_scale(ctx, -c.gamma, u + off[2], d2));                              // d2 = -gamma invW lambda
    HIPC(hipMemcpyAsync(ctx->q_tmp, u + off[1], n1p * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    RC(spmv(ctx, ALFD_M, d2, ctx->q_tmp, 1, 1.0));                              // u2 + M d2
    RC(inner_solve(ctx, OP_A22, ctx->q_tmp, d1));                               // d1 = A22_inv (...)
    if (c.w_inverse != ALFD_W_DIAGONAL) {                                        // t = invW M d1
      RC(spmv(ctx, ALFD_M, d1, ctx->t_lam, 0));
      RC(winv_scale(ctx, 1.0, ctx->t_lam, ctx->t_lam));
    } else {
      RC(spmv(ctx, ALFD_M, d1, ctx->t_lam, 2, 0.0, w));
    }
    VEC_LAUNCH(scale_copy_kernel, n0p, 16, 1.0, u + off[0], ctx->rhs_tmp);
    RC(spmv(ctx, ALFD_CT, ctx->t_lam, ctx->rhs_tmp, 1, c.gamma));               // u + gamma Ct t
    RC(spmv(ctx, ALFD_CT, d2, ctx->rhs_tmp, 1, -1.0));                          //   - Ct d2
    return inner_solve(ctx, OP_AUG, ctx->rhs_tmp, d0);                          // d0 = A11_inv (...)
  }
  if (c.variant == ALFD_AL_ELL_IDEAL) {
    // BlockTriangularALPreconditioner::vmult, ...preconditioner.h:130-156
    RC(winv_scale(ctx, -c.gamma, u + off[2], v + off[2]));
    HIPC(hipMemcpyAsync(ctx->rhs_tmp, u, off[2] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    RC(spmv(ctx, ALFD_CT, v + off[2], ctx->rhs_tmp, 1, -1.0));                  // u0 - Ct v2
    RC(spmv(ctx, ALFD_M, v + off[2], ctx->rhs_tmp + off[1], 1, 1.0));           // u1 + M v2
    return inner_solve(ctx, OP_AUG2, ctx->rhs_tmp, v);                          // [v0;v1] = Aug_inv (...)
  }
  if (c.variant == ALFD_RATIONAL) {
    // RationalPreconditioner::vmult (block diagonal, SPD): v0 = K_inv u0, v1 = rational(u1)
    RC(inner_solve(ctx, OP_K, u + off[0], v + off[0]));
    return rational_apply(ctx, u + off[1], v + off[1]);
  }
  return ctx->err = "preconditioner variant not implemented yet", ALFD_E_UNSUPPORTED;
}

// AA.vmult on padded device block vectors.
static int system_apply(alfd_ctx *ctx, const double *x, double *y) {
  const alfd_config &c = ctx->cfg;
  const int64_t *off = ctx->off;
  const int last = ctx->nblocks - 1;
  const double *w = ctx->diag[ALFD_INVW];
  if (c.variant == ALFD_AL2 || c.variant == ALFD_AL_STOKES || c.variant == ALFD_AL_STOKES_DIAG) {
    const double *x0 = x + off[0];
    double *y0 = y + off[0];
    RC(spmv(ctx, ALFD_A, x0, y0, 0));
    if (c.aug_assembled) {
      RC(spmv(ctx, ALFD_C, x0, y + off[last], 0));
    } else {
      if (c.w_inverse != ALFD_W_DIAGONAL) {
        RC(spmv(ctx, ALFD_C, x0, y + off[last], 0));
        RC(winv_scale(ctx, 1.0, y + off[last], ctx->t_lam));
      } else {
        RC(spmv(ctx, ALFD_C, x0, y + off[last], 3, 0.0, w, ctx->t_lam));
      }
      RC(spmv(ctx, ALFD_CT, ctx->t_lam, y0, 1, c.gamma));
    }
    if (ctx->nblocks == 3) {
      RC(spmv(ctx, ALFD_BT, x + off[1], y0, 1, 1.0));
      RC(spmv(ctx, ALFD_B, x0, y + off[1], 0));
    }
    RC(spmv(ctx, ALFD_CT, x + off[last], y0, 1, 1.0));
    return ALFD_OK;
  }
  if (c.variant == ALFD_RATIONAL) {
    // AA = [[K, Ct],[C, 0]] (immersed_laplace.cc:596-597)
    RC(spmv(ctx, ALFD_A, x + off[0], y + off[0], 0));
    RC(spmv(ctx, ALFD_CT, x + off[1], y + off[0], 1, 1.0));
    return spmv(ctx, ALFD_C, x + off[0], y + off[1], 0);
  }
  if (is_elliptic(c.variant)) {
    // elliptic_interface.cc:810-819
    const double *x0 = x + off[0], *x1 = x + off[1], *x2 = x + off[2];
    double *y0 = y + off[0], *y1 = y + off[1], *y2 = y + off[2];
    RC(spmv(ctx, ALFD_C, x0, y2, 0));
    RC(spmv(ctx, ALFD_M, x1, y2, 1, -1.0));                                     // y2 = C x0 - M x1
    RC(winv_scale(ctx, 1.0, y2, ctx->t_lam));
    RC(spmv(ctx, ALFD_A, x0, y0, 0));
    RC(spmv(ctx, ALFD_CT, ctx->t_lam, y0, 1, c.gamma));
    RC(spmv(ctx, ALFD_CT, x2, y0, 1, 1.0));
    RC(spmv(ctx, ALFD_A2, x1, y1, 0));
    RC(spmv(ctx, ALFD_M, ctx->t_lam, y1, 1, -c.gamma2));
    return spmv(ctx, ALFD_M, x2, y1, 1, -1.0);
  }
  return ctx->err = "system operator variant not implemented yet", ALFD_E_UNSUPPORTED;
}

// ---------------------------------------------------------------- FGMRES
static int block_dots(alfd_ctx *ctx, const double *Vb, int count, const double *w, int out) {
  const int64_t N = ctx->ntot(), nb = N / kChunk;
  if (nb > 0) {
    Timer tm(ctx, ALFD_T_DOT, 8.0 * N * (count + 1));
    hipLaunchKernelGGL(multi_dot_partial_kernel, dim3((unsigned)nb), dim3(kBlock), 0, ctx->stream, Vb, N,
                       count, w, ctx->partial, ctx->pstride);
  }
  HIPC(hipGetLastError());
  return finish_dots(ctx, nb, count, out, FIN_STORE);
}

static int fgmres(alfd_ctx *ctx, alfd_result *out) {
  const alfd_config &c = ctx->cfg;
  const int m = c.restart;
  const int64_t N = ctx->ntot();
  double *x = ctx->xb, *b = ctx->bb;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), h(m + 2), h2(m + 2), y(m);
  Control sc{c.outer};
  int k = 0;
  State st = ITERATE;
  double res = 0;
  ctx->history.clear();
  auto Vj = [&](int j) { return ctx->V + (int64_t)j * N; };
  auto Zj = [&](int j) { return ctx->Z + (int64_t)j * N; };
  do {
    RC(system_apply(ctx, x, Vj(0)));
    VEC_LAUNCH(sub_from_kernel, N, 24, b, Vj(0));
    RC(dot_async(ctx, N, Vj(0), Vj(0), S_TMP));
    RC(read_scalars(ctx, S_TMP, 1));
    res = std::sqrt(ctx->sc_host[S_TMP]);
    st = sc.check(k, res);
    if (k == 0) ctx->history.push_back(res);
    if (c.log_level >= 2 && ctx->rank == 0) std::printf("DEAL:FGMRES::Check %d\t%.17g\n", k, res);
    if (st != ITERATE) break;
    if (res != 0.0) VEC_LAUNCH(scale_kernel, N, 16, (const double *)nullptr, 0, 0, 1.0 / res, Vj(0));
    g[0] = res;
    int j = 0;
    for (; j < m && st == ITERATE; ++j) {
      RC(precond_apply(ctx, Vj(j), Zj(j)));
      double *wv = Vj(j + 1);
      RC(system_apply(ctx, Zj(j), wv));
      if (c.orthogonalization == ALFD_ORTH_MGS) {
        for (int i = 0; i <= j; ++i) {
          RC(dot_async(ctx, N, Vj(i), wv, S_H + i));
          VEC_LAUNCH(multi_axpy_neg_kernel, N, 24, Vj(i), N, 1, ctx->sc, (int)S_H + i, wv);
        }
        RC(read_scalars(ctx, S_H, j + 1));
        for (int i = 0; i <= j; ++i) h[i] = ctx->sc_host[S_H + i];
      } else {
        RC(block_dots(ctx, ctx->V, j + 1, wv, S_H));
        VEC_LAUNCH(multi_axpy_neg_kernel, N, 8.0 * (j + 3), ctx->V, N, j + 1, ctx->sc, (int)S_H, wv);
        if (c.orthogonalization == ALFD_ORTH_CGS2) {
          RC(block_dots(ctx, ctx->V, j + 1, wv, S_H + kMaxBasis));
          VEC_LAUNCH(multi_axpy_neg_kernel, N, 8.0 * (j + 3), ctx->V, N, j + 1, ctx->sc,
                     (int)S_H + kMaxBasis, wv);
          RC(read_scalars(ctx, S_H, 2 * kMaxBasis));
          for (int i = 0; i <= j; ++i) h[i] = ctx->sc_host[S_H + i] + ctx->sc_host[S_H + kMaxBasis + i];
        } else {
          RC(read_scalars(ctx, S_H, j + 1));
          for (int i = 0; i <= j; ++i) h[i] = ctx->sc_host[S_H + i];
        }
      }
      RC(dot_async(ctx, N, wv, wv, S_TMP));
      RC(read_scalars(ctx, S_TMP, 1));
      h[j + 1] = std::sqrt(ctx->sc_host[S_TMP]);
      if (h[j + 1] != 0.0)
        VEC_LAUNCH(scale_kernel, N, 16, (const double *)nullptr, 0, 0, 1.0 / h[j + 1], wv);
      for (int i = 0; i < j; ++i) {
        const double t = cs[i] * h[i] + sn[i] * h[i + 1];
        h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
        h[i] = t;
      }
      const double denom = std::sqrt(h[j] * h[j] + h[j + 1] * h[j + 1]);
      cs[j] = h[j] / denom;
      sn[j] = h[j + 1] / denom;
      h[j] = denom;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] = h[i];
      res = std::fabs(g[j + 1]);
      ++k;
      st = sc.check(k, res);
      ctx->history.push_back(res);
      if (c.log_level >= 2 && ctx->rank == 0) std::printf("DEAL:FGMRES::Check %d\t%.17g\n", k, res);
    }
    for (int i = j - 1; i >= 0; --i) {
      double s = g[i];
      for (int l = i + 1; l < j; ++l) s -= H[(size_t)i * m + l] * y[l];
      y[i] = s / H[(size_t)i * m + i];
    }
    for (int i = 0; i < j; ++i) ctx->sc_host[S_H + i] = y[i];
    HIPC(hipMemcpyAsync(ctx->sc + S_H, ctx->sc_host + S_H, j * sizeof(double), hipMemcpyHostToDevice,
                        ctx->stream));
    VEC_LAUNCH(multi_axpy_kernel, N, 8.0 * (j + 2), ctx->Z, N, j, ctx->sc, (int)S_H, x);
    HIPC(hipStreamSynchronize(ctx->stream));  // sc_host is reused next cycle
  } while (st == ITERATE);
  out->outer_iterations = k;
  out->initial_residual = sc.initial;
  out->last_residual = res;
  if (c.log_level >= 1 && ctx->rank == 0)
    std::printf(st == SUCCESS ? "DEAL:FGMRES::Convergence step %d value %.17g\n"
                              : "DEAL:FGMRES::Failure step %d value %.17g\n",
                k, res);
  if (st != SUCCESS) {
    ctx->err = "FGMRES did not converge (SolverControl::NoConvergence)";
    return std::isnan(res) ? ALFD_E_BREAKDOWN : ALFD_E_NO_CONVERGENCE_OUTER;
  }
  return ALFD_OK;
}

// Least squares min || beta e1 - H(0:n, 0:n-1) y || of the (n+1) x n leading Hessenberg block by
// Givens rotations (deal.II <= 9.5 uses Householder::least_squares: the same minimiser).  H is
// stored row-major with leading dimension m.  Returns the residual norm, y[0..n).
static double hessenberg_lsq(const std::vector<double> &H, int m, int n, double beta, std::vector<double> &y) {
  std::vector<double> R((size_t)(n + 1) * n), g(n + 1, 0.0);
  for (int i = 0; i <= n; ++i)
    for (int j = 0; j < n; ++j) R[(size_t)i * n + j] = H[(size_t)i * m + j];
  g[0] = beta;
  for (int j = 0; j < n; ++j) {
    const double a = R[(size_t)j * n + j], b = R[(size_t)(j + 1) * n + j];
    const double denom = std::sqrt(a * a + b * b);
    const double c = a / denom, sn = b / denom;
    for (int l = j; l < n; ++l) {
      const double t = c * R[(size_t)j * n + l] + sn * R[(size_t)(j + 1) * n + l];
      R[(size_t)(j + 1) * n + l] = -sn * R[(size_t)j * n + l] + c * R[(size_t)(j + 1) * n + l];
      R[(size_t)j * n + l] = t;
    }
    const double t = c * g[j];
    g[j + 1] = -sn * g[j];
    g[j] = t;
  }
  for (int i = n - 1; i >= 0; --i) {
    double sum = g[i];
    for (int l = i + 1; l < n; ++l) sum -= R[(size_t)i * n + l] * y[l];
    y[i] = sum / R[(size_t)i * n + i];
  }
  return std::fabs(g[n]);
}

// SolverFGMRES of deal.II <= 9.5 [EXT] (alfd_fgmres_flavour): modified Gram-Schmidt, delayed
// least-squares check, no counter increment at j = 0 of a cycle.
static int fgmres_dealii95(alfd_ctx *ctx, alfd_result *out) {
  const alfd_config &c = ctx->cfg;
  const int m = c.restart;
  const int64_t N = ctx->ntot();
  double *x = ctx->xb, *b = ctx->bb;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), y(m, 0.0);
  Control sc{c.outer};
  int k = 0;
  State st = ITERATE;
  double res = 0;
  ctx->history.clear();
  auto Vj = [&](int j) { return ctx->V + (int64_t)j * N; };
  auto Zj = [&](int j) { return ctx->Z + (int64_t)j * N; };
  double *aux = Vj(m);
  auto dotv = [&](const double *p, const double *q, double *r) -> int {
    RC(dot_async(ctx, N, p, q, S_TMP));
    RC(read_scalars(ctx, S_TMP, 1));
    *r = ctx->sc_host[S_TMP];
    return ALFD_OK;
  };
  do {
    RC(system_apply(ctx, x, aux));
    VEC_LAUNCH(sub_from_kernel, N, 24, b, aux);                  // aux = b - AA x
    double bb2 = 0;
    RC(dotv(aux, aux, &bb2));
    const double beta = std::sqrt(bb2);
    res = beta;
    st = sc.check(k, res);
    if (k == 0) ctx->history.push_back(res);
    if (c.log_level >= 2 && ctx->rank == 0) std::printf("DEAL:FGMRES::Check %d\t%.17g\n", k, res);
    if (st != ITERATE) break;
    std::fill(H.begin(), H.end(), 0.0);
    double a = beta;
    int ny = 0;
    for (int j = 0; j < m; ++j) {
      if (a != 0.0) VEC_LAUNCH(scale_copy_kernel, N, 16, 1.0 / a, aux, Vj(j));   // v_j = aux / a
      else HIPC(hipMemsetAsync(Vj(j), 0, N * sizeof(double), ctx->stream));
      RC(precond_apply(ctx, Vj(j), Zj(j)));
      RC(system_apply(ctx, Zj(j), aux));
      double h = 0;
      RC(dotv(aux, Vj(0), &h));
      H[(size_t)0 * m + j] = h;
      for (int i = 0; i < j; ++i) {                              // H(i+1,j) = (aux -= H(i,j) v_i) . v_{i+1}
        VEC_LAUNCH(axpy_kernel, N, 24, (const double *)nullptr, 0, -H[(size_t)i * m + j], Vj(i), aux);
        RC(dotv(aux, Vj(i + 1), &h));
        H[(size_t)(i + 1) * m + j] = h;
      }
      VEC_LAUNCH(axpy_kernel, N, 24, (const double *)nullptr, 0, -H[(size_t)j * m + j], Vj(j), aux);
      RC(dotv(aux, aux, &h));
      H[(size_t)(j + 1) * m + j] = a = std::sqrt(h);
      if (j > 0) {
        res = hessenberg_lsq(H, m, j, beta, y);
        ny = j;
        ++k;
        st = sc.check(k, res);
        ctx->history.push_back(res);
        if (c.log_level >= 2 && ctx->rank == 0) std::printf("DEAL:FGMRES::Check %d\t%.17g\n", k, res);
        if (st != ITERATE) break;
      }
    }
    for (int i = 0; i < ny; ++i) ctx->sc_host[S_H + i] = y[i];
    if (ny > 0) {
      HIPC(hipMemcpyAsync(ctx->sc + S_H, ctx->sc_host + S_H, ny * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      VEC_LAUNCH(multi_axpy_kernel, N, 8.0 * (ny + 2), ctx->Z, N, ny, ctx->sc, (int)S_H, x);
    }
    HIPC(hipStreamSynchronize(ctx->stream));
  } while (st == ITERATE);
  out->outer_iterations = k;
  out->initial_residual = sc.initial;
  out->last_residual = res;
  if (c.log_level >= 1 && ctx->rank == 0)
    std::printf(st == SUCCESS ? "DEAL:FGMRES::Convergence step %d value %.17g\n"
                              : "DEAL:FGMRES::Failure step %d value %.17g\n",
                k, res);
  if (st != SUCCESS) {
    ctx->err = "FGMRES did not converge (SolverControl::NoConvergence)";
    return std::isnan(res) ? ALFD_E_BREAKDOWN : ALFD_E_NO_CONVERGENCE_OUTER;
  }
  return ALFD_OK;
}

// deal.II SolverMinRes [EXT] on device vectors (immersed_laplace.cc:629-631,
// stokes...:1057-1064).  Vectors: u0,u1,u2 | m0,m1,m2 | v live in the Krylov arenas.
static int minres(alfd_ctx *ctx, alfd_result *out) {
  const alfd_config &c = ctx->cfg;
  const int64_t N = ctx->ntot();
  double *x = ctx->xb, *b = ctx->bb;
  double *u0 = ctx->V, *u1 = ctx->V + N, *u2 = ctx->V + 2 * N;
  double *m0 = ctx->Z, *m1 = ctx->Z + N, *m2 = ctx->Z + 2 * N;
  double *v = ctx->V + 3 * N;
  double delta[3] = {0, 0, 0}, f[2] = {0, 0}, e[2] = {0, 0};
  double r_l2 = 0, r0 = 0, tau = 0, cc = 0, ss = 0, d_ = 0, phibar = 0;
  int j = 1;
  Control sc{c.outer};
  ctx->history.clear();
  auto dotv = [&](const double *a, const double *bb_, double *res) -> int {
    RC(dot_async(ctx, N, a, bb_, S_TMP));
    RC(read_scalars(ctx, S_TMP, 1));
    *res = ctx->sc_host[S_TMP];
    return ALFD_OK;
  };
  RC(system_apply(ctx, x, m0));
  HIPC(hipMemcpyAsync(u1, m0, N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  VEC_LAUNCH(sub_from_kernel, N, 24, b, u1);                       // u1 = b - A x
  HIPC(hipMemsetAsync(v, 0, N * sizeof(double), ctx->stream));
  RC(precond_apply(ctx, u1, v));
  RC(dotv(v, u1, &delta[1]));
  if (delta[1] < 0) return ctx->err = "MinRes: preconditioner not positive definite", ALFD_E_BREAKDOWN;
  r0 = std::sqrt(delta[1]);
  r_l2 = r0;
  phibar = r0;
  HIPC(hipMemsetAsync(u0, 0, N * sizeof(double), ctx->stream));
  delta[0] = 1.0;
  HIPC(hipMemsetAsync(ctx->Z, 0, 3 * N * sizeof(double), ctx->stream));
  State st = sc.check(0, r_l2);
  ctx->history.push_back(r_l2);
  if (c.log_level >= 2 && ctx->rank == 0) std::printf("DEAL:minres::Check 0\t%.17g\n", r_l2);
  while (st == ITERATE) {
    if (delta[1] != 0)
      VEC_LAUNCH(scale_kernel, N, 16, (const double *)nullptr, 0, 0, 1.0 / std::sqrt(delta[1]), v);
    else
      HIPC(hipMemsetAsync(v, 0, N * sizeof(double), ctx->stream));
    RC(system_apply(ctx, v, u2));
    VEC_LAUNCH(axpy_kernel, N, 24, (const double *)nullptr, 0, -std::sqrt(delta[1] / delta[0]), u0, u2);
    double gamma = 0;
    RC(dotv(u2, v, &gamma));
    VEC_LAUNCH(axpy_kernel, N, 24, (const double *)nullptr, 0, -gamma / std::sqrt(delta[1]), u1, u2);
    HIPC(hipMemcpyAsync(m0, v, N * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    RC(precond_apply(ctx, u2, v));
    RC(dotv(v, u2, &delta[2]));
    if (delta[2] < 0) return ctx->err = "MinRes: preconditioner not positive definite", ALFD_E_BREAKDOWN;
    if (j == 1) {
      d_ = gamma;
      e[1] = std::sqrt(delta[2]);
    }
    if (j > 1) {
      d_ = ss * e[0] - cc * gamma;
      e[0] = cc * e[0] + ss * gamma;
      f[1] = ss * std::sqrt(delta[2]);
      e[1] = -cc * std::sqrt(delta[2]);
    }
    const double d = std::sqrt(d_ * d_ + delta[2]);
    // tau_j = c_j phibar_{j-1}, phibar_j = s_j phibar_{j-1}: division-free form of deal.II's
    // "tau *= s/c; tau *= c" (which is 0*inf when the first Lanczos coefficient is 0)
    cc = d_ / d;
    ss = std::sqrt(delta[2]) / d;
    tau = cc * phibar;
    phibar = ss * phibar;
    VEC_LAUNCH(axpy_kernel, N, 24, (const double *)nullptr, 0, -e[0], m1, m0);
    if (j > 1) VEC_LAUNCH(axpy_kernel, N, 24, (const double *)nullptr, 0, -f[0], m2, m0);
    VEC_LAUNCH(scale_kernel, N, 16, (const double *)nullptr, 0, 0, 1.0 / d, m0);
    VEC_LAUNCH(axpy_kernel, N, 24, (const double *)nullptr, 0, tau, m0, x);
    r_l2 *= std::fabs(ss);
    st = sc.check(j, r_l2);
    ctx->history.push_back(r_l2);
    if (c.log_level >= 2 && ctx->rank == 0) std::printf("DEAL:minres::Check %d\t%.17g\n", j, r_l2);
    ++j;
    double *t = u0;
    u0 = u1;
    u1 = u2;
    u2 = t;
    t = m2;
    m2 = m1;
    m1 = m0;
    m0 = t;
    delta[0] = delta[1];
    delta[1] = delta[2];
    f[0] = f[1];
    e[0] = e[1];
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  out->outer_iterations = j - 1;
  out->initial_residual = sc.initial;
  out->last_residual = r_l2;
  if (c.log_level >= 1 && ctx->rank == 0)
    std::printf(st == SUCCESS ? "DEAL:minres::Convergence step %d value %.17g\n"
                              : "DEAL:minres::Failure step %d value %.17g\n",
                j - 1, r_l2);
  if (st != SUCCESS) {
    ctx->err = "MinRes did not converge (SolverControl::NoConvergence)";
    return std::isnan(r_l2) ? ALFD_E_BREAKDOWN : ALFD_E_NO_CONVERGENCE_OUTER;
  }
  return ALFD_OK;
}

// ---------------------------------------------------------------- halo
static int halo_exchange(alfd_ctx *ctx, DevCsr &m, const double *x, hipStream_t st) {
  if (!st) st = ctx->stream;
  const int64_t nsend = m.send_off.back();
  if (nsend > 0) {
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((nsend + 255) / 256)), dim3(256), 0, st,
                       nsend, m.send_idx, x, m.send_buf);
    HIPC(hipGetLastError());
  }
  RC(comm_alltoallv(ctx, m.send_buf, m.send_off.data(), m.halo, m.recv_off.data(), sizeof(double), st));
  return ALFD_OK;
}

// Which vector blocks a slot maps between: {row block, col block}; -1 = last.
static void slot_blocks(const alfd_ctx *ctx, int slot, int *rb, int *cb) {
  const int last = ctx->nblocks - 1;
  switch (slot) {
    case ALFD_A: *rb = 0, *cb = 0; break;
    case ALFD_BT: *rb = 0, *cb = 1; break;
    case ALFD_B: *rb = 1, *cb = 0; break;
    case ALFD_CT: *rb = 0, *cb = last; break;
    case ALFD_C: *rb = last, *cb = 0; break;
    case ALFD_M: *rb = last, *cb = last; break;
    case ALFD_MP: *rb = 1, *cb = 1; break;
    case ALFD_A2: *rb = 1, *cb = 1; break;
    default: *rb = last, *cb = last; break;
  }
}

static void choose_lanes(DevCsr &m, int64_t nonempty) {
  const double avg = nonempty ? (double)m.nnz / (double)nonempty : 0.0;
  if (avg > 48) m.L = 64;
  else if (avg > 24) m.L = 32;
  else if (avg > 12) m.L = 16;
  else if (avg > 6) m.L = 8;
  else m.L = 4;
}

// Host-only part of the halo plan of one row-partitioned matrix (no HIP, no
// communication): off-rank columns sorted unique (ascending global index =>
// grouped by owner), columns rewritten to the local index space
// [owned | halo], and the receive counts per owner.  Exposed through the ABI
// as alfd_host_halo_plan so it can be exercised without a GPU.
static void host_halo_plan(int64_t nnz, const int32_t *col, const int64_t *col_offsets, int nranks, int rank,
                           int32_t *col_local, std::vector<int32_t> &halo_globals, int64_t *recv_off) {
  const int64_t c0 = col_offsets[rank], c1 = col_offsets[rank + 1];
  halo_globals.clear();
  for (int64_t k = 0; k < nnz; ++k)
    if (col[k] < c0 || col[k] >= c1) halo_globals.push_back(col[k]);
  std::sort(halo_globals.begin(), halo_globals.end());
  halo_globals.erase(std::unique(halo_globals.begin(), halo_globals.end()), halo_globals.end());
  const int32_t n_local = (int32_t)(c1 - c0);
  for (int64_t k = 0; k < nnz; ++k) {
    const int32_t c = col[k];
    if (c >= c0 && c < c1)
      col_local[k] = (int32_t)(c - c0);
    else
      col_local[k] = n_local + (int32_t)(std::lower_bound(halo_globals.begin(), halo_globals.end(), c) -
                                         halo_globals.begin());
  }
  for (int p = 0; p <= nranks; ++p) recv_off[p] = 0;
  for (int32_t c : halo_globals) {
    const int owner =
        (int)(std::upper_bound(col_offsets, col_offsets + nranks + 1, (int64_t)c) - col_offsets) - 1;
    recv_off[owner + 1]++;
  }
  for (int p = 0; p < nranks; ++p) recv_off[p + 1] += recv_off[p];
}

// Build the LDS-window format of a long-row matrix (host, multi-threaded).
// col: column indices in the LOCAL index space [local | halo].
// ---- window / value-index format: pure host planning (also exported as
// alfd_host_window_plan so that CPU tests can decode and check it), then upload.
struct WindowParams {
  int RB_long = 96, short_scale = 2, RB_vi = 96, maxW = 4096, gap = 8;
  bool want_vi = true;
};
struct WindowPlan {
  bool win = false, vi = false;
  int RB = 0;
  int32_t maxW = 0;
  int64_t nb = 0, fallback = 0;
  std::vector<uint16_t> lcol;
  std::vector<int32_t> blkW, seg_begin, seg_col, seg_off;
  std::vector<uint8_t> vidx;
  std::vector<uint16_t> vidw;
  std::vector<int32_t> blk_dn, doff;
  std::vector<double> dict;
  int64_t vi_blocks = 0, vi_nnz = 0, wide_nnz = 0;
  std::vector<uint64_t> tab;
  std::vector<int32_t> cnt;
  int stride = 0;
};
static int window_row_block(const WindowParams &wp, int L) {
  // short rows: larger row blocks, so a window serves about as many entries as for L = 64
  return L == 64 ? wp.RB_long : std::min(512, wp.RB_long * wp.short_scale * 64 / L);
}

static void plan_window(int64_t nrows, int L, const int64_t *rp, const int32_t *col, const double *val,
                        const WindowParams &wp, WindowPlan &pl) {
  const int64_t nnz = rp[nrows];
  int RB = window_row_block(wp, L);
  const bool want_vi = wp.want_vi && L == 64;
  if (want_vi && wp.RB_vi > 0 && wp.RB_vi != RB) {
    // The value-indexed kernel has its own best block size.  Sample a few blocks: if their
    // values look dictionary-codable, build the window format with that block.
    const int64_t nb0 = (nrows + RB - 1) / RB;
    int good = 0, seen = 0;
    for (int64_t q = 0; q < 32 && q < nb0; ++q) {
      const int64_t b = nb0 * q / std::min<int64_t>(32, nb0);
      const int64_t k0 = rp[b * RB], k1 = rp[std::min<int64_t>((b + 1) * RB, nrows)];
      std::unordered_set<uint64_t> u;
      for (int64_t k = k0; k < k1 && u.size() <= 1024; ++k) {
        uint64_t bits;
        std::memcpy(&bits, &val[k], 8);
        u.insert(bits);
      }
      ++seen;
      good += u.size() <= 1024;
    }
    if (good * 4 >= seen * 3) RB = wp.RB_vi;
  }
  const int maxW = wp.maxW, GAP = wp.gap;
  const int64_t nb = (nrows + RB - 1) / RB;
  if (nb == 0 || nb > 2147483000LL) return;
  pl.RB = RB;
  pl.nb = nb;
  pl.lcol.assign(nnz, 0);
  pl.blkW.assign(nb, 0);
  std::vector<int32_t> blk_nseg(nb, 0);
  pl.vidx.assign(want_vi ? nnz : 0, 0);
  pl.vidw.assign(want_vi ? nnz : 0, 0);  // 16-bit codes (uploaded only if some block needs them)
  pl.blk_dn.assign(nb, -1);
  std::vector<std::vector<double>> t_dict(want_vi ? nb : 0);
  std::vector<int64_t> t_wide(64, 0);
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::vector<int32_t>> t_seg_col(T), t_seg_off(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      const int64_t b0 = nb * t / T, b1 = nb * (t + 1) / T;
      std::vector<uint8_t> mark;
      std::vector<int32_t> pos;
      for (int64_t b = b0; b < b1; ++b) {
        const int64_t r0 = b * RB, r1 = std::min<int64_t>(r0 + RB, nrows);
        const int64_t k0 = rp[r0], k1 = rp[r1];
        if (k1 == k0) continue;
        int32_t clo = INT32_MAX, chi = -1;
        for (int64_t k = k0; k < k1; ++k) {
          clo = std::min(clo, col[k]);
          chi = std::max(chi, col[k]);
        }
        const int64_t range = (int64_t)chi - clo + 1;
        if (range > (int64_t)(1 << 22)) {
          pl.blkW[b] = -1;
          continue;
        }
        mark.assign(range, 0);
        for (int64_t k = k0; k < k1; ++k) mark[col[k] - clo] = 1;
        pos.assign(range, -1);
        int32_t W = 0, nseg = 0;
        const size_t seg_base = t_seg_col[t].size();
        int64_t c = 0;
        while (c < range) {
          if (!mark[c]) {
            ++c;
            continue;
          }
          // segment starts at c; extend over gaps shorter than GAP
          int64_t e = c, last = c;
          while (e < range) {
            if (mark[e]) last = e;
            else if (e - last >= GAP) break;
            ++e;
          }
          t_seg_col[t].push_back((int32_t)(clo + c));
          t_seg_off[t].push_back(W);
          for (int64_t q = c; q <= last; ++q) pos[q] = W++;
          ++nseg;
          c = last + 1;
        }
        if (W > maxW || W > 65535) {
          t_seg_col[t].resize(seg_base);
          t_seg_off[t].resize(seg_base);
          pl.blkW[b] = -1;
          continue;
        }
        pl.blkW[b] = W;
        blk_nseg[b] = nseg;
        for (int64_t k = k0; k < k1; ++k) pl.lcol[k] = (uint16_t)pos[col[k] - clo];
        if (want_vi) {
          // open-addressing table over the 64-bit patterns.  Up to 256 distinct values:
          // 8-bit codes; up to kDictMaxEntries: 16-bit codes; beyond that: raw block.
          constexpr int kTab = 2048;
          std::vector<uint64_t> keys(kTab);
          std::vector<int16_t> ids(kTab, (int16_t)-1);
          std::vector<double> &dv = t_dict[b];
          bool ok = true;
          for (int64_t k = k0; k < k1 && ok; ++k) {
            uint64_t bits;
            std::memcpy(&bits, &val[k], 8);
            uint32_t h = (uint32_t)((bits * 0x9E3779B97F4A7C15ull) >> 53);
            for (;;) {
              if (ids[h] < 0) {
                if ((int)dv.size() == kDictMaxEntries) {
                  ok = false;
                  break;
                }
                keys[h] = bits;
                ids[h] = (int16_t)dv.size();
                dv.push_back(val[k]);
                break;
              }
              if (keys[h] == bits) break;
              h = (h + 1) & (kTab - 1);
            }
            if (ok) pl.vidw[k] = (uint16_t)ids[h];
          }
          if (!ok) {
            dv.clear();
          } else if (dv.size() <= 256) {
            for (int64_t k = k0; k < k1; ++k) pl.vidx[k] = (uint8_t)pl.vidw[k];
            pl.blk_dn[b] = (int32_t)dv.size();
          } else {
            pl.blk_dn[b] = (int32_t)dv.size() | kDictWide;
            t_wide[t] += k1 - k0;
          }
        }
      }
    });
  for (auto &x : th) x.join();
  pl.seg_begin.assign(nb + 1, 0);
  for (int64_t b = 0; b < nb; ++b) pl.seg_begin[b + 1] = pl.seg_begin[b] + blk_nseg[b];
  for (int t = 0; t < T; ++t) {
    pl.seg_col.insert(pl.seg_col.end(), t_seg_col[t].begin(), t_seg_col[t].end());
    pl.seg_off.insert(pl.seg_off.end(), t_seg_off[t].begin(), t_seg_off[t].end());
  }
  int32_t mw = 0;
  for (int64_t b = 0; b < nb; ++b) {
    mw = std::max(mw, pl.blkW[b]);
    pl.fallback += pl.blkW[b] < 0;
  }
  pl.maxW = std::max(mw, 1);
  if (pl.fallback * 2 > nb) return;  // windows do not pay for this matrix: keep the plain kernels
  pl.win = true;
  if (!want_vi) return;
  pl.doff.assign(nb, 0);
  for (int64_t b = 0; b < nb; ++b) {
    pl.doff[b] = (int32_t)pl.dict.size();
    if (pl.blk_dn[b] < 0) continue;
    pl.dict.insert(pl.dict.end(), t_dict[b].begin(), t_dict[b].end());
    ++pl.vi_blocks;
    pl.vi_nnz += rp[std::min<int64_t>((b + 1) * RB, nrows)] - rp[b * RB];
  }
  for (int64_t e : t_wide) pl.wide_nnz += e;
  if (!(pl.vi_blocks * 2 > nb && pl.dict.size() < 2000000000ull)) return;  // worthwhile only if most blocks are coded
  pl.vi = true;
  int64_t longest = 0;
  for (int64_t r = 0; r < nrows; ++r) longest = std::max(longest, rp[r + 1] - rp[r]);
  if (RB > 250 || longest > 65535) return;
  // class-sorted batches of 4 rows per block
  const int stride = (RB + 3) / 4 + kVibMaxClass + 2;
  pl.stride = stride;
  pl.tab.assign((size_t)nb * stride * 4, 0);
  pl.cnt.assign(nb, 0);
  std::vector<std::thread> th2;
  for (int t = 0; t < T; ++t)
    th2.emplace_back([&, t]() {
      std::vector<int> ids[kVibMaxClass + 2];
      for (int64_t b = nb * t / T; b < nb * (t + 1) / T; ++b) {
        const int64_t r0 = b * RB, r1 = std::min<int64_t>(r0 + RB, nrows);
        for (auto &v : ids) v.clear();
        for (int64_t r = r0; r < r1; ++r) {
          const int64_t len = rp[r + 1] - rp[r];
          const int64_t cls = (len + 63) / 64;
          ids[cls > kVibMaxClass ? kVibMaxClass + 1 : cls].push_back((int)(r - r0));
        }
        int nbt = 0;
        for (int cls = 0; cls <= kVibMaxClass + 1; ++cls)
          for (size_t q = 0; q < ids[cls].size(); q += 4) {
            uint64_t *dst = &pl.tab[((size_t)b * stride + nbt++) * 4];
            for (int i = 0; i < 4; ++i) {
              const bool real = q + i < ids[cls].size();
              const int id = ids[cls][real ? q + i : q];  // filler: repeat the batch's first row
              const uint64_t ks = (uint64_t)(rp[r0 + id] - rp[r0]);
              const uint64_t len = (uint64_t)(rp[r0 + id + 1] - rp[r0 + id]);
              dst[i] = ks | (len << 32) | ((uint64_t)(real ? id : 0xff) << 48) | ((uint64_t)cls << 56);
            }
          }
        pl.cnt[b] = nbt;
      }
    });
  for (auto &x : th2) x.join();
}

template <class T>
static int upload_vec(alfd_ctx *ctx, DevCsr &m, T **dst, const std::vector<T> &v) {
  RC(csr_alloc(ctx, m, dst, (int64_t)v.size()));
  HIPC(hipMemcpyAsync(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return ALFD_OK;
}

static int build_window(alfd_ctx *ctx, DevCsr &m, const int64_t *rp, const int32_t *col, const double *val,
                        bool slot_is_user, bool skip_vi = false) {
  WindowParams wp;
  wp.RB_long = ctx->win_RB;
  wp.short_scale = ctx->win_short_scale;
  wp.RB_vi = ctx->win_RB_vi;
  wp.maxW = ctx->win_maxW;
  wp.gap = ctx->win_gap;
  // skip_vi: the batch-major form already holds the matrix with dictionaries of its own; the window plan then only
  // provides the 16-bit columns of the general 10 B/nnz kernel (a third of the planning time at N = 74)
  wp.want_vi = !skip_vi && ctx->win_vi && (slot_is_user || ctx->vi_levels);
  WindowPlan pl;
  plan_window(m.nrows, m.L, rp, col, val, wp, pl);
  if (!pl.win) return ALFD_OK;
  m.win_RB = pl.RB;
  m.win_maxW = pl.maxW;
  m.win_nblocks = pl.nb;
  m.win_fallback_blocks = pl.fallback;
  m.win_nseg = (int64_t)pl.seg_col.size();
  RC(upload_vec(ctx, m, &m.lcol, pl.lcol));
  RC(upload_vec(ctx, m, &m.blk_seg_begin, pl.seg_begin));
  RC(upload_vec(ctx, m, &m.blk_W, pl.blkW));
  RC(upload_vec(ctx, m, &m.seg_col, pl.seg_col));
  RC(upload_vec(ctx, m, &m.seg_off, pl.seg_off));
  HIPC(hipStreamSynchronize(ctx->stream));
  m.win = true;
  if (!pl.vi) return ALFD_OK;
  RC(upload_vec(ctx, m, &m.vidx, pl.vidx));
  if (pl.wide_nnz > 0) RC(upload_vec(ctx, m, &m.vidw, pl.vidw));
  RC(upload_vec(ctx, m, &m.blk_dict_off, pl.doff));
  RC(upload_vec(ctx, m, &m.blk_dict_n, pl.blk_dn));
  RC(upload_vec(ctx, m, &m.dict, pl.dict));
  if (pl.stride > 0) {
    RC(upload_vec(ctx, m, &m.vib_tab, pl.tab));
    RC(upload_vec(ctx, m, &m.vib_cnt, pl.cnt));
    m.vib_stride = pl.stride;
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  m.vi = true;
  m.vi_blocks = pl.vi_blocks;
  m.vi_nnz = pl.vi_nnz;
  m.vi_wide_nnz = pl.wide_nnz;
  m.vi_dict_total = (int64_t)pl.dict.size();
  if (ctx->cfg.log_level > 0)
    std::fprintf(stderr, "[alfd] value-indexed %s matrix: %lld rows, %lld of %lld blocks coded, dict %zu\n",
                 slot_is_user ? "user" : "level", (long long)m.nrows, (long long)pl.vi_blocks, (long long)pl.nb,
                 pl.dict.size());
  return ALFD_OK;
}


// ---- batch-major format (kernels_vs.hpp): host planning, then upload.
// Row blocks are lists of rows: runs of RB rows of the numbering, or the caller's blocks
// (alfd_set_row_blocks: bricks of the mesh graph).  The format is all-or-nothing: every block
// needs an x window of at most maxW slots, at most 256 distinct values, rows of at most
// kVsMaxLen entries; otherwise the matrix keeps the formats of plan_window.
struct VsPlan {
  bool ok = false;
  int64_t nb = 0, nbatch = 0, shared_nnz = 0, nb_interior = 0;
  bool dict_split = false;   // blocks were halved for their dictionaries (or row counts): wide codes may pay
  int rbs = 0, stride = 0;
  int32_t maxW = 0;
  std::vector<int32_t> blkW, seg_begin, seg_col, seg_off, doff, dn, cnt;
  std::vector<double> dict;
  std::vector<int64_t> sb;
  std::vector<uint8_t> stream;
  std::vector<uint64_t> tab;
  int wide = 0;       // 1: 10-bit codes / 11-bit window columns (VsFmt<1>)
  int64_t nb_in = 0;  // blocks before the dictionary limit halved any
};

constexpr int kVsBatchRows = 16;   // rows a batch descriptor names (16 x uint64 = 128 bytes)
constexpr int kVsRefineW = 2048;   // window a block halved for its window may keep (slots): 8 workgroups per CU; B at N = 74: 0.129 ms (4096: 0.138, 1536: 0.170, general kernel 0.414)
struct VsBatch {
  int cls, nreal, id[kVsBatchRows];   // plain: up to 4 rows; shared: up to 16
  bool shared;                        // the rows are translates of one another: one template + a window shift per row
  int32_t shift[kVsBatchRows];        // window slots, relative to row id[0]
};

// x window of a row block: maximal runs of used columns, gaps shorter than GAP bridged, cut into pieces
// of at most 64 slots (one wave-load each).  slot(c) = window slot of column c.
struct VsWindow {
  int32_t clo = 0, W = 0, nseg = 0;
  // the window as runs of consecutive columns: run i holds columns run_col[i] .. run_col[i] + run_len[i] - 1 in the
  // slots run_slot[i] ..  (sparse: a block of an operator whose rows reach far -- the divergence block of a Taylor-Hood
  // pair -- spans 300 k columns with 2 k of them used; dense per-column tables made planning such operators cost more
  // than planning A)
  std::vector<int32_t> run_col, run_slot, run_len;
  std::vector<uint8_t> mark;       // scratch, all zero between calls, grown on demand
  std::vector<int32_t> ucol;       // scratch
  int32_t slot(int32_t c) const {  // window slot of a column of the block
    size_t lo = 0, hi = run_col.size();
    while (hi - lo > 1) {
      const size_t mid = (lo + hi) / 2;
      if (run_col[mid] <= c) lo = mid;
      else hi = mid;
    }
    return run_slot[lo] + (c - run_col[lo]);
  }
};
static bool vs_window(const std::vector<int32_t> &rows, const int64_t *rp, const int32_t *col, int GAP, int maxW,
                      VsWindow &w, std::vector<int32_t> *seg_col, std::vector<int32_t> *seg_off) {
  w.ucol.clear();
  for (int32_t r : rows)
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
      const int32_t c = col[k];
      if ((size_t)c >= w.mark.size()) w.mark.resize((size_t)c + 1 + w.mark.size() / 2, 0);
      if (!w.mark[c]) {
        w.mark[c] = 1;
        w.ucol.push_back(c);
      }
    }
  for (int32_t c : w.ucol) w.mark[c] = 0;
  std::sort(w.ucol.begin(), w.ucol.end());
  w.clo = w.ucol.empty() ? 0 : w.ucol.front();
  w.run_col.clear();
  w.run_slot.clear();
  w.run_len.clear();
  w.W = 0;
  w.nseg = 0;
  size_t i = 0;
  while (i < w.ucol.size()) {
    const int32_t c = w.ucol[i];
    int32_t last = c;
    size_t j = i + 1;
    while (j < w.ucol.size() && w.ucol[j] - last <= GAP) last = w.ucol[j++];   // gaps shorter than GAP are bridged
    const int32_t len = last - c + 1;
    for (int32_t q0 = 0; q0 < len; q0 += 64) {   // pieces of at most 64 slots (one wave-load each)
      if (seg_col) {
        seg_col->push_back(c + q0);
        seg_off->push_back(w.W + q0);
      }
      ++w.nseg;
    }
    w.run_col.push_back(c);
    w.run_slot.push_back(w.W);
    w.run_len.push_back(len);
    w.W += len;
    if (w.W > 4096 && !seg_col) return false;   // hopeless already (the planner's callers pass no segment lists when probing)
    i = j;
  }
  if (w.run_col.empty()) {
    w.run_col.push_back(0);
    w.run_slot.push_back(0);
    w.run_len.push_back(0);
  }
  return w.W <= maxW && w.W <= 4096;
}

// Batches of one block.  Rows are grouped by chunk count (class); inside a class, rows that are
// translates of one another -- same length, same values entry by entry, window columns differing by one
// constant -- form SHARED batches (one stored template, a shift per row: what a uniform mesh gives for
// the rows of one node type inside a brick); the rest form plain batches, longest rows first.
static bool vs_batches(const std::vector<int32_t> &rows, const int64_t *rp, const int32_t *col, const double *val,
                       const VsWindow &w, std::vector<VsBatch> &out, bool share = true) {
  const int nr = (int)rows.size();
  out.clear();
  std::vector<int64_t> len(nr);
  std::vector<uint64_t> key(nr);
  std::vector<int32_t> first(nr, 0);
  for (int i = 0; i < nr; ++i) {
    const int64_t k0 = rp[rows[i]], n = rp[rows[i] + 1] - k0;
    if (n > kVsMaxLen) return false;
    len[i] = n;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    const int32_t p0 = n ? w.slot(col[k0]) : 0;
    first[i] = p0;
    for (int64_t k = 0; k < n; ++k) {
      uint64_t bits;
      std::memcpy(&bits, &val[k0 + k], 8);
      h = (h ^ bits) * 0xff51afd7ed558ccdull;
      h = (h ^ (uint64_t)(uint32_t)(w.slot(col[k0 + k]) - p0)) * 0xc4ceb9fe1a85ec53ull;
      h ^= h >> 29;
    }
    key[i] = h;
  }
  auto same = [&](int a, int b) {   // exact test behind the hash
    if (len[a] != len[b]) return false;
    const int64_t ka = rp[rows[a]], kb = rp[rows[b]];
    for (int64_t k = 0; k < len[a]; ++k) {
      if (std::memcmp(&val[ka + k], &val[kb + k], 8) != 0) return false;
      if (w.slot(col[ka + k]) - first[a] != w.slot(col[kb + k]) - first[b]) return false;
    }
    return true;
  };
  std::vector<int> ids, plain;
  for (int cls = 0; cls <= 6; ++cls) {
    ids.clear();
    plain.clear();
    for (int i = 0; i < nr; ++i)
      if ((int)((len[i] + 63) / 64) == cls) ids.push_back(i);
    if (cls == 0) {
      plain = ids;
    } else {
      std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return key[a] < key[b]; });
      size_t g0 = 0;
      while (g0 < ids.size()) {
        size_t g1 = g0 + 1;
        while (share && g1 < ids.size() && key[ids[g1]] == key[ids[g0]] && same(ids[g0], ids[g1])) ++g1;
        size_t q = g0;
        while (g1 - q >= 2) {   // 2..16 rows per shared batch (never a lone leftover if it can be avoided)
          const size_t left = g1 - q;
          const size_t take = left == kVsBatchRows + 1 ? kVsBatchRows / 2 + 1 : std::min<size_t>(kVsBatchRows, left);
          VsBatch bt{cls, 0, {}, true, {}};
          for (size_t i = q; i < q + take; ++i) {
            bt.id[bt.nreal] = ids[i];
            bt.shift[bt.nreal] = first[ids[i]] - first[ids[q]];
            ++bt.nreal;
          }
          out.push_back(bt);
          q += take;
        }
        for (; q < g1; ++q) plain.push_back(ids[q]);
        g0 = g1;
      }
    }
    // plain batches: longest first, so that the rows of a batch have equal or close remainders in the last chunk
    std::stable_sort(plain.begin(), plain.end(), [&](int a, int b) { return len[a] != len[b] ? len[a] > len[b] : a < b; });
    for (size_t q = 0; q < plain.size(); q += 4) {
      VsBatch bt{cls, 0, {}, false, {}};
      for (size_t i = q; i < std::min(q + 4, plain.size()); ++i) bt.id[bt.nreal++] = plain[i];
      out.push_back(bt);
    }
  }
  return true;
}

// 16-byte units a batch occupies.  Plain: 12 bytes (4 row slots x 24 bits) per lane and chunk; shared: 4 bytes
// (one 24-bit field in a dword).  Of the last chunk only the lanes below the longest remainder (rounded up to 4).
static int64_t vs_batch_units(const VsBatch &q, const std::vector<int32_t> &rows, const int64_t *rp) {
  if (q.cls == 0) return 0;
  const int64_t full = 64 * (q.cls - 1);
  int64_t mr = 1;
  for (int i = 0; i < q.nreal; ++i) mr = std::max(mr, rp[rows[q.id[i]] + 1] - rp[rows[q.id[i]]] - full);
  const int64_t lanes = full + (mr + 3) / 4 * 4;
  return q.shared ? lanes / 4 : 3 * lanes / 4;
}

// Blocks whose entries take more than kVsMaxDict distinct values (9-bit codes), or that list more than
// kVsMaxRows rows, are halved until they fit; false if a single row does not fit.
static bool vs_refine_blocks(int64_t nrows, const int64_t *rp, const double *val, int RB, int64_t nb_in,
                             const int64_t *bptr, const int32_t *brows, std::vector<int64_t> &optr,
                             std::vector<int32_t> &orows, int max_rows = kVsMaxRows, int max_dict = kVsMaxDict) {
  const bool nat = bptr == nullptr;
  const int64_t nb = nat ? (nrows + RB - 1) / RB : nb_in;
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::vector<int64_t>> t_sizes(T);
  std::atomic<bool> bad(false);
  orows.resize(nrows);
  if (nat) {
    for (int64_t r = 0; r < nrows; ++r) orows[r] = (int32_t)r;
  } else {
    std::copy(brows, brows + nrows, orows.begin());
  }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      constexpr int kTab = 4096;   // open addressing over the 64-bit patterns; stamps avoid clearing
      std::vector<uint64_t> keys(kTab);
      std::vector<uint32_t> stamp(kTab, 0);
      uint32_t gen = 0;
      std::vector<std::pair<int64_t, int64_t>> stack;
      for (int64_t b = nb * t / T; b < nb * (t + 1) / T && !bad; ++b) {
        const int64_t lo = nat ? b * RB : bptr[b], hi = nat ? std::min<int64_t>((b + 1) * RB, nrows) : bptr[b + 1];
        stack.clear();
        stack.emplace_back(lo, hi);
        while (!stack.empty() && !bad) {   // depth-first, left half first: pieces come out in row-list order
          const auto [a, e] = stack.back();
          stack.pop_back();
          bool fits = e - a <= max_rows;
          if (fits) {
            ++gen;
            int distinct = 0;
            for (int64_t i = a; i < e && fits; ++i) {
              const int32_t r = orows[i];
              for (int64_t k = rp[r]; k < rp[r + 1] && fits; ++k) {
                uint64_t bits;
                std::memcpy(&bits, &val[k], 8);
                uint32_t h = (uint32_t)((bits * 0x9E3779B97F4A7C15ull) >> 52);
                while (stamp[h] == gen && keys[h] != bits) h = (h + 1) & (kTab - 1);
                if (stamp[h] != gen) {
                  stamp[h] = gen;
                  keys[h] = bits;
                  fits = ++distinct <= max_dict;
                }
              }
            }
          }
          if (fits) {
            t_sizes[t].push_back(e - a);
          } else if (e - a <= 1) {
            bad = true;
          } else {
            const int64_t mid = a + (e - a) / 2;
            stack.emplace_back(mid, e);
            stack.emplace_back(a, mid);
          }
        }
      }
    });
  for (auto &x : th) x.join();
  if (bad) return false;
  optr.assign(1, 0);
  for (int t = 0; t < T; ++t)
    for (int64_t sz : t_sizes[t]) optr.push_back(optr.back() + sz);
  // blocks that had to be cut to a quarter of their size on average: the values do not repeat
  // enough for this format (a window per handful of rows costs more than the codes save)
  return (int64_t)optr.size() - 1 <= 4 * nb + 16;
}

static void plan_vs(int64_t nrows, const int64_t *rp, const int32_t *col, const double *val, int RB, int maxW,
                    int GAP, int64_t nb_in, const int64_t *bptr_in, const int32_t *brows_in, VsPlan &pl,
                    bool share = true, int wide = 0, int64_t n_local_cols = -1) {
  std::vector<int64_t> r_ptr;
  std::vector<int32_t> r_rows;
  const int max_dict = wide ? VsFmt<1>::kMaxDict : VsFmt<0>::kMaxDict, code_shift = wide ? VsFmt<1>::kCodeShift : VsFmt<0>::kCodeShift;
  maxW = std::min(maxW, wide ? VsFmt<1>::kMaxSlots : VsFmt<0>::kMaxSlots);
  pl.wide = wide;
  pl.nb_in = nb_in > 0 ? nb_in : (nrows + RB - 1) / RB;
  if (nrows == 0 || !vs_refine_blocks(nrows, rp, val, RB, nb_in, bptr_in, brows_in, r_ptr, r_rows, kVsMaxRows, max_dict)) return;
  pl.dict_split = (int64_t)r_ptr.size() - 1 > pl.nb_in;
  if (n_local_cols >= 0) {
    // partitioned operator: the blocks that read no halo column (columns >= n_local_cols) first, so that they can run
    // while the halo is still on its way (spmv_m); the order of the blocks is free, every row names its own result
    const int64_t nb0 = (int64_t)r_ptr.size() - 1;
    std::vector<uint8_t> bnd(nb0, 0);
    for (int64_t b = 0; b < nb0; ++b)
      for (int64_t i = r_ptr[b]; i < r_ptr[b + 1] && !bnd[b]; ++i) {
        const int32_t r = r_rows[i];
        for (int64_t k = rp[r]; k < rp[r + 1] && !bnd[b]; ++k) bnd[b] = col[k] >= n_local_cols;   // halo columns sit anywhere in a row
      }
    std::vector<int64_t> np(1, 0);
    std::vector<int32_t> nr;
    nr.reserve(r_rows.size());
    for (int pass = 0; pass < 2; ++pass)
      for (int64_t b = 0; b < nb0; ++b)
        if (bnd[b] == pass) {
          nr.insert(nr.end(), r_rows.begin() + r_ptr[b], r_rows.begin() + r_ptr[b + 1]);
          np.push_back((int64_t)nr.size());
        }
    r_ptr.swap(np);
    r_rows.swap(nr);
  }
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::vector<VsBatch>> batches;
  std::vector<int64_t> blk_units;
  std::atomic<bool> bad(false), too_wide(false);
  std::atomic<int64_t> n_shared_nnz(0);
  int64_t nb = 0;
  const int64_t *bptr = nullptr;
  const int32_t *brows = nullptr;
  for (int attempt = 0; attempt < 3; ++attempt) {
    if (attempt == 1) {
      // cheap first: how often do a few sample blocks have to be halved?  Split every block that often, untested (the
      // pass below tests all of them; what still does not fit goes through the exact refinement of attempt 2)
      if (!too_wide) return;
      int k = 0;
      std::vector<int32_t> rows;
      VsWindow w;
      for (int64_t b : {(int64_t)0, nb / 2, nb - 1}) {
        int64_t len = bptr[b + 1] - bptr[b];
        int kb = 0;
        for (; len > 1; ++kb, len = (len + 1) / 2) {
          rows.assign(brows + bptr[b], brows + bptr[b] + len);
          if (vs_window(rows, rp, col, GAP, std::min(maxW, kVsRefineW), w, nullptr, nullptr)) break;
        }
        k = std::max(k, kb);
      }
      if (k == 0) continue;   // the samples fit: straight to the exact refinement
      std::vector<int64_t> np(1, 0);
      for (int64_t b = 0; b < nb; ++b) {
        const int64_t a = bptr[b], len = bptr[b + 1] - a, pieces = std::min<int64_t>((int64_t)1 << k, std::max<int64_t>(len, 1));
        for (int64_t q = 1; q <= pieces; ++q)
          if (a + len * q / pieces > np.back()) np.push_back(a + len * q / pieces);
      }
      r_ptr.swap(np);
      bad = false;
      too_wide = false;
      n_shared_nnz = 0;
    }
    if (attempt == 2) {
      // A block's x window did not fit the LDS budget (operators whose rows reach far, e.g. the divergence block B of a
      // Taylor-Hood pair: 96 pressure rows touch 15 000 velocity columns): halve such blocks until their windows fit.
      // Only paid by operators that need it -- the first attempt is the plan of everything else.
      if (!too_wide) return;
      std::vector<std::vector<int64_t>> t_sizes(T);
      std::atomic<bool> hopeless(false);
      const int64_t nb0 = nb;
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t]() {
          std::vector<int32_t> rows;
          VsWindow w;
          std::vector<std::pair<int64_t, int64_t>> stack;
          for (int64_t b = nb0 * t / T; b < nb0 * (t + 1) / T && !hopeless; ++b) {
            stack.clear();
            stack.emplace_back(bptr[b], bptr[b + 1]);
            while (!stack.empty() && !hopeless) {   // depth-first, left half first: pieces come out in row-list order
              const auto [a, e] = stack.back();
              stack.pop_back();
              rows.assign(brows + a, brows + e);
              if (vs_window(rows, rp, col, GAP, std::min(maxW, kVsRefineW), w, nullptr, nullptr)) {
                t_sizes[t].push_back(e - a);
              } else if (e - a <= 1) {
                hopeless = true;
              } else {
                const int64_t mid = a + (e - a) / 2;
                stack.emplace_back(mid, e);
                stack.emplace_back(a, mid);
              }
            }
          }
        });
      for (auto &x : th) x.join();
      if (hopeless) return;
      std::vector<int64_t> np(1, 0);
      for (int t = 0; t < T; ++t)
        for (int64_t sz : t_sizes[t]) np.push_back(np.back() + sz);
      r_ptr.swap(np);
      bad = false;
      n_shared_nnz = 0;
    }
  bptr = r_ptr.data();
  brows = r_rows.data();
  nb = (int64_t)r_ptr.size() - 1;
  if (nb == 0 || nb > 2147483000LL) return;
  // pass 1: windows, batches (kept), stream extents
  batches.assign(nb, std::vector<VsBatch>());
  blk_units.assign(nb, 0);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        std::vector<int32_t> rows;
        VsWindow w;
        int64_t sh = 0;
        for (int64_t b = nb * t / T; b < nb * (t + 1) / T && !bad; ++b) {
          rows.assign(brows + bptr[b], brows + bptr[b + 1]);
          if (rows.size() <= (size_t)kVsMaxRows && !vs_window(rows, rp, col, GAP, maxW, w, nullptr, nullptr)) {
            too_wide = true;
            bad = true;
            break;
          }
          if (rows.size() > (size_t)kVsMaxRows || !vs_batches(rows, rp, col, val, w, batches[b], share)) {
            bad = true;
            break;
          }
          int64_t e = 0;
          for (const VsBatch &q : batches[b]) {
            e += vs_batch_units(q, rows, rp);
            if (q.shared)
              for (int i = 0; i < q.nreal; ++i) sh += rp[rows[q.id[i]] + 1] - rp[rows[q.id[i]]];
          }
          blk_units[b] = e;
        }
        n_shared_nnz += sh;
      });
    for (auto &x : th) x.join();
  }
    if (!bad) break;
  }
  if (bad) return;
  pl.nb = nb;
  pl.nb_interior = 0;
  if (n_local_cols >= 0) {   // leading blocks without a halo column (splits for the window keep the order)
    bool interior = true;
    for (int64_t b = 0; b < nb && interior; ++b) {
      for (int64_t i = bptr[b]; i < bptr[b + 1] && interior; ++i) {
        const int32_t r = brows[i];
        for (int64_t k = rp[r]; k < rp[r + 1] && interior; ++k) interior = col[k] < n_local_cols;
      }
      if (interior) pl.nb_interior = b + 1;
    }
  }
  pl.shared_nnz = n_shared_nnz;
  pl.sb.assign(nb, 0);
  int64_t tot = 0;
  int maxb = 1, maxr = 1;
  for (int64_t b = 0; b < nb; ++b) {
    pl.sb[b] = tot;
    tot += 16 * blk_units[b];
    if (blk_units[b] > 0xfffffLL) return;
    maxb = std::max(maxb, (int)batches[b].size());
    maxr = std::max(maxr, (int)(bptr[b + 1] - bptr[b]));
    pl.nbatch += (int64_t)batches[b].size();
  }
  if (maxb > 0xffff) return;
  pl.stride = maxb;
  pl.rbs = maxr;
  pl.stream.assign((size_t)tot + 4096, 0);
  pl.tab.assign((size_t)nb * maxb * kVsBatchRows, 0);
  pl.cnt.assign(nb, 0);
  pl.blkW.assign(nb, 0);
  pl.dn.assign(nb, 0);
  std::vector<int32_t> blk_nseg(nb, 0);
  std::vector<std::vector<double>> t_dict(T);
  std::vector<std::vector<int32_t>> t_seg_col(T), t_seg_off(T), t_doff(T);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        std::vector<int32_t> rows;
        VsWindow w;
        constexpr int kTab = 4096;
        std::vector<uint64_t> keys(kTab);
        std::vector<int16_t> ids(kTab);
        for (int64_t b = nb * t / T; b < nb * (t + 1) / T && !bad; ++b) {
          rows.assign(brows + bptr[b], brows + bptr[b + 1]);
          const std::vector<VsBatch> &bts = batches[b];
          pl.cnt[b] = (int32_t)bts.size();
          t_doff[t].push_back((int32_t)t_dict[t].size());
          vs_window(rows, rp, col, GAP, maxW, w, &t_seg_col[t], &t_seg_off[t]);
          pl.blkW[b] = w.W;
          blk_nseg[b] = w.nseg;
          // dictionary (<= kVsMaxDict bit patterns) and the batch-major stream
          std::fill(ids.begin(), ids.end(), (int16_t)-1);
          const size_t d0 = t_dict[t].size();
          uint8_t *sp = pl.stream.data() + pl.sb[b];
          uint32_t eoff = 0;  // 16-byte units from the block's first byte
          auto field_of = [&](int64_t k) -> int64_t {   // (code << 15) | (window slot << 3) of CSR entry k; -1: dictionary full
            uint64_t bits;
            std::memcpy(&bits, &val[k], 8);
            uint32_t h = (uint32_t)((bits * 0x9E3779B97F4A7C15ull) >> 52);
            for (;;) {
              if (ids[h] < 0) {
                if (t_dict[t].size() - d0 == (size_t)max_dict) return -1;
                keys[h] = bits;
                ids[h] = (int16_t)(t_dict[t].size() - d0);
                t_dict[t].push_back(val[k]);
                break;
              }
              if (keys[h] == bits) break;
              h = (h + 1) & (kTab - 1);
            }
            return ((int64_t)ids[h] << code_shift) | ((int64_t)w.slot(col[k]) << 3);
          };
          for (size_t q = 0; q < bts.size() && !bad; ++q) {
            const VsBatch &bt = bts[q];
            uint8_t *fb = sp + 16 * (size_t)eoff;
            uint64_t *dst = &pl.tab[((size_t)b * maxb + q) * kVsBatchRows];
            for (int i = 0; i < kVsBatchRows && !bad; ++i) {
              const int src = i < bt.nreal ? i : 0;     // fillers repeat the batch's first row
              const int32_t r = rows[bt.id[src]];
              const int64_t k0 = rp[r], n = rp[r + 1] - k0;
              if (bt.shared ? i == 0 : i < 4)
                for (int64_t k = 0; k < n; ++k) {   // entry k: chunk k / 64, lane k % 64 (row slot i of a plain batch)
                  const int64_t f = field_of(k0 + k);
                  if (f < 0) { bad = true; break; }
                  if (bt.shared) {
                    const uint32_t f32 = (uint32_t)f;
                    std::memcpy(fb + 256 * (size_t)(k / 64) + 4 * (size_t)(k % 64), &f32, 4);
                  } else {
                    uint8_t *cell = fb + 768 * (size_t)(k / 64) + 12 * (size_t)(k % 64) + 3 * i;
                    cell[0] = (uint8_t)f;
                    cell[1] = (uint8_t)(f >> 8);
                    cell[2] = (uint8_t)(f >> 16);
                  }
                }
              // low dword: offset | count | class; rows 1.. of a shared batch: their window shift in bytes (signed)
              const uint64_t low = (bt.shared && i > 0) ? (uint64_t)(uint32_t)(i < bt.nreal ? bt.shift[i] * 8 : 0)
                                                        : ((uint64_t)eoff | ((uint64_t)n << 20) | ((uint64_t)bt.cls << 29));
              dst[i] = low | ((uint64_t)(i < bt.nreal ? (uint32_t)r : 0xffffffffu) << 32);
            }
            if (bt.shared) dst[0] |= 1ull << 63;
            eoff += (uint32_t)vs_batch_units(bt, rows, rp);
          }
          pl.dn[b] = (int32_t)(t_dict[t].size() - d0);
        }
      });
    for (auto &x : th) x.join();
  }
  if (bad) return;
  pl.seg_begin.assign(nb + 1, 0);
  for (int64_t b = 0; b < nb; ++b) pl.seg_begin[b + 1] = pl.seg_begin[b] + blk_nseg[b];
  pl.doff.reserve(nb);
  for (int t = 0; t < T; ++t) {
    const int32_t base = (int32_t)pl.dict.size();
    for (int32_t o : t_doff[t]) pl.doff.push_back(base + o);
    pl.dict.insert(pl.dict.end(), t_dict[t].begin(), t_dict[t].end());
    pl.seg_col.insert(pl.seg_col.end(), t_seg_col[t].begin(), t_seg_col[t].end());
    pl.seg_off.insert(pl.seg_off.end(), t_seg_off[t].begin(), t_seg_off[t].end());
  }
  if (pl.dict.size() > 2000000000ull) return;
  int32_t mw = 1;
  for (int64_t b = 0; b < nb; ++b) mw = std::max(mw, pl.blkW[b]);
  pl.maxW = mw;
  pl.ok = true;
}

// ---- the same idea for SHORT rows (canonical L = 8, 16 or 32 lanes per row: Q1 stencils, the level operators of
// scalar problems).  A 64-lane wave holds G = 64 / L rows side by side, a batch is ONE stored template row (a dword
// per entry: 9-bit value code, 12-bit window column, ready-shifted) shared by up to 4 G translate rows -- on a
// uniform mesh every interior row of a run is a translate of its neighbour, so the matrix stream all but vanishes
// (27-point Laplace: 12 B/nnz CSR -> ~0.6 B/nnz incl. descriptors) and what is left is one LDS gather and one fma
// per entry.  Rows without a translate partner are batches of one.  Descriptor per batch (uint64 words):
//   [0] eb (20 bits, 16-byte units) | entry count (9) | class = ceil(count / L) (3) | rows in the batch (32)
//   [1 + i] global row (32, low) | window shift in bytes (32, high, signed)      i = 0 .. 4 G - 1
constexpr int kVssMaxClass = 4;   // rows of at most 4 L entries
constexpr int kVssMaxRows = 512;  // rows per block
struct VssBatch {
  int cls, nreal;
  std::vector<int> id;
  std::vector<int32_t> shift;
};

static void plan_vss(int64_t nrows, int L, const int64_t *rp, const int32_t *col, const double *val, int RB, int maxW,
                     int GAP, VsPlan &pl) {
  const int RBb = 4 * (64 / L);   // rows per batch
  std::vector<int64_t> r_ptr;
  std::vector<int32_t> r_rows;
  if (nrows == 0 || !vs_refine_blocks(nrows, rp, val, RB, 0, nullptr, nullptr, r_ptr, r_rows, kVssMaxRows)) return;
  {
    // blocks whose x window would not fit (operators whose rows reach far: the restriction from the Q2 grid, 384 coarse
    // rows touch 7 000 fine columns): how often do three sample blocks have to be halved?  Every block is split that often.
    int k = 0;
    std::vector<int32_t> rows;
    VsWindow w;
    const int64_t nb0 = (int64_t)r_ptr.size() - 1;
    for (int64_t b : {(int64_t)0, nb0 / 2, nb0 - 1}) {
      int64_t len = r_ptr[b + 1] - r_ptr[b];
      int kb = 0;
      for (; len > 1; ++kb, len = (len + 1) / 2) {
        rows.assign(r_rows.begin() + r_ptr[b], r_rows.begin() + r_ptr[b] + len);
        if (vs_window(rows, rp, col, GAP, maxW, w, nullptr, nullptr)) break;
      }
      k = std::max(k, kb);
    }
    if (k > 0) {
      std::vector<int64_t> np(1, 0);
      for (int64_t b = 0; b < nb0; ++b) {
        const int64_t a = r_ptr[b], len = r_ptr[b + 1] - a, pieces = std::min<int64_t>((int64_t)1 << k, std::max<int64_t>(len, 1));
        for (int64_t q = 1; q <= pieces; ++q)
          if (a + len * q / pieces > np.back()) np.push_back(a + len * q / pieces);
      }
      r_ptr.swap(np);
    }
  }
  const int64_t *bptr = r_ptr.data();
  const int32_t *brows = r_rows.data();
  const int64_t nb = (int64_t)r_ptr.size() - 1;
  if (nb == 0 || nb > 2147483000LL) return;
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::vector<VssBatch>> batches(nb);
  std::vector<int64_t> blk_units(nb, 0);
  std::atomic<bool> bad(false);
  std::atomic<int64_t> n_shared_nnz(0);
  auto units_of = [&](const VssBatch &q) { return (int64_t)(q.cls * L + 3) / 4; };   // cls * L dwords, 16-byte units
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        std::vector<int32_t> rows, first;
        std::vector<int64_t> len;
        std::vector<uint64_t> key;
        std::vector<int> ids;
        VsWindow w;
        int64_t sh = 0;
        for (int64_t b = nb * t / T; b < nb * (t + 1) / T && !bad; ++b) {
          rows.assign(brows + bptr[b], brows + bptr[b + 1]);
          if (!vs_window(rows, rp, col, GAP, maxW, w, nullptr, nullptr)) { bad = true; break; }
          const int nr = (int)rows.size();
          len.assign(nr, 0);
          key.assign(nr, 0);
          first.assign(nr, 0);
          for (int i = 0; i < nr && !bad; ++i) {
            const int64_t k0 = rp[rows[i]], n = rp[rows[i] + 1] - k0;
            if (n > (int64_t)kVssMaxClass * L) { bad = true; break; }
            len[i] = n;
            uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
            const int32_t p0 = n ? w.slot(col[k0]) : 0;
            first[i] = p0;
            for (int64_t k = 0; k < n; ++k) {
              uint64_t bits;
              std::memcpy(&bits, &val[k0 + k], 8);
              h = (h ^ bits) * 0xff51afd7ed558ccdull;
              h = (h ^ (uint64_t)(uint32_t)(w.slot(col[k0 + k]) - p0)) * 0xc4ceb9fe1a85ec53ull;
              h ^= h >> 29;
            }
            key[i] = h;
          }
          if (bad) break;
          auto same = [&](int a, int c) {
            if (len[a] != len[c]) return false;
            const int64_t ka = rp[rows[a]], kc = rp[rows[c]];
            for (int64_t k = 0; k < len[a]; ++k) {
              if (std::memcmp(&val[ka + k], &val[kc + k], 8) != 0) return false;
              if (w.slot(col[ka + k]) - first[a] != w.slot(col[kc + k]) - first[c]) return false;
            }
            return true;
          };
          std::vector<VssBatch> &out = batches[b];
          for (int cls = 0; cls <= kVssMaxClass; ++cls) {
            ids.clear();
            for (int i = 0; i < nr; ++i)
              if ((int)((len[i] + L - 1) / L) == cls) ids.push_back(i);
            std::stable_sort(ids.begin(), ids.end(), [&](int a, int c) { return key[a] < key[c]; });
            size_t g0 = 0;
            while (g0 < ids.size()) {
              size_t g1 = g0 + 1;
              while (cls > 0 && g1 < ids.size() && key[ids[g1]] == key[ids[g0]] && same(ids[g0], ids[g1])) ++g1;
              if (cls == 0) g1 = ids.size();   // empty rows: any number per batch
              for (size_t q = g0; q < g1; q += RBb) {
                VssBatch bt{cls, 0, {}, {}};
                for (size_t i = q; i < std::min(q + (size_t)RBb, g1); ++i) {
                  bt.id.push_back(ids[i]);
                  bt.shift.push_back(first[ids[i]] - first[ids[q]]);
                  ++bt.nreal;
                }
                if (bt.nreal > 1) sh += (int64_t)bt.nreal * len[bt.id[0]];
                out.push_back(std::move(bt));
              }
              g0 = g1;
            }
          }
          int64_t e = 0;
          for (const VssBatch &q : out) e += units_of(q);
          blk_units[b] = e;
        }
        n_shared_nnz += sh;
      });
    for (auto &x : th) x.join();
  }
  if (bad) return;
  pl.nb = nb;
  pl.shared_nnz = n_shared_nnz;
  pl.sb.assign(nb, 0);
  int64_t tot = 0;
  int maxb = 1, maxr = 1;
  for (int64_t b = 0; b < nb; ++b) {
    pl.sb[b] = tot;
    tot += 16 * blk_units[b];
    if (blk_units[b] > 0xfffffLL) return;
    maxb = std::max(maxb, (int)batches[b].size());
    maxr = std::max(maxr, (int)(bptr[b + 1] - bptr[b]));
    pl.nbatch += (int64_t)batches[b].size();
  }
  if (maxb > 0xffff) return;
  const int W64 = 1 + RBb;   // descriptor words per batch
  pl.stride = maxb;
  pl.rbs = maxr;
  pl.stream.assign((size_t)tot + 4096, 0);
  pl.tab.assign((size_t)nb * maxb * W64, 0);
  pl.cnt.assign(nb, 0);
  pl.blkW.assign(nb, 0);
  pl.dn.assign(nb, 0);
  std::vector<int32_t> blk_nseg(nb, 0);
  std::vector<std::vector<double>> t_dict(T);
  std::vector<std::vector<int32_t>> t_seg_col(T), t_seg_off(T), t_doff(T);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        std::vector<int32_t> rows;
        VsWindow w;
        constexpr int kTab = 1024;
        std::vector<uint64_t> keys(kTab);
        std::vector<int16_t> ids(kTab);
        for (int64_t b = nb * t / T; b < nb * (t + 1) / T && !bad; ++b) {
          rows.assign(brows + bptr[b], brows + bptr[b + 1]);
          const std::vector<VssBatch> &bts = batches[b];
          pl.cnt[b] = (int32_t)bts.size();
          t_doff[t].push_back((int32_t)t_dict[t].size());
          vs_window(rows, rp, col, GAP, maxW, w, &t_seg_col[t], &t_seg_off[t]);
          pl.blkW[b] = w.W;
          blk_nseg[b] = w.nseg;
          std::fill(ids.begin(), ids.end(), (int16_t)-1);
          const size_t d0 = t_dict[t].size();
          uint8_t *sp = pl.stream.data() + pl.sb[b];
          uint32_t eoff = 0;
          for (size_t q = 0; q < bts.size() && !bad; ++q) {
            const VssBatch &bt = bts[q];
            uint64_t *dst = &pl.tab[((size_t)b * maxb + q) * W64];
            const int32_t r0 = rows[bt.id[0]];
            const int64_t k0 = rp[r0], n = rp[r0 + 1] - k0;
            for (int64_t k = 0; k < n; ++k) {
              uint64_t bits;
              std::memcpy(&bits, &val[k0 + k], 8);
              uint32_t h = (uint32_t)((bits * 0x9E3779B97F4A7C15ull) >> 54);
              for (;;) {
                if (ids[h] < 0) {
                  if (t_dict[t].size() - d0 == (size_t)kVsMaxDict) { bad = true; break; }
                  keys[h] = bits;
                  ids[h] = (int16_t)(t_dict[t].size() - d0);
                  t_dict[t].push_back(val[k0 + k]);
                  break;
                }
                if (keys[h] == bits) break;
                h = (h + 1) & (kTab - 1);
              }
              if (bad) break;
              const uint32_t f = ((uint32_t)ids[h] << 15) | ((uint32_t)w.slot(col[k0 + k]) << 3);
              std::memcpy(sp + 16 * (size_t)eoff + 4 * (size_t)k, &f, 4);
            }
            dst[0] = (uint64_t)eoff | ((uint64_t)n << 20) | ((uint64_t)bt.cls << 29) | ((uint64_t)bt.nreal << 32);
            for (int i = 0; i < RBb; ++i) {
              const uint32_t row = i < bt.nreal ? (uint32_t)rows[bt.id[i]] : 0xffffffffu;
              const int32_t shb = i < bt.nreal ? bt.shift[i] * 8 : 0;
              dst[1 + i] = (uint64_t)row | ((uint64_t)(uint32_t)shb << 32);
            }
            eoff += (uint32_t)units_of(bt);
          }
          pl.dn[b] = (int32_t)(t_dict[t].size() - d0);
        }
      });
    for (auto &x : th) x.join();
  }
  if (bad) return;
  pl.seg_begin.assign(nb + 1, 0);
  for (int64_t b = 0; b < nb; ++b) pl.seg_begin[b + 1] = pl.seg_begin[b] + blk_nseg[b];
  pl.doff.reserve(nb);
  for (int t = 0; t < T; ++t) {
    const int32_t base = (int32_t)pl.dict.size();
    for (int32_t o : t_doff[t]) pl.doff.push_back(base + o);
    pl.dict.insert(pl.dict.end(), t_dict[t].begin(), t_dict[t].end());
    pl.seg_col.insert(pl.seg_col.end(), t_seg_col[t].begin(), t_seg_col[t].end());
    pl.seg_off.insert(pl.seg_off.end(), t_seg_off[t].begin(), t_seg_off[t].end());
  }
  if (pl.dict.size() > 2000000000ull) return;
  int32_t mw = 1;
  for (int64_t b = 0; b < nb; ++b) mw = std::max(mw, pl.blkW[b]);
  pl.maxW = mw;
  pl.ok = true;
}

static int build_vs(alfd_ctx *ctx, DevCsr &m, int slot, const int64_t *rp, const int32_t *col, const double *val) {
  VsPlan pl;
  const int64_t nlc = (ctx->nranks > 1 && !m.rep) ? (int64_t)m.n_local_cols : -1;   // columns >= nlc are halo columns
  bool hint = slot >= 0 && slot < ALFD_NSLOTS && !ctx->rb_ptr[slot].empty();
  if (hint && ctx->rb_ptr[slot].back() != m.nrows) {
    // the hint was given for a matrix of another size (a re-upload of the slot): it no longer applies
    if (ctx->cfg.log_level > 0)
      std::fprintf(stderr, "[alfd] row-block hint of slot %d dropped: it lists %lld rows, the matrix has %lld\n", slot,
                   (long long)ctx->rb_ptr[slot].back(), (long long)m.nrows);
    ctx->rb_ptr[slot].clear();
    ctx->rb_rows[slot].clear();
    hint = false;
  }
  if (hint) {
    // the hint must list every row exactly once
    const auto &bp = ctx->rb_ptr[slot];
    const auto &br = ctx->rb_rows[slot];
    bool good = bp.front() == 0 && bp.back() == m.nrows && (int64_t)br.size() == m.nrows;
    std::vector<uint8_t> seen(good ? m.nrows : 0, 0);
    for (size_t i = 0; good && i < br.size(); ++i) {
      good = br[i] >= 0 && br[i] < m.nrows && !seen[br[i]];
      if (good) seen[br[i]] = 1;
    }
    for (size_t i = 0; good && i + 1 < bp.size(); ++i) good = bp[i] <= bp[i + 1];
    if (!good) return ctx->err = "alfd_set_row_blocks: the blocks are not a partition of the matrix rows", ALFD_E_INVALID;
    plan_vs(m.nrows, rp, col, val, ctx->vs_RB, ctx->win_maxW, ctx->win_gap, (int64_t)bp.size() - 1, bp.data(),
            br.data(), pl, ctx->vs_share != 0, 0, nlc);
  } else {
    plan_vs(m.nrows, rp, col, val, ctx->vs_RB, ctx->win_maxW, ctx->win_gap, 0, nullptr, nullptr, pl, ctx->vs_share != 0, 0, nlc);
  }
  if (ctx->vs_wide && (!pl.ok || pl.dict_split)) {
    // blocks with more than 512 distinct values had to be halved (or the plan failed): the same blocks with 10-bit
    // codes and 11-bit window columns, kept if the stream gets smaller
    VsPlan pw;
    if (hint)
      plan_vs(m.nrows, rp, col, val, ctx->vs_RB, ctx->win_maxW, ctx->win_gap, (int64_t)ctx->rb_ptr[slot].size() - 1,
              ctx->rb_ptr[slot].data(), ctx->rb_rows[slot].data(), pw, ctx->vs_share != 0, 1, nlc);
    else
      plan_vs(m.nrows, rp, col, val, ctx->vs_RB, ctx->win_maxW, ctx->win_gap, 0, nullptr, nullptr, pw, ctx->vs_share != 0, 1, nlc);
    if (pw.ok && (!pl.ok || pw.stream.size() + 128 * (size_t)pw.nbatch < pl.stream.size() + 128 * (size_t)pl.nbatch)) pl = std::move(pw);
  }
  if (!pl.ok) return ALFD_OK;
  DevCsr::Vs &v = m.vs;
  // block headers and the fixed-stride segment table (the kernel reads one segment past a block's last, and its first
  // round of 8 x 4 segments unconditionally; the descriptor table is read up to 16 batches past a block's last)
  int32_t max_nseg = 0;
  for (int64_t b = 0; b < pl.nb; ++b) max_nseg = std::max(max_nseg, pl.seg_begin[b + 1] - pl.seg_begin[b]);
  const int32_t S = std::max(max_nseg + 1, 33);
  if ((double)pl.nb * S * 8.0 > 2.0e9) return ALFD_OK;   // a block with thousands of window pieces: not this format
  std::vector<int32_t> hdrb((size_t)pl.nb * 8, 0), segx((size_t)pl.nb * S * 2 + 16, 0);   // + the reads past the last block
  for (int64_t b = 0; b < pl.nb; ++b) {
    const int32_t s0 = pl.seg_begin[b], ns = pl.seg_begin[b + 1] - s0;
    int32_t *h = &hdrb[(size_t)b * 8];
    h[0] = pl.cnt[b];
    h[1] = pl.blkW[b];
    h[2] = ns;
    h[3] = pl.dn[b];
    h[4] = pl.doff[b];
    h[5] = (int32_t)(uint32_t)((uint64_t)pl.sb[b] & 0xffffffffull);
    h[6] = (int32_t)((uint64_t)pl.sb[b] >> 32);
    for (int32_t q = 0; q < ns; ++q) {
      segx[((size_t)b * S + q) * 2] = pl.seg_col[s0 + q];
      segx[((size_t)b * S + q) * 2 + 1] = pl.seg_off[s0 + q];
    }
  }
  pl.tab.resize(pl.tab.size() + (size_t)16 * kVsBatchRows, 0);
  RC(upload_vec(ctx, m, &v.stream, pl.stream));
  RC(upload_vec(ctx, m, &v.tab, pl.tab));
  RC(upload_vec(ctx, m, &v.hdrb, hdrb));
  RC(upload_vec(ctx, m, &v.segx, segx));
  RC(upload_vec(ctx, m, &v.dict, pl.dict));
  HIPC(hipStreamSynchronize(ctx->stream));
  v.seg_stride = S;
  v.nb_interior = pl.nb_interior;
  v.nb = pl.nb;
  v.nseg = (int64_t)pl.seg_col.size();
  v.stream_bytes = (int64_t)pl.stream.size() - 4096;
  v.nbatch = pl.nbatch;
  v.shared_nnz = pl.shared_nnz;
  v.dict_total = (int64_t)pl.dict.size();
  v.stride = pl.stride;
  v.maxW = pl.maxW;
  v.wide = pl.wide;
  v.bricks = hint;
  v.on = true;
  if (ctx->cfg.log_level > 0)
    std::fprintf(stderr,
                 "[alfd] batch-major format%s: %lld blocks (%s), %lld batches, window <= %d slots, %.2f B/nnz, %.1f %% of the "
                 "entries in template-shared batches\n", pl.wide ? " (10-bit codes)" : "",
                 (long long)pl.nb, hint ? "caller's row blocks" : "runs of the numbering", (long long)pl.nbatch, pl.maxW,
                 (double)v.stream_bytes / (double)std::max<int64_t>(m.nnz, 1),
                 100.0 * (double)pl.shared_nnz / (double)std::max<int64_t>(m.nnz, 1));
  return ALFD_OK;
}

static int build_vss(alfd_ctx *ctx, DevCsr &m, const int64_t *rp, const int32_t *col, const double *val) {
  VsPlan pl;
  const int RB = std::min(kVssMaxRows, std::max(64, ctx->win_RB * std::max(1, ctx->win_short_scale) * 64 / m.L));
  plan_vss(m.nrows, m.L, rp, col, val, RB, ctx->win_maxW, ctx->win_gap, pl);
  // worthwhile only if rows do share templates (otherwise a dword per entry + descriptors buys nothing over the
  // 10 B/nnz window format... it still halves the stream, but the L-lane window kernel is the tested default)
  if (!pl.ok || pl.shared_nnz * 2 < m.nnz) return ALFD_OK;
  DevCsr::Vs &v = m.vs;
  RC(upload_vec(ctx, m, &v.stream, pl.stream));
  RC(upload_vec(ctx, m, &v.sb, pl.sb));
  RC(upload_vec(ctx, m, &v.tab, pl.tab));
  RC(upload_vec(ctx, m, &v.cnt, pl.cnt));
  RC(upload_vec(ctx, m, &v.seg_begin, pl.seg_begin));
  RC(upload_vec(ctx, m, &v.blkW, pl.blkW));
  RC(upload_vec(ctx, m, &v.seg_col, pl.seg_col));
  RC(upload_vec(ctx, m, &v.seg_off, pl.seg_off));
  RC(upload_vec(ctx, m, &v.doff, pl.doff));
  RC(upload_vec(ctx, m, &v.dn, pl.dn));
  RC(upload_vec(ctx, m, &v.dict, pl.dict));
  HIPC(hipStreamSynchronize(ctx->stream));
  v.nb = pl.nb;
  v.nseg = (int64_t)pl.seg_col.size();
  v.stream_bytes = (int64_t)pl.stream.size() - 4096;
  v.nbatch = pl.nbatch;
  v.shared_nnz = pl.shared_nnz;
  v.dict_total = (int64_t)pl.dict.size();
  v.stride = pl.stride;
  v.maxW = pl.maxW;
  v.L = m.L;
  v.tab_u64 = 1 + 4 * (64 / m.L);
  v.bricks = false;
  v.on = true;
  if (ctx->cfg.log_level > 0)
    std::fprintf(stderr, "[alfd] batch-major format (L = %d): %lld blocks, %lld batches, window <= %d slots, %.2f B/nnz, "
                 "%.1f %% of the entries in shared batches\n", m.L, (long long)pl.nb, (long long)pl.nbatch, pl.maxW,
                 (double)(v.stream_bytes + 8.0 * v.tab_u64 * v.nbatch) / (double)std::max<int64_t>(m.nnz, 1),
                 100.0 * (double)pl.shared_nnz / (double)std::max<int64_t>(m.nnz, 1));
  return ALFD_OK;
}

static int upload_matrix(alfd_ctx *ctx, int slot, int64_t nrows, int64_t ncols, const int64_t *rp,
                         const int32_t *col, const double *val) {
  DevCsr &m = ctx->mat[slot];
  csr_free(m);  // a re-upload releases the previous arrays of this slot
  const auto t_up0 = std::chrono::steady_clock::now();
  m.nrows = nrows;
  m.ncols = ncols;
  m.nnz = rp[nrows];
  int64_t nonempty = 0;
  for (int64_t r = 0; r < nrows; ++r) nonempty += rp[r + 1] > rp[r];
  m.sparse = nonempty * 2 < nrows;
  if (ctx->nranks > 1) {
    // lanes-per-row follows the GLOBAL matrix (so every rank, and the oracle, use one L)
    int64_t mine[2] = {m.nnz, nonempty};
    std::vector<int64_t> all(2 * (size_t)ctx->nranks);
    int64_t *d_m = nullptr, *d_a = nullptr;
    HIPC(hipMalloc((void **)&d_m, 2 * sizeof(int64_t)));
    HIPC(hipMalloc((void **)&d_a, 2 * sizeof(int64_t) * ctx->nranks));
    HIPC(hipMemcpyAsync(d_m, mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
    RC(comm_allgather(ctx, d_m, d_a, sizeof(mine)));
    HIPC(hipMemcpyAsync(all.data(), d_a, all.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    hipFree(d_m);
    hipFree(d_a);
    int64_t gn = 0, ge = 0;
    for (int p = 0; p < ctx->nranks; ++p) gn += all[2 * p], ge += all[2 * p + 1];
    DevCsr tmp;
    tmp.nnz = gn;
    choose_lanes(tmp, ge);
    m.L = tmp.L;
  } else {
    choose_lanes(m, nonempty);
  }
  const int32_t *col_up = col;
  std::vector<int32_t> remap;
  m.n_local_cols = (int32_t)ncols;
  if (ctx->nranks > 1 && ctx->up_local_only) {
    // rank-local operator (multigrid transfers): no halo, but the plan arrays must exist
    m.send_off.assign(ctx->nranks + 1, 0);
    m.recv_off.assign(ctx->nranks + 1, 0);
  } else if (ctx->nranks > 1) {
    if (ctx->part.empty()) return ctx->err = "alfd_set_partition must precede alfd_set_matrix", ALFD_E_INVALID;
    int rb, cb;
    slot_blocks(ctx, slot, &rb, &cb);
    const int64_t *po = ctx->up_col_offsets ? ctx->up_col_offsets : ctx->part[cb].data();
    const int64_t c0 = po[ctx->rank], c1 = po[ctx->rank + 1];
    m.n_local_cols = (int32_t)(c1 - c0);
    std::vector<int32_t> hal;
    remap.resize(m.nnz);
    m.recv_off.assign(ctx->nranks + 1, 0);
    host_halo_plan(m.nnz, col, po, ctx->nranks, ctx->rank, remap.data(), hal, m.recv_off.data());
    m.n_halo = (int64_t)hal.size();
    m.halo_globals = hal;
    col_up = remap.data();
    // counts all-to-all through an all-gather of the nranks x nranks matrix
    std::vector<int32_t> cnt_local(ctx->nranks), cnt_all((size_t)ctx->nranks * ctx->nranks);
    for (int p = 0; p < ctx->nranks; ++p) cnt_local[p] = (int32_t)(m.recv_off[p + 1] - m.recv_off[p]);
    int32_t *d_cl = nullptr, *d_ca = nullptr;
    HIPC(hipMalloc((void **)&d_cl, sizeof(int32_t) * ctx->nranks));
    HIPC(hipMalloc((void **)&d_ca, sizeof(int32_t) * ctx->nranks * ctx->nranks));
    HIPC(hipMemcpyAsync(d_cl, cnt_local.data(), ctx->nranks * 4, hipMemcpyHostToDevice, ctx->stream));
    RC(comm_allgather(ctx, d_cl, d_ca, (size_t)ctx->nranks * 4));
    HIPC(hipMemcpyAsync(cnt_all.data(), d_ca, cnt_all.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    m.send_off.assign(ctx->nranks + 1, 0);
    for (int p = 0; p < ctx->nranks; ++p)
      m.send_off[p + 1] = m.send_off[p] + cnt_all[(size_t)p * ctx->nranks + ctx->rank];
    const int64_t nsend = m.send_off.back();
    // exchange requested ids: I send my halo id list slices, receive what peers want from me
    int32_t *d_req = nullptr, *d_want = nullptr;
    HIPC(hipMalloc((void **)&d_req, sizeof(int32_t) * std::max<int64_t>(m.n_halo, 1)));
    HIPC(hipMalloc((void **)&d_want, sizeof(int32_t) * std::max<int64_t>(nsend, 1)));
    HIPC(hipMemcpyAsync(d_req, hal.data(), m.n_halo * 4, hipMemcpyHostToDevice, ctx->stream));
    // I send my wanted-id list slices (grouped by owner), I receive what peers want from me
    RC(comm_alltoallv(ctx, d_req, m.recv_off.data(), d_want, m.send_off.data(), sizeof(int32_t)));
    std::vector<int32_t> want(nsend);
    HIPC(hipMemcpyAsync(want.data(), d_want, nsend * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    hipFree(d_cl);
    hipFree(d_ca);
    hipFree(d_req);
    hipFree(d_want);
    // the column space of this matrix is block cb; my owned range there starts at c0
    for (auto &g : want) g = (int32_t)(g - c0);
    RC(csr_alloc(ctx, m, &m.send_idx, nsend));
    HIPC(hipMemcpyAsync(m.send_idx, want.data(), nsend * 4, hipMemcpyHostToDevice, ctx->stream));
    RC(csr_alloc(ctx, m, &m.send_buf, nsend));
    RC(csr_alloc(ctx, m, &m.halo, m.n_halo));
    HIPC(hipStreamSynchronize(ctx->stream));
  }
  RC(csr_alloc(ctx, m, &m.col, m.nnz));
  RC(csr_alloc(ctx, m, &m.val, m.nnz));
  // The CSR copy of a large operator (21 GB at N = 74, 0.9 s of PCIe time from pageable memory) runs on a helper thread
  // while this one plans the storage formats from the host arrays; joined before the call returns (the guard also joins
  // on the error paths).
  struct CopyJob {
    std::thread th;
    hipError_t rc = hipSuccess;
    ~CopyJob() { if (th.joinable()) th.join(); }
  } copy_job;
  if (m.nnz > 50000000) {
    const int dev = ctx->device;
    int32_t *dcol = m.col;
    double *dval = m.val;
    const int64_t nn = m.nnz;
    copy_job.th = std::thread([&copy_job, dev, dcol, dval, col_up, val, nn] {
      hipStream_t st = nullptr;   // a stream of its own: nothing here orders against the plan uploads on the context's
      hipError_t e = hipSetDevice(dev);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipMemcpyAsync(dcol, col_up, nn * sizeof(int32_t), hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipMemcpyAsync(dval, val, nn * sizeof(double), hipMemcpyHostToDevice, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (st) hipStreamDestroy(st);
      copy_job.rc = e;
    });
  } else {
    HIPC(hipMemcpyAsync(m.col, col_up, m.nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(m.val, val, m.nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  if (m.sparse) {
    std::vector<int32_t> rows;
    std::vector<int64_t> crp(1, 0);
    for (int64_t r = 0; r < nrows; ++r)
      if (rp[r + 1] > rp[r]) {
        rows.push_back((int32_t)r);
        crp.push_back(rp[r + 1]);
      }
    // compact row pointer: entries of non-empty rows are contiguous in CSR order
    for (size_t i = 0; i < rows.size(); ++i) crp[i] = rp[rows[i]];
    crp[rows.size()] = m.nnz;
    m.n_list = (int64_t)rows.size();
    RC(csr_alloc(ctx, m, &m.rows, m.n_list));
    RC(csr_alloc(ctx, m, &m.rp, m.n_list + 1));
    HIPC(hipMemcpyAsync(m.rows, rows.data(), m.n_list * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(m.rp, crp.data(), (m.n_list + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                        ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
  } else {
    m.n_list = nrows;
    RC(csr_alloc(ctx, m, &m.rp, nrows + 1));
    HIPC(hipMemcpyAsync(m.rp, rp, (nrows + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  // the windowed kernel launches one workgroup per row block: only for matrices with
  // enough row blocks to fill the chip (256 CUs x several workgroups)
  const bool win_long = m.L == 64 && m.nrows >= (int64_t)ctx->win_RB * ctx->win_min_blocks;
  const bool win_short = ctx->win_short_scale > 0 && m.L >= 8 && m.L < 64 &&
                         m.nrows >= (int64_t)std::min(512, ctx->win_RB * ctx->win_short_scale * 64 / m.L) * ctx->win_short_min_blocks;
  const bool windows = ctx->win_enable && (win_long || win_short) && !m.sparse && m.nnz > 0;
  const auto t_plan0 = std::chrono::steady_clock::now();
  // long rows: the batch-major form first -- it has dictionaries of its own (per row block, up to 1024 values with wide
  // codes) and is tried even when 96-row window blocks cannot be coded (cell-wise assembled operators)
  if (windows && ctx->vs_enable && ctx->win_vi && m.L == 64 && (slot != kScratchSlot || ctx->vi_levels))
    RC(build_vs(ctx, m, slot, rp, col_up, val));
  const auto t_plan1 = std::chrono::steady_clock::now();
  // a matrix the batch-major form holds needs the window plan (16-bit columns of the general 10 B/nnz kernel) only
  // when that kernel is asked for (tunables "value_index" / "batch_major" = 0, alfd_bench_spmv_format(..., 0)):
  // planned then, from the device copy (ensure_window_plan) -- 1.5 s of the upload at N = 74 otherwise
  m.win_deferred = windows && m.vs.on && m.L == 64;
  m.win_user = slot != kScratchSlot;
  if (windows && !m.win_deferred) RC(build_window(ctx, m, rp, col_up, val, slot != kScratchSlot, m.vs.on));
  if ((ctx->cfg.log_level > 0 || std::getenv("ALFD_LOG_UPLOAD")) && m.nnz > 50000000)
    std::fprintf(stderr, "[alfd] upload of a %lld-nnz matrix: CSR copy %.2f s, batch-major plan + copy %.2f s, window plan + copy %.2f s\n",
                 (long long)m.nnz, std::chrono::duration<double>(t_plan0 - t_up0).count(),
                 std::chrono::duration<double>(t_plan1 - t_plan0).count(),
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_plan1).count());
  // (also when the 10 B/nnz window plan did not fit its row blocks: plan_vss halves blocks whose x window is too wide --
  // transfer operators, whose rows reach across two grids)
  if (ctx->vs_enable && windows && !m.sparse && (m.L == 32 || m.L == 16 || m.L == 8)) RC(build_vss(ctx, m, rp, col_up, val));
  if (copy_job.th.joinable()) {
    copy_job.th.join();
    if (copy_job.rc != hipSuccess) return ctx->err = hipGetErrorString(copy_job.rc), ALFD_E_HIP;
  }
  m.present = true;
  RC(pick_short_row_format(ctx, m));
  return ALFD_OK;
}

static int upload_transpose(alfd_ctx *ctx, int slot, int64_t nrows, int64_t ncols, const int64_t *rp,
                            const int32_t *col, const double *val) {
  const int64_t nnz = rp[nrows];
  std::vector<int64_t> trp(ncols + 1, 0);
  std::vector<int32_t> tcol(nnz);
  std::vector<double> tval(nnz);
  for (int64_t k = 0; k < nnz; ++k) trp[col[k] + 1]++;
  for (int64_t r = 0; r < ncols; ++r) trp[r + 1] += trp[r];
  std::vector<int64_t> cur(trp.begin(), trp.end() - 1);
  for (int64_t r = 0; r < nrows; ++r)
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) {
      const int64_t p = cur[col[k]]++;
      tcol[p] = (int32_t)r;
      tval[p] = val[k];
    }
  return upload_matrix(ctx, slot, ncols, nrows, trp.data(), tcol.data(), tval.data());
}

static int nblocks_of(int variant) { return variant == ALFD_AL2 || variant == ALFD_RATIONAL ? 2 : 3; }

static int ws_alloc_zero(alfd_ctx *ctx, double **p, int64_t count) {
  void *q = nullptr;
  HIPC(hipMalloc(&q, std::max<int64_t>(count, 1) * sizeof(double)));
  ctx->ws_allocs.push_back(q);
  *p = static_cast<double *>(q);
  HIPC(hipMemsetAsync(q, 0, std::max<int64_t>(count, 1) * sizeof(double), ctx->stream));
  return ALFD_OK;
}

// dinv = 1 / (diag(Adiag) + g * sum_k w_k R_ik^2): the diagonal of Adiag + g R diag(w) R^T
static int diag_plus_m(alfd_ctx *ctx, const DevCsr &A, DevCsr &R, double g, int64_t n, double *dinv);
static int diag_plus(alfd_ctx *ctx, int slot_diag, int slot_rows, double g, int64_t n, double *dinv) {
  return diag_plus_m(ctx, ctx->mat[slot_diag], ctx->mat[slot_rows], g, n, dinv);
}
static int diag_plus_m(alfd_ctx *ctx, const DevCsr &A, DevCsr &R, double g, int64_t n, double *dinv) {
  if (pad_chunk(n) > ctx->diag_cap) {
    // a replicated multigrid level may be longer than this rank's largest block (many ranks): grow the two scratch vectors
    RC(ws_alloc_zero(ctx, &ctx->dA, pad_chunk(n)));
    RC(ws_alloc_zero(ctx, &ctx->s_aug, pad_chunk(n)));
    ctx->diag_cap = pad_chunk(n);
  }
  HIPC(hipMemsetAsync(ctx->dA, 0, pad_chunk(n) * sizeof(double), ctx->stream));
  HIPC(hipMemsetAsync(ctx->s_aug, 0, pad_chunk(n) * sizeof(double), ctx->stream));
  const int grid = grid_for_rows(A.nrows, A.L);
#define ALFD_DIAG(LL)                                                                                   \
  hipLaunchKernelGGL((extract_diag_kernel<LL>), dim3(grid), dim3(kBlock), 0, ctx->stream, A.nrows, A.rp, \
                     A.col, A.val, ctx->dA)
  switch (A.L) {
    case 4: ALFD_DIAG(4); break;
    case 8: ALFD_DIAG(8); break;
    case 16: ALFD_DIAG(16); break;
    case 32: ALFD_DIAG(32); break;
    default: ALFD_DIAG(64); break;
  }
#undef ALFD_DIAG
  // a rank with no rows of R of its own may still own W entries its peers need
  if (ctx->nranks > 1 && (ctx->local || ctx->host_alltoallv || R.n_halo > 0 || R.send_off.back() > 0))
    RC(halo_exchange(ctx, R, ctx->diag[ALFD_INVW]));
  if (R.n_list > 0)
    hipLaunchKernelGGL(aug_diag_rows_kernel, dim3((unsigned)((R.n_list + 255) / 256)), dim3(256), 0,
                       ctx->stream, R.n_list, R.rp, R.col, R.val, R.sparse ? R.rows : nullptr,
                       ctx->diag[ALFD_INVW], R.halo, R.n_local_cols, ctx->s_aug);
  hipLaunchKernelGGL(aug_diag_finish_kernel, dim3((unsigned)std::max<int64_t>(1, (n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                     g, ctx->dA, ctx->s_aug, dinv);
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// lambda_max(D^-1 Op) by power iteration from the integer-hash start vector
static int power_iteration(alfd_ctx *ctx, int op) {
  const alfd_config &c = ctx->cfg;
  const int64_t npad = op_npad(ctx, op);
  double *v = ctx->w_p, *wv = ctx->w_Ap;
  HIPC(hipMemsetAsync(v, 0, npad * sizeof(double), ctx->stream));
  HIPC(hipMemsetAsync(wv, 0, npad * sizeof(double), ctx->stream));
  auto fill = [&](int blk, double *dst) {
    const int64_t goff = ctx->nranks > 1 ? ctx->part[blk][ctx->rank] : 0;
    hipLaunchKernelGGL(hash_vector_kernel, dim3((unsigned)std::max<int64_t>(1, (ctx->n[blk] + 255) / 256)), dim3(256), 0,
                       ctx->stream, ctx->n[blk], goff, dst);
  };
  if (op == OP_AUG2) {
    fill(0, v);
    fill(1, v + ctx->off[1]);
  } else {
    fill((op == OP_AUG || op == OP_K) ? 0 : 1, v);
  }
  double lam = 0;
  for (int it = 0; it < c.cheb_power_its; ++it) {
    RC(dot_async(ctx, npad, v, v, S_TMP));
    RC(read_scalars(ctx, S_TMP, 1));
    const double nv = std::sqrt(ctx->sc_host[S_TMP]);
    VEC_LAUNCH(scale_kernel, npad, 16, (const double *)nullptr, 0, 0, 1.0 / nv, v);
    RC(op_apply(ctx, op, v, wv));
    VEC_LAUNCH(pmul_scale_kernel, npad, 24, 1.0, op_dinv(ctx, op), wv, wv);
    RC(dot_async(ctx, npad, wv, wv, S_TMP));
    RC(read_scalars(ctx, S_TMP, 1));
    lam = std::sqrt(ctx->sc_host[S_TMP]);
    std::swap(v, wv);
  }
  ctx->lam_max[op] = lam * c.cheb_safety;
  HIPC(hipMemsetAsync(ctx->w_p, 0, ctx->wmax * sizeof(double), ctx->stream));
  HIPC(hipMemsetAsync(ctx->w_Ap, 0, ctx->wmax * sizeof(double), ctx->stream));
  return ALFD_OK;
}

// ======================================================================
// Aggregation multigrid for the augmented block (ALFD_PREC_MULTILEVEL).
// Level l+1 = Galerkin coarsening of level l through the caller's aggregates:
//   A_{l+1} = P^T A_l P,  C_{l+1} = C_l P,  Aug_{l+1} = A_{l+1} + gamma C_{l+1}^T invW C_{l+1}
// (kept factored on every level, like the fine operator).  The cycle is a
// symmetric V-cycle: Chebyshev pre-smoothing from zero, coarse correction,
// Chebyshev post-smoothing of the residual; the coarsest level is "solved" by a
// high-degree Chebyshev sweep.  Every piece is a fixed polynomial in SPD
// operators, so the preconditioner is a fixed SPD operator and plain CG applies.
static void free_levels(alfd_ctx *ctx) {
  for (MlLevel &L : ctx->ml)
    for (DevCsr *m : {&L.A, &L.C, &L.Ct, &L.P, &L.R, &L.gA, &L.gC, &L.gCt, &L.gP, &L.gR}) csr_free(*m);
  ctx->ml.clear();
  csr_free(ctx->ml_inv);
  for (DevCsr *m : {&ctx->patch.Ass, &ctx->patch.As, &ctx->patch.Ats, &ctx->patch.Cs, &ctx->patch.Cts, &ctx->patch.Cts_loc,
                    &ctx->patch.Ctg})
    csr_free(*m);
  ctx->patch = alfd_ctx::Patch();   // its vectors belong to the setup workspace
  ctx->ml_rep_level = -1;
}

static int download_csr(alfd_ctx *ctx, const DevCsr &m, HostCsr &h) {
  h.nrows = m.nrows;
  h.ncols = m.ncols;
  h.col.resize(m.nnz);
  h.val.resize(m.nnz);
  HIPC(hipMemcpyAsync(h.col.data(), m.col, m.nnz * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipMemcpyAsync(h.val.data(), m.val, m.nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  h.rp.assign(m.nrows + 1, 0);
  if (m.sparse) {
    std::vector<int32_t> rows(m.n_list);
    std::vector<int64_t> crp(m.n_list + 1);
    HIPC(hipMemcpyAsync(rows.data(), m.rows, m.n_list * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(crp.data(), m.rp, (m.n_list + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < m.n_list; ++i) h.rp[rows[i] + 1] = crp[i + 1] - crp[i];
    for (int64_t r = 0; r < m.nrows; ++r) h.rp[r + 1] += h.rp[r];
  } else {
    HIPC(hipMemcpyAsync(h.rp.data(), m.rp, (m.nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
  }
  return ALFD_OK;
}

// The deferred window plan of a matrix held in the batch-major form (upload_matrix), built from its device copy.
static int ensure_window_plan(alfd_ctx *ctx, DevCsr &m) {
  if (!m.present || !m.win_deferred) return ALFD_OK;
  HostCsr h;
  RC(download_csr(ctx, m, h));
  m.win_deferred = false;
  const int tag = m.tag;
  RC(build_window(ctx, m, h.rp.data(), h.col.data(), h.val.data(), m.win_user, true));
  m.tag = tag;
  return ALFD_OK;
}
static int ensure_window_plans(alfd_ctx *ctx) {
  for (DevCsr &m : ctx->mat) RC(ensure_window_plan(ctx, m));
  for (MlLevel &L : ctx->ml)
    for (DevCsr *m : {&L.A, &L.gA}) RC(ensure_window_plan(ctx, *m));
  return ALFD_OK;
}

static void transpose_host(const HostCsr &a, HostCsr &t) {
  t.nrows = a.ncols;
  t.ncols = a.nrows;
  const int64_t nnz = a.nnz();
  t.rp.assign(t.nrows + 1, 0);
  t.col.resize(nnz);
  t.val.resize(nnz);
  for (int64_t k = 0; k < nnz; ++k) t.rp[a.col[k] + 1]++;
  for (int64_t r = 0; r < t.nrows; ++r) t.rp[r + 1] += t.rp[r];
  std::vector<int64_t> cur(t.rp.begin(), t.rp.end() - 1);
  for (int64_t r = 0; r < a.nrows; ++r)
    for (int64_t k = a.rp[r]; k < a.rp[r + 1]; ++k) {
      const int64_t p = cur[a.col[k]]++;
      t.col[p] = (int32_t)r;
      t.val[p] = a.val[k];
    }
}

// out = R A Q for aggregation-type transfers.  Row I of out collects the fine rows i with
// agg_row[i] == I (agg_row == nullptr: I == i); column J collects agg_col[j] == J.
// Canonical accumulation order (the oracle follows it too): fine rows ascending, entries
// in CSR order, each added to its coarse entry as it is met; the finished row is sorted
// by column.
static void galerkin(const HostCsr &A, const int32_t *agg_row, const double *w_row, int64_t n_rows_c,
                     const int32_t *agg_col, const double *w_col, int64_t n_cols_c, HostCsr &out) {
  std::vector<int64_t> mem_ptr;
  std::vector<int32_t> mem;
  if (agg_row) {
    mem_ptr.assign(n_rows_c + 1, 0);
    for (int64_t i = 0; i < A.nrows; ++i)
      if (agg_row[i] >= 0) mem_ptr[agg_row[i] + 1]++;
    for (int64_t I = 0; I < n_rows_c; ++I) mem_ptr[I + 1] += mem_ptr[I];
    mem.resize(mem_ptr[n_rows_c]);
    std::vector<int64_t> cur(mem_ptr.begin(), mem_ptr.end() - 1);
    for (int64_t i = 0; i < A.nrows; ++i)
      if (agg_row[i] >= 0) mem[cur[agg_row[i]]++] = (int32_t)i;
  }
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::vector<int64_t>> t_cnt(T);
  std::vector<std::vector<int32_t>> t_col(T);
  std::vector<std::vector<double>> t_val(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      const int64_t I0 = n_rows_c * t / T, I1 = n_rows_c * (t + 1) / T;
      std::vector<int64_t> marker(n_cols_c, -1);
      std::vector<std::pair<int32_t, double>> row;
      for (int64_t I = I0; I < I1; ++I) {
        row.clear();
        const int64_t m0 = agg_row ? mem_ptr[I] : I, m1 = agg_row ? mem_ptr[I + 1] : I + 1;
        for (int64_t mi = m0; mi < m1; ++mi) {
          const int64_t i = agg_row ? mem[mi] : mi;
          const double wi = w_row ? w_row[i] : 1.0;
          for (int64_t k = A.rp[i]; k < A.rp[i + 1]; ++k) {
            const int32_t j = A.col[k];
            const int32_t J = agg_col ? agg_col[j] : j;
            if (J < 0) continue;
            const double wj = w_col ? w_col[j] : 1.0;
            const double c = (w_row || w_col) ? (wi * wj) * A.val[k] : A.val[k];
            if (marker[J] < 0) {
              marker[J] = (int64_t)row.size();
              row.emplace_back(J, c);
            } else {
              row[marker[J]].second = row[marker[J]].second + c;
            }
          }
        }
        for (auto &e : row) marker[e.first] = -1;
        std::sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        t_cnt[t].push_back((int64_t)row.size());
        for (auto &e : row) {
          t_col[t].push_back(e.first);
          t_val[t].push_back(e.second);
        }
      }
    });
  for (auto &x : th) x.join();
  out.nrows = n_rows_c;
  out.ncols = n_cols_c;
  out.rp.assign(1, 0);
  out.col.clear();
  out.val.clear();
  for (int t = 0; t < T; ++t) {
    for (int64_t c : t_cnt[t]) out.rp.push_back(out.rp.back() + c);
    out.col.insert(out.col.end(), t_col[t].begin(), t_col[t].end());
    out.val.insert(out.val.end(), t_val[t].begin(), t_val[t].end());
  }
}

static int upload_matrix(alfd_ctx *ctx, int slot, int64_t nrows, int64_t ncols, const int64_t *rp,
                         const int32_t *col, const double *val);
static int upload_level(alfd_ctx *ctx, DevCsr &dst, const HostCsr &h) {
  static const int32_t no_col = 0;
  static const double no_val = 0;
  RC(upload_matrix(ctx, kScratchSlot, h.nrows, h.ncols, h.rp.data(), h.col.empty() ? &no_col : h.col.data(),
                   h.val.empty() ? &no_val : h.val.data()));
  csr_free(dst);
  dst = std::move(ctx->mat[kScratchSlot]);
  ctx->mat[kScratchSlot] = DevCsr();
  dst.tag = 1;
  return ALFD_OK;
}

// y = Aug_l x
static int level_op(alfd_ctx *ctx, int l, const double *x, double *y) {
  if (l == 0) return op_apply(ctx, OP_AUG, x, y);
  MlLevel &L = ctx->ml[l];
  RC(spmv_m(ctx, L.A, ALFD_T_SPMV_OTHER, x, y, 0));
  if (ctx->cfg.aug_assembled) return ALFD_OK;
  RC(spmv_m(ctx, L.C, ALFD_T_SPMV_OTHER, x, ctx->t_lam, 2, 0.0, ctx->diag[ALFD_INVW]));
  return spmv_m(ctx, L.Ct, ALFD_T_SPMV_OTHER, ctx->t_lam, y, 1, ctx->cfg.gamma);
}

// z = p_k(D^-1 Aug_l) D^-1 r on level l (Chebyshev, zero start)
static int level_cheb(alfd_ctx *ctx, int l, int degree, double ratio, const double *r, double *z) {
  MlLevel &L = ctx->ml[l];
  const double lmax = L.lmax, lmin = lmax / ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  const double sigma = theta / delta;
  double rho = 1.0 / sigma;
  VEC_LAUNCH(cheb_init_kernel, L.npad, degree > 1 ? 40 : 32, 1.0 / theta, L.dinv, r, L.cd, z, L.cres,
             degree > 1 ? 1 : 0);
  for (int j = 1; j < degree; ++j) {
    RC(level_op(ctx, l, L.cd, L.ctmp));
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
    VEC_LAUNCH(cheb_step_kernel, L.npad, 64, c1, c2, L.dinv, L.ctmp, L.cres, L.cd, z);
    rho = rho_new;
  }
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// ---- replicated levels: every rank holds the global operators and vectors and does the same work
static int level_op_rep(alfd_ctx *ctx, int l, const double *x, double *y) {
  MlLevel &L = ctx->ml[l];
  RC(spmv_m(ctx, L.gA, ALFD_T_SPMV_OTHER, x, y, 0));
  if (ctx->cfg.aug_assembled) return ALFD_OK;
  RC(spmv_m(ctx, L.gC, ALFD_T_SPMV_OTHER, x, ctx->g_tlam, 2, 0.0, ctx->g_w));
  return spmv_m(ctx, L.gCt, ALFD_T_SPMV_OTHER, ctx->g_tlam, y, 1, ctx->cfg.gamma);
}

static int level_cheb_rep(alfd_ctx *ctx, int l, int degree, double ratio, const double *r, double *z) {
  MlLevel &L = ctx->ml[l];
  const double lmax = L.lmax, lmin = lmax / ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  const double sigma = theta / delta;
  double rho = 1.0 / sigma;
  VEC_LAUNCH(cheb_init_kernel, L.gnpad, degree > 1 ? 40 : 32, 1.0 / theta, L.gdinv, r, L.gcd, z, L.gcres,
             degree > 1 ? 1 : 0);
  for (int j = 1; j < degree; ++j) {
    RC(level_op_rep(ctx, l, L.gcd, L.gctmp));
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
    VEC_LAUNCH(cheb_step_kernel, L.gnpad, 64, c1, c2, L.gdinv, L.gctmp, L.gcres, L.gcd, z);
    rho = rho_new;
  }
  HIPC(hipGetLastError());
  return ALFD_OK;
}

static int ml_cycle_rep(alfd_ctx *ctx, int l, const double *r, double *z) {
  const alfd_config &c = ctx->cfg;
  const int last = (int)ctx->ml.size() - 1;
  if (l == last) {
    if (ctx->ml_inv.present) return spmv_m(ctx, ctx->ml_inv, ALFD_T_SPMV_OTHER, r, z, 0);
    return level_cheb_rep(ctx, l, c.ml_coarse_degree, c.ml_coarse_ratio, r, z);
  }
  MlLevel &L = ctx->ml[l], &N = ctx->ml[l + 1];
  const int sdeg = l > 0 && c.ml_smooth_degree_coarse > 0 ? c.ml_smooth_degree_coarse : c.ml_smooth_degree;
  RC(level_cheb_rep(ctx, l, sdeg, c.ml_smooth_ratio, r, z));
  RC(level_op_rep(ctx, l, z, L.gt));
  VEC_LAUNCH(sub_from_kernel, L.gnpad, 24, r, L.gt);
  RC(spmv_m(ctx, N.gR, ALFD_T_SPMV_OTHER, L.gt, N.gr, 0));
  RC(ml_cycle_rep(ctx, l + 1, N.gr, N.gz));
  RC(spmv_m(ctx, N.gP, ALFD_T_SPMV_OTHER, N.gz, z, 1, 1.0));
  RC(level_op_rep(ctx, l, z, L.gt));
  VEC_LAUNCH(sub_from_kernel, L.gnpad, 24, r, L.gt);
  RC(level_cheb_rep(ctx, l, sdeg, c.ml_smooth_ratio, L.gt, L.gr));
  VEC_LAUNCH(axpy_kernel, L.gnpad, 24, (const double *)nullptr, 0, 1.0, L.gr, z);
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// z = V-cycle(r) on level l
static int ml_cycle(alfd_ctx *ctx, int l, const double *r, double *z) {
  const alfd_config &c = ctx->cfg;
  const int last = (int)ctx->ml.size() - 1;
  if (l == last) {
    if (ctx->ml_inv.present) return spmv_m(ctx, ctx->ml_inv, ALFD_T_SPMV_OTHER, r, z, 0);   // z = Aug_c^-1 r
    return level_cheb(ctx, l, c.ml_coarse_degree, c.ml_coarse_ratio, r, z);
  }
  MlLevel &L = ctx->ml[l], &N = ctx->ml[l + 1];
  const int sdeg = l > 0 && c.ml_smooth_degree_coarse > 0 ? c.ml_smooth_degree_coarse : c.ml_smooth_degree;
  RC(level_cheb(ctx, l, sdeg, c.ml_smooth_ratio, r, z));                     // pre-smoothing from zero
  RC(level_op(ctx, l, z, L.t));
  VEC_LAUNCH(sub_from_kernel, L.npad, 24, r, L.t);                           // t = r - Aug z
  RC(spmv_m(ctx, N.R, ALFD_T_SPMV_OTHER, L.t, N.r, 0));                      // r_c = P^T t
  if (l + 1 == ctx->ml_rep_level) {
    // the restricted residual is gathered once; everything below runs replicated, without exchanges
    HIPC(hipMemcpyAsync(N.g_send, N.r, N.n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    RC(comm_allgather(ctx, N.g_send, N.g_stage, (size_t)N.g_maxpiece * sizeof(double)));
    for (int p = 0; p < ctx->nranks; ++p) {
      const int64_t np = N.g_offs[p + 1] - N.g_offs[p];
      if (np > 0)
        HIPC(hipMemcpyAsync(N.gr + N.g_offs[p], N.g_stage + (int64_t)p * N.g_maxpiece, np * sizeof(double),
                            hipMemcpyDeviceToDevice, ctx->stream));
    }
    RC(ml_cycle_rep(ctx, l + 1, N.gr, N.gz));
    // z += P e_c: aggregates -> my slice of e_c; CSR prolongators address the replicated vector by global ids
    RC(spmv_m(ctx, N.P, ALFD_T_SPMV_OTHER, N.P_global_cols ? N.gz : N.gz + N.g_offs[ctx->rank], z, 1, 1.0));
  } else {
    RC(ml_cycle(ctx, l + 1, N.r, N.z));
    RC(spmv_m(ctx, N.P, ALFD_T_SPMV_OTHER, N.z, z, 1, 1.0));                 // z += P e_c
  }
  RC(level_op(ctx, l, z, L.t));
  VEC_LAUNCH(sub_from_kernel, L.npad, 24, r, L.t);
  RC(level_cheb(ctx, l, sdeg, c.ml_smooth_ratio, L.t, L.r));                 // post-smoothing correction
  VEC_LAUNCH(axpy_kernel, L.npad, 24, (const double *)nullptr, 0, 1.0, L.r, z);
  HIPC(hipGetLastError());
  return ALFD_OK;
}

static int ws_alloc_zero(alfd_ctx *ctx, double **p, int64_t count);

// ---- interface patch (alfd_config::ml_patch_degree): y = Aug_SS x on patch-compact vectors
static int patch_op(alfd_ctx *ctx, const double *x, double *y) {
  alfd_ctx::Patch &Q = ctx->patch;
  RC(spmv_m(ctx, Q.Ass, ALFD_T_SPMV_OTHER, x, y, 0));
  if (ctx->cfg.aug_assembled) return ALFD_OK;
  double *t = Q.rep ? ctx->g_tlam : ctx->t_lam;                       // replicated patch: the whole multiplier space
  const double *w = Q.rep ? ctx->g_w : ctx->diag[ALFD_INVW];
  RC(spmv_m(ctx, Q.Cs, ALFD_T_SPMV_OTHER, x, t, 2, 0.0, w));
  return spmv_m(ctx, Q.Cts, ALFD_T_SPMV_OTHER, t, y, 1, ctx->cfg.gamma);
}

// z = q(D^-1 Aug_SS) D^-1 r, q = Chebyshev polynomial of degree ml_patch_degree (zero start)
static int patch_cheb(alfd_ctx *ctx, const double *r, double *z) {
  alfd_ctx::Patch &Q = ctx->patch;
  const int degree = ctx->cfg.ml_patch_degree;
  const double lmax = Q.lmax, lmin = lmax / ctx->cfg.ml_patch_ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  const double sigma = theta / delta;
  double rho = 1.0 / sigma;
  VEC_LAUNCH(cheb_init_kernel, Q.mpad, degree > 1 ? 40 : 32, 1.0 / theta, Q.dinv, r, Q.cd, z, Q.cres, degree > 1 ? 1 : 0);
  for (int j = 1; j < degree; ++j) {
    RC(patch_op(ctx, Q.cd, Q.ctmp));
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
    VEC_LAUNCH(cheb_step_kernel, Q.mpad, 64, c1, c2, Q.dinv, Q.ctmp, Q.cres, Q.cd, z);
    rho = rho_new;
  }
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// The multilevel inner preconditioner: the V-cycle, wrapped (ml_patch_degree > 0) into two corrections on the
// interface patch:  z1 = E q E^T r;  z2 = z1 + V(r - Aug z1);  z = z2 + E q E^T (r - Aug z2)  -- symmetric.
// device vector pieces (offs[p+1] - offs[p] doubles on rank p) -> the whole vector on every rank
static int allgather_pieces(alfd_ctx *ctx, const double *mine, const std::vector<int64_t> &offs, int64_t piece, double *send,
                            double *stage, double *whole) {
  const int64_t n_me = offs[ctx->rank + 1] - offs[ctx->rank];
  if (n_me > 0) HIPC(hipMemcpyAsync(send, mine, n_me * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  RC(comm_allgather(ctx, send, stage, (size_t)piece * sizeof(double)));
  for (int p = 0; p < ctx->nranks; ++p) {
    const int64_t np = offs[p + 1] - offs[p];
    if (np > 0)
      HIPC(hipMemcpyAsync(whole + offs[p], stage + (int64_t)p * piece, np * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  return ALFD_OK;
}

// ml_apply on a partitioned context: same arithmetic, the patch polynomial runs redundantly on every rank
static int ml_apply_rep(alfd_ctx *ctx, const double *r, double *z) {
  alfd_ctx::Patch &Q = ctx->patch;
  const int64_t n0p = pad_chunk(ctx->n[0]);
  const unsigned gm = (unsigned)std::max<int64_t>(1, (Q.m_loc + 255) / 256);
  const bool pen = !ctx->cfg.aug_assembled;
  const int last = ctx->nblocks - 1;
  const int64_t s0 = Q.soff[ctx->rank];
  if (Q.m_loc > 0) hipLaunchKernelGGL(gather_kernel, dim3(gm), dim3(256), 0, ctx->stream, Q.m_loc, Q.S, r, Q.loc);
  RC(allgather_pieces(ctx, Q.loc, Q.soff, Q.piece, Q.send, Q.stage, Q.rS));
  RC(patch_cheb(ctx, Q.rS, Q.zS));
  VEC_LAUNCH(scale_copy_kernel, n0p, 16, 1.0, r, Q.rr);
  RC(spmv_m(ctx, Q.Ats, ALFD_T_SPMV_OTHER, Q.zS, Q.rr, 1, -1.0));
  if (pen) {
    RC(spmv_m(ctx, Q.Cs, ALFD_T_SPMV_OTHER, Q.zS, ctx->g_tlam, 2, 0.0, ctx->g_w));
    RC(spmv_m(ctx, Q.Ctg, ALFD_T_SPMV_OTHER, ctx->g_tlam, Q.rr, 1, -ctx->cfg.gamma));
  }
  RC(ml_cycle(ctx, 0, Q.rr, z));
  if (Q.m_loc > 0) hipLaunchKernelGGL(scatter_add_kernel, dim3(gm), dim3(256), 0, ctx->stream, Q.m_loc, Q.S, Q.zS + s0, z);
  RC(spmv_m(ctx, Q.As, ALFD_T_SPMV_OTHER, z, Q.loc, 0));                          // my rows of (A z) on S
  if (pen) {
    RC(spmv(ctx, ALFD_C, z, ctx->t_lam, 2, 0.0, ctx->diag[ALFD_INVW]));          // my multipliers of W^-1 C z
    RC(allgather_pieces(ctx, ctx->t_lam, ctx->part[last], Q.lam_piece, Q.lam_send, Q.lam_stage, ctx->g_tlam));
    RC(spmv_m(ctx, Q.Cts_loc, ALFD_T_SPMV_OTHER, ctx->g_tlam, Q.loc, 1, ctx->cfg.gamma));
  }
  RC(allgather_pieces(ctx, Q.loc, Q.soff, Q.piece, Q.send, Q.stage, Q.uS));
  VEC_LAUNCH(sub_from_kernel, Q.mpad, 24, Q.rS, Q.uS);
  RC(patch_cheb(ctx, Q.uS, Q.eS));
  if (Q.m_loc > 0) hipLaunchKernelGGL(scatter_add_kernel, dim3(gm), dim3(256), 0, ctx->stream, Q.m_loc, Q.S, Q.eS + s0, z);
  HIPC(hipGetLastError());
  return ALFD_OK;
}

static int ml_apply(alfd_ctx *ctx, const double *r, double *z) {
  alfd_ctx::Patch &Q = ctx->patch;
  if (!Q.on) return ml_cycle(ctx, 0, r, z);
  if (Q.rep) return ml_apply_rep(ctx, r, z);
  const int64_t n0p = pad_chunk(ctx->n[0]);
  const unsigned gm = (unsigned)((Q.m + 255) / 256);
  const bool pen = !ctx->cfg.aug_assembled;
  const double *w = ctx->diag[ALFD_INVW];
  hipLaunchKernelGGL(gather_kernel, dim3(gm), dim3(256), 0, ctx->stream, Q.m, Q.S, r, Q.rS);
  RC(patch_cheb(ctx, Q.rS, Q.zS));
  VEC_LAUNCH(scale_copy_kernel, n0p, 16, 1.0, r, Q.rr);                           // rr = r (a blit copy costs 10x this kernel)
  RC(spmv_m(ctx, Q.Ats, ALFD_T_SPMV_OTHER, Q.zS, Q.rr, 1, -1.0));                 // rr -= A[:,S] zS
  if (pen) {
    RC(spmv_m(ctx, Q.Cs, ALFD_T_SPMV_OTHER, Q.zS, ctx->t_lam, 2, 0.0, w));
    RC(spmv(ctx, ALFD_CT, ctx->t_lam, Q.rr, 1, -ctx->cfg.gamma));                 // rr -= gamma Ct W^-1 C[:,S] zS
  }
  RC(ml_cycle(ctx, 0, Q.rr, z));
  hipLaunchKernelGGL(scatter_add_kernel, dim3(gm), dim3(256), 0, ctx->stream, Q.m, Q.S, Q.zS, z);
  RC(spmv_m(ctx, Q.As, ALFD_T_SPMV_OTHER, z, Q.uS, 0));                           // (Aug z) on S
  if (pen) {
    RC(spmv(ctx, ALFD_C, z, ctx->t_lam, 2, 0.0, w));
    RC(spmv_m(ctx, Q.Cts, ALFD_T_SPMV_OTHER, ctx->t_lam, Q.uS, 1, ctx->cfg.gamma));
  }
  VEC_LAUNCH(sub_from_kernel, Q.mpad, 24, Q.rS, Q.uS);                            // uS = r_S - (Aug z)_S
  RC(patch_cheb(ctx, Q.uS, Q.eS));
  hipLaunchKernelGGL(scatter_add_kernel, dim3(gm), dim3(256), 0, ctx->stream, Q.m, Q.S, Q.eS, z);
  HIPC(hipGetLastError());
  return ALFD_OK;
}

// out = A * Pm on the host, multi-threaded over row ranges.  Every output entry (i, J) is ONE sequential fma
// chain: entries k of row i of A in CSR order, entries of row col_k of Pm in CSR order,
// acc_J = fma(a_ik, p_kJ, acc_J) from 0; the finished row is sorted by column (the oracle repeats this).
static void spgemm_host(const HostCsr &A, const HostCsr &Pm, HostCsr &out) {
  const int64_t nrows = A.nrows, nc = Pm.ncols;
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())),
                                                           (nrows + 4095) / 4096));
  std::vector<std::vector<int32_t>> t_cnt(T), t_col(T);
  std::vector<std::vector<double>> t_val(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      const int64_t i0 = nrows * t / T, i1 = nrows * (t + 1) / T;
      std::vector<int64_t> stamp(nc, -1);
      std::vector<double> acc(nc, 0.0);
      std::vector<int32_t> touched;
      t_cnt[t].reserve(i1 - i0);
      for (int64_t i = i0; i < i1; ++i) {
        touched.clear();
        for (int64_t k = A.rp[i]; k < A.rp[i + 1]; ++k) {
          const double a = A.val[k];
          const int64_t j = A.col[k];
          for (int64_t e = Pm.rp[j]; e < Pm.rp[j + 1]; ++e) {
            const int32_t J = Pm.col[e];
            if (stamp[J] != i) {
              stamp[J] = i;
              acc[J] = std::fma(a, Pm.val[e], 0.0);
              touched.push_back(J);
            } else {
              acc[J] = std::fma(a, Pm.val[e], acc[J]);
            }
          }
        }
        std::sort(touched.begin(), touched.end());
        t_cnt[t].push_back((int32_t)touched.size());
        for (int32_t J : touched) {
          t_col[t].push_back(J);
          t_val[t].push_back(acc[J]);
        }
      }
    });
  for (auto &x : th) x.join();
  out.nrows = nrows;
  out.ncols = nc;
  out.rp.assign(1, 0);
  out.rp.reserve(nrows + 1);
  int64_t total = 0;
  for (int t = 0; t < T; ++t) total += (int64_t)t_col[t].size();
  out.col.clear();
  out.val.clear();
  out.col.reserve(total);
  out.val.reserve(total);
  for (int t = 0; t < T; ++t) {
    for (int32_t c : t_cnt[t]) out.rp.push_back(out.rp.back() + c);
    out.col.insert(out.col.end(), t_col[t].begin(), t_col[t].end());
    out.val.insert(out.val.end(), t_val[t].begin(), t_val[t].end());
    std::vector<int32_t>().swap(t_col[t]);
    std::vector<double>().swap(t_val[t]);
  }
}

// rows `rows` of A with compact row numbering (full_rows = 0) or as rows of an nrows_out-row matrix whose other
// rows are empty (full_rows = 1); columns kept where colmap[col] >= 0 (renumbered) or all (colmap == nullptr)
static void extract_host(const HostCsr &A, const std::vector<int32_t> &rows, const int32_t *colmap, int64_t ncols,
                         bool full_rows, HostCsr &out) {
  out.nrows = full_rows ? A.nrows : (int64_t)rows.size();
  out.ncols = ncols;
  out.rp.assign(out.nrows + 1, 0);
  out.col.clear();
  out.val.clear();
  int64_t next = 0;   // next output row to close
  for (size_t q = 0; q < rows.size(); ++q) {
    const int64_t i = rows[q], orow = full_rows ? i : (int64_t)q;
    for (; next <= orow; ++next) out.rp[next] = (int64_t)out.col.size();
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      const int32_t c = colmap ? colmap[A.col[k]] : A.col[k];
      if (c < 0) continue;
      out.col.push_back(c);
      out.val.push_back(A.val[k]);
    }
  }
  for (; next <= out.nrows; ++next) out.rp[next] = (int64_t)out.col.size();
}

static int upload_level(alfd_ctx *ctx, DevCsr &dst, const HostCsr &h);
static int upload_level_part(alfd_ctx *ctx, DevCsr &dst, const HostCsr &h, const int64_t *col_offsets, bool local_only);
static int level_op(alfd_ctx *ctx, int l, const double *x, double *y);

// A CSR matrix in plain device arrays (intermediate products of the Galerkin setup)
struct DevRawCsr {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int64_t *rp = nullptr;
  int32_t *col = nullptr;
  double *val = nullptr;
  void release() {
    if (rp) hipFree(rp);
    if (col) hipFree(col);
    if (val) hipFree(val);
    rp = nullptr, col = nullptr, val = nullptr;
  }
};

// out = A * P on the device (spgemm_rows_kernel: the canonical fma chains of spgemm_host, bit for bit).
// *fits = false when a row has more distinct columns than the kernel's hash set holds (nothing is returned then).
static int dev_spgemm(alfd_ctx *ctx, int64_t nrows, const int64_t *arp, const int32_t *acol, const double *aval,
                      const int64_t *prp, const int32_t *pcol, const double *pval, int64_t ncols_out, DevRawCsr &out,
                      bool *fits) {
  *fits = false;
  out = DevRawCsr();
  out.nrows = nrows;
  out.ncols = ncols_out;
  int32_t *counts = nullptr, *ovf = nullptr;
  HIPC(hipMalloc((void **)&counts, std::max<int64_t>(nrows, 1) * sizeof(int32_t)));
  HIPC(hipMalloc((void **)&ovf, sizeof(int32_t)));
  HIPC(hipMemsetAsync(ovf, 0, sizeof(int32_t), ctx->stream));
  HIPC(hipMalloc((void **)&out.rp, (nrows + 1) * sizeof(int64_t)));
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nrows, 256 * 64));
  hipLaunchKernelGGL(spgemm_rows_kernel, dim3(grid), dim3(64), 0, ctx->stream, nrows, arp, acol, aval, prp, pcol, pval, 0,
                     counts, (const int64_t *)nullptr, (int32_t *)nullptr, (double *)nullptr, ovf);
  std::vector<int32_t> hc(nrows);
  int32_t hovf = 0;
  HIPC(hipMemcpyAsync(hc.data(), counts, nrows * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipMemcpyAsync(&hovf, ovf, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  if (hovf) {
    hipFree(counts);
    hipFree(ovf);
    out.release();
    return ALFD_OK;
  }
  std::vector<int64_t> rp(nrows + 1, 0);
  for (int64_t i = 0; i < nrows; ++i) rp[i + 1] = rp[i] + hc[i];
  out.nnz = rp[nrows];
  HIPC(hipMemcpyAsync(out.rp, rp.data(), (nrows + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMalloc((void **)&out.col, std::max<int64_t>(out.nnz, 1) * sizeof(int32_t)));
  HIPC(hipMalloc((void **)&out.val, std::max<int64_t>(out.nnz, 1) * sizeof(double)));
  hipLaunchKernelGGL(spgemm_rows_kernel, dim3(grid), dim3(64), 0, ctx->stream, nrows, arp, acol, aval, prp, pcol, pval, 1,
                     counts, (const int64_t *)out.rp, out.col, out.val, ovf);
  HIPC(hipGetLastError());
  HIPC(hipStreamSynchronize(ctx->stream));
  hipFree(counts);
  hipFree(ovf);
  *fits = true;
  return ALFD_OK;
}

static int download_raw(alfd_ctx *ctx, const DevRawCsr &d, HostCsr &h) {
  h.nrows = d.nrows;
  h.ncols = d.ncols;
  h.rp.resize(d.nrows + 1);
  h.col.resize(d.nnz);
  h.val.resize(d.nnz);
  HIPC(hipMemcpyAsync(h.rp.data(), d.rp, (d.nrows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  if (d.nnz) {
    HIPC(hipMemcpyAsync(h.col.data(), d.col, d.nnz * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipMemcpyAsync(h.val.data(), d.val, d.nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  return ALFD_OK;
}

// Host CSR with the shape of the device matrix A but only the rows that reach the interface patch filled in (the rows
// with a column in S, which contain S itself): found and gathered on the device, so that the patch setup does not need
// a host copy of A (21 GB at N = 74).  pos[j] >= 0 marks the columns of S.
static int fetch_patch_rows(alfd_ctx *ctx, const DevCsr &A, const std::vector<int32_t> &pos, HostCsr &out,
                            std::vector<int32_t> &Trows) {
  const int64_t n = A.nrows;
  int32_t *d_pos = nullptr, *d_rows = nullptr, *d_col = nullptr;
  uint8_t *d_flag = nullptr;
  int64_t *d_orp = nullptr;
  double *d_val = nullptr;
  HIPC(hipMalloc((void **)&d_pos, std::max<int64_t>(A.ncols, 1) * sizeof(int32_t)));
  HIPC(hipMalloc((void **)&d_flag, std::max<int64_t>(n, 1)));
  HIPC(hipMemcpyAsync(d_pos, pos.data(), pos.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(rows_touching_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, n, A.rp, A.col, d_pos, d_flag);
  std::vector<uint8_t> flag(n);
  std::vector<int64_t> rp(n + 1);
  HIPC(hipMemcpyAsync(flag.data(), d_flag, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipMemcpyAsync(rp.data(), A.rp, (n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  Trows.clear();
  for (int64_t i = 0; i < n; ++i)
    if (flag[i]) Trows.push_back((int32_t)i);
  const int64_t nt = (int64_t)Trows.size();
  std::vector<int64_t> orp(nt + 1, 0);
  for (int64_t q = 0; q < nt; ++q) orp[q + 1] = orp[q] + (rp[Trows[q] + 1] - rp[Trows[q]]);
  const int64_t nnz = orp[nt];
  HIPC(hipMalloc((void **)&d_rows, std::max<int64_t>(nt, 1) * sizeof(int32_t)));
  HIPC(hipMalloc((void **)&d_orp, (nt + 1) * sizeof(int64_t)));
  HIPC(hipMalloc((void **)&d_col, std::max<int64_t>(nnz, 1) * sizeof(int32_t)));
  HIPC(hipMalloc((void **)&d_val, std::max<int64_t>(nnz, 1) * sizeof(double)));
  HIPC(hipMemcpyAsync(d_rows, Trows.data(), nt * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(d_orp, orp.data(), (nt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  if (nt > 0)
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, ctx->stream, nt, d_rows, A.rp, A.col,
                       A.val, d_orp, d_col, d_val);
  out.nrows = n;
  out.ncols = A.ncols;
  out.col.resize(nnz);
  out.val.resize(nnz);
  HIPC(hipMemcpyAsync(out.col.data(), d_col, nnz * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipMemcpyAsync(out.val.data(), d_val, nnz * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  out.rp.assign(n + 1, 0);
  for (int64_t q = 0; q < nt; ++q) out.rp[Trows[q] + 1] = orp[q + 1] - orp[q];
  for (int64_t i = 0; i < n; ++i) out.rp[i + 1] += out.rp[i];
  for (void *q : {(void *)d_pos, (void *)d_flag, (void *)d_rows, (void *)d_orp, (void *)d_col, (void *)d_val}) hipFree(q);
  return ALFD_OK;
}

// S, the patch operators and lambda_max(D^-1 Aug_SS); A / C / Ct are the host copies of the level-0 operators
static int patch_setup(alfd_ctx *ctx, const HostCsr *A_full, const HostCsr &C, const HostCsr &Ct) {
  alfd_ctx::Patch &Q = ctx->patch;
  const alfd_config &c = ctx->cfg;
  const int64_t n = ctx->n[0];
  std::vector<int32_t> S, Trows, pos(n, -1);
  for (int64_t i = 0; i < n; ++i)
    if (Ct.rp[i + 1] > Ct.rp[i]) {
      pos[i] = (int32_t)S.size();
      S.push_back((int32_t)i);
    }
  const int64_t m = (int64_t)S.size();
  if (m == 0) return ALFD_OK;
  HostCsr A_patch;      // without a host copy of A: only the rows that reach the patch, gathered on the device
  if (!A_full) RC(fetch_patch_rows(ctx, ctx->mat[ALFD_A], pos, A_patch, Trows));
  const HostCsr &A = A_full ? *A_full : A_patch;
  if (A_full) {
    // rows of A with a column in S (the rows E^T-corrections reach): scanned in parallel, kept in order
    const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::vector<int32_t>> part(T);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
          bool hit = false;
          for (int64_t k = A.rp[i]; k < A.rp[i + 1] && !hit; ++k) hit = pos[A.col[k]] >= 0;
          if (hit) part[t].push_back((int32_t)i);
        }
      });
    for (auto &x : th) x.join();
    for (auto &v : part) Trows.insert(Trows.end(), v.begin(), v.end());
  }
  std::vector<int32_t> lam_rows(C.nrows);
  for (int64_t k = 0; k < C.nrows; ++k) lam_rows[k] = (int32_t)k;
  HostCsr h;
  extract_host(A, S, pos.data(), m, false, h);
  RC(upload_level(ctx, Q.Ass, h));
  extract_host(A, S, nullptr, n, false, h);
  RC(upload_level(ctx, Q.As, h));
  extract_host(A, Trows, pos.data(), m, true, h);
  RC(upload_level(ctx, Q.Ats, h));
  extract_host(C, lam_rows, pos.data(), m, false, h);
  RC(upload_level(ctx, Q.Cs, h));
  extract_host(Ct, S, nullptr, Ct.ncols, false, h);
  RC(upload_level(ctx, Q.Cts, h));
  Q.m = m;
  Q.mpad = pad_chunk(m);
  void *q = nullptr;
  HIPC(hipMalloc(&q, m * sizeof(int32_t)));
  ctx->ws_allocs.push_back(q);
  Q.S = static_cast<int32_t *>(q);
  HIPC(hipMemcpyAsync(Q.S, S.data(), m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  for (double **v : {&Q.dinv, &Q.rS, &Q.zS, &Q.uS, &Q.eS, &Q.cd, &Q.cres, &Q.ctmp}) RC(ws_alloc_zero(ctx, v, Q.mpad));
  RC(ws_alloc_zero(ctx, &Q.rr, pad_chunk(n)));
  const unsigned gm = (unsigned)((m + 255) / 256);
  hipLaunchKernelGGL(gather_kernel, dim3(gm), dim3(256), 0, ctx->stream, m, Q.S, ctx->dinv_aug, Q.dinv);
  // lambda_max(D^-1 Aug_SS): power iteration from the integer-hash vector (patch-compact index)
  double *v = Q.rS, *wv = Q.zS;
  hipLaunchKernelGGL(hash_vector_kernel, dim3(gm), dim3(256), 0, ctx->stream, m, (int64_t)0, v);
  double lam = 0;
  for (int it = 0; it < c.cheb_power_its; ++it) {
    RC(dot_async(ctx, Q.mpad, v, v, S_TMP));
    RC(read_scalars(ctx, S_TMP, 1));
    VEC_LAUNCH(scale_kernel, Q.mpad, 16, (const double *)nullptr, 0, 0, 1.0 / std::sqrt(ctx->sc_host[S_TMP]), v);
    RC(patch_op(ctx, v, wv));
    VEC_LAUNCH(pmul_scale_kernel, Q.mpad, 24, 1.0, Q.dinv, wv, wv);
    RC(dot_async(ctx, Q.mpad, wv, wv, S_TMP));
    RC(read_scalars(ctx, S_TMP, 1));
    lam = std::sqrt(ctx->sc_host[S_TMP]);
    std::swap(v, wv);
  }
  Q.lmax = lam * c.cheb_safety;
  HIPC(hipMemsetAsync(Q.rS, 0, Q.mpad * sizeof(double), ctx->stream));
  HIPC(hipMemsetAsync(Q.zS, 0, Q.mpad * sizeof(double), ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  Q.on = true;
  if (c.log_level > 0 && ctx->rank == 0)
    std::fprintf(stderr, "[alfd] interface patch: %lld rows, A[S,S] %lld nnz, %lld rows of A touch it, lambda_max %.3f\n",
                 (long long)m, (long long)Q.Ass.nnz, (long long)Trows.size(), Q.lmax);
  return ALFD_OK;
}

// Explicit inverse of the coarsest Aug_c = A_c + gamma Ct_c W^-1 C_c: dense Cholesky on the host, every entry a
// sequential fma chain (the oracle repeats the loops), uploaded as a dense CSR so that z = Aug_c^-1 r runs in
// the canonical SpMV order.  ML solves its coarsest level directly as well (KLU, utilities.h:304-317).
static int coarse_inverse(alfd_ctx *ctx, const HostCsr &A, const HostCsr &C, const HostCsr &Ct, const std::vector<double> &w) {
  const int64_t n = A.nrows;
  std::vector<double> D((size_t)n * n, 0.0);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; ++k) D[i * n + A.col[k]] = A.val[k];
  if (!ctx->cfg.aug_assembled)
    for (int64_t i = 0; i < n; ++i)
      for (int64_t k = Ct.rp[i]; k < Ct.rp[i + 1]; ++k) {
        const int64_t lam = Ct.col[k];
        const double sfac = (ctx->cfg.gamma * w[lam]) * Ct.val[k];
        for (int64_t e = C.rp[lam]; e < C.rp[lam + 1]; ++e)
          D[i * n + C.col[e]] = std::fma(sfac, C.val[e], D[i * n + C.col[e]]);
      }
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  auto par_for = [&](int64_t lo, int64_t hi, const std::function<void(int64_t)> &body) {
    if (hi - lo < 256) {
      for (int64_t i = lo; i < hi; ++i) body(i);
      return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        for (int64_t i = lo + (hi - lo) * t / T; i < lo + (hi - lo) * (t + 1) / T; ++i) body(i);
      });
    for (auto &x : th) x.join();
  };
  for (int64_t j = 0; j < n; ++j) {
    double d = D[j * n + j];
    for (int64_t k = 0; k < j; ++k) d = std::fma(-D[j * n + k], D[j * n + k], d);
    if (!(d > 0.0)) return ctx->err = "coarsest multigrid operator is not positive definite", ALFD_E_BREAKDOWN;
    const double ljj = std::sqrt(d);
    D[j * n + j] = ljj;
    par_for(j + 1, n, [&](int64_t i) {
      double sacc = D[i * n + j];
      for (int64_t k = 0; k < j; ++k) sacc = std::fma(-D[i * n + k], D[j * n + k], sacc);
      D[i * n + j] = sacc / ljj;
    });
  }
  HostCsr X;
  X.nrows = X.ncols = n;
  X.rp.resize(n + 1);
  X.col.resize((size_t)n * n);
  X.val.assign((size_t)n * n, 0.0);
  for (int64_t i = 0; i <= n; ++i) X.rp[i] = i * n;
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = 0; j < n; ++j) X.col[i * n + j] = (int32_t)j;
  {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t]() {
        std::vector<double> y(n), x(n);
        for (int64_t c = n * t / T; c < n * (t + 1) / T; ++c) {
          std::fill(y.begin(), y.end(), 0.0);
          for (int64_t i = c; i < n; ++i) {
            double sacc = i == c ? 1.0 : 0.0;
            for (int64_t k = c; k < i; ++k) sacc = std::fma(-D[i * n + k], y[k], sacc);
            y[i] = sacc / D[i * n + i];
          }
          for (int64_t i = n - 1; i >= 0; --i) {
            double sacc = y[i];
            for (int64_t k = i + 1; k < n; ++k) sacc = std::fma(-D[k * n + i], x[k], sacc);
            x[i] = sacc / D[i * n + i];
          }
          for (int64_t i = 0; i < n; ++i) X.val[i * n + c] = x[i];
        }
      });
    for (auto &x : th) x.join();
  }
  RC(upload_level_part(ctx, ctx->ml_inv, X, nullptr, true));   // never partitioned: every rank that has it has it whole
  return ALFD_OK;
}

// Aggregate ids over the LOCAL index space [owned | halo] of a matrix' column space:
// the owned part is given, the halo part is fetched from the owners through the
// matrix' own halo plan (ids travel as exactly representable doubles).
static int exchange_ids(alfd_ctx *ctx, DevCsr &m, const std::vector<int32_t> &owned, std::vector<int32_t> &all) {
  all.assign(owned.begin(), owned.end());
  if (ctx->nranks == 1) return ALFD_OK;
  std::vector<double> v(owned.begin(), owned.end());
  double *d = nullptr;
  HIPC(hipMalloc((void **)&d, std::max<size_t>(v.size(), 1) * sizeof(double)));
  HIPC(hipMemcpyAsync(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  int rc = ALFD_OK;
  if (ctx->local || ctx->host_alltoallv || m.n_halo > 0 || m.send_off.back() > 0) rc = halo_exchange(ctx, m, d);
  if (rc == ALFD_OK && m.n_halo > 0) {
    std::vector<double> h(m.n_halo);
    hipMemcpyAsync(h.data(), m.halo, m.n_halo * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
    for (double x : h) all.push_back((int32_t)x);
  }
  hipStreamSynchronize(ctx->stream);
  hipFree(d);
  return rc;
}

static int upload_level_part(alfd_ctx *ctx, DevCsr &dst, const HostCsr &h, const int64_t *col_offsets,
                             bool local_only) {
  ctx->up_col_offsets = col_offsets;
  ctx->up_local_only = local_only;
  const int rc = upload_level(ctx, dst, h);
  ctx->up_col_offsets = nullptr;
  ctx->up_local_only = false;
  return rc;
}

// All ranks contribute `bytes` bytes (sizes may differ); out = the contributions in rank order.
static int allgather_bytes(alfd_ctx *ctx, const void *data, size_t bytes, std::vector<char> &out,
                           std::vector<size_t> &sizes) {
  const int P = ctx->nranks;
  int64_t mine = (int64_t)bytes;
  std::vector<int64_t> all(P);
  int64_t *d_m = nullptr, *d_a = nullptr;
  HIPC(hipMalloc((void **)&d_m, sizeof(int64_t)));
  HIPC(hipMalloc((void **)&d_a, sizeof(int64_t) * P));
  HIPC(hipMemcpyAsync(d_m, &mine, sizeof(mine), hipMemcpyHostToDevice, ctx->stream));
  int rc = comm_allgather(ctx, d_m, d_a, sizeof(int64_t));
  if (rc == ALFD_OK) {
    hipMemcpyAsync(all.data(), d_a, sizeof(int64_t) * P, hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
  }
  hipFree(d_m);
  hipFree(d_a);
  if (rc != ALFD_OK) return rc;
  size_t maxb = 8;
  sizes.assign(P, 0);
  for (int p = 0; p < P; ++p) {
    sizes[p] = (size_t)all[p];
    maxb = std::max(maxb, (sizes[p] + 7) / 8 * 8);
  }
  char *d_s = nullptr, *d_r = nullptr;
  HIPC(hipMalloc((void **)&d_s, maxb));
  HIPC(hipMalloc((void **)&d_r, maxb * P));
  HIPC(hipMemsetAsync(d_s, 0, maxb, ctx->stream));
  if (bytes) HIPC(hipMemcpyAsync(d_s, data, bytes, hipMemcpyHostToDevice, ctx->stream));
  rc = comm_allgather(ctx, d_s, d_r, maxb);
  std::vector<char> stage(maxb * P);
  if (rc == ALFD_OK) {
    hipMemcpyAsync(stage.data(), d_r, stage.size(), hipMemcpyDeviceToHost, ctx->stream);
    hipStreamSynchronize(ctx->stream);
  }
  hipFree(d_s);
  hipFree(d_r);
  if (rc != ALFD_OK) return rc;
  out.clear();
  for (int p = 0; p < P; ++p) out.insert(out.end(), stage.begin() + p * maxb, stage.begin() + p * maxb + sizes[p]);
  return ALFD_OK;
}

// Global CSR (rows of all ranks in rank order) from every rank's local rows.  shift: per-rank
// amount added to the column ids of that rank's piece (rank-local ids -> global), or nullptr.
static int gather_csr(alfd_ctx *ctx, const HostCsr &loc, const int64_t *shift, int64_t ncols_global, HostCsr &g) {
  std::vector<int32_t> len(loc.nrows);
  for (int64_t r = 0; r < loc.nrows; ++r) len[r] = (int32_t)(loc.rp[r + 1] - loc.rp[r]);
  std::vector<char> blen, bcol, bval;
  std::vector<size_t> slen, scol, sval;
  RC(allgather_bytes(ctx, len.data(), len.size() * 4, blen, slen));
  RC(allgather_bytes(ctx, loc.col.data(), (size_t)loc.nnz() * 4, bcol, scol));
  RC(allgather_bytes(ctx, loc.val.data(), (size_t)loc.nnz() * 8, bval, sval));
  const int64_t nrows = (int64_t)(blen.size() / 4), nnz = (int64_t)(bcol.size() / 4);
  g.nrows = nrows;
  g.ncols = ncols_global;
  g.rp.assign(nrows + 1, 0);
  const int32_t *gl = reinterpret_cast<const int32_t *>(blen.data());
  for (int64_t r = 0; r < nrows; ++r) g.rp[r + 1] = g.rp[r] + gl[r];
  g.col.resize(nnz);
  g.val.resize(nnz);
  std::memcpy(g.col.data(), bcol.data(), (size_t)nnz * 4);
  std::memcpy(g.val.data(), bval.data(), (size_t)nnz * 8);
  if (g.rp[nrows] != nnz) return ctx->err = "gather_csr: inconsistent pieces", ALFD_E_COMM;
  if (shift) {
    int64_t k = 0;
    for (int p = 0; p < ctx->nranks; ++p) {
      const int64_t np = (int64_t)(scol[p] / 4);
      for (int64_t q = 0; q < np; ++q) g.col[k + q] = (int32_t)(g.col[k + q] + shift[p]);
      k += np;
    }
  }
  return ALFD_OK;
}

// device vector pieces (n_local each, rank order) -> global device vector on every rank
static int gather_vec(alfd_ctx *ctx, const double *d_local, int64_t n_local, double *d_global) {
  std::vector<double> h(n_local);
  HIPC(hipMemcpyAsync(h.data(), d_local, n_local * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  std::vector<char> out;
  std::vector<size_t> sizes;
  RC(allgather_bytes(ctx, h.data(), h.size() * 8, out, sizes));
  HIPC(hipMemcpyAsync(d_global, out.data(), out.size(), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return ALFD_OK;
}

// ---------------------------------------------------------------------------
// Algebraic aggregation (alfd_build_aggregates): the counterpart of ML's uncoupled aggregation
// (utilities.h:304-317: aggregation_threshold, constant modes per component) for operators that
// come without grid information -- a matrix replayed from an .alfd file, a locally refined mesh
// with hanging-node rows.  Unknowns are taken node-major with `bs` components per node.  Per
// level: node graph with w_IJ = max |a_ij| over the bs x bs block, strong if
// w_IJ >= theta sqrt(d_I d_J) (d = max |a_ii| of the node); rows that hold only their diagonal
// (Dirichlet / constrained rows) stay out (-1).  Greedy passes in natural order (deterministic):
// (1) a free node whose strong neighbours are all free founds an aggregate with its strongest
// free neighbours (at most max_size nodes); (2) leftovers join the founded aggregate they are most
// strongly tied to; (3) what is still free founds aggregates of its own.  The next level's graph
// is that of the Galerkin product with the piecewise-constant prolongator.
static void aggregate_level(const HostCsr &A, int bs, double theta, int max_size, std::vector<int32_t> &agg,
                            int64_t &n_coarse) {
  const int64_t n = A.nrows, nn = n / bs;
  std::vector<double> d(nn, 0.0);
  std::vector<char> fixed(nn, 1);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t I = i / bs;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      if (A.col[k] == i) d[I] = std::max(d[I], std::fabs(A.val[k]));
      else if (A.val[k] != 0.0) fixed[I] = 0;
    }
  }
  // node graph (strong edges only), CSR with weights
  std::vector<int64_t> gp(nn + 1, 0);
  std::vector<int32_t> gc;
  std::vector<double> gw;
  {
    std::vector<double> w(nn, 0.0);
    std::vector<int32_t> touched;
    for (int64_t I = 0; I < nn; ++I) {
      touched.clear();
      if (!fixed[I])
        for (int64_t i = I * bs; i < (I + 1) * bs; ++i)
          for (int64_t k = A.rp[i]; k < A.rp[i + 1]; ++k) {
            const int64_t J = A.col[k] / bs;
            if (J == I || J >= nn || fixed[J]) continue;
            const double v = std::fabs(A.val[k]);
            if (v == 0.0) continue;
            if (w[J] == 0.0) touched.push_back((int32_t)J);
            w[J] = std::max(w[J], v);
          }
      std::sort(touched.begin(), touched.end());
      for (int32_t J : touched) {
        if (w[J] >= theta * std::sqrt(d[I] * d[J])) {
          gc.push_back(J);
          gw.push_back(w[J]);
        }
        w[J] = 0.0;
      }
      gp[I + 1] = (int64_t)gc.size();
    }
  }
  std::vector<int32_t> na(nn, -1);  // node -> aggregate
  int32_t nagg = 0;
  std::vector<std::pair<double, int32_t>> cand;
  // pass 1
  for (int64_t I = 0; I < nn; ++I) {
    if (fixed[I] || na[I] >= 0) continue;
    bool all_free = true;
    for (int64_t k = gp[I]; k < gp[I + 1] && all_free; ++k) all_free = na[gc[k]] < 0;
    if (!all_free) continue;
    cand.clear();
    for (int64_t k = gp[I]; k < gp[I + 1]; ++k) cand.emplace_back(-gw[k], gc[k]);
    std::sort(cand.begin(), cand.end());   // strongest first, ties by node id
    na[I] = nagg;
    for (size_t q = 0; q < cand.size() && (int)q + 1 < max_size; ++q) na[cand[q].second] = nagg;
    ++nagg;
  }
  // pass 2: join the most strongly tied aggregate of pass 1
  const int32_t nagg1 = nagg;
  std::vector<int32_t> join(nn, -1);
  for (int64_t I = 0; I < nn; ++I) {
    if (fixed[I] || na[I] >= 0) continue;
    double best = 0.0;
    for (int64_t k = gp[I]; k < gp[I + 1]; ++k) {
      const int32_t a = na[gc[k]];
      if (a >= 0 && a < nagg1 && gw[k] > best) best = gw[k], join[I] = a;
    }
  }
  for (int64_t I = 0; I < nn; ++I)
    if (join[I] >= 0) na[I] = join[I];
  // pass 3: the rest founds its own aggregates (with whatever free strong neighbours are left)
  for (int64_t I = 0; I < nn; ++I) {
    if (fixed[I] || na[I] >= 0) continue;
    na[I] = nagg;
    int taken = 1;
    for (int64_t k = gp[I]; k < gp[I + 1] && taken < max_size; ++k)
      if (na[gc[k]] < 0) na[gc[k]] = nagg, ++taken;
    ++nagg;
  }
  agg.assign(n, -1);
  for (int64_t i = 0; i < nn * bs; ++i)
    if (na[i / bs] >= 0) agg[i] = (int32_t)(na[i / bs] * bs + i % bs);
  n_coarse = (int64_t)nagg * bs;
}

// Rows of a row-partitioned CSR matrix (this rank holds the global rows [row_off[rank], row_off[rank + 1]) as `loc`,
// any column space) for an arbitrary list of GLOBAL row ids: out gets one row per entry of `want`, in that order.
// Collective (setup only): requests and answers travel through all-gathers, so every rank sees all of them.
static int fetch_rows(alfd_ctx *ctx, const HostCsr &loc, const int64_t *row_off, const std::vector<int64_t> &want,
                      HostCsr &out) {
  const int P = ctx->nranks, me = ctx->rank;
  std::vector<char> breq, bans;
  std::vector<size_t> sreq, sans;
  RC(allgather_bytes(ctx, want.data(), want.size() * sizeof(int64_t), breq, sreq));
  std::vector<std::vector<int64_t>> req(P);
  {
    size_t pos = 0;
    for (int p = 0; p < P; ++p) {
      req[p].resize(sreq[p] / sizeof(int64_t));
      std::memcpy(req[p].data(), breq.data() + pos, sreq[p]);
      pos += sreq[p];
    }
  }
  // my answers: for every requester, the rows I own in its request order: int64 len, int32 cols (padded to 8), double vals
  std::vector<char> ans;
  auto put = [&](const void *q, size_t n) { ans.insert(ans.end(), (const char *)q, (const char *)q + n); };
  for (int p = 0; p < P; ++p)
    for (int64_t g : req[p]) {
      if (g < row_off[me] || g >= row_off[me + 1]) continue;
      const int64_t i = g - row_off[me], k0 = loc.rp[i], len = loc.rp[i + 1] - k0;
      put(&len, 8);
      put(loc.col.data() + k0, (size_t)len * 4);
      if (len & 1) {
        const int32_t z = 0;
        put(&z, 4);
      }
      put(loc.val.data() + k0, (size_t)len * 8);
    }
  RC(allgather_bytes(ctx, ans.data(), ans.size(), bans, sans));
  out.nrows = (int64_t)want.size();
  out.ncols = loc.ncols;
  out.rp.assign(1, 0);
  out.col.clear();
  out.val.clear();
  std::vector<size_t> base(P + 1, 0);
  for (int o = 0; o < P; ++o) base[o + 1] = base[o] + sans[o];
  // walk every owner's blob in the order it was written; keep what was meant for me, placed by request position
  std::vector<std::pair<const char *, int64_t>> mine(want.size(), {nullptr, 0});
  for (int o = 0; o < P; ++o) {
    const char *q = bans.data() + base[o];
    for (int p = 0; p < P; ++p)
      for (size_t t = 0; t < req[p].size(); ++t) {
        const int64_t g = req[p][t];
        if (g < row_off[o] || g >= row_off[o + 1]) continue;
        int64_t len;
        std::memcpy(&len, q, 8);
        if (p == me) mine[t] = {q + 8, len};
        q += 8 + (size_t)((len + 1) / 2 * 2) * 4 + (size_t)len * 8;
      }
  }
  for (size_t t = 0; t < want.size(); ++t) {
    if (!mine[t].first) return ctx->err = "fetch_rows: a requested row has no owner", ALFD_E_COMM;
    const int64_t len = mine[t].second;
    const int32_t *c = (const int32_t *)mine[t].first;
    const double *v = (const double *)(mine[t].first + (size_t)((len + 1) / 2 * 2) * 4);
    out.col.insert(out.col.end(), c, c + len);
    out.val.insert(out.val.end(), v, v + len);
    out.rp.push_back((int64_t)out.col.size());
  }
  return ALFD_OK;
}

static int upload_raw(alfd_ctx *ctx, const HostCsr &h, DevRawCsr &d) {
  d = DevRawCsr();
  d.nrows = h.nrows;
  d.ncols = h.ncols;
  d.nnz = h.nnz();
  HIPC(hipMalloc((void **)&d.rp, (h.nrows + 1) * sizeof(int64_t)));
  HIPC(hipMalloc((void **)&d.col, std::max<int64_t>(d.nnz, 1) * sizeof(int32_t)));
  HIPC(hipMalloc((void **)&d.val, std::max<int64_t>(d.nnz, 1) * sizeof(double)));
  HIPC(hipMemcpyAsync(d.rp, h.rp.data(), (h.nrows + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  if (d.nnz) {
    HIPC(hipMemcpyAsync(d.col, h.col.data(), d.nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipMemcpyAsync(d.val, h.val.data(), d.nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  return ALFD_OK;
}

// out = A * Pm with the device kernel when it fits, else on the host (both in the canonical order)
static int product(alfd_ctx *ctx, const HostCsr &A, const HostCsr &Pm, HostCsr &out) {
  if (ctx->ml_gpu_galerkin && A.nnz() > 200000) {
    DevRawCsr dA, dP, dC;
    RC(upload_raw(ctx, A, dA));
    RC(upload_raw(ctx, Pm, dP));
    bool fits = false;
    const int rc = dev_spgemm(ctx, A.nrows, dA.rp, dA.col, dA.val, dP.rp, dP.col, dP.val, Pm.ncols, dC, &fits);
    dA.release();
    dP.release();
    if (rc != ALFD_OK) return rc;
    if (fits) {
      const int rc2 = download_raw(ctx, dC, out);
      dC.release();
      return rc2;
    }
  }
  spgemm_host(A, Pm, out);
  return ALFD_OK;
}

static int coarse_inverse(alfd_ctx *ctx, const HostCsr &A, const HostCsr &C, const HostCsr &Ct, const std::vector<double> &w);
static int patch_setup_rep(alfd_ctx *ctx);

// The interface patch of a partitioned context: S is split by row ownership, the patch operators are gathered whole on
// every rank (they are small), the polynomial runs redundantly; see ml_apply_rep.
static int patch_setup_rep(alfd_ctx *ctx) {
  alfd_ctx::Patch &Q = ctx->patch;
  const alfd_config &c = ctx->cfg;
  const int P = ctx->nranks, rk = ctx->rank, last = ctx->nblocks - 1;
  const std::vector<int64_t> &off0 = ctx->part[0], &offl = ctx->part[last];
  DevCsr &dA = ctx->mat[ALFD_A], &dC = ctx->mat[ALFD_C], &dCt = ctx->mat[ALFD_CT];
  const int64_t n_loc = ctx->n[0];
  HostCsr Ct, C;
  RC(download_csr(ctx, dCt, Ct));
  RC(download_csr(ctx, dC, C));
  std::vector<int32_t> S;
  for (int64_t i = 0; i < n_loc; ++i)
    if (Ct.rp[i + 1] > Ct.rp[i]) S.push_back((int32_t)i);
  // global S: every rank's list of global row ids, in rank order
  std::vector<int64_t> Sg(S.size());
  for (size_t q = 0; q < S.size(); ++q) Sg[q] = off0[rk] + S[q];
  std::vector<char> ball;
  std::vector<size_t> sz;
  RC(allgather_bytes(ctx, Sg.data(), Sg.size() * sizeof(int64_t), ball, sz));
  Q.soff.assign(P + 1, 0);
  for (int p = 0; p < P; ++p) Q.soff[p + 1] = Q.soff[p] + (int64_t)(sz[p] / sizeof(int64_t));
  const int64_t m = Q.soff[P];
  if (m == 0) return ALFD_OK;
  std::vector<int64_t> Sall(m);
  std::memcpy(Sall.data(), ball.data(), (size_t)m * sizeof(int64_t));     // ascending: rank order = global order
  auto patch_id = [&](int64_t g) -> int32_t {
    auto it = std::lower_bound(Sall.begin(), Sall.end(), g);
    return it != Sall.end() && *it == g ? (int32_t)(it - Sall.begin()) : -1;
  };
  // patch ids over the LOCAL column spaces [owned | halo] of A and of C
  auto make_pos = [&](const DevCsr &mtx, std::vector<int32_t> &pos) {
    pos.assign((size_t)mtx.n_local_cols + mtx.halo_globals.size(), -1);
    for (size_t q = 0; q < S.size(); ++q) pos[S[q]] = (int32_t)(Q.soff[rk] + (int64_t)q);
    for (size_t h = 0; h < mtx.halo_globals.size(); ++h) pos[mtx.n_local_cols + h] = patch_id(mtx.halo_globals[h]);
  };
  std::vector<int32_t> posA, posC, Trows;
  make_pos(dA, posA);
  make_pos(dC, posC);
  HostCsr A_patch, h, g;
  RC(fetch_patch_rows(ctx, dA, posA, A_patch, Trows));
  // replicated operators: my rows, then gathered in rank order
  extract_host(A_patch, S, posA.data(), m, false, h);
  RC(gather_csr(ctx, h, nullptr, m, g));
  RC(upload_level_part(ctx, Q.Ass, g, nullptr, true));
  std::vector<int32_t> lam_rows(C.nrows);
  for (int64_t k = 0; k < C.nrows; ++k) lam_rows[k] = (int32_t)k;
  extract_host(C, lam_rows, posC.data(), m, false, h);
  RC(gather_csr(ctx, h, nullptr, m, g));
  RC(upload_level_part(ctx, Q.Cs, g, nullptr, true));
  // Ct rows over GLOBAL multiplier ids
  std::vector<int32_t> lamg((size_t)dCt.n_local_cols + dCt.halo_globals.size());
  for (int32_t j = 0; j < dCt.n_local_cols; ++j) lamg[j] = (int32_t)(offl[rk] + j);
  for (size_t j = 0; j < dCt.halo_globals.size(); ++j) lamg[dCt.n_local_cols + j] = dCt.halo_globals[j];
  HostCsr Ctg = Ct;
  Ctg.ncols = offl.back();
  for (int64_t i = 0; i < Ctg.nrows; ++i) {   // rows re-sorted by the global id (the local order puts halo ids last)
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t k = Ct.rp[i]; k < Ct.rp[i + 1]; ++k) row.emplace_back(lamg[Ct.col[k]], Ct.val[k]);
    std::sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    for (size_t k = 0; k < row.size(); ++k) Ctg.col[Ct.rp[i] + k] = row[k].first, Ctg.val[Ct.rp[i] + k] = row[k].second;
  }
  extract_host(Ctg, S, nullptr, Ctg.ncols, false, h);
  RC(upload_level_part(ctx, Q.Cts_loc, h, nullptr, true));
  RC(gather_csr(ctx, h, nullptr, Ctg.ncols, g));
  RC(upload_level_part(ctx, Q.Cts, g, nullptr, true));
  RC(upload_level_part(ctx, Q.Ctg, Ctg, nullptr, true));
  Q.Ass.rep = Q.Cs.rep = Q.Cts.rep = true;
  // my rows of A[:, S] (patch ids, local-only) and of A[S, :] (global fine ids, halo on the fine vector)
  extract_host(A_patch, Trows, posA.data(), m, true, h);
  RC(upload_level_part(ctx, Q.Ats, h, nullptr, true));
  extract_host(A_patch, S, nullptr, A_patch.ncols, false, h);
  for (size_t k = 0; k < h.col.size(); ++k)
    h.col[k] = (int32_t)(h.col[k] < dA.n_local_cols ? off0[rk] + h.col[k] : dA.halo_globals[h.col[k] - dA.n_local_cols]);
  h.ncols = off0.back();
  RC(upload_level_part(ctx, Q.As, h, off0.data(), false));
  Q.m = m;
  Q.mpad = pad_chunk(m);
  Q.m_loc = (int64_t)S.size();
  for (int p = 0; p < P; ++p) {
    Q.piece = std::max(Q.piece, Q.soff[p + 1] - Q.soff[p]);
    Q.lam_piece = std::max(Q.lam_piece, offl[p + 1] - offl[p]);
  }
  void *q = nullptr;
  HIPC(hipMalloc(&q, std::max<int64_t>(Q.m_loc, 1) * sizeof(int32_t)));
  ctx->ws_allocs.push_back(q);
  Q.S = static_cast<int32_t *>(q);
  if (Q.m_loc) HIPC(hipMemcpyAsync(Q.S, S.data(), Q.m_loc * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  for (double **v : {&Q.dinv, &Q.rS, &Q.zS, &Q.uS, &Q.eS, &Q.cd, &Q.cres, &Q.ctmp}) RC(ws_alloc_zero(ctx, v, Q.mpad));
  RC(ws_alloc_zero(ctx, &Q.rr, pad_chunk(n_loc)));
  RC(ws_alloc_zero(ctx, &Q.loc, pad_chunk(std::max<int64_t>(Q.m_loc, 1))));
  RC(ws_alloc_zero(ctx, &Q.send, Q.piece));
  RC(ws_alloc_zero(ctx, &Q.stage, Q.piece * P));
  RC(ws_alloc_zero(ctx, &Q.lam_send, Q.lam_piece));
  RC(ws_alloc_zero(ctx, &Q.lam_stage, Q.lam_piece * P));
  Q.rep = true;
  // 1 / diag(Aug) on S: my entries, gathered
  const unsigned gm = (unsigned)std::max<int64_t>(1, (Q.m_loc + 255) / 256);
  if (Q.m_loc) hipLaunchKernelGGL(gather_kernel, dim3(gm), dim3(256), 0, ctx->stream, Q.m_loc, Q.S, ctx->dinv_aug, Q.loc);
  RC(allgather_pieces(ctx, Q.loc, Q.soff, Q.piece, Q.send, Q.stage, Q.dinv));
  // lambda_max(D^-1 Aug_SS) on the replicated patch: plain reductions, the same on every rank
  ctx->dots_replicated = true;
  double *v = Q.rS, *wv = Q.zS;
  hipLaunchKernelGGL(hash_vector_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, m, (int64_t)0, v);
  double lam = 0;
  int rc = ALFD_OK;
  for (int it = 0; it < c.cheb_power_its && rc == ALFD_OK; ++it) {
    auto step = [&]() -> int {
      RC(dot_async(ctx, Q.mpad, v, v, S_TMP));
      RC(read_scalars(ctx, S_TMP, 1));
      VEC_LAUNCH(scale_kernel, Q.mpad, 16, (const double *)nullptr, 0, 0, 1.0 / std::sqrt(ctx->sc_host[S_TMP]), v);
      RC(patch_op(ctx, v, wv));
      VEC_LAUNCH(pmul_scale_kernel, Q.mpad, 24, 1.0, Q.dinv, wv, wv);
      RC(dot_async(ctx, Q.mpad, wv, wv, S_TMP));
      RC(read_scalars(ctx, S_TMP, 1));
      lam = std::sqrt(ctx->sc_host[S_TMP]);
      std::swap(v, wv);
      return ALFD_OK;
    };
    rc = step();
  }
  ctx->dots_replicated = false;
  if (rc != ALFD_OK) return rc;
  Q.lmax = lam * c.cheb_safety;
  HIPC(hipMemsetAsync(Q.rS, 0, Q.mpad * sizeof(double), ctx->stream));
  HIPC(hipMemsetAsync(Q.zS, 0, Q.mpad * sizeof(double), ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  Q.on = true;
  return ALFD_OK;
}

// ALFD_PREC_MULTILEVEL through CSR prolongators on a ROW-PARTITIONED context (round 3): the fine level stays
// partitioned, every level below it -- and the interface patch -- is REPLICATED on all ranks.  Level 0: this rank holds
// the prolongator rows of its own fine unknowns (global coarse ids) and alfd_set_aggregate_partition(0, ...) names the
// rank that forms each coarse row; levels >= 1: the whole prolongator on every rank.  The hierarchy is the single-rank
// one bit for bit: a coarse row's Galerkin chain and restriction sum run over its fine rows in GLOBAL order, remote
// prolongator / A P rows are fetched from their owners (fetch_rows), finished rows are all-gathered.  Only the level-0
// reductions differ from a single-rank run (rank-ordered sums, as on every partitioned path).
static int ml_setup_rep_prolongators(alfd_ctx *ctx, int nlev) {
  const alfd_config &c = ctx->cfg;
  const int P = ctx->nranks, rk = ctx->rank, last = ctx->nblocks - 1;
  for (int l = 0; l < nlev; ++l)
    if (ctx->ml_P[l].rp.empty())
      return ctx->err = "partitioned context: CSR prolongators and aggregates cannot be mixed", ALFD_E_UNSUPPORTED;
  const std::vector<int64_t> &off0 = ctx->part[0], &coff = ctx->ml_coff[0];
  const HostCsr &P0 = ctx->ml_P[0];
  const int64_t n_loc = ctx->n[0], n1 = P0.ncols;
  if ((int)coff.size() != P + 1 || coff.back() != n1 || P0.nrows != n_loc)
    return ctx->err = "partitioned CSR prolongator: level 0 needs this rank's rows and alfd_set_aggregate_partition(0, coarse offsets)",
           ALFD_E_INVALID;
  for (int l = 1; l < nlev; ++l)
    if (ctx->ml_P[l].nrows != ctx->ml_P[l - 1].ncols)
      return ctx->err = "partitioned CSR prolongator: levels >= 1 must be given whole on every rank", ALFD_E_INVALID;
  free_levels(ctx);
  ctx->ml.assign(nlev + 1, MlLevel());
  // ---- level 0 (partitioned): vectors only, operators are the slots
  {
    MlLevel &L = ctx->ml[0];
    L.n = n_loc;
    L.npad = pad_chunk(n_loc);
    for (double **v : {&L.r, &L.z, &L.t, &L.cd, &L.cres, &L.ctmp}) RC(ws_alloc_zero(ctx, v, L.npad));
    L.dinv = ctx->dinv_aug;
    L.lmax = ctx->lam_max[OP_AUG];
  }
  HostCsr A1, C1, Ct1;
  {
    PhaseClock pc(ctx, ALFD_SETUP_ML_GALERKIN);
    // A's rows in its LOCAL column space [owned | halo]: the prolongator rows in that order
    DevCsr &dA = ctx->mat[ALFD_A], &dC = ctx->mat[ALFD_C];
    auto perm_rows = [&](const DevCsr &m, HostCsr &Pperm, std::vector<int64_t> &colg) -> int {
      std::vector<int64_t> want(m.halo_globals.begin(), m.halo_globals.end());
      HostCsr halo;
      RC(fetch_rows(ctx, P0, off0.data(), want, halo));
      Pperm = P0;
      Pperm.nrows = P0.nrows + halo.nrows;
      for (int64_t i = 0; i < halo.nrows; ++i) Pperm.rp.push_back(Pperm.rp.back() + (halo.rp[i + 1] - halo.rp[i]));
      Pperm.col.insert(Pperm.col.end(), halo.col.begin(), halo.col.end());
      Pperm.val.insert(Pperm.val.end(), halo.val.begin(), halo.val.end());
      colg.resize(Pperm.nrows);
      for (int64_t q = 0; q < P0.nrows; ++q) colg[q] = off0[rk] + q;
      for (size_t q = 0; q < want.size(); ++q) colg[P0.nrows + q] = want[q];
      return ALFD_OK;
    };
    HostCsr PpA, PpC, AP_loc, AP_halo, CP_loc;
    std::vector<int64_t> colgA, colgC;
    RC(perm_rows(dA, PpA, colgA));
    {   // A_r * P on the device, fetched (its rows are served to the neighbours below)
      DevRawCsr dP, dAP;
      RC(upload_raw(ctx, PpA, dP));
      bool fits = false;
      RC(dev_spgemm(ctx, dA.nrows, dA.rp, dA.col, dA.val, dP.rp, dP.col, dP.val, n1, dAP, &fits));
      dP.release();
      if (!fits) return ctx->err = "partitioned Galerkin product: a row exceeds the device kernel's column set", ALFD_E_UNSUPPORTED;
      RC(download_raw(ctx, dAP, AP_loc));
      dAP.release();
    }
    {
      std::vector<int64_t> want(dA.halo_globals.begin(), dA.halo_globals.end());
      RC(fetch_rows(ctx, AP_loc, off0.data(), want, AP_halo));
    }
    // R_owned: one row per owned coarse unknown, entries over the local fine space [owned | halo] in GLOBAL fine order
    const int64_t c0 = coff[rk], nc_loc = coff[rk + 1] - c0;
    HostCsr Rl, APp;
    {
      std::vector<int64_t> cnt(nc_loc + 1, 0);
      for (int64_t q = 0; q < PpA.nrows; ++q)
        for (int64_t k = PpA.rp[q]; k < PpA.rp[q + 1]; ++k) {
          const int64_t I = PpA.col[k];
          if (I >= c0 && I < c0 + nc_loc) cnt[I - c0 + 1]++;
        }
      for (int64_t I = 0; I < nc_loc; ++I) cnt[I + 1] += cnt[I];
      std::vector<std::pair<int64_t, std::pair<int32_t, double>>> ent(cnt[nc_loc]);   // (global fine id, (local index, value))
      std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
      for (int64_t q = 0; q < PpA.nrows; ++q)
        for (int64_t k = PpA.rp[q]; k < PpA.rp[q + 1]; ++k) {
          const int64_t I = PpA.col[k];
          if (I >= c0 && I < c0 + nc_loc) ent[cur[I - c0]++] = {colgA[q], {(int32_t)q, PpA.val[k]}};
        }
      Rl.nrows = nc_loc;
      Rl.ncols = PpA.nrows;
      Rl.rp = cnt;
      Rl.col.resize(ent.size());
      Rl.val.resize(ent.size());
      for (int64_t I = 0; I < nc_loc; ++I) {
        std::sort(ent.begin() + cnt[I], ent.begin() + cnt[I + 1], [](const auto &a, const auto &b) { return a.first < b.first; });
        for (int64_t k = cnt[I]; k < cnt[I + 1]; ++k) Rl.col[k] = ent[k].second.first, Rl.val[k] = ent[k].second.second;
      }
      APp = AP_loc;
      APp.nrows = AP_loc.nrows + AP_halo.nrows;
      for (int64_t i = 0; i < AP_halo.nrows; ++i) APp.rp.push_back(APp.rp.back() + (AP_halo.rp[i + 1] - AP_halo.rp[i]));
      APp.col.insert(APp.col.end(), AP_halo.col.begin(), AP_halo.col.end());
      APp.val.insert(APp.val.end(), AP_halo.val.begin(), AP_halo.val.end());
    }
    {
      // every prolongator entry must have been seen by the rank that forms its coarse row: the fine rows in the support
      // of an owned coarse unknown have to lie in this rank's rows or in A's halo (coarse partition aligned with the fine one)
      int64_t cnt2[2] = {(int64_t)Rl.col.size(), P0.nnz()};
      std::vector<char> ball;
      std::vector<size_t> sz;
      RC(allgather_bytes(ctx, cnt2, sizeof(cnt2), ball, sz));
      int64_t seen = 0, total = 0;
      for (int p = 0; p < P; ++p) {
        int64_t v[2];
        std::memcpy(v, ball.data() + (size_t)p * sizeof(cnt2), sizeof(cnt2));
        seen += v[0];
        total += v[1];
      }
      if (seen != total)
        return ctx->err = "partitioned CSR prolongator: a coarse unknown's fine support leaves its rank's rows + halo of A "
                          "(alfd_set_aggregate_partition must follow the fine partition)", ALFD_E_INVALID;
    }
    HostCsr A1_loc;
    RC(product(ctx, Rl, APp, A1_loc));
    RC(gather_csr(ctx, A1_loc, nullptr, n1, A1));
    // C_1 = C P: rows of the owned multipliers, then gathered
    RC(perm_rows(dC, PpC, colgC));
    {
      HostCsr Cl;
      RC(download_csr(ctx, dC, Cl));
      Cl.ncols = PpC.nrows;
      RC(product(ctx, Cl, PpC, CP_loc));
    }
    RC(gather_csr(ctx, CP_loc, nullptr, n1, C1));
    transpose_host(C1, Ct1);
    // the partitioned transfer pair of level 0 <-> 1: R rows of the owned coarse unknowns (global fine columns, halo on
    // the fine vector), P rows of the owned fine unknowns addressing the replicated coarse vector
    HostCsr Rg = Rl;
    Rg.ncols = off0.back();
    for (size_t k = 0; k < Rg.col.size(); ++k) Rg.col[k] = (int32_t)colgA[Rl.col[k]];
    MlLevel &N = ctx->ml[1];
    PhaseClock pu(ctx, ALFD_SETUP_ML_UPLOAD);
    RC(upload_level_part(ctx, N.R, Rg, off0.data(), false));
    RC(upload_level_part(ctx, N.P, P0, nullptr, true));
    N.P_global_cols = true;
  }
  // ---- replicated levels 1 .. nlev
  const int64_t lam_global = ctx->part[last].back(), lam_pad = pad_chunk(lam_global);
  RC(ws_alloc_zero(ctx, &ctx->g_w, lam_pad));
  RC(ws_alloc_zero(ctx, &ctx->g_tlam, lam_pad));
  RC(gather_vec(ctx, ctx->diag[ALFD_INVW], ctx->n[last], ctx->g_w));
  ctx->ml_rep_level = 1;
  HostCsr A = std::move(A1), C = std::move(C1), Ct = std::move(Ct1), An, Cn, Ctn, R;
  for (int l = 1; l <= nlev; ++l) {
    MlLevel &L = ctx->ml[l];
    L.gn = A.nrows;
    L.gnpad = pad_chunk(L.gn);
    L.g_offs.assign(P + 1, 0);
    if (l == 1) L.g_offs = coff;
    else for (int p = 0; p <= P; ++p) L.g_offs[p] = L.gn * p / P;      // no partitioned piece below level 1
    L.n = l == 1 ? coff[rk + 1] - coff[rk] : 0;
    L.npad = pad_chunk(std::max<int64_t>(L.n, 1));
    for (double **v : {&L.r, &L.z}) RC(ws_alloc_zero(ctx, v, L.npad));
    {
      PhaseClock pu(ctx, ALFD_SETUP_ML_UPLOAD);
      RC(upload_level_part(ctx, L.gA, A, nullptr, true));
      RC(upload_level_part(ctx, L.gC, C, nullptr, true));
      RC(upload_level_part(ctx, L.gCt, Ct, nullptr, true));
      L.gA.rep = L.gC.rep = L.gCt.rep = true;
    }
    for (double **v : {&L.gdinv, &L.gr, &L.gz, &L.gt, &L.gcd, &L.gcres, &L.gctmp}) RC(ws_alloc_zero(ctx, v, L.gnpad));
    {
      // diagonal and lambda_max on the replicated operator: plain (single-rank) reductions, the same on every rank
      PhaseClock pl(ctx, ALFD_SETUP_ML_LAMBDA);
      ctx->dots_replicated = true;
      double *w_save = ctx->diag[ALFD_INVW];
      ctx->diag[ALFD_INVW] = ctx->g_w;        // diag_plus_m reads the weight through the slot
      const int rc = diag_plus_m(ctx, L.gA, L.gCt, c.aug_assembled ? 0.0 : c.gamma, L.gn, L.gdinv);
      ctx->diag[ALFD_INVW] = w_save;
      if (rc != ALFD_OK) return ctx->dots_replicated = false, rc;
      double *v = L.gt, *wv = L.gr;
      hipLaunchKernelGGL(hash_vector_kernel, dim3((unsigned)((L.gn + 255) / 256)), dim3(256), 0, ctx->stream, L.gn, (int64_t)0, v);
      double lam = 0;
      for (int it = 0; it < c.cheb_power_its; ++it) {
        RC(dot_async(ctx, L.gnpad, v, v, S_TMP));
        RC(read_scalars(ctx, S_TMP, 1));
        VEC_LAUNCH(scale_kernel, L.gnpad, 16, (const double *)nullptr, 0, 0, 1.0 / std::sqrt(ctx->sc_host[S_TMP]), v);
        RC(level_op_rep(ctx, l, v, wv));
        VEC_LAUNCH(pmul_scale_kernel, L.gnpad, 24, 1.0, L.gdinv, wv, wv);
        RC(dot_async(ctx, L.gnpad, wv, wv, S_TMP));
        RC(read_scalars(ctx, S_TMP, 1));
        lam = std::sqrt(ctx->sc_host[S_TMP]);
        std::swap(v, wv);
      }
      ctx->dots_replicated = false;
      L.lmax = lam * c.cheb_safety;
      HIPC(hipMemsetAsync(L.gt, 0, L.gnpad * sizeof(double), ctx->stream));
      HIPC(hipMemsetAsync(L.gr, 0, L.gnpad * sizeof(double), ctx->stream));
    }
    if (l == 1) {
      for (int p = 0; p < P; ++p) L.g_maxpiece = std::max(L.g_maxpiece, coff[p + 1] - coff[p]);
      L.g_maxpiece = std::max<int64_t>(L.g_maxpiece, 1);
      RC(ws_alloc_zero(ctx, &L.g_send, L.g_maxpiece));
      RC(ws_alloc_zero(ctx, &L.g_stage, L.g_maxpiece * P));
    }
    if (l == nlev) break;
    const HostCsr &Pm = ctx->ml_P[l];
    {
      PhaseClock pg(ctx, ALFD_SETUP_ML_GALERKIN);
      HostCsr AP;
      transpose_host(Pm, R);
      RC(product(ctx, A, Pm, AP));
      RC(product(ctx, R, AP, An));
      RC(product(ctx, C, Pm, Cn));
      transpose_host(Cn, Ctn);
    }
    MlLevel &N = ctx->ml[l + 1];
    PhaseClock pu(ctx, ALFD_SETUP_ML_UPLOAD);
    RC(upload_level_part(ctx, N.gP, Pm, nullptr, true));
    RC(upload_level_part(ctx, N.gR, R, nullptr, true));
    N.gP.rep = N.gR.rep = true;
    A = std::move(An);
    C = std::move(Cn);
    Ct = std::move(Ctn);
  }
  if (c.ml_coarse_direct > 0 && ctx->ml[nlev].gn > 0 && ctx->ml[nlev].gn <= c.ml_coarse_direct) {
    PhaseClock pc(ctx, ALFD_SETUP_ML_COARSE);
    std::vector<double> w(lam_global);
    HIPC(hipMemcpyAsync(w.data(), ctx->g_w, w.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    RC(coarse_inverse(ctx, A, C, Ct, w));
    ctx->ml_inv.rep = true;
  }
  if (c.ml_patch_degree > 0) {
    PhaseClock pc(ctx, ALFD_SETUP_ML_PATCH);
    RC(patch_setup_rep(ctx));
  }
  return ALFD_OK;
}

static int ml_setup(alfd_ctx *ctx) {
  const alfd_config &c = ctx->cfg;
  if (c.ml_smooth_degree < 1 || c.ml_coarse_degree < 1 || !(c.ml_smooth_ratio > 1.0) || !(c.ml_coarse_ratio > 1.0))
    return ctx->err = "bad multilevel parameters", ALFD_E_INVALID;
  int nlev = 0;
  bool any_P = false;
  while (nlev < ALFD_MAX_LEVELS && (!ctx->ml_agg[nlev].empty() || !ctx->ml_P[nlev].rp.empty())) {
    any_P = any_P || !ctx->ml_P[nlev].rp.empty();
    ++nlev;
  }
  if (nlev == 0)
    return ctx->err = "alfd_set_aggregates / alfd_set_prolongator must precede alfd_setup for ALFD_PREC_MULTILEVEL", ALFD_E_NOT_SETUP;
  if (ctx->nranks > 1 && any_P) return ml_setup_rep_prolongators(ctx, nlev);
  if (ctx->nranks > 1 && (c.ml_patch_degree > 0 || c.ml_coarse_direct > 0))
    return ctx->err = "partitioned context: the interface patch and the direct coarsest solve need CSR prolongators", ALFD_E_UNSUPPORTED;
  if (c.ml_patch_degree > 0 && !(c.ml_patch_ratio > 1.0)) return ctx->err = "bad ml_patch_ratio", ALFD_E_INVALID;
  const int rk = ctx->rank, last = ctx->nblocks - 1;
  // rank offsets of every level's unknowns: level 0 = block 0, level l+1 = ml_coff[l]
  std::vector<std::vector<int64_t>> off(nlev + 1);
  if (ctx->nranks > 1) {
    off[0] = ctx->part[0];
    for (int l = 0; l < nlev; ++l) {
      if ((int)ctx->ml_coff[l].size() != ctx->nranks + 1 || ctx->ml_coff[l].back() != ctx->ml_ncoarse[l])
        return ctx->err = "alfd_set_aggregate_partition missing or inconsistent for level " + std::to_string(l),
               ALFD_E_INVALID;
      off[l + 1] = ctx->ml_coff[l];
    }
  } else {
    off[0] = {0, ctx->n[0]};
    for (int l = 0; l < nlev; ++l) off[l + 1] = {0, ctx->ml_ncoarse[l]};
  }
  for (int l = 0; l < nlev; ++l) {
    const bool isP = !ctx->ml_P[l].rp.empty();
    if ((isP ? ctx->ml_P[l].nrows : (int64_t)ctx->ml_agg[l].size()) != off[l][rk + 1] - off[l][rk])
      return ctx->err = "aggregates / prolongator of level " + std::to_string(l) + " do not match this rank's unknowns", ALFD_E_INVALID;
  }
  free_levels(ctx);
  ctx->ml.assign(nlev + 1, MlLevel());
  // multi-rank: levels with few unknowns are replicated on every rank after the partitioned build
  // (their kernels take microseconds, their halo exchanges would dominate)
  int rep_from = -1;
  if (ctx->nranks > 1 && ctx->ml_rep_threshold > 0)
    for (int l = 1; l <= nlev; ++l)
      if (off[l].back() <= ctx->ml_rep_threshold) {
        rep_from = l;
        break;
      }
  struct Stash {
    HostCsr A, C, Ct, P, R;
  };
  std::vector<Stash> stash(nlev + 1);
  HostCsr A, C, Ct, An, Cn, Ctn, P, R;
  // Host copies: C and Ct are small and always fetched.  A (21 GB at N = 74) only when a level needs it on the host:
  // the aggregation levels' product, or a CSR-prolongator level whose product does not fit the device kernel.  The
  // device products (ALFD_ML_GPU_GALERKIN=0 switches them off) and the device-side patch-row gather need no host A.
  bool have_host_A = false;
  auto fetch_A = [&](const DevCsr &dA) -> int {
    if (have_host_A) return ALFD_OK;
    PhaseClock pc(ctx, ALFD_SETUP_ML_FETCH);
    RC(download_csr(ctx, dA, A));
    have_host_A = true;
    return ALFD_OK;
  };
  {
    PhaseClock pc(ctx, ALFD_SETUP_ML_FETCH);
    RC(download_csr(ctx, ctx->mat[ALFD_C], C));
    RC(download_csr(ctx, ctx->mat[ALFD_CT], Ct));
  }
  const bool gpu_products = ctx->ml_gpu_galerkin && ctx->nranks == 1;
  if (!gpu_products || ctx->ml_P[0].rp.empty()) RC(fetch_A(ctx->mat[ALFD_A]));
  const int64_t lam0 = ctx->nranks > 1 ? ctx->part[last][rk] : 0;
  const int64_t lam_global = ctx->nranks > 1 ? ctx->part[last].back() : ctx->n[last];
  if (c.ml_patch_degree > 0) {
    PhaseClock pc(ctx, ALFD_SETUP_ML_PATCH);
    RC(patch_setup(ctx, have_host_A ? &A : nullptr, C, Ct));
  }
  for (int l = 0; l <= nlev; ++l) {
    MlLevel &L = ctx->ml[l];
    const int64_t n = off[l][rk + 1] - off[l][rk];
    L.n = n;
    L.npad = pad_chunk(n);
    RC(ws_alloc_zero(ctx, &L.r, L.npad));
    RC(ws_alloc_zero(ctx, &L.z, L.npad));
    RC(ws_alloc_zero(ctx, &L.t, L.npad));
    RC(ws_alloc_zero(ctx, &L.cd, L.npad));
    RC(ws_alloc_zero(ctx, &L.cres, L.npad));
    RC(ws_alloc_zero(ctx, &L.ctmp, L.npad));
    if (l == 0) {
      L.dinv = ctx->dinv_aug;
      L.lmax = ctx->lam_max[OP_AUG];
    } else {
      PhaseClock pc(ctx, ALFD_SETUP_ML_LAMBDA);
      RC(ws_alloc_zero(ctx, &L.dinv, L.npad));
      RC(diag_plus_m(ctx, L.A, L.Ct, c.aug_assembled ? 0.0 : c.gamma, n, L.dinv));
      // lambda_max(D^-1 Aug_l): power iteration from the integer-hash vector (global index)
      double *v = L.t, *wv = L.r;
      if (n > 0)
        hipLaunchKernelGGL(hash_vector_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                           off[l][rk], v);
      double lam = 0;
      for (int it = 0; it < c.cheb_power_its; ++it) {
        RC(dot_async(ctx, L.npad, v, v, S_TMP));
        RC(read_scalars(ctx, S_TMP, 1));
        VEC_LAUNCH(scale_kernel, L.npad, 16, (const double *)nullptr, 0, 0,
                   1.0 / std::sqrt(ctx->sc_host[S_TMP]), v);
        RC(level_op(ctx, l, v, wv));
        VEC_LAUNCH(pmul_scale_kernel, L.npad, 24, 1.0, L.dinv, wv, wv);
        RC(dot_async(ctx, L.npad, wv, wv, S_TMP));
        RC(read_scalars(ctx, S_TMP, 1));
        lam = std::sqrt(ctx->sc_host[S_TMP]);
        std::swap(v, wv);
      }
      L.lmax = lam * c.cheb_safety;
      HIPC(hipMemsetAsync(L.t, 0, L.npad * sizeof(double), ctx->stream));
      HIPC(hipMemsetAsync(L.r, 0, L.npad * sizeof(double), ctx->stream));
    }
    if (l == nlev) break;
    if (!ctx->ml_P[l].rp.empty()) {
      // ---- next level through a general CSR prolongator (single rank): A_c = P^T (A P), C_c = C P, Ct_c = C_c^T
      const HostCsr &Pm = ctx->ml_P[l];
      MlLevel &Nx = ctx->ml[l + 1];
      DevCsr &dA = l == 0 ? ctx->mat[ALFD_A] : L.A;
      DevCsr &dC = l == 0 ? ctx->mat[ALFD_C] : L.C;
      {
        PhaseClock pc(ctx, ALFD_SETUP_ML_UPLOAD);
        transpose_host(Pm, R);
        RC(upload_level_part(ctx, Nx.P, Pm, nullptr, true));
        RC(upload_level_part(ctx, Nx.R, R, nullptr, true));
      }
      bool done = false;
      if (gpu_products && !dA.sparse && !dC.sparse && !Nx.P.sparse && !Nx.R.sparse && dA.nnz > 0) {
        // the three products on the device (spgemm_rows_kernel), results fetched for the format planner
        PhaseClock pc(ctx, ALFD_SETUP_ML_GALERKIN);
        DevRawCsr AP, RAP, CP;
        bool f1 = false, f2 = false, f3 = false;
        RC(dev_spgemm(ctx, dA.nrows, dA.rp, dA.col, dA.val, Nx.P.rp, Nx.P.col, Nx.P.val, Pm.ncols, AP, &f1));
        if (f1) RC(dev_spgemm(ctx, Nx.R.nrows, Nx.R.rp, Nx.R.col, Nx.R.val, AP.rp, AP.col, AP.val, Pm.ncols, RAP, &f2));
        AP.release();
        if (f2) RC(dev_spgemm(ctx, dC.nrows, dC.rp, dC.col, dC.val, Nx.P.rp, Nx.P.col, Nx.P.val, Pm.ncols, CP, &f3));
        if (f3) {
          RC(download_raw(ctx, RAP, An));
          RC(download_raw(ctx, CP, Cn));
          transpose_host(Cn, Ctn);
          done = true;
        }
        RAP.release();
        CP.release();
      }
      if (!done) {
        if (l == 0) RC(fetch_A(dA));
        PhaseClock pc(ctx, ALFD_SETUP_ML_GALERKIN);
        HostCsr AP;
        spgemm_host(A, Pm, AP);
        spgemm_host(R, AP, An);
        AP = HostCsr();
        spgemm_host(C, Pm, Cn);
        transpose_host(Cn, Ctn);
      }
      PhaseClock pc(ctx, ALFD_SETUP_ML_UPLOAD);
      RC(upload_level_part(ctx, Nx.A, An, nullptr, false));
      RC(upload_level_part(ctx, Nx.C, Cn, nullptr, false));
      RC(upload_level_part(ctx, Nx.Ct, Ctn, nullptr, false));
      if (l + 1 < nlev) {
        A = std::move(An);      // host copies of the new level: the next level's fallback product, aggregation levels
        have_host_A = true;
        C = std::move(Cn);
        Ct = std::move(Ctn);
      }
      continue;
    }
    // ---- next level: Galerkin products of this rank's rows
    if (l == 0) RC(fetch_A(ctx->mat[ALFD_A]));
    const std::vector<int32_t> &aggG = ctx->ml_agg[l];  // owned fine dof -> GLOBAL coarse id (or -1)
    const double *w = ctx->ml_wgt[l].empty() ? nullptr : ctx->ml_wgt[l].data();
    if (w && ctx->nranks > 1) return ctx->err = "weighted aggregates are single-rank for now", ALFD_E_UNSUPPORTED;
    const int64_t c0 = off[l + 1][rk], nc_loc = off[l + 1][rk + 1] - c0, nc_glob = off[l + 1].back();
    std::vector<int32_t> agg_rows(n);  // owned fine dof -> LOCAL coarse row
    for (int64_t i = 0; i < n; ++i) {
      if (aggG[i] >= 0 && (aggG[i] < c0 || aggG[i] >= c0 + nc_loc))
        return ctx->err = "an aggregate spans two ranks", ALFD_E_INVALID;
      agg_rows[i] = aggG[i] < 0 ? -1 : (int32_t)(aggG[i] - c0);
    }
    DevCsr &dA = l == 0 ? ctx->mat[ALFD_A] : L.A;
    DevCsr &dC = l == 0 ? ctx->mat[ALFD_C] : L.C;
    DevCsr &dCt = l == 0 ? ctx->mat[ALFD_CT] : L.Ct;
    std::vector<int32_t> aggA, aggC;
    RC(exchange_ids(ctx, dA, aggG, aggA));   // columns of A_l: [owned | halo of A_l]
    RC(exchange_ids(ctx, dC, aggG, aggC));   // columns of C_l: [owned | halo of C_l]
    auto tg0 = std::chrono::steady_clock::now();
    galerkin(A, agg_rows.data(), w, nc_loc, aggA.data(), w, nc_glob, An);
    galerkin(C, nullptr, nullptr, C.nrows, aggC.data(), w, nc_glob, Cn);
    // Ct_{l+1} = P^T Ct_l: rows grouped by aggregate, multiplier columns back to GLOBAL ids
    std::vector<int32_t> lam_ids((size_t)dCt.n_local_cols + dCt.halo_globals.size());
    for (int32_t j = 0; j < dCt.n_local_cols; ++j) lam_ids[j] = (int32_t)(lam0 + j);
    for (size_t j = 0; j < dCt.halo_globals.size(); ++j) lam_ids[dCt.n_local_cols + j] = dCt.halo_globals[j];
    galerkin(Ct, agg_rows.data(), w, nc_loc, lam_ids.data(), nullptr, lam_global, Ctn);
    // transfers are rank-local: P (n x nc_loc, one entry per represented row), R = P^T
    P.nrows = n;
    P.ncols = nc_loc;
    P.rp.assign(n + 1, 0);
    P.col.clear();
    P.val.clear();
    for (int64_t i = 0; i < n; ++i) {
      if (agg_rows[i] >= 0) {
        P.col.push_back(agg_rows[i]);
        P.val.push_back(w ? w[i] : 1.0);
      }
      P.rp[i + 1] = (int64_t)P.col.size();
    }
    transpose_host(P, R);
    ctx->setup_s[ALFD_SETUP_ML_GALERKIN] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
    MlLevel &Nx = ctx->ml[l + 1];
    PhaseClock pc(ctx, ALFD_SETUP_ML_UPLOAD);
    RC(upload_level_part(ctx, Nx.A, An, off[l + 1].data(), false));
    RC(upload_level_part(ctx, Nx.C, Cn, off[l + 1].data(), false));
    RC(upload_level_part(ctx, Nx.Ct, Ctn, ctx->nranks > 1 ? ctx->part[last].data() : nullptr, false));
    RC(upload_level_part(ctx, Nx.P, P, nullptr, true));
    RC(upload_level_part(ctx, Nx.R, R, nullptr, true));
    if (rep_from > 0 && l + 1 >= rep_from) {
      stash[l + 1].A = An;  // global column ids
      stash[l + 1].C = Cn;
      stash[l + 1].Ct = Ctn;
      stash[l + 1].P = P;   // rank-local ids
      stash[l + 1].R = R;
    }
    // host copies of the new level in its LOCAL column space, for the next Galerkin step
    if (l + 1 < nlev) {
      RC(download_csr(ctx, Nx.A, A));
      RC(download_csr(ctx, Nx.C, C));
      RC(download_csr(ctx, Nx.Ct, Ct));
    }
  }
  if (c.ml_coarse_direct > 0 && ctx->ml[nlev].n > 0 && ctx->ml[nlev].n <= c.ml_coarse_direct) {
    // An / Cn / Ctn still hold the coarsest level (global = local column ids on one rank)
    std::vector<double> w(ctx->n[last]);
    HIPC(hipMemcpyAsync(w.data(), ctx->diag[ALFD_INVW], w.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    PhaseClock pc(ctx, ALFD_SETUP_ML_COARSE);
    RC(coarse_inverse(ctx, An, Cn, Ctn, w));
  }
  if (rep_from > 0) {
    // ---- replicate levels rep_from .. nlev: gather operators, diagonals and W on every rank
    const int64_t lam_pad = pad_chunk(lam_global);
    RC(ws_alloc_zero(ctx, &ctx->g_w, lam_pad));
    RC(ws_alloc_zero(ctx, &ctx->g_tlam, lam_pad));
    RC(gather_vec(ctx, ctx->diag[ALFD_INVW], ctx->n[last], ctx->g_w));
    for (int l = rep_from; l <= nlev; ++l) {
      MlLevel &L = ctx->ml[l];
      L.g_offs = off[l];
      L.gn = off[l].back();
      L.gnpad = pad_chunk(L.gn);
      HostCsr g;
      RC(gather_csr(ctx, stash[l].A, nullptr, L.gn, g));
      RC(upload_level_part(ctx, L.gA, g, nullptr, true));
      RC(gather_csr(ctx, stash[l].C, nullptr, L.gn, g));
      RC(upload_level_part(ctx, L.gC, g, nullptr, true));
      RC(gather_csr(ctx, stash[l].Ct, nullptr, lam_global, g));
      RC(upload_level_part(ctx, L.gCt, g, nullptr, true));
      L.gA.rep = L.gC.rep = L.gCt.rep = true;
      if (l > rep_from) {
        RC(gather_csr(ctx, stash[l].P, off[l].data(), L.gn, g));        // n_{l-1} x n_l
        RC(upload_level_part(ctx, L.gP, g, nullptr, true));
        RC(gather_csr(ctx, stash[l].R, off[l - 1].data(), off[l - 1].back(), g));  // n_l x n_{l-1}
        RC(upload_level_part(ctx, L.gR, g, nullptr, true));
        L.gP.rep = L.gR.rep = true;
      }
      for (double **v : {&L.gdinv, &L.gr, &L.gz, &L.gt, &L.gcd, &L.gcres, &L.gctmp}) RC(ws_alloc_zero(ctx, v, L.gnpad));
      RC(gather_vec(ctx, L.dinv, L.n, L.gdinv));
      if (l == rep_from) {
        for (int p = 0; p < ctx->nranks; ++p) L.g_maxpiece = std::max(L.g_maxpiece, off[l][p + 1] - off[l][p]);
        L.g_maxpiece = std::max<int64_t>(L.g_maxpiece, 1);
        RC(ws_alloc_zero(ctx, &L.g_send, L.g_maxpiece));
        RC(ws_alloc_zero(ctx, &L.g_stage, L.g_maxpiece * ctx->nranks));
      }
    }
    ctx->ml_rep_level = rep_from;
  }
  return ALFD_OK;
}

static int setup(alfd_ctx *ctx) {
  if (!ctx->configured) return ctx->err = "alfd_configure not called", ALFD_E_NOT_SETUP;
  for (int ph = ALFD_SETUP_DIAG_LAMBDA; ph < ALFD_SETUP_NPHASES; ++ph) ctx->setup_s[ph] = 0.0;
  ctx->upload_since_setup = false;
  const alfd_config &c = ctx->cfg;
  const bool rat = c.variant == ALFD_RATIONAL;
  if (rat && ctx->nranks > 1) return ctx->err = "rational variant is single-rank", ALFD_E_UNSUPPORTED;
  if (c.outer_solver == ALFD_OUTER_MINRES && c.restart < 4)
    return ctx->err = "MinRes needs restart >= 4 (vector arena)", ALFD_E_INVALID;
  if (c.restart < 1 || c.restart > kMaxBasis - 1) return ctx->err = "restart out of range", ALFD_E_INVALID;
  if (c.fgmres_flavour != ALFD_FGMRES_DEALII_96 && c.fgmres_flavour != ALFD_FGMRES_DEALII_95)
    return ctx->err = "unknown alfd_config::fgmres_flavour", ALFD_E_INVALID;
  if (c.fgmres_flavour == ALFD_FGMRES_DEALII_95 && c.restart < 2)
    return ctx->err = "the deal.II <= 9.5 FGMRES loop needs restart >= 2", ALFD_E_INVALID;
  const bool ell = is_elliptic(c.variant);
  if (!c.grad_div_in_A && (c.variant == ALFD_AL_STOKES || c.variant == ALFD_AL_STOKES_DIAG))
    return ctx->err = "grad_div_in_A = 0 (nested Bt Mp^-1 B in Aug) not implemented", ALFD_E_UNSUPPORTED;
  if ((c.inner_prec == ALFD_PREC_CHEBYSHEV || c.inner_prec == ALFD_PREC_MULTILEVEL) &&
      (c.cheb_degree < 1 || c.cheb_power_its < 1 || !(c.cheb_eig_ratio > 1.0)))
    return ctx->err = "bad Chebyshev parameters", ALFD_E_INVALID;
  // parameter sanity the reference asserts (elliptic_interface.cc:874-884, 912-920)
  if (c.variant == ALFD_AL_ELL_MODIFIED && c.gamma2 > 20.0)
    return ctx->err = "gamma_AL_immersed is too large for modified AL preconditioner", ALFD_E_INVALID;
  if (c.variant == ALFD_AL_ELL_MODIFIED && std::fabs(c.gamma2 - c.gamma) <= 1e-1)
    return ctx->err = "modified AL preconditioner: gamma_1 and gamma_2 should not be too close", ALFD_E_INVALID;
  if (c.variant == ALFD_AL_ELL_IDEAL && std::fabs(c.gamma - c.gamma2) >= 1e-12)
    return ctx->err = "ideal AL preconditioner: gamma must be identical", ALFD_E_INVALID;
  ctx->nblocks = nblocks_of(c.variant);
  const int last = ctx->nblocks - 1;
  if (!ctx->mat[ALFD_A].present || !ctx->mat[ALFD_CT].present || !ctx->mat[ALFD_C].present ||
      (!ctx->diag[ALFD_INVW] && !rat))
    return ctx->err = "A, CT (and C) and INVW must be set", ALFD_E_NOT_SETUP;
  ctx->n[0] = ctx->mat[ALFD_A].nrows;
  ctx->n[last] = ctx->mat[ALFD_C].nrows;
  if (rat) {
    const HostCsr &K = ctx->h_K, &M = ctx->h_M;
    if (K.rp.empty() || M.rp.empty()) return ctx->err = "KIMM and M must be set for the rational variant", ALFD_E_NOT_SETUP;
    if (K.nrows != ctx->n[1] || M.nrows != ctx->n[1] || K.col != M.col || K.rp != M.rp)
      return ctx->err = "KIMM and M must share one sparsity pattern (matrix.add)", ALFD_E_INVALID;
    if (!(c.rho_bound > 0)) return ctx->err = "rho_bound must be positive", ALFD_E_INVALID;
  }
  if (ell) {
    if (!ctx->mat[ALFD_A2].present || !ctx->mat[ALFD_M].present)
      return ctx->err = "A2 and M must be set for the elliptic-interface variants", ALFD_E_NOT_SETUP;
    ctx->n[1] = ctx->mat[ALFD_A2].nrows;
    if (ctx->n[1] != ctx->n[2] || ctx->mat[ALFD_M].nrows != ctx->n[1])
      return ctx->err = "elliptic interface: blocks 1 and 2 must have the same size", ALFD_E_INVALID;
  } else if (ctx->nblocks == 3) {
    if (!ctx->mat[ALFD_BT].present || !ctx->mat[ALFD_B].present || !ctx->mat[ALFD_MP].present ||
        !ctx->diag[ALFD_MP_LUMPED_INV])
      return ctx->err = "BT, B, MP and MP_LUMPED_INV must be set for the Stokes variants", ALFD_E_NOT_SETUP;
    ctx->n[1] = ctx->mat[ALFD_B].nrows;
  }
  if (ctx->mat[ALFD_CT].nrows != ctx->n[0] || (!rat && ctx->diag_n[ALFD_INVW] != ctx->n[last]))
    return ctx->err = "inconsistent block sizes", ALFD_E_INVALID;
  ctx->nmax = 0;
  for (int b = 0; b < ctx->nblocks; ++b) {
    ctx->off[b + 1] = ctx->off[b] + pad_chunk(ctx->n[b]);
    ctx->nmax = std::max(ctx->nmax, pad_chunk(ctx->n[b]));
  }
  // release the workspace of a previous setup()
  HIPC(hipStreamSynchronize(ctx->stream));
  for (void *p : ctx->ws_allocs) hipFree(p);
  ctx->ws_allocs.clear();
  if (ctx->sc_host) hipHostFree(ctx->sc_host), ctx->sc_host = nullptr;
  const int64_t N = ctx->ntot(), n0p = pad_chunk(ctx->n[0]);
  ctx->wmax = c.variant == ALFD_AL_ELL_IDEAL ? ctx->off[2] : ctx->nmax;
  const int64_t wm = ctx->wmax;
  ctx->pstride = N / kChunk + 1;
  RC(ws_alloc_zero(ctx, &ctx->sc, kNumScalars));
  HIPC(hipHostMalloc((void **)&ctx->sc_host, kNumScalars * sizeof(double), hipHostMallocMapped));
  RC(ws_alloc_zero(ctx, &ctx->partial, (int64_t)(kMaxBasis + 2) * ctx->pstride));
  RC(ws_alloc_zero(ctx, &ctx->gather, (int64_t)ctx->nranks * (kMaxBasis + 2)));
  RC(ws_alloc_zero(ctx, &ctx->dinv_aug, n0p));
  RC(ws_alloc_zero(ctx, &ctx->dA, ctx->nmax));
  RC(ws_alloc_zero(ctx, &ctx->s_aug, ctx->nmax));
  ctx->diag_cap = ctx->nmax;
  RC(ws_alloc_zero(ctx, &ctx->w_r, wm));
  RC(ws_alloc_zero(ctx, &ctx->w_z, wm));
  RC(ws_alloc_zero(ctx, &ctx->w_p, wm));
  RC(ws_alloc_zero(ctx, &ctx->w_Ap, wm));
  RC(ws_alloc_zero(ctx, &ctx->c_d, wm));
  RC(ws_alloc_zero(ctx, &ctx->c_res, wm));
  RC(ws_alloc_zero(ctx, &ctx->c_tmp, wm));
  RC(ws_alloc_zero(ctx, &ctx->t_lam, pad_chunk(ctx->n[last])));
  RC(ws_alloc_zero(ctx, &ctx->q_tmp, ctx->nmax));
  RC(ws_alloc_zero(ctx, &ctx->rhs_tmp, std::max(n0p, wm)));
  RC(ws_alloc_zero(ctx, &ctx->V, (int64_t)(c.restart + 1) * N));
  RC(ws_alloc_zero(ctx, &ctx->Z, (int64_t)c.restart * N));
  RC(ws_alloc_zero(ctx, &ctx->xb, N));
  RC(ws_alloc_zero(ctx, &ctx->bb, N));
  RC(ws_alloc_zero(ctx, &ctx->io, N));
  RC(ws_alloc_zero(ctx, &ctx->st_in, N));
  RC(ws_alloc_zero(ctx, &ctx->st_out, N));
  for (int k = 0; k < 7; ++k) ctx->lam_max[k] = 0;
  if (ctx->n_sc_host) hipHostFree(ctx->n_sc_host), ctx->n_sc_host = nullptr;
  if (c.w_inverse != ALFD_W_DIAGONAL) {
    // exact W^-1 = (M^-1)^2 or M^-1: nested Jacobi CG on the immersed mass matrix
    if (c.w_inverse != ALFD_W_MASS_INV_SQUARED && c.w_inverse != ALFD_W_MASS_INV)
      return ctx->err = "unknown alfd_config::w_inverse", ALFD_E_INVALID;
    if (rat) return ctx->err = "the rational variant has no W^-1", ALFD_E_UNSUPPORTED;
    const HostCsr &M = ctx->h_M;
    if (!ctx->mat[ALFD_M].present || M.rp.empty() || M.nrows != ctx->n[last])
      return ctx->err = "slot M (immersed mass matrix) must be set for the exact W^-1", ALFD_E_NOT_SETUP;
    const int64_t nlp = pad_chunk(ctx->n[last]);
    const int64_t row0 = ctx->nranks > 1 ? ctx->part[last][ctx->rank] : 0;
    std::vector<double> dm(nlp, 0.0);
    for (int64_t i = 0; i < M.nrows; ++i) {
      double dii = 0.0;
      for (int64_t k = M.rp[i]; k < M.rp[i + 1]; ++k)
        if (M.col[k] == row0 + i) dii = M.val[k];
      if (!(dii > 0.0)) return ctx->err = "mass matrix needs a positive diagonal", ALFD_E_INVALID;
      dm[i] = 1.0 / dii;
    }
    RC(ws_alloc_zero(ctx, &ctx->dinv_m, nlp));
    HIPC(hipMemcpyAsync(ctx->dinv_m, dm.data(), nlp * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    RC(ws_alloc_zero(ctx, &ctx->m_tmp, nlp));
    RC(ws_alloc_zero(ctx, &ctx->m_tmp2, nlp));
    RC(ws_alloc_zero(ctx, &ctx->n_r, nlp));
    RC(ws_alloc_zero(ctx, &ctx->n_z, nlp));
    RC(ws_alloc_zero(ctx, &ctx->n_p, nlp));
    RC(ws_alloc_zero(ctx, &ctx->n_Ap, nlp));
    RC(ws_alloc_zero(ctx, &ctx->n_sc, kNumScalars));
    HIPC(hipHostMalloc((void **)&ctx->n_sc_host, kNumScalars * sizeof(double), hipHostMallocMapped));
    RC(ws_alloc_zero(ctx, &ctx->n_partial, (int64_t)(kMaxBasis + 2) * ctx->pstride));
    RC(ws_alloc_zero(ctx, &ctx->n_gather, (int64_t)ctx->nranks * (kMaxBasis + 2)));
  }
  const bool cheb = c.inner_prec == ALFD_PREC_CHEBYSHEV || c.inner_prec == ALFD_PREC_MULTILEVEL;
  if (rat) {
    // K_inv: Jacobi / Chebyshev-Jacobi CG on K itself
    RC(ws_alloc_zero(ctx, &ctx->dinv_k, n0p));
    {
      const DevCsr &A = ctx->mat[ALFD_A];
      const int grid = grid_for_rows(A.nrows, A.L);
#define ALFD_DIAG(LL)                                                                                   \
  hipLaunchKernelGGL((extract_diag_kernel<LL>), dim3(grid), dim3(kBlock), 0, ctx->stream, A.nrows, A.rp, \
                     A.col, A.val, ctx->dA)
      switch (A.L) {
        case 4: ALFD_DIAG(4); break;
        case 8: ALFD_DIAG(8); break;
        case 16: ALFD_DIAG(16); break;
        case 32: ALFD_DIAG(32); break;
        default: ALFD_DIAG(64); break;
      }
#undef ALFD_DIAG
      hipLaunchKernelGGL(inv_diag_kernel, dim3((unsigned)std::max<int64_t>(1, (ctx->n[0] + 255) / 256)), dim3(256), 0, ctx->stream,
                         ctx->n[0], ctx->dA, ctx->dinv_k);
    }
    if (cheb) RC(power_iteration(ctx, OP_K));
    // block-diagonal matrix of the 21 immersed systems, segment s at rows/cols s*npl
    const HostCsr &K = ctx->h_K, &M = ctx->h_M;
    const int64_t n1 = ctx->n[1], npl = pad_chunk(n1), nnz1 = K.rp[n1];
    std::vector<int64_t> rp((size_t)npl * kRatSystems + 1, 0);
    std::vector<int32_t> col((size_t)nnz1 * kRatSystems);
    std::vector<double> val((size_t)nnz1 * kRatSystems);
    for (int sidx = 0; sidx < kRatSystems; ++sidx) {
      const double sh = sidx < 20 ? -(c.rho_bound * kRatPoles[sidx]) : 0.0;  // matrix.add(-rho p_i, M)
      for (int64_t r = 0; r < npl; ++r) {
        const int64_t gr = (int64_t)sidx * npl + r;
        rp[gr + 1] = rp[gr] + (r < n1 ? K.rp[r + 1] - K.rp[r] : 0);
      }
      for (int64_t k = 0; k < nnz1; ++k) {
        col[(size_t)sidx * nnz1 + k] = (int32_t)(K.col[k] + sidx * npl);
        val[(size_t)sidx * nnz1 + k] = sidx < 20 ? std::fma(sh, M.val[k], K.val[k]) : M.val[k];
      }
    }
    {
      // upload through the generic path into a scratch slot object
      RC(upload_matrix(ctx, kScratchSlot, npl * kRatSystems, npl * kRatSystems, rp.data(), col.data(),
                       val.data()));
      csr_free(ctx->rat_mat);
      ctx->rat_mat = std::move(ctx->mat[kScratchSlot]);
      ctx->mat[kScratchSlot] = DevCsr();
    }
    const int64_t nt = npl * kRatSystems;
    RC(ws_alloc_zero(ctx, &ctx->rt_r, nt));
    RC(ws_alloc_zero(ctx, &ctx->rt_z, nt));
    RC(ws_alloc_zero(ctx, &ctx->rt_p, nt));
    RC(ws_alloc_zero(ctx, &ctx->rt_Ap, nt));
    RC(ws_alloc_zero(ctx, &ctx->rt_x, nt));
    RC(ws_alloc_zero(ctx, &ctx->rt_dinv, nt));
    RC(ws_alloc_zero(ctx, &ctx->rt_partial, nt / kChunk + 1));
    RC(ws_alloc_zero(ctx, &ctx->rt_scb, kRatSystems * kBS));
    RC(ws_alloc_zero(ctx, &ctx->rt_coef, kRatSystems));
    if (ctx->rt_scb_host) hipHostFree(ctx->rt_scb_host), ctx->rt_scb_host = nullptr;
    HIPC(hipHostMalloc((void **)&ctx->rt_scb_host, kRatSystems * kBS * sizeof(double)));
    {
      // 1/diag of the shifted systems (same fma as the values above), 1 for the
      // unpreconditioned mass solve, 0 in the padding
      std::vector<double> dinv((size_t)nt, 0.0);
      for (int sidx = 0; sidx < kRatSystems; ++sidx) {
        const double sh = sidx < 20 ? -(c.rho_bound * kRatPoles[sidx]) : 0.0;
        for (int64_t r = 0; r < n1; ++r) {
          double d = 1.0;
          if (sidx < 20)
            for (int64_t k = K.rp[r]; k < K.rp[r + 1]; ++k)
              if (K.col[k] == r) d = 1.0 / std::fma(sh, M.val[k], K.val[k]);
          dinv[(size_t)sidx * npl + r] = d;
        }
      }
      HIPC(hipMemcpyAsync(ctx->rt_dinv, dinv.data(), nt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      HIPC(hipStreamSynchronize(ctx->stream));
    }
    double coef[kRatSystems];
    for (int sidx = 0; sidx < 20; ++sidx) coef[sidx] = c.rho_bound * kRatRes[sidx + 1];
    coef[20] = kRatRes[0];
    HIPC(hipMemcpyAsync(ctx->rt_coef, coef, sizeof(coef), hipMemcpyHostToDevice, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
    ctx->lambda_max = ctx->lam_max[OP_K];
    ctx->is_setup = true;
    return ALFD_OK;
  }
  // diag(Aug) = diag(A) + gamma sum_k w_k Ct_ik^2 (no product matrix is formed)
  std::unique_ptr<PhaseClock> diag_clock(new PhaseClock(ctx, ALFD_SETUP_DIAG_LAMBDA));
  if (c.aug_assembled && is_elliptic(c.variant))
    return ctx->err = "aug_assembled (operator form) is implemented for the AL2 / Stokes variants", ALFD_E_UNSUPPORTED;
  RC(diag_plus(ctx, ALFD_A, ALFD_CT, c.aug_assembled ? 0.0 : c.gamma, ctx->n[0], ctx->dinv_aug));
  if (ell) {
    // diag(A22_aug) = diag(A2) + gamma2 sum_k w_k M_ik^2
    RC(ws_alloc_zero(ctx, &ctx->dinv_a22, pad_chunk(ctx->n[1])));
    RC(diag_plus(ctx, ALFD_A2, ALFD_M, c.gamma2, ctx->n[1], ctx->dinv_a22));
    if (c.variant == ALFD_AL_ELL_IDEAL) {
      RC(ws_alloc_zero(ctx, &ctx->dinv_aug2, ctx->off[2]));
      HIPC(hipMemcpyAsync(ctx->dinv_aug2, ctx->dinv_aug, n0p * sizeof(double), hipMemcpyDeviceToDevice,
                          ctx->stream));
      HIPC(hipMemcpyAsync(ctx->dinv_aug2 + ctx->off[1], ctx->dinv_a22, pad_chunk(ctx->n[1]) * sizeof(double),
                          hipMemcpyDeviceToDevice, ctx->stream));
      if (cheb) RC(power_iteration(ctx, OP_AUG2));
    } else if (cheb) {
      RC(power_iteration(ctx, OP_AUG));
      RC(power_iteration(ctx, OP_A22));
    }
  } else if (cheb) {
    RC(power_iteration(ctx, OP_AUG));
  }
  ctx->lambda_max = ctx->lam_max[c.variant == ALFD_AL_ELL_IDEAL ? OP_AUG2 : OP_AUG];
  diag_clock.reset();
  free_levels(ctx);
  if (c.inner_prec == ALFD_PREC_MULTILEVEL) {
    if (c.variant == ALFD_AL_ELL_IDEAL)
      return ctx->err = "multilevel inner preconditioner: not for the 2x2 block CG of the ideal variant", ALFD_E_UNSUPPORTED;
    RC(ml_setup(ctx));
  }
  HIPC(hipStreamSynchronize(ctx->stream));
  ctx->is_setup = true;
  return ALFD_OK;
}

// host blocks <-> padded device block vector
static int to_device(alfd_ctx *ctx, const double *const *blocks, double *dev) {
  HIPC(hipMemsetAsync(dev, 0, ctx->ntot() * sizeof(double), ctx->stream));
  for (int b = 0; b < ctx->nblocks; ++b)
    HIPC(hipMemcpyAsync(dev + ctx->off[b], blocks[b], ctx->n[b] * sizeof(double), hipMemcpyHostToDevice,
                        ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return ALFD_OK;
}
static int to_host(alfd_ctx *ctx, const double *dev, double *const *blocks) {
  for (int b = 0; b < ctx->nblocks; ++b)
    HIPC(hipMemcpyAsync(blocks[b], dev + ctx->off[b], ctx->n[b] * sizeof(double), hipMemcpyDeviceToHost,
                        ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return ALFD_OK;
}

static void reset_stats(alfd_ctx *ctx) {
  ctx->inner_its = ctx->mp_its = 0;
  ctx->inner_failures = ctx->precond_applications = 0;
  ctx->rational_its = 0;
  ctx->mass_its = 0;
}
static void fill_result(alfd_ctx *ctx, alfd_result *res, int status) {
  res->status = status;
  res->inner_iterations = ctx->inner_its;
  res->mp_iterations = ctx->mp_its;
  res->inner_failures = ctx->inner_failures;
  res->precond_applications = ctx->precond_applications;
  res->lambda_max = ctx->lambda_max;
  res->rational_iterations = ctx->rational_its;
  res->mass_iterations = ctx->mass_its;
}

}  // namespace alfd

// =================================================================== C ABI
#define CHECK_CTX()                       \
  if (!ctx) return ALFD_E_INVALID;        \
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->err = "hipSetDevice failed", ALFD_E_HIP
#define CHECK_SETUP() \
  if (!ctx->is_setup) return ctx->err = "alfd_setup not called", ALFD_E_NOT_SETUP

extern "C" {

int alfd_abi_version(void) { return ALFD_ABI_VERSION; }

const char *alfd_strerror(int s) {
  switch (s) {
    case ALFD_OK: return "ok";
    case ALFD_E_INVALID: return "invalid argument";
    case ALFD_E_HIP: return "HIP runtime error";
    case ALFD_E_NO_CONVERGENCE_OUTER: return "FGMRES: no convergence (SolverControl::NoConvergence)";
    case ALFD_E_NO_CONVERGENCE_INNER: return "inner CG: no convergence (SolverControl::NoConvergence)";
    case ALFD_E_BREAKDOWN: return "breakdown (NaN residual)";
    case ALFD_E_NOT_SETUP: return "context not set up";
    case ALFD_E_COMM: return "RCCL error";
    case ALFD_E_UNSUPPORTED: return "not implemented";
    default: return "unknown status";
  }
}

const char *alfd_last_error(alfd_ctx_t ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int alfd_create(alfd_ctx_t *out, int device_id) {
  if (!out) return ALFD_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
    return ALFD_E_HIP;  // no GPU: fail loudly, there is no CPU path
  if (hipSetDevice(device_id) != hipSuccess) return ALFD_E_HIP;
  alfd_ctx *ctx = new alfd_ctx;
  ctx->device = device_id;
  if (hipStreamCreateWithFlags(&ctx->xstream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming) != hipSuccess)
    ctx->xstream = nullptr;   // no overlap then
  if (const char *e = std::getenv("ALFD_SPMV_OVERLAP_HALO")) ctx->overlap_halo = std::atoi(e);
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return ALFD_E_HIP;
  }
  alfd_default_config(&ctx->cfg, ALFD_AL_STOKES);
  if (const char *e = std::getenv("ALFD_SPMV_STREAM_R")) ctx->spmv_stream_R = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_STREAM_U")) ctx->spmv_stream_U = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_NT")) ctx->spmv_nt = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_GRID")) ctx->spmv_grid_mult = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW")) ctx->win_enable = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_RB")) ctx->win_RB = std::max(4, std::atoi(e));
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_MIN_BLOCKS")) ctx->win_min_blocks = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_SHORT_MIN_BLOCKS")) ctx->win_short_min_blocks = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_MAXW")) ctx->win_maxW = std::min(16384, std::max(256, std::atoi(e)));
  if (const char *e = std::getenv("ALFD_ML_REPLICATE")) ctx->ml_rep_threshold = std::atoll(e);
  if (const char *e = std::getenv("ALFD_ML_GPU_GALERKIN")) ctx->ml_gpu_galerkin = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_VI_LEVELS")) ctx->vi_levels = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_BATCH_MAJOR")) ctx->vs_enable = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_BATCH_MAJOR_ROWS")) ctx->vs_RB = std::max(4, std::min(kVsMaxRows, std::atoi(e)));
  if (const char *e = std::getenv("ALFD_SPMV_VI_XCD")) ctx->vi_xcd = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_VI_BATCHED")) ctx->vi_batched = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_RB_VI")) ctx->win_RB_vi = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_VI_R")) ctx->vi_rows_R = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_VI_J")) ctx->vi_rows_J = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_VALUE_INDEX")) ctx->win_vi = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_GROUP_R")) ctx->spmv_group_R = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_GROUP_U")) ctx->spmv_group_U = std::atoi(e);
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_SHORT")) ctx->win_short_scale = std::max(0, std::atoi(e));
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_GAP")) ctx->win_gap = std::max(1, std::atoi(e));
  if (const char *e = std::getenv("ALFD_SPMV_WINDOW_XCD")) ctx->win_xcd = std::atoi(e);
  *out = ctx;
  return ALFD_OK;
}

int alfd_destroy(alfd_ctx_t ctx) {
  if (!ctx) return ALFD_E_INVALID;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  flush_timers(ctx);
  free_levels(ctx);
  csr_free(ctx->rat_mat);
  for (DevCsr &m : ctx->mat) csr_free(m);
  for (void *p : ctx->ws_allocs) hipFree(p);
  for (void *p : ctx->allocs) hipFree(p);
  if (ctx->sc_host) hipHostFree(ctx->sc_host);
  if (ctx->n_sc_host) hipHostFree(ctx->n_sc_host);
  if (ctx->rt_scb_host) hipHostFree(ctx->rt_scb_host);
  if (ctx->nccl) ncclCommDestroy(ctx->nccl);
  if (ctx->xstream) hipStreamDestroy(ctx->xstream);
  if (ctx->ev_x) hipEventDestroy(ctx->ev_x);
  if (ctx->ev_halo) hipEventDestroy(ctx->ev_halo);
  hipStreamDestroy(ctx->stream);
  delete ctx;
  return ALFD_OK;
}

int alfd_comm_unique_id(void *id_out, size_t bytes) {
  static_assert(sizeof(ncclUniqueId) <= ALFD_UNIQUE_ID_BYTES, "unique id size");
  if (!id_out || bytes < sizeof(ncclUniqueId)) return ALFD_E_INVALID;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return ALFD_E_COMM;
  std::memset(id_out, 0, bytes);
  std::memcpy(id_out, &id, sizeof(id));
  return ALFD_OK;
}

int alfd_comm_init(alfd_ctx_t ctx, int rank, int nranks, const void *id, size_t bytes) {
  CHECK_CTX();
  if (nranks < 1 || rank < 0 || rank >= nranks) return ALFD_E_INVALID;
  ctx->rank = rank;
  ctx->nranks = nranks;
  if (nranks == 1) return ALFD_OK;
  if (!id || bytes < sizeof(ncclUniqueId)) return ALFD_E_INVALID;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  if (ncclCommInitRank(&ctx->nccl, nranks, uid, rank) != ncclSuccess)
    return ctx->err = "ncclCommInitRank failed", ALFD_E_COMM;
  if (rank == 0) {   // which RCCL this process resolved (the library links $(ROCM)/lib/librccl, torch bundles its own copy)
    int v = 0;
    if (ncclGetVersion(&v) == ncclSuccess) std::fprintf(stderr, "[alfd] RCCL version %d, %d ranks\n", v, nranks);
  }
  return ALFD_OK;
}

int alfd_local_group_create(int nranks, alfd_local_group **out) {
  if (!out || nranks < 1) return ALFD_E_INVALID;
  alfd_local_group *g = new alfd_local_group;
  g->n = nranks;
  g->buf.assign(nranks, nullptr);
  g->off.assign(nranks, nullptr);
  *out = g;
  return ALFD_OK;
}

int alfd_local_group_destroy(alfd_local_group *g) {
  if (!g) return ALFD_E_INVALID;
  delete g;
  return ALFD_OK;
}

int alfd_comm_init_local(alfd_ctx_t ctx, alfd_local_group *g, int rank) {
  CHECK_CTX();
  if (!g || rank < 0 || rank >= g->n) return ALFD_E_INVALID;
  ctx->rank = rank;
  ctx->nranks = g->n;
  ctx->local = g->n > 1 ? g : nullptr;
  return ALFD_OK;
}

int alfd_comm_init_host(alfd_ctx_t ctx, int rank, int nranks, alfd_host_allgather_fn allgather,
                        alfd_host_alltoallv_fn alltoallv, void *user) {
  CHECK_CTX();
  if (nranks < 1 || rank < 0 || rank >= nranks || !allgather || !alltoallv) return ALFD_E_INVALID;
  ctx->rank = rank;
  ctx->nranks = nranks;
  if (nranks == 1) return ALFD_OK;
  ctx->host_allgather = allgather;
  ctx->host_alltoallv = alltoallv;
  ctx->host_user = user;
  return ALFD_OK;
}

int alfd_set_partition(alfd_ctx_t ctx, int nblocks, const int64_t *const *offsets) {
  CHECK_CTX();
  if (nblocks < 2 || nblocks > ALFD_MAX_BLOCKS || !offsets) return ALFD_E_INVALID;
  ctx->part.assign(nblocks, {});
  for (int b = 0; b < nblocks; ++b) {
    ctx->part[b].assign(offsets[b], offsets[b] + ctx->nranks + 1);
    for (int r = 0; r < ctx->nranks; ++r)
      if (ctx->part[b][r + 1] < ctx->part[b][r]) return ctx->err = "partition not monotone", ALFD_E_INVALID;
  }
  ctx->nblocks = nblocks;
  return ALFD_OK;
}

int alfd_set_matrix(alfd_ctx_t ctx, int slot, int64_t nrows, int64_t ncols, const int64_t *row_ptr,
                    const int32_t *col, const double *val) {
  CHECK_CTX();
  if (slot < 0 || slot >= ALFD_NSLOTS || nrows < 0 || ncols < 0 || !row_ptr) return ALFD_E_INVALID;
  if (ncols > 2147483647LL) return ctx->err = "more than 2^31-1 columns", ALFD_E_INVALID;
  if (row_ptr[0] != 0) return ctx->err = "row_ptr[0] != 0", ALFD_E_INVALID;
  const int64_t nnz = row_ptr[nrows];
  if (nnz > 0 && (!col || !val)) return ALFD_E_INVALID;
  for (int64_t r = 0; r < nrows; ++r)
    if (row_ptr[r + 1] < row_ptr[r]) return ctx->err = "row_ptr not monotone", ALFD_E_INVALID;
  for (int64_t k = 0; k < nnz; ++k)
    if (col[k] < 0 || col[k] >= ncols) return ctx->err = "column index out of range", ALFD_E_INVALID;
  ctx->is_setup = false;
  if (ctx->nranks > 1 && ctx->nblocks == 0)
    return ctx->err = "alfd_set_partition must precede alfd_set_matrix", ALFD_E_INVALID;
  if (!ctx->upload_since_setup) ctx->setup_s[ALFD_SETUP_UPLOAD] = 0.0, ctx->upload_since_setup = true;
  struct UploadClock {
    alfd_ctx *c;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~UploadClock() {
      hipStreamSynchronize(c->stream);
      c->setup_s[ALFD_SETUP_UPLOAD] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  } upload_clock{ctx};
  RC(upload_matrix(ctx, slot, nrows, ncols, row_ptr, col, val));
  if ((slot == ALFD_M || slot == ALFD_KIMM) && nnz <= (int64_t)1 << 26) {
    // the rational preconditioner builds its 21 shifted systems from these on the host
    HostCsr &h = slot == ALFD_M ? ctx->h_M : ctx->h_K;
    h.nrows = nrows;
    h.ncols = ncols;
    h.rp.assign(row_ptr, row_ptr + nrows + 1);
    h.col.assign(col, col + nnz);
    h.val.assign(val, val + nnz);
  }
  // single rank: the transposed operators are derived on upload, like
  // transpose_operator(Ct) (stokes...:927), and re-derived whenever CT / BT is uploaded
  // again (new values must not meet the transpose of the old ones); an explicit upload of
  // C / B overrides and is then left alone.
  if (slot == ALFD_C || slot == ALFD_B) ctx->mat[slot].derived = false;
  if (ctx->nranks == 1) {
    const int pairs[2][2] = {{ALFD_CT, ALFD_C}, {ALFD_BT, ALFD_B}};
    for (const auto &pr : pairs) {
      const DevCsr &t = ctx->mat[pr[1]];
      if (slot == pr[0] && (!t.present || t.derived)) {
        RC(upload_transpose(ctx, pr[1], nrows, ncols, row_ptr, col, val));
        ctx->mat[pr[1]].derived = true;
      }
    }
  }
  return ALFD_OK;
}

int alfd_set_diag(alfd_ctx_t ctx, int slot, int64_t n, const double *d) {
  CHECK_CTX();
  if (slot < 0 || slot >= ALFD_NDIAGS || n < 0 || (n > 0 && !d)) return ALFD_E_INVALID;
  ctx->is_setup = false;
  if (ctx->diag[slot]) {
    ctx->allocs.erase(std::remove(ctx->allocs.begin(), ctx->allocs.end(), (void *)ctx->diag[slot]), ctx->allocs.end());
    hipFree(ctx->diag[slot]);
    ctx->diag[slot] = nullptr;
  }
  RC(dev_alloc_zero(ctx, &ctx->diag[slot], pad_chunk(n)));
  HIPC(hipMemcpyAsync(ctx->diag[slot], d, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  ctx->diag_n[slot] = n;
  return ALFD_OK;
}

void alfd_default_config(alfd_config *c, int variant) {
  std::memset(c, 0, sizeof(*c));
  c->variant = variant;
  c->restart = (variant == ALFD_AL_ELL_IDEAL || variant == ALFD_AL_ELL_MODIFIED) ? 50 : 30;
  c->orthogonalization = ALFD_ORTH_CGS2;
  c->grad_div_in_A = 1;
  c->gamma = 10.0;
  c->gamma_grad_div = 10.0;
  c->gamma2 = 1e-2;
  c->outer = {ALFD_CTRL_REDUCTION, 1000, 1e-8, 1e-12};
  c->inner = {ALFD_CTRL_ABS, 100, 1e-2, 0.0};
  c->mp_inner = {ALFD_CTRL_ABS, 100, 1e-6, 0.0};
  c->inner_prec = ALFD_PREC_CHEBYSHEV;
  c->cheb_degree = 4;
  c->cheb_power_its = 20;
  c->on_inner_failure = ALFD_INNER_THROW;
  c->cheb_eig_ratio = 30.0;
  c->cheb_safety = 1.2;
  c->log_level = 0;
  c->outer_solver = variant == ALFD_RATIONAL ? ALFD_OUTER_MINRES : ALFD_OUTER_FGMRES;
  c->rho_bound = 0.0;
  c->rational = {ALFD_CTRL_ABS, 2000, 1e-14, 0.0};
  c->ml_smooth_degree = 3;
  c->ml_coarse_degree = 40;
  c->ml_smooth_ratio = 4.0;
  c->ml_coarse_ratio = 400.0;
  c->w_inverse = ALFD_W_DIAGONAL;
  c->mass = {ALFD_CTRL_REDUCTION, 1000, 1e-30, 1e-14};
  c->ml_patch_degree = 0;
  c->ml_coarse_direct = 0;
  c->ml_patch_ratio = 30.0;
}

int alfd_set_aggregates(alfd_ctx_t ctx, int level, int64_t n_fine, const int32_t *agg, const double *weight,
                        int64_t n_coarse) {
  CHECK_CTX();
  if (level < 0 || level >= ALFD_MAX_LEVELS || n_fine < 1 || n_coarse < 1 || !agg) return ALFD_E_INVALID;
  for (int64_t i = 0; i < n_fine; ++i)
    if (agg[i] < -1 || agg[i] >= n_coarse) return ctx->err = "aggregate id out of range", ALFD_E_INVALID;
  ctx->ml_agg[level].assign(agg, agg + n_fine);
  if (weight)
    ctx->ml_wgt[level].assign(weight, weight + n_fine);
  else
    ctx->ml_wgt[level].clear();
  ctx->ml_ncoarse[level] = n_coarse;
  ctx->ml_P[level] = HostCsr();
  for (int l = level + 1; l < ALFD_MAX_LEVELS; ++l) ctx->ml_agg[l].clear(), ctx->ml_wgt[l].clear(), ctx->ml_P[l] = HostCsr();
  ctx->is_setup = false;
  return ALFD_OK;
}

int alfd_set_prolongator(alfd_ctx_t ctx, int level, int64_t n_fine, int64_t n_coarse, const int64_t *row_ptr,
                         const int32_t *col, const double *val) {
  CHECK_CTX();
  if (level < 0 || level >= ALFD_MAX_LEVELS || n_fine < 0 || n_coarse <= 0 || n_coarse > INT32_MAX || !row_ptr ||
      row_ptr[0] != 0)
    return ctx->err = "alfd_set_prolongator: bad arguments", ALFD_E_INVALID;
  for (int64_t i = 0; i < n_fine; ++i)
    if (row_ptr[i + 1] < row_ptr[i]) return ctx->err = "alfd_set_prolongator: row_ptr must be monotone", ALFD_E_INVALID;
  const int64_t nnz = row_ptr[n_fine];
  if (nnz > 0 && (!col || !val)) return ctx->err = "alfd_set_prolongator: col / val missing", ALFD_E_INVALID;
  for (int64_t i = 0; i < n_fine; ++i)
    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k)
      if (col[k] < 0 || col[k] >= n_coarse || (k > row_ptr[i] && col[k] <= col[k - 1]))
        return ctx->err = "alfd_set_prolongator: columns must be ascending and inside [0, n_coarse)", ALFD_E_INVALID;
  HostCsr &P = ctx->ml_P[level];
  P.nrows = n_fine;
  P.ncols = n_coarse;
  P.rp.assign(row_ptr, row_ptr + n_fine + 1);
  P.col.assign(col, col + nnz);
  P.val.assign(val, val + nnz);
  ctx->ml_agg[level].clear();
  ctx->ml_wgt[level].clear();
  ctx->ml_ncoarse[level] = n_coarse;
  for (int l = level + 1; l < ALFD_MAX_LEVELS; ++l) {   // levels below are redefined by later calls
    ctx->ml_agg[l].clear();
    ctx->ml_P[l] = HostCsr();
  }
  ctx->is_setup = false;
  return ALFD_OK;
}

int alfd_build_aggregates(alfd_ctx_t ctx, int32_t block_size, double threshold, int32_t max_aggregate_nodes,
                          int64_t min_coarse, int32_t max_levels, int32_t *levels_out) {
  CHECK_CTX();
  if (ctx->nranks > 1) return ctx->err = "alfd_build_aggregates is single-rank", ALFD_E_UNSUPPORTED;
  if (!ctx->mat[ALFD_A].present) return ctx->err = "upload slot A first", ALFD_E_NOT_SETUP;
  if (block_size < 1 || !(threshold >= 0.0) || max_aggregate_nodes < 2 || min_coarse < 1) return ALFD_E_INVALID;
  if (ctx->mat[ALFD_A].nrows % block_size) return ctx->err = "rows of A are not a multiple of block_size", ALFD_E_INVALID;
  if (max_levels < 1 || max_levels > ALFD_MAX_LEVELS - 1) max_levels = ALFD_MAX_LEVELS - 1;
  HostCsr A, An;
  RC(download_csr(ctx, ctx->mat[ALFD_A], A));
  for (int l = 0; l < ALFD_MAX_LEVELS; ++l) ctx->ml_agg[l].clear(), ctx->ml_wgt[l].clear(), ctx->ml_P[l] = HostCsr();
  int nlev = 0;
  while (nlev < max_levels) {
    std::vector<int32_t> agg;
    int64_t nc = 0;
    aggregate_level(A, block_size, threshold, max_aggregate_nodes, agg, nc);
    if (nc < 1 || nc >= A.nrows) break;     // nothing left to coarsen
    ctx->ml_agg[nlev] = agg;
    ctx->ml_ncoarse[nlev] = nc;
    ++nlev;
    if (nc <= min_coarse) break;
    galerkin(A, agg.data(), nullptr, nc, agg.data(), nullptr, nc, An);
    std::swap(A, An);
  }
  if (nlev == 0) return ctx->err = "algebraic aggregation found nothing to coarsen", ALFD_E_INVALID;
  if (levels_out) *levels_out = nlev;
  ctx->is_setup = false;
  return ALFD_OK;
}

int alfd_host_aggregate_level(int64_t nrows, const int64_t *rp, const int32_t *col, const double *val,
                               int32_t block_size, double threshold, int32_t max_aggregate_nodes, int32_t *agg,
                               int64_t *n_coarse) {
  if (nrows < 1 || !rp || !col || !val || !agg || !n_coarse || block_size < 1 || nrows % block_size ||
      max_aggregate_nodes < 2 || !(threshold >= 0.0))
    return ALFD_E_INVALID;
  HostCsr A;
  A.nrows = A.ncols = nrows;
  A.rp.assign(rp, rp + nrows + 1);
  A.col.assign(col, col + rp[nrows]);
  A.val.assign(val, val + rp[nrows]);
  std::vector<int32_t> a;
  aggregate_level(A, block_size, threshold, max_aggregate_nodes, a, *n_coarse);
  std::copy(a.begin(), a.end(), agg);
  return ALFD_OK;
}

int alfd_get_aggregates(alfd_ctx_t ctx, int level, int32_t *agg, int64_t capacity, int64_t *n_fine, int64_t *n_coarse) {
  if (!ctx || level < 0 || level >= ALFD_MAX_LEVELS || ctx->ml_agg[level].empty()) return ALFD_E_INVALID;
  const std::vector<int32_t> &a = ctx->ml_agg[level];
  if (n_fine) *n_fine = (int64_t)a.size();
  if (n_coarse) *n_coarse = ctx->ml_ncoarse[level];
  if (agg) {
    if (capacity < (int64_t)a.size()) return ALFD_E_INVALID;
    std::copy(a.begin(), a.end(), agg);
  }
  return ALFD_OK;
}

int alfd_set_aggregate_partition(alfd_ctx_t ctx, int level, const int64_t *coarse_offsets) {
  CHECK_CTX();
  if (level < 0 || level >= ALFD_MAX_LEVELS || !coarse_offsets) return ALFD_E_INVALID;
  ctx->ml_coff[level].assign(coarse_offsets, coarse_offsets + ctx->nranks + 1);
  ctx->is_setup = false;
  return ALFD_OK;
}

int alfd_configure(alfd_ctx_t ctx, const alfd_config *cfg) {
  CHECK_CTX();
  if (!cfg) return ALFD_E_INVALID;
  if (cfg->variant < ALFD_AL2 || cfg->variant > ALFD_RATIONAL) return ALFD_E_INVALID;
  ctx->cfg = *cfg;
  ctx->configured = true;
  ctx->is_setup = false;
  return ALFD_OK;
}

int alfd_set_controls(alfd_ctx_t ctx, const alfd_control *outer, const alfd_control *inner, const alfd_control *mp_inner) {
  CHECK_CTX();
  if (!ctx->configured) return ctx->err = "alfd_configure not called", ALFD_E_NOT_SETUP;
  for (const alfd_control *c : {outer, inner, mp_inner})
    if (c && (c->kind < ALFD_CTRL_ABS || c->kind > ALFD_CTRL_FIXED_ITERS || c->max_steps < 0))
      return ctx->err = "alfd_set_controls: bad control", ALFD_E_INVALID;
  if (outer) ctx->cfg.outer = *outer;
  if (inner) ctx->cfg.inner = *inner;
  if (mp_inner) ctx->cfg.mp_inner = *mp_inner;
  return ALFD_OK;
}

int alfd_setup(alfd_ctx_t ctx) {
  CHECK_CTX();
  return setup(ctx);
}

int alfd_precond_apply(alfd_ctx_t ctx, const double *const *src, double *const *dst, alfd_result *res) {
  CHECK_CTX();
  CHECK_SETUP();
  if (!src || !dst) return ALFD_E_INVALID;
  reset_stats(ctx);
  RC(to_device(ctx, src, ctx->st_in));
  HIPC(hipMemsetAsync(ctx->st_out, 0, ctx->ntot() * sizeof(double), ctx->stream));
  const int rc = precond_apply(ctx, ctx->st_in, ctx->st_out);
  RC(to_host(ctx, ctx->st_out, dst));
  if (res) {
    std::memset(res, 0, sizeof(*res));
    fill_result(ctx, res, rc);
  }
  return rc;
}

int alfd_system_apply(alfd_ctx_t ctx, const double *const *src, double *const *dst) {
  CHECK_CTX();
  CHECK_SETUP();
  if (!src || !dst) return ALFD_E_INVALID;
  RC(to_device(ctx, src, ctx->st_in));
  HIPC(hipMemsetAsync(ctx->st_out, 0, ctx->ntot() * sizeof(double), ctx->stream));
  RC(system_apply(ctx, ctx->st_in, ctx->st_out));
  return to_host(ctx, ctx->st_out, dst);
}

int alfd_augment_rhs(alfd_ctx_t ctx, double *const *rhs) {
  CHECK_CTX();
  CHECK_SETUP();
  if (!rhs) return ALFD_E_INVALID;
  if (!ctx->diag[ALFD_INVW] || ctx->cfg.variant == ALFD_RATIONAL)
    return ctx->err = "rhs augmentation applies to the AL variants only", ALFD_E_UNSUPPORTED;
  const int last = ctx->nblocks - 1;
  RC(to_device(ctx, rhs, ctx->st_in));
  RC(winv_scale(ctx, 1.0, ctx->st_in + ctx->off[last], ctx->t_lam));
  RC(spmv(ctx, ALFD_CT, ctx->t_lam, ctx->st_in + ctx->off[0], 1, ctx->cfg.gamma));
  return to_host(ctx, ctx->st_in, rhs);
}

int alfd_upload_rhs(alfd_ctx_t ctx, const double *const *rhs, const double *const *x0) {
  CHECK_CTX();
  CHECK_SETUP();
  if (!rhs) return ALFD_E_INVALID;
  RC(to_device(ctx, rhs, ctx->bb));
  if (x0)
    RC(to_device(ctx, x0, ctx->io));
  else
    HIPC(hipMemsetAsync(ctx->io, 0, ctx->ntot() * sizeof(double), ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  return ALFD_OK;
}

int alfd_solve_resident(alfd_ctx_t ctx, alfd_result *res) {
  CHECK_CTX();
  CHECK_SETUP();
  if (!res) return ALFD_E_INVALID;
  std::memset(res, 0, sizeof(*res));
  reset_stats(ctx);
  // x <- initial guess kept in io (so the solve can be repeated)
  HIPC(hipMemcpyAsync(ctx->xb, ctx->io, ctx->ntot() * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = ctx->cfg.outer_solver == ALFD_OUTER_MINRES               ? minres(ctx, res)
                 : ctx->cfg.fgmres_flavour == ALFD_FGMRES_DEALII_95 ? fgmres_dealii95(ctx, res)
                                                                    : fgmres(ctx, res);
  hipStreamSynchronize(ctx->stream);
  res->solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  flush_timers(ctx);
  fill_result(ctx, res, rc);
  return rc;
}

int alfd_download_solution(alfd_ctx_t ctx, double *const *x) {
  CHECK_CTX();
  CHECK_SETUP();
  if (!x) return ALFD_E_INVALID;
  return to_host(ctx, ctx->xb, x);
}

int alfd_solve(alfd_ctx_t ctx, const double *const *rhs, double *const *x, alfd_result *res) {
  CHECK_CTX();
  CHECK_SETUP();
  if (!rhs || !x || !res) return ALFD_E_INVALID;
  RC(alfd_upload_rhs(ctx, rhs, x));
  const int rc = alfd_solve_resident(ctx, res);
  const int rc2 = alfd_download_solution(ctx, x);
  return rc != ALFD_OK ? rc : rc2;
}

int alfd_get_history(alfd_ctx_t ctx, double *out, int32_t capacity, int32_t *count) {
  if (!ctx) return ALFD_E_INVALID;
  if (count) *count = (int32_t)ctx->history.size();
  if (out)
    for (int32_t i = 0; i < capacity && i < (int32_t)ctx->history.size(); ++i) out[i] = ctx->history[i];
  return ALFD_OK;
}

int alfd_spmv(alfd_ctx_t ctx, int slot, const double *x, double *y, int mode, double alpha) {
  CHECK_CTX();
  if (slot < 0 || slot >= ALFD_NSLOTS || !ctx->mat[slot].present) return ALFD_E_INVALID;
  if (!x || !y || (mode != 0 && mode != 1)) return ALFD_E_INVALID;
  const DevCsr &m = ctx->mat[slot];
  // partitioned context: collective over the ranks; x = this rank's owned columns, y = its rows
  const int64_t nx = ctx->nranks > 1 ? (int64_t)m.n_local_cols : m.ncols;
  double *dx = nullptr, *dy = nullptr;
  HIPC(hipMalloc((void **)&dx, std::max<int64_t>(nx, 1) * sizeof(double)));
  HIPC(hipMalloc((void **)&dy, std::max<int64_t>(m.nrows, 1) * sizeof(double)));
  HIPC(hipMemcpyAsync(dx, x, nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(dy, y, m.nrows * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  const int rc = spmv(ctx, slot, dx, dy, mode, alpha);
  if (rc == ALFD_OK) {
    HIPC(hipMemcpyAsync(y, dy, m.nrows * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPC(hipStreamSynchronize(ctx->stream));
  }
  hipFree(dx);
  hipFree(dy);
  return rc;
}

int alfd_dot(alfd_ctx_t ctx, int64_t n, const double *x, const double *y, double *result) {
  CHECK_CTX();
  if (n < 0 || !x || !y || !result) return ALFD_E_INVALID;
  const int64_t npad = pad_chunk(std::max<int64_t>(n, 1));
  double *dx = nullptr, *dy = nullptr, *part = nullptr, *sc = nullptr;
  HIPC(hipMalloc((void **)&dx, npad * sizeof(double)));
  HIPC(hipMalloc((void **)&dy, npad * sizeof(double)));
  HIPC(hipMalloc((void **)&part, (npad / kChunk) * sizeof(double)));
  HIPC(hipMalloc((void **)&sc, kNumScalars * sizeof(double)));
  HIPC(hipMemsetAsync(dx, 0, npad * sizeof(double), ctx->stream));
  HIPC(hipMemsetAsync(dy, 0, npad * sizeof(double), ctx->stream));
  HIPC(hipMemcpyAsync(dx, x, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPC(hipMemcpyAsync(dy, y, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(dot_partial_kernel, dim3((unsigned)(npad / kChunk)), dim3(kBlock), 0, ctx->stream, dx,
                     dy, part);
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, part, npad / kChunk,
                     (int64_t)0, sc, (int)S_TMP, (int)FIN_STORE);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(result, sc + S_TMP, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIPC(hipStreamSynchronize(ctx->stream));
  hipFree(dx);
  hipFree(dy);
  hipFree(part);
  hipFree(sc);
  return ALFD_OK;
}

int alfd_host_halo_plan(int64_t nnz, const int32_t *col, const int64_t *col_offsets, int nranks, int rank,
                        int32_t *col_local, int32_t *halo_globals, int64_t halo_capacity, int64_t *n_halo,
                        int64_t *recv_off) {
  if (nnz < 0 || nranks < 1 || rank < 0 || rank >= nranks || !col_offsets || !col_local || !n_halo || !recv_off ||
      (nnz > 0 && !col))
    return ALFD_E_INVALID;
  std::vector<int32_t> hal;
  host_halo_plan(nnz, col, col_offsets, nranks, rank, col_local, hal, recv_off);
  *n_halo = (int64_t)hal.size();
  if (halo_globals) {
    if ((int64_t)hal.size() > halo_capacity) return ALFD_E_INVALID;
    std::copy(hal.begin(), hal.end(), halo_globals);
  }
  return ALFD_OK;
}

int alfd_matrix_lanes(alfd_ctx_t ctx, int slot, int32_t *lanes) {
  if (!ctx || slot < 0 || slot >= ALFD_NSLOTS || !lanes || !ctx->mat[slot].present) return ALFD_E_INVALID;
  *lanes = ctx->mat[slot].L;
  return ALFD_OK;
}

int alfd_bench_spmv(alfd_ctx_t ctx, int slot, int32_t reps, double *ms_per_launch, double *bytes) {
  CHECK_CTX();
  if (slot < 0 || slot >= ALFD_NSLOTS || !ctx->mat[slot].present || reps < 1) return ALFD_E_INVALID;
  if (ctx->nranks > 1) return ALFD_E_UNSUPPORTED;
  const DevCsr &m = ctx->mat[slot];
  double *dx = nullptr, *dy = nullptr;
  HIPC(hipMalloc((void **)&dx, std::max<int64_t>(m.ncols, 1) * sizeof(double)));
  HIPC(hipMalloc((void **)&dy, std::max<int64_t>(m.nrows, 1) * sizeof(double)));
  hipLaunchKernelGGL(hash_vector_kernel, dim3((unsigned)std::max<int64_t>(1, (m.ncols + 255) / 256)), dim3(256), 0, ctx->stream,
                     m.ncols, (int64_t)0, dx);
  const int was = ctx->timing;
  ctx->timing = 0;
  for (int i = 0; i < 2; ++i) RC(spmv(ctx, slot, dx, dy, 0));
  hipEvent_t a, b;
  HIPC(hipEventCreate(&a));
  HIPC(hipEventCreate(&b));
  HIPC(hipEventRecord(a, ctx->stream));
  for (int i = 0; i < reps; ++i) RC(spmv(ctx, slot, dx, dy, 0));
  HIPC(hipEventRecord(b, ctx->stream));
  HIPC(hipEventSynchronize(b));
  float ms = 0;
  HIPC(hipEventElapsedTime(&ms, a, b));
  ctx->timing = was;
  hipEventDestroy(a);
  hipEventDestroy(b);
  hipFree(dx);
  hipFree(dy);
  if (ms_per_launch) *ms_per_launch = (double)ms / reps;
  if (bytes) *bytes = m.algorithmic_bytes();
  return ALFD_OK;
}

int alfd_bench_spmv_format(alfd_ctx_t ctx, int slot, int32_t reps, int use_value_index, double *ms_per_launch,
                           double *streamed_bytes) {
  CHECK_CTX();
  if (slot < 0 || slot >= ALFD_NSLOTS || !ctx->mat[slot].present) return ALFD_E_INVALID;
  const bool was_off = ctx->vi_off;
  if (!use_value_index) RC(ensure_window_plan(ctx, ctx->mat[slot]));
  ctx->vi_off = !use_value_index;
  const int rc = alfd_bench_spmv(ctx, slot, reps, ms_per_launch, nullptr);
  ctx->vi_off = was_off;
  if (streamed_bytes) *streamed_bytes = ctx->mat[slot].streamed_bytes(use_value_index != 0, ctx->vs_enable != 0);
  return rc;
}

int alfd_host_window_plan(int64_t nrows, const int64_t *rp, const int32_t *col, const double *val, int32_t lanes,
                          int32_t want_value_index, alfd_window_plan_info *out) {
  if (nrows < 0 || !rp || !out || (rp[nrows] > 0 && (!col || !val))) return ALFD_E_INVALID;
  if (lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32 && lanes != 64) return ALFD_E_INVALID;
  WindowParams wp;
  wp.want_vi = want_value_index != 0;
  WindowPlan pl;
  plan_window(nrows, lanes, rp, col, val, wp, pl);
  std::memset(out, 0, sizeof(*out));
  out->windowed = pl.win;
  out->value_indexed = pl.vi;
  out->row_block = pl.RB;
  out->max_window = pl.maxW;
  out->blocks = pl.nb;
  out->fallback_blocks = pl.fallback;
  out->segments = (int64_t)pl.seg_col.size();
  out->value_indexed_blocks = pl.vi ? pl.vi_blocks : 0;
  out->value_indexed_nnz = pl.vi ? pl.vi_nnz : 0;
  out->value_wide_nnz = pl.vi ? pl.wide_nnz : 0;
  out->dictionary_entries = pl.vi ? (int64_t)pl.dict.size() : 0;
  if (!pl.win) return ALFD_OK;
  // decode the plan back and compare with the CSR it was made from
  int64_t bad = 0, batches = 0;
  std::vector<int32_t> wcol;
  std::vector<int> seen;
  for (int64_t b = 0; b < pl.nb; ++b) {
    const int64_t r0 = b * pl.RB, r1 = std::min<int64_t>(r0 + pl.RB, nrows);
    const int64_t k0 = rp[r0], k1 = rp[r1];
    const int32_t W = pl.blkW[b];
    if (W >= 0) {
      wcol.assign(W, -1);
      for (int32_t sg = pl.seg_begin[b]; sg < pl.seg_begin[b + 1]; ++sg) {
        const int32_t end = (sg + 1 < pl.seg_begin[b + 1]) ? pl.seg_off[sg + 1] : W;
        for (int32_t o = pl.seg_off[sg]; o < end; ++o) wcol[o] = pl.seg_col[sg] + (o - pl.seg_off[sg]);
      }
      for (int64_t k = k0; k < k1; ++k) bad += pl.lcol[k] >= W || wcol[pl.lcol[k]] != col[k];
    }
    if (pl.vi && W >= 0 && pl.blk_dn[b] >= 0) {
      const bool wide = (pl.blk_dn[b] & kDictWide) != 0;
      const int32_t nd = pl.blk_dn[b] & 0xffff;
      for (int64_t k = k0; k < k1; ++k) {
        const int32_t code = wide ? (int32_t)pl.vidw[k] : (int32_t)pl.vidx[k];
        if (code >= nd || std::memcmp(&pl.dict[pl.doff[b] + code], &val[k], 8) != 0) ++bad;
      }
    }
    if (pl.stride > 0) {
      seen.assign(r1 - r0, 0);
      batches += pl.cnt[b];
      for (int32_t q = 0; q < pl.cnt[b]; ++q) {
        const uint64_t *dsc = &pl.tab[((size_t)b * pl.stride + q) * 4];
        const int cls = (int)(dsc[0] >> 56);
        for (int i = 0; i < 4; ++i) {
          const int id = (int)((dsc[i] >> 48) & 0xff);
          const int64_t ks = (int64_t)(dsc[i] & 0xffffffffull), len = (int64_t)((dsc[i] >> 32) & 0xffff);
          if ((int)(dsc[i] >> 56) != cls) ++bad;
          if (id == 0xff) {  // filler repeats the batch's first row
            bad += ks != (int64_t)(dsc[0] & 0xffffffffull) || i == 0;
            continue;
          }
          if (id >= r1 - r0) {
            ++bad;
            continue;
          }
          ++seen[id];
          const int64_t rl = rp[r0 + id + 1] - rp[r0 + id];
          const int64_t want_cls = std::min<int64_t>((rl + 63) / 64, kVibMaxClass + 1);
          bad += ks != rp[r0 + id] - k0 || len != rl || cls != want_cls;
        }
      }
      for (int v : seen) bad += v != 1;
    }
  }
  out->batches = batches;
  out->decode_mismatches = bad;
  return ALFD_OK;
}

int alfd_get_matrix_info(alfd_ctx_t ctx, int slot, alfd_matrix_info *out) {
  CHECK_CTX();
  if (slot < 0 || slot >= ALFD_NSLOTS || !ctx->mat[slot].present || !out) return ALFD_E_INVALID;
  const DevCsr &m = ctx->mat[slot];
  std::memset(out, 0, sizeof(*out));
  out->lanes = m.L;
  out->windowed = (m.win || m.win_deferred) ? 1 : 0;
  out->value_indexed = (m.vi || (m.vs.on && ctx->vs_enable)) ? 1 : 0;
  out->batch_major = (m.vs.on && ctx->vs_enable) ? (m.vs.bricks ? 2 : 1) : 0;
  out->nnz = m.nnz;
  out->window_blocks = m.win_nblocks;
  out->window_fallback_blocks = m.win_fallback_blocks;
  out->value_indexed_blocks = m.vi_blocks;
  out->value_indexed_nnz = m.vs.on && ctx->vs_enable ? m.nnz : m.vi_nnz;   // every entry of a batch-major matrix is dictionary-coded
  out->dictionary_entries = m.vi_dict_total;
  out->value_wide_nnz = m.vi_wide_nnz;
  out->algorithmic_bytes = m.algorithmic_bytes();
  out->streamed_bytes = m.streamed_bytes(true, ctx->vs_enable != 0);
  out->shared_nnz = m.vs.on ? m.vs.shared_nnz : 0;
  out->batch_major_blocks = m.vs.on ? m.vs.nb : 0;
  out->batch_major_wide = m.vs.on ? m.vs.wide : 0;
  out->batch_major_interior_blocks = (m.vs.on && m.vs.L == 64 && ctx->nranks > 1 && !m.rep) ? m.vs.nb_interior : 0;
  return ALFD_OK;
}

// Recursive coordinate bisection of the rows' support points: leaves of at most max_rows rows, cut at the
// median of the axis with the largest (weighted) extent.  The axis along which consecutive rows mostly
// advance -- the fast axis of the numbering -- counts half, so that bricks come out about twice as long
// there: the x window of a block is staged in runs of consecutive columns.
namespace {
struct Rcb {
  int dim;
  const double *xyz;
  int64_t max_rows;
  double weight[3];
  std::vector<int32_t> idx;
  std::vector<std::vector<int64_t>> leaves;   // leaf sizes per top-level task, in order
  void split(int64_t a, int64_t e, std::vector<int64_t> &out) {
    if (e - a <= max_rows) {
      if (e > a) out.push_back(e - a);
      return;
    }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = a; i < e; ++i)
      for (int d = 0; d < dim; ++d) {
        const double c = xyz[(int64_t)idx[i] * dim + d];
        lo[d] = std::min(lo[d], c);
        hi[d] = std::max(hi[d], c);
      }
    int ax = 0;
    for (int d = 1; d < dim; ++d)
      if ((hi[d] - lo[d]) * weight[d] > (hi[ax] - lo[ax]) * weight[ax]) ax = d;
    const int64_t mid = cut(a, e, ax);
    split(a, mid, out);
    split(mid, e, out);
  }
  // Cut [a, e) along axis ax at a coordinate PLANE near the median: every point below the plane goes left, so
  // that leaves are whole boxes of the mesh (their x windows are regular, which is what lets translate rows
  // share a template).  Returns the split position.
  int64_t cut(int64_t a, int64_t e, int ax) {
    const int64_t mid = a + (e - a) / 2;
    auto less = [&](int32_t p, int32_t q) {
      const double cp = xyz[(int64_t)p * dim + ax], cq = xyz[(int64_t)q * dim + ax];
      return cp < cq || (cp == cq && p < q);
    };
    std::nth_element(idx.begin() + a, idx.begin() + mid, idx.begin() + e, less);
    const double cm = xyz[(int64_t)idx[mid] * dim + ax];
    // points with coordinate == cm sit on both sides of mid: move the boundary to the nearer end of that plane
    auto lo = std::partition(idx.begin() + a, idx.begin() + mid, [&](int32_t p) { return xyz[(int64_t)p * dim + ax] < cm; });
    auto hi = std::partition(idx.begin() + mid, idx.begin() + e, [&](int32_t p) { return xyz[(int64_t)p * dim + ax] <= cm; });
    const int64_t l = lo - idx.begin(), h = hi - idx.begin();   // [l, h) = the plane cm
    int64_t pos = (mid - l <= h - mid) ? l : h;
    if (pos == a) pos = h;
    if (pos == e) pos = l;
    if (pos == a || pos == e) pos = mid;   // a single plane (cannot happen: ax has a positive extent)
    return pos;
  }
};
}  // namespace

int alfd_host_row_blocks_from_points(int64_t nrows, int32_t dim, const double *points, int32_t max_rows,
                                     int64_t *n_blocks_out, int64_t *block_ptr_out, int32_t *rows_out) {
  if (nrows < 0 || dim < 1 || dim > 3 || !points || max_rows < 1 || max_rows > kVsMaxRows || !n_blocks_out ||
      !block_ptr_out || !rows_out || nrows > 2147483000LL)
    return ALFD_E_INVALID;
  Rcb r;
  r.dim = dim;
  r.xyz = points;
  r.max_rows = max_rows;
  r.idx.resize(nrows);
  for (int64_t i = 0; i < nrows; ++i) r.idx[i] = (int32_t)i;
  // fast axis of the numbering: the axis along which consecutive rows most often differ
  int64_t moves[3] = {0, 0, 0};
  for (int64_t i = 0; i + 1 < nrows; i += std::max<int64_t>(1, nrows / 100000)) {
    int best = -1;
    double bd = 0;
    for (int d = 0; d < dim; ++d) {
      const double dd = std::fabs(points[(i + 1) * dim + d] - points[i * dim + d]);
      if (dd > bd) bd = dd, best = d;
    }
    if (best >= 0) ++moves[best];
  }
  int fast = 0;
  for (int d = 1; d < dim; ++d)
    if (moves[d] > moves[fast]) fast = d;
  for (int d = 0; d < 3; ++d) r.weight[d] = d == fast ? 0.5 : 1.0;
  // top levels sequentially into 2^k tasks, the tasks in parallel
  std::vector<std::pair<int64_t, int64_t>> tasks(1, {0, nrows});
  const int T = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  while ((int)tasks.size() < T && nrows > (int64_t)max_rows * 64) {
    std::vector<std::pair<int64_t, int64_t>> next;
    for (auto [a, e] : tasks) {
      if (e - a <= max_rows) {
        next.push_back({a, e});
        continue;
      }
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      for (int64_t i = a; i < e; ++i)
        for (int d = 0; d < dim; ++d) {
          const double c = points[(int64_t)r.idx[i] * dim + d];
          lo[d] = std::min(lo[d], c);
          hi[d] = std::max(hi[d], c);
        }
      int ax = 0;
      for (int d = 1; d < dim; ++d)
        if ((hi[d] - lo[d]) * r.weight[d] > (hi[ax] - lo[ax]) * r.weight[ax]) ax = d;
      const int64_t mid = r.cut(a, e, ax);
      next.push_back({a, mid});
      next.push_back({mid, e});
    }
    tasks.swap(next);
  }
  r.leaves.assign(tasks.size(), {});
  {
    std::vector<std::thread> th;
    std::atomic<size_t> nextTask(0);
    for (int t = 0; t < T; ++t)
      th.emplace_back([&]() {
        for (size_t q = nextTask++; q < tasks.size(); q = nextTask++) r.split(tasks[q].first, tasks[q].second, r.leaves[q]);
      });
    for (auto &x : th) x.join();
  }
  int64_t nb = 0;
  block_ptr_out[0] = 0;
  for (auto &lv : r.leaves)
    for (int64_t sz : lv) {
      block_ptr_out[nb + 1] = block_ptr_out[nb] + sz;
      ++nb;
    }
  // rows of a block in ascending order (the order the numbering visits them)
  for (int64_t b = 0; b < nb; ++b) std::sort(r.idx.begin() + block_ptr_out[b], r.idx.begin() + block_ptr_out[b + 1]);
  std::copy(r.idx.begin(), r.idx.end(), rows_out);
  *n_blocks_out = nb;
  return ALFD_OK;
}

int alfd_host_stream_plan(int64_t nrows, const int64_t *rp, const int32_t *col, const double *val, int32_t row_block,
                          int64_t n_blocks, const int64_t *block_ptr, const int32_t *rows,
                          alfd_stream_plan_info *out) {
  if (!rp || !out || nrows < 0 || row_block < 1 || row_block > kVsMaxRows) return ALFD_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  VsPlan pl;
  if (n_blocks > 0) plan_vs(nrows, rp, col, val, row_block, 4096, 8, n_blocks, block_ptr, rows, pl);
  else plan_vs(nrows, rp, col, val, row_block, 4096, 8, 0, nullptr, nullptr, pl);
  if (!pl.ok || pl.dict_split) {   // as build_vs: wide codes when the 512-value limit halved blocks
    VsPlan pw;
    if (n_blocks > 0) plan_vs(nrows, rp, col, val, row_block, 4096, 8, n_blocks, block_ptr, rows, pw, true, 1);
    else plan_vs(nrows, rp, col, val, row_block, 4096, 8, 0, nullptr, nullptr, pw, true, 1);
    if (pw.ok && (!pl.ok || pw.stream.size() + 128 * (size_t)pw.nbatch < pl.stream.size() + 128 * (size_t)pl.nbatch)) pl = std::move(pw);
  }
  const int code_shift = pl.wide ? VsFmt<1>::kCodeShift : VsFmt<0>::kCodeShift;
  out->ok = pl.ok ? 1 : 0;
  if (!pl.ok) return ALFD_OK;
  out->max_window = pl.maxW;
  out->max_rows = pl.rbs;
  out->max_batches = pl.stride;
  out->blocks = pl.nb;
  out->batches = pl.nbatch;
  out->segments = (int64_t)pl.seg_col.size();
  out->dictionary_entries = (int64_t)pl.dict.size();
  out->stream_bytes = (int64_t)pl.stream.size() - 4096;
  out->shared_nnz = pl.shared_nnz;
  std::vector<uint8_t> seen(nrows, 0);
  int64_t bad = 0, covered = 0;
  std::vector<int32_t> slot_col;
  for (int64_t b = 0; b < pl.nb; ++b) {
    // window slot -> column
    slot_col.assign(pl.blkW[b], -1);
    for (int32_t s = pl.seg_begin[b]; s < pl.seg_begin[b + 1]; ++s) {
      const int32_t end = s + 1 < pl.seg_begin[b + 1] ? pl.seg_off[s + 1] : pl.blkW[b];
      for (int32_t o = pl.seg_off[s]; o < end; ++o) slot_col[o] = pl.seg_col[s] + (o - pl.seg_off[s]);
    }
    const int nbt = pl.cnt[b];
    const uint8_t *sp = pl.stream.data() + pl.sb[b];
    if (pl.sb[b] % 16) ++bad;
    for (int q = 0; q < nbt; ++q) {
      const uint64_t *dsc = &pl.tab[((size_t)b * pl.stride + q) * kVsBatchRows];
      auto off = [](uint64_t dd) { return (uint32_t)dd & 0xfffffu; };
      auto cnt_of = [](uint64_t dd) { return ((uint32_t)dd >> 20) & 0x1ffu; };
      const uint32_t eb = off(dsc[0]);
      const bool shared = (dsc[0] >> 63) != 0;
      const int cls = (int)(((uint32_t)dsc[0] >> 29) & 7u);
      const uint8_t *fb = sp + 16 * (size_t)eb;
      for (int i = 0; i < kVsBatchRows; ++i) {
        const uint32_t rfield = (uint32_t)(dsc[i] >> 32) & (i == 0 ? 0x7fffffffu : 0xffffffffu);
        if (!shared && i >= 4 && (int32_t)rfield >= 0) ++bad;   // plain batches hold 4 rows
        const int64_t r = (int32_t)rfield;
        if (r < 0) continue;  // filler
        if (r >= nrows || seen[r]) { ++bad; continue; }
        seen[r] = 1;
        ++covered;
        const uint32_t n = cnt_of(dsc[shared ? 0 : i]);        // translates share the template's count
        if ((int64_t)n != rp[r + 1] - rp[r] || (int)((n + 63) / 64) != cls) { ++bad; continue; }
        if (!shared && off(dsc[i]) != eb) { ++bad; continue; }
        const int32_t shift = (shared && i > 0) ? (int32_t)(uint32_t)dsc[i] : 0;   // bytes
        if (shift % 8) ++bad;
        for (uint32_t k = 0; k < n; ++k) {
          uint32_t f;
          if (shared) {
            std::memcpy(&f, fb + 256 * (size_t)(k / 64) + 4 * (size_t)(k % 64), 4);
          } else {
            const uint8_t *cell = fb + 768 * (size_t)(k / 64) + 12 * (size_t)(k % 64) + 3 * i;
            f = cell[0] | ((uint32_t)cell[1] << 8) | ((uint32_t)cell[2] << 16);
          }
          if ((f & 7u) || (f >> 24)) ++bad;
          const int32_t lcv = (int32_t)((f >> 3) & ((1u << (code_shift - 3)) - 1u)) + shift / 8;
          const uint32_t vcv = f >> code_shift;
          const double v = (int32_t)vcv < pl.dn[b] ? pl.dict[pl.doff[b] + vcv] : std::nan("");
          const int32_t c = (lcv >= 0 && lcv < pl.blkW[b]) ? slot_col[lcv] : -1;
          if (c != col[rp[r] + k] || std::memcmp(&v, &val[rp[r] + k], 8) != 0) ++bad;
        }
      }
    }
  }
  out->decode_mismatches = bad;
  out->rows_covered = covered;
  return ALFD_OK;
}

int alfd_host_stream_plan_short(int64_t nrows, const int64_t *rp, const int32_t *col, const double *val, int32_t lanes,
                                alfd_stream_plan_info *out) {
  if (!rp || !out || nrows < 0 || (lanes != 8 && lanes != 16 && lanes != 32)) return ALFD_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  VsPlan pl;
  const int L = lanes, RBb = 4 * (64 / L), W64 = 1 + RBb;
  plan_vss(nrows, L, rp, col, val, std::min(kVssMaxRows, 96 * 2 * 64 / L), 4096, 8, pl);
  out->ok = pl.ok ? 1 : 0;
  if (!pl.ok) return ALFD_OK;
  out->max_window = pl.maxW;
  out->max_rows = pl.rbs;
  out->max_batches = pl.stride;
  out->blocks = pl.nb;
  out->batches = pl.nbatch;
  out->segments = (int64_t)pl.seg_col.size();
  out->dictionary_entries = (int64_t)pl.dict.size();
  out->stream_bytes = (int64_t)pl.stream.size() - 4096 + 8 * (int64_t)W64 * pl.nbatch;
  out->shared_nnz = pl.shared_nnz;
  std::vector<uint8_t> seen(nrows, 0);
  int64_t bad = 0, covered = 0;
  std::vector<int32_t> slot_col;
  for (int64_t b = 0; b < pl.nb; ++b) {
    slot_col.assign(pl.blkW[b], -1);
    for (int32_t s = pl.seg_begin[b]; s < pl.seg_begin[b + 1]; ++s) {
      const int32_t end = s + 1 < pl.seg_begin[b + 1] ? pl.seg_off[s + 1] : pl.blkW[b];
      for (int32_t o = pl.seg_off[s]; o < end; ++o) slot_col[o] = pl.seg_col[s] + (o - pl.seg_off[s]);
    }
    const uint8_t *sp = pl.stream.data() + pl.sb[b];
    for (int q = 0; q < pl.cnt[b]; ++q) {
      const uint64_t *dsc = &pl.tab[((size_t)b * pl.stride + q) * W64];
      const uint32_t eb = (uint32_t)dsc[0] & 0xfffffu, n = ((uint32_t)dsc[0] >> 20) & 0x1ffu;
      const int cls = (int)(((uint32_t)dsc[0] >> 29) & 7u), nreal = (int)(dsc[0] >> 32);
      if ((int)((n + L - 1) / L) != cls || nreal < 1 || nreal > RBb) ++bad;
      for (int i = 0; i < RBb; ++i) {
        const int64_t r = (int32_t)(uint32_t)dsc[1 + i];
        const int32_t shift = (int32_t)(dsc[1 + i] >> 32);
        if (r < 0) {
          if (i < nreal) ++bad;
          continue;
        }
        if (i >= nreal || r >= nrows || seen[r] || (int64_t)n != rp[r + 1] - rp[r] || shift % 8) { ++bad; continue; }
        seen[r] = 1;
        ++covered;
        for (uint32_t k = 0; k < n; ++k) {
          uint32_t f;
          std::memcpy(&f, sp + 16 * (size_t)eb + 4 * (size_t)k, 4);
          if ((f & 7u) || (f >> 24)) ++bad;
          const int32_t lcv = (int32_t)((f >> 3) & 0xfffu) + shift / 8;
          const uint32_t vcv = f >> 15;
          const double v = (int32_t)vcv < pl.dn[b] ? pl.dict[pl.doff[b] + vcv] : std::nan("");
          const int32_t c = (lcv >= 0 && lcv < pl.blkW[b]) ? slot_col[lcv] : -1;
          if (c != col[rp[r] + k] || std::memcmp(&v, &val[rp[r] + k], 8) != 0) ++bad;
        }
      }
    }
  }
  out->decode_mismatches = bad;
  out->rows_covered = covered;
  return ALFD_OK;
}

int alfd_host_numbering_from_points(int64_t nrows, int32_t dim, const double *points, int64_t *new_to_old) {
  if (nrows < 0 || dim < 1 || dim > 3 || !points || !new_to_old) return ALFD_E_INVALID;
  for (int64_t i = 0; i < nrows; ++i) new_to_old[i] = i;
  std::stable_sort(new_to_old, new_to_old + nrows, [&](int64_t a, int64_t b) {
    for (int d = dim - 1; d >= 0; --d) {
      const double pa = points[a * dim + d], pb = points[b * dim + d];
      if (pa != pb) return pa < pb;
    }
    return false;
  });
  return ALFD_OK;
}

int alfd_host_brick_blocks_from_points(int64_t nrows, int32_t dim, const double *points, const int32_t *brick,
                                       int32_t max_rows, int64_t *n_blocks_out, int64_t *block_ptr_out, int32_t *rows_out) {
  if (nrows < 0 || dim < 1 || dim > 3 || !points || !brick || max_rows < 1 || max_rows > kVsMaxRows || !n_blocks_out ||
      !block_ptr_out || !rows_out || nrows > 2147483000LL)
    return ALFD_E_INVALID;
  for (int d = 0; d < dim; ++d)
    if (brick[d] < 1) return ALFD_E_INVALID;
  // grid index of a point along an axis = rank of its coordinate among the distinct coordinates of the axis
  std::vector<std::vector<int32_t>> gi(dim, std::vector<int32_t>(nrows));
  int64_t span[3] = {1, 1, 1};
  for (int d = 0; d < dim; ++d) {
    std::vector<double> u(nrows);
    for (int64_t i = 0; i < nrows; ++i) u[i] = points[i * dim + d];
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
    for (int64_t i = 0; i < nrows; ++i)
      gi[d][i] = (int32_t)(std::lower_bound(u.begin(), u.end(), points[i * dim + d]) - u.begin());
    span[d] = (int64_t)u.size() / brick[d] + 1;
  }
  std::vector<std::pair<int64_t, int32_t>> key(nrows);   // (brick id, row): rows of a brick in ascending order
  for (int64_t i = 0; i < nrows; ++i) {
    int64_t k = 0;
    for (int d = dim - 1; d >= 0; --d) k = k * span[d] + gi[d][i] / brick[d];
    key[i] = {k, (int32_t)i};
  }
  std::sort(key.begin(), key.end());
  int64_t nb = 0;
  block_ptr_out[0] = 0;
  for (int64_t i = 0; i < nrows;) {
    int64_t j = i;
    while (j < nrows && key[j].first == key[i].first) ++j;
    for (int64_t a = i; a < j; a += max_rows) block_ptr_out[++nb] = std::min<int64_t>(a + max_rows, j);
    i = j;
  }
  for (int64_t i = 0; i < nrows; ++i) rows_out[i] = key[i].second;
  *n_blocks_out = nb;
  return ALFD_OK;
}

int alfd_host_permute_csr(int64_t nrows, const int64_t *rp, const int32_t *col, const double *val,
                          const int64_t *row_new_to_old, const int64_t *col_old_to_new, int64_t *orp, int32_t *ocol,
                          double *oval) {
  if (nrows < 0 || !rp || !orp || (rp[nrows] > 0 && (!col || !val || !ocol || !oval))) return ALFD_E_INVALID;
  orp[0] = 0;
  for (int64_t r = 0; r < nrows; ++r) {
    const int64_t src = row_new_to_old ? row_new_to_old[r] : r;
    if (src < 0 || src >= nrows) return ALFD_E_INVALID;
    orp[r + 1] = orp[r] + (rp[src + 1] - rp[src]);
  }
  if (orp[nrows] != rp[nrows]) return ALFD_E_INVALID;   // row_new_to_old is not a permutation
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())),
                                                           (nrows + 8191) / 8192));
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t]() {
      std::vector<std::pair<int32_t, double>> row;
      for (int64_t r = nrows * t / T; r < nrows * (t + 1) / T; ++r) {
        const int64_t src = row_new_to_old ? row_new_to_old[r] : r;
        const int64_t k0 = rp[src], len = rp[src + 1] - k0, o0 = orp[r];
        if (!col_old_to_new) {
          for (int64_t k = 0; k < len; ++k) ocol[o0 + k] = col[k0 + k], oval[o0 + k] = val[k0 + k];
          continue;
        }
        row.resize(len);
        for (int64_t k = 0; k < len; ++k) row[k] = {(int32_t)col_old_to_new[col[k0 + k]], val[k0 + k]};
        std::sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        for (int64_t k = 0; k < len; ++k) ocol[o0 + k] = row[k].first, oval[o0 + k] = row[k].second;
      }
    });
  for (auto &x : th) x.join();
  return ALFD_OK;
}

int alfd_get_device_memory(alfd_ctx_t ctx, int64_t *free_bytes, int64_t *total_bytes) {
  CHECK_CTX();
  size_t f = 0, t = 0;
  HIPC(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return ALFD_OK;
}

int alfd_set_tunable(alfd_ctx_t ctx, const char *name, int value) {
  if (!ctx || !name) return ALFD_E_INVALID;
  if (std::strcmp(name, "value_index") == 0) {
    ctx->vi_off = value == 0;
    if (value == 0) RC(ensure_window_plans(ctx));   // the general kernel needs the (deferred) window plans
    return ALFD_OK;
  }
  if (std::strcmp(name, "batch_major") == 0) {   // takes effect at the next alfd_set_matrix
    ctx->vs_enable = value != 0;
    if (value == 0) RC(ensure_window_plans(ctx));
    return ALFD_OK;
  }
  if (std::strcmp(name, "batch_major_share") == 0) {   // 0: no template-shared batches (takes effect at the next alfd_set_matrix)
    ctx->vs_share = value != 0;
    return ALFD_OK;
  }
  if (std::strcmp(name, "batch_major_wide") == 0) {   // 0: never plan 10-bit codes (at the next alfd_set_matrix)
    ctx->vs_wide = value != 0;
    return ALFD_OK;
  }
  if (std::strcmp(name, "batch_major_waves") == 0) {
    if (value != 2 && value != 4 && value != 8) return ctx->err = "batch_major_waves: 2, 4 or 8", ALFD_E_INVALID;
    ctx->vs_NW = value;
    return ALFD_OK;
  }
  if (std::strcmp(name, "batch_major_rows") == 0) {
    if (value < 4 || value > kVsMaxRows) return ctx->err = "batch_major_rows: 4..250", ALFD_E_INVALID;
    ctx->vs_RB = value;
    return ALFD_OK;
  }
  if (std::strcmp(name, "batch_major_xcd") == 0) {
    ctx->vs_xcd = value != 0;
    return ALFD_OK;
  }
  return ctx->err = std::string("unknown tunable ") + name, ALFD_E_INVALID;
}

int alfd_set_row_blocks(alfd_ctx_t ctx, int slot, int64_t n_blocks, const int64_t *block_ptr, const int32_t *rows) {
  if (!ctx || slot < 0 || slot >= ALFD_NSLOTS) return ALFD_E_INVALID;
  ctx->rb_ptr[slot].clear();
  ctx->rb_rows[slot].clear();
  if (n_blocks <= 0 || !block_ptr || !rows) return ALFD_OK;   // hint removed
  // the prefix is checked before anything is copied with it (a negative, huge or non-monotone value would be
  // undefined behaviour in vector::assign); that the blocks partition the rows is checked against the matrix at upload
  if (block_ptr[0] != 0) return ctx->err = "alfd_set_row_blocks: block_ptr[0] must be 0", ALFD_E_INVALID;
  for (int64_t b = 0; b < n_blocks; ++b)
    if (block_ptr[b + 1] < block_ptr[b]) return ctx->err = "alfd_set_row_blocks: block_ptr must be monotone", ALFD_E_INVALID;
  if (block_ptr[n_blocks] > 2147483000LL) return ctx->err = "alfd_set_row_blocks: more than 2^31 rows", ALFD_E_INVALID;
  ctx->rb_ptr[slot].assign(block_ptr, block_ptr + n_blocks + 1);
  ctx->rb_rows[slot].assign(rows, rows + block_ptr[n_blocks]);
  return ALFD_OK;
}

int alfd_enable_timing(alfd_ctx_t ctx, int on) {
  if (!ctx) return ALFD_E_INVALID;
  ctx->timing = on < 0 ? 0 : on;
  for (int i = 0; i < ALFD_T_NCLASSES; ++i) ctx->t_ms[i] = 0, ctx->t_launches[i] = 0, ctx->t_bytes[i] = 0, ctx->t_fbytes[i] = 0;
  return ALFD_OK;
}
int alfd_get_timing_streamed(alfd_ctx_t ctx, double *streamed_bytes) {
  if (!ctx || !streamed_bytes) return ALFD_E_INVALID;
  for (int i = 0; i < ALFD_T_NCLASSES; ++i) streamed_bytes[i] = ctx->t_fbytes[i];
  return ALFD_OK;
}
int alfd_get_setup_seconds(alfd_ctx_t ctx, double *seconds) {
  if (!ctx || !seconds) return ALFD_E_INVALID;
  for (int i = 0; i < ALFD_SETUP_NPHASES; ++i) seconds[i] = ctx->setup_s[i];
  return ALFD_OK;
}

int alfd_get_timing(alfd_ctx_t ctx, double *ms, int64_t *launches, double *bytes) {
  if (!ctx) return ALFD_E_INVALID;
  flush_timers(ctx);
  for (int i = 0; i < ALFD_T_NCLASSES; ++i) {
    if (ms) ms[i] = ctx->t_ms[i];
    if (launches) launches[i] = ctx->t_launches[i];
    if (bytes) bytes[i] = ctx->t_bytes[i];
  }
  return ALFD_OK;
}

}  // extern "C"
