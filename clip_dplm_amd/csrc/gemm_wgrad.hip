// gemm_wgrad.hip — dW[N,K] (+)= dY[M,N]^T · X[M,K], db[N] (+)= colsum(dY): the weight gradient of
// every Linear on the path (autograd's mm(dY.t(), X) + dY.sum(0) in the reference's backward).
//
// gfx950 design:
//   * the contraction runs over the M tokens, which are the ROW index of both operands in memory, so
//     both MFMA fragments are "8 consecutive m at a fixed column": exactly what the CDNA4 transposed
//     LDS read ds_read_b64_tr_b16 delivers from a row-major tile.  No transposed copy of an activation ever
//     exists in HBM;
//   * tiles are [64 m][128 cols] bf16 (256-B rows) filled by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 4 rows
//     per wave-instruction).  An LDS-DMA writes lane-linear bytes, so rows cannot be padded; instead the
//     32-byte column segments are XOR-swizzled with (row & 7) — applied to the per-lane SOURCE address and to
//     the transposed-read address — which makes every 4x16 transposed block read conflict-free
//     (tools/lds_conflicts.py);
//   * one 32 KiB step buffer, <= 128 VGPRs: 4 workgroups per CU overlap each other's load and MFMA phases;
//   * 128(n) x 128(k) f32 output tile per workgroup, 4 waves as 2x2, 16x16x32 bf16 MFMA;
//   * M is split across workgroups; each split writes an f32 slab and a second kernel sums the slabs in a
//     fixed order — deterministic, no float atomics;
//   * the bias gradient rides along as one extra MFMA per n-tile against an all-ones B fragment in the
//     k-tile-0 workgroups (no second pass over dY).
#include "common.h"
#include <stdlib.h>

// gemm_wgrad_v3.hip: 256 x 256 output tiles, 8 waves, phase-interleaved (large problems)
struct clipk_wgrad_v3_args {
  const unsigned short* dY; long lddy;
  const unsigned short* X; long ldx;
  float* slab; float* bslab;
  int M, N, K;
  int ntn, ntk, splits, m_per_split;
};
extern "C" void clipk_wgrad_v3_plan(int M, int N, int K, int* ntn, int* ntk, int* splits, int* mps);
extern "C" int clipk_wgrad_v3_launch(const clipk_wgrad_v3_args* a, void* stream);

namespace {

constexpr int BN = 128, BKO = 128, BMS = 64, NTHREADS = 256;
constexpr int ROWB = 256;                               // LDS row bytes (128 bf16, unpadded: LDS-DMA image)
constexpr int TILE_BYTES = BMS * ROWB;                  // 16 KiB per operand tile
constexpr int LDS_BYTES = 2 * TILE_BYTES;               // 32 KiB

struct WP {
  const unsigned short* dY; long lddy;
  const unsigned short* X; long ldx;
  float* slab;        // [splits][N][K]
  float* bslab;       // [splits][N] or null
  int M, N, K;
  int ntn, ntk, splits, m_per_split;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

// logical (row, 32-byte segment) -> byte offset in the swizzled tile
__device__ __forceinline__ int seg_off(int row, int seg) { return row * ROWB + ((seg ^ (row & 7)) << 5); }

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row0, int seg, int pbyte) {
  // rows row0 (+16) supplied by this lane, 16-column block `seg`; returns 8 m-values of one column
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(tile + seg_off(row0, seg) + pbyte));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(tile + seg_off(row0 + 16, seg) + pbyte));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(gptr, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(NTHREADS, 3) void wgrad_kernel(const WP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid >> 1, wk = wid & 1;
  const int ntiles = p.ntn * p.ntk;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid / ntiles;
  const int tile = bid - split * ntiles;
  const int tn = tile / p.ntk, tk = tile - tn * p.ntk;
  const int n0 = tn * BN, k0 = tk * BKO;
  const int m_beg = split * p.m_per_split;
  int m_end = m_beg + p.m_per_split; m_end = m_end < p.M ? m_end : p.M;
  const bool do_bias = (p.bslab != nullptr) && (tk == 0) && (wk == 0);

  // ---- LDS-DMA assignment: wave w, piece i (0..3) fills rows 4*(4w+i) .. +3 of each operand tile.
  // lane -> (row in piece = lane>>4, physical 16-B slot = lane&15); the slot's 32-B segment is (slot>>1)
  // and holds logical segment (slot>>1) ^ (row&7).
  const int prow = lane >> 4, pslot = lane & 15;
  const int row0 = 16 * wid + prow;                               // piece i adds 4*i rows
  int ncol[4], kcol[4];                                           // swizzled source columns of this lane's chunks
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + 4 * i;
    const int lseg = (pslot >> 1) ^ (row & 7);
    const int col = lseg * 16 + (pslot & 1) * 8;                   // logical column (elements) of this 16-B chunk
    int nc = n0 + col; nc = nc < p.N ? nc : p.N - 8;               // N, K % 8 == 0: clamp to a valid chunk
    int kc = k0 + col; kc = kc < p.K ? kc : p.K - 8;
    ncol[i] = nc;
    kcol[i] = kc;
  }

  f32x4 acc[4][4], accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;          // bf16 1.0

  // transposed-read lane addressing: group g = lane>>4, in-group i = lane&15, q = i>>2, p = i&3
  const int g = lane >> 4, li = lane & 15;
  const int trow = 4 * g + (li >> 2);
  const int pbyte = 8 * (li & 3);
  char* ytile = smem;
  char* xtile = smem + TILE_BYTES;

  const int nsteps = (m_end - m_beg + BMS - 1) / BMS;
  for (int st = 0; st < nsteps; ++st) {
    const int ms0 = m_beg + st * BMS;
    const unsigned short* ybase = p.dY + (long)ms0 * p.lddy;      // wave-uniform (SGPR) base of this step
    const unsigned short* xbase = p.X + (long)ms0 * p.ldx;
    const int ldy = (int)p.lddy, ldxx = (int)p.ldx;
    if (ms0 + BMS <= m_end) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        char* dst = smem + (4 * wid + i) * 1024;
        glds16(ybase + ((row0 + 4 * i) * ldy + ncol[i]), dst);       // SGPR base + 32-bit lane offset
        glds16(xbase + ((row0 + 4 * i) * ldxx + kcol[i]), dst + TILE_BYTES);
      }
    } else {
      // ragged last step of the last split: rows past M must contribute zeros -> register staging + select
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = ms0 + row0 + 4 * i < m_end;
        const int rr = ok ? row0 + 4 * i : 0;                        // row ms0 itself is always valid
        u32x4 vy = *reinterpret_cast<const u32x4*>(ybase + (rr * ldy + ncol[i]));
        u32x4 vx = *reinterpret_cast<const u32x4*>(xbase + (rr * ldxx + kcol[i]));
        const u32x4 z = {0u, 0u, 0u, 0u};
        char* dst = smem + (4 * wid + i) * 1024 + lane * 16;
        *reinterpret_cast<u32x4*>(dst) = ok ? vy : z;
        *reinterpret_cast<u32x4*>(dst + TILE_BYTES) = ok ? vx : z;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      bf16x8 af[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = tr_frag(ytile, ms * 32 + trow, wn * 4 + t, pbyte);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 bfj = tr_frag(xtile, ms * 32 + trow, wk * 4 + j, pbyte);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfj, acc[i][j], 0, 0, 0);
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- store the f32 partial tile: D rows = n (4 per lane), cols = k (lane&15)
  float* slab = p.slab + (long)split * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * g + r;
        if (n < p.N && k < p.K) slab[(long)n * p.K + k] = acc[i][j][r];
      }
    }
  if (do_bias && li == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + 4 * g + r;
        if (n < p.N) p.bslab[(long)split * p.N + n] = accb[i][r];
      }
  }
}

// (il_hd, il_rows): dY's first il_rows columns are in pair-interleaved head order (common.h il_src): gradient row n of the
// slabs belongs to row il_src(n) of dW / dbias (identity when il_rows == 0)
__global__ void wgrad_reduce_kernel(const float* slab, const float* bslab, int splits, int bsplits, int N, int K,
                                    float* dW, long lddw, float* dbias, int accumulate, int il_hd, int il_rows) {
  const long total4 = (long)N * K / 4;
  const int k4 = K >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += stride) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    const long sstride = (long)N * K / 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(slab) + i;
    int s = 0;
    for (; s + 8 <= splits; s += 8) {                        // eight loads in flight (16 - 64 slabs: the launch is a chain of
      f32x4 v[8];                                            // memory round trips); same summation order as the 4-wide step
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[(long)(s + e) * sstride];
      a += (v[0] + v[1]) + (v[2] + v[3]);
      a += (v[4] + v[5]) + (v[6] + v[7]);
    }
    for (; s + 4 <= splits; s += 4) {                        // four loads in flight; fixed summation order
      const f32x4 v0 = src[(long)s * sstride], v1 = src[(long)(s + 1) * sstride];
      const f32x4 v2 = src[(long)(s + 2) * sstride], v3 = src[(long)(s + 3) * sstride];
      a += (v0 + v1) + (v2 + v3);
    }
    for (; s < splits; ++s) a += src[(long)s * sstride];
    const long n = i / k4; const int c = (int)(i - n * k4);
    f32x4* o = reinterpret_cast<f32x4*>(dW + (long)il_src((int)n, il_hd, il_rows) * lddw + 4 * c);
    *o = accumulate ? (*o + a) : a;
  }
  if (dbias) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += stride) {
      float a = 0.f;
      int s = 0;
      for (; s + 8 <= bsplits; s += 8) {                     // eight loads in flight; fixed summation order
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = bslab[(long)(s + e) * N + i];
        a += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
      }
      for (; s < bsplits; ++s) a += bslab[(long)s * N + i];
      const int o = il_src((int)i, il_hd, il_rows);
      dbias[o] = accumulate ? dbias[o] + a : a;
    }
  }
}

struct Plan { int ntn, ntk, splits, mps; };
Plan make_plan(int M, int N, int K) {
  Plan pl;
  pl.ntn = (N + BN - 1) / BN; pl.ntk = (K + BKO - 1) / BKO;
  const int ntiles = pl.ntn * pl.ntk;
  int splits = (1024 + ntiles - 1) / ntiles;                 // ~4 workgroups per CU resident
  const int max_splits = (M + 8 * BMS - 1) / (8 * BMS);      // at least 512 rows per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = (M + splits - 1) / splits;
  mps = (mps + BMS - 1) / BMS * BMS;
  pl.mps = mps;
  pl.splits = (M + mps - 1) / mps;
  return pl;
}

// large problems go to the 256 x 256 phase-interleaved kernel; option wgrad_kernel = 2 / 3 forces the choice
bool use_v3(int M, int N, int K) {
  const int mode = clipk_opt_get(OPT_WGRAD_KERNEL);
  if (mode == 2) return false;
  if (mode == 3 || mode == 4) return M >= 1024;
  return M >= 16384 && N >= 128 && K >= 128;
}

}  // namespace

extern "C" size_t clipk_gemm_wgrad_workspace(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan pl = make_plan(M, N, K);
  int ntn, ntk, splits, mps;
  clipk_wgrad_v3_plan(M, N, K, &ntn, &ntk, &splits, &mps);
  // either kernel may be chosen at launch time; the 256 x 256 kernel keeps one bias partial per (split, k-tile)
  const size_t a = (size_t)pl.splits * ((size_t)N * K + N), b = (size_t)splits * ((size_t)N * K + (size_t)ntk * N);
  return (a > b ? a : b) * sizeof(float);
}

extern "C" int clipk_gemm_wgrad(const void* dY, int64_t lddy, const void* X, int64_t ldx,
                                float* dW, int64_t lddw, float* dbias, int M, int N, int K, int accumulate,
                                int il_hd, int il_rows, void* workspace, size_t workspace_bytes, void* stream) {
  if (!dY || !X || !dW || !workspace || M <= 0 || N <= 0 || K <= 0) return CLIPK_ERR_BAD_ARG;
  if (il_rows < 0 || il_rows > N || (il_rows > 0 && (il_hd < 2 || (il_hd & 1) || il_rows % il_hd))) return CLIPK_ERR_BAD_ARG;
  if (il_rows == 0) il_hd = 2;
  if ((N & 7) || (K & 7) || (lddy & 7) || (ldx & 7) || (lddw & 3)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(dY) || !aligned16(X) || !aligned16(dW) || !aligned16(workspace)) return CLIPK_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (use_v3(M, N, K)) {
    clipk_wgrad_v3_args a;
    clipk_wgrad_v3_plan(M, N, K, &a.ntn, &a.ntk, &a.splits, &a.m_per_split);
    if (workspace_bytes < (size_t)a.splits * ((size_t)N * K + (size_t)a.ntk * N) * sizeof(float)) return CLIPK_ERR_BAD_ARG;
    a.dY = (const unsigned short*)dY; a.lddy = lddy;
    a.X = (const unsigned short*)X; a.ldx = ldx;
    a.slab = (float*)workspace;
    a.bslab = dbias ? (float*)workspace + (size_t)a.splits * N * K : nullptr;
    a.M = M; a.N = N; a.K = K;
    int rc = clipk_wgrad_v3_launch(&a, stream);
    if (rc) return rc;
    long total4 = (long)N * K / 4;
    int blocks = (int)((total4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)a.slab, (const float*)a.bslab,
                       a.splits, a.splits * a.ntk, N, K, dW, (long)lddw, dbias, accumulate, il_hd, il_rows);
    return clipk_check_launch();
  }
  const Plan pl = make_plan(M, N, K);
  const size_t need = (size_t)pl.splits * ((size_t)N * K + N) * sizeof(float);
  if (workspace_bytes < need) return CLIPK_ERR_BAD_ARG;
  WP p;
  p.dY = (const unsigned short*)dY; p.lddy = lddy;
  p.X = (const unsigned short*)X; p.ldx = ldx;
  p.slab = (float*)workspace;
  p.bslab = dbias ? (float*)workspace + (size_t)pl.splits * N * K : nullptr;
  p.M = M; p.N = N; p.K = K;
  p.ntn = pl.ntn; p.ntk = pl.ntk; p.splits = pl.splits; p.m_per_split = pl.mps;
  hipLaunchKernelGGL(wgrad_kernel, dim3(pl.ntn * pl.ntk * pl.splits), dim3(NTHREADS), LDS_BYTES, st, p);
  int rc = clipk_check_launch();
  if (rc) return rc;
  long total4 = (long)N * K / 4;
  int blocks = (int)((total4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)p.slab, (const float*)p.bslab,
                     pl.splits, pl.splits, N, K, dW, (long)lddw, dbias, accumulate, il_hd, il_rows);
  return clipk_check_launch();
}
