// gemm_wgrad.hip — dW[N,K] (+)= dY[M,N]^T · X[M,K], db[N] (+)= colsum(dY): the weight gradient of
// every Linear on the path (autograd's mm(dY.t(), X) + dY.sum(0) in the reference's backward).
//
// gfx950 design:
//   * the contraction runs over the M tokens, which are the ROW index of both operands in memory, so
//     both MFMA fragments are "8 consecutive m at a fixed column": exactly what the CDNA4 transposed
//     LDS read ds_read_b64_tr_b16 delivers from a row-major tile.  Tiles are staged row-major
//     (coalesced 256-B rows) with a 288-byte LDS row stride that tools/lds_conflicts.py shows
//     conflict-free for the 4x16 transposed blocks; no transposed copies of activations ever exist;
//   * 128(n) x 128(k) f32 output tile per workgroup, 4 waves as 2x2, 16x16x32 bf16 MFMA;
//   * M is split across workgroups (enough splits to fill 256 CUs); each split writes an f32 slab and a
//     second kernel sums the slabs in a fixed order — deterministic, no float atomics;
//   * the bias gradient rides along as one extra MFMA per n-tile against an all-ones B fragment in
//     the k-tile-0 workgroups (no second pass over dY).
#include "common.h"

namespace {

constexpr int BN = 128, BKO = 128, BMS = 64, NTHREADS = 256;
constexpr int ROWB = 288;                               // LDS row stride in bytes (256 + 32)
constexpr int TILE_BYTES = BMS * ROWB;                  // 18 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;              // 72 KiB

struct WP {
  const unsigned short* dY; long lddy;
  const unsigned short* X; long ldx;
  float* slab;        // [splits][N][K]
  float* bslab;       // [splits][N] or null
  int M, N, K;
  int ntn, ntk, splits, m_per_split;
};

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row0, int colbyte) {
  // rows row0..row0+3 and row0+16..row0+19 of a 16-column block -> 8 k-values of one column per lane
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(tile + row0 * ROWB + colbyte));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(tile + (row0 + 16) * ROWB + colbyte));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const WP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
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

  // staging: chunk c = tid + 256*i -> row = (tid>>4) + 16*i, 16-byte column chunk = tid & 15
  const int srow = tid >> 4, scc = tid & 15;
  int ncol = n0 + scc * 8; ncol = ncol < p.N ? ncol : p.N - 8;     // N % 8 == 0: clamp to a valid chunk
  int kcol = k0 + scc * 8; kcol = kcol < p.K ? kcol : p.K - 8;
  const unsigned short* yg = p.dY + ncol;
  const unsigned short* xg = p.X + kcol;
  const int soff = srow * ROWB + scc * 16;

  u32x4 ry[4], rx[4];
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + srow + 16 * i;
      const bool ok = m < m_end;
      const long mm = ok ? m : m_beg;
      const u32x4 vy = *reinterpret_cast<const u32x4*>(yg + mm * p.lddy);
      const u32x4 vx = *reinterpret_cast<const u32x4*>(xg + mm * p.ldx);
      const u32x4 z = {0u, 0u, 0u, 0u};
      ry[i] = ok ? vy : z;
      rx[i] = ok ? vx : z;
    }
  };
  auto lstore = [&](int buf) {
    char* base = smem + buf * STAGE_BYTES + soff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(base + i * 16 * ROWB) = ry[i];
      *reinterpret_cast<u32x4*>(base + TILE_BYTES + i * 16 * ROWB) = rx[i];
    }
  };

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
  const int tcolb = 8 * (li & 3);                                // 4 bf16 = 8 bytes per p
  const int a_colb = (wn * 64) * 2 + tcolb;
  const int b_colb = (wk * 64) * 2 + tcolb;

  const int nsteps = (m_end - m_beg + BMS - 1) / BMS;
  if (nsteps > 0) {
    gload(m_beg);
    lstore(0);
  }
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    const bool more = (st + 1) < nsteps;
    if (more) gload(m_beg + (st + 1) * BMS);
    const char* ytile = smem + (st & 1) * STAGE_BYTES;
    const char* xtile = ytile + TILE_BYTES;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = tr_frag(ytile, ms * 32 + trow, a_colb + t * 32);
        bf[t] = tr_frag(xtile, ms * 32 + trow, b_colb + t * 32);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], ones, accb[i], 0, 0, 0);
      }
    }
    if (more) lstore((st + 1) & 1);
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

__global__ void wgrad_reduce_kernel(const float* slab, const float* bslab, int splits, int N, int K,
                                    float* dW, long lddw, float* dbias, int accumulate) {
  const long total4 = (long)N * K / 4;
  const int k4 = K >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += stride) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) a += reinterpret_cast<const f32x4*>(slab + (long)s * N * K)[i];
    const long n = i / k4; const int c = (int)(i - n * k4);
    f32x4* o = reinterpret_cast<f32x4*>(dW + n * lddw + 4 * c);
    *o = accumulate ? (*o + a) : a;
  }
  if (dbias) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < N; i += stride) {
      float a = 0.f;
      for (int s = 0; s < splits; ++s) a += bslab[(long)s * N + i];
      dbias[i] = accumulate ? dbias[i] + a : a;
    }
  }
}

struct Plan { int ntn, ntk, splits, mps; };
Plan make_plan(int M, int N, int K) {
  Plan pl;
  pl.ntn = (N + BN - 1) / BN; pl.ntk = (K + BKO - 1) / BKO;
  const int ntiles = pl.ntn * pl.ntk;
  int splits = (512 + ntiles - 1) / ntiles;
  const int max_splits = (M + 4 * BMS - 1) / (4 * BMS);      // at least 256 rows per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = (M + splits - 1) / splits;
  mps = (mps + BMS - 1) / BMS * BMS;
  pl.mps = mps;
  pl.splits = (M + mps - 1) / mps;
  return pl;
}

}  // namespace

extern "C" size_t clipk_gemm_wgrad_workspace(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const Plan pl = make_plan(M, N, K);
  return (size_t)pl.splits * ((size_t)N * K + N) * sizeof(float);
}

extern "C" int clipk_gemm_wgrad(const void* dY, int64_t lddy, const void* X, int64_t ldx,
                                float* dW, int64_t lddw, float* dbias, int M, int N, int K, int accumulate,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!dY || !X || !dW || !workspace || M <= 0 || N <= 0 || K <= 0) return CLIPK_ERR_BAD_ARG;
  if ((N & 7) || (K & 7) || (lddy & 7) || (ldx & 7) || (lddw & 3)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(dY) || !aligned16(X) || !aligned16(dW) || !aligned16(workspace)) return CLIPK_ERR_BAD_ARG;
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
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wgrad_kernel, dim3(pl.ntn * pl.ntk * pl.splits), dim3(NTHREADS), LDS_BYTES, st, p);
  int rc = clipk_check_launch();
  if (rc) return rc;
  long total4 = (long)N * K / 4;
  int blocks = (int)((total4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)p.slab, (const float*)p.bslab,
                     pl.splits, N, K, dW, (long)lddw, dbias, accumulate);
  return clipk_check_launch();
}
