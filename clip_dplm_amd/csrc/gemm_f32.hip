// gemm_f32.hip — tiled exact-f32 GEMM on the f32 matrix pipe (v_mfma_f32_32x32x2_f32) for the ICNN transport maps
// (the reference forces f32 there: triple_flow/2_icnn_core.py:195) and for the materialised-logit gradients.
//
//   out[M,N] = alpha[0] * opA(A) · opB(B) (+ bias[N]) (+ addend_scale[0] * addend[M,N])
//   opA(A) = A[M,K] (row-major, lda)            or  A stored [K,M] (transA: contraction-major, e.g. dY for dW = dY^T X)
//   opB(B) = B[N,K]^T (nn.Linear weight layout)  or  B stored [K,N] (transB: input-gradient products dA = dZ · W)
//
// Numerics: each output element is a k-ordered chain of fmaf (the MFMA's f32 arithmetic, bit-for-bit), k ascending in
// steps of the instruction's K = 2 with the two lane halves interleaved — deterministic, no split-K, no atomics.
//
// Structure (HBM/L2-light, MFMA-bound by design: 64 FLOP/clk/SIMD is the f32 VECTOR rate, so the only way to beat a VALU
// kernel is to keep the matrix pipe issuing): 128 x 64 output tile, 256 threads = 4 waves, wave w owns rows
// [32w, 32w+32) x 64 columns = two 32x32 accumulators; BK = 16 per step; operands go global -> registers -> LDS
// (issue-early / write-late: the next step's loads are in flight under this step's 16 MFMAs), LDS rows padded to 20
// floats so that the float4 fragment reads of 32 consecutive rows spread over the banks; a transposed operand is
// transposed while it is written to LDS, so all four op combinations share one main loop.  24 KiB of LDS and < 128
// VGPRs: four workgroups per CU hide the L2 latency of the staging loads.
#include "common.h"

namespace {

constexpr int BN = 64, BK = 16, LDP = BK + 4;                 // LDS row = 20 floats (80 B, 16-B aligned)
constexpr int B_TILE = BN * LDP;                              // floats

struct GP {
  const float* A; long lda; const float* B; long ldb;
  float* out; long ldo;
  const float* bias; const float* addend; long ldadd; const float* addend_scale; const float* alpha;
  int M, N, K, transA, transB;
};

// stage one BK-deep slice of an operand tile [ROWS][BK] into registers (RPT float4 per thread)
//   !trans: memory [rows][K]: thread -> (row = idx / 4, kq = idx % 4): float4 along k
//    trans: memory [K][rows]: thread -> (k = idx / (ROWS/4), rq = idx % (ROWS/4)): float4 along rows
template <int ROWS>
struct Stager {
  static constexpr int RPT = ROWS * BK / 4 / 256;             // float4 per thread: 2 (A), 1 (B)
  f32x4 v[RPT];
  __device__ __forceinline__ void load(const float* base, long ld, int row0, int nrows, int k0, int K, bool trans, int tid) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int idx = tid + i * 256;
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (!trans) {
        const int r = idx >> 2, kq = (idx & 3) * 4;
        const int gr = row0 + r, gk = k0 + kq;
        if (gr < nrows) {
          const float* p = base + (long)gr * ld + gk;
          if (gk + 3 < K) t = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (gk + e < K) t[e] = p[e];
          }
        }
      } else {
        constexpr int RQ = ROWS / 4;
        const int k = idx / RQ, rq = (idx % RQ) * 4;
        const int gk = k0 + k, gr = row0 + rq;
        if (gk < K) {
          const float* p = base + (long)gk * ld + gr;
          if (gr + 3 < nrows) t = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (gr + e < nrows) t[e] = p[e];
          }
        }
      }
      v[i] = t;
    }
  }
  __device__ __forceinline__ void store(float* tile, bool trans, int tid) const {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int idx = tid + i * 256;
      if (!trans) {
        const int r = idx >> 2, kq = (idx & 3) * 4;
        *reinterpret_cast<f32x4*>(tile + r * LDP + kq) = v[i];
      } else {
        constexpr int RQ = ROWS / 4;
        const int k = idx / RQ, rq = (idx % RQ) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[(rq + e) * LDP + k] = v[i][e];
      }
    }
  }
};

// WM x WN waves (WM * WN = 4): tile = 32 WM rows x 64 columns; a wave owns rows [32 wm, +32) and 2 / WN column tiles.
// 4 x 1 (128 x 64) for large problems; 2 x 2 (64 x 64) when the larger tile would leave CUs without a second
// workgroup to overlap with (the transport maps' 4096 x 512 products: 256 -> 512 workgroups).
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GP p) {
  constexpr int BM = 32 * WM, A_TILE = BM * LDP, NT = 2 / WN;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, h = lane >> 5;
  // tile order: N fastest, so that the workgroups running together share A row panels (L2)
  const int ntn = (p.N + BN - 1) / BN;
  const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int K = p.K;

  Stager<BM> sa;
  Stager<BN> sb;
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  sa.load(p.A, p.lda, m0, p.M, 0, K, p.transA != 0, tid);
  sb.load(p.B, p.ldb, n0, p.N, 0, K, p.transB != 0, tid);
  sa.store(smem, p.transA != 0, tid);
  sb.store(smem + A_TILE, p.transB != 0, tid);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    float* at = smem + (ks & 1) * (A_TILE + B_TILE);
    float* bt = at + A_TILE;
    if (ks + 1 < nk) {                                        // issue early: lands under the MFMAs below
      sa.load(p.A, p.lda, m0, p.M, (ks + 1) * BK, K, p.transA != 0, tid);
      sb.load(p.B, p.ldb, n0, p.N, (ks + 1) * BK, K, p.transB != 0, tid);
    }
    // fragments: lane (row li, half h) holds k = 8 j + 4 h + e (e = 0..3) of its row for j = 0, 1: the SAME k map for
    // both operands, so MFMA number (j, e) contracts k = 8 j + 4 h' + e over its two lane halves h'
    f32x4 af[2], bf[NT][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      af[j] = *reinterpret_cast<const f32x4*>(at + (wm * 32 + li) * LDP + 8 * j + 4 * h);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        bf[t][j] = *reinterpret_cast<const f32x4*>(bt + ((wn * NT + t) * 32 + li) * LDP + 8 * j + 4 * h);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][e], bf[t][j][e], acc[t], 0, 0, 0);
    if (ks + 1 < nk) {                                        // write late, into the other buffer
      float* an = smem + ((ks + 1) & 1) * (A_TILE + B_TILE);
      sa.store(an, p.transA != 0, tid);
      sb.store(an + A_TILE, p.transB != 0, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
  const float asc = (p.addend && p.addend_scale) ? p.addend_scale[0] : 1.0f;
  const float alpha = p.alpha ? p.alpha[0] : 1.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int gn = n0 + (wn * NT + t) * 32 + li;
    if (gn >= p.N) continue;
    const float bv = p.bias ? p.bias[gn] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gm = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (gm < p.M) {
        float v = alpha * acc[t][r] + bv;
        if (p.addend) v += asc * p.addend[(long)gm * p.ldadd + gn];
        p.out[(long)gm * p.ldo + gn] = v;
      }
    }
  }
}

}  // namespace

extern "C" int clipk_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB,
                              int M, int N, int K, const float* alpha, const float* bias, const float* addend,
                              int64_t ldadd, const float* addend_scale, float* out, int64_t ldo, void* stream) {
  if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0) return CLIPK_ERR_BAD_ARG;
  if ((lda & 3) || (ldb & 3) || !aligned16(A) || !aligned16(B)) return CLIPK_ERR_UNSUPPORTED;   // float4 staging
  GP p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.out = out; p.ldo = ldo;
  p.bias = bias; p.addend = addend; p.ldadd = ldadd; p.addend_scale = addend_scale; p.alpha = alpha;
  p.M = M; p.N = N; p.K = K; p.transA = transA; p.transB = transB;
  const long ntn = (N + BN - 1) / BN;
  const long t128 = (long)((M + 127) / 128) * ntn, t64 = (long)((M + 63) / 64) * ntn;
  if (t64 > 0x7fffffffL) return CLIPK_ERR_UNSUPPORTED;
  if (t128 >= 1024)                                          // >= 4 workgroups per CU anyway: the larger tile
    hipLaunchKernelGGL((gemm_f32_kernel<4, 1>), dim3((unsigned)t128), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<2, 2>), dim3((unsigned)t64), dim3(256), 0, (hipStream_t)stream, p);
  return clipk_check_launch();
}
