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
  // skinny form only: S-way split of the contraction ACROSS workgroups (grid.y = S); the partial tiles
  // [S][N / 32][MB][32][32] live in the caller's workspace
  int S; float* parts;
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

// ---------------------------------------------------------------------------------------------------------------------
// Skinny form: M <= 64 rows against a large weight - every Linear (forward and input gradient) of the models that pool
// ONE position and are sliced to it before their encoders (RNARBPCLIPModel / ContrastiveModel: M = batch = 32 rows
// through 71.6 M f32 parameters, rna_clip_codes.ipynb:1925-1954).  Such a launch is a single pass over the weight: HBM-bound
// at 2 M FLOP per weight byte / 4, and the tiled kernel above walked its whole contraction in ONE workgroup per 64
// output columns (40 - 80 workgroups, 50 - 220 us for 6 - 26 MB).  Here:
//   * one workgroup of SIXTEEN waves per 32 output columns; the waves split the contraction 16 ways (runs of 32-index
//     units), so a CU has 16 independent load streams in flight, and N / 32 workgroups x 16 waves cover the chip's load
//     issue capacity;
//   * each lane feeds the f32 MFMA straight from its loads: A (the M rows) as consecutive float4 along the contraction
//     (whole 128-byte lines per lane and trip), B either the same way (weight stored [N][K]: forward) or as dwords of
//     consecutive weight rows (weight stored [K][N]: input gradient, 128-byte row segments per half wave) - no LDS
//     staging, no barrier in the main loop;
//   * the 16 partial tiles meet in LDS and are summed in wave order (deterministic; no atomics, no workspace), then the
//     same alpha / bias / addend epilogue as above, 128-byte rows per store.
// v_mfma_f32_32x32x2_f32 operands: lane (i = lane & 31, h = lane >> 5) supplies A[i][k = h] and B[k = h][j = i].  The
// pairing of contraction indices into k = 0 / 1 is free as long as A and B agree: lane half h takes c0 + 4 h + t.
template <int MB, bool TB>
__global__ __launch_bounds__(1024) void gemm_f32_skinny_kernel(const GP p) {
  extern __shared__ float red[];                             // [16 waves][MB][32 x 32]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int j0 = blockIdx.x * 32, gj = j0 + li;
  const bool jok = gj < p.N;
  // the contraction in UNITS of 4 U indices = what one lane reads contiguously per trip (U float4: a whole 128-byte line
  // of its row at U = 8); each wave owns a contiguous run of units, a trip feeds two of them (one per lane half)
  constexpr int U = MB == 1 ? 8 : 4;
  constexpr int UNIT = 4 * U;
  const int nunit = (p.K + UNIT - 1) / UNIT;
  const int S = p.S, sid = blockIdx.y;
  const int per_wg = (nunit + S - 1) / S;                      // this workgroup's run of units ...
  const int g0 = sid * per_wg, g1 = (g0 + per_wg < nunit) ? g0 + per_wg : nunit;
  const int per = (g1 - g0 + 15) >> 4;                         // ... split over its 16 waves
  const int u0 = g0 + wid * per, u1 = (u0 + per < g1) ? u0 + per : g1;
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
  const float* arow[MB];
  bool mok[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = mb * 32 + li;
    mok[mb] = m < p.M;
    arow[mb] = p.A + (long)(mok[mb] ? m : 0) * p.lda;
  }
  const float* bcol = TB ? p.B + (jok ? gj : 0) : p.B + (long)(jok ? gj : 0) * p.ldb;
  // Every load of a trip is issued (unconditionally, from clamped addresses) before the first MFMA consumes one; what lies
  // outside the matrix is zeroed by a bit mask (a select would come back as a branch around the load, and the loop as one
  // load pair per wait).  A lane's U float4 are CONSECUTIVE: with one float4 per row and trip (first version) every
  // 128-byte line of the weight crossed L2 -> L1 four times (16 waves x 8 trips in flight do not fit the 32 KiB L1).
  const unsigned jmask = jok ? 0xffffffffu : 0u;
  for (int ub = u0; ub < u1; ub += 2) {
    const int unit = ub + h;
    const bool uok = unit < u1;
    const int cb = UNIT * unit;
    f32x4 a[U][MB], b[U];
    unsigned cm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = cb + 4 * u;                              // this lane's contraction indices c .. c + 3
      const bool cok = uok && c < p.K;                       // (K % 4 == 0: checked by the host)
      const int cc = cok ? c : 0;
      cm[u] = cok ? 0xffffffffu : 0u;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) a[u][mb] = *reinterpret_cast<const f32x4*>(arow[mb] + cc);
      if constexpr (TB) {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[u][t] = bcol[(long)(cc + t) * p.ldb];
      } else {
        b[u] = *reinterpret_cast<const f32x4*>(bcol + cc);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float bv = __uint_as_float(__float_as_uint(b[u][t]) & (cm[u] & jmask));
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float av = __uint_as_float(__float_as_uint(a[u][mb][t]) & (mok[mb] ? cm[u] : 0u));
          acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mb], 0, 0, 0);
        }
      }
    }
  }
  // C / D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      red[((wid * MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + li] = acc[mb][r];
  __syncthreads();
  const float asc = (p.addend && p.addend_scale) ? p.addend_scale[0] : 1.0f;
  const float alpha = p.alpha ? p.alpha[0] : 1.0f;
  float vsum[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int e = tid + i * 1024;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += red[w * MB * 1024 + e];
    vsum[i] = v;
  }
  if (S > 1) {
    // cross-workgroup split: park this workgroup's partial tile; gemm_f32_skinny_reduce_kernel adds the S of them in split
    // order and runs the epilogue.  (One kernel with a "last workgroup to arrive reduces" counter was built first: its
    // agent-scope release fence is an L2 write-back on this multi-XCD part - 320 workgroups x one write-back each took
    // a 22 us launch to 142 us.  The kernel boundary is the cheap fence.)
    const long tile = (long)MB * 1024;
    float* mine = p.parts + ((long)sid * gridDim.x + blockIdx.x) * tile;
#pragma unroll
    for (int i = 0; i < MB; ++i) mine[tid + i * 1024] = vsum[i];
    return;
  }
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int e = tid + i * 1024;
    const int m = e >> 5, j = j0 + (e & 31);
    if (m < p.M && j < p.N) {
      float v = alpha * vsum[i] + (p.bias ? p.bias[j] : 0.f);
      if (p.addend) v += asc * p.addend[(long)m * p.ldadd + j];
      p.out[(long)m * p.ldo + j] = v;
    }
  }
}

// second pass of a split launch: out = epilogue(sum over the S partial tiles, in split order)
template <int MB>
__global__ __launch_bounds__(1024) void gemm_f32_skinny_reduce_kernel(const GP p, int njb) {
  const int tid = threadIdx.x, j0 = blockIdx.x * 32;
  const long tile = (long)MB * 1024;
  const float asc = (p.addend && p.addend_scale) ? p.addend_scale[0] : 1.0f;
  const float alpha = p.alpha ? p.alpha[0] : 1.0f;
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int e = tid + i * 1024;
    // all S <= 8 partial tiles requested at once (a run-time loop of dependent adds was S memory round trips); summed in
    // split order as before
    float part[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) part[s2] = p.parts[((long)(s2 < p.S ? s2 : 0) * njb + blockIdx.x) * tile + e];
    const int m = e >> 5, j = j0 + (e & 31);
    const bool ok = m < p.M && j < p.N;
    // (bias / addend requested with the partial tiles: clamped addresses, used under `ok`)
    const float bv = p.bias ? p.bias[ok ? j : 0] : 0.f;
    const float av = p.addend ? p.addend[ok ? (long)m * p.ldadd + j : 0] : 0.f;
    float v = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) v += s2 < p.S ? part[s2] : 0.f;
    if (ok) {
      v = alpha * v + bv;
      if (p.addend) v += asc * av;
      p.out[(long)m * p.ldo + j] = v;
    }
  }
}

// split of the contraction across workgroups: enough workgroups to put every CU to work (an f32-MFMA-fed CU takes in
// 16 B / clk of weight at M = 32: the chip's 8 TB/s need ~all 256 of them), never less than one 32-index unit per wave
static int skinny_splits(int N, int K, int MB) {
  // one 1024-thread workgroup fits a CU (101 VGPRs x 16 waves), so S * njb workgroups run in ceil(S njb / 256) rounds
  // of 1 / S of the contraction each: take the S <= 8 with the least rounds / S (ties: the smaller S - fewer partial
  // tiles), never less than one 32-index unit per wave
  const int njb = (N + 31) / 32, nunit = (K + (MB == 1 ? 32 : 16) - 1) / (MB == 1 ? 32 : 16);
  const int forced = clipk_opt_get(OPT_GEMM_F32_SPLITS);
  if (forced >= 1 && forced <= 8) return (nunit / forced >= 1) ? forced : 1;
  int best = 1;
  double cost = (double)((njb + 255) / 256);
  for (int S = 2; S <= 8; ++S) {
    if (nunit / S < 16) break;
    const double c = (double)((S * njb + 255) / 256) / S;
    if (c < cost - 1e-9) { cost = c; best = S; }
  }
  return best;
}
static size_t skinny_ws_bytes(int M, int N, int K) {
  const int MB = M <= 32 ? 1 : 2, njb = (N + 31) / 32, S = skinny_splits(N, K, MB);
  if (S <= 1) return 0;
  return (size_t)S * njb * MB * 1024 * sizeof(float);
}
static bool skinny_applies(int transA, int transB, int M, int K, int64_t ldb) {
  return !transA && M <= 64 && K >= 256 && (K & 3) == 0 && (transB || (ldb & 3) == 0);
}

template <int MB, bool TB>
int launch_skinny(const GP& p, hipStream_t st) {
  static std::atomic<uint64_t> once{0};
  const size_t lds = (size_t)16 * MB * 1024 * sizeof(float);
  clipk_once_per_device(once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_skinny_kernel<MB, TB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int njb = (p.N + 31) / 32;
  hipLaunchKernelGGL((gemm_f32_skinny_kernel<MB, TB>), dim3((unsigned)njb, (unsigned)p.S), dim3(1024), lds, st, p);
  if (p.S > 1) hipLaunchKernelGGL((gemm_f32_skinny_reduce_kernel<MB>), dim3((unsigned)njb), dim3(1024), 0, st, p, njb);
  return clipk_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Parameter gradients of an exact-f32 Linear whose input has FEW rows (M <= 64: the position-0 models, M = batch = 32):
//   dW[N, K] (+)= dY[M, N]^T X[M, K]        db[N] (+)= sum over rows of dY
// Such a product is ONE PASS OVER dW - 2 M FLOP per 4 (8 when accumulating) bytes of dW - and the tiled kernel above gave
// every 64 x 64 tile a staging round through LDS, two barriers and a two-step main loop: 16 us for 6.5 - 26 MB, plus a
// column-reduce launch for db (53 + 52 launches per step of the notebook model).  Here a wave owns a 32 x 32 block of dW:
// each lane loads its dY / X values straight into the f32 MFMA's operand registers (for one m: 32 lanes = 128 contiguous
// bytes of the row), ceil(M / 2) MFMAs, read-add-write of 128-byte row segments.  No LDS, no barrier; every load of a wave
// is in flight before its first MFMA.  The waves of column block 0 also sum their dY operands: db comes out of the same
// launch.  Contraction order = the tiled kernel's (m = 16 ks + 8 j + 4 h + e, steps (ks, j, e) ascending): dW is
// bit-identical to clipk_gemm_f32(transA, transB) on the same operands.
struct WGP {
  const float* dY; long lddy; const float* X; long ldx;
  float* dW; long lddw; float* db;
  int M, N, K, accumulate;
};

template <int MT>                                            // MT = ceil(M / 32): 16 MT MFMAs per block
__global__ __launch_bounds__(256) void wgrad_f32_small_kernel(const WGP p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 32, k0 = (blockIdx.x * 4 + wid) * 32;
  if (k0 >= p.K) return;                                      // (wave-uniform; no barrier in this kernel)
  const int gn = n0 + li, gk = k0 + li;
  const bool n_ok = gn < p.N, k_ok = gk < p.K;
  // clamped addresses + masks instead of branches around the loads: every load is issued before the first wait
  const float* ap = p.dY + (n_ok ? gn : 0);
  const float* bp = p.X + (k_ok ? gk : 0);
  float a[16 * MT], b[16 * MT];
#pragma unroll
  for (int s = 0; s < 16 * MT; ++s) {
    const int m = (s >> 3) * 16 + ((s >> 2) & 1) * 8 + (s & 3) + 4 * h;
    const int mc = m < p.M ? m : 0;
    a[s] = ap[(long)mc * p.lddy];
    b[s] = bp[(long)mc * p.ldx];
  }
  f32x16 old;
  if (p.accumulate) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gr = n0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      old[r] = p.dW[(long)(gr < p.N ? gr : 0) * p.lddw + (k_ok ? gk : 0)];     // (clamped; used under the store's mask)
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
#pragma unroll
  for (int s = 0; s < 16 * MT; ++s) {
    const int m = (s >> 3) * 16 + ((s >> 2) & 1) * 8 + (s & 3) + 4 * h;
    const float av = (m < p.M && n_ok) ? a[s] : 0.f;
    const float bv = (m < p.M && k_ok) ? b[s] : 0.f;
    bsum += av;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gr = n0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (gr < p.N && k_ok) {
      float v = acc[r];
      if (p.accumulate) v += old[r];
      p.dW[(long)gr * p.lddw + gk] = v;
    }
  }
  if (p.db != nullptr && k0 == 0) {                           // (wave-uniform) db[n] = sum_m dY[m][n]: both lane halves
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0 && n_ok) p.db[gn] = p.accumulate ? p.db[gn] + bsum : bsum;
  }
}

}  // namespace

extern "C" size_t clipk_gemm_f32_workspace(int M, int N, int K, int transA, int transB) {
  if (M <= 0 || N <= 0 || K <= 0 || !skinny_applies(transA, transB, M, K, 4)) return 0;
  return skinny_ws_bytes(M, N, K);
}

extern "C" int clipk_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB,
                              int M, int N, int K, const float* alpha, const float* bias, const float* addend,
                              int64_t ldadd, const float* addend_scale, float* out, int64_t ldo,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0) return CLIPK_ERR_BAD_ARG;
  if ((lda & 3) || (ldb & 3) || !aligned16(A) || !aligned16(B)) return CLIPK_ERR_UNSUPPORTED;   // float4 staging
  GP p;
  p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.out = out; p.ldo = ldo;
  p.bias = bias; p.addend = addend; p.ldadd = ldadd; p.addend_scale = addend_scale; p.alpha = alpha;
  p.M = M; p.N = N; p.K = K; p.transA = transA; p.transB = transB;
  p.S = 1; p.parts = nullptr;
  // a few rows against a large weight: one bandwidth-bound pass over B (see gemm_f32_skinny_kernel)
  if (skinny_applies(transA, transB, M, K, ldb)) {
    hipStream_t st = (hipStream_t)stream;
    const size_t need = skinny_ws_bytes(M, N, K);
    if (need && workspace && workspace_bytes >= need && aligned16(workspace)) {   // else: one workgroup per column block
      p.S = skinny_splits(N, K, M <= 32 ? 1 : 2);
      p.parts = reinterpret_cast<float*>(workspace);
    }
    if (M <= 32) return transB ? launch_skinny<1, true>(p, st) : launch_skinny<1, false>(p, st);
    return transB ? launch_skinny<2, true>(p, st) : launch_skinny<2, false>(p, st);
  }
  const long ntn = (N + BN - 1) / BN;
  const long t128 = (long)((M + 127) / 128) * ntn, t64 = (long)((M + 63) / 64) * ntn;
  if (t64 > 0x7fffffffL) return CLIPK_ERR_UNSUPPORTED;
  if (t128 >= 1024)                                          // >= 4 workgroups per CU anyway: the larger tile
    hipLaunchKernelGGL((gemm_f32_kernel<4, 1>), dim3((unsigned)t128), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<2, 2>), dim3((unsigned)t64), dim3(256), 0, (hipStream_t)stream, p);
  return clipk_check_launch();
}

extern "C" int clipk_colsum_f32(const float* x, int rows, int cols, float* out, int accumulate, void* stream);
extern "C" int clipk_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB,
                              int M, int N, int K, const float* alpha, const float* bias, const float* addend,
                              int64_t ldadd, const float* addend_scale, float* out, int64_t ldo,
                              void* workspace, size_t workspace_bytes, void* stream);

extern "C" int clipk_gemm_wgrad_f32(const float* dY, int64_t lddy, const float* X, int64_t ldx, float* dW, int64_t lddw,
                                    float* dbias, int M, int N, int K, int accumulate, void* stream) {
  if (!dY || !X || (!dW && !dbias) || M <= 0 || N <= 0 || K <= 0) return CLIPK_ERR_BAD_ARG;
  if (dW && M <= 64) {
    WGP p;
    p.dY = dY; p.lddy = lddy; p.X = X; p.ldx = ldx; p.dW = dW; p.lddw = lddw; p.db = dbias;
    p.M = M; p.N = N; p.K = K; p.accumulate = accumulate;
    const dim3 grid((unsigned)((K + 127) / 128), (unsigned)((N + 31) / 32));
    if (grid.y > 65535u) return CLIPK_ERR_UNSUPPORTED;
    if (M <= 32) hipLaunchKernelGGL((wgrad_f32_small_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((wgrad_f32_small_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return clipk_check_launch();
  }
  // many rows: the tiled kernel (contraction-major operands) and the column reduce
  if (dW) {
    const int rc = clipk_gemm_f32(dY, lddy, 1, X, ldx, 1, N, K, M, nullptr, nullptr, accumulate ? dW : nullptr, lddw, nullptr,
                                  dW, lddw, nullptr, 0, stream);
    if (rc != CLIPK_OK) return rc;
  }
  if (dbias) {
    if (lddy != N) return CLIPK_ERR_UNSUPPORTED;             // (the column reduce takes dense rows)
    return clipk_colsum_f32(dY, M, N, dbias, accumulate, stream);
  }
  return CLIPK_OK;
}

