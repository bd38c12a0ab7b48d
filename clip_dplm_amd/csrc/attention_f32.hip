// attention_f32.hip — exact-f32 multi-head self-attention, forward and backward, for the models whose caller is fp32.
//
// Reference arithmetic replaced: nn.MultiheadAttention inside nn.TransformerEncoderLayer as the notebook models call it
// WITHOUT autocast (current/rna_clip_codes.ipynb:1911-1954, current/tf_clip_codes (1).ipynb:13056-13072): f32 q / k / v,
// f32 scores, f32 softmax, f32 P·V.  Those models pool ONE position (`enc[:, 0]`, ipynb:1948-1949) of a stack whose
// attention mixes the batch axis, so after the position-0 slice their attention problem is tiny - one "sequence" of B
// samples per head - and what matters is that no operand is rounded to bf16 (a bf16-rounded weight perturbs the model
// identically for every sample and no batch size averages that away: DESIGN.md §3.3).
//
// gfx950 design: the f32 matrix pipe (v_mfma_f32_32x32x2_f32) and the packed-f32 vector pipe have the same 157 TFLOP/s
// peak on this part, so these kernels use plain VALU fmas in a fixed order (bitwise reproducible, no atomics) and spend
// their care on the LDS access pattern instead of on MFMA fragment layouts:
//   * scores: lane = key.  A 64-key block of K (and V) sits in LDS as rows of S = 4 * (odd) floats, so that the 64
//     lanes' ds_read_b128 of their own row hit disjoint bank groups (16 lanes x 4 banks per pass); the query rows are
//     read as wave-uniform (broadcast) b128.  Four query rows per wave share every K read: 5 LDS reads per 16 fmas;
//   * online softmax per query row with wave-wide max / sum (DPP butterflies), statistics in f32, accurate expf;
//   * P·V (and every "sum over rows of an LDS tile" product of the backward): lane = head-dim column d, the
//     probabilities of the wave's four query rows come back from LDS as one broadcast b128 per key, the V row is read
//     by consecutive lanes (conflict-free);
//   * backward = a dQ kernel (16 queries per workgroup, sweeping keys; also writes delta = rowsum(dO * O)) and a dK/dV
//     kernel (16 keys per workgroup, sweeping queries), each recomputing P from q, k and the saved log-sum-exp: 7
//     products instead of 5, no float atomics, one fixed summation order;
//   * dropout on the probabilities uses the SAME counter-based mask and element index as the bf16 kernels
//     (attention.hip attn_drop), so a model switched between the two arithmetics drops the same elements.
// Head dims up to 192 (any value, no multiple-of-8 rule: the notebook's 120 / 8 = 15 runs unpadded).
#include "common.h"
#include <math.h>

namespace {

struct AF {
  const float* qkv; const uint8_t* key_mask; float* out; float* lse;
  const float* dout; float* delta; float* dqkv;
  int B, L, H, D;
  float scale;
  unsigned drop_thr, drop_seed; float drop_scale;
  const unsigned* drop_epoch;       // common.h drop_seed_eff (nullptr outside a captured step)
};

constexpr int KB = 64;        // keys (dQ / forward) or queries (dK/dV) per LDS block = one per lane
constexpr int NW = 4;         // waves per workgroup; each owns RW rows (template parameter: 4, or 1 for short sequences -
                              // the sliced notebook models attend over L = batch = 32 samples: with 4 rows per wave that
                              // is 2 workgroups per head, 16 on the whole chip, and the launch is pure latency)

__host__ __device__ inline int row_stride(int D) { return 4 * ((((D + 3) >> 2)) | 1); }   // floats; S / 4 odd

__device__ __forceinline__ float drop_at(const AF& p, unsigned dseed, long qrow, int h, int key) {
  return drop_mul(dseed, ((unsigned long long)qrow * p.H + h) * (unsigned long long)p.L + key, p.drop_thr, p.drop_scale);
}

// rows [r0, r0 + nrows) of one head-slice of a [B*L, ld] f32 tensor -> LDS rows of stride S, columns >= D and rows past
// the sequence zero-filled; consecutive threads on consecutive columns (coalesced 4-byte loads: head slices of odd head
// dims are not 16-byte aligned).  mul: scale applied on the way (q * q_scale).
__device__ __forceinline__ void stage_rows(float* dst, const float* src, long ld, long row0, int r0, int nrows, int L, int D,
                                           int S, float mul, int tid) {
  const int lpr_log = D <= 16 ? 4 : (D <= 32 ? 5 : 6);
  const int tx = tid & ((1 << lpr_log) - 1), ty = tid >> lpr_log, rows_per_pass = 256 >> lpr_log;
  // Up to 4 rows x 4 column strides = 16 loads in flight per thread before the first LDS store: left as one load -> wait ->
  // store per iteration (run-time trip counts, no unrolling) a 64-row tile was 48 dependent memory round trips - 20 us of the
  // 27-us forward of the sliced notebook model (L = 32 samples, D = 160).  Clamped addresses + selects: no branch around a
  // load; rows past the sequence end are zero-filled without touching memory.
  constexpr int U = 4, C = 4;                                  // S <= 196 floats = 4 strides of 64 (3 of 16 / 32 for short heads)
  const float* base = src + row0 * ld;
  const int live = L - r0 < nrows ? (L - r0 > 0 ? L - r0 : 0) : nrows;      // rows [0, live) exist
  for (int j0 = ty; j0 < live; j0 += rows_per_pass * U) {
    float v[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + j0 + u * rows_per_pass;
      const unsigned ro = (unsigned)(r < L ? r : 0) * (unsigned)ld;     // 32-bit offsets from one uniform base: L * ld < 2^31
#pragma unroll                                                          // (checked on the host), one VGPR per address
      for (int c = 0; c < C; ++c) {
        const int d = tx + (c << lpr_log);
        v[u][c] = base[ro + (unsigned)(d < D ? d : 0)];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * rows_per_pass;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int d = tx + (c << lpr_log);
        if (j < live && d < S) dst[j * S + d] = d < D ? v[u][c] * mul : 0.f;
      }
    }
  }
  for (int j = live + ty; j < nrows; j += rows_per_pass)
    for (int d = tx; d < S; d += (1 << lpr_log)) dst[j * S + d] = 0.f;
}

// two tensors with the same row range staged together (K and V; Q and dO): their loads are in flight at once, 8 rows x 4
// column strides each - 32 rows of K and V in ONE memory round trip (the sliced models attend over 32 samples)
__device__ __forceinline__ void stage_rows2(float* dstA, const float* srcA, long ldA, float mulA, float* dstB, const float* srcB,
                                            long ldB, float mulB, long row0, int r0, int nrows, int L, int D, int S, int tid) {
  const int lpr_log = D <= 16 ? 4 : (D <= 32 ? 5 : 6);
  const int tx = tid & ((1 << lpr_log) - 1), ty = tid >> lpr_log, rows_per_pass = 256 >> lpr_log;
  constexpr int U = 8, C = 4;
  const float* baseA = srcA + row0 * ldA;
  const float* baseB = srcB + row0 * ldB;
  const int live = L - r0 < nrows ? (L - r0 > 0 ? L - r0 : 0) : nrows;
  for (int j0 = ty; j0 < live; j0 += rows_per_pass * U) {
    float va[U][C], vb[U][C];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + j0 + u * rows_per_pass;
      const unsigned rc = (unsigned)(r < L ? r : 0);
      const unsigned roA = rc * (unsigned)ldA, roB = rc * (unsigned)ldB;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int d = tx + (c << lpr_log);
        const unsigned dc = (unsigned)(d < D ? d : 0);
        va[u][c] = baseA[roA + dc];
        vb[u][c] = baseB[roB + dc];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * rows_per_pass;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int d = tx + (c << lpr_log);
        if (j < live && d < S) {
          dstA[j * S + d] = d < D ? va[u][c] * mulA : 0.f;
          dstB[j * S + d] = d < D ? vb[u][c] * mulB : 0.f;
        }
      }
    }
  }
  for (int j = live + ty; j < nrows; j += rows_per_pass)
    for (int d = tx; d < S; d += (1 << lpr_log)) { dstA[j * S + d] = 0.f; dstB[j * S + d] = 0.f; }
}

// acc[r] += <rowsA[r] (wave-uniform rows, broadcast reads), rowB (this lane's row)> over the padded head dim, d ascending
template <int RW>
__device__ __forceinline__ void dot4(const float* rowsA, const float* rowB, int S, int D4, float (&acc)[RW]) {
  for (int d = 0; d < D4; d += 4) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(rowB + d);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(rowsA + r * S + d);
      acc[r] = fmaf(a[0], b[0], acc[r]);
      acc[r] = fmaf(a[1], b[1], acc[r]);
      acc[r] = fmaf(a[2], b[2], acc[r]);
      acc[r] = fmaf(a[3], b[3], acc[r]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: workgroup = 16 queries of one (batch, head); wave w owns queries q0 + 4w .. + 3
template <int DT, int RW>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const AF p) {
  constexpr int RB = NW * RW;
  extern __shared__ float smem[];
  const unsigned dseed = p.drop_thr ? drop_seed_eff(p.drop_seed, p.drop_epoch) : 0u;
  const int L = p.L, H = p.H, D = p.D, S = row_stride(D), D4 = (D + 3) & ~3;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * RB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long ld = 3L * H * D, row0 = (long)b * L;
  float* Ks = smem;                        // [KB][S]
  float* Vs = Ks + KB * S;                 // [KB][S]
  float* Qs = Vs + KB * S;                 // [RB][S], pre-multiplied by q_scale
  float* Ps = Qs + RB * S;                 // [NW][KB][RW]
  stage_rows(Qs, p.qkv + (long)h * D, ld, row0, q0, RB, L, D, S, p.scale, tid);
  float m[RW], l[RW], acc[RW][DT];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    m[r] = -INFINITY; l[r] = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) acc[r][t] = 0.f;
  }
  for (int k0 = 0; k0 < L; k0 += KB) {
    __syncthreads();                                        // previous block's readers are done (and Qs is written)
    stage_rows2(Ks, p.qkv + (long)(H + h) * D, ld, 1.f, Vs, p.qkv + (long)(2 * H + h) * D, ld, 1.f, row0, k0, KB, L, D, S, tid);
    __syncthreads();
    const int key = k0 + lane;
    const bool valid = key < L && (!p.key_mask || p.key_mask[row0 + key]);
    float s[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) s[r] = 0.f;
    dot4<RW>(Qs + w * RW * S, Ks + lane * S, S, D4, s);
    float pt[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float sv = valid ? s[r] : -INFINITY;
      const float m_new = fmaxf(m[r], wave_max(sv));
      float pv = 0.f;
      if (m_new != -INFINITY) {                             // wave-uniform
        const float corr = expf(m[r] - m_new);              // m = -inf: 0
        pv = expf(sv - m_new);
        l[r] = fmaf(l[r], corr, wave_sum(pv));
#pragma unroll
        for (int t = 0; t < DT; ++t) acc[r][t] *= corr;
        m[r] = m_new;
      }
      if (p.drop_thr) pv *= drop_at(p, dseed, row0 + q0 + w * RW + r, h, key);
      pt[r] = pv;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) Ps[(w * KB + lane) * RW + r] = pt[r];
    __syncthreads();
    const int nk = L - k0 < KB ? L - k0 : KB;
    for (int j = 0; j < nk; ++j) {
      float pj[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) pj[r] = Ps[(w * KB + j) * RW + r];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int d = lane + 64 * t;
        const float v = Vs[j * S + (d < D ? d : 0)];
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[r][t] = fmaf(pj[r], v, acc[r][t]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int q = q0 + w * RW + r;
    if (q >= L) continue;
    const float inv = l[r] > 0.f ? 1.0f / l[r] : 0.f;       // every key masked: zero row, lse = -inf (as attention.hip)
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int d = lane + 64 * t;
      if (d < D) p.out[(row0 + q) * ((long)H * D) + (long)h * D + d] = acc[r][t] * inv;
    }
    if (lane == 0) p.lse[((long)b * H + h) * L + q] = l[r] > 0.f ? m[r] + logf(l[r]) : -INFINITY;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dQ (+ delta): same ownership as the forward
template <int DT, int RW>
__global__ __launch_bounds__(256) void attn_f32_bwd_dq_kernel(const AF p) {
  constexpr int RB = NW * RW;
  extern __shared__ float smem[];
  const unsigned dseed = p.drop_thr ? drop_seed_eff(p.drop_seed, p.drop_epoch) : 0u;
  const int L = p.L, H = p.H, D = p.D, S = row_stride(D), D4 = (D + 3) & ~3;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * RB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long ld = 3L * H * D, ldo = (long)H * D, row0 = (long)b * L;
  float* Ks = smem;
  float* Vs = Ks + KB * S;
  float* Qs = Vs + KB * S;                 // [RB][S] q * q_scale
  float* Gs = Qs + RB * S;                 // [RB][S] dO
  float* Ds = Gs + RB * S;                 // [NW][KB][RW] dS
  // the rows' own loads (dO and O for delta, lse) are REQUESTED first and consumed after the staging below: one memory
  // round trip for all of it (clamped addresses, masked at use: no branch around a load)
  float lse[RW], delta[RW], dq[RW][DT], gv[RW][DT], ov[RW][DT];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int q = q0 + w * RW + r;
    const int qc = q < L ? q : 0;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int d = lane + 64 * t;
      const long o = (row0 + qc) * ldo + (long)h * D + (d < D ? d : 0);
      gv[r][t] = p.dout[o];
      ov[r][t] = p.out[o];
    }
    lse[r] = p.lse[((long)b * H + h) * L + qc];
  }
  stage_rows2(Qs, p.qkv + (long)h * D, ld, p.scale, Gs, p.dout + (long)h * D, ldo, 1.f, row0, q0, RB, L, D, S, tid);
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int q = q0 + w * RW + r;
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int d = lane + 64 * t;
      if (q < L && d < D) part = fmaf(gv[r][t], ov[r][t], part);
    }
    delta[r] = wave_sum(part);
    if (q >= L) lse[r] = -INFINITY;
    if (lane == 0 && q < L) p.delta[((long)b * H + h) * L + q] = delta[r];
#pragma unroll
    for (int t = 0; t < DT; ++t) dq[r][t] = 0.f;
  }
  for (int k0 = 0; k0 < L; k0 += KB) {
    __syncthreads();
    stage_rows2(Ks, p.qkv + (long)(H + h) * D, ld, 1.f, Vs, p.qkv + (long)(2 * H + h) * D, ld, 1.f, row0, k0, KB, L, D, S, tid);
    __syncthreads();
    const int key = k0 + lane;
    const bool valid = key < L && (!p.key_mask || p.key_mask[row0 + key]);
    float s[RW], dp[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    dot4<RW>(Qs + w * RW * S, Ks + lane * S, S, D4, s);
    dot4<RW>(Gs + w * RW * S, Vs + lane * S, S, D4, dp);
    float dst[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float pv = (valid && lse[r] != -INFINITY) ? expf(s[r] - lse[r]) : 0.f;
      const float dm = p.drop_thr ? drop_at(p, dseed, row0 + q0 + w * RW + r, h, key) : 1.f;
      dst[r] = pv * fmaf(dp[r], dm, -delta[r]);              // dS = P o (dP~ - delta)
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) Ds[(w * KB + lane) * RW + r] = dst[r];
    __syncthreads();
    const int nk = L - k0 < KB ? L - k0 : KB;
    for (int j = 0; j < nk; ++j) {
      float dj[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) dj[r] = Ds[(w * KB + j) * RW + r];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int d = lane + 64 * t;
        const float kv = Ks[j * S + (d < D ? d : 0)];
#pragma unroll
        for (int r = 0; r < RW; ++r) dq[r][t] = fmaf(dj[r], kv, dq[r][t]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int q = q0 + w * RW + r;
    if (q >= L) continue;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int d = lane + 64 * t;
      if (d < D) p.dqkv[(row0 + q) * ld + (long)h * D + d] = dq[r][t] * p.scale;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dK / dV: workgroup = 16 keys of one (batch, head), sweeping 64-query blocks (lane = query); needs delta
template <int DT, int RW>
__global__ __launch_bounds__(256) void attn_f32_bwd_dkv_kernel(const AF p) {
  constexpr int RB = NW * RW;
  extern __shared__ float smem[];
  const unsigned dseed = p.drop_thr ? drop_seed_eff(p.drop_seed, p.drop_epoch) : 0u;
  const int L = p.L, H = p.H, D = p.D, S = row_stride(D), D4 = (D + 3) & ~3;
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * RB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long ld = 3L * H * D, ldo = (long)H * D, row0 = (long)b * L;
  float* Qs = smem;                        // [KB][S] q * q_scale
  float* Gs = Qs + KB * S;                 // [KB][S] dO
  float* Kr = Gs + KB * S;                 // [RB][S]
  float* Vr = Kr + RB * S;                 // [RB][S]
  float* Ps = Vr + RB * S;                 // [NW][KB][RW] P~
  float* Ds = Ps + NW * KB * RW;           // [NW][KB][RW] dS
  stage_rows2(Kr, p.qkv + (long)(H + h) * D, ld, 1.f, Vr, p.qkv + (long)(2 * H + h) * D, ld, 1.f, row0, k0, RB, L, D, S, tid);
  bool kvalid[RW];
  float dk[RW][DT], dv[RW][DT];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int key = k0 + w * RW + r;
    kvalid[r] = key < L && (!p.key_mask || p.key_mask[row0 + key]);
#pragma unroll
    for (int t = 0; t < DT; ++t) { dk[r][t] = 0.f; dv[r][t] = 0.f; }
  }
  for (int q0 = 0; q0 < L; q0 += KB) {
    const int q = q0 + lane;                                 // (requested before the staging: one round trip, not two)
    const float lse = q < L ? p.lse[((long)b * H + h) * L + q] : -INFINITY;
    const float delta = q < L ? p.delta[((long)b * H + h) * L + q] : 0.f;
    __syncthreads();
    stage_rows2(Qs, p.qkv + (long)h * D, ld, p.scale, Gs, p.dout + (long)h * D, ldo, 1.f, row0, q0, KB, L, D, S, tid);
    __syncthreads();
    float s[RW], dp[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    dot4<RW>(Kr + w * RW * S, Qs + lane * S, S, D4, s);
    dot4<RW>(Vr + w * RW * S, Gs + lane * S, S, D4, dp);
    float pt[RW], dst[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const float pv = (kvalid[r] && lse != -INFINITY) ? expf(s[r] - lse) : 0.f;
      const float dm = p.drop_thr ? drop_at(p, dseed, row0 + q, h, k0 + w * RW + r) : 1.f;
      pt[r] = pv * dm;
      dst[r] = pv * fmaf(dp[r], dm, -delta);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      Ps[(w * KB + lane) * RW + r] = pt[r];
      Ds[(w * KB + lane) * RW + r] = dst[r];
    }
    __syncthreads();
    const int nq = L - q0 < KB ? L - q0 : KB;
    for (int i = 0; i < nq; ++i) {
      float pi[RW], di[RW];
#pragma unroll
      for (int r = 0; r < RW; ++r) { pi[r] = Ps[(w * KB + i) * RW + r]; di[r] = Ds[(w * KB + i) * RW + r]; }
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const int d = lane + 64 * t;
        const int dd = d < D ? d : 0;
        const float g = Gs[i * S + dd], qv = Qs[i * S + dd];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          dv[r][t] = fmaf(pi[r], g, dv[r][t]);
          dk[r][t] = fmaf(di[r], qv, dk[r][t]);              // Qs holds q * q_scale: d s / d k
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int key = k0 + w * RW + r;
    if (key >= L) continue;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const int d = lane + 64 * t;
      if (d < D) {
        p.dqkv[(row0 + key) * ld + (long)(H + h) * D + d] = dk[r][t];
        p.dqkv[(row0 + key) * ld + (long)(2 * H + h) * D + d] = dv[r][t];
      }
    }
  }
}

size_t lds_fwd(int D, int RW) { return (size_t)(2 * KB * row_stride(D) + NW * RW * row_stride(D) + NW * KB * RW) * 4; }
size_t lds_dq(int D, int RW) { return (size_t)(2 * KB * row_stride(D) + 2 * NW * RW * row_stride(D) + NW * KB * RW) * 4; }
size_t lds_dkv(int D, int RW) { return (size_t)(2 * KB * row_stride(D) + 2 * NW * RW * row_stride(D) + 2 * NW * KB * RW) * 4; }

template <typename K>
int launch(K kern, std::atomic<uint64_t>& once, const AF& p, int rb, size_t lds, hipStream_t st) {
  clipk_once_per_device(once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const dim3 grid((unsigned)((p.L + rb - 1) / rb), (unsigned)p.H, (unsigned)p.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  return clipk_check_launch();
}

// rows per wave: 4 (each K / V read from LDS serves four query rows), or 1 when that would leave the chip empty
int rows_per_wave(const AF& p) { return ((long)((p.L + 15) / 16) * p.H * p.B < 128) ? 1 : 4; }

template <int DT>
int launch_fwd(const AF& p, hipStream_t st) {
  static std::atomic<uint64_t> o1{0}, o4{0};
  if (rows_per_wave(p) == 1) return launch(attn_f32_fwd_kernel<DT, 1>, o1, p, NW, lds_fwd(p.D, 1), st);
  return launch(attn_f32_fwd_kernel<DT, 4>, o4, p, NW * 4, lds_fwd(p.D, 4), st);
}
template <int DT>
int launch_bwd(const AF& p, hipStream_t st) {
  static std::atomic<uint64_t> a1{0}, a4{0}, b1{0}, b4{0};
  if (rows_per_wave(p) == 1) {
    const int rc = launch(attn_f32_bwd_dq_kernel<DT, 1>, a1, p, NW, lds_dq(p.D, 1), st);
    return rc ? rc : launch(attn_f32_bwd_dkv_kernel<DT, 1>, b1, p, NW, lds_dkv(p.D, 1), st);
  }
  const int rc = launch(attn_f32_bwd_dq_kernel<DT, 4>, a4, p, NW * 4, lds_dq(p.D, 4), st);
  return rc ? rc : launch(attn_f32_bwd_dkv_kernel<DT, 4>, b4, p, NW * 4, lds_dkv(p.D, 4), st);
}

int set_dropout(AF& p, float dropout_p, unsigned seed) {
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f) return CLIPK_ERR_BAD_ARG;
  p.drop_thr = 0; p.drop_seed = 0; p.drop_scale = 1.f; p.drop_epoch = clipk_drop_epoch();
  if (dropout_p == 0.f) return CLIPK_OK;
  const double t = (double)dropout_p * 4294967296.0;
  p.drop_thr = t < 1.0 ? 1u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
  p.drop_seed = seed;
  p.drop_scale = 1.0f / (1.0f - dropout_p);
  return CLIPK_OK;
}

int check(const void* qkv, int B, int L, int H, int D) {
  if (!qkv || B <= 0 || L <= 0 || H <= 0 || D <= 0) return CLIPK_ERR_BAD_ARG;
  if (D > 192 || H > 65535 || B > 65535 || (long)L * 3 * H * D >= (1L << 31)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(qkv)) return CLIPK_ERR_BAD_ARG;
  return CLIPK_OK;
}

}  // namespace

extern "C" int clipk_attn_f32_fwd(const float* qkv, const uint8_t* key_mask, float* out, float* lse, int B, int L, int H,
                                  int D, float q_scale, float dropout_p, uint32_t dropout_seed, void* stream) {
  int rc = check(qkv, B, L, H, D);
  if (rc) return rc;
  if (!out || !lse) return CLIPK_ERR_BAD_ARG;
  AF p{};
  p.qkv = qkv; p.key_mask = key_mask; p.out = out; p.lse = lse; p.B = B; p.L = L; p.H = H; p.D = D; p.scale = q_scale;
  rc = set_dropout(p, dropout_p, dropout_seed);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (D <= 64) return launch_fwd<1>(p, st);
  if (D <= 128) return launch_fwd<2>(p, st);
  return launch_fwd<3>(p, st);
}

extern "C" int clipk_attn_f32_bwd(const float* qkv, const uint8_t* key_mask, const float* out, const float* dout,
                                  const float* lse, float* delta, float* dqkv, int B, int L, int H, int D, float q_scale,
                                  float dropout_p, uint32_t dropout_seed, void* stream) {
  int rc = check(qkv, B, L, H, D);
  if (rc) return rc;
  if (!out || !dout || !lse || !delta || !dqkv) return CLIPK_ERR_BAD_ARG;
  AF p{};
  p.qkv = qkv; p.key_mask = key_mask; p.out = const_cast<float*>(out); p.lse = const_cast<float*>(lse);
  p.dout = dout; p.delta = delta; p.dqkv = dqkv; p.B = B; p.L = L; p.H = H; p.D = D; p.scale = q_scale;
  rc = set_dropout(p, dropout_p, dropout_seed);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (D <= 64) return launch_bwd<1>(p, st);
  if (D <= 128) return launch_bwd<2>(p, st);
  return launch_bwd<3>(p, st);
}
