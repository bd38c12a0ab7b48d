// rowwise.hip — HBM-bound row kernels: LayerNorm fwd/bwd (+fused activation), L2-normalise fwd/bwd.
// One wave (64 lanes) per row, whole row held in registers as 4-element chunks (16-byte f32 loads,
// 8-byte bf16 loads), f32 statistics with wave-shuffle reductions — no LDS, no re-reads.
// Reference ops: nn.LayerNorm at old/clip.py:12,28,32, the LayerNorms inside nn.TransformerEncoderLayer
// (rna_clip_codes.ipynb:1915-1916) and EsmLayer (modeling_esm.py:429-438,552-553); F.normalize at
// old/clip.py:63-64.
#include "common.h"

namespace {

template <bool BF16>
__device__ __forceinline__ f32x4 load4(const void* base, long off) {
  if (BF16) {
    const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + off);
    return f32x4{bf16_to_f32((unsigned short)(r[0] & 0xffffu)), bf16_to_f32((unsigned short)(r[0] >> 16)),
                 bf16_to_f32((unsigned short)(r[1] & 0xffffu)), bf16_to_f32((unsigned short)(r[1] >> 16))};
  } else {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
  }
}
__device__ __forceinline__ void store4_bf16(void* base, long off, f32x4 v) {
  u32x2 o;
  o[0] = pack_bf16x2(v[0], v[1]);
  o[1] = pack_bf16x2(v[2], v[3]);
  *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(base) + off) = o;
}

struct LnFwd {
  const void* x; long ldx;
  const float* gamma; const float* beta; float eps; int act;
  float* y_f32; void* y_bf16; long ldy;
  float* mean; float* rstd; int rows, cols;
};

template <int VPL, bool XBF16>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwd p) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = p.cols >> 2;
  f32x4 g[VPL], b[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    g[v] = f32x4{0.f, 0.f, 0.f, 0.f}; b[v] = g[v];
    if (c < nch) {
      g[v] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c);
      b[v] = *reinterpret_cast<const f32x4*>(p.beta + 4 * c);
    }
  }
  const float inv_n = 1.0f / (float)p.cols;
  for (int row = wave; row < p.rows; row += nwaves) {
    f32x4 x[VPL];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      x[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch) x[v] = load4<XBF16>(p.x, (long)row * p.ldx + 4 * c);
      s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
    }
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[v][e] - mean; q += d * d; }
      }
    }
    const float var = wave_sum(q) * inv_n;
    const float rstd = rsqrtf(var + p.eps);
    if (lane == 0) {
      if (p.mean) p.mean[row] = mean;
      if (p.rstd) p.rstd[row] = rstd;
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      if (c < nch) {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = act_apply((x[v][e] - mean) * rstd * g[v][e] + b[v][e], p.act);
        if (p.y_f32) *reinterpret_cast<f32x4*>(p.y_f32 + (long)row * p.ldy + 4 * c) = y;
        if (p.y_bf16) store4_bf16(p.y_bf16, (long)row * p.ldy + 4 * c, y);
      }
    }
  }
}

// Wide rows (cols > 2048: the 2560-wide hidden LayerNorms of the notebook's projection heads, rna_clip_codes.ipynb:1887-1903):
// one WORKGROUP per row, the four waves take interleaved 256-float segments (chunk = lane + 64 (4 v + wave)), row sums meet
// in LDS.  The one-wave-per-row form needs 20 float4 per lane and array (560 registers in the backward: it lived in scratch,
// 48 us for 32 rows).  f32 in / f32 or bf16 out like the narrow kernel; the row statistics are sums of four wave sums.
template <int VPW, bool XBF16>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(const LnFwd p) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = p.cols >> 2;
  f32x4 g[VPW], b[VPW];
#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = lane + 64 * (4 * v + wid);
    g[v] = f32x4{0.f, 0.f, 0.f, 0.f}; b[v] = g[v];
    if (c < nch) {
      g[v] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c);
      b[v] = *reinterpret_cast<const f32x4*>(p.beta + 4 * c);
    }
  }
  const float inv_n = 1.0f / (float)p.cols;
  for (int row = blockIdx.x; row < p.rows; row += gridDim.x) {
    f32x4 x[VPW];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c = lane + 64 * (4 * v + wid);
      x[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch) x[v] = load4<XBF16>(p.x, (long)row * p.ldx + 4 * c);
      s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
    }
    s = wave_sum(s);
    if (lane == 0) red[0][wid] = s;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c = lane + 64 * (4 * v + wid);
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[v][e] - mean; q += d * d; }
      }
    }
    q = wave_sum(q);
    if (lane == 0) red[1][wid] = q;
    __syncthreads();                       // (also: every wave has read red[0] before the next row overwrites it)
    const float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * inv_n;
    const float rstd = rsqrtf(var + p.eps);
    if (threadIdx.x == 0) {
      if (p.mean) p.mean[row] = mean;
      if (p.rstd) p.rstd[row] = rstd;
    }
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c = lane + 64 * (4 * v + wid);
      if (c < nch) {
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = act_apply((x[v][e] - mean) * rstd * g[v][e] + b[v][e], p.act);
        if (p.y_f32) *reinterpret_cast<f32x4*>(p.y_f32 + (long)row * p.ldy + 4 * c) = y;
        if (p.y_bf16) store4_bf16(p.y_bf16, (long)row * p.ldy + 4 * c, y);
      }
    }
  }
}

// LayerNorm + masked mean over the L rows of each sample in one pass (the encoders' final LayerNorm followed by the
// pooling of configuration_hybrid_clip.py:109 use_mean_pooling / tf_clip_codes (1).ipynb:1188): the normalised rows
// are never written (2 x rows x cols x 4 B less per tower and step: one write here, one read by the pooling kernel).
// One workgroup per sample; wave w takes rows w, w + 4, ...; the four partial sums are combined in wave order, so a
// sample's pooled row does not depend on where it sits in the batch.  mean / rstd per row are kept for the backward.
struct LnPool {
  const void* x; long ldx;        // f32 or bf16 (XBF16)
  const float* gamma; const float* beta; float eps;
  const unsigned char* mask;      // [B * L], 1 = valid, or null
  float* pooled; float* mean; float* rstd; float* wrow;   // wrow [B * L]: the row's weight in its mean, 1 / #valid or 0
  int L, cols;
};

template <int VPL, bool XBF16>
__global__ __launch_bounds__(256) void ln_pool_fwd_kernel(const LnPool p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sm = reinterpret_cast<float*>(smem);                    // [4][cols] partial sums, then [4] counts
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int nch = p.cols >> 2;
  f32x4 g[VPL], be[VPL], acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    g[v] = f32x4{0.f, 0.f, 0.f, 0.f}; be[v] = g[v]; acc[v] = g[v];
    if (c < nch) {
      g[v] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c);
      be[v] = *reinterpret_cast<const f32x4*>(p.beta + 4 * c);
    }
  }
  const float inv_n = 1.0f / (float)p.cols;
  int cnt = 0;
  for (int l = wid; l < p.L; l += 4) {
    const long row = (long)b * p.L + l;
    f32x4 x[VPL];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      x[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch) x[v] = load4<XBF16>(p.x, row * p.ldx + 4 * c);
      s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
    }
    const float mean = wave_sum(s) * inv_n;                      // same statistics arithmetic as ln_fwd_kernel
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x[v][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_n + p.eps);
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
    const bool ok = !p.mask || p.mask[row];                      // wave-uniform
    if (ok) {
      ++cnt;
#pragma unroll
      for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[v][e] += (x[v][e] - mean) * rstd * g[v][e] + be[v][e];
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    if (c < nch) *reinterpret_cast<f32x4*>(sm + wid * p.cols + 4 * c) = acc[v];
  }
  if (lane == 0) sm[4 * p.cols + wid] = (float)cnt;
  __syncthreads();
  const float n = (sm[4 * p.cols] + sm[4 * p.cols + 1]) + (sm[4 * p.cols + 2] + sm[4 * p.cols + 3]);
  const float inv = n > 0.f ? 1.0f / n : 0.f;
  for (int l = threadIdx.x; l < p.L; l += blockDim.x) {
    const long row = (long)b * p.L + l;
    p.wrow[row] = (!p.mask || p.mask[row]) ? inv : 0.f;          // what the backward multiplies the pooled gradient by
  }
  for (int i = threadIdx.x; i < p.cols; i += blockDim.x) {
    const float a = ((sm[i] + sm[p.cols + i]) + sm[2 * p.cols + i]) + sm[3 * p.cols + i];
    p.pooled[(long)b * p.cols + i] = a * inv;
  }
}

struct LnBwd {
  const void* dy; long lddy;
  const void* x; long ldx;
  const float* gamma; const float* beta; const float* mean; const float* rstd; int act;
  const void* dx_add; float* dx_f32; void* dx_bf16; long lddx;
  float* part;   // [nblocks][2][cols]
  int rows, cols;
  // dropout mask on the bf16 output only: it is the gradient of a Linear's dropped-out output (d(W x) = keep / (1-p) *
  // d(residual sum)), the f32 output stays the residual-path gradient.  Index = row * cols + col.  thr 0 = off.
  unsigned drop_thr, drop_seed; float drop_scale;
  int add_bf16;     // dx_add holds bf16 instead of f32
  // POOL instantiation (LayerNorm followed by a masked mean over the L rows of each sample): dy is the gradient of the
  // POOLED row, f32 [rows / pool_L][cols]; row r receives dy[r / pool_L] * wrow[r] (1 / #valid rows of its sample, or 0:
  // written by ln_pool_fwd_kernel next to mean / rstd, so that the row's scalars are three independent loads)
  const float* wrow; int pool_L;
  const unsigned* drop_epoch;   // common.h drop_seed_eff (last member: the aggregate initialisers below leave it nullptr)
};

template <int VPL, bool DYBF16, bool XBF16, bool ADD16 = false, bool POOL = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwd p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = p.cols >> 2;
  f32x4 g[VPL], b[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    g[v] = f32x4{0.f, 0.f, 0.f, 0.f}; b[v] = g[v]; dg[v] = g[v]; db[v] = g[v];
    if (c < nch) {
      g[v] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c);
      if (p.act != CLIPK_ACT_NONE) b[v] = *reinterpret_cast<const f32x4*>(p.beta + 4 * c);
    }
  }
  const float inv_n = 1.0f / (float)p.cols;
  for (int row = wave; row < p.rows; row += nwaves) {
    const float mean = p.mean[row], rstd = p.rstd[row];
    f32x4 xh[VPL], gy[VPL], addv[VPL];
    float s1 = 0.f, s2 = 0.f;
    long dyrow = row;
    float pw = 1.f;
    if constexpr (POOL) {
      dyrow = row / p.pool_L;
      pw = p.wrow[row];
    }
    // the residual-path gradient is only needed after the row reductions: issue its load with the others, so the
    // row costs one memory round trip instead of two
    u32x2 addp[VPL];                            // ADD16: the residual-path gradient stays packed bf16 until it is added
#pragma unroll                                  // (2 instead of 4 registers per chunk: 100 -> 96 VGPRs at VPL 2 = 5 waves / SIMD)
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      addv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      addp[v] = u32x2{0u, 0u};
      if constexpr (!POOL) {                    // (the pooled form has no residual-path gradient: 16 registers at VPL 4,
        if (p.dx_add && c < nch) {              //  the difference between 2 and 3 waves per SIMD)
          if constexpr (ADD16)
            addp[v] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p.dx_add) + (long)row * p.lddx + 4 * c);
          else
            addv[v] = load4<false>(p.dx_add, (long)row * p.lddx + 4 * c);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      xh[v] = f32x4{0.f, 0.f, 0.f, 0.f}; gy[v] = xh[v];
      if (c < nch) {
        const f32x4 xv = load4<XBF16>(p.x, (long)row * p.ldx + 4 * c);
        f32x4 dyv = load4<DYBF16>(p.dy, dyrow * p.lddy + 4 * c);
        if constexpr (POOL) dyv = dyv * pw;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xhat = (xv[e] - mean) * rstd;
          if (p.act != CLIPK_ACT_NONE) dyv[e] *= act_grad(xhat * g[v][e] + b[v][e], p.act);
          xh[v][e] = xhat;
          dg[v][e] += dyv[e] * xhat;
          db[v][e] += dyv[e];
          const float gg = dyv[e] * g[v][e];
          gy[v][e] = gg;
          s1 += gg; s2 += gg * xhat;
        }
      }
    }
    const float c1 = wave_sum(s1) * inv_n, c2 = wave_sum(s2) * inv_n;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      if (c < nch) {
        f32x4 dx;
#pragma unroll
        for (int e = 0; e < 4; ++e) dx[e] = rstd * (gy[v][e] - c1 - xh[v][e] * c2);
        if constexpr (!POOL) {
          if constexpr (ADD16)
            dx += f32x4{bf16_to_f32((unsigned short)(addp[v][0] & 0xffffu)), bf16_to_f32((unsigned short)(addp[v][0] >> 16)),
                        bf16_to_f32((unsigned short)(addp[v][1] & 0xffffu)), bf16_to_f32((unsigned short)(addp[v][1] >> 16))};
          else
            dx += addv[v];
        }
        if (p.dx_f32) *reinterpret_cast<f32x4*>(p.dx_f32 + (long)row * p.lddx + 4 * c) = dx;
        if (p.dx_bf16) {
          if (p.drop_thr) {
            const unsigned long long base = (unsigned long long)row * (unsigned)p.cols + 4u * c;
            const unsigned dseed = drop_seed_eff(p.drop_seed, p.drop_epoch);
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[e] *= drop_mul(dseed, base + e, p.drop_thr, p.drop_scale);
          }
          store4_bf16(p.dx_bf16, (long)row * p.lddx + 4 * c, dx);
        }
      }
    }
  }
  // block-level combine of the 4 waves' dgamma/dbeta partials through LDS, then one partial row / block
  float* sm = reinterpret_cast<float*>(smem);          // [4][2][cols]
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    if (c < nch) {
      *reinterpret_cast<f32x4*>(sm + (wid * 2 + 0) * p.cols + 4 * c) = dg[v];
      *reinterpret_cast<f32x4*>(sm + (wid * 2 + 1) * p.cols + 4 * c) = db[v];
    }
  }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < 2 * p.cols; i += blockDim.x) {
    float a = 0.f;
    for (int w = 0; w < nw; ++w) a += sm[w * 2 * p.cols + i];
    p.part[(long)blockIdx.x * 2 * p.cols + i] = a;
  }
}

// Wide rows, backward (see ln_fwd_wide_kernel): one workgroup per row, f32 dy / x, optional f32 residual-path gradient;
// every wave keeps the dgamma / dbeta partials of ITS columns and writes them to the block's partial row itself.
template <int VPW>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(const LnBwd p) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = p.cols >> 2;
  f32x4 g[VPW], b[VPW], dg[VPW], db[VPW];
#pragma unroll
  for (int v = 0; v < VPW; ++v) {
    const int c = lane + 64 * (4 * v + wid);
    g[v] = f32x4{0.f, 0.f, 0.f, 0.f}; b[v] = g[v]; dg[v] = g[v]; db[v] = g[v];
    if (c < nch) {
      g[v] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c);
      if (p.act != CLIPK_ACT_NONE) b[v] = *reinterpret_cast<const f32x4*>(p.beta + 4 * c);
    }
  }
  const float inv_n = 1.0f / (float)p.cols;
  for (int row = blockIdx.x; row < p.rows; row += gridDim.x) {
    const float mean = p.mean[row], rstd = p.rstd[row];
    f32x4 xh[VPW], gy[VPW], addv[VPW];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c = lane + 64 * (4 * v + wid);
      addv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.dx_add && c < nch) addv[v] = load4<false>(p.dx_add, (long)row * p.lddx + 4 * c);
    }
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c = lane + 64 * (4 * v + wid);
      xh[v] = f32x4{0.f, 0.f, 0.f, 0.f}; gy[v] = xh[v];
      if (c < nch) {
        const f32x4 xv = load4<false>(p.x, (long)row * p.ldx + 4 * c);
        f32x4 dyv = load4<false>(p.dy, (long)row * p.lddy + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xhat = (xv[e] - mean) * rstd;
          if (p.act != CLIPK_ACT_NONE) dyv[e] *= act_grad(xhat * g[v][e] + b[v][e], p.act);
          xh[v][e] = xhat;
          dg[v][e] += dyv[e] * xhat;
          db[v][e] += dyv[e];
          const float gg = dyv[e] * g[v][e];
          gy[v][e] = gg;
          s1 += gg; s2 += gg * xhat;
        }
      }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    __syncthreads();                       // the previous row's sums have been read by every wave
    if (lane == 0) { red[0][wid] = s1; red[1][wid] = s2; }
    __syncthreads();
    const float c1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * inv_n;
    const float c2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * inv_n;
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      const int c = lane + 64 * (4 * v + wid);
      if (c < nch) {
        f32x4 dx;
#pragma unroll
        for (int e = 0; e < 4; ++e) dx[e] = rstd * (gy[v][e] - c1 - xh[v][e] * c2);
        dx += addv[v];
        if (p.dx_f32) *reinterpret_cast<f32x4*>(p.dx_f32 + (long)row * p.lddx + 4 * c) = dx;
        if (p.dx_bf16) {
          if (p.drop_thr) {
            const unsigned long long base = (unsigned long long)row * (unsigned)p.cols + 4u * c;
            const unsigned dseed = drop_seed_eff(p.drop_seed, p.drop_epoch);
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[e] *= drop_mul(dseed, base + e, p.drop_thr, p.drop_scale);
          }
          store4_bf16(p.dx_bf16, (long)row * p.lddx + 4 * c, dx);
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VPW; ++v) {            // this block's partial row: every wave writes its own columns
    const int c = lane + 64 * (4 * v + wid);
    if (c < nch) {
      *reinterpret_cast<f32x4*>(p.part + (long)blockIdx.x * 2 * p.cols + 4 * c) = dg[v];
      *reinterpret_cast<f32x4*>(p.part + (long)blockIdx.x * 2 * p.cols + p.cols + 4 * c) = db[v];
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Backward OF the LayerNorm(+activation) backward (second order).  The ICNN transport map is T(x) = dPsi/dx
// (2_icnn_core.py:181-211, create_graph=True) and its training loss is a function of T, so autograd differentiates
// the first backward  da = r (u - mean u - xh mean(u xh)),  u = dy * act'(n) * gamma,  n = xh gamma + beta,
// xh = (a - mean a) r,  r = rstd(a)  with respect to (dy, a, gamma, beta).  With g the cotangent of da, per row:
//   w  = r (g - mean g - xh mean(g xh))                     (the same projection applied to g)
//   d_dy = w gamma act'(n)
//   q  = w gamma dy act''(n)                                (cotangent of n through act')
//   tt = q gamma - r (u mean(g xh) + g mean(u xh))          (cotangent of xh: through n, and explicit in da)
//   d_a = r (tt - mean tt - xh (mean(tt xh) + mean(g da)))  (through xh and through r)
//   d_gamma += w dy act'(n) + q xh,   d_beta += q           (column sums over the rows)
// One wave per row, the row in registers, two rounds of shuffle reductions; dgamma / dbeta as per-block partial rows
// + the deterministic column reduce of the first-order kernel.
// -------------------------------------------------------------------------------------------------
struct LnBwd2 {
  const float* g; const float* dy; const float* a; long ld;
  const float* gamma; const float* beta; const float* mean; const float* rstd; int act;
  float* d_dy; float* d_a; float* part; int rows, cols;
};

template <int VPL>
__global__ __launch_bounds__(256) void ln_bwd2_kernel(const LnBwd2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = p.cols >> 2;
  f32x4 gm[VPL], bt[VPL], dgm[VPL], dbt[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    gm[v] = f32x4{0.f, 0.f, 0.f, 0.f}; bt[v] = gm[v]; dgm[v] = gm[v]; dbt[v] = gm[v];
    if (c < nch) {
      gm[v] = *reinterpret_cast<const f32x4*>(p.gamma + 4 * c);
      if (p.act != CLIPK_ACT_NONE) bt[v] = *reinterpret_cast<const f32x4*>(p.beta + 4 * c);
    }
  }
  const float inv_n = 1.0f / (float)p.cols;
  for (int row = wave; row < p.rows; row += nwaves) {
    const float mean = p.mean[row], r = p.rstd[row];
    f32x4 xh[VPL], gv[VPL], u[VPL], k1[VPL], k2[VPL];      // u = dy act'(n), k1 = act'(n), k2 = gamma dy act''(n)
    float su = 0.f, sux = 0.f, sg = 0.f, sgx = 0.f, sgu = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      xh[v] = f32x4{0.f, 0.f, 0.f, 0.f}; gv[v] = xh[v]; u[v] = xh[v]; k1[v] = xh[v]; k2[v] = xh[v];
      if (c < nch) {
        const long o = (long)row * p.ld + 4 * c;
        const f32x4 av = *reinterpret_cast<const f32x4*>(p.a + o);
        const f32x4 dyv = *reinterpret_cast<const f32x4*>(p.dy + o);
        gv[v] = *reinterpret_cast<const f32x4*>(p.g + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = (av[e] - mean) * r;
          float a1 = 1.f, a2 = 0.f;
          if (p.act != CLIPK_ACT_NONE) {
            const float n = x * gm[v][e] + bt[v][e];
            a1 = act_grad(n, p.act);
            a2 = act_grad2(n, p.act);
          }
          xh[v][e] = x;
          k1[v][e] = a1;
          k2[v][e] = gm[v][e] * dyv[e] * a2;
          const float ud = dyv[e] * a1;                     // dy act'(n): d_gamma's first term pairs it with w
          u[v][e] = ud;
          const float ug = ud * gm[v][e];                   // the first backward's u
          su += ug; sux += ug * x; sg += gv[v][e]; sgx += gv[v][e] * x; sgu += gv[v][e] * ug;
        }
      }
    }
    const float mu = wave_sum(su) * inv_n, cux = wave_sum(sux) * inv_n, mg = wave_sum(sg) * inv_n,
                dgx = wave_sum(sgx) * inv_n, mgu = wave_sum(sgu) * inv_n;
    const float e_gda = r * (mgu - mg * mu - dgx * cux);   // mean(g da)
    f32x4 tt[VPL], w[VPL];
    float st = 0.f, stx = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      tt[v] = f32x4{0.f, 0.f, 0.f, 0.f}; w[v] = tt[v];
      if (c < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = xh[v][e];
          const float ww = r * (gv[v][e] - mg - x * dgx);
          w[v][e] = ww;
          const float q = ww * k2[v][e];                   // w gamma dy act''(n)
          const float ug = u[v][e] * gm[v][e];              // u with gamma
          const float t = q * gm[v][e] - r * (ug * dgx + gv[v][e] * cux);
          tt[v][e] = t;
          st += t; stx += t * x;
          dgm[v][e] += ww * u[v][e] + q * x;
          dbt[v][e] += q;
        }
      }
    }
    const float mt = wave_sum(st) * inv_n, mtx = wave_sum(stx) * inv_n;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = lane + 64 * v;
      if (c < nch) {
        const long o = (long)row * p.ld + 4 * c;
        f32x4 ddy, da;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ddy[e] = w[v][e] * gm[v][e] * k1[v][e];
          da[e] = r * (tt[v][e] - mt - xh[v][e] * (mtx + e_gda));
        }
        if (p.d_dy) *reinterpret_cast<f32x4*>(p.d_dy + o) = ddy;
        if (p.d_a) *reinterpret_cast<f32x4*>(p.d_a + o) = da;
      }
    }
  }
  float* sm = reinterpret_cast<float*>(smem);          // [4][2][cols]
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    if (c < nch) {
      *reinterpret_cast<f32x4*>(sm + (wid * 2 + 0) * p.cols + 4 * c) = dgm[v];
      *reinterpret_cast<f32x4*>(sm + (wid * 2 + 1) * p.cols + 4 * c) = dbt[v];
    }
  }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < 2 * p.cols; i += blockDim.x) {
    float acc = 0.f;
    for (int ww = 0; ww < nw; ++ww) acc += sm[ww * 2 * p.cols + i];
    p.part[(long)blockIdx.x * 2 * p.cols + i] = acc;
  }
}

// part: [nparts][ncols] with ncols = 2*cols ([dgamma | dbeta]).  Block = 16 columns x 16 row groups: a thread sums
// every 16th partial row with 8 independent loads in flight (the old 64 x 4 shape ran 15 blocks of 256-deep
// dependent chains: 33 us for a 4 MB input), LDS combines the 16 groups in a fixed order (deterministic).
__global__ __launch_bounds__(256) void colreduce_kernel(const float* part, int nparts, int ncols, float* out0,
                                                        float* out1, int cols, int accumulate) {
  __shared__ float sm[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + c;
  float a = 0.f;
  if (i < ncols) {
    int s = rg;
    for (; s + 7 * 16 < nparts; s += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(s + 16 * u) * ncols + i];
      a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; s < nparts; s += 16) a += part[(long)s * ncols + i];
  }
  sm[rg][c] = a;
  __syncthreads();
  if (rg == 0 && i < ncols) {
    a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += sm[r][c];
    float* o = (i < cols) ? (out0 ? out0 + i : nullptr) : (out1 ? out1 + (i - cols) : nullptr);
    if (o) *o = accumulate ? (*o + a) : a;
  }
}

// Several column reduces in ONE launch (blockIdx.y = problem): the dgamma / dbeta partial rows of every LayerNorm whose
// backward ran in this pass, reduced when the pass is over (functional.LayerNormFn defers them: one launch instead of one
// per LayerNorm - 20 per step of the sliced notebook model, each a 4-us launch on the critical chain).  desc: 6 int64 per
// problem {part, nparts, cols, out0 (dgamma), out1 (dbeta), accumulate}; same block shape and summation order as
// colreduce_kernel.
__global__ __launch_bounds__(256) void colreduce_batched_kernel(const long long* desc) {
  const long long* d = desc + 6 * (long)blockIdx.y;
  const float* part = reinterpret_cast<const float*>(d[0]);
  const int nparts = (int)d[1], cols = (int)d[2], ncols = 2 * cols, accumulate = (int)d[5];
  float* out0 = reinterpret_cast<float*>(d[3]);
  float* out1 = reinterpret_cast<float*>(d[4]);
  if ((int)blockIdx.x * 16 >= ncols) return;                 // (block-uniform)
  __shared__ float sm[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + c;
  float a = 0.f;
  if (i < ncols) {
    int s = rg;
    for (; s + 7 * 16 < nparts; s += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(s + 16 * u) * ncols + i];
      a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; s < nparts; s += 16) a += part[(long)s * ncols + i];
  }
  sm[rg][c] = a;
  __syncthreads();
  if (rg == 0 && i < ncols) {
    a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += sm[r][c];
    float* o = (i < cols) ? (out0 ? out0 + i : nullptr) : (out1 ? out1 + (i - cols) : nullptr);
    if (o) *o = accumulate ? (*o + a) : a;
  }
}

// ---- L2 normalise -------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* y, float* norm, int rows, int cols,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int nch = cols >> 2;
  f32x4 xv[VPL];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    xv[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < nch) xv[v] = *reinterpret_cast<const f32x4*>(x + (long)row * cols + 4 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) s += xv[v][e] * xv[v][e];
  }
  const float n = sqrtf(wave_sum(s));
  const float inv = 1.0f / fmaxf(n, eps);
  if (lane == 0 && norm) norm[row] = n;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    if (c < nch) *reinterpret_cast<f32x4*>(y + (long)row * cols + 4 * c) = xv[v] * inv;
  }
}

template <int VPL>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* dy, const float* y, const float* norm, float* dx,
                                                         int rows, int cols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int nch = cols >> 2;
  f32x4 yv[VPL], dv[VPL];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    yv[v] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[v] = yv[v];
    if (c < nch) {
      yv[v] = *reinterpret_cast<const f32x4*>(y + (long)row * cols + 4 * c);
      dv[v] = *reinterpret_cast<const f32x4*>(dy + (long)row * cols + 4 * c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) s += yv[v][e] * dv[v][e];
  }
  const float dot = wave_sum(s);
  const float n = norm[row];
  // y = x / max(n, eps): for n >= eps, dx = (dy - y*(y.dy)) / n ; below eps the clamp is constant: dx = dy/eps
  const bool clamped = n < eps;
  const float inv = 1.0f / fmaxf(n, eps);
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = lane + 64 * v;
    if (c < nch) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = clamped ? dv[v][e] * inv : (dv[v][e] - yv[v][e] * dot) * inv;
      *reinterpret_cast<f32x4*>(dx + (long)row * cols + 4 * c) = o;
    }
  }
}

// one wave per row, 4 waves per block.  These kernels are pure HBM streams: they need ~72 KiB in flight per CU
// (MI355X_MICROARCH.md) and a 480-wide f32 row is 1.9 KiB, i.e. >= 32 resident waves per CU.  The backward also
// writes one [2][cols] partial row per block for dgamma / dbeta, hence its smaller cap.
int ln_blocks_cap(int rows, int cap) {
  int b = (rows + 3) / 4;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return b;
}
// backward: as many blocks as are resident at once (96 VGPRs at VPL 2 = 5 waves per SIMD, ~160 at VPL 4 = 3), never a
// second round
int ln_blocks(int rows, int cols) { return ln_blocks_cap(rows, cols <= 512 ? 1280 : (cols <= 1024 ? 768 : 512)); }
int ln_blocks_fwd(int rows) { return ln_blocks_cap(rows, 2048); }

template <int VPL, bool DYBF16, bool XBF16, bool ADD16 = false, bool POOL = false>
void launch_ln_bwd(const LnBwd& p, int blocks, size_t lds, hipStream_t st) {
  if (lds > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_kernel<VPL, DYBF16, XBF16, ADD16, POOL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ln_bwd_kernel<VPL, DYBF16, XBF16, ADD16, POOL>), dim3(blocks), dim3(256), lds, st, p);
}

}  // namespace

#define LN_DISPATCH_VPL(cols, CALL)                      \
  do {                                                   \
    const int nch_ = (cols) >> 2;                        \
    if (nch_ <= 64 * 2) { CALL(2); }                     \
    else if (nch_ <= 64 * 4) { CALL(4); }                \
    else if (nch_ <= 64 * 8) { CALL(8); }                \
    else if (nch_ <= 64 * 20) { CALL(20); }              \
    else return CLIPK_ERR_UNSUPPORTED;                   \
  } while (0)

extern "C" int clipk_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta,
                                   float eps, int act, float* y_f32, void* y_bf16, int64_t ldy,
                                   float* mean, float* rstd, int rows, int cols, void* stream) {
  if (!x || !gamma || !beta || rows <= 0 || cols <= 0 || (!y_f32 && !y_bf16)) return CLIPK_ERR_BAD_ARG;
  if ((cols & 3) || (ldx & 3) || (ldy & 3)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(gamma) || !aligned16(beta)) return CLIPK_ERR_BAD_ARG;
  LnFwd p{x, (long)ldx, gamma, beta, eps, act, y_f32, y_bf16, (long)ldy, mean, rstd, rows, cols};
  const int blocks = ln_blocks_fwd(rows);
  hipStream_t st = (hipStream_t)stream;
  if (cols > 2048 && cols <= 5120) {                        // wide rows: one workgroup per row (the choice depends on the
    const int wb = rows < 2048 ? rows : 2048;               // row WIDTH only: a row's arithmetic never depends on the batch)
    if (x_dtype == CLIPK_BF16) hipLaunchKernelGGL((ln_fwd_wide_kernel<5, true>), dim3(wb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((ln_fwd_wide_kernel<5, false>), dim3(wb), dim3(256), 0, st, p);
    return clipk_check_launch();
  }
#define CALL(V)                                                                                   \
  if (x_dtype == CLIPK_BF16) hipLaunchKernelGGL((ln_fwd_kernel<V, true>), dim3(blocks), dim3(256), 0, st, p); \
  else hipLaunchKernelGGL((ln_fwd_kernel<V, false>), dim3(blocks), dim3(256), 0, st, p)
  LN_DISPATCH_VPL(cols, CALL);
#undef CALL
  return clipk_check_launch();
}

extern "C" size_t clipk_layernorm_bwd_workspace(int rows, int cols) {
  return (size_t)ln_blocks(rows, cols) * 2 * cols * sizeof(float);
}

extern "C" int clipk_layernorm_bwd(const void* dy, int dy_dtype, int64_t lddy, const void* x, int x_dtype, int64_t ldx,
                                   const float* gamma, const float* beta, const float* mean, const float* rstd, int act,
                                   const void* dx_add, int dx_add_dtype, float* dx_f32, void* dx_bf16, int64_t lddx,
                                   float* dgamma, float* dbeta, int accumulate,
                                   int rows, int cols, float drop_p, uint32_t drop_seed,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || rows <= 0 || cols <= 0 || !workspace) return CLIPK_ERR_BAD_ARG;
  if (act != CLIPK_ACT_NONE && !beta) return CLIPK_ERR_BAD_ARG;
  if (!(drop_p >= 0.f) || drop_p >= 1.f) return CLIPK_ERR_BAD_ARG;
  if ((cols & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3)) return CLIPK_ERR_UNSUPPORTED;
  if (dx_add && dx_add_dtype == CLIPK_BF16 && !(dy_dtype == CLIPK_BF16 && x_dtype == CLIPK_F32)) return CLIPK_ERR_UNSUPPORTED;
  const int blocks = ln_blocks(rows, cols);
  if (workspace_bytes < (size_t)blocks * 2 * cols * sizeof(float)) return CLIPK_ERR_BAD_ARG;
  LnBwd p{dy, (long)lddy, x, (long)ldx, gamma, beta, mean, rstd, act, dx_add, dx_f32, dx_bf16, (long)lddx,
          (float*)workspace, rows, cols, 0u, drop_seed, 1.0f, (dx_add && dx_add_dtype == CLIPK_BF16) ? 1 : 0,
          nullptr, 1};
  p.drop_epoch = clipk_drop_epoch();
  if (drop_p > 0.f && drop_p < 1.f) {
    const double t = (double)drop_p * 4294967296.0;
    p.drop_thr = t < 1.0 ? 1u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
    p.drop_scale = 1.0f / (1.0f - drop_p);
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)4 * 2 * cols * sizeof(float);
  const bool wide = cols > 2048 && cols <= 5120 && dy_dtype == CLIPK_F32 && x_dtype == CLIPK_F32 && !p.add_bf16;
  if (wide)                                     // one workgroup per row (ln_bwd_wide_kernel); `blocks` partial rows as below
    hipLaunchKernelGGL((ln_bwd_wide_kernel<5>), dim3(blocks), dim3(256), 0, st, p);
#define CALL(V)                                                                                                   \
  do {                                                                                                            \
    if (wide) break;                                                                                              \
    if (p.add_bf16)                             /* bf16 gradient stream: bf16 dy, f32 x (checked above) */       \
      launch_ln_bwd<V, true, false, true>(p, blocks, lds, st);                                                    \
    else if (dy_dtype == CLIPK_BF16 && x_dtype == CLIPK_BF16)                                                     \
      launch_ln_bwd<V, true, true>(p, blocks, lds, st);                                                           \
    else if (dy_dtype == CLIPK_BF16)                                                                              \
      launch_ln_bwd<V, true, false>(p, blocks, lds, st);                                                          \
    else if (x_dtype == CLIPK_BF16)                                                                               \
      launch_ln_bwd<V, false, true>(p, blocks, lds, st);                                                          \
    else                                                                                                          \
      launch_ln_bwd<V, false, false>(p, blocks, lds, st);                                                         \
  } while (0)
  LN_DISPATCH_VPL(cols, CALL);
#undef CALL
  int rc = clipk_check_launch();
  if (rc) return rc;
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((2 * cols + 15) / 16), dim3(256), 0, st, (const float*)workspace,
                       blocks, 2 * cols, dgamma, dbeta, cols, accumulate);
    rc = clipk_check_launch();
  }
  return rc;
}

// out[c] (+)= sum over rows of x[r][c]: the bias gradient of an exact-f32 Linear (db = dY.sum(0)), fixed summation order
extern "C" int clipk_colsum_f32(const float* x, int rows, int cols, float* out, int accumulate, void* stream) {
  if (!x || !out || rows <= 0 || cols <= 0) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(colreduce_kernel, dim3((cols + 15) / 16), dim3(256), 0, (hipStream_t)stream, x, rows, cols, out,
                     (float*)nullptr, cols, accumulate);
  return clipk_check_launch();
}

// The deferred form of clipk_layernorm_bwd's parameter gradients: call it with dgamma = dbeta = NULL and a workspace of your
// own per LayerNorm (it keeps the [blocks][2][cols] partial rows, blocks = workspace bytes / (2 cols 4)), then reduce all of
// them here in one launch.  desc_dev: n x 6 int64 in device memory {partial rows, blocks, cols, dgamma, dbeta, accumulate}.
extern "C" int clipk_colreduce_batched(const int64_t* desc_dev, int n, int max_cols, void* stream) {
  if (!desc_dev || n <= 0 || max_cols <= 0 || n > 65535) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(colreduce_batched_kernel, dim3((2 * max_cols + 15) / 16, n), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(desc_dev));
  return clipk_check_launch();
}

// Second-order LayerNorm(+activation) backward (see ln_bwd2_kernel): f32 only, [rows, cols] row-major with one leading
// dimension.  workspace: clipk_layernorm_bwd_workspace(rows, cols) bytes.  d_gamma / d_beta are overwritten
// (accumulate = 0) or added to (accumulate = 1); either both or neither.
extern "C" int clipk_layernorm_bwd2(const float* g, const float* dy, const float* a, int64_t ld, const float* gamma,
                                    const float* beta, const float* mean, const float* rstd, int act, float* d_dy,
                                    float* d_a, float* d_gamma, float* d_beta, int accumulate, int rows, int cols,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !dy || !a || !gamma || !mean || !rstd || rows <= 0 || cols <= 0 || !workspace) return CLIPK_ERR_BAD_ARG;
  if (act != CLIPK_ACT_NONE && !beta) return CLIPK_ERR_BAD_ARG;
  if (act != CLIPK_ACT_NONE && act != CLIPK_ACT_CELU && act != CLIPK_ACT_SOFTPLUS) return CLIPK_ERR_UNSUPPORTED;
  if ((d_gamma == nullptr) != (d_beta == nullptr)) return CLIPK_ERR_BAD_ARG;
  if ((cols & 3) || (ld & 3)) return CLIPK_ERR_UNSUPPORTED;
  const int blocks = ln_blocks_cap(rows, 512);
  if (workspace_bytes < (size_t)blocks * 2 * cols * sizeof(float)) return CLIPK_ERR_BAD_ARG;
  LnBwd2 p{g, dy, a, (long)ld, gamma, beta, mean, rstd, act, d_dy, d_a, (float*)workspace, rows, cols};
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)4 * 2 * cols * sizeof(float);
#define LN_BWD2_CALL(V)                                                                                      \
  do {                                                                                                       \
    if (lds > 65536)                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd2_kernel<V>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    hipLaunchKernelGGL((ln_bwd2_kernel<V>), dim3(blocks), dim3(256), lds, st, p);                            \
  } while (0)
  const int nch_ = cols >> 2;
  if (nch_ <= 64 * 2) LN_BWD2_CALL(2);
  else if (nch_ <= 64 * 4) LN_BWD2_CALL(4);
  else if (nch_ <= 64 * 8) LN_BWD2_CALL(8);
  else return CLIPK_ERR_UNSUPPORTED;
#undef LN_BWD2_CALL
  if (d_gamma)
    hipLaunchKernelGGL(colreduce_kernel, dim3((2 * cols + 15) / 16), dim3(256), 0, st, (const float*)workspace,
                       blocks, 2 * cols, d_gamma, d_beta, cols, accumulate);
  return clipk_check_launch();
}

extern "C" int clipk_layernorm_meanpool_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, float eps,
                                           const uint8_t* mask, int B, int L, int cols, float* pooled, float* mean,
                                           float* rstd, float* row_weight, void* stream) {
  if (!x || !gamma || !beta || !pooled || !mean || !rstd || !row_weight || B <= 0 || L <= 0 || cols <= 0) return CLIPK_ERR_BAD_ARG;
  if ((cols & 3) || (ldx & 3)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(beta)) return CLIPK_ERR_BAD_ARG;
  LnPool p{x, (long)ldx, gamma, beta, eps, mask, pooled, mean, rstd, row_weight, L, cols};
  const size_t lds = ((size_t)4 * cols + 4) * sizeof(float);
  if (lds > 65536) return CLIPK_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define CALL(V)                                                                                          \
  do {                                                                                                   \
    if (x_dtype == CLIPK_BF16) hipLaunchKernelGGL((ln_pool_fwd_kernel<V, true>), dim3(B), dim3(256), lds, st, p);  \
    else hipLaunchKernelGGL((ln_pool_fwd_kernel<V, false>), dim3(B), dim3(256), lds, st, p);             \
  } while (0)
  LN_DISPATCH_VPL(cols, CALL);
#undef CALL
  return clipk_check_launch();
}

extern "C" int clipk_layernorm_meanpool_bwd(const float* dpooled, const float* row_weight, int B, int L,
                                           const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* mean,
                                           const float* rstd, float* dx_f32, void* dx_bf16, int64_t lddx, float* dgamma,
                                           float* dbeta, int accumulate, int cols, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (!dpooled || !row_weight || !x || !gamma || !mean || !rstd || B <= 0 || L <= 0 || cols <= 0 || !workspace)
    return CLIPK_ERR_BAD_ARG;
  if (!dx_f32 && !dx_bf16) return CLIPK_ERR_BAD_ARG;
  if ((cols & 3) || (ldx & 3) || (lddx & 3)) return CLIPK_ERR_UNSUPPORTED;
  const long rows_l = (long)B * L;
  if (rows_l > 0x7fffffffL) return CLIPK_ERR_UNSUPPORTED;
  const int rows = (int)rows_l;
  const int blocks = ln_blocks(rows, cols);
  if (workspace_bytes < (size_t)blocks * 2 * cols * sizeof(float)) return CLIPK_ERR_BAD_ARG;
  LnBwd p{dpooled, (long)cols, x, (long)ldx, gamma, nullptr, mean, rstd, CLIPK_ACT_NONE, nullptr, dx_f32, dx_bf16,
          (long)lddx, (float*)workspace, rows, cols, 0u, 0u, 1.0f, 0, row_weight, L};
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)4 * 2 * cols * sizeof(float);
#define CALL(V)                                                                          \
  do {                                                                                   \
    if (x_dtype == CLIPK_BF16) launch_ln_bwd<V, false, true, false, true>(p, blocks, lds, st);   \
    else launch_ln_bwd<V, false, false, false, true>(p, blocks, lds, st);                \
  } while (0)
  LN_DISPATCH_VPL(cols, CALL);
#undef CALL
  int rc = clipk_check_launch();
  if (rc) return rc;
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(colreduce_kernel, dim3((2 * cols + 15) / 16), dim3(256), 0, st, (const float*)workspace,
                       blocks, 2 * cols, dgamma, dbeta, cols, accumulate);
    rc = clipk_check_launch();
  }
  return rc;
}

extern "C" int clipk_l2norm_fwd(const float* x, float* y, float* norm, int rows, int cols, float eps, void* stream) {
  if (!x || !y || rows <= 0 || cols <= 0) return CLIPK_ERR_BAD_ARG;
  if ((cols & 3) || cols > 64 * 4 * 8) return CLIPK_ERR_UNSUPPORTED;
  const int blocks = (rows + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  if (cols <= 512) hipLaunchKernelGGL((l2norm_fwd_kernel<2>), dim3(blocks), dim3(256), 0, st, x, y, norm, rows, cols, eps);
  else hipLaunchKernelGGL((l2norm_fwd_kernel<8>), dim3(blocks), dim3(256), 0, st, x, y, norm, rows, cols, eps);
  return clipk_check_launch();
}

extern "C" int clipk_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx,
                                int rows, int cols, float eps, void* stream) {
  if (!dy || !y || !norm || !dx || rows <= 0 || cols <= 0) return CLIPK_ERR_BAD_ARG;
  if ((cols & 3) || cols > 64 * 4 * 8) return CLIPK_ERR_UNSUPPORTED;
  const int blocks = (rows + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  if (cols <= 512) hipLaunchKernelGGL((l2norm_bwd_kernel<2>), dim3(blocks), dim3(256), 0, st, dy, y, norm, dx, rows, cols, eps);
  else hipLaunchKernelGGL((l2norm_bwd_kernel<8>), dim3(blocks), dim3(256), 0, st, dy, y, norm, dx, rows, cols, eps);
  return clipk_check_launch();
}
