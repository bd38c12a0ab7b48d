// celogits.hip — the reference's module API hands MATERIALISED logits to its loss functions
// (old/clip.py:66-73 returns "logits_per_rna_protein"; old/ablation.py:16 and old/clip_opt.py:130-151 apply
// F.cross_entropy(logits, arange) to them).  Training should use the fused path (simce.hip, no logits in HBM); these
// kernels keep the drop-in API itself on the HIP path:
//   * ce_diag_lse_{row,col}: logsumexp over the rows / the columns of S (optionally [S | S2], the cache columns of
//     old/clip_opt.py:136) and the diagonal "positive" logit;
//   * ce_diag_bwd: dS = g * ( w_row (softmax_row - onehot) / M  +  w_col (softmax_col - onehot) / M ), dS2 alike;
//   * transpose_scale_f32: out[C, R] = s * in[R, C]^T (operands of the exact-f32 products of SimLogitsFn.backward and
//     of the ICNN's transposed weights).
// All HBM-bound, f32, coalesced along the fastest dimension; no atomics.
#include "common.h"
#include <math.h>

namespace {

// one wave per row i: lse[i] = logsumexp_j [S | S2][i, j], pos[i] = S[i, i + label_offset]
__global__ __launch_bounds__(256) void ce_lse_row_kernel(const float* S, long ld, int M, int N, const float* S2, long ld2,
                                                         int N2, int label_offset, float* lse, float* pos) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* r = S + (long)row * ld;
  float m = -INFINITY, l = 0.f;
  auto upd = [&](float v) {
    if (v > m) { l = l * __expf(m - v) + 1.f; m = v; }
    else l += __expf(v - m);
  };
  for (int j = lane; j < N; j += 64) upd(r[j]);
  if (S2) {
    const float* r2 = S2 + (long)row * ld2;
    for (int j = lane; j < N2; j += 64) upd(r2[j]);
  }
  // merge the 64 (m, l) pairs
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64);
    const float mn = fmaxf(m, m2);
    const float a = (m == -INFINITY) ? 0.f : l * __expf(m - mn);
    const float b = (m2 == -INFINITY) ? 0.f : l2 * __expf(m2 - mn);
    m = mn; l = a + b;
  }
  if (lane == 0) {
    lse[row] = m + __logf(l);
    const int lab = row + label_offset;
    pos[row] = (lab >= 0 && lab < N) ? r[lab] : 0.f;
  }
}

// columns: block = 64 columns x 4 row slices; lse[j] = logsumexp_i S[i, j], pos[j] = S[j + label_offset, j]
__global__ __launch_bounds__(256) void ce_lse_col_kernel(const float* S, long ld, int M, int N, int label_offset,
                                                         float* lse, float* pos) {
  __shared__ float sm[4][64], sl[4][64];
  const int c = threadIdx.x & 63, sl_id = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  float m = -INFINITY, l = 0.f;
  if (col < N) {
    for (int i = sl_id; i < M; i += 4) {
      const float v = S[(long)i * ld + col];
      if (v > m) { l = l * __expf(m - v) + 1.f; m = v; }
      else l += __expf(v - m);
    }
  }
  sm[sl_id][c] = m; sl[sl_id][c] = l;
  __syncthreads();
  if (sl_id == 0 && col < N) {
    float mm = sm[0][c], ll = sl[0][c];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float m2 = sm[k][c], l2 = sl[k][c];
      const float mn = fmaxf(mm, m2);
      const float a = (mm == -INFINITY) ? 0.f : ll * __expf(mm - mn);
      const float b = (m2 == -INFINITY) ? 0.f : l2 * __expf(m2 - mn);
      mm = mn; ll = a + b;
    }
    lse[col] = mm + __logf(ll);
    const int lab = col + label_offset;
    pos[col] = (lab >= 0 && lab < M) ? S[(long)lab * ld + col] : 0.f;
  }
}

// dS[i, j] = g * ( wr * (exp(S_ij - lse_r[i]) - [j == i + off]) + wc * (exp(S_ij - lse_c[j]) - [i == j + offc]) );
// dS2[i, j] = g * wr * exp(S2_ij - lse_r[i]).  wr / wc already carry the 1 / batch factors.
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* S, long ld, int M, int N, const float* S2, long ld2, int N2,
                                                     const float* lse_r, const float* lse_c, float wr, float wc,
                                                     int off_r, int off_c, const float* g, float* dS, long ldd,
                                                     float* dS2, long ldd2) {
  const int row = blockIdx.y;
  const float gv = g[0];
  const float lr = lse_r ? lse_r[row] : 0.f;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < N) {
    const float v = S[(long)row * ld + j];
    float d = 0.f;
    if (lse_r) d += wr * (__expf(v - lr) - ((j == row + off_r) ? 1.f : 0.f));
    if (lse_c) d += wc * (__expf(v - lse_c[j]) - ((row == j + off_c) ? 1.f : 0.f));
    dS[(long)row * ldd + j] = gv * d;
  }
  if (S2 && dS2 && j < N2) {
    const float v = S2[(long)row * ld2 + j];
    dS2[(long)row * ldd2 + j] = lse_r ? gv * wr * __expf(v - lr) : 0.f;
  }
}

// out[C, R] = s * in[R, C]^T through a 32 x 33 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void transpose_scale_kernel(const float* in, int R, int C, const float* s, float* out) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const float sv = s ? s[0] : 1.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? in[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < R) out[(long)c * R + r] = sv * tile[tx][ty + 8 * k];
  }
}

}  // namespace

extern "C" int clipk_ce_logits_lse(const float* S, int64_t ld, int M, int N, const float* S2, int64_t ld2, int N2,
                                   int columns, int label_offset, float* lse, float* pos, void* stream) {
  if (!S || !lse || !pos || M <= 0 || N <= 0 || N2 < 0 || (N2 > 0 && !S2)) return CLIPK_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (columns) {
    if (N2 > 0) return CLIPK_ERR_UNSUPPORTED;              // cache columns exist in the row direction only
    hipLaunchKernelGGL(ce_lse_col_kernel, dim3((N + 63) / 64), dim3(256), 0, st, S, (long)ld, M, N, label_offset, lse, pos);
  } else {
    hipLaunchKernelGGL(ce_lse_row_kernel, dim3((M + 3) / 4), dim3(256), 0, st, S, (long)ld, M, N, N2 > 0 ? S2 : nullptr,
                       (long)ld2, N2, label_offset, lse, pos);
  }
  return clipk_check_launch();
}

extern "C" int clipk_ce_logits_bwd(const float* S, int64_t ld, int M, int N, const float* S2, int64_t ld2, int N2,
                                   const float* lse_row, const float* lse_col, float w_row, float w_col,
                                   int label_offset_row, int label_offset_col, const float* gscale,
                                   float* dS, int64_t ldd, float* dS2, int64_t ldd2, void* stream) {
  if (!S || !dS || !gscale || M <= 0 || N <= 0 || N2 < 0 || (N2 > 0 && (!S2 || !dS2))) return CLIPK_ERR_BAD_ARG;
  const int nmax = N > N2 ? N : N2;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3((nmax + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, S, (long)ld, M, N,
                     N2 > 0 ? S2 : nullptr, (long)ld2, N2, lse_row, lse_col, w_row, w_col, label_offset_row,
                     label_offset_col, gscale, dS, (long)ldd, N2 > 0 ? dS2 : nullptr, (long)ldd2);
  return clipk_check_launch();
}

extern "C" int clipk_transpose_scale_f32(const float* in, int rows, int cols, const float* scale_dev, float* out,
                                         void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(transpose_scale_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                     in, rows, cols, scale_dev, out);
  return clipk_check_launch();
}
