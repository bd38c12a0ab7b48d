// elementwise.hip — HBM-bound elementwise / gather / optimiser kernels (16-byte accesses, grid-stride).
// Reference ops: dtype casts done by autocast (old/clip_opt.py:163), F.relu / nn.GELU (old/clip.py:16,29),
// skip + layer_scale * projected (old/clip_opt.py:41-44), EsmEmbeddings (modeling_esm.py:203-270),
// position-0 / mean pooling (rna_clip_codes.ipynb:1948; configuration_hybrid_clip.py:109),
// AdamW + clip_grad_norm_ (rna_clip_codes.ipynb:2033,2076; old/clip_opt.py:168-171).
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;
inline int ew_blocks(long n_items) {
  long b = (n_items + EW_THREADS - 1) / EW_THREADS;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ void cast_f32_bf16_kernel(const float* x, unsigned short* y, long n) {
  const long n8 = n >> 3;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += stride) {
    const f32x4 a = reinterpret_cast<const f32x4*>(x)[2 * i], b = reinterpret_cast<const f32x4*>(x)[2 * i + 1];
    u32x4 o = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    reinterpret_cast<u32x4*>(y)[i] = o;
  }
  for (long i = (n8 << 3) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = f32_to_bf16(x[i]);
}

__global__ void cast_bf16_f32_kernel(const unsigned short* x, float* y, long n) {
  const long n8 = n >> 3;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += stride) {
    const u32x4 a = reinterpret_cast<const u32x4*>(x)[i];
    f32x4 lo, hi;
    lo[0] = bf16_to_f32(a[0] & 0xffffu); lo[1] = bf16_to_f32(a[0] >> 16);
    lo[2] = bf16_to_f32(a[1] & 0xffffu); lo[3] = bf16_to_f32(a[1] >> 16);
    hi[0] = bf16_to_f32(a[2] & 0xffffu); hi[1] = bf16_to_f32(a[2] >> 16);
    hi[2] = bf16_to_f32(a[3] & 0xffffu); hi[3] = bf16_to_f32(a[3] >> 16);
    reinterpret_cast<f32x4*>(y)[2 * i] = lo;
    reinterpret_cast<f32x4*>(y)[2 * i + 1] = hi;
  }
  for (long i = (n8 << 3) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = bf16_to_f32(x[i]);
}

// 64x64 tile through LDS: coalesced f32 reads along cols, coalesced bf16 writes along rows of W^T
// (il_hd, il_rows): rows [0, il_rows) of BOTH copies in pair-interleaved head order (common.h il_src): copy row r = w row il_src(r)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* w, unsigned short* wb, unsigned short* wt,
                                                             int rows, int cols, int il_hd, int il_rows) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = w[(long)il_src(r, il_hd, il_rows) * cols + c];
      if (wb) wb[(long)r * cols + c] = f32_to_bf16(v);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (wt) {
    for (int i = ty; i < 64; i += 4) {
      const int c = c0 + i, r = r0 + tx;
      if (r < rows && c < cols) wt[(long)c * rows + r] = f32_to_bf16(tile[tx][i]);
    }
  }
}

// every weight of a model in ONE launch (the optimiser step refreshes all bf16 copies at once: 76 launches of
// ~12 us each otherwise).  desc[t] = {w, w_bf16, wt_bf16, rows, cols, il_hd, il_rows} as int64; blockIdx.y = tensor, blocks
// stride over its 64x64 tiles.
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(const long long* desc) {
  __shared__ float tile[64][65];
  const long long* d = desc + 7 * (long)blockIdx.y;
  const float* w = reinterpret_cast<const float*>(d[0]);
  unsigned short* wb = reinterpret_cast<unsigned short*>(d[1]);
  unsigned short* wt = reinterpret_cast<unsigned short*>(d[2]);
  const int rows = (int)d[3], cols = (int)d[4], il_hd = (int)d[5], il_rows = (int)d[6];
  const int tx_n = (cols + 63) / 64, ntiles = tx_n * ((rows + 63) / 64);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int r0 = (t / tx_n) * 64, c0 = (t % tx_n) * 64;
    for (int i = ty; i < 64; i += 4) {
      const int r = r0 + i, c = c0 + tx;
      float v = 0.f;
      if (r < rows && c < cols) {
        v = w[(long)il_src(r, il_hd, il_rows) * cols + c];
        if (wb) wb[(long)r * cols + c] = f32_to_bf16(v);
      }
      tile[i][tx] = v;
    }
    __syncthreads();
    if (wt) {
      for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (r < rows && c < cols) wt[(long)c * rows + r] = f32_to_bf16(tile[tx][i]);
      }
    }
    __syncthreads();
  }
}

__global__ void act_fwd_kernel(const float* x, float* y, int act, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = act_apply(x[i], act);
}
__global__ void act_bwd_kernel(const float* dy, const float* x, float* dx, int act, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) dx[i] = dy[i] * act_grad(x[i], act);
}
__global__ void axpby_dev_kernel(const float* a, const float* b, const float* s, float* y, long n) {
  const float sc = s[0];
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) y[i] = a ? a[i] + sc * b[i] : sc * b[i];
}

// y = x * keep(seed, i) / (1 - p) (+ addend): nn.Dropout on an f32 tensor with the kernels' counter-based mask (index =
// the element's row-major position), optionally followed by the residual add it sits in front of
__global__ void dropout_f32_kernel(const float* x, const float* addend, float* y, long n, unsigned thr, unsigned seed,
                                   float scale, const unsigned* epoch) {
  seed = drop_seed_eff(seed, epoch);
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i] * drop_mul(seed, (unsigned long long)i, thr, scale);
    y[i] = addend ? v + addend[i] : v;
  }
}

// out_bf16 = dy * act'(aux): the activation backward between two Linear layers (dy f32 or bf16)
template <bool DYBF16>
__global__ void dact_kernel(const void* dy, const unsigned short* aux, int act, unsigned short* out, long n) {
  const long n8 = n >> 3;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += stride) {
    float d[8];
    if (DYBF16) {
      const u32x4 a = reinterpret_cast<const u32x4*>(dy)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[2 * e] = bf16_to_f32(a[e] & 0xffffu); d[2 * e + 1] = bf16_to_f32(a[e] >> 16); }
    } else {
      const f32x4 a = reinterpret_cast<const f32x4*>(dy)[2 * i], b = reinterpret_cast<const f32x4*>(dy)[2 * i + 1];
#pragma unroll
      for (int e = 0; e < 4; ++e) { d[e] = a[e]; d[4 + e] = b[e]; }
    }
    const u32x4 x = reinterpret_cast<const u32x4*>(aux)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = pack_bf16x2(d[2 * e] * act_grad(bf16_to_f32(x[e] & 0xffffu), act),
                         d[2 * e + 1] * act_grad(bf16_to_f32(x[e] >> 16), act));
    reinterpret_cast<u32x4*>(out)[i] = o;
  }
  for (long i = (n8 << 3) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = DYBF16 ? bf16_to_f32(reinterpret_cast<const unsigned short*>(dy)[i]) : reinterpret_cast<const float*>(dy)[i];
    out[i] = f32_to_bf16(d * act_grad(bf16_to_f32(aux[i]), act));
  }
}

// ---- embedding ----------------------------------------------------------------------------------
__global__ void embed_fwd_kernel(const int64_t* ids, const float* table, const float* row_scale, const uint8_t* mask,
                                 int mask_token_id, float* x, int B, int L, int d, int V) {
  const int nch = d >> 2;
  const long total = (long)B * L * nch;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long t = i / nch; const int c = (int)(i - t * nch);
    const int64_t id = ids[t];
    float sc = row_scale ? row_scale[t / L] : 1.0f;
    if (mask && !mask[t]) sc = 0.f;
    if ((int)id == mask_token_id) sc = 0.f;
    // an id outside the table never reads memory: its row comes out as NaN, so the mistake surfaces in the loss
    // (F.embedding would trip a device assert; a silent wild read is the one thing that must not happen)
    const bool ok = id >= 0 && id < V;
    const float nanv = __uint_as_float(0x7fc00000u);
    f32x4 v = ok ? *reinterpret_cast<const f32x4*>(table + id * d + 4 * c) : f32x4{nanv, nanv, nanv, nanv};
    *reinterpret_cast<f32x4*>(x + t * d + 4 * c) = ok ? v * sc : v;
  }
}

// dtable[v,:] += sum over tokens with ids == v.  Per-block private [V][dc] table in LDS (ds_add_f32) for a chunk
// of dc columns (blockIdx.y), then one global atomic per (block, v, column).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* ids, const float* dx, const float* row_scale,
                                                        const uint8_t* mask, int mask_token_id, float* dtable,
                                                        int B, int L, int d, int V, int dc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  const int c0 = blockIdx.y * dc;
  const int ncol = (c0 + dc <= d) ? dc : d - c0;
  for (int i = threadIdx.x; i < V * dc; i += blockDim.x) tab[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  const long T = (long)B * L;
  for (long t = wave; t < T; t += nwaves) {
    const int id = (int)ids[t];
    float sc = row_scale ? row_scale[t / L] : 1.0f;
    if (mask && !mask[t]) sc = 0.f;
    if (id == mask_token_id) sc = 0.f;
    if (sc == 0.f) continue;
    for (int c = lane; c < ncol; c += 64) atomicAdd(&tab[id * dc + c], dx[t * d + c0 + c] * sc);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < V * dc; i += blockDim.x) {
    const int v = i / dc, c = i - v * dc;
    const float val = tab[i];
    if (c < ncol && val != 0.f) atomicAdd(&dtable[(long)v * d + c0 + c], val);
  }
}

// ---- deterministic embedding gradient for small vocabularies (V <= 64: ESM-2 has 33 tokens) --------------------
// dtable[V][d] += onehot(ids)^T [V x T] . (dx * sc) [T x d] on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32):
// the one-hot operand is built in registers from the ids, dx streams in as one float4 per lane and token (512 B per
// token and wave), a wave owns 128 columns x a slice of tokens.  Slices are combined in a fixed order (4 waves
// through LDS, then the slice partials by embed_bwd_reduce_kernel): no atomics, bitwise reproducible, and ~5x
// faster than the LDS-atomic table below, which 33 hot rows serialise.
constexpr int EB_COLS = 128;                             // columns per workgroup (4 interleaved strips of 32)
__global__ __launch_bounds__(256, 3) void embed_bwd_mfma_kernel(const int64_t* ids, const float* dx, const float* row_scale,
                                                             const uint8_t* mask, int mask_token_id, float* part,
                                                             int B, int L, int d, int V, int tok_per_slice) {
  __shared__ float red[3][64][33];                       // waves 1..3 hand their accumulators to wave 0, one at a time
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int cg = blockIdx.y, slice = blockIdx.x;
  const long T = (long)B * L;
  const long t_beg = (long)slice * tok_per_slice;
  long t_end = t_beg + tok_per_slice; t_end = t_end < T ? t_end : T;
  const int c0 = cg * EB_COLS + 4 * (lane & 31);         // this lane's 4 columns: strip j owns column c0 + j
  const bool col_ok = c0 < d;                            // d % 4 == 0
  const int kk = lane >> 5;                              // which token of the pair this lane feeds (MFMA k index)
  const int vrow = lane & 31;
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  f32x16 acc[2][4];
#pragma unroll
  for (int vt = 0; vt < 2; ++vt)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[vt][j][r] = 0.f;
  // wave w takes token pairs w, w+4, ... of the slice, four pairs per iteration so that their loads are in flight
  // together (the MFMAs of a pair depend on its load)
  constexpr int U = 4;
  for (long t0 = t_beg + 2 * wid; t0 < t_end; t0 += 8 * U) {
    f32x4 v[U];
    float a0[U], a1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long t = t0 + 8 * u + kk;
      const bool tok_ok = t < t_end;
      const unsigned tc = (unsigned)(tok_ok ? t : t_beg);      // B * L < 2^31 (launcher)
      const int id = (int)ids[tc];
      float sc = row_scale ? row_scale[tc / (unsigned)L] : 1.0f;
      if (mask && !mask[tc]) sc = 0.f;
      if (id == mask_token_id || !tok_ok) sc = 0.f;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (col_ok) v[u] = *reinterpret_cast<const f32x4*>(dx + (long)tc * d + c0);
      a0[u] = (id == vrow) ? sc : 0.f;
      a1[u] = (id == vrow + 32) ? sc : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], v[u][j], acc[0][j], 0, 0, 0);
        if (V > 32) acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], v[u][j], acc[1][j], 0, 0, 0);
      }
  }
  // C layout: lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r = 0..15
  float* out = part + ((long)slice * V) * d;
  const int nvt = V > 32 ? 2 : 1;
#pragma unroll
  for (int vt = 0; vt < 2; ++vt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (vt >= nvt) break;                              // block-uniform
      // fixed-order sum over the 4 waves: waves 1..3 write, wave 0 adds in order 1, 2, 3
      __syncthreads();
      if (wid > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wid - 1][lane][r] = acc[vt][j][r];
      }
      __syncthreads();
      if (wid == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sum = ((acc[vt][j][r] + red[0][lane][r]) + red[1][lane][r]) + red[2][lane][r];
          const int row = vt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = cg * EB_COLS + 4 * (lane & 31) + j;
          if (row < V && col < d) out[(long)row * d + col] = sum;
        }
      }
    }
}
// dtable[i] += sum over slices of part[s][i], fixed order.  Block = 16 elements x 16 slice groups, 8 loads in flight
// per thread (a thread per element with a serial loop over 256 slices is a 100-us latency chain).
__global__ __launch_bounds__(256) void embed_bwd_reduce_kernel(const float* part, int nslices, long n, float* dtable) {
  __shared__ float sm[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + c;
  float a = 0.f;
  if (i < n) {
    int s = rg;
    for (; s + 7 * 16 < nslices; s += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(s + 16 * u) * n + i];
      a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; s < nslices; s += 16) a += part[(long)s * n + i];
  }
  sm[rg][c] = a;
  __syncthreads();
  if (rg == 0 && i < n) {
    a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += sm[r][c];
    dtable[i] += a;
  }
}

// ---- pooling ------------------------------------------------------------------------------------
// y[b, c] = mean over valid positions of x[b, :, c] (mode 1) or x[b, 0, c] (mode 0).  One 64-thread block per
// (b, 64 columns) walks the L positions with 8 independent loads in flight (256-thread blocks with a plain serial
// loop ran at 1.6 TB/s: 2 blocks per sequence, one load in flight per thread).
__global__ void pool_fwd_kernel(const float* x, const uint8_t* mask, float* y, int B, int L, int d, int mode) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const float* xb = x + ((long)b * L) * d + c;
  if (mode == 0) { y[(long)b * d + c] = xb[0]; return; }
  const uint8_t* mb = mask ? mask + (long)b * L : nullptr;
  float s = 0.f; int n = 0;
  int l = 0;
  for (; l + 8 <= L; l += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = xb[(long)(l + u) * d];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool ok = mb ? mb[l + u] != 0 : true;
      if (ok) { s += v[u]; ++n; }
    }
  }
  for (; l < L; ++l) {
    const bool ok = mb ? mb[l] != 0 : true;
    if (ok) { s += xb[(long)l * d]; ++n; }
  }
  y[(long)b * d + c] = n > 0 ? s / (float)n : 0.f;
}
// packed variable-length batches: sequence b = rows [cu[b], cu[b+1]); mode 0 = its first row, 1 = mean over its rows
__global__ void pool_varlen_fwd_kernel(const float* x, const int* cu, float* y, int d, int mode) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const int r0 = cu[b], n = cu[b + 1] - r0;
  const float* xb = x + (long)r0 * d + c;
  if (mode == 0 || n <= 0) { y[(long)b * d + c] = n > 0 ? xb[0] : 0.f; return; }
  float s = 0.f;
  int l = 0;
  for (; l + 8 <= n; l += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = xb[(long)(l + u) * d];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; l < n; ++l) s += xb[(long)l * d];
  y[(long)b * d + c] = s / (float)n;
}
// one block row per sequence: dx[rows of b] = dy[b] / n (mode 1) or dy[b] on the first row only (mode 0)
__global__ void pool_varlen_bwd_kernel(const float* dy, const int* cu, float* dx, int d, int mode) {
  const int b = blockIdx.y;
  const int r0 = cu[b], n = cu[b + 1] - r0;
  const int nch = d >> 2;
  const long total = (long)n * nch;
  const float inv = (mode == 1 && n > 0) ? 1.0f / (float)n : 1.0f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int l = (int)(i / nch), c = (int)(i - (long)l * nch);
    const float sc = (mode == 0) ? (l == 0 ? 1.f : 0.f) : inv;
    const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (long)b * d + 4 * c);
    *reinterpret_cast<f32x4*>(dx + ((long)r0 + l) * d + 4 * c) = v * sc;
  }
}
__global__ void pool_bwd_kernel(const float* dy, const uint8_t* mask, float* dx, int B, int L, int d, int mode) {
  const int nch = d >> 2;
  const long total = (long)B * L * nch;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long t = i / nch; const int c = (int)(i - t * nch);
    const int b = (int)(t / L), l = (int)(t - (long)b * L);
    float sc;
    if (mode == 0) sc = (l == 0) ? 1.f : 0.f;
    else {
      // valid-token count of this sequence (L is small; recomputed per thread from the L1-resident mask)
      int n = L;
      if (mask) { n = 0; for (int k = 0; k < L; ++k) n += mask[(long)b * L + k] != 0; }
      const bool ok = mask ? mask[t] != 0 : true;
      sc = (ok && n > 0) ? 1.0f / (float)n : 0.f;
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(dy + (long)b * d + 4 * c);
    *reinterpret_cast<f32x4*>(dx + t * d + 4 * c) = v * sc;
  }
}

// ---- optimiser ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long n, float* part) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* part, int nparts, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void adamw_kernel(float* w, const float* g, float* m, float* v, unsigned short* wb, long n, float lr,
                             float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                             const float* gnsq, float max_norm, float grad_scale, const float* hyper) {
  if (hyper) {                    // {lr, 1 - beta1^t, sqrt(1 - beta2^t)} from device memory: a step replayed from a hipGraph
    lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2];
  }
  float clip = grad_scale;
  if (gnsq) {
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    const float norm = sqrtf(gnsq[0]) * grad_scale;
    const float coef = max_norm / (norm + 1e-6f);
    clip *= coef < 1.0f ? coef : 1.0f;
  }
  const long stride = (long)gridDim.x * blockDim.x;
  const float step = lr / bc1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * clip;
    float wi = w[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    wi -= step * (mi / denom);
    w[i] = wi;
    if (wb) wb[i] = f32_to_bf16(wi);
  }
}

}  // namespace

extern "C" int clipk_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0 || !aligned16(x) || !aligned16(y)) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, x,
                     (unsigned short*)y, (long)n);
  return clipk_check_launch();
}
extern "C" int clipk_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0 || !aligned16(x) || !aligned16(y)) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const unsigned short*)x, y, (long)n);
  return clipk_check_launch();
}
extern "C" int clipk_cast_transpose(const float* w, void* w_bf16, void* wt_bf16, int rows, int cols, int il_hd, int il_rows,
                                    void* stream) {
  if (!w || rows <= 0 || cols <= 0 || (!w_bf16 && !wt_bf16)) return CLIPK_ERR_BAD_ARG;
  if (il_rows < 0 || il_rows > rows || (il_rows > 0 && (il_hd < 2 || (il_hd & 1) || il_rows % il_hd))) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     w, (unsigned short*)w_bf16, (unsigned short*)wt_bf16, rows, cols, il_rows ? il_hd : 2, il_rows);
  return clipk_check_launch();
}
extern "C" int clipk_cast_transpose_batched(const void* desc_dev, int n, void* stream) {
  if (!desc_dev || n <= 0 || n > 65535) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(32, n), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)desc_dev);
  return clipk_check_launch();
}
extern "C" int clipk_act_fwd(const float* x, float* y, int act, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, y, act, (long)n);
  return clipk_check_launch();
}
extern "C" int clipk_act_bwd(const float* dy, const float* x, float* dx, int act, int64_t n, void* stream) {
  if (!dy || !x || !dx || n <= 0) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, dy, x, dx, act, (long)n);
  return clipk_check_launch();
}
extern "C" int clipk_dact(const void* dy, int dy_dtype, const void* aux_bf16, int act, void* out_bf16, int64_t n,
                          void* stream) {
  if (!dy || !aux_bf16 || !out_bf16 || n <= 0 || !aligned16(dy) || !aligned16(aux_bf16) || !aligned16(out_bf16))
    return CLIPK_ERR_BAD_ARG;
  if (dy_dtype == CLIPK_BF16)
    hipLaunchKernelGGL(dact_kernel<true>, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, dy,
                       (const unsigned short*)aux_bf16, act, (unsigned short*)out_bf16, (long)n);
  else
    hipLaunchKernelGGL(dact_kernel<false>, dim3(ew_blocks(n / 8 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream, dy,
                       (const unsigned short*)aux_bf16, act, (unsigned short*)out_bf16, (long)n);
  return clipk_check_launch();
}
extern "C" int clipk_axpby_dev(const float* a, const float* b, const float* s, float* y, int64_t n, void* stream) {
  if (!b || !s || !y || n <= 0) return CLIPK_ERR_BAD_ARG;        // a == NULL: y = s * b
  hipLaunchKernelGGL(axpby_dev_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, a, b, s, y, (long)n);
  return clipk_check_launch();
}

extern "C" int clipk_dropout_f32(const float* x, const float* addend, float* y, int64_t n, float p, uint32_t seed,
                                 void* stream) {
  if (!x || !y || n <= 0 || !(p >= 0.f) || p >= 1.f) return CLIPK_ERR_BAD_ARG;
  const double t = (double)p * 4294967296.0;
  const unsigned thr = p == 0.f ? 0u : (t < 1.0 ? 1u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t));
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, addend, y,
                     (long)n, thr, seed, 1.0f / (1.0f - p), clipk_drop_epoch());
  return clipk_check_launch();
}

extern "C" int clipk_embed_fwd(const int64_t* ids, const float* table, const float* row_scale, const uint8_t* mask,
                               int mask_token_id, float* x, int B, int L, int d, int V, void* stream) {
  if (!ids || !table || !x || B <= 0 || L <= 0 || d <= 0 || V <= 0) return CLIPK_ERR_BAD_ARG;
  if (d & 3) return CLIPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(ew_blocks((long)B * L * (d / 4))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     ids, table, row_scale, mask, mask_token_id, x, B, L, d, V);
  return clipk_check_launch();
}
static int embed_bwd_slices(int B, int L) {
  long T = (long)B * L;
  int s = (int)((T + 511) / 512);                            // >= 512 tokens per slice, ~4 workgroups per CU at d = 480
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return s;
}
extern "C" size_t clipk_embed_bwd_workspace(int B, int L, int d, int V) {
  if (B <= 0 || L <= 0 || d <= 0 || V <= 0 || V > 64) return 0;   // larger vocabularies use the LDS-table kernel
  return (size_t)embed_bwd_slices(B, L) * V * d * sizeof(float);
}
extern "C" int clipk_embed_bwd(const int64_t* ids, const float* dx, const float* row_scale, const uint8_t* mask,
                               int mask_token_id, float* dtable, int B, int L, int d, int V, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (!ids || !dx || !dtable || B <= 0 || L <= 0 || d <= 0 || V <= 0) return CLIPK_ERR_BAD_ARG;
  if (V <= 64 && (d & 3) == 0 && workspace && aligned16(dx)) {
    const int slices = embed_bwd_slices(B, L);
    if (workspace_bytes < (size_t)slices * V * d * sizeof(float)) return CLIPK_ERR_BAD_ARG;
    const long T = (long)B * L;
    int tps = (int)((T + slices - 1) / slices);
    tps = (tps + 31) & ~31;                                  // whole iterations of 4 waves x 4 pairs
    if ((long)B * L >= (1L << 31)) return CLIPK_ERR_UNSUPPORTED;
    const int ncg = (d + EB_COLS - 1) / EB_COLS;
    hipLaunchKernelGGL(embed_bwd_mfma_kernel, dim3(slices, ncg), dim3(256), 0, (hipStream_t)stream, ids, dx, row_scale,
                       mask, mask_token_id, (float*)workspace, B, L, d, V, tps);
    const long n = (long)V * d;
    hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3((int)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, slices, n, dtable);
    return clipk_check_launch();
  }
  // columns per block so that the private table is <= 16 KiB: the kernel streams B*L*d floats and needs many
  // resident blocks; small vocabularies keep >= 64 columns
  int dc = (16 * 1024) / (V * (int)sizeof(float));
  if (dc < 64) dc = (48 * 1024) / (V * (int)sizeof(float));
  dc &= ~3;
  if (dc < 4) return CLIPK_ERR_UNSUPPORTED;                  // vocabulary too large for the LDS-table scheme
  if (dc > d) dc = d;
  const size_t lds = (size_t)V * dc * sizeof(float);
  const int nchunks = (d + dc - 1) / dc;
  long waves = (long)B * L;
  int blocks = (int)((waves + 3) / 4); if (blocks > 1024 / nchunks + 1) blocks = 1024 / nchunks + 1; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(blocks, nchunks), dim3(256), lds, (hipStream_t)stream, ids, dx, row_scale, mask,
                     mask_token_id, dtable, B, L, d, V, dc);
  return clipk_check_launch();
}
extern "C" int clipk_pool_fwd(const float* x, const uint8_t* mask, float* y, int B, int L, int d, int mode, void* stream) {
  if (!x || !y || B <= 0 || L <= 0 || d <= 0) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3((d + 63) / 64, B), dim3(64), 0, (hipStream_t)stream, x, mask, y, B, L, d, mode);
  return clipk_check_launch();
}
extern "C" int clipk_pool_bwd(const float* dy, const uint8_t* mask, float* dx, int B, int L, int d, int mode, void* stream) {
  if (!dy || !dx || B <= 0 || L <= 0 || d <= 0) return CLIPK_ERR_BAD_ARG;
  if (d & 3) return CLIPK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(ew_blocks((long)B * L * (d / 4))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     dy, mask, dx, B, L, d, mode);
  return clipk_check_launch();
}

extern "C" int clipk_pool_varlen_fwd(const float* x, const int* cu_seqlens, float* y, int B, int d, int mode, void* stream) {
  if (!x || !cu_seqlens || !y || B <= 0 || d <= 0) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(pool_varlen_fwd_kernel, dim3((d + 63) / 64, B), dim3(64), 0, (hipStream_t)stream, x, cu_seqlens, y, d, mode);
  return clipk_check_launch();
}
extern "C" int clipk_pool_varlen_bwd(const float* dy, const int* cu_seqlens, float* dx, int B, int d, int mode, void* stream) {
  if (!dy || !cu_seqlens || !dx || B <= 0 || d <= 0 || (d & 3)) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(pool_varlen_bwd_kernel, dim3(8, B), dim3(256), 0, (hipStream_t)stream, dy, cu_seqlens, dx, d, mode);
  return clipk_check_launch();
}

extern "C" size_t clipk_sumsq_workspace(int64_t n) { return (size_t)ew_blocks(n / 4 + 1) * sizeof(float); }
extern "C" int clipk_sumsq(const float* g, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || !out || !workspace || n <= 0 || !aligned16(g)) return CLIPK_ERR_BAD_ARG;
  const int blocks = ew_blocks(n / 4 + 1);
  if (workspace_bytes < (size_t)blocks * sizeof(float)) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (long)n, (float*)workspace);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, blocks, out);
  return clipk_check_launch();
}
extern "C" int clipk_adamw_step(float* w, const float* g, float* m, float* v, void* w_bf16, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int step,
                                const float* grad_norm_sq, float max_norm, float grad_scale, const float* hyper_dev,
                                void* stream) {
  if (!w || !g || !m || !v || n <= 0 || (step < 1 && !hyper_dev)) return CLIPK_ERR_BAD_ARG;
  if (hyper_dev && step < 1) step = 1;
  // bias corrections in double on the host, as torch.optim.AdamW computes them (f32 powf is ~6e-5 off in 1 - beta2^t
  // at small t)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, w, g, m, v,
                     (unsigned short*)w_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                     grad_norm_sq, max_norm, grad_scale, hyper_dev);
  return clipk_check_launch();
}
