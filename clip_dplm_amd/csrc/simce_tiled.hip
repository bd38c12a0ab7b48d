// simce_tiled.hip — second-generation LSE pass of the fused similarity + cross-entropy (simce.hip has the contract).
//
// The first kernel gives one workgroup 32 queries and lets its waves split the contraction dimension P: every 32-key
// tile then costs a synchronous staging step, a cross-wave reduction through LDS and two barriers, and only one wave
// does the softmax — 102 us for one rank's config-3 block (512 x 4096 x 512), 16 TFLOP/s of exact-f32 matrix work.
// This one is the tiled exact-f32 GEMM of gemm_f32.hip with the softmax statistics as its epilogue:
//   * a workgroup owns 64 queries and walks 64-key tiles; per tile the 64 x 64 block of S^T = Y X^T is a BK = 16 K-loop
//     over P with register-staged double buffering (keys are the MFMA rows, queries the lanes), 4 waves = 2 (keys) x 2
//     (queries), one 32 x 32 accumulator each: nothing crosses waves inside a tile;
//   * each lane keeps the running (max, sum) of its query over the key rows it sees (16 per tile) in registers; the two
//     lane halves and the two key-waves are merged once, at the end (one lane^32 exchange, one LDS hop);
//   * key-range splits write (m, l) partials that simce_lse_finalize merges in a fixed order, as before.
// Numerics: the same k-ordered f32 fmaf chains per logit (the MFMA's arithmetic), a different but fixed summation order
// of the exponentials; deterministic.
#include "common.h"
#include <math.h>

namespace {

constexpr int TQ = 64, TK = 64;                   // queries per workgroup, keys per tile

struct LP {
  const float* X; int Mx;
  const float* Y; int Ny;
  const float* Yc; int Nc;
  int P;
  const float* scale; int label_offset;
  float* part_ml;      // [ksplit][Mx][2]
  float* pos;          // [Mx]
  int tiles_per_split, ntiles;
};

__device__ __forceinline__ int keyrow32(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x4 ld4(const float* row, int k, int P) {
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  if (k + 3 < P) t = *reinterpret_cast<const f32x4*>(row + k);
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (k + e < P) t[e] = row[k + e];
  }
  return t;
}

// S^T tile [64 keys][64 queries] += Y_tile X_tile^T over the whole contraction P: BKT-deep K-steps, operands global ->
// registers -> LDS (issue early / write late, two LDS buffers), wave (wm, wn) accumulates keys [32 wm, +32) x queries
// [32 wn, +32).  yrows / xrows: this thread's staging rows (thread -> row = idx / (BKT / 4), float4 idx % (BKT / 4)).
// Ends with a barrier: every wave has finished reading the buffers.
template <int BKT>
__device__ __forceinline__ void s_tile(f32x16& acc, const float* const (&yrows)[BKT / 16], const float* const (&xrows)[BKT / 16],
                                       float* smem, int P, int tid, int wm, int wn, int li, int h) {
  constexpr int LD = BKT + 4, TL = 64 * LD, NF = BKT / 16, NJJ = BKT / 8, QPR = BKT / 4;
  f32x4 ya[NF], xa[NF];
  int srow[NF], skq[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int idx = tid + i * 256;
    srow[i] = idx / QPR; skq[i] = (idx % QPR) * 4;
    ya[i] = ld4(yrows[i], skq[i], P); xa[i] = ld4(xrows[i], skq[i], P);
  }
  __syncthreads();                                                        // whoever read these buffers last is done
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    *reinterpret_cast<f32x4*>(smem + srow[i] * LD + skq[i]) = ya[i];
    *reinterpret_cast<f32x4*>(smem + TL + srow[i] * LD + skq[i]) = xa[i];
  }
  __syncthreads();
  const int nk = (P + BKT - 1) / BKT;
  for (int s = 0; s < nk; ++s) {
    const float* yt = smem + (s & 1) * 2 * TL;
    const float* xt = yt + TL;
    if (s + 1 < nk) {
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        ya[i] = ld4(yrows[i], (s + 1) * BKT + skq[i], P); xa[i] = ld4(xrows[i], (s + 1) * BKT + skq[i], P);
      }
    }
    f32x4 af[NJJ], bf[NJJ];
#pragma unroll
    for (int jj = 0; jj < NJJ; ++jj) {
      af[jj] = *reinterpret_cast<const f32x4*>(yt + (wm * 32 + li) * LD + 8 * jj + 4 * h);
      bf[jj] = *reinterpret_cast<const f32x4*>(xt + (wn * 32 + li) * LD + 8 * jj + 4 * h);
    }
#pragma unroll
    for (int jj = 0; jj < NJJ; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[jj][e], bf[jj][e], acc, 0, 0, 0);
    if (s + 1 < nk) {
      float* nb = smem + ((s + 1) & 1) * 2 * TL;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        *reinterpret_cast<f32x4*>(nb + srow[i] * LD + skq[i]) = ya[i];
        *reinterpret_cast<f32x4*>(nb + TL + srow[i] * LD + skq[i]) = xa[i];
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256, 2) void simce_lse_tiled_kernel(const LP p) {
  constexpr int BKL = 32;                                                 // 16 MFMAs per wave between barriers
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * 64 * (BKL + 4)];   // 2 buffers x (keys | queries)
  __shared__ float mrg[2][2][TQ];                                         // [key-wave][m | l][query]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;                                  // key half, query half
  const int li = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * TQ, ks = blockIdx.y;
  const int P = p.P, Nkeys = p.Ny + p.Nc;
  const float scale = p.scale[0];
  const int qg = q0 + wn * 32 + li;                                       // this lane's query
  const int label = p.label_offset + qg;
  float m_run = -INFINITY, l_run = 0.f, pos_v = 0.f;
  bool pos_hit = false;
  const float* xrows[BKL / 16];
#pragma unroll
  for (int i = 0; i < BKL / 16; ++i) {
    int q = q0 + (tid + i * 256) / (BKL / 4); q = q < p.Mx ? q : p.Mx - 1;
    xrows[i] = p.X + (long)q * P;
  }
  const int t_beg = ks * p.tiles_per_split;
  int t_end = t_beg + p.tiles_per_split; t_end = t_end < p.ntiles ? t_end : p.ntiles;

  for (int kt = t_beg; kt < t_end; ++kt) {
    const int j0 = kt * TK;
    const float* yrows[BKL / 16];
#pragma unroll
    for (int i = 0; i < BKL / 16; ++i) {
      int j = j0 + (tid + i * 256) / (BKL / 4); j = j < Nkeys ? j : Nkeys - 1;      // clamped: masked in the epilogue
      yrows[i] = (j < p.Ny) ? p.Y + (long)j * P : p.Yc + (long)(j - p.Ny) * P;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    s_tile<BKL>(acc, yrows, xrows, smem, P, tid, wm, wn, li, h);
    // ---- softmax statistics of this lane's query over its 16 key rows of the tile
    float sv[16], tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = j0 + wm * 32 + keyrow32(r, h);
      sv[r] = key < Nkeys ? scale * acc[r] : -INFINITY;
      tmax = fmaxf(tmax, sv[r]);
      if (key == label && key < p.Ny) { pos_v = sv[r]; pos_hit = true; }
    }
    if (tmax > -INFINITY) {
      const float m_new = fmaxf(m_run, tmax);
      float a = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) a += expf(sv[r] - m_new);             // exp(-inf) = 0 for masked keys
      l_run = l_run * expf(m_run - m_new) + a;
      m_run = m_new;
    }
  }

  // ---- merge: lane halves, then the two key-waves
  {
    const float m_o = __shfl_xor(m_run, 32, 64), l_o = __shfl_xor(l_run, 32, 64);
    const float m_n = fmaxf(m_run, m_o);
    float l_n = 0.f;
    if (m_n > -INFINITY) l_n = l_run * expf(m_run - m_n) + l_o * expf(m_o - m_n);
    m_run = m_n; l_run = l_n;
  }
  if (pos_hit && qg < p.Mx) p.pos[qg] = pos_v;                            // exactly one lane of the grid holds it
  if (h == 0) { mrg[wm][0][wn * 32 + li] = m_run; mrg[wm][1][wn * 32 + li] = l_run; }
  __syncthreads();
  if (wm == 0 && h == 0 && qg < p.Mx) {
    const float m1 = mrg[1][0][wn * 32 + li], l1 = mrg[1][1][wn * 32 + li];
    const float m_n = fmaxf(m_run, m1);
    float l_n = 0.f;
    if (m_n > -INFINITY) l_n = l_run * expf(m_run - m_n) + l1 * expf(m1 - m_n);
    float* o = p.part_ml + ((long)ks * p.Mx + qg) * 2;
    o[0] = m_n; o[1] = l_n;
  }
}

// ------------------------------------------------------------------------------------------------ gradient pass
// dX[q, :] = sum_key G[q, key] Y[key, :] with G = dL/dS formed from the two LSE vectors (simce.hip has the formula).
// Same 64 x 64 tiles: (1) S^T tile by the K-loop over P; (2) G in the accumulator layout (keys on rows, queries on
// lanes), written once to a 16 KiB LDS tile; (3) dX^T[p, q] += Y^T[p, key] G^T[key, q] as a second MFMA product whose
// M dimension is p: wave w owns p in [w P/4, (w+1) P/4) for all 64 queries (up to 8 accumulators), the key tile comes
// back from L2 in four 16-key blocks [16][P] staged in LDS (A operand: Y[key][p], one ds_read_b32 per MFMA, conflict
// free), G^T is the B operand straight from the LDS tile.  70 KiB of LDS and < 256 VGPRs: TWO workgroups per CU, so
// one's staging latencies sit under the other's MFMAs (with 64 KiB key halves and one workgroup per CU the same kernel
// took 79 us instead of the figure in DESIGN.md).  P <= 512 (128 accumulator registers); larger P keeps the
// first-generation kernel.  Key-range splits write dX slabs that simce_grad_finalize sums in a fixed order.
struct GP2 {
  const float* X; int Mx;
  const float* Y; int Ny;
  const float* Yc; int Nc;
  int P;
  const float* scale; int label_offset;
  const float* lse_x; const float* lse_y;
  float w_row, w_col, inv_bg;
  const float* upstream;   // device scalar multiplied into inv_bg, or null
  float* slab;         // [ksplit][Mx][P]
  float* dsc_part;     // [ksplit][Mx]
  int tiles_per_split, ntiles;
};

constexpr int GPMAX = 512;                         // contraction / output width limit of the tiled gradient pass
constexpr int YH_LD = GPMAX + 4;                   // floats per staged key row
constexpr int KSB = 16;                            // keys per staged block of the second product
constexpr int BKG = 16;                            // K-step of the gradient pass's S tile (LDS budget: 2 workgroups per CU)
constexpr int GRAD_LDS_FLOATS = 2 * 2 * 64 * (BKG + 4) + TK * TQ + KSB * YH_LD + 2 * TQ;

__global__ __launch_bounds__(256, 2) void simce_grad_tiled_kernel(const GP2 p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);                      // K-loop buffers
  float* gl = smem + 2 * 2 * 64 * (BKG + 4);                              // G tile [64 keys][64 queries]
  float* yh = gl + TK * TQ;                                               // key block [16][YH_LD] / output transposes
  float* dsl = yh + KSB * YH_LD;                                          // [64 queries] dscale partials of key-wave 1
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int li = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * TQ, ks = blockIdx.y;
  const int P = p.P, Nkeys = p.Ny + p.Nc;
  const float scale = p.scale[0];
  const int qg = q0 + wn * 32 + li;
  const int label = p.label_offset + qg;
  const float lse_xi = p.lse_x[qg < p.Mx ? qg : p.Mx - 1];
  const float ibg = p.upstream ? p.inv_bg * p.upstream[0] : p.inv_bg;
  const int npt = (P + 127) / 128;                                        // 32-row p tiles per wave: P/4 / 32
  const int pw = npt * 32;                                                // p rows per wave
  f32x16 dx[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) dx[a][b][r] = 0.f;
  float dsc = 0.f;

  const float* xrows[1];
  { int q = q0 + (tid >> 2); q = q < p.Mx ? q : p.Mx - 1; xrows[0] = p.X + (long)q * P; }
  auto key_row = [&](int j) {
    j = j < Nkeys ? j : Nkeys - 1;
    return (j < p.Ny) ? p.Y + (long)j * P : p.Yc + (long)(j - p.Ny) * P;
  };
  const int t_beg = ks * p.tiles_per_split;
  int t_end = t_beg + p.tiles_per_split; t_end = t_end < p.ntiles ? t_end : p.ntiles;

  for (int kt = t_beg; kt < t_end; ++kt) {
    const int j0 = kt * TK;
    const float* yrows[1] = {key_row(j0 + (tid >> 2))};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    s_tile<BKG>(acc, yrows, xrows, smem, P, tid, wm, wn, li, h);          // (its first barrier also frees gl / yh)
    // ---- G (accumulator layout: rows = keys, lanes = queries) -> LDS tile gl[key][query]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = wm * 32 + keyrow32(r, h);
      const int key = j0 + kl;
      const float sv = scale * acc[r];
      float gv = 0.f;
      if (key < Nkeys && qg < p.Mx) {
        gv = p.w_row * expf(sv - lse_xi);
        if (key < p.Ny) {
          gv += p.w_col * expf(sv - p.lse_y[key]);
          if (key == label) gv -= (p.w_row + p.w_col);
        }
        gv *= ibg;
      }
      dsc += gv * acc[r];
      gl[kl * TQ + wn * 32 + li] = gv;
    }
    // ---- dX^T += Y^T G^T, the key tile in blocks of KSB keys
    for (int kb = 0; kb < TK / KSB; ++kb) {
      __syncthreads();                                                    // gl complete (kb = 0) / yh free again
      {
        // stage Y[16 keys][P]: thread -> (key = tid / 16, 16-B chunks c = tid % 16 + 16 i), loads first, then stores
        const float* yr = key_row(j0 + kb * KSB + (tid >> 4));
        float* dst = yh + (tid >> 4) * YH_LD;
#pragma unroll
        for (int g = 0; g < 2; ++g) {                                     // two groups of four: 16 staging registers
          f32x4 tmp[GPMAX / 128];
#pragma unroll
          for (int i = 0; i < GPMAX / 128; ++i) {
            const int c = (tid & 15) + 16 * (g * (GPMAX / 128) + i);
            tmp[i] = (c * 4 < P) ? ld4(yr, c * 4, P) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int i = 0; i < GPMAX / 128; ++i) {
            const int c = (tid & 15) + 16 * (g * (GPMAX / 128) + i);
            if (c * 4 < P) *reinterpret_cast<f32x4*>(dst + c * 4) = tmp[i];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < KSB / 2; ++u) {                                 // MFMA u contracts keys 2u (h = 0) and 2u + 1
        const int kl = 2 * u + h;
        const float b0 = gl[(kb * KSB + kl) * TQ + li], b1 = gl[(kb * KSB + kl) * TQ + 32 + li];
#pragma unroll
        for (int a = 0; a < 4; ++a)
          if (a < npt) {
            const int prow = wid * pw + a * 32 + li;
            const float av = prow < P ? yh[kl * YH_LD + prow] : 0.f;
            dx[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, dx[a][0], 0, 0, 0);
            dx[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, dx[a][1], 0, 0, 0);
          }
      }
    }
  }

  // ---- dX^T accumulators -> [q][p] rows through LDS (one 32 x 32 block per wave at a time), whole 128-B row segments
  __syncthreads();
  float* tb = yh + wid * (32 * 33);
#pragma unroll
  for (int a = 0; a < 4; ++a) {                             // (fully unrolled: the accumulators are register arrays)
    if (a < npt) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tb[li * 33 + keyrow32(r, h)] = dx[a][b][r];  // [query][p]
        // wave-private region: the wave's own writes are visible to its reads in program order
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int ql = it * 2 + h;                                      // 2 query rows per pass, 32 consecutive p each
          const int q = q0 + b * 32 + ql, pp = wid * pw + a * 32 + li;
          if (q < p.Mx && pp < P) p.slab[((long)ks * p.Mx + q) * P + pp] = tb[ql * 33 + li];
        }
      }
    }
  }
  // ---- d scale partials: lane halves, then the two key-waves
  dsc += __shfl_xor(dsc, 32, 64);
  if (wm == 1 && h == 0) dsl[wn * 32 + li] = dsc;
  __syncthreads();
  if (wm == 0 && h == 0 && qg < p.Mx) p.dsc_part[(long)ks * p.Mx + qg] = dsc + dsl[wn * 32 + li];
}

}  // namespace

// plan: one workgroup per (64-query block, key split); splits so that the grid holds >= 2 workgroups per CU
extern "C" void clipk_simce_tiled_plan(int Mx, int Nkeys, int* nqb, int* ksplit, int* tps, int* ntiles) {
  *nqb = (Mx + TQ - 1) / TQ;
  *ntiles = (Nkeys + TK - 1) / TK;
  int ks = (512 + *nqb - 1) / *nqb;
  if (ks > *ntiles) ks = *ntiles;
  if (ks < 1) ks = 1;
  *tps = (*ntiles + ks - 1) / ks;
  *ksplit = (*ntiles + *tps - 1) / *tps;
}

extern "C" void clipk_simce_grad_tiled_plan(int Mx, int Nkeys, int* nqb, int* ksplit, int* tps, int* ntiles) {
  *nqb = (Mx + TQ - 1) / TQ;
  *ntiles = (Nkeys + TK - 1) / TK;
  int ks = (512 + *nqb - 1) / *nqb;                           // two workgroups per CU (70 KiB of LDS each)
  if (ks > *ntiles) ks = *ntiles;
  if (ks < 1) ks = 1;
  *tps = (*ntiles + ks - 1) / ks;
  *ksplit = (*ntiles + *tps - 1) / *tps;
}

extern "C" int clipk_simce_grad_tiled_launch(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                                             int P, const float* scale, int label_offset, const float* lse_x,
                                             const float* lse_y, float w_row, float w_col, float inv_bg,
                                             const float* upstream, float* slab, float* dsc_part, void* stream) {
  if (P > GPMAX) return CLIPK_ERR_UNSUPPORTED;
  GP2 p;
  p.X = X; p.Mx = Mx; p.Y = Y; p.Ny = Ny; p.Yc = Yc ? Yc : Y; p.Nc = Nc; p.P = P;
  p.scale = scale; p.label_offset = label_offset; p.lse_x = lse_x; p.lse_y = lse_y;
  p.w_row = w_row; p.w_col = w_col; p.inv_bg = inv_bg; p.upstream = upstream; p.slab = slab; p.dsc_part = dsc_part;
  int nqb, ksplit;
  clipk_simce_grad_tiled_plan(Mx, Ny + Nc, &nqb, &ksplit, &p.tiles_per_split, &p.ntiles);
  const size_t lds = (size_t)GRAD_LDS_FLOATS * sizeof(float);
  static std::atomic<uint64_t> attr_set{0};
  clipk_once_per_device(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(simce_grad_tiled_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  hipLaunchKernelGGL(simce_grad_tiled_kernel, dim3(nqb, ksplit), dim3(256), lds, (hipStream_t)stream, p);
  return clipk_check_launch();
}

extern "C" int clipk_simce_lse_tiled_launch(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc, int P,
                                            const float* scale, int label_offset, float* part_ml, float* pos,
                                            void* stream) {
  LP p;
  p.X = X; p.Mx = Mx; p.Y = Y; p.Ny = Ny; p.Yc = Yc ? Yc : Y; p.Nc = Nc; p.P = P;
  p.scale = scale; p.label_offset = label_offset; p.part_ml = part_ml; p.pos = pos;
  int nqb, ksplit;
  clipk_simce_tiled_plan(Mx, Ny + Nc, &nqb, &ksplit, &p.tiles_per_split, &p.ntiles);
  hipLaunchKernelGGL(simce_lse_tiled_kernel, dim3(nqb, ksplit), dim3(256), 0, (hipStream_t)stream, p);
  return clipk_check_launch();
}
