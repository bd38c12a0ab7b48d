// simce.hip — fused similarity + cross-entropy for the CLIP loss; the B x B logits never reach HBM.
//
// Reference arithmetic replaced (see include/clipk.h): old/clip.py:66-67 (scaled A·B^T),
// old/ablation.py:16 / rna_clip_codes.ipynb:1952-1953 (row / column cross-entropy on the diagonal),
// old/clip_opt.py:115-121,130-151 (extra cache columns in the row direction).
//
// gfx950 design:
//   * exact-f32 matrix cores: v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, bit-for-bit f32), so
//     the loss meets the 1e-3 parity bar without a bf16 rounding of the unit-norm embeddings;
//   * "swapped" product S^T = Y·X^T: keys on the accumulator rows (registers), queries on the lanes,
//     so the per-query softmax statistics are in-register reductions plus one lane^32 exchange;
//   * one workgroup = 32 queries x a contiguous range of 32-key tiles; its NW = P/128 waves split the
//     contraction dimension P, each wave keeps its slice of the query block in registers as MFMA B
//     fragments and its slice of the key tile in a wave-private LDS region (coalesced 512-B row
//     segments from HBM/L2), partial S tiles are summed through LDS;
//   * backward: G = dL/dS is formed in the accumulator layout and is directly the B operand of the
//     second product dX^T += Y^T·G^T (same key tile, still in LDS) — no LDS round trip for G;
//   * key-range splits write f32 partials (online-softmax pairs / dX slabs) that a tiny finalize
//     kernel merges in a fixed order: deterministic, no float atomics.
#include "common.h"
#include <math.h>

namespace {

constexpr int QB = 32;          // queries per workgroup
constexpr int KT = 32;          // keys per tile
constexpr int PWMAX = 128;      // contraction slice per wave
constexpr int MAXW = 8;         // P <= 1024
constexpr int MAXZ = 6;         // problems per batched launch

enum { MODE_LSE = 0, MODE_GRAD = 1, MODE_LOGITS = 2 };

struct SP {
  const float* X; int Mx;
  const float* Y; int Ny;
  const float* Yc; int Nc;
  int P, Pw, NW;
  const float* scale;
  int label_offset;
  // LSE outputs
  float* part_ml;      // [ksplit][Mx][2]
  float* pos;          // [Mx]
  // grad
  const float* lse_x; const float* lse_y;
  float w_row, w_col, inv_bg;
  const float* upstream;   // device scalar multiplied into inv_bg (the loss' incoming gradient), or null
  float* slab;         // [ksplit][Mx][P]
  float* dsc_part;     // [ksplit][Mx]
  // logits / f32 Linear epilogue
  float* S; long lds_out;
  const float* ep_bias; const float* ep_add; const float* ep_add_scale;
  int ksplit, tiles_per_split, ntiles;
  // batched launch (blockIdx.z = problem): several same-shape (X, Y) problems in one grid — the three pairwise blocks
  // of the tri-modal loss in both directions.  nz == 0: the single problem described above.
  int nz;
  const float* Xz[MAXZ]; const float* Yz[MAXZ]; const float* lse_xz[MAXZ]; const float* lse_yz[MAXZ];
  long z_part, z_pos, z_slab, z_dsc;     // per-problem strides (floats) of part_ml / pos / slab / dsc_part
};

__device__ __forceinline__ int keyrow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int MODE>
__global__ __launch_bounds__(512) void simce_kernel(const SP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int NW = p.NW, Pw = p.Pw, P = p.P;
  const int YLD = Pw + 4;
  const int q = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * QB, ks = blockIdx.y;
  const int Nkeys = p.Ny + p.Nc;
  const int z = blockIdx.z;
  const float* Xb = p.nz ? p.Xz[z] : p.X;
  const float* Yb = p.nz ? p.Yz[z] : p.Y;
  const float* Ycb = p.nz ? Yb : p.Yc;
  const float* lse_xb = p.nz ? p.lse_xz[z] : p.lse_x;
  const float* lse_yb = p.nz ? p.lse_yz[z] : p.lse_y;
  float* part_ml = p.part_ml + (p.nz ? z * p.z_part : 0);
  float* posb = p.pos + (p.nz ? z * p.z_pos : 0);
  float* slabb = p.slab + (p.nz ? z * p.z_slab : 0);
  float* dsc_partb = p.dsc_part + (p.nz ? z * p.z_dsc : 0);
  float* ylds = reinterpret_cast<float*>(smem) + w * KT * YLD;
  float* red = reinterpret_cast<float*>(smem) + NW * KT * YLD;     // [NW][16*64]
  const int pbeg = w * Pw;
  const float scale = p.scale ? p.scale[0] : 1.0f;

  // ---- this wave's slice of the query block, as B fragments of S^T = Y·X^T (kept in registers)
  f32x4 xf[PWMAX / 8];
  {
    int qi = q0 + q; qi = qi < p.Mx ? qi : p.Mx - 1;
    const float* xr = Xb + (long)qi * P;
#pragma unroll
    for (int t = 0; t < PWMAX / 8; ++t) {
      xf[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t * 8 < Pw) {
        const int pp = pbeg + t * 8 + 4 * h;
        if (pp < P) xf[t] = *reinterpret_cast<const f32x4*>(xr + pp);
      }
    }
  }

  const int qg = q0 + q;                        // global query row within X
  const int label = p.label_offset + qg;        // its positive key
  float m_run = -INFINITY, l_run = 0.f, pos_v = 0.f;
  bool pos_hit = false;
  const float ibg = (MODE == MODE_GRAD && p.upstream) ? p.inv_bg * p.upstream[0] : p.inv_bg;
  float lse_xi = 0.f, dsc = 0.f;
  f32x16 dx[PWMAX / 32];
  if (MODE == MODE_GRAD) {
    lse_xi = lse_xb[qg < p.Mx ? qg : p.Mx - 1];
#pragma unroll
    for (int t = 0; t < PWMAX / 32; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dx[t][r] = 0.f;
  }

  const int t_beg = ks * p.tiles_per_split;
  int t_end = t_beg + p.tiles_per_split; t_end = t_end < p.ntiles ? t_end : p.ntiles;
  const int c4_per_row = Pw >> 2;               // float4 per key row of this wave's slice
  const int n_it = (KT * c4_per_row) >> 6;      // = Pw/8 passes of 64 lanes

  for (int kt = t_beg; kt < t_end; ++kt) {
    const int j0 = kt * KT;
    __syncthreads();                            // previous tile fully consumed (red[] and ylds reuse)
    // ---- stage this wave's [32 keys][Pw] slice of the key tile: whole row segments, 16 B per lane
    for (int it = 0; it < n_it; ++it) {
      const int idx = lane + 64 * it;
      const int row = idx / c4_per_row, c4 = idx - row * c4_per_row;
      int j = j0 + row; j = j < Nkeys ? j : Nkeys - 1;
      const float* yr = (j < p.Ny) ? Yb + (long)j * P : Ycb + (long)(j - p.Ny) * P;
      const int pp = pbeg + c4 * 4;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pp < P) v = *reinterpret_cast<const f32x4*>(yr + pp);
      *reinterpret_cast<f32x4*>(ylds + row * YLD + c4 * 4) = v;
    }
    // wave-private region: a wave's own LDS writes are visible to its later reads in program order
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int t = 0; t < PWMAX / 8; ++t) {
      if (t * 8 < Pw) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ylds + q * YLD + t * 8 + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], xf[t][e], s, 0, 0, 0);
      }
    }
    if (NW > 1) {                               // sum the per-wave partial tiles (fixed order)
      float* mine = red + w * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = s[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = 0.f;
        for (int ww = 0; ww < NW; ++ww) a += red[ww * 1024 + r * 64 + lane];
        s[r] = a;
      }
    }

    if (MODE == MODE_LSE) {
      if (w == 0) {
        float tmax = -INFINITY;
        float sv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j0 + keyrow(r, h);
          sv[r] = key < Nkeys ? scale * s[r] : -INFINITY;
          tmax = fmaxf(tmax, sv[r]);
          if (key == label && key < p.Ny) { pos_v = sv[r]; pos_hit = true; }
        }
        if (tmax > -INFINITY) {
          const float m_new = fmaxf(m_run, tmax);
          float acc = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc += expf(sv[r] - m_new);   // exp(-inf) = 0 for masked keys
          l_run = l_run * expf(m_run - m_new) + acc;
          m_run = m_new;
        }
      }
    } else if (MODE == MODE_GRAD) {
      f32x16 g;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j0 + keyrow(r, h);
        const float sv = scale * s[r];
        float gv = 0.f;
        if (key < Nkeys) {
          gv = p.w_row * expf(sv - lse_xi);
          if (key < p.Ny) {
            gv += p.w_col * expf(sv - lse_yb[key]);
            if (key == label) gv -= (p.w_row + p.w_col);
          }
          gv *= ibg;
        }
        g[r] = gv;
        dsc += gv * s[r];
      }
      // dX^T[p][q] += sum_key Y[key][p] * G^T[key][q]; G registers are the B operand as they stand
#pragma unroll
      for (int tt = 0; tt < PWMAX / 32; ++tt) {
        if (tt * 32 < Pw) {
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            const float a = ylds[keyrow(u, h) * YLD + tt * 32 + q];
            dx[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, g[u], dx[tt], 0, 0, 0);
          }
        }
      }
    } else {  // MODE_LOGITS: roles are swapped by the host (queries = Y rows of the caller)
      if (w == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = j0 + keyrow(r, h);
          if (key < Nkeys && qg < p.Mx) {
            float v = scale * s[r];
            if (p.ep_bias) v += p.ep_bias[qg];
            if (p.ep_add) v += (p.ep_add_scale ? p.ep_add_scale[0] : 1.0f) * p.ep_add[(long)key * p.lds_out + qg];
            p.S[(long)key * p.lds_out + qg] = v;
          }
        }
      }
    }
  }

  if (MODE == MODE_LSE) {
    if (w == 0) {
      // merge the two lane halves (keys 4h.. interleaved) of each query
      const float m_o = __shfl_xor(m_run, 32, 64), l_o = __shfl_xor(l_run, 32, 64);
      const float m_n = fmaxf(m_run, m_o);
      float l_n = 0.f;
      if (m_n > -INFINITY) l_n = l_run * expf(m_run - m_n) + l_o * expf(m_o - m_n);
      if (qg < p.Mx) {
        if (h == 0) {
          float* o = part_ml + ((long)ks * p.Mx + qg) * 2;
          o[0] = m_n; o[1] = l_n;
        }
        if (pos_hit) posb[qg] = pos_v;
      }
    }
  } else if (MODE == MODE_GRAD) {
    // stage dX^T tiles as [q][p] rows in this wave's LDS region, then write whole row segments
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < PWMAX / 32; ++tt)
      if (tt * 32 < Pw)
#pragma unroll
        for (int r = 0; r < 16; ++r) ylds[q * YLD + tt * 32 + keyrow(r, h)] = dx[tt][r];
    for (int it = 0; it < n_it; ++it) {
      const int idx = lane + 64 * it;
      const int row = idx / c4_per_row, c4 = idx - row * c4_per_row;
      const int pp = pbeg + c4 * 4;
      if (q0 + row < p.Mx && pp < P) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ylds + row * YLD + c4 * 4);
        *reinterpret_cast<f32x4*>(slabb + ((long)ks * p.Mx + q0 + row) * P + pp) = v;
      }
    }
    if (w == 0) {
      const float d = dsc + __shfl_xor(dsc, 32, 64);
      if (h == 0 && qg < p.Mx) dsc_partb[(long)ks * p.Mx + qg] = d;
    }
  }
}

// one wave per query: lanes take the key-split partials (m, l) s = lane, lane + 64, ..., merged by wave reductions (fixed
// order: deterministic).  The serial one-thread-per-query loop was a 64-deep dependent chain of loads + expf (23 us at
// 64 splits).
__global__ __launch_bounds__(256) void simce_lse_finalize(const float* part_ml, int ksplit, int Mx, float* lse, long z_part) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= Mx) return;
  part_ml += blockIdx.y * z_part;                 // batched launch: blockIdx.y = problem
  lse += (long)blockIdx.y * Mx;
  float m = -INFINITY;
  for (int s = lane; s < ksplit; s += 64) m = fmaxf(m, part_ml[((long)s * Mx + i) * 2]);
  m = wave_max(m);
  float l = 0.f;
  for (int s = lane; s < ksplit; s += 64) {
    const float ms = part_ml[((long)s * Mx + i) * 2], ls = part_ml[((long)s * Mx + i) * 2 + 1];
    if (ms > -INFINITY) l += ls * expf(ms - m);
  }
  l = wave_sum(l);
  if (lane == 0) lse[i] = m + logf(l);
}

__global__ void simce_grad_finalize(const float* slab, const float* dsc_part, int ksplit, int Mx, int P,
                                    const float* scale, float* dX, float* dscale_partial, long z_slab,
                                    long z_dsc) {
  const long n4 = (long)Mx * P / 4;
  const float sc = scale[0];
  slab += blockIdx.y * z_slab;                    // batched launch: blockIdx.y = problem
  dsc_part += blockIdx.y * z_dsc;
  dX += (long)blockIdx.y * Mx * P;
  if (dscale_partial) dscale_partial += (long)blockIdx.y * Mx;
  // 64 float4 columns x 4 slab groups per workgroup: group g sums slabs [g ksplit/4, (g+1) ksplit/4) in order (eight
  // loads in flight), the four group sums are added in group order: a fixed summation tree, so the result does not
  // depend on scheduling (and equals the plain slab-ordered sum's tree for ksplit <= 4).
  __shared__ f32x4 red[3][64];
  const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int per = (ksplit + 3) / 4;
  const int s0 = g * per, s1 = (s0 + per < ksplit) ? s0 + per : ksplit;
  const long slab_n = (long)Mx * P;
  for (long i0 = blockIdx.x * 64L; i0 < n4; i0 += gridDim.x * 64L) {     // (uniform trip count per workgroup)
    const long i = i0 + col;
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < n4) {
      int s = s0;
      for (; s + 8 <= s1; s += 8) {
        f32x4 t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = reinterpret_cast<const f32x4*>(slab + (long)(s + e) * slab_n)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) a += t[e];
      }
      for (; s < s1; ++s) a += reinterpret_cast<const f32x4*>(slab + (long)s * slab_n)[i];
    }
    if (g) red[g - 1][col] = a;
    __syncthreads();
    if (g == 0 && i < n4) {
      a += red[0][col]; a += red[1][col]; a += red[2][col];
      reinterpret_cast<f32x4*>(dX)[i] = a * sc;
    }
    __syncthreads();
  }
  if (dscale_partial) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < Mx; i += (long)gridDim.x * blockDim.x) {
      float a = 0.f;
      for (int s = 0; s < ksplit; ++s) a += dsc_part[(long)s * Mx + i];
      dscale_partial[i] = a;
    }
  }
}

struct Plan { int NW, Pw, nqb, ntiles, ksplit, tps; size_t lds; };

bool make_plan(int Mx, int Nkeys, int P, Plan* pl) {
  if (Mx <= 0 || Nkeys <= 0 || P <= 0 || (P & 3) || P > PWMAX * MAXW) return false;
  int nw = 1;
  while (nw < MAXW && (P + nw - 1) / nw > PWMAX) nw <<= 1;
  int pw = (P + nw - 1) / nw; pw = (pw + 7) & ~7;
  pl->NW = nw; pl->Pw = pw;
  pl->nqb = (Mx + QB - 1) / QB;
  pl->ntiles = (Nkeys + KT - 1) / KT;
  int ks = (512 + pl->nqb - 1) / pl->nqb;
  if (ks > pl->ntiles) ks = pl->ntiles;
  if (ks < 1) ks = 1;
  pl->tps = (pl->ntiles + ks - 1) / ks;
  pl->ksplit = (pl->ntiles + pl->tps - 1) / pl->tps;
  pl->lds = (size_t)nw * KT * (pw + 4) * 4 + (size_t)nw * 1024 * 4;
  if (pl->lds > 160 * 1024) return false;
  return true;
}

template <int MODE>
int launch(const SP& sp, const Plan& pl, hipStream_t st) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(simce_kernel<MODE>),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  hipLaunchKernelGGL(simce_kernel<MODE>, dim3(pl.nqb, pl.ksplit, sp.nz > 0 ? sp.nz : 1), dim3(pl.NW * 64), pl.lds, st, sp);
  return clipk_check_launch();
}

}  // namespace

// simce_tiled.hip: second-generation LSE pass (64 x 64 tiles, gemm_f32-style K-loop)
extern "C" void clipk_simce_tiled_plan(int Mx, int Nkeys, int* nqb, int* ksplit, int* tps, int* ntiles);
extern "C" int clipk_simce_lse_tiled_launch(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc, int P,
                                            const float* scale, int label_offset, float* part_ml, float* pos,
                                            void* stream);
extern "C" void clipk_simce_grad_tiled_plan(int Mx, int Nkeys, int* nqb, int* ksplit, int* tps, int* ntiles);
static bool use_tiled_grad(int Mx, int Nkeys, int P, int Pw) {
  const int mode = clipk_opt_get(OPT_SIMCE_KERNEL);
  if (P > 512) return false;
  // the first-generation gradient pass walks its P slice in blocks of 32 columns: it needs Pw % 32 == 0
  if (Pw % 32) return true;
  if (mode == 1) return false;
  if (mode == 2) return true;
  return Mx >= 64 && Nkeys >= 64;
}
static bool use_tiled_lse(int Mx, int Nkeys) {
  const int mode = clipk_opt_get(OPT_SIMCE_KERNEL);
  if (mode == 1) return false;
  if (mode == 2) return true;
  return Mx >= 64 && Nkeys >= 64;
}

extern "C" size_t clipk_simce_workspace(int Mx, int Nkeys, int P) {
  Plan pl;
  if (!make_plan(Mx, Nkeys, P, &pl)) return 0;
  int nqb, ks2, tps, nt;
  clipk_simce_tiled_plan(Mx, Nkeys, &nqb, &ks2, &tps, &nt);
  const size_t a = (size_t)pl.ksplit * Mx * ((size_t)P + 2) * sizeof(float);
  const size_t b = (size_t)ks2 * Mx * 2 * sizeof(float);          // (m, l) partials of the tiled LSE pass
  int ks3;
  clipk_simce_grad_tiled_plan(Mx, Nkeys, &nqb, &ks3, &tps, &nt);
  const size_t c = (size_t)ks3 * Mx * ((size_t)P + 1) * sizeof(float);   // dX slabs + dscale partials of the tiled grad pass
  const size_t ab = a > b ? a : b;
  return ab > c ? ab : c;
}

extern "C" int clipk_simce_lse(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                               int P, const float* scale, int label_offset, float* lse, float* pos,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (!X || !Y || !scale || !lse || !pos || !workspace || Nc < 0 || (Nc > 0 && !Yc)) return CLIPK_ERR_BAD_ARG;
  Plan pl;
  if (!make_plan(Mx, Ny + Nc, P, &pl)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(X) || !aligned16(Y) || (Yc && !aligned16(Yc))) return CLIPK_ERR_BAD_ARG;
  if (use_tiled_lse(Mx, Ny + Nc)) {
    int nqb, ks2, tps, nt;
    clipk_simce_tiled_plan(Mx, Ny + Nc, &nqb, &ks2, &tps, &nt);
    if (workspace_bytes < (size_t)ks2 * Mx * 2 * sizeof(float)) return CLIPK_ERR_BAD_ARG;
    int rc = clipk_simce_lse_tiled_launch(X, Mx, Y, Ny, Yc, Nc, P, scale, label_offset, (float*)workspace, pos, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(simce_lse_finalize, dim3((Mx + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, ks2, Mx, lse, 0L);
    return clipk_check_launch();
  }
  if (workspace_bytes < (size_t)pl.ksplit * Mx * 2 * sizeof(float)) return CLIPK_ERR_BAD_ARG;
  SP sp{};
  sp.X = X; sp.Mx = Mx; sp.Y = Y; sp.Ny = Ny; sp.Yc = Yc ? Yc : Y; sp.Nc = Nc;
  sp.P = P; sp.Pw = pl.Pw; sp.NW = pl.NW; sp.scale = scale; sp.label_offset = label_offset;
  sp.part_ml = (float*)workspace; sp.pos = pos;
  sp.ksplit = pl.ksplit; sp.tiles_per_split = pl.tps; sp.ntiles = pl.ntiles;
  int rc = launch<MODE_LSE>(sp, pl, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(simce_lse_finalize, dim3((Mx + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, pl.ksplit, Mx, lse, 0L);
  return clipk_check_launch();
}

// loss = (w_row * sum(lse_r - pos_r) + w_col * sum(lse_c - pos_c)) / bg in ONE launch (fixed summation order: strided
// partial sums per thread, then a tree over the 256 threads): the reference's two F.cross_entropy means and their
// average (rna_clip_codes.ipynb:1952-1953, old/ablation.py:16) were eight elementwise / reduce launches of 4.5 us each on
// [B] vectors - a fifth of config 1's captured step.
namespace {
__global__ __launch_bounds__(256) void ce_combine_kernel(const float* lse_r, const float* pos_r, const float* lse_c,
                                                         const float* pos_c, int n, float w_row, float w_col, float bg,
                                                         float* loss) {
  __shared__ float sr[256], sc[256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    a += lse_r[i] - pos_r[i];
    if (lse_c) b += lse_c[i] - pos_c[i];
  }
  sr[threadIdx.x] = a; sc[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sr[threadIdx.x] += sr[threadIdx.x + s]; sc[threadIdx.x] += sc[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float local = w_row * sr[0];
    if (lse_c) local = local + w_col * sc[0];
    loss[0] = local / bg;
  }
}
}  // namespace
extern "C" int clipk_ce_combine(const float* lse_r, const float* pos_r, const float* lse_c, const float* pos_c, int n,
                                float w_row, float w_col, float bg, float* loss, void* stream) {
  if (!lse_r || !pos_r || !loss || n <= 0 || (lse_c && !pos_c) || !(bg > 0.f)) return CLIPK_ERR_BAD_ARG;
  hipLaunchKernelGGL(ce_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lse_r, pos_r, lse_c, pos_c, n, w_row,
                     w_col, bg, loss);
  return clipk_check_launch();
}

extern "C" int clipk_simce_grad_tiled_launch(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                                             int P, const float* scale, int label_offset, const float* lse_x,
                                             const float* lse_y, float w_row, float w_col, float inv_bg,
                                             const float* upstream, float* slab, float* dsc_part, void* stream);

extern "C" int clipk_simce_grad_scaled(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                                       int P, const float* scale, int label_offset,
                                       const float* lse_x, const float* lse_y, float w_row, float w_col, float inv_bg,
                                       const float* upstream, float* dX, float* dscale_partial,
                                       void* workspace, size_t workspace_bytes, void* stream);
extern "C" int clipk_simce_grad(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                                int P, const float* scale, int label_offset,
                                const float* lse_x, const float* lse_y, float w_row, float w_col, float inv_bg,
                                float* dX, float* dscale_partial,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return clipk_simce_grad_scaled(X, Mx, Y, Ny, Yc, Nc, P, scale, label_offset, lse_x, lse_y, w_row, w_col, inv_bg, nullptr,
                                 dX, dscale_partial, workspace, workspace_bytes, stream);
}

extern "C" int clipk_simce_grad_scaled(const float* X, int Mx, const float* Y, int Ny, const float* Yc, int Nc,
                                       int P, const float* scale, int label_offset,
                                       const float* lse_x, const float* lse_y, float w_row, float w_col, float inv_bg,
                                       const float* upstream, float* dX, float* dscale_partial,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!X || !Y || !scale || !lse_x || !lse_y || !dX || !workspace || Nc < 0 || (Nc > 0 && !Yc))
    return CLIPK_ERR_BAD_ARG;
  Plan pl;
  if (!make_plan(Mx, Ny + Nc, P, &pl)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(X) || !aligned16(Y) || (Yc && !aligned16(Yc)) || !aligned16(dX) || !aligned16(workspace))
    return CLIPK_ERR_BAD_ARG;
  if (use_tiled_grad(Mx, Ny + Nc, P, pl.Pw)) {
    int nqb, ks3, tps, nt;
    clipk_simce_grad_tiled_plan(Mx, Ny + Nc, &nqb, &ks3, &tps, &nt);
    if (workspace_bytes < (size_t)ks3 * Mx * ((size_t)P + 1) * sizeof(float)) return CLIPK_ERR_BAD_ARG;
    float* slab = (float*)workspace;
    float* dscp = slab + (size_t)ks3 * Mx * P;
    int rc = clipk_simce_grad_tiled_launch(X, Mx, Y, Ny, Yc, Nc, P, scale, label_offset, lse_x, lse_y, w_row, w_col,
                                           inv_bg, upstream, slab, dscp, stream);
    if (rc) return rc;
    long n4 = (long)Mx * P / 4;
    int blocks = (int)((n4 + 63) / 64); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(simce_grad_finalize, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)slab,
                       (const float*)dscp, ks3, Mx, P, scale, dX, dscale_partial, 0L, 0L);
    return clipk_check_launch();
  }
  if (pl.Pw % 32) return CLIPK_ERR_UNSUPPORTED;             // P > 512 with a P slice that is not a multiple of 32
  const size_t need = (size_t)pl.ksplit * Mx * ((size_t)P + 1) * sizeof(float);
  if (workspace_bytes < need) return CLIPK_ERR_BAD_ARG;
  SP sp{};
  sp.X = X; sp.Mx = Mx; sp.Y = Y; sp.Ny = Ny; sp.Yc = Yc ? Yc : Y; sp.Nc = Nc;
  sp.P = P; sp.Pw = pl.Pw; sp.NW = pl.NW; sp.scale = scale; sp.label_offset = label_offset;
  sp.lse_x = lse_x; sp.lse_y = lse_y; sp.w_row = w_row; sp.w_col = w_col; sp.inv_bg = inv_bg; sp.upstream = upstream;
  sp.slab = (float*)workspace; sp.dsc_part = (float*)workspace + (size_t)pl.ksplit * Mx * P;
  sp.ksplit = pl.ksplit; sp.tiles_per_split = pl.tps; sp.ntiles = pl.ntiles;
  int rc = launch<MODE_GRAD>(sp, pl, (hipStream_t)stream);
  if (rc) return rc;
  long n4 = (long)Mx * P / 4;
  int blocks = (int)((n4 + 63) / 64); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(simce_grad_finalize, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float*)sp.slab, (const float*)sp.dsc_part, pl.ksplit, Mx, P, scale, dX,
                     dscale_partial, 0L, 0L);
  return clipk_check_launch();
}

// ---- batched form: npairs same-shape problems (X = E[pairs[2i]], Y = E[pairs[2i+1]]) in ONE launch each for the LSE
// pass and the gradient pass (blockIdx.z = problem).  The tri-modal ContrastiveModel of current/tf_clip_codes (1).ipynb
// :13150-13163 is three pairwise symmetric losses on one logit_scale = six directed problems.
extern "C" size_t clipk_simce_pairs_workspace(int npairs, int B, int P) {
  Plan pl;
  if (npairs <= 0 || npairs > MAXZ || !make_plan(B, B, P, &pl)) return 0;
  return (size_t)npairs * pl.ksplit * B * ((size_t)P + 2) * sizeof(float);
}

extern "C" int clipk_simce_lse_pairs(const float* E, int nmod, int B, int P, const int* pairs, int npairs,
                                     const float* scale, float* lse, float* pos, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  if (!E || !pairs || !scale || !lse || !pos || !workspace || npairs <= 0 || npairs > MAXZ || nmod <= 0)
    return CLIPK_ERR_BAD_ARG;
  Plan pl;
  if (!make_plan(B, B, P, &pl)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(E) || (((size_t)B * P * 4) & 15)) return CLIPK_ERR_BAD_ARG;
  const long z_part = (long)pl.ksplit * B * 2;
  if (workspace_bytes < (size_t)npairs * z_part * sizeof(float)) return CLIPK_ERR_BAD_ARG;
  SP sp{};
  sp.Mx = B; sp.Ny = B; sp.Nc = 0;
  sp.P = P; sp.Pw = pl.Pw; sp.NW = pl.NW; sp.scale = scale; sp.label_offset = 0;
  sp.part_ml = (float*)workspace; sp.pos = pos;
  sp.ksplit = pl.ksplit; sp.tiles_per_split = pl.tps; sp.ntiles = pl.ntiles;
  sp.nz = npairs; sp.z_part = z_part; sp.z_pos = B;
  for (int i = 0; i < npairs; ++i) {
    const int a = pairs[2 * i], b = pairs[2 * i + 1];
    if (a < 0 || a >= nmod || b < 0 || b >= nmod) return CLIPK_ERR_BAD_ARG;
    sp.Xz[i] = E + (size_t)a * B * P; sp.Yz[i] = E + (size_t)b * B * P;
  }
  int rc = launch<MODE_LSE>(sp, pl, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(simce_lse_finalize, dim3((B + 3) / 4, npairs), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, pl.ksplit, B, lse, z_part);
  return clipk_check_launch();
}

extern "C" int clipk_simce_grad_pairs(const float* E, int nmod, int B, int P, const int* pairs, const int* reverse,
                                      int npairs, const float* scale, const float* lse /*[npairs][B]*/, float w_row,
                                      float w_col, float inv_bg, float* dX /*[npairs][B][P]*/,
                                      float* dscale_partial /*[npairs][B]*/, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (!E || !pairs || !reverse || !scale || !lse || !dX || !workspace || npairs <= 0 || npairs > MAXZ || nmod <= 0)
    return CLIPK_ERR_BAD_ARG;
  Plan pl;
  if (!make_plan(B, B, P, &pl)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(E) || !aligned16(dX) || !aligned16(workspace) || (((size_t)B * P * 4) & 15)) return CLIPK_ERR_BAD_ARG;
  const long z_slab = (long)pl.ksplit * B * P, z_dsc = (long)pl.ksplit * B;
  if (workspace_bytes < (size_t)npairs * (z_slab + z_dsc) * sizeof(float)) return CLIPK_ERR_BAD_ARG;
  SP sp{};
  sp.Mx = B; sp.Ny = B; sp.Nc = 0;
  sp.P = P; sp.Pw = pl.Pw; sp.NW = pl.NW; sp.scale = scale; sp.label_offset = 0;
  sp.w_row = w_row; sp.w_col = w_col; sp.inv_bg = inv_bg;
  sp.slab = (float*)workspace; sp.dsc_part = (float*)workspace + (size_t)npairs * z_slab;
  sp.ksplit = pl.ksplit; sp.tiles_per_split = pl.tps; sp.ntiles = pl.ntiles;
  sp.nz = npairs; sp.z_slab = z_slab; sp.z_dsc = z_dsc;
  for (int i = 0; i < npairs; ++i) {
    const int a = pairs[2 * i], b = pairs[2 * i + 1], r = reverse[i];
    if (a < 0 || a >= nmod || b < 0 || b >= nmod || r < 0 || r >= npairs) return CLIPK_ERR_BAD_ARG;
    sp.Xz[i] = E + (size_t)a * B * P; sp.Yz[i] = E + (size_t)b * B * P;
    sp.lse_xz[i] = lse + (size_t)i * B;            // rows of X over the keys Y
    sp.lse_yz[i] = lse + (size_t)r * B;            // each key of Y over the queries X = the reverse problem's LSE
  }
  int rc = launch<MODE_GRAD>(sp, pl, (hipStream_t)stream);
  if (rc) return rc;
  long n4 = (long)B * P / 4;
  int blocks = (int)((n4 + 63) / 64); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(simce_grad_finalize, dim3(blocks, npairs), dim3(256), 0, (hipStream_t)stream,
                     (const float*)sp.slab, (const float*)sp.dsc_part, pl.ksplit, B, P, scale, dX, dscale_partial,
                     z_slab, z_dsc);
  return clipk_check_launch();
}

extern "C" int clipk_sim_logits(const float* X, int Mx, const float* Y, int Ny, int P, const float* scale,
                                float* S, int64_t lds, void* stream) {
  if (!X || !Y || !scale || !S) return CLIPK_ERR_BAD_ARG;
  // swapped roles: the kernel's "queries" (lanes) are the columns of S so stores are row-contiguous
  Plan pl;
  if (!make_plan(Ny, Mx, P, &pl)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(X) || !aligned16(Y)) return CLIPK_ERR_BAD_ARG;
  SP sp{};
  sp.X = Y; sp.Mx = Ny; sp.Y = X; sp.Ny = Mx; sp.Yc = X; sp.Nc = 0;
  sp.P = P; sp.Pw = pl.Pw; sp.NW = pl.NW; sp.scale = scale; sp.label_offset = 0;
  sp.S = S; sp.lds_out = lds;
  sp.ksplit = pl.ksplit; sp.tiles_per_split = pl.tps; sp.ntiles = pl.ntiles;
  return launch<MODE_LOGITS>(sp, pl, (hipStream_t)stream);
}

extern "C" int clipk_gemm_f32_nt(const float* X, int M, const float* W, int N, int K, const float* bias,
                                 const float* addend, const float* addend_scale, float* out, void* stream) {
  if (!X || !W || !out) return CLIPK_ERR_BAD_ARG;
  Plan pl;
  if (!make_plan(N, M, K, &pl)) return CLIPK_ERR_UNSUPPORTED;       // kernel "queries" = output columns (rows of W)
  if (!aligned16(X) || !aligned16(W)) return CLIPK_ERR_BAD_ARG;
  SP sp{};
  sp.X = W; sp.Mx = N; sp.Y = X; sp.Ny = M; sp.Yc = X; sp.Nc = 0;
  sp.P = K; sp.Pw = pl.Pw; sp.NW = pl.NW; sp.scale = nullptr; sp.label_offset = 0;
  sp.S = out; sp.lds_out = N;
  sp.ep_bias = bias; sp.ep_add = addend; sp.ep_add_scale = addend_scale;
  sp.ksplit = pl.ksplit; sp.tiles_per_split = pl.tps; sp.ntiles = pl.ntiles;
  return launch<MODE_LOGITS>(sp, pl, (hipStream_t)stream);
}
