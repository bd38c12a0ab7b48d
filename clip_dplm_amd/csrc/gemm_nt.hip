// gemm_nt.hip — C[M,N] = epilogue(A[M,K] · B[N,K]^T): the Linear layer of every encoder / head.
//
// gfx950 design:
//   * 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA
//     16x16x32 bf16 tiles, 64 f32 accumulator registers);
//   * K walked in 64-deep steps; both operands are K-contiguous (activations [M,K], nn.Linear weight
//     [N,K]) so fragments are single ds_read_b128 reads; LDS image is XOR-swizzled per 128-byte row
//     (chunk ^= (row>>1)&7) which tools/lds_conflicts.py shows conflict-free for the fragment reads;
//   * register-staged double buffering: the global loads of step k+1 are issued before the MFMAs of
//     step k and written to the other LDS buffer after them (one barrier per K-step);
//   * epilogue goes through LDS so that every global access of bias / residual / aux / output is a
//     whole 16-byte-per-lane row segment (8 lanes = one 128-B line of bf16), and fuses bias,
//     ReLU/GELU, activation-derivative, residual add and the bf16/f32 cast;
//   * 1-D grid with an XCD-aware bijective remap; N-tiles vary fastest so one XCD's L2 keeps the A
//     row panel while it sweeps the (small, L2-resident) weight.
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

extern "C" int clipk_gemm_nt_v2_launch(const clipk_gemm_args* a, void* stream);
extern "C" int clipk_gemm_nt_v3_launch(const clipk_gemm_args* a, void* stream);
extern "C" int clipk_gemm_nt_v4_launch(const clipk_gemm_args* a, void* stream);

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int A_TILE_BYTES = BM * BK * 2;               // 16 KiB
constexpr int B_TILE_BYTES = BN * BK * 2;               // 16 KiB
constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
constexpr int EPI_LD = 68;                              // f32 per staged row (64 + 4 pad: conflict-free)
constexpr int EPI_BYTES = 4 * 64 * EPI_LD * 4;          // 4 waves x 64 rows
constexpr int LDS_BYTES = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;

struct Params {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  void* C; long ldc; int c_f32;
  int M, N, K;
  const float* bias;
  int act;
  unsigned short* out_preact; long ldp;
  const unsigned short* dact_aux; long ldd; int dact;
  const void* residual; long ldr; int r_f32;
  float alpha;
  int ntn;
  unsigned drop_thr, drop_seed; float drop_scale; const unsigned* drop_epoch;
};

__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = p.M, N = p.N, K = p.K;

  // ---- staging assignment: chunk c = tid + 256*i -> row = (tid>>3) + 32*i, 16-byte chunk = tid&7
  const int srow = tid >> 3, skc = tid & 7;
  const unsigned short* ag[4];
  const unsigned short* bg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ra = m0 + srow + 32 * i; ra = ra < M ? ra : M - 1;
    int rb = n0 + srow + 32 * i; rb = rb < N ? rb : N - 1;
    ag[i] = p.A + (long)ra * p.lda;
    bg[i] = p.B + (long)rb * p.ldb;
  }
  const int soff0 = srow * 128 + ((skc ^ ((srow >> 1) & 7)) << 4);   // (row+32i)>>1 & 7 == (row>>1)&7

  u32x4 ra_[4], rb_[4];
  auto gload = [&](int kt) {
    int k = kt * BK + skc * 8;
    const bool ok = k < K;
    k = ok ? k : 0;                                  // always a valid address; zero-select afterwards
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 va = *reinterpret_cast<const u32x4*>(ag[i] + k);
      u32x4 vb = *reinterpret_cast<const u32x4*>(bg[i] + k);
      const u32x4 z = {0u, 0u, 0u, 0u};
      ra_[i] = ok ? va : z;
      rb_[i] = ok ? vb : z;
    }
  };
  auto lstore = [&](int buf) {
    char* base = smem + buf * STAGE_BYTES + soff0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(base + i * 4096) = ra_[i];
      *reinterpret_cast<u32x4*>(base + A_TILE_BYTES + i * 4096) = rb_[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lane_sw = (lane >> 1) & 7;                 // ((row)>>1)&7 with row = 16*t + (lane&15)
  const int frow = lane & 15, fch = lane >> 4;
  const int a_frag_off = (wm * 64 + frow) * 128;
  const int b_frag_off = A_TILE_BYTES + (wn * 64 + frow) * 128;

  const int nk = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) gload(kt + 1);
    const char* base = smem + (kt & 1) * STAGE_BYTES;
    const int ksub = (kt * BK + 32 < K) ? 2 : 1;       // K tail: skip an all-zero second half step
    for (int kk = 0; kk < ksub; ++kk) {
      const int choff = (((kk * 4 + fch) ^ lane_sw) << 4);
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = *reinterpret_cast<const bf16x8*>(base + a_frag_off + t * 2048 + choff);
        bf[t] = *reinterpret_cast<const bf16x8*>(base + b_frag_off + t * 2048 + choff);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> per-wave LDS slab (row = m, col = n) -> row-contiguous 8-wide pieces
  float* eb = reinterpret_cast<float*>(smem) + wid * 64 * EPI_LD;
  const float alpha = p.alpha;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        eb[(i * 16 + (lane >> 4) * 4 + r) * EPI_LD + j * 16 + (lane & 15)] = acc[i][j][r] * alpha;
  __syncthreads();

  const int erow = lane >> 3, ecol = (lane & 7) * 8;
  const int gn = n0 + wn * 64 + ecol;
  if (gn >= N) return;
  float bv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bv[c] = 0.f;
  if (p.bias) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + gn);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { bv[c] = b0[c]; bv[4 + c] = b1[c]; }
  }
#pragma unroll
  for (int ps = 0; ps < 8; ++ps) {
    const int row = ps * 8 + erow;
    const int gm = m0 + wm * 64 + row;
    if (gm >= M) continue;
    float v[8];
    {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(eb + row * EPI_LD + ecol);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(eb + row * EPI_LD + ecol + 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] = v0[c] + bv[c]; v[4 + c] = v1[c] + bv[4 + c]; }
    }
    if (p.out_preact) {
      u32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
      *reinterpret_cast<u32x4*>(p.out_preact + (long)gm * p.ldp + gn) = o;
    }
    if (p.act != CLIPK_ACT_NONE) {
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = act_apply(v[c], p.act);
    }
    if (p.drop_thr) {                                         // nn.Dropout on this tensor: index = m * N + n
      const unsigned long long base = (unsigned long long)gm * (unsigned)p.N + (unsigned)gn;
      const unsigned dseed = drop_seed_eff(p.drop_seed, p.drop_epoch);
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] *= drop_mul(dseed, base + c, p.drop_thr, p.drop_scale);
    }
    if (p.dact_aux) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(p.dact_aux + (long)gm * p.ldd + gn);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v[2 * c] *= act_grad(bf16_to_f32((unsigned short)(a[c] & 0xffffu)), p.dact);
        v[2 * c + 1] *= act_grad(bf16_to_f32((unsigned short)(a[c] >> 16)), p.dact);
      }
    }
    if (p.residual) {
      if (p.r_f32) {
        const float* r = reinterpret_cast<const float*>(p.residual) + (long)gm * p.ldr + gn;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(r);
        const f32x4 r1 = *reinterpret_cast<const f32x4*>(r + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[c] += r0[c]; v[4 + c] += r1[c]; }
      } else {
        const u32x4 a = *reinterpret_cast<const u32x4*>(
            reinterpret_cast<const unsigned short*>(p.residual) + (long)gm * p.ldr + gn);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[2 * c] += bf16_to_f32((unsigned short)(a[c] & 0xffffu));
          v[2 * c + 1] += bf16_to_f32((unsigned short)(a[c] >> 16));
        }
      }
    }
    if (p.c_f32) {
      float* c = reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn;
      *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(c + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      u32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
      *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.C) + (long)gm * p.ldc + gn) = o;
    }
  }
}

}  // namespace

static int gemm_nt_one(const clipk_gemm_args* a, void* stream);

// The fast kernels address operands and outputs through buffer descriptors with 32-bit byte offsets (outputs and
// epilogue operands < 2 GiB, the activation operand < 4 GiB); a larger problem (ESM-2-650M at B = 256, L = 1024: the
// 262144 x 5120 bf16 FFN activation is 2.7 GB) is cut along M into slabs that fit — same kernels, same results, instead
// of falling back to the run-time epilogue (measured: 5.5 ms -> see DESIGN.md §3.2).
extern "C" int clipk_gemm_nt(const clipk_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) return CLIPK_ERR_BAD_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return CLIPK_ERR_BAD_ARG;
  if (!(a->drop_p >= 0.f) || a->drop_p >= 1.f) return CLIPK_ERR_BAD_ARG;      // nn.Dropout(p = 1) zeroes: not a mask
  if (a->aux_dtype != CLIPK_BF16 && a->aux_dtype != CLIPK_U8) return CLIPK_ERR_BAD_ARG;
  const long c_elt = a->c_dtype == CLIPK_F32 ? 4 : 2, r_elt = a->r_dtype == CLIPK_F32 ? 4 : 2;
  long row_bytes = a->ldc * c_elt;
  if (a->out_preact && a->ldp * 2 > row_bytes) row_bytes = a->ldp * 2;
  if (a->dact_aux && a->ldd * 2 > row_bytes) row_bytes = a->ldd * 2;
  if (a->residual && a->ldr * r_elt > row_bytes) row_bytes = a->ldr * r_elt;
  if (a->lda * 2 > 2 * row_bytes) row_bytes = a->lda;          // A operand: 4 GiB limit = rows * lda * 2 < 2^32
  const long max_rows = ((0x7fffffffL - (long)a->N * 4) / row_bytes) / 256 * 256;
  if (a->M <= max_rows || max_rows <= 0) return gemm_nt_one(a, stream);
  for (long m0 = 0; m0 < a->M; m0 += max_rows) {
    clipk_gemm_args c = *a;
    c.M = (int)((a->M - m0) < max_rows ? (a->M - m0) : max_rows);
    c.A = (const char*)a->A + m0 * a->lda * 2;
    c.C = (char*)a->C + m0 * a->ldc * c_elt;
    const long aux_elt = a->aux_dtype == CLIPK_U8 ? 1 : 2;
    if (a->out_preact) c.out_preact = (char*)a->out_preact + m0 * a->ldp * aux_elt;
    if (a->dact_aux) c.dact_aux = (const char*)a->dact_aux + m0 * a->ldd * aux_elt;
    if (a->residual) c.residual = (const char*)a->residual + m0 * a->ldr * r_elt;
    c.rope_row0 = a->rope_row0 + (int)m0;                 // positions count from the first row of the whole problem
    // dropout masks are indexed by the global element position m * N + n: not supported across slabs
    if (a->drop_p > 0.f) return CLIPK_ERR_UNSUPPORTED;
    const int rc = gemm_nt_one(&c, stream);
    if (rc) return rc;
  }
  return CLIPK_OK;
}

static int gemm_nt_one(const clipk_gemm_args* a, void* stream) {
  if (!a || !a->A || !a->B || !a->C) return CLIPK_ERR_BAD_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return CLIPK_ERR_BAD_ARG;
  if ((a->K & 7) || (a->N & 7)) return CLIPK_ERR_UNSUPPORTED;
  if ((a->lda & 7) || (a->ldb & 7) || (a->ldc & 7)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(a->A) || !aligned16(a->B) || !aligned16(a->C)) return CLIPK_ERR_BAD_ARG;
  if (a->bias && !aligned16(a->bias)) return CLIPK_ERR_BAD_ARG;
  if (a->out_preact && (!aligned16(a->out_preact) || (a->ldp & 7))) return CLIPK_ERR_BAD_ARG;
  if (a->dact_aux && (!aligned16(a->dact_aux) || (a->ldd & 7))) return CLIPK_ERR_BAD_ARG;
  if (a->residual && (!aligned16(a->residual) || (a->ldr & 7))) return CLIPK_ERR_BAD_ARG;
  // fast paths, both LDS-DMA staged and needing whole 32-deep K steps:
  //   gemm_nt_v3.hip  persistent 256 x 256 tiles, phase-interleaved: problems with at least ~3/4 of a tile per CU
  //   gemm_nt_v2.hip  128 x 128 tiles, 4 workgroups per CU: everything else
  // option gemm_kernel = 2 / 3 forces the choice (tools/bench_kernels.py, tests), 1 the generic kernel.
  const int kmode = clipk_opt_get(OPT_GEMM_KERNEL);
  const bool force_v1 = kmode == 1;
  const int v3mode = kmode == 3 ? 1 : kmode == 2 ? 0 : -1;
  const long tiles256 = (long)((a->M + 255) / 256) * ((a->N + 255) / 256);
  const bool v3_ok = (a->K & 31) == 0 && a->K >= 160 && (long)a->M * a->lda * 2 < (1L << 32) &&
                     (long)a->N * a->ldb * 2 < (1L << 32);     // 32-bit buffer offsets
  //   gemm_nt_v4.hip  persistent 128 x 256 tiles, two 4-wave workgroups per CU (option gemm_kernel = 4 only)
  if (kmode == 4 && v3_ok && a->M >= 2048) return clipk_gemm_nt_v4_launch(a, stream);
  if (!force_v1 && v3_ok && ((v3mode == 1 && a->M >= 2048) || (v3mode < 0 && tiles256 >= 192)))
    return clipk_gemm_nt_v3_launch(a, stream);
  if (!force_v1 && (a->K & 31) == 0) return clipk_gemm_nt_v2_launch(a, stream);
  if (a->rope_cos) return CLIPK_ERR_UNSUPPORTED;           // the generic-K kernel has no rotation
  if (a->aux_dtype == CLIPK_U8 && (a->out_preact || a->dact_aux)) return CLIPK_ERR_UNSUPPORTED;   // ... and no 8-bit aux
  Params p;
  p.A = (const unsigned short*)a->A; p.lda = a->lda;
  p.B = (const unsigned short*)a->B; p.ldb = a->ldb;
  p.C = a->C; p.ldc = a->ldc; p.c_f32 = (a->c_dtype == CLIPK_F32);
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.bias = a->bias; p.act = a->act;
  p.out_preact = (unsigned short*)a->out_preact; p.ldp = a->ldp;
  p.dact_aux = (const unsigned short*)a->dact_aux; p.ldd = a->ldd; p.dact = a->dact;
  p.residual = a->residual; p.ldr = a->ldr; p.r_f32 = (a->r_dtype == CLIPK_F32);
  p.alpha = a->alpha;
  { const EpiArgs e = epi_args_from(a); p.drop_thr = e.drop_thr; p.drop_seed = e.drop_seed; p.drop_scale = e.drop_scale;
    p.drop_epoch = e.drop_epoch; }
  const int ntm = (a->M + BM - 1) / BM, ntn = (a->N + BN - 1) / BN;
  p.ntn = ntn;
  static std::atomic<uint64_t> attr_set{0};
  clipk_once_per_device(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  hipLaunchKernelGGL(gemm_nt_kernel, dim3(ntm * ntn), dim3(NTHREADS), LDS_BYTES, (hipStream_t)stream, p);
  return clipk_check_launch();
}
