// gemm_nt_v3.hip — 256 x 256 tile variant of clipk_gemm_nt for the large-M Linear layers (K % 32 == 0).
//
// Why a bigger tile: profiles/r01 + tools/exp_gemm.py show the 128 x 128 kernel's main loop running at the
// L2 -> LDS bandwidth of the chip (15.3 TB/s of operand tiles at K = N = 1920, ~90 % of what the fabric delivers),
// i.e. bound by bytes per FLOP (32 KiB of staged operands per 2.1 MFLOP step), not by MFMA issue.  A 256 x 256 tile
// stages 64 KiB per 8.4 MFLOP step: half the L2 traffic per FLOP.
//
// Structure: 8 waves (2 along M x 4 along N, each wave 128 x 64 = 8 x 4 MFMA 16x16x32 tiles, 128 accumulator
// registers), BK = 64, two 64 KiB LDS stages filled by LDS-DMA (global_load_lds_dwordx4, swizzle on the source
// address as in v2).  The DMA of step k+1 stays in flight across the barrier while step k's MFMAs run: counted
// `s_waitcnt vmcnt(8)` + raw s_barrier (a __syncthreads() would drain it).  One workgroup per CU, so the overlap
// of loads and MFMAs is inside the workgroup.  Epilogue identical to v2 (swapped operand roles, LDS slab per wave,
// 16-byte row-contiguous global accesses, fused bias / GELU / GELU' / residual / cast).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int A_TILE_BYTES = BM * BK * 2;               // 32 KiB
constexpr int B_TILE_BYTES = BN * BK * 2;               // 32 KiB
constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;              // 128 KiB
constexpr int EPI_LD = 68;

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
};

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(gptr, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int ACT_T, int DACT_T>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_v3_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = p.M, N = p.N, K = p.K;

  // ---- LDS-DMA assignment: wave w, piece i (0..3) fills rows 8*(4w+i) .. +7 of each 256-row operand tile
  const int prow = lane >> 3, pslot = lane & 7;
  const unsigned short* asrc[4];
  const unsigned short* bsrc[4];
  int kchunk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wid + i) + prow;
    kchunk[i] = (pslot ^ ((row >> 1) & 7)) * 8;
    int ra = m0 + row; ra = ra < M ? ra : M - 1;
    int rb = n0 + row; rb = rb < N ? rb : N - 1;
    asrc[i] = p.A + (long)ra * p.lda;
    bsrc[i] = p.B + (long)rb * p.ldb;
  }

  f32x4 acc[4][8];          // [n-tile i][m-tile j]: rows n = 4g+r, col m = lane&15
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lane_sw = (lane >> 1) & 7;
  const int frow = lane & 15, fch = lane >> 4;
  const int x_frag_off = (wm * 128 + frow) * 128;                  // activation rows (B operand)
  const int w_frag_off = A_TILE_BYTES + (wn * 64 + frow) * 128;    // weight rows (A operand)

  const int nk = (K + BK - 1) / BK;
  auto issue = [&](int kt, char* stage) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = k0 + kchunk[i];
      k = k < K ? k : 0;                                 // K tail: slot is never read, keep the address valid
      glds16(asrc[i] + k, stage + (4 * wid + i) * 1024);
      glds16(bsrc[i] + k, stage + A_TILE_BYTES + (4 * wid + i) * 1024);
    }
  };

  issue(0, smem);
  for (int kt = 0; kt < nk; ++kt) {
    const char* cur = smem + (kt & 1) * STAGE_BYTES;
    if (kt + 1 < nk) {
      issue(kt + 1, smem + ((kt + 1) & 1) * STAGE_BYTES);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // everything but the 8 youngest DMA (= next stage) has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const int ksub = (kt * BK + 32 < K) ? 2 : 1;
    for (int kk = 0; kk < ksub; ++kk) {
      const int choff = (((kk * 4 + fch) ^ lane_sw) << 4);
      bf16x8 wf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(cur + w_frag_off + t * 2048 + choff);
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {                   // two halves of the wave's 8 m-tiles: 16 fragment registers live
        bf16x8 xf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          xf[t] = *reinterpret_cast<const bf16x8*>(cur + x_frag_off + (jh * 4 + t) * 2048 + choff);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[i][jh * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[t], acc[i][jh * 4 + t], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // stage (kt&1) fully read before step kt+2's DMA refills it
  }

  // ---- epilogue, one 16-row m-tile at a time through a wave-private LDS slab [16 m][64 n (+4)] f32
  float* eb = reinterpret_cast<float*>(smem) + wid * 16 * EPI_LD;
  const float alpha = p.alpha;
  const int g = lane >> 4, li = lane & 15;
  const int ecol = (lane & 7) * 8;
  const int gn = n0 + wn * 64 + ecol;
  float bv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) bv[c] = 0.f;
  if (p.bias && gn < N) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + gn);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) { bv[c] = b0[c]; bv[4 + c] = b1[c]; }
  }
  const int act = (ACT_T >= 0) ? ACT_T : p.act;
  const int dact = (DACT_T >= 0) ? DACT_T : p.dact;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(eb + li * EPI_LD + i * 16 + 4 * g) = acc[i][j] * alpha;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = half * 8 + (lane >> 3);
      const int gm = m0 + wm * 128 + j * 16 + row;
      float v[8];
      {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(eb + row * EPI_LD + ecol);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(eb + row * EPI_LD + ecol + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[c] = v0[c] + bv[c]; v[4 + c] = v1[c] + bv[4 + c]; }
      }
      if (gm < M && gn < N) {
        if (p.out_preact) {
          u32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
          *reinterpret_cast<u32x4*>(p.out_preact + (long)gm * p.ldp + gn) = o;
        }
        if (act != CLIPK_ACT_NONE) {
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = act_apply(v[c], act);
        }
        if ((DACT_T < 0 || DACT_T != CLIPK_ACT_NONE) && p.dact_aux) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(p.dact_aux + (long)gm * p.ldd + gn);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v[2 * c] *= act_grad(bf16_to_f32((unsigned short)(a[c] & 0xffffu)), dact);
            v[2 * c + 1] *= act_grad(bf16_to_f32((unsigned short)(a[c] >> 16)), dact);
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
  }
}

template <int ACT_T, int DACT_T>
void launch(const Params& p, int grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_v3_kernel<ACT_T, DACT_T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_v3_kernel<ACT_T, DACT_T>), dim3(grid), dim3(NTHREADS), LDS_BYTES, st, p);
}

}  // namespace

extern "C" int clipk_gemm_nt_v3_launch(const clipk_gemm_args* a, void* stream) {
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
  const int ntm = (a->M + BM - 1) / BM, ntn = (a->N + BN - 1) / BN;
  p.ntn = ntn;
  hipStream_t st = (hipStream_t)stream;
  const bool has_dact = a->dact_aux != nullptr;
  if (a->act == CLIPK_ACT_GELU && !has_dact) launch<CLIPK_ACT_GELU, CLIPK_ACT_NONE>(p, ntm * ntn, st);
  else if (a->act == CLIPK_ACT_NONE && has_dact && a->dact == CLIPK_ACT_GELU) launch<CLIPK_ACT_NONE, CLIPK_ACT_GELU>(p, ntm * ntn, st);
  else if (a->act == CLIPK_ACT_NONE && !has_dact) launch<CLIPK_ACT_NONE, CLIPK_ACT_NONE>(p, ntm * ntn, st);
  else launch<-1, -1>(p, ntm * ntn, st);
  return clipk_check_launch();
}
