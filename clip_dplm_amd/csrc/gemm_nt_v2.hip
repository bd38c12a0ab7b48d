// gemm_nt_v2.hip — fast path of clipk_gemm_nt for K % 32 == 0: same contract and epilogue as gemm_nt.hip.
//
// What changed against the register-staged v1 kernel, and why (measured on MI355X, profiles/):
//   * operand tiles go HBM/L2 -> LDS directly with global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction):
//     no staging VGPRs, no ds_write pass.  The LDS image stays the XOR-swizzled one of v1; because an LDS-DMA
//     writes lane-linear bytes, the swizzle is applied to the per-lane SOURCE address instead
//     (cdna_hip_programming.md §5.4 rule 21);
//   * one 32 KiB K-step buffer instead of two: 4 workgroups per CU fit (LDS 4 x 32 KiB, <= 128 VGPRs), and the
//     other resident workgroups' MFMAs cover this one's load phase — with K = 480..768 a tile has only 8-12
//     K-steps, so cross-workgroup overlap hides prologue / epilogue better than in-kernel double buffering;
//   * swapped operand roles (A fragment = weight rows, B fragment = activation rows): each lane then owns 4
//     consecutive output columns of one row, so the accumulator -> LDS staging of the epilogue is 16
//     ds_write_b128 per lane instead of 64 ds_write_b32.
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

constexpr int BN = 128, BK = 64;
constexpr int B_TILE_BYTES = BN * BK * 2;               // weights      [128 n][64 k]
// BM = 128: 4 waves (2x2), 32 KiB LDS, 4 workgroups / CU.   BM = 256: 8 waves (4x2), 48 KiB LDS, 2 workgroups / CU
// — a third fewer bytes per FLOP through the CU's L2->LDS path, which is what bounds this kernel.

struct Params {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  int M, N, K;
  EpiArgs e;
  int ntn;
  int stagger;      // s_sleep(127) units (~3.4 us each) per quarter-phase, 0 = off
};

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(gptr, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// MODE: compile-time epilogue variant (gemm_epilogue.h); EPI_GENERIC decides everything at run time
template <int BM, int STAGES, int MODE = EPI_GENERIC>
__global__ __launch_bounds__(BM * 2, (STAGES == 1 ? 4 : 2)) void gemm_nt_v2_kernel(const Params p) {
  constexpr int A_TILE_BYTES = BM * BK * 2;             // activations  [BM m][64 k]
  constexpr int STAGE_BYTES = A_TILE_BYTES + B_TILE_BYTES;
  constexpr int NWAVES = BM / 32;
  constexpr int A_PIECES = BM / 8 / NWAVES;             // 1-KiB LDS-DMA pieces of the A tile per wave (= 4)
  constexpr int B_PIECES = BN / 8 / NWAVES;             // 4 (BM = 128) or 2 (BM = 256)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = p.M, N = p.N, K = p.K;
  // Phase stagger: every workgroup runs the same load / MFMA / store sequence for the same time, so the
  // workgroups sharing a CU (and the whole chip) march in lockstep — all streaming outputs to HBM together, then
  // all on the matrix pipe together.  Delaying the first-round workgroups by 0..3 quarter periods de-phases
  // them for the rest of the launch (each CU slot keeps its offset as tiles are re-dispatched).
  if (p.stagger > 0 && blockIdx.x < 2048) {
    const int q = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < q * p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- LDS-DMA assignment: wave w, piece i (0..3) fills rows 8*(4w+i) .. +7 of each operand tile.
  // lane -> (row in piece = lane>>3, physical 16-B slot = lane&7); source chunk = slot ^ ((row>>1)&7)
  const int prow = lane >> 3, pslot = lane & 7;
  const unsigned short* asrc[A_PIECES];
  const unsigned short* bsrc[B_PIECES];
  int akchunk[A_PIECES], bkchunk[B_PIECES];
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int row = 8 * (A_PIECES * wid + i) + prow;
    akchunk[i] = (pslot ^ ((row >> 1) & 7)) * 8;
    int ra = m0 + row; ra = ra < M ? ra : M - 1;
    asrc[i] = p.A + (long)ra * p.lda;
  }
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int row = 8 * (B_PIECES * wid + i) + prow;
    bkchunk[i] = (pslot ^ ((row >> 1) & 7)) * 8;
    int rb = n0 + row; rb = rb < N ? rb : N - 1;
    bsrc[i] = p.B + (long)rb * p.ldb;
  }

  f32x4 acc[4][4];          // [n-tile i][m-tile j]: rows n = 4g+r, col m = lane&15
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lane_sw = (lane >> 1) & 7;
  const int frow = lane & 15, fch = lane >> 4;
  const int x_frag_off = (wm * 64 + frow) * 128;                   // activation rows (B operand)
  const int w_frag_off = A_TILE_BYTES + (wn * 64 + frow) * 128;    // weight rows (A operand)

  const int nk = (K + BK - 1) / BK;
  auto issue = [&](int kt, char* stage) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      int k = k0 + akchunk[i];
      k = k < K ? k : 0;                                 // K tail (K % 64 == 32): slot is never read, keep the address valid
      glds16(asrc[i] + k, stage + (A_PIECES * wid + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      int k = k0 + bkchunk[i];
      k = k < K ? k : 0;
      glds16(bsrc[i] + k, stage + A_TILE_BYTES + (B_PIECES * wid + i) * 1024);
    }
  };
  auto compute = [&](int kt, const char* stage) {
    const int ksub = (kt * BK + 32 < K) ? 2 : 1;
    for (int kk = 0; kk < ksub; ++kk) {
      const int choff = (((kk * 4 + fch) ^ lane_sw) << 4);
      bf16x8 wf[4], xf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        wf[t] = *reinterpret_cast<const bf16x8*>(stage + w_frag_off + t * 2048 + choff);
        xf[t] = *reinterpret_cast<const bf16x8*>(stage + x_frag_off + t * 2048 + choff);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  };
  if (STAGES == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      issue(kt, smem);
      __syncthreads();                                   // drains the LDS-DMA (vmcnt(0)) and publishes the tile
      compute(kt, smem);
      __syncthreads();                                   // everyone done reading before the next DMA lands
    }
  } else {
    // two stages: the LDS-DMA of step kt+1 stays in flight across the barrier while step kt's MFMAs run
    // (counted vmcnt + raw s_barrier: __syncthreads() would drain it, cdna_hip_programming.md §5)
    issue(0, smem);
    for (int kt = 0; kt < nk; ++kt) {
      char* cur = smem + (kt & 1) * STAGE_BYTES;
      if (kt + 1 < nk) {
        issue(kt + 1, smem + ((kt + 1) & 1) * STAGE_BYTES);
        if (A_PIECES + B_PIECES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                      // step kt's tile is complete for every wave
      compute(kt, cur);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // all fragment reads of this stage retired before it is refilled
    }
  }

  // ---- epilogue (gemm_epilogue.h): wave-private LDS slab, 16 rows at a time
  float* eb = reinterpret_cast<float*>(smem) + wid * 16 * EPI_LD;
  float bv[8];
  epi_load_bias(p.e, n0 + wn * 64 + (lane & 7) * 8, bv);
  gemm_epilogue<MODE, 4>(p.e, acc, eb, lane, m0 + wm * 64, n0 + wn * 64 + (lane & 7) * 8, bv);
}

}  // namespace

// called by clipk_gemm_nt (gemm_nt.hip) after it validated the arguments; returns CLIPK_OK / launch error
extern "C" int clipk_gemm_nt_v2_launch(const clipk_gemm_args* a, void* stream) {
  Params p;
  p.A = (const unsigned short*)a->A; p.lda = a->lda;
  p.B = (const unsigned short*)a->B; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.e = epi_args_from(a);
  const int ntn = (a->N + BN - 1) / BN;
  p.ntn = ntn;
  p.stagger = clipk_opt_get(OPT_GEMM_STAGGER);
  const bool big = clipk_opt_get(OPT_GEMM_BM) == 256;      // A/B switches for tools/bench_kernels.py
  const int stages = clipk_opt_get(OPT_GEMM_STAGES);
  const int mode = clipk_opt_get(OPT_GEMM_EPI_GENERIC) == 1 ? EPI_GENERIC : epi_mode_for(a);
  if (mode == EPI_UNSUPPORTED) return CLIPK_ERR_UNSUPPORTED;
  if (a->rope_cos && ((mode != EPI_ROPE && mode != EPI_ROPE_IL) || big || stages == 2)) return CLIPK_ERR_UNSUPPORTED;   // never unrotated
  static std::atomic<uint64_t> attr_set{0};
  clipk_once_per_device(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_v2_kernel<256, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 * BK * 2 + B_TILE_BYTES));
  });
  hipStream_t st = (hipStream_t)stream;
  if (big) {
    const int ntm = (a->M + 255) / 256;
    const int lds = 256 * BK * 2 + B_TILE_BYTES;
    if (stages == 2) hipLaunchKernelGGL((gemm_nt_v2_kernel<256, 2>), dim3(ntm * ntn), dim3(512), 2 * lds, st, p);
    else hipLaunchKernelGGL((gemm_nt_v2_kernel<256, 1>), dim3(ntm * ntn), dim3(512), lds, st, p);
  } else {
    const int ntm = (a->M + 127) / 128;
    const int lds = 128 * BK * 2 + B_TILE_BYTES;
    const dim3 grid(ntm * ntn), blk(256);
    if (stages == 2) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 2>), grid, blk, 2 * lds, st, p);
    else if (mode == EPI_PLAIN) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_PLAIN>), grid, blk, lds, st, p);
    else if (mode == EPI_RES32) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_RES32>), grid, blk, lds, st, p);
    else if (mode == EPI_GELU_PRE) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_GELU_PRE>), grid, blk, lds, st, p);
    else if (mode == EPI_DGELU) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_DGELU>), grid, blk, lds, st, p);
    else if (mode == EPI_RES16) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_RES16>), grid, blk, lds, st, p);
    else if (mode == EPI_PRES16) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_PRES16>), grid, blk, lds, st, p);
    else if (mode == EPI_ROPE) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_ROPE>), grid, blk, lds, st, p);
    else if (mode == EPI_GELU_D8) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_GELU_D8>), grid, blk, lds, st, p);
    else if (mode == EPI_DGELU8) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_DGELU8>), grid, blk, lds, st, p);
    else if (mode == EPI_PLAIN_NB) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_PLAIN_NB>), grid, blk, lds, st, p);
    else if (mode == EPI_ROPE_IL) hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1, EPI_ROPE_IL>), grid, blk, lds, st, p);
    else hipLaunchKernelGGL((gemm_nt_v2_kernel<128, 1>), grid, blk, lds, st, p);
  }
  return clipk_check_launch();
}
