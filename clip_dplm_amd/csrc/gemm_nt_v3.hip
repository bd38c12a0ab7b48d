// gemm_nt_v3.hip — 256 x 256 tile, 8 waves, phase-interleaved schedule for clipk_gemm_nt (K % 32 == 0, K >= 128).
// Same contract and epilogue as gemm_nt_v2.hip; selected by CLIPK_GEMM_V3 (gemm_nt.hip).
//
// Why a second structure: the 128 x 128 kernel tops out where the CU's L2 -> LDS path saturates.
// A 256 x 256 tile halves the operand bytes per FLOP, but only pays with ~1 workgroup per CU if the loads stay in
// flight across barriers and the two waves of each SIMD alternate between "fetch fragments" and "issue MFMAs"
// (cdna_hip_programming.md §5, 8-phase template).  Structure:
//   * 8 waves = 2 (m) x 4 (n), wave tile 128 m x 64 n, 128 accumulator VGPRs; one K-tile (BK = 64) = 4 phases of
//     16 MFMAs, each phase one quadrant (64 m x 32 n) of the wave tile: (m0,n0) (m0,n1) (m1,n1) (m1,n0);
//   * LDS = 2 buffers x 4 half-tiles of 16 KiB.  A half-tile is defined by CONSUMPTION order, not by position:
//     "X mh" holds the mh-th 64 rows of BOTH m-waves, "W nh" the nh-th 32 rows of all four n-waves, so a half-tile
//     is dead after the phase that read it and can be refilled while the rest of the buffer is still in use;
//   * every phase refills one half-tile (2 x global_load_lds_dwordx4 per lane) two K-tiles ahead; the only vmcnt
//     waits are a counted vmcnt(6) once per K-tile (three half-tiles stay in flight) — never 0 in the main loop;
//   * raw s_barrier twice per phase; the m = 1 waves run one barrier behind the m = 0 waves, so on every SIMD one
//     wave is in its MFMA block (s_setprio 1) while the other fetches fragments and issues the refill.
// Hazard bookkeeping (phases numbered 4T + ph for K-tile T):
//   RAW  tile T+1 is complete at the vmcnt(6) of phase 4T+3, both wave groups have executed that wait before the
//        barrier that opens phase 4T+4, where it is first read;
//   WAR  W nh0: read first in phase 4T (retired by lgkmcnt(8) before that phase's barrier), refilled in 4T+1;
//        X mh0: read 4T, refilled 4T+2;  W nh1: read 4T+1, refilled 4T+3;  X mh1: read 4T+2, refilled 4T+4.
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;        // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;       // X mh0 | X mh1 | W nh0 | W nh1
constexpr int LDS_BYTES = 2 * BUF_BYTES;        // 128 KiB

__device__ __attribute__((aligned(16))) const unsigned int kZeroChunk[4] = {0u, 0u, 0u, 0u};

struct Params {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  int M, N, K;
  EpiArgs e;
  int ntn;
};

__device__ __forceinline__ void glds16(const void* gptr, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(gptr, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// one quadrant: 2 n-tiles x 4 m-tiles x 2 k-halves = 16 MFMAs (k outer so dependent accumulations sit 8 apart)
template <int NH, int MH>
__device__ __forceinline__ void quad(f32x4 (&acc)[4][8], const bf16x8 (&wf)[2][2][2], const bf16x8 (&xf)[4][2]) {
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[NH * 2 + t][MH * 4 + j] =
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[NH][t][kk], xf[j][kk], acc[NH * 2 + t][MH * 4 + j], 0, 0, 0);
}

#define CLIPK_BAR() __builtin_amdgcn_s_barrier()
#define CLIPK_SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_nt_v3_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = p.M, N = p.N, K = p.K;

  // ---- LDS-DMA assignment: wave w fills pieces 2w, 2w+1 (8 rows x 128 B each) of every half-tile.
  // lane -> (row in piece = lane>>3, physical 16-B slot = lane&7); source chunk = slot ^ ((row>>1)&7)
  const int prow = lane >> 3, pslot = lane & 7;
  const unsigned short* xs0[2]; const unsigned short* xs1[2];
  const unsigned short* ws0[2]; const unsigned short* ws1[2];
  int kch[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (2 * wid + i) + prow;                     // row of the half-tile image
    kch[i] = (pslot ^ ((r >> 1) & 7)) * 8;
    const int mb = m0 + (r >> 6) * 128 + (r & 63);              // X half mh: rows of m-wave r>>6
    const int nb = n0 + (r >> 5) * 64 + (r & 31);               // W half nh: rows of n-wave r>>5
    int ma = mb, mc = mb + 64, na = nb, nc = nb + 32;
    ma = ma < M ? ma : M - 1; mc = mc < M ? mc : M - 1;
    na = na < N ? na : N - 1; nc = nc < N ? nc : N - 1;
    xs0[i] = p.A + (long)ma * p.lda + kch[i];
    xs1[i] = p.A + (long)mc * p.lda + kch[i];
    ws0[i] = p.B + (long)na * p.ldb + kch[i];
    ws1[i] = p.B + (long)nc * p.ldb + kch[i];
  }
  const unsigned short* zsrc = reinterpret_cast<const unsigned short*>(kZeroChunk);
  auto stage = [&](const unsigned short* const (&src)[2], int T, int region) {
    const int k0 = T * BK;
    char* dst = smem + (T & 1) * BUF_BYTES + region + wid * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned short* g = (k0 + kch[i] < K) ? src[i] + k0 : zsrc;   // K tail (K % 64 == 32): zero x zero
      glds16(g, dst + i * 1024);
    }
  };
  constexpr int XH0 = 0, XH1 = HALF_BYTES, WH0 = 2 * HALF_BYTES, WH1 = 3 * HALF_BYTES;

  f32x4 acc[4][8];          // [n-tile i][m-tile j]: rows n = 4g+r, col m = lane&15
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fch = lane >> 4, lane_sw = (frow >> 1) & 7;
  int xo[2], wo[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int choff = ((kk * 4 + fch) ^ lane_sw) << 4;
    xo[kk] = (wm * 64 + frow) * 128 + choff;
    wo[kk] = WH0 + (wn * 32 + frow) * 128 + choff;
  }

  bf16x8 xf[4][2], wf[2][2][2];
  const int nk = (K + BK - 1) / BK;

  // TM 0: steady state; 1: K-tile nk-2 (only the last half-tile of tile nk-1 left to fetch); 2: last K-tile
  auto tile_body = [&](auto mode_c, int T) {
    constexpr int TM = decltype(mode_c)::value;
    const char* buf = smem + (T & 1) * BUF_BYTES;
    // ---- phase 0: quadrant (n0, m0); fetch W nh0 (4 reads, first) + X mh0 (8 reads); refill X mh1 of tile T+1
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[0][t][kk] = *reinterpret_cast<const bf16x8*>(buf + wo[kk] + t * 2048);
    CLIPK_SB();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(buf + xo[kk] + XH0 + j * 2048);
    if (TM <= 1) stage(xs1, T + 1, XH1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          // W nh0 reads retired: refilled next phase
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    quad<0, 0>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    // ---- phase 1: quadrant (n1, m0); fetch W nh1; refill W nh0 of tile T+2
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        wf[1][t][kk] = *reinterpret_cast<const bf16x8*>(buf + wo[kk] + HALF_BYTES + t * 2048);
    if (TM == 0) stage(ws0, T + 2, WH0);
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    quad<1, 0>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    // ---- phase 2: quadrant (n1, m1); fetch X mh1; refill X mh0 of tile T+2
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(buf + xo[kk] + XH1 + j * 2048);
    if (TM == 0) stage(xs0, T + 2, XH0);
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    quad<1, 1>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    // ---- phase 3: quadrant (n0, m1); nothing to fetch; refill W nh1 of tile T+2; tile T+1 must be complete
    if (TM == 0) {
      stage(ws1, T + 2, WH1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (TM == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    quad<0, 1>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
  };

  // ---- prologue: tile 0 complete, three half-tiles of tile 1 in flight (nk >= 2 guaranteed by the launcher)
  stage(ws0, 0, WH0); stage(xs0, 0, XH0); stage(ws1, 0, WH1); stage(xs1, 0, XH1);
  stage(ws0, 1, WH0); stage(xs0, 1, XH0); stage(ws1, 1, WH1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
  if (wm == 1) CLIPK_BAR();                                     // m = 1 waves run one barrier behind
  for (int T = 0; T < nk - 2; ++T) tile_body(std::integral_constant<int, 0>{}, T);
  tile_body(std::integral_constant<int, 1>{}, nk - 2);
  tile_body(std::integral_constant<int, 2>{}, nk - 1);
  if (wm == 0) CLIPK_BAR();                                     // re-align the two groups
  __syncthreads();   // compiler-visible drain: without it hipcc waits vmcnt(0) before every epilogue LDS read

  // ---- epilogue (gemm_epilogue.h): wave-private LDS slab, 16 rows at a time
  float* eb = reinterpret_cast<float*>(smem) + wid * 16 * EPI_LD;
  gemm_epilogue<MODE, 8>(p.e, acc, eb, lane, m0 + wm * 128, n0 + wn * 64 + (lane & 7) * 8);
}

template <int MODE>
void launch_v3(const Params& p, dim3 grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_v3_kernel<MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_v3_kernel<MODE>), grid, dim3(512), LDS_BYTES, st, p);
}

}  // namespace

// called by clipk_gemm_nt (gemm_nt.hip) after it validated the arguments (K % 32 == 0, K >= 128)
extern "C" int clipk_gemm_nt_v3_launch(const clipk_gemm_args* a, void* stream) {
  Params p;
  p.A = (const unsigned short*)a->A; p.lda = a->lda;
  p.B = (const unsigned short*)a->B; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.e = epi_args_from(a);
  const int ntm = (a->M + BM - 1) / BM, ntn = (a->N + BN - 1) / BN;
  p.ntn = ntn;
  const dim3 grid(ntm * ntn);
  hipStream_t st = (hipStream_t)stream;
  const char* ge = getenv("CLIPK_GEMM_EPI_GENERIC");
  const int mode = (ge && atoi(ge) == 1) ? EPI_GENERIC : epi_mode_for(a);
  if (mode == EPI_PLAIN) launch_v3<EPI_PLAIN>(p, grid, st);
  else if (mode == EPI_RES32) launch_v3<EPI_RES32>(p, grid, st);
  else if (mode == EPI_GELU_PRE) launch_v3<EPI_GELU_PRE>(p, grid, st);
  else if (mode == EPI_DGELU) launch_v3<EPI_DGELU>(p, grid, st);
  else launch_v3<EPI_GENERIC>(p, grid, st);
  return clipk_check_launch();
}
