// gemm_nt_v4.hip — 256 x 128 tile, 4 waves, TWO workgroups per CU, phase-interleaved main loop (K % 32 == 0, K >= 96).
// Experimental sibling of gemm_nt_v3.hip, selected by CLIPK_GEMM_V4=1 (gemm_nt.hip).
//
// Question it answers: gemm_nt_v3 (one 8-wave workgroup per CU) cannot overlap a tile's epilogue traffic with MFMA
// work — its time is FLOP / 1250 TFLOP/s + output bytes / 5 TB/s, the second term 16 ms of a 96 ms step.  Here a CU
// holds two independent 4-wave workgroups (80 KiB LDS and <= 256 VGPRs each, one wave of each per SIMD), so one
// can stream its output while the other multiplies — at the price of 1.5x the operand bytes per FLOP through the
// L2 -> LDS path (256 x 128 instead of 256 x 256).
//
// Main loop: the wave tile (128 m x 64 n), the four 16-MFMA quadrant phases per K-tile, the consumption-ordered
// half-tiles ("X mh": the mh-th 64 rows of both m-waves, 16 KiB; "W nh": the nh-th 32 rows of both n-waves, 8 KiB)
// and the buffer-descriptor LDS-DMA are those of gemm_nt_v3.hip.  With 80 KiB there is no room for two whole K-tiles
// (2 x 48 KiB), so the X half-tiles rotate through THREE 16-KiB slots and the W half-tiles through two 16-KiB
// buffers:
//     phase 0 of K-tile T: read W nh0(T), X mh0(T);  refill X mh0(T+1) into the slot X mh1(T-1) left in phase 2
//     phase 1            : read W nh1(T);            refill W nh0(T+2) over W nh0(T)   [retired by lgkmcnt(8)]
//     phase 2            : read X mh1(T);            refill X mh1(T+1) over X mh0(T)   [read two phases ago]
//     phase 3            : -                         refill W nh1(T+2) over W nh1(T)   [read two phases ago]
// Every X half-tile is issued one K-tile (four phases) before its first read, every W half-tile 6-7 phases before.
// Counted waits: vmcnt(8) at the end of phases 1 and 3 (the 8 newest loads = the two X / two W refills issued since
// the half-tile that is read next); one s_barrier per phase, between the wait and the MFMA block, is enough:
// a refill always targets a slot whose reads were retired before the PREVIOUS barrier.
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 128, BK = 64;
constexpr int XSLOT = 128 * BK * 2;             // 16 KiB
constexpr int WSLOT = 64 * BK * 2;              // 8 KiB
constexpr int W_BASE = 3 * XSLOT;               // X slots 0..2, then W [tile parity][nh]
constexpr int LDS_BYTES = 3 * XSLOT + 4 * WSLOT;   // 80 KiB

struct Params {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  int M, N, K;
  EpiArgs e;
  int ntn;
};

template <int NH, int MH, int KLO = 0>
__device__ __forceinline__ void quad(f32x4 (&acc)[4][8], const bf16x8 (&wf)[2][2][2], const bf16x8 (&xf)[4][2]) {
#pragma unroll
  for (int kk = KLO; kk < 2; ++kk)
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
__global__ __launch_bounds__(256, 2) void gemm_nt_v4_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int M = p.M, N = p.N, K = p.K;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDS-DMA assignment: 1-KiB pieces (8 rows x 128 B); an X half-tile has 16 (4 per wave), a W half-tile 8 (2)
  unsigned xv[4], wv[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * (4 * wid + i) + (lane >> 3);
    const int kch = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    xv[i] = (unsigned)(((long)((r >> 6) * 128 + (r & 63)) * p.lda + kch) * 2);      // rows of m-wave r>>6
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (2 * wid + i) + (lane >> 3);
    const int kch = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    wv[i] = (unsigned)(((long)((r >> 5) * 64 + (r & 31)) * p.ldb + kch) * 2);       // rows of n-wave r>>5
  }
  auto desc = [&](const unsigned short* base, long ld, int row0, int rows) {
    const long left = (long)rows - row0;
    const int bytes = left > 0 ? (int)(((left - 1) * ld + K) * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)row0 * ld), 0, bytes, 0x00020000);
  };
  using rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0));
  rsrc_t dx0 = desc(p.A, p.lda, m0, M), dx1 = desc(p.A, p.lda, m0 + 64, M);
  rsrc_t dw0 = desc(p.B, p.ldb, n0, N), dw1 = desc(p.B, p.ldb, n0 + 32, N);
  const int nk = (K + BK - 1) / BK;
  const bool tail = (K & 63) != 0;
  // (k offset computed inline: a nested lambda call in the builtin's argument list makes hipcc's host pass drop the
  //  kernel without a diagnostic)
  auto stage_x = [&](rsrc_t d, int T, int slot) {
    const int k0 = (tail && T == nk - 1) ? K - BK : T * BK;    // see gemm_nt_v3.hip
    char* dst = smem + slot * XSLOT + wid * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               (int)xv[i], k0 * 2, 0, 0);
  };
  auto stage_w = [&](rsrc_t d, int T, int nh) {
    const int k0 = (tail && T == nk - 1) ? K - BK : T * BK;
    char* dst = smem + W_BASE + (T & 1) * 2 * WSLOT + nh * WSLOT + wid * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               (int)wv[i], k0 * 2, 0, 0);
  };

  f32x4 acc[4][8];
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
    wo[kk] = W_BASE + (wn * 32 + frow) * 128 + choff;
  }
  bf16x8 xf[4][2], wf[2][2][2];

  // X slot rotation: sa holds X mh0(T), sb X mh1(T), sc is free (X mh1(T-1) was there); after a K-tile (sa,sb,sc) <- (sc,sa,sb)
  int sa = 0, sb = 1, sc = 2;

  // TM 0: T <= nk-3; 1: T = nk-2 (no W left to fetch); 2: last K-tile
  auto tile_body = [&](auto mode_c, int T) {
    constexpr int TM = decltype(mode_c)::value;
    const char* wb = smem + (T & 1) * 2 * WSLOT;               // + wo[] (contains W_BASE) + nh * WSLOT
    const char* xa = smem + sa * XSLOT;
    const char* xb = smem + sb * XSLOT;
    // ---- phase 0
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[0][t][kk] = *reinterpret_cast<const bf16x8*>(wb + wo[kk] + t * 2048);
    CLIPK_SB();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(xa + xo[kk] + j * 2048);
    if (TM <= 1) stage_x(dx0, T + 1, sc);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          // W nh0 reads retired: its slot is refilled next phase
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<0, 0, 1>(acc, wf, xf); else quad<0, 0>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
    // ---- phase 1
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[1][t][kk] = *reinterpret_cast<const bf16x8*>(wb + wo[kk] + WSLOT + t * 2048);
    if (TM == 0) stage_w(dw0, T + 2, 0);
    if (TM == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // X mh1(T) landed (read next phase)
    else if (TM == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<1, 0, 1>(acc, wf, xf); else quad<1, 0>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
    // ---- phase 2
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(xb + xo[kk] + j * 2048);
    if (TM <= 1) stage_x(dx1, T + 1, sa);
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<1, 1, 1>(acc, wf, xf); else quad<1, 1>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
    // ---- phase 3
    if (TM == 0) stage_w(dw1, T + 2, 1);
    if (TM == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // X mh0(T+1) (and W(T+1)) landed
    else if (TM == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<0, 1, 1>(acc, wf, xf); else quad<0, 1>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
    const int t_ = sa; sa = sc; sc = sb; sb = t_;               // (sa, sb, sc) <- (sc, sa, sb)
  };

  // ---- prologue: K-tile 0 whole, W of K-tile 1; one full drain (the other workgroup on this CU covers the latency)
  stage_w(dw0, 0, 0); stage_x(dx0, 0, sa); stage_w(dw1, 0, 1); stage_x(dx1, 0, sb);
  if (nk > 1) { stage_w(dw0, 1, 0); stage_w(dw1, 1, 1); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
  for (int T = 0; T < nk - 2; ++T) tile_body(std::integral_constant<int, 0>{}, T);
  if (nk > 1) tile_body(std::integral_constant<int, 1>{}, nk - 2);
  tile_body(std::integral_constant<int, 2>{}, nk - 1);
  __syncthreads();

  // ---- epilogue (gemm_epilogue.h); the slab reuses the first X slot
  float* eb = reinterpret_cast<float*>(smem) + wid * (16 * 64);
  const int gn = n0 + wn * 64 + (lane & 7) * 8;
  float bv[8];
  epi_load_bias(p.e, gn, bv);
  gemm_epilogue<MODE, 8, true>(p.e, acc, eb, lane, m0 + wm * 128, gn, bv);
}

template <int MODE>
void launch_v4(const Params& p, dim3 grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_v4_kernel<MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_v4_kernel<MODE>), grid, dim3(256), LDS_BYTES, st, p);
}

}  // namespace

// called by clipk_gemm_nt (gemm_nt.hip) after it validated the arguments (K % 32 == 0, K >= 96, 32-bit offsets)
extern "C" int clipk_gemm_nt_v4_launch(const clipk_gemm_args* a, void* stream) {
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
  if (mode == EPI_PLAIN) launch_v4<EPI_PLAIN>(p, grid, st);
  else if (mode == EPI_RES32) launch_v4<EPI_RES32>(p, grid, st);
  else if (mode == EPI_GELU_PRE) launch_v4<EPI_GELU_PRE>(p, grid, st);
  else if (mode == EPI_DGELU) launch_v4<EPI_DGELU>(p, grid, st);
  else launch_v4<EPI_GENERIC>(p, grid, st);
  return clipk_check_launch();
}
