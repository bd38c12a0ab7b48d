// gemm_wgrad_v3.hip — 256(n) x 256(k) output tile, 8 waves, phase-interleaved schedule for the weight gradient
// dW[N,K] (+)= dY[M,N]^T · X[M,K], db[N] (+)= colsum(dY).  Same contract, slab layout and reduce kernel as
// gemm_wgrad.hip, which selects this kernel for large problems (CLIPK_WGRAD_V3 forces the choice).
//
// The schedule is the one of gemm_nt_v3.hip (read its header first) with the roles
//     activation half-tile "X mh"  ->  dY half "nh": the nh-th 64 columns of BOTH n-waves,   [64 m][128 cols]
//     weight half-tile     "W nh"  ->  X  half "kh": the kh-th 32 columns of all four k-waves, [64 m][128 cols]
// and one "K-tile" = 64 token rows m.  Differences that matter:
//   * the contraction index m is the ROW index of both operands in memory, so both MFMA fragments come from
//     ds_read_b64_tr_b16 (transposed LDS read) on the row-major half-tiles; the LDS image is the 256-byte-row,
//     32-byte-segment XOR-swizzled one of gemm_wgrad.hip (conflict-free for those reads, tools/lds_conflicts.py);
//   * a fragment is two 8-byte reads, so phase 0 issues 24 LDS reads; lgkmcnt saturates at 15, and
//     lgkmcnt(15) retires at least the 8 reads of X kh0 issued first (the half-tile refilled one phase later);
//   * LDS-DMA sources are buffer descriptors re-based per 64-row step (scalar work): rows past M fail the range
//     check and load zeros, which is exactly what the contraction needs at the ragged end;
//   * M is split over workgroups so that one launch puts ~one workgroup on every CU; each workgroup runs ONE long
//     main loop (M / splits / 64 = 30..100 steps), so unlike the forward GEMM nothing needs to be persistent;
//   * the bias gradient is MFMAs against an all-ones fragment: the four k-waves of an n-wave hold identical dY
//     fragments, so wave wk sums n-tile wk of each half (2 MFMAs in phase 1, 2 in phase 3: every wave of the
//     workgroup gets the same 18 instead of 16 MFMAs in those phases), and the ntk workgroups that share a dY panel
//     take turns by step (T mod ntk == tk), so no workgroup of the launch is slower than the others.  Partial sums
//     per (split, tk) go to the bias slab [splits * ntk][N].
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct clipk_wgrad_v3_args {
  const unsigned short* dY; long lddy;
  const unsigned short* X; long ldx;
  float* slab;        // [splits][N][K]
  float* bslab;       // [splits * ntk][N] or null
  int M, N, K;
  int ntn, ntk, splits, m_per_split;
};

namespace {

constexpr int BN = 256, BKO = 256, BMS = 64;
constexpr int ROWB = 256;                               // LDS row bytes of a half-tile (128 bf16)
constexpr int HALF_BYTES = BMS * ROWB;                  // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;               // dY nh0 | dY nh1 | X kh0 | X kh1
constexpr int LDS_BYTES = 2 * BUF_BYTES;                // 128 KiB
constexpr int YH0 = 0, YH1 = HALF_BYTES, XH0 = 2 * HALF_BYTES, XH1 = 3 * HALF_BYTES;

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// Transposed reads are inline asm on purpose: for the ds_read_tr builtin hipcc (ROCm 7.2) cannot tell that the read
// does not alias an LDS-DMA still in flight and puts s_waitcnt vmcnt(0) in front of every batch, which drains the
// three half-tiles this schedule keeps in flight.  The counters are therefore managed by hand below: every batch is
// followed (after the phase barrier) by s_waitcnt lgkmcnt(0) + sched_barrier before its first use.
template <int OFF>
__device__ __forceinline__ u32x2 ds_tr16(unsigned lds_addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF));
  return v;
}
// 8 m-values of one column: rows r and r + 16 of the transposed-read block (address already swizzled for this lane)
template <int OFF>
__device__ __forceinline__ bf16x8 tr_pair(unsigned lds_addr) {
  const u32x2 lo = ds_tr16<OFF>(lds_addr);
  const u32x2 hi = ds_tr16<OFF + 16 * ROWB>(lds_addr);
  const u32x4 f = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, f);
}

// one quadrant: 4 n-tiles x 2 k-tiles x 2 m-halves = 16 MFMAs (m-half outer: dependent accumulations sit 8 apart)
template <int KH, int NH>
__device__ __forceinline__ void quad(f32x4 (&acc)[8][4], const bf16x8 (&yf)[4][2], const bf16x8 (&xf)[2][2][2]) {
#pragma unroll
  for (int ms = 0; ms < 2; ++ms)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[NH * 4 + t][KH * 2 + u] =
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[t][ms], xf[KH][u][ms], acc[NH * 4 + t][KH * 2 + u], 0, 0, 0);
}

#define CLIPK_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef CLIPK_WGRAD_TRACE
// experiment builds (tools/exp_wgrad_trace.py): cycle stamps after every barrier of steps 8..11, workgroup 0,
// waves 0 (n-wave group 0) and 4 (group 1) -> where a K-step's time goes
__device__ unsigned long long* g_wgrad_trace = nullptr;
#define CLIPK_BAR()                                                                                       \
  do {                                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                         \
    if (tr_on && tr_T >= 8 && tr_T < 12 && tr_i < 16) tr[(tr_T - 8) * 16 + tr_i] = __builtin_readcyclecounter(); \
    ++tr_i;                                                                                               \
  } while (0)
#define CLIPK_BAR2() __builtin_amdgcn_s_barrier()
#define CLIPK_STAMP()                                                                                     \
  do {                                                                                                    \
    if (tr_on && tr_T >= 8 && tr_T < 12 && tr_i < 16) tr[(tr_T - 8) * 16 + tr_i] = __builtin_readcyclecounter(); \
    ++tr_i;                                                                                               \
  } while (0)
#else
#define CLIPK_BAR() __builtin_amdgcn_s_barrier()
#define CLIPK_BAR2() __builtin_amdgcn_s_barrier()
#define CLIPK_STAMP() do { } while (0)
#endif

// SCHED 1: the 8-phase schedule (two wave groups one barrier apart, reads issued right before the barrier of the
// phase that consumes them).  SCHED 2: software-pipelined schedule, see the main loop below.
template <int SCHED>
__global__ __launch_bounds__(512, 1) void wgrad_v3_kernel(const clipk_wgrad_v3_args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wid >> 2, wk = wid & 3;
  const int ntiles = p.ntn * p.ntk;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid / ntiles;
  const int tile = bid - split * ntiles;
  const int tn = tile / p.ntk, tk = tile - tn * p.ntk;
  const int n0 = tn * BN, k0 = tk * BKO;
  const int M = p.M, N = p.N, K = p.K;
  const int m_beg = split * p.m_per_split;
  int m_end = m_beg + p.m_per_split; m_end = m_end < M ? m_end : M;
  const int nkt = (m_end - m_beg + BMS - 1) / BMS;             // >= 1
#ifdef CLIPK_WGRAD_TRACE
  const bool tr_on = bid == 0 && (wid == 0 || wid == 4) && lane == 0 && g_wgrad_trace != nullptr;
  unsigned long long* tr = g_wgrad_trace + (wid == 4 ? 64 : 0);
  int tr_T = -1, tr_i = 0;
#endif
  const bool has_bias = p.bslab != nullptr;
  int bias_T = tk;                                             // next step whose dY rows this workgroup sums

  // ---- LDS-DMA assignment: wave w fills pieces 2w, 2w+1 (4 rows x 256 B each) of every half-tile.
  // lane -> (row in piece = lane>>4, physical 16-B slot = lane&15); the slot's 32-B segment (slot>>1) holds the
  // logical segment (slot>>1) ^ (row&7).  Offsets are relative to (row m_t, first column of the half).
  unsigned yv[2], xv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 4 * (2 * wid + i) + (lane >> 4);
    const int slot = lane & 15;
    const int col = (((slot >> 1) ^ (row & 7)) << 4) + (slot & 1) * 8;      // logical column of this 16-B chunk
    yv[i] = (unsigned)(((long)row * p.lddy + (col >> 6) * 128 + (col & 63)) * 2);   // n-wave col>>6
    xv[i] = (unsigned)(((long)row * p.ldx + (col >> 5) * 64 + (col & 31)) * 2);     // k-wave col>>5
  }
  const unsigned short* y_end = p.dY + ((long)(M - 1) * p.lddy + N);       // one past the last valid element
  const unsigned short* x_end = p.X + ((long)(M - 1) * p.ldx + K);
  // half-tile `half` of step T: descriptor based at (row m_beg + 64 T, first column of the half); everything from
  // there to the end of the operand is in range, i.e. rows >= M (and only those) read as zero
  auto stage = [&](bool is_y, int half, int T, int region) {
    const long mt = (long)m_beg + (long)T * BMS;
    const unsigned short* base = is_y ? p.dY + mt * p.lddy + (n0 + half * 64) : p.X + mt * p.ldx + (k0 + half * 32);
    const long left = ((is_y ? y_end : x_end) - base) * 2;
    const int bytes = left > 0 ? (int)(left < 0x7fffffffL ? left : 0x7fffffffL) : 0;
    const auto d = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    char* dst = smem + (T & 1) * BUF_BYTES + region + wid * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               (int)(is_y ? yv[i] : xv[i]), 0, 0, 0);
  };
  // one of the wave's two pieces of a half-tile (the software-pipelined schedule spreads them: one per MFMA block)
  auto stage1 = [&](bool is_y, int half, int T, int region, int i) {
    const long mt = (long)m_beg + (long)T * BMS;
    const unsigned short* base = is_y ? p.dY + mt * p.lddy + (n0 + half * 64) : p.X + mt * p.ldx + (k0 + half * 32);
    const long left = ((is_y ? y_end : x_end) - base) * 2;
    const int bytes = left > 0 ? (int)(left < 0x7fffffffL ? left : 0x7fffffffL) : 0;
    const auto d = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    char* dst = smem + (T & 1) * BUF_BYTES + region + wid * 2048 + i * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)dst, 16,
                                             (int)(is_y ? (i ? yv[1] : yv[0]) : (i ? xv[1] : xv[0])), 0, 0, 0);
  };

  f32x4 acc[8][4];          // [n-tile i][k-tile j]: rows n = 4g+r, col k = lane&15
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb[2][2];                                            // [n half][m half of the step]
#pragma unroll
  for (int u = 0; u < 2; ++u) { accb[u][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[u][1] = accb[u][0]; }
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (short)0x3F80;          // bf16 1.0

  // ---- transposed-read addressing.  Lane group g = lane>>4, in-group li = lane&15: the lane supplies row
  // 4g + (li>>2) (+16, +32, +48) at byte 8 (li&3) of a 16-column segment.  Segment s of the wave sits at physical
  // segment s ^ (row & 7), row & 7 = 4 (g&1) + (li>>2): an XOR of the segment's low bits with a lane constant, so
  // one address register per segment of the quadrant and immediates for everything else.
  const int g = lane >> 4, li = lane & 15;
  const int trow = 4 * g + (li >> 2);
  const int sw = trow & 7;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of smem
  unsigned ya[4], xa[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) ya[t] = lds0 + trow * ROWB + (((wn * 4 + t) ^ sw) << 5) + 8 * (li & 3);
#pragma unroll
  for (int u = 0; u < 2; ++u) xa[u] = lds0 + trow * ROWB + (((wk * 2 + u) ^ sw) << 5) + 8 * (li & 3);

  if constexpr (SCHED == 1) {
    bf16x8 yf[4][2], xf[2][2][2];

    // dY column sums: accb[half][ms] += yf[wk][ms] x ones.  The tile a wave sums depends on wk; as C++ branches the
    // accumulators become phi copies around each branch, and hipcc then waits for the matrix pipe to copy them
    // (s_nop 7 + v_mov, ~200 cycles per phase).  So the four-way choice is ONE asm statement with the accumulators in
    // place: two independent MFMAs per wave.  s_nop 1 first: `ones` may have been written by a VALU move just before.
    // Nothing but these MFMAs touches accb until the epilogue, hundreds of instructions after the last one.
    auto bias_tile = [&](f32x4& a0, f32x4& a1) {
      asm volatile(
          "s_nop 1\n\t"
          "s_cmp_lg_u32 %[wk], 0\n\t"
          "s_cbranch_scc1 1f\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a0], %[y00], %[on], %[a0]\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a1], %[y01], %[on], %[a1]\n\t"
          "s_branch 4f\n"
          "1:\n\t"
          "s_cmp_lg_u32 %[wk], 1\n\t"
          "s_cbranch_scc1 2f\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a0], %[y10], %[on], %[a0]\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a1], %[y11], %[on], %[a1]\n\t"
          "s_branch 4f\n"
          "2:\n\t"
          "s_cmp_lg_u32 %[wk], 2\n\t"
          "s_cbranch_scc1 3f\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a0], %[y20], %[on], %[a0]\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a1], %[y21], %[on], %[a1]\n\t"
          "s_branch 4f\n"
          "3:\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a0], %[y30], %[on], %[a0]\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a1], %[y31], %[on], %[a1]\n"
          "4:\n\t"
          : [a0] "+v"(a0), [a1] "+v"(a1)
          : [wk] "s"(wk), [on] "v"(ones), [y00] "v"(yf[0][0]), [y01] "v"(yf[0][1]), [y10] "v"(yf[1][0]),
            [y11] "v"(yf[1][1]), [y20] "v"(yf[2][0]), [y21] "v"(yf[2][1]), [y30] "v"(yf[3][0]), [y31] "v"(yf[3][1])
          : "scc");
    };

    // TM 0: steady state; 1: step nkt-2 (only the last half-tile of step nkt-1 left to fetch); 2: last step
    auto step_body = [&](auto mode_c, int T) {
      constexpr int TM = decltype(mode_c)::value;
      const unsigned bo = (T & 1) * BUF_BYTES;
  #ifdef CLIPK_WGRAD_TRACE
      tr_T = T; tr_i = 0;
  #endif
      // ---- phase 0: quadrant (k0, n0); fetch X kh0 (8 reads, first) + dY nh0 (16 reads); refill dY nh1 of step T+1
  #pragma unroll
      for (int u = 0; u < 2; ++u)
  #pragma unroll
        for (int ms = 0; ms < 2; ++ms)
          xf[0][u][ms] = ms ? tr_pair<XH0 + 32 * ROWB>(xa[u] + bo) : tr_pair<XH0>(xa[u] + bo);
      CLIPK_SB();
  #pragma unroll
      for (int t = 0; t < 4; ++t)
  #pragma unroll
        for (int ms = 0; ms < 2; ++ms)
          yf[t][ms] = ms ? tr_pair<YH0 + 32 * ROWB>(ya[t] + bo) : tr_pair<YH0>(ya[t] + bo);
      if (TM <= 1) stage(true, 1, T + 1, YH1);
      asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");         // >= 9 of 24 reads retired: all of X kh0
      CLIPK_SB(); CLIPK_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      CLIPK_SB();
      __builtin_amdgcn_s_setprio(1);
      quad<0, 0>(acc, yf, xf);
      __builtin_amdgcn_s_setprio(0);
      CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
      // ---- phase 1: quadrant (k1, n0); fetch X kh1; refill X kh0 of step T+2
  #pragma unroll
      for (int u = 0; u < 2; ++u)
  #pragma unroll
        for (int ms = 0; ms < 2; ++ms)
          xf[1][u][ms] = ms ? tr_pair<XH1 + 32 * ROWB>(xa[u] + bo) : tr_pair<XH1>(xa[u] + bo);
      if (TM == 0) stage(false, 0, T + 2, XH0);
      CLIPK_SB(); CLIPK_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      CLIPK_SB();
      __builtin_amdgcn_s_setprio(1);
      quad<1, 0>(acc, yf, xf);
      const bool bias_now = has_bias && T == bias_T;              // (workgroup-uniform)
      if (bias_now) bias_tile(accb[0][0], accb[0][1]);            // n-tiles 0..3 are live: wave wk sums tile wk
      __builtin_amdgcn_s_setprio(0);
      CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
      // ---- phase 2: quadrant (k1, n1); fetch dY nh1; refill dY nh0 of step T+2
  #pragma unroll
      for (int t = 0; t < 4; ++t)
  #pragma unroll
        for (int ms = 0; ms < 2; ++ms)
          yf[t][ms] = ms ? tr_pair<YH1 + 32 * ROWB>(ya[t] + bo) : tr_pair<YH1>(ya[t] + bo);
      if (TM == 0) stage(true, 0, T + 2, YH0);
      CLIPK_SB(); CLIPK_BAR();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      CLIPK_SB();
      __builtin_amdgcn_s_setprio(1);
      quad<1, 1>(acc, yf, xf);
      __builtin_amdgcn_s_setprio(0);
      CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
      // ---- phase 3: quadrant (k0, n1); nothing to fetch; refill X kh1 of step T+2; step T+1 must be complete
      if (TM == 0) {
        stage(false, 1, T + 2, XH1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else if (TM == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
      __builtin_amdgcn_s_setprio(1);
      quad<0, 1>(acc, yf, xf);
      if (bias_now) { bias_tile(accb[1][0], accb[1][1]); bias_T += p.ntk; }   // n-tiles 4..7 are live
      __builtin_amdgcn_s_setprio(0);
      CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    };

    // ---- prologue: step 0 complete, three half-tiles of step 1 in flight
    stage(false, 0, 0, XH0); stage(true, 0, 0, YH0); stage(false, 1, 0, XH1); stage(true, 1, 0, YH1);
    if (nkt > 1) {
      stage(false, 0, 1, XH0); stage(true, 0, 1, YH0); stage(false, 1, 1, XH1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    if (wn == 1) CLIPK_BAR();                                     // n = 1 waves run one barrier behind
  #ifdef CLIPK_WGRAD_TRACE
    // in-kernel clock (MI355X_MICROARCH.md, DVFS item 6): shader cycles and 100 MHz ticks around the main loop
    unsigned long long tc0 = 0, tr0 = 0;
    if (tr_on && wid == 0) { tc0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
    __builtin_amdgcn_s_waitcnt(0xC07F);
  #endif
    for (int T = 0; T < nkt - 2; ++T) step_body(std::integral_constant<int, 0>{}, T);
    if (nkt > 1) step_body(std::integral_constant<int, 1>{}, nkt - 2);
    step_body(std::integral_constant<int, 2>{}, nkt - 1);
    if (wn == 0) CLIPK_BAR();                                     // re-align
  #ifdef CLIPK_WGRAD_TRACE
    if (tr_on && wid == 0) {
      g_wgrad_trace[128] = __builtin_amdgcn_s_memtime() - tc0;
      g_wgrad_trace[129] = __builtin_amdgcn_s_memrealtime() - tr0;
      g_wgrad_trace[130] = (unsigned long long)nkt;
    }
  #endif
  } else {
    // ================================================================================================
    // SCHED 2.  Stamps of the 8-phase loop (tools/exp_wgrad_trace.py) show every MFMA block preceded by the full
    // latency of the LDS reads issued just before its barrier: 256 cycles of MFMAs per ~450-cycle window, the matrix
    // pipe 57 % busy - the second wave group only issues DMA in that window, it does not compute.  Here every wave
    // issues the reads of a LATER block before the MFMAs of the current one, without more registers: a K-step is 8
    // blocks of 8 MFMAs (k half x n quarter), the dY fragments are two QUARTER buffers of 16 registers (a quarter is
    // read while the previous one is multiplied), both X halves stay resident, and the block order
    //     (k0,q0) (k1,q0) (k1,q1) (k0,q1) (k0,q2) (k1,q2) (k0,q3) (k1,q3)
    // ends on k1 after k0, so X k0 of the next step is read during the last block and X k1 during the next first one.
    //   block : reads issued before its MFMAs          barrier
    //     0   : dY q1, X k1
    //     2   : dY q2                                  alpha (dY nh0 and X of step T all read)
    //     4   : dY q3
    //     6   : dY q0 of T+1                           beta (dY nh1 of T all read; dY nh0, X of T+1 landed)
    //     7   : X k0 of T+1
    // Two barriers per step instead of eight; each is preceded by the counted vmcnt that makes the half-tiles read
    // after it complete (alpha: dY nh1 of T; beta: dY nh0 / X kh0 / X kh1 of T+1).  The LDS-DMA pieces: see step2.
    bf16x8 yq[2][2][2], xf[2][2][2];              // yq[buffer][tile in quarter][m half], xf[k half][u][m half]
    auto read_yq = [&](auto q_c, unsigned bo) {   // quarter q (tiles 2 (q & 1) + {0, 1} of half q >> 1) -> yq[q & 1]
      constexpr int q = decltype(q_c)::value;
      constexpr int REG = (q >> 1) ? YH1 : YH0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        yq[q & 1][t][0] = tr_pair<REG>(ya[(q & 1) * 2 + t] + bo);
        yq[q & 1][t][1] = tr_pair<REG + 32 * ROWB>(ya[(q & 1) * 2 + t] + bo);
      }
    };
    auto read_x = [&](auto kh_c, unsigned bo) {
      constexpr int kh = decltype(kh_c)::value;
      constexpr int REG = kh ? XH1 : XH0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xf[kh][u][0] = tr_pair<REG>(xa[u] + bo);
        xf[kh][u][1] = tr_pair<REG + 32 * ROWB>(xa[u] + bo);
      }
    };
    auto blk = [&](auto kh_c, auto q_c, auto&& mid) {   // 8 MFMAs: 2 n-tiles x 2 k-tiles x 2 m-halves; `mid` between the halves
      constexpr int kh = decltype(kh_c)::value, q = decltype(q_c)::value;
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[q * 2 + t][kh * 2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yq[q & 1][t][ms], xf[kh][u][ms],
                                                                                 acc[q * 2 + t][kh * 2 + u], 0, 0, 0);
        if (ms == 0) { CLIPK_SB(); mid(); CLIPK_SB(); }
      }
    };
    // dY column sums: wave wk sums n-tile wk of each half = tile wk & 1 of quarter 2 half + (wk >> 1): waves 0, 1 take
    // part in quarters 0 and 2, waves 2, 3 in quarters 1 and 3.  One asm statement, accumulators in place (see SCHED 1).
    const int b_odd = wk >> 1, b_sel = wk & 1;
    auto bias_mfma = [](int odd, int sel, int qo, f32x4& a0, f32x4& a1, const bf16x8& on, const bf16x8& y00,
                        const bf16x8& y01, const bf16x8& y10, const bf16x8& y11) {
      asm volatile(
          "s_nop 1\n\t"
          "s_cmp_lg_u32 %[odd], %[qo]\n\t"
          "s_cbranch_scc1 2f\n\t"
          "s_cmp_lg_u32 %[sel], 0\n\t"
          "s_cbranch_scc1 1f\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a0], %[y00], %[on], %[a0]\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a1], %[y01], %[on], %[a1]\n\t"
          "s_branch 2f\n"
          "1:\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a0], %[y10], %[on], %[a0]\n\t"
          "v_mfma_f32_16x16x32_bf16 %[a1], %[y11], %[on], %[a1]\n"
          "2:\n\t"
          : [a0] "+v"(a0), [a1] "+v"(a1)
          : [odd] "s"(odd), [sel] "s"(sel), [qo] "s"(qo), [on] "v"(on), [y00] "v"(y00), [y01] "v"(y01),
            [y10] "v"(y10), [y11] "v"(y11)
          : "scc");
    };
#define CLIPK_BIAS_Q(q, a0, a1) \
  bias_mfma(b_odd, b_sel, (q) & 1, a0, a1, ones, yq[(q) & 1][0][0], yq[(q) & 1][0][1], yq[(q) & 1][1][0], yq[(q) & 1][1][1])
#if defined(CLIPK_WGRAD_TRACE) && defined(CLIPK_WGRAD_ABL)
    // timing ablations of the experiment build (results are garbage): 1 no LDS-DMA in the loop, 2 no LDS reads in the
    // loop, 4 no barriers in the loop
#define CLIPK_ABL_DMA(x) do { if (!(CLIPK_WGRAD_ABL & 1)) { x; } } while (0)
#define CLIPK_ABL_RD(x) do { if (!(CLIPK_WGRAD_ABL & 2)) { x; } } while (0)
#define CLIPK_ABL_BAR(x) do { if (!(CLIPK_WGRAD_ABL & 4)) { x; } } while (0)
#else
#define CLIPK_ABL_DMA(x) do { x; } while (0)
#define CLIPK_ABL_RD(x) do { x; } while (0)
#define CLIPK_ABL_BAR(x) do { x; } while (0)
#endif
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
#define CLIPK_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    // TM 0: steady state (refills for step T+2); 1: step nkt-2 (only dY nh1 of the last step left); 2: last step.
    // LDS-DMA: ONE piece per block, between the block's two MFMA halves (a burst of pieces from all eight waves queues
    // up in the CU's vector-memory path and every issuing wave waits for it): per wave and step, in issue order,
    //     blocks 0, 1: dY nh1 of T+1 (slot free since beta of T-1) | blocks 2..5: dY nh0, X kh0 of T+2 (free since alpha)
    //     blocks 6, 7: X kh1 of T+2
    // so before beta "all but the newest 6" covers dY nh0 / X of T+1, before alpha "all but the newest 8" dY nh1 of T.
    auto step2 = [&](auto mode_c, int T) {
      constexpr int TM = decltype(mode_c)::value;
      const unsigned bo = (T & 1) * BUF_BYTES, bn = ((T + 1) & 1) * BUF_BYTES;
      const bool bias_now = has_bias && T == bias_T;              // (workgroup-uniform)
      const bool y1_next = TM != 2 && T >= 1;                     // (step 1's dY nh1 comes with the prologue)
      auto none = [] {};
#ifdef CLIPK_WGRAD_TRACE
      tr_T = T; tr_i = 0;
#endif
      // ---- block 0 (k0, q0): needs dY q0 + X k0 (read during blocks 6 / 7 of the previous step)
      CLIPK_LGKM0(); CLIPK_SB(); CLIPK_STAMP();
      CLIPK_ABL_RD(read_yq(I1{}, bo); read_x(I1{}, bo));
      CLIPK_SB();
      blk(I0{}, I0{}, [&] { if (y1_next) CLIPK_ABL_DMA(stage1(true, 1, T + 1, YH1, 0)); });
      if (bias_now) CLIPK_BIAS_Q(0, accb[0][0], accb[0][1]);
      CLIPK_SB();
      // ---- block 1 (k1, q0)
      CLIPK_LGKM0(); CLIPK_SB(); CLIPK_STAMP();
      blk(I1{}, I0{}, [&] { if (y1_next) CLIPK_ABL_DMA(stage1(true, 1, T + 1, YH1, 1)); });
      CLIPK_SB();
      // ---- block 2 (k1, q1): barrier alpha
      if (TM == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      CLIPK_SB(); CLIPK_ABL_BAR(CLIPK_BAR2()); CLIPK_SB(); CLIPK_STAMP();
      CLIPK_ABL_RD(read_yq(I2{}, bo));
      CLIPK_SB();
      blk(I1{}, I1{}, [&] { if (TM == 0) CLIPK_ABL_DMA(stage1(true, 0, T + 2, YH0, 0)); });
      if (bias_now) CLIPK_BIAS_Q(1, accb[0][0], accb[0][1]);
      CLIPK_SB();
      // ---- block 3 (k0, q1)
      CLIPK_STAMP();
      blk(I0{}, I1{}, [&] { if (TM == 0) CLIPK_ABL_DMA(stage1(true, 0, T + 2, YH0, 1)); });
      CLIPK_SB();
      // ---- block 4 (k0, q2)
      CLIPK_LGKM0(); CLIPK_SB(); CLIPK_STAMP();
      CLIPK_ABL_RD(read_yq(I3{}, bo));
      CLIPK_SB();
      blk(I0{}, I2{}, [&] { if (TM == 0) CLIPK_ABL_DMA(stage1(false, 0, T + 2, XH0, 0)); });
      if (bias_now) CLIPK_BIAS_Q(2, accb[1][0], accb[1][1]);
      CLIPK_SB();
      // ---- block 5 (k1, q2)
      CLIPK_STAMP();
      blk(I1{}, I2{}, [&] { if (TM == 0) CLIPK_ABL_DMA(stage1(false, 0, T + 2, XH0, 1)); });
      CLIPK_SB();
      // ---- block 6 (k0, q3): barrier beta
      CLIPK_LGKM0();
      if (TM == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (TM == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      CLIPK_SB(); CLIPK_ABL_BAR(CLIPK_BAR2()); CLIPK_SB(); CLIPK_STAMP();
      if (TM != 2) CLIPK_ABL_RD(read_yq(I0{}, bn));
      CLIPK_SB();
      blk(I0{}, I3{}, [&] { if (TM == 0) CLIPK_ABL_DMA(stage1(false, 1, T + 2, XH1, 0)); });
      if (bias_now) CLIPK_BIAS_Q(3, accb[1][0], accb[1][1]);
      CLIPK_SB();
      // ---- block 7 (k1, q3)
      CLIPK_STAMP();
      if (TM != 2) CLIPK_ABL_RD(read_x(I0{}, bn));
      CLIPK_SB();
      blk(I1{}, I3{}, [&] { if (TM == 0) CLIPK_ABL_DMA(stage1(false, 1, T + 2, XH1, 1)); });
      if (bias_now) bias_T += p.ntk;
      CLIPK_SB();
      (void)none;
    };

    // ---- prologue: steps 0 and 1 requested in the steady-state piece order, step 0 complete, its first reads issued
    stage(true, 0, 0, YH0); stage(false, 0, 0, XH0); stage(false, 1, 0, XH1); stage(true, 1, 0, YH1);
    if (nkt > 1) {
      stage(true, 0, 1, YH0); stage(false, 0, 1, XH0); stage(false, 1, 1, XH1); stage(true, 1, 1, YH1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CLIPK_SB(); CLIPK_BAR2(); CLIPK_SB();
    read_yq(I0{}, 0u); read_x(I0{}, 0u);
#ifdef CLIPK_WGRAD_TRACE
    unsigned long long tc0 = 0, tr0 = 0;
    if (tr_on && wid == 0) { tc0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
    __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
    for (int T = 0; T < nkt - 2; ++T) step2(std::integral_constant<int, 0>{}, T);
    if (nkt > 1) step2(std::integral_constant<int, 1>{}, nkt - 2);
    step2(std::integral_constant<int, 2>{}, nkt - 1);
#ifdef CLIPK_WGRAD_TRACE
    if (tr_on && wid == 0) {
      g_wgrad_trace[128] = __builtin_amdgcn_s_memtime() - tc0;
      g_wgrad_trace[129] = __builtin_amdgcn_s_memrealtime() - tr0;
      g_wgrad_trace[130] = (unsigned long long)nkt;
    }
#endif
  }

  // ---- store the f32 partial tile: rows n (4 per lane), cols k (lane&15)
  float* slab = p.slab + (long)split * N * K;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wn * 128 + i * 16 + 4 * g + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + li;
        if (n < N && k < K) slab[(long)n * K + k] = acc[i][j][r];
      }
    }
  if (has_bias && li == 0) {
    // accb[u]: n-tile wk of half u of this n-wave, summed over this workgroup's turns (possibly none: zeros)
    float* bs = p.bslab + ((long)split * p.ntk + tk) * N;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 128 + u * 64 + wk * 16 + 4 * g + r;
        if (n < N) bs[n] = accb[u][0][r] + accb[u][1][r];
      }
  }
}

}  // namespace

// plan: one workgroup per (tile, split), about one per CU; splits are multiples of 64 rows
extern "C" void clipk_wgrad_v3_plan(int M, int N, int K, int* ntn, int* ntk, int* splits, int* mps) {
  *ntn = (N + BN - 1) / BN; *ntk = (K + BKO - 1) / BKO;
  const int ntiles = *ntn * *ntk;
  int s = 256 / ntiles;
  if (clipk_opt_get(OPT_WGRAD_SPLITS) > 0) s = clipk_opt_get(OPT_WGRAD_SPLITS);
  if (s < 1) s = 1;
  const int max_splits = (M + 8 * BMS - 1) / (8 * BMS);         // at least 512 rows per split
  if (s > max_splits) s = max_splits;
  int m = (M + s - 1) / s;
  m = (m + BMS - 1) / BMS * BMS;
  *mps = m;
  *splits = (M + m - 1) / m;
}

#ifdef CLIPK_WGRAD_TRACE
extern "C" int clipk_wgrad_v3_set_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int clipk_wgrad_v3_launch(const clipk_wgrad_v3_args* a, void* stream) {
  static std::atomic<uint64_t> attr_set{0};
  clipk_once_per_device(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_v3_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_v3_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  const dim3 grid(a->ntn * a->ntk * a->splits);
  // option wgrad_kernel = 3: the 8-phase schedule; 4 (and auto): the software-pipelined one.  In cycles the second is
  // 8 % ahead (3326 vs 3602 per step, matrix pipe 0.62 vs 0.57 busy), in wall time 1-2 %: the chip answers the
  // denser MFMA stream with a lower clock (1.82 vs 1.95 GHz in-kernel)
  if (clipk_opt_get(OPT_WGRAD_KERNEL) == 3)
    hipLaunchKernelGGL(wgrad_v3_kernel<1>, grid, dim3(512), LDS_BYTES, (hipStream_t)stream, *a);
  else
    hipLaunchKernelGGL(wgrad_v3_kernel<2>, grid, dim3(512), LDS_BYTES, (hipStream_t)stream, *a);
  return clipk_check_launch();
}
