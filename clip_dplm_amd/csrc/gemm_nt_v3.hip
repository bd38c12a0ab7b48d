// gemm_nt_v3.hip — persistent 256 x 256-tile, 8-wave, phase-interleaved kernel for clipk_gemm_nt
// (K % 32 == 0, K >= 192).  Same contract and epilogue as gemm_nt_v2.hip; selected by gemm_nt.hip (option gemm_kernel).
//
// Why a second structure: the 128 x 128 kernel tops out where the CU's L2 -> LDS path saturates.
// A 256 x 256 tile halves the operand bytes per FLOP, but only pays with ~1 workgroup per CU if the loads stay in
// flight across barriers and the two waves of each SIMD alternate between "fetch fragments" and "issue MFMAs"
// (cdna_hip_programming.md §5, 8-phase template).  Main loop:
//   * 8 waves = 2 (m) x 4 (n), wave tile 128 m x 64 n, 128 accumulator VGPRs; one K-tile (BK = 64) = 4 phases of
//     16 MFMAs, each phase one quadrant (64 m x 32 n) of the wave tile: (m0,n0) (m0,n1) (m1,n1) (m1,n0);
//   * LDS = 2 buffers x 4 half-tiles of 16 KiB.  A half-tile is defined by CONSUMPTION order, not by position:
//     "X mh" holds the mh-th 64 rows of BOTH m-waves, "W nh" the nh-th 32 rows of all four n-waves, so a half-tile
//     is dead after the phase that read it and can be refilled while the rest of the buffer is still in use;
//   * every phase refills one half-tile (2 x global_load_lds_dwordx4 per lane) two K-tiles ahead; the only vmcnt
//     waits are counted, once per K-tile (three half-tiles stay in flight) — never 0 in the main loop;
//   * raw s_barrier twice per phase; the m = 1 waves run one barrier behind the m = 0 waves, so on every SIMD one
//     wave is in its MFMA block (s_setprio 1) while the other fetches fragments and issues the refill.
// Hazard bookkeeping (phases numbered 4T + ph for K-tile T):
//   RAW  tile T+1 is complete at the counted vmcnt of phase 4T+3, both wave groups have executed that wait before
//        the barrier that opens phase 4T+4, where it is first read;
//   WAR  W nh0: read first in phase 4T (retired by lgkmcnt(8) before that phase's barrier), refilled in 4T+1;
//        X mh0: read 4T, refilled 4T+2;  W nh1: read 4T+1, refilled 4T+3;  X mh1: read 4T+2, refilled 4T+4.
//
// Persistent over output tiles, because with one workgroup per CU nothing else hides a tile's fixed costs: measured
// (profiles/r02, K = 480) the first-tile fetch (~4 us) and the output stream (128 KiB per tile at the ~10 B/clk a
// CU gets from HBM, ~5.5 us) were purely additive to a 12 us main loop, and a wave cannot retire (s_endpgm) before
// its stores are acknowledged.  So each workgroup walks tiles bid, bid + grid, ... and
//   * issues the LDS-DMA of the NEXT tile's first two K-tiles (16 loads per lane) right after its last MFMA phase,
//     before the epilogue: the fetch latency hides under the epilogue;
//   * the epilogue (gemm_epilogue.h, slab in the 32 KiB of LDS beside the two buffers) only ISSUES its stores; they
//     drain under the next tile's main loop.  vmcnt is an in-order counter shared by loads and stores, so the waits
//     of the next tile's first K-tile allow for the NS store instructions issued after the prefetch:
//     vmcnt(8 + NS) / vmcnt(6 + NS) instead of vmcnt(8) / vmcnt(6).
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;        // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;       // X mh0 | X mh1 | W nh0 | W nh1
constexpr int SLAB_BYTES = 16 * 64 * 4;         // per-wave epilogue slab (XOR-swizzled, unpadded)
constexpr int LDS_BYTES = 2 * BUF_BYTES + 8 * SLAB_BYTES;   // 160 KiB: the whole CU

struct Params {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  int M, N, K;
  EpiArgs e;
  int ntn, ntiles;
  int abl;          // timing-only ablation (tools/bench_kernels.py): 1 = no epilogue
  int stagger;      // workgroup b starts ((b >> 3) & 3) * stagger * ~3.4 us late, 0 = off
};

// one quadrant: 2 n-tiles x 4 m-tiles x 2 k-halves = 16 MFMAs (k outer so dependent accumulations sit 8 apart);
// KLO = 1: upper k-half only, for the last K-tile of a K % 64 == 32 problem (fetched as [K - 64, K), whose lower
// half was already accumulated by the K-tile before)
template <int NH, int MH, int KLO = 0>
__device__ __forceinline__ void quad(f32x4 (&acc)[4][8], const bf16x8 (&wf)[2][2][2], const bf16x8 (&xf)[4][2]) {
#ifdef CLIPK_GEMM_MFMA32
  // TIMING-ONLY experiment (tools/exp_gemm_mfma_shape.py): the same quadrant as 8 x
  // v_mfma_f32_32x32x16_bf16 (2 m-tiles of 32 x 1 n-tile of 32 x 4 k-steps of 16; 8 x 32 = 256 pipe cycles, as 16 x
  // 16) on the SAME 12 fragment registers and the same 32 accumulator registers.  The fragments were fetched in the
  // 16x16x32 lane layout, so the products are garbage (random bf16 data all the same: realistic operand toggling);
  // instruction mix, LDS traffic, register footprint and MFMA pipe cycles are those of a real 32x32x16 loop.
  typedef __attribute__((ext_vector_type(16))) float f32x16;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    f32x16 c;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = acc[NH * 2 + (e >> 3)][MH * 4 + mt * 2 + ((e >> 2) & 1)][e & 3];
#pragma unroll
    for (int ks = 2 * KLO; ks < 4; ++ks)
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[NH][ks & 1][ks >> 1], xf[mt * 2 + (ks & 1)][ks >> 1], c, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[NH * 2 + (e >> 3)][MH * 4 + mt * 2 + ((e >> 2) & 1)][e & 3] = c[e];
  }
  return;
#endif
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
#ifdef CLIPK_GEMM_TRACE
// experiment builds (tools/exp_gemm_mfma_shape.py): shader cycles (s_memtime) and 100 MHz ticks (s_memrealtime) summed
// over the main loops of workgroup 0, and its K-tile count -> cycles per K-tile and the in-kernel clock
// (MI355X_MICROARCH.md, DVFS item 6).  The stamps go to a buffer of their own; no output depends on them.
__device__ unsigned long long* g_gemm_trace = nullptr;
#endif
#define CLIPK_STR2(x) #x
#define CLIPK_STR(x) CLIPK_STR2(x)
// counted wait that tolerates NS younger-than-the-loads store instructions (NS is a template constant 0..32)
#define CLIPK_VMCNT_PLUS(base, ns)                                                              \
  do {                                                                                          \
    if ((ns) == 0) asm volatile("s_waitcnt vmcnt(" CLIPK_STR(base) ")" ::: "memory");           \
    else if ((ns) == 16) asm volatile("s_waitcnt vmcnt(" CLIPK_STR(base) "+16)" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(" CLIPK_STR(base) "+32)" ::: "memory");                  \
  } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm_nt_v3_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = epi_stores(MODE, 8) < 0 ? 0 : epi_stores(MODE, 8);
  static_assert(NS == 0 || NS == 16 || NS == 32, "vmcnt bookkeeping below knows 0 / 16 / 32 stores");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int M = p.M, N = p.N, K = p.K;

  // ---- LDS-DMA assignment: wave w fills pieces 2w, 2w+1 (8 rows x 128 B each) of every half-tile.
  // lane -> (row in piece = lane>>3, physical 16-B slot = lane&7); source chunk = slot ^ ((row>>1)&7).
  // Sources are buffer descriptors, one per half-tile kind and re-based per output tile (scalar work only): the
  // per-lane offsets below never change, rows past M / N fail the range check and load zeros, and a refill costs
  // no VALU instruction at all (buffer_load_dwordx4 v_off, s[desc], s_k0 offen lds).
  unsigned xv[2], wv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (2 * wid + i) + (lane >> 3);              // row of the half-tile image
    const int kch = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    xv[i] = (unsigned)(((long)((r >> 6) * 128 + (r & 63)) * p.lda + kch) * 2);   // X half mh: rows of m-wave r>>6
    wv[i] = (unsigned)(((long)((r >> 5) * 64 + (r & 31)) * p.ldb + kch) * 2);    // W half nh: rows of n-wave r>>5
  }
  int m0 = 0, n0 = 0;
  using rsrc_t = decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0));
  rsrc_t dx0, dx1, dw0, dw1;
  auto desc = [&](const unsigned short* base, long ld, int row0, int rows) {      // rows [row0, rows) of a [rows][K] operand
    const long left = (long)rows - row0;
    const int bytes = left > 0 ? (int)(((left - 1) * ld + K) * 2) : 0;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (long)row0 * ld), 0, bytes, 0x00020000);
  };
  auto setup = [&](int bid) {                                   // tile -> origin and the four descriptors
    const int tile = xcd_remap(bid, p.ntiles);
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    m0 = tm * BM; n0 = tn * BN;
    dx0 = desc(p.A, p.lda, m0, M); dx1 = desc(p.A, p.lda, m0 + 64, M);
    dw0 = desc(p.B, p.ldb, n0, N); dw1 = desc(p.B, p.ldb, n0 + 32, N);
  };
  const int nk = (K + BK - 1) / BK;                             // >= 3 (launcher)
  const bool tail = (K & 63) != 0;                              // K % 64 == 32
  // K-tile T covers k in [64 T, 64 T + 64), except the last one of a K % 64 == 32 problem, which is fetched as
  // [K - 64, K): always in range, and only its upper half is multiplied (quad<.., KLO = 1>)
  auto stage = [&](rsrc_t d, const unsigned (&off)[2], int T, int region) {
    const int k0 = (tail && T == nk - 1) ? K - BK : T * BK;
    char* dst = smem + (T & 1) * BUF_BYTES + region + wid * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               (int)off[i], k0 * 2, 0, 0);
  };
  constexpr int XH0 = 0, XH1 = HALF_BYTES, WH0 = 2 * HALF_BYTES, WH1 = 3 * HALF_BYTES;
  auto prefetch_two = [&]() {                                   // K-tiles 0 and 1 of the tile `setup` pointed at
    stage(dw0, wv, 0, WH0); stage(dx0, xv, 0, XH0); stage(dw1, wv, 0, WH1); stage(dx1, xv, 0, XH1);
    stage(dw0, wv, 1, WH0); stage(dx0, xv, 1, XH0); stage(dw1, wv, 1, WH1); stage(dx1, xv, 1, XH1);
  };

  const int frow = lane & 15, fch = lane >> 4, lane_sw = (frow >> 1) & 7;
  int xo[2], wo[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int choff = ((kk * 4 + fch) ^ lane_sw) << 4;
    xo[kk] = (wm * 64 + frow) * 128 + choff;
    wo[kk] = WH0 + (wn * 32 + frow) * 128 + choff;
  }

  f32x4 acc[4][8];          // [n-tile i][m-tile j]: rows n = 4g+r, col m = lane&15
  bf16x8 xf[4][2], wf[2][2][2];
  bool first = true;                                            // no epilogue stores of a previous tile in flight

  // TM 0: steady state; 1: K-tile nk-2 (only the last half-tile of K-tile nk-1 left to fetch); 2: last K-tile;
  // 3: K-tile 0 (K-tile 1 was prefetched whole; its wait also covers the previous tile's NS stores)
  auto tile_body = [&](auto mode_c, int T) {
    constexpr int TM = decltype(mode_c)::value;
    const char* buf = smem + (T & 1) * BUF_BYTES;
    // ---- phase 0: quadrant (n0, m0); fetch W nh0 (4 reads, first) + X mh0 (8 reads); refill X mh1 of K-tile T+1
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[0][t][kk] = *reinterpret_cast<const bf16x8*>(buf + wo[kk] + t * 2048);
    CLIPK_SB();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(buf + xo[kk] + XH0 + j * 2048);
    if (TM <= 1) stage(dx1, xv, T + 1, XH1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          // W nh0 reads retired: refilled next phase
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<0, 0, 1>(acc, wf, xf);
    else quad<0, 0>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    // ---- phase 1: quadrant (n1, m0); fetch W nh1; refill W nh0 of K-tile T+2
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        wf[1][t][kk] = *reinterpret_cast<const bf16x8*>(buf + wo[kk] + HALF_BYTES + t * 2048);
    if (TM == 0 || TM == 3) stage(dw0, wv, T + 2, WH0);
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<1, 0, 1>(acc, wf, xf);
    else quad<1, 0>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    // ---- phase 2: quadrant (n1, m1); fetch X mh1; refill X mh0 of K-tile T+2
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(buf + xo[kk] + XH1 + j * 2048);
    if (TM == 0 || TM == 3) stage(dx0, xv, T + 2, XH0);
    CLIPK_SB(); CLIPK_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<1, 1, 1>(acc, wf, xf);
    else quad<1, 1>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    // ---- phase 3: quadrant (n0, m1); nothing to fetch; refill W nh1 of K-tile T+2; K-tile T+1 must be complete
    if (TM == 0) {
      stage(dw1, wv, T + 2, WH1);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (TM == 3) {
      stage(dw1, wv, T + 2, WH1);
      // K-tile 1 came with the prefetch issued BEFORE the previous tile's epilogue: queue = prefetch, NS stores, 6
      if (first) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else CLIPK_VMCNT_PLUS(6, NS);
    } else if (TM == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    __builtin_amdgcn_s_setprio(1);
    if (TM == 2 && tail) quad<0, 1, 1>(acc, wf, xf);
    else quad<0, 1>(acc, wf, xf);
    __builtin_amdgcn_s_setprio(0);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
  };

  if (p.stagger > 0) {
    const int q = (blockIdx.x >> 3) & 3;
    for (int i = 0; i < q * p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  int bid = blockIdx.x;
  setup(bid);
  prefetch_two();
  while (true) {
    const int cm0 = m0, cn0 = n0;
    // K-tile 0 complete: the 8 loads of K-tile 1 (and the previous tile's NS stores) may stay in flight
    if (first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else CLIPK_VMCNT_PLUS(8, NS);
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    if (wm == 1) CLIPK_BAR();                                   // m = 1 waves run one barrier behind
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef CLIPK_GEMM_TRACE
    const bool tr_on = blockIdx.x == 0 && tid == 0 && g_gemm_trace != nullptr;
    unsigned long long tc0 = 0, tr0 = 0;
    if (tr_on) { tc0 = __builtin_amdgcn_s_memtime(); tr0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    tile_body(std::integral_constant<int, 3>{}, 0);
    for (int T = 1; T < nk - 2; ++T) tile_body(std::integral_constant<int, 0>{}, T);
    tile_body(std::integral_constant<int, 1>{}, nk - 2);
    // epilogue index math is recomputed per tile from a fresh lane id: kept live (or hoisted) it would occupy
    // ~20 VGPRs through the main loop, which runs at the 256-VGPR cap
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int gn_e = cn0 + wn * 64 + (lane_e & 7) * 8;
    tile_body(std::integral_constant<int, 2>{}, nk - 1);
#ifdef CLIPK_GEMM_TRACE
    if (tr_on) {
      g_gemm_trace[0] += __builtin_amdgcn_s_memtime() - tc0;
      g_gemm_trace[1] += __builtin_amdgcn_s_memrealtime() - tr0;
      g_gemm_trace[2] += (unsigned long long)nk;
    }
#endif
    if (wm == 0) CLIPK_BAR();                                   // re-align: every fragment read of this tile retired
    CLIPK_SB();
    // bias before the prefetch: vmcnt is in-order, a load issued after the prefetch could not be waited for
    // without waiting for the prefetch too.  (Issuing it before the last K-tile would hide its latency, but the
    // 8 extra live VGPRs make the main loop spill.)
    float bv[8];
    epi_load_bias(p.e, gn_e, bv);
    bid += gridDim.x;
    const bool more = bid < p.ntiles;
    if (more) {
      setup(bid);
      prefetch_two();                                           // lands under the epilogue
    }
    // ---- epilogue (gemm_epilogue.h): issues its stores and moves on; they drain under the next main loop
    float* eb = reinterpret_cast<float*>(smem + 2 * BUF_BYTES) + wid * (SLAB_BYTES / 4);
#ifdef CLIPK_EXPERIMENTS
    if (p.abl & 1) {
      float sacc = bv[0];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sacc == 1.2345e-30f) reinterpret_cast<float*>(p.e.C)[0] = sacc;
    } else
#endif
    gemm_epilogue<MODE, 8, true>(p.e, acc, eb, lane_e, cm0 + wm * 128, gn_e, bv);
    if (!more) break;
#ifdef CLIPK_EXPERIMENTS
    const bool drain = MODE == EPI_GENERIC || (p.abl & 1);
#else
    constexpr bool drain = MODE == EPI_GENERIC;
#endif
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // store count unknown: drain
    first = drain;
  }
}

template <int MODE>
void launch_v3(const Params& p, dim3 grid, hipStream_t st) {
  static std::atomic<uint64_t> attr_set{0};
  clipk_once_per_device(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_v3_kernel<MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  hipLaunchKernelGGL((gemm_nt_v3_kernel<MODE>), grid, dim3(512), LDS_BYTES, st, p);
}

int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace

#ifdef CLIPK_GEMM_TRACE
extern "C" int clipk_gemm_v3_set_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

// called by clipk_gemm_nt (gemm_nt.hip) after it validated the arguments (K % 32 == 0, K >= 192)
extern "C" int clipk_gemm_nt_v3_launch(const clipk_gemm_args* a, void* stream) {
  Params p;
  p.A = (const unsigned short*)a->A; p.lda = a->lda;
  p.B = (const unsigned short*)a->B; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.e = epi_args_from(a);
  const int ntm = (a->M + BM - 1) / BM, ntn = (a->N + BN - 1) / BN;
  p.ntn = ntn;
  p.ntiles = ntm * ntn;
  // one persistent workgroup per CU; a multiple of 8 keeps "workgroup b runs on XCD b % 8" true for every tile
  // it walks, so the XCD-contiguous tile order (xcd_remap) still holds
  int nwg = cu_count() & ~7;
  { const int e = clipk_opt_get(OPT_GEMM_NWG); if (e >= 8) nwg = e & ~7; }                   // experiments only
  if (nwg > p.ntiles) nwg = p.ntiles;
  const dim3 grid(nwg);
#ifdef CLIPK_EXPERIMENTS
  p.abl = clipk_opt_get(OPT_GEMM_ABL);                      // timing-only ablations: experiment builds only
  if (p.abl & 4) p.e.N = 0;                                 // every store out of range: same instructions, no traffic
#else
  p.abl = 0;
#endif
  p.stagger = clipk_opt_get(OPT_GEMM_STAGGER);
  hipStream_t st = (hipStream_t)stream;
  const int mode = clipk_opt_get(OPT_GEMM_EPI_GENERIC) == 1 ? EPI_GENERIC : epi_mode_for(a);
  if (mode == EPI_PLAIN) launch_v3<EPI_PLAIN>(p, grid, st);
  else if (mode == EPI_RES32) launch_v3<EPI_RES32>(p, grid, st);
  else if (mode == EPI_GELU_PRE) launch_v3<EPI_GELU_PRE>(p, grid, st);
  else if (mode == EPI_DGELU) launch_v3<EPI_DGELU>(p, grid, st);
  else if (mode == EPI_RES16) launch_v3<EPI_RES16>(p, grid, st);
  else if (mode == EPI_PRES16) launch_v3<EPI_PRES16>(p, grid, st);
  else if (mode == EPI_ROPE) launch_v3<EPI_ROPE>(p, grid, st);
  else if (mode == EPI_GELU_D8) launch_v3<EPI_GELU_D8>(p, grid, st);
  else if (mode == EPI_DGELU8) launch_v3<EPI_DGELU8>(p, grid, st);
  else if (mode == EPI_PLAIN_NB) launch_v3<EPI_PLAIN_NB>(p, grid, st);
  else if (mode == EPI_ROPE_IL) launch_v3<EPI_ROPE_IL>(p, grid, st);
  else if (mode == EPI_UNSUPPORTED || a->rope_cos) return CLIPK_ERR_UNSUPPORTED;   // rotation: its own mode only
  else launch_v3<EPI_GENERIC>(p, grid, st);
  return clipk_check_launch();
}
