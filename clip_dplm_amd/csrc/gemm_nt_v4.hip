// gemm_nt_v4.hip — persistent 128 x 256-tile, 4-wave kernel for clipk_gemm_nt, TWO workgroups per CU
// (K % 32 == 0, K >= 128).  Same contract and epilogue as gemm_nt_v2.hip / gemm_nt_v3.hip; option gemm_kernel = 4.
//
// Why a third structure.  In gemm_nt_v3 (one 8-wave workgroup per CU, 256 x 256 tile) nothing runs on a CU while its
// workgroup is in the epilogue: measured (DESIGN.md 3.1 / 3.2, K = 480) a 12 us main loop is followed by ~5 us of
// epilogue instructions - 8 us for the GELU epilogues, which are VALU-bound - and the tile's output stream.  The matrix
// pipe idles for all of it.  Here a CU holds two independent 4-wave workgroups (256 VGPRs per wave, 64 KiB of LDS
// each): they drift apart by themselves, so one workgroup's epilogue VALU work and store drain run under the other's
// MFMAs, and one workgroup's fragment fetches under the other's MFMA blocks (what the two wave groups of v3 do for
// each other by construction).  The price is operand reuse: a 128 x 256 tile moves 1.5x the L2 -> LDS bytes per FLOP of
// a 256 x 256 tile.
//
// Workgroup: 4 waves = 1 (m) x 4 (n), wave tile 128 m x 64 n = the wave tile of v3 (128 accumulator VGPRs, same
// fragment layout, same quadrant order, same epilogue call).  K-tile = 64, rows of 128 B with the 16-byte chunks
// XOR-swizzled by (row >> 1) & 7, filled by buffer-descriptor LDS-DMA in pieces of 8 rows x 128 B: WHOLE cache lines
// per row (a first version with 32-deep K-tiles - 16 rows x 64 B per wave instruction - ran at 0.57 of v3's main
// loop: every line crossed the L2 -> L1 path twice and the address unit saw twice the lines per byte,
// cdna_hip_programming.md "full 128-B lines").  ONE buffer of four half-tiles, defined by consumption order as in v3:
//     X mh = the mh-th 64 rows of the m-wave (8 KiB), W nh = the nh-th 32 rows of all four n-waves (16 KiB)
// and each half-tile is refilled for the NEXT K-tile as soon as every wave has retired its reads of it:
//     phase 0  barrier (W nh0, X mh0 of T landed)  reads W nh0 + X mh0     quadrant (n0, m0)
//     phase 1  barrier (W nh1 landed)              refill W nh0, X mh0 of T+1   reads W nh1   quadrant (n1, m0)
//     phase 2  barrier (X mh1 landed)              refill W nh1 of T+1          reads X mh1   quadrant (n1, m1)
//     phase 3  barrier                             refill X mh1 of T+1                        quadrant (n0, m1)
// One barrier per phase: it says both "what this phase reads has landed" (every wave waited, with a counted vmcnt, for
// its own pieces) and "what the previous phase read is dead".  A refill has three phases to land; a wave that has to
// wait is covered by the other workgroup.  The ring runs across output tiles: in a tile's last K-tile the refills
// fetch K-tile 0 of the workgroup's next tile, whose stores then drain under the next main loop (vmcnt is in-order:
// the three waits of a tile's first K-tile allow for the NS store instructions of the epilogue before it).
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 256, BK = 64;
constexpr int XH_BYTES = 64 * BK * 2;            // 8 KiB
constexpr int WH_BYTES = 128 * BK * 2;           // 16 KiB
constexpr int XH0 = 0, XH1 = XH_BYTES, WH0 = 2 * XH_BYTES, WH1 = 2 * XH_BYTES + WH_BYTES;
constexpr int BUF_BYTES = 2 * XH_BYTES + 2 * WH_BYTES;     // 48 KiB
constexpr int SLAB_BYTES = 16 * 64 * 4;          // per-wave epilogue slab (XOR-swizzled, unpadded)
constexpr int LDS_BYTES = BUF_BYTES + 4 * SLAB_BYTES;      // 64 KiB: two workgroups per CU

struct Params {
  const unsigned short* A; long lda;
  const unsigned short* B; long ldb;
  int M, N, K;
  EpiArgs e;
  int ntn, ntiles;
  int abl;          // timing-only ablation (CLIPK_EXPERIMENTS builds): 1 = no epilogue
  int stagger;      // the second workgroup of every CU (blockIdx >= gridDim / 2) starts stagger x ~3.9 us late
};

// one quadrant: 2 n-tiles x 4 m-tiles x 2 k-halves = 16 MFMAs (k outer so dependent accumulations sit 8 apart);
// KLO = 1: upper k-half only, for the last K-tile of a K % 64 == 32 problem (fetched as [K - 64, K), whose lower
// half was already accumulated by the K-tile before)
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
#define CLIPK_STR2(x) #x
#define CLIPK_STR(x) CLIPK_STR2(x)
// counted wait that tolerates NS younger-than-the-loads store instructions (NS is a template constant 0 / 16 / 32)
#define CLIPK_VMCNT_PLUS(base, ns)                                                              \
  do {                                                                                          \
    if ((ns) == 0) asm volatile("s_waitcnt vmcnt(" CLIPK_STR(base) ")" ::: "memory");           \
    else if ((ns) == 16) asm volatile("s_waitcnt vmcnt(" CLIPK_STR(base) "+16)" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(" CLIPK_STR(base) "+32)" ::: "memory");                  \
  } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_nt_v4_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NS = epi_stores(MODE, 8) < 0 ? 0 : epi_stores(MODE, 8);
  static_assert(NS == 0 || NS == 16 || NS == 32, "vmcnt bookkeeping below knows 0 / 16 / 32 stores");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.M, N = p.N, K = p.K;

  // ---- LDS-DMA assignment: a wave instruction fills one piece = 8 rows x 128 B.  Wave w takes pieces 2w, 2w + 1 of an
  // X half-tile and 4w .. 4w + 3 of a W half-tile.  lane -> (row in piece = lane >> 3, physical 16-B slot = lane & 7);
  // source chunk = slot ^ ((row >> 1) & 7).  Per-lane byte offsets are fixed; the descriptors (one per half-tile kind)
  // move per output tile.
  unsigned xv[2], wv[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = 8 * (2 * wn + i) + (lane >> 3);               // row of the X half image (0 .. 63)
    const int kch = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    xv[i] = (unsigned)(((long)r * p.lda + kch) * 2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * (4 * wn + i) + (lane >> 3);               // row of the W half image (0 .. 127): n-wave r >> 5
    const int kch = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
    wv[i] = (unsigned)(((long)((r >> 5) * 64 + (r & 31)) * p.ldb + kch) * 2);
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
  const int nk = (K + BK - 1) / BK;                             // >= 2 (launcher)
  const bool tail = (K & 63) != 0;                              // K % 64 == 32
  // K-tile T covers k in [64 T, 64 T + 64), except the last one of a K % 64 == 32 problem, which is fetched as
  // [K - 64, K): always in range, and only its upper half is multiplied (quad<.., KLO = 1>)
  auto k_of = [&](int T) { return (tail && T == nk - 1) ? K - BK : T * BK; };
  auto stage_x = [&](rsrc_t d, int T, int region) {
    const int k0 = k_of(T);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)(smem + region + (2 * wn + i) * 1024),
                                               16, (int)xv[i], k0 * 2, 0, 0);
  };
  auto stage_w = [&](rsrc_t d, int T, int region) {
    const int k0 = k_of(T);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(d, (__attribute__((address_space(3))) void*)(smem + region + (4 * wn + i) * 1024),
                                               16, (int)wv[i], k0 * 2, 0, 0);
  };

  const int frow = lane & 15, fch = lane >> 4, lane_sw = (frow >> 1) & 7;
  int xo[2], wo[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int choff = ((kk * 4 + fch) ^ lane_sw) << 4;
    xo[kk] = frow * 128 + choff;
    wo[kk] = WH0 + (wn * 32 + frow) * 128 + choff;
  }

  f32x4 acc[4][8];          // [n-tile i][m-tile j]: rows n = 4g+r, col m = lane&15
  bf16x8 xf[4][2], wf[2][2][2];
#ifdef CLIPK_EXPERIMENTS
  const bool prio = !(p.abl & 8);                               // ablation 8: MFMA blocks without s_setprio 1
#else
  constexpr bool prio = true;
#endif
  bool first = true;                                            // no epilogue stores of a previous tile in flight
  bool more = true;                                             // set per tile before its last K-tile

  // KT 0: a tile's first K-tile (the previous tile's NS stores sit in the queue behind this K-tile's pieces);
  //    1: a K-tile in the middle;  2: the last K-tile (refills fetch K-tile 0 of the NEXT output tile, if there is one,
  //       through descriptors `setup` has already moved; T + 1 -> 0)
  // Queue per wave, in issue order, for K-tile T: [W nh0 + X mh0: 6] [W nh1: 4] [X mh1: 2], each group requested one
  // K-tile earlier in phases 1 / 2 / 3.
  auto k_tile = [&](auto kt_c, int T) {
    constexpr int KT = decltype(kt_c)::value;
    const int Tn = KT == 2 ? 0 : T + 1;                         // the K-tile the refills fetch
    const bool refill = KT != 2 || more;
    // ---- phase 0: W nh0 + X mh0 of this K-tile landed (younger: W nh1 4, X mh1 2 [, stores])
    if (KT == 0 && !first) CLIPK_VMCNT_PLUS(6, NS);
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[0][t][kk] = *reinterpret_cast<const bf16x8*>(smem + wo[kk] + t * 2048);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(smem + xo[kk] + XH0 + j * 2048);
    CLIPK_SB();
    if (prio) __builtin_amdgcn_s_setprio(1);
    if (KT == 2 && tail) quad<0, 0, 1>(acc, wf, xf);
    else quad<0, 0>(acc, wf, xf);
    if (prio) __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
    // ---- phase 1: W nh1 landed (younger: X mh1 2 [, stores]); W nh0 / X mh0 are dead -> refill
    if (KT == 0 && !first) CLIPK_VMCNT_PLUS(2, NS);
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    if (refill) { stage_w(dw0, Tn, WH0); stage_x(dx0, Tn, XH0); }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) wf[1][t][kk] = *reinterpret_cast<const bf16x8*>(smem + wo[kk] + WH_BYTES + t * 2048);
    CLIPK_SB();
    if (prio) __builtin_amdgcn_s_setprio(1);
    if (KT == 2 && tail) quad<1, 0, 1>(acc, wf, xf);
    else quad<1, 0>(acc, wf, xf);
    if (prio) __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
    // ---- phase 2: X mh1 landed (younger: [stores,] the 6 pieces just requested - or nothing); W nh1 is dead -> refill
    if (!refill) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (KT == 0 && !first) CLIPK_VMCNT_PLUS(6, NS);
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    CLIPK_SB(); CLIPK_BAR(); CLIPK_SB();
    if (refill) stage_w(dw1, Tn, WH1);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) xf[j][kk] = *reinterpret_cast<const bf16x8*>(smem + xo[kk] + XH1 + j * 2048);
    CLIPK_SB();
    if (prio) __builtin_amdgcn_s_setprio(1);
    if (KT == 2 && tail) quad<1, 1, 1>(acc, wf, xf);
    else quad<1, 1>(acc, wf, xf);
    if (prio) __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
    // ---- phase 3: nothing to fetch; X mh1 is dead -> refill
    CLIPK_BAR(); CLIPK_SB();
    if (refill) stage_x(dx1, Tn, XH1);
    if (prio) __builtin_amdgcn_s_setprio(1);
    if (KT == 2 && tail) quad<0, 1, 1>(acc, wf, xf);
    else quad<0, 1>(acc, wf, xf);
    if (prio) __builtin_amdgcn_s_setprio(0);
    CLIPK_SB();
  };

  // Two workgroups that start together and walk tiles of equal cost stay in lockstep - both in their main loops, then
  // both in their epilogues - and nothing overlaps.  Half a tile of head start for one of them is what makes one's
  // epilogue fall under the other's main loop, and equal periods keep it there.
  // (which of the two is "second" on its CU: the wave slot inside the SIMD, HW_REG_HW_ID[3:0] - the dispatcher's
  // workgroup -> CU order is not ours to know)
  if (p.stagger > 0 && (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 0xF) != 0)
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  int bid = blockIdx.x;
  setup(bid);
  stage_w(dw0, 0, WH0); stage_x(dx0, 0, XH0); stage_w(dw1, 0, WH1); stage_x(dx1, 0, XH1);
  while (true) {
    const int cm0 = m0, cn0 = n0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    k_tile(std::integral_constant<int, 0>{}, 0);
    for (int T = 1; T < nk - 1; ++T) k_tile(std::integral_constant<int, 1>{}, T);
    bid += gridDim.x;
    more = bid < p.ntiles;
    if (more) setup(bid);                                       // this tile's operands are all requested: move on
    // epilogue index math from a fresh lane id (kept live across the main loop it costs registers there)
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int gn_e = cn0 + wn * 64 + (lane_e & 7) * 8;
    k_tile(std::integral_constant<int, 2>{}, nk - 1);
    float bv[8];
    epi_load_bias(p.e, gn_e, bv);
    float* eb = reinterpret_cast<float*>(smem + BUF_BYTES) + wn * (SLAB_BYTES / 4);
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
    gemm_epilogue<MODE, 8, true>(p.e, acc, eb, lane_e, cm0, gn_e, bv);
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
void launch_v4(const Params& p, dim3 grid, hipStream_t st) {
  static std::atomic<uint64_t> attr_set{0};
  clipk_once_per_device(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_v4_kernel<MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  });
  hipLaunchKernelGGL((gemm_nt_v4_kernel<MODE>), grid, dim3(256), LDS_BYTES, st, p);
}

int cu_count4() {
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

// called by clipk_gemm_nt (gemm_nt.hip) after it validated the arguments (K % 32 == 0, K >= 128)
extern "C" int clipk_gemm_nt_v4_launch(const clipk_gemm_args* a, void* stream) {
  Params p;
  p.A = (const unsigned short*)a->A; p.lda = a->lda;
  p.B = (const unsigned short*)a->B; p.ldb = a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.e = epi_args_from(a);
  const int ntm = (a->M + BM - 1) / BM, ntn = (a->N + BN - 1) / BN;
  p.ntn = ntn;
  p.ntiles = ntm * ntn;
  // two persistent workgroups per CU; a multiple of 8 keeps "workgroup b runs on XCD b % 8" true for every tile it walks
  int nwg = (2 * cu_count4()) & ~7;
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
  if (mode == EPI_PLAIN) launch_v4<EPI_PLAIN>(p, grid, st);
  else if (mode == EPI_RES32) launch_v4<EPI_RES32>(p, grid, st);
  else if (mode == EPI_GELU_PRE) launch_v4<EPI_GELU_PRE>(p, grid, st);
  else if (mode == EPI_DGELU) launch_v4<EPI_DGELU>(p, grid, st);
  else if (mode == EPI_RES16) launch_v4<EPI_RES16>(p, grid, st);
  else if (mode == EPI_PRES16) launch_v4<EPI_PRES16>(p, grid, st);
  else if (mode == EPI_ROPE) launch_v4<EPI_ROPE>(p, grid, st);
  else if (mode == EPI_GELU_D8) launch_v4<EPI_GELU_D8>(p, grid, st);
  else if (mode == EPI_DGELU8) launch_v4<EPI_DGELU8>(p, grid, st);
  else if (mode == EPI_PLAIN_NB) launch_v4<EPI_PLAIN_NB>(p, grid, st);
  else if (mode == EPI_ROPE_IL) launch_v4<EPI_ROPE_IL>(p, grid, st);
  else if (mode == EPI_UNSUPPORTED || a->rope_cos) return CLIPK_ERR_UNSUPPORTED;   // rotation: its own mode only
  else launch_v4<EPI_GENERIC>(p, grid, st);
  return clipk_check_launch();
}
