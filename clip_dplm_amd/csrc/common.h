// common.h — shared device helpers for the gfx950 (CDNA4) kernels of libclipk.
// Wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/clipk.h"

#define CLIPK_ABI_VERSION 6
#define WAVE 64

// Kernel-selection options (core.hip): set explicitly through clipk_set_option(), never read from the environment.
// Every value of every option computes the same results; they choose between kernels / schedules that tests and
// tools/ want to compare.  Result-changing ablation switches exist only in builds with -DCLIPK_EXPERIMENTS.
enum clipk_opt {
  OPT_GEMM_KERNEL = 0,     // -1 auto, 1 generic kernel (gemm_nt.hip), 2 128x128 (v2), 3 persistent 256x256 (v3)
  OPT_GEMM_EPI_GENERIC,    // 1: run-time epilogue instead of the specialised modes
  OPT_GEMM_BM,             // v2: 256 = 256-row tile variant
  OPT_GEMM_STAGES,         // v2: 2 = two-stage LDS pipeline
  OPT_GEMM_NWG,            // v3: persistent grid size (multiple of 8), 0 = one workgroup per CU
  OPT_GEMM_STAGGER,        // start-up stagger of workgroups (0 = off)
  OPT_EPI_NT,              // 1: non-temporal epilogue stores
  OPT_WGRAD_KERNEL,        // -1 auto, 2 128x128 kernel, 3 256x256 kernel (8-phase schedule), 4 256x256 software-pipelined
  OPT_ATTN_WHOLE_FWD,      // -1 auto, 0 off, 1 on: whole-head forward for short heads
  OPT_ATTN_FUSED_BWD,      // -1 auto, 0 off, 1 on: whole-head backward for short heads
  OPT_ATTN_FUSED_WAVES,    // waves per workgroup in the whole-head backward: 0 auto (hd <= 32: 4, hd 96: 8), 4, 8
  OPT_WGRAD_SPLITS,        // v3 weight-gradient kernel: M splits (0 = about one workgroup per CU)
  OPT_SIMCE_KERNEL,        // -1 auto (tiled LSE pass for >= 64 queries), 1 first-generation kernel, 2 tiled
  OPT_GEMM_ABL,            // CLIPK_EXPERIMENTS builds only: timing ablations that change results
  OPT_ATTN_ROW_STORES,     // whole-head forward: 2 = rotated q / k rows written back four lanes to a row from LDS (measured slower; default off)
  OPT_GEMM_F32_SPLITS,     // skinny exact-f32 Linear: cross-workgroup splits of the contraction (0 = auto, 1 .. 8)
  OPT_COUNT
};
int clipk_opt_get(int which);      // core.hip

typedef __attribute__((ext_vector_type(8))) short bf16x8;     // 8 bf16 = one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;     // 4 bf16 = one ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(4))) float f32x4;      // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;    // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
// two at a time: ONE v_cvt_pk_bf16_f32 (converting the halves separately costs two of them plus an sdwa-or)
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float v2f;
  typedef __attribute__((ext_vector_type(2))) __bf16 v2b;
  const v2b v = __builtin_convertvector(v2f{lo, hi}, v2b);
  return __builtin_bit_cast(unsigned int, v);
}

// erf-GELU with erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7: f32-rounding level, far below the bf16
// rounding of every consumer), written for the raw v_rcp_f32 / v_exp_f32 instructions: ~15 VALU per element
// instead of ocml erff's branchy ~40.  The GEMM epilogues that apply it are VALU-bound (profiles/).
//   erf(|x|/sqrt2) = 1 - (a1 t + ... + a5 t^5) * exp(-x^2/2),  t = 1 / (1 + p |x| / sqrt2)
// exp(-x^2/2) is shared by the erf and by the Gaussian pdf of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& erf_abs, float& e) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);        // exp(-x^2/2) = 2^(-x^2 * log2(e) / 2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  erf_abs = fmaf(-poly * t, e, 1.0f);                                  // erf(|x| / sqrt 2)
}
__device__ __forceinline__ float gelu_erf(float x) {          // transformers modeling_esm.py:82-86 / nn.GELU()
  float erf_abs, e;
  gelu_parts(x, erf_abs, e);
  return fmaf(0.5f * fabsf(x), erf_abs, 0.5f * x);            // 0.5 x (1 + sign(x) erf|.|)
}
__device__ __forceinline__ float gelu_erf_grad(float x) {     // Phi(x) + x phi(x)
  float erf_abs, e;
  gelu_parts(x, erf_abs, e);
  const float cdf = fmaf(copysignf(0.5f, x), erf_abs, 0.5f);
  return fmaf(x * 0.39894228040143267794f, e, cdf);
}
// packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32: 2 results per issue slot): see gelu_erf8 below
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { return f32x2{c, c}; }
__device__ __forceinline__ float act_apply(float x, int act) {
  if (act == CLIPK_ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == CLIPK_ACT_GELU) return gelu_erf(x);
  if (act == CLIPK_ACT_CELU) return x > 0.f ? x : expm1f(x);                    // F.celu, alpha = 1
  if (act == CLIPK_ACT_SOFTPLUS) return x > 20.f ? x : log1pf(expf(x));         // F.softplus (beta 1, threshold 20)
  return x;
}
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == CLIPK_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (act == CLIPK_ACT_GELU) return gelu_erf_grad(x);
  if (act == CLIPK_ACT_CELU) return x > 0.f ? 1.f : expf(x);
  if (act == CLIPK_ACT_SOFTPLUS) return x > 20.f ? 1.f : 1.0f / (1.0f + expf(-x));
  return 1.f;
}

// ---- GELU'(u) as an 8-bit code: what the FFN keeps for its backward instead of the bf16 pre-activation u.
// GELU' takes values in [-0.1290, 1.1290]; the code is the nearest of 256 levels of step 0.005 starting at -0.13
// (levels -0.13 ... 1.145; error <= 0.0025 absolute - the size of a bf16 rounding of a value near 1): half the bytes of u
// in the two store-bound FFN epilogues, no erf in the backward one.  The forward value GELU(u) is untouched.
// The grid is chosen so that the two values most hidden units take ARE code points (on the round-3 grid
// [-0.13, 1.13] / 255 a dead unit decoded to -0.0015 and a saturated one to 1.0015, a bias of one sign per class of unit
// in every layer): GELU' = 0 -> code 26 -> fma(26, 0.005f, -0.13f) = 1.9e-9, GELU' = 1 -> code 226 -> exactly 1.0f.
constexpr float CLIPK_GD8_MIN = -0.13f, CLIPK_GD8_STEP = 0.005f, CLIPK_GD8_INV = 200.0f, CLIPK_GD8_OFF = 26.0f;
__device__ __forceinline__ unsigned gelu_grad_code(float d) {                 // d = GELU'(u) -> 0 .. 255
  float t = fmaf(d, CLIPK_GD8_INV, CLIPK_GD8_OFF + 0.5f);                     // + 0.5: v_cvt_u32_f32 truncates
  t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
  return (unsigned)t;
}
__device__ __forceinline__ float gelu_grad_decode(unsigned q) { return fmaf((float)q, CLIPK_GD8_STEP, CLIPK_GD8_MIN); }
// eight codes -> two dwords (columns in byte order)
__device__ __forceinline__ u32x2 gelu_grad_pack8(const float (&d)[8]) {
  u32x2 w;
  w[0] = gelu_grad_code(d[0]) | (gelu_grad_code(d[1]) << 8) | (gelu_grad_code(d[2]) << 16) | (gelu_grad_code(d[3]) << 24);
  w[1] = gelu_grad_code(d[4]) | (gelu_grad_code(d[5]) << 8) | (gelu_grad_code(d[6]) << 16) | (gelu_grad_code(d[7]) << 24);
  return w;
}
// GELU(x) as packed bf16 and (CODE) the code of GELU'(x) for EIGHT elements from one evaluation of erf / exp each - the
// FFN epilogue of the Linear kernels, whose instruction time is purely additive to the main loop (MFMA and VALU of one
// SIMD do not overlap on gfx950).  Round 4 rewrite, counted in the ISA of gemm_nt_v3_kernel<GELU_D8>
// per output pair: 17 packed + 7 plain VALU + 4 transcendentals + 3.3 hazard s_nop  ->  13 + 7 + 4 + 0:
//   * written STAGE BY STAGE over the four pairs: a v_pk_*_f32 whose operand is the previous packed result needs a wait
//     state, and with one pair's dependent chain after the other hipcc filled those with s_nop (214 per tile and wave);
//   * |x| as the source modifier of a plain v_fma_f32 (VOP3P has none): no v_and, no separate packed fma for 1 + p|x|;
//   * Phi(x) = 1/2 + copysign(1/2, x) erf(|x| / sqrt 2) once, GELU = x Phi and GELU' = Phi + x phi(x) both from it (the
//     round-3 form evaluated 0.5 |x| erf + 0.5 x and a separately scaled cdf: one packed multiply more);
//   * erf still by Abramowitz-Stegun 7.1.26 (five coefficients, |err| <= 1.5e-7): with the three-coefficient 7.1.25
//     the error reaches 3.2 half-ulps of the bf16 result at x = -3.4 (measured over the f32 grid), 7.1.26 stays at 0.11.
template <bool CODE>
__device__ __forceinline__ void gelu_erf8(const float (&v)[8], u32x4& o, unsigned (&q)[2]) {
  constexpr float P = 0.3275911f * 0.70710678118654752440f;
  // (sched_barrier: hipcc's scheduler otherwise re-serialises the stages into one dependent chain per pair to save
  // registers, and pays for it with a wait state between every two packed instructions)
#define GELU8_FENCE() __builtin_amdgcn_sched_barrier(0)
  f32x2 x[4], t[4], e[4], pl[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) x[c] = f32x2{v[2 * c], v[2 * c + 1]};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    t[c] = f32x2{__builtin_amdgcn_rcpf(fmaf(fabsf(v[2 * c]), P, 1.0f)), __builtin_amdgcn_rcpf(fmaf(fabsf(v[2 * c + 1]), P, 1.0f))};
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x2 xx = (x[c] * splat2(-0.72134752044448170368f)) * x[c];
    e[c] = f32x2{__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  }
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) pl[c] = __builtin_elementwise_fma(splat2(1.061405429f), t[c], splat2(-1.453152027f));
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) pl[c] = __builtin_elementwise_fma(pl[c], t[c], splat2(1.421413741f));
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) pl[c] = __builtin_elementwise_fma(pl[c], t[c], splat2(-0.284496736f));
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) pl[c] = __builtin_elementwise_fma(pl[c], t[c], splat2(0.254829592f));
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) pl[c] = pl[c] * t[c];
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) pl[c] = __builtin_elementwise_fma(-pl[c], e[c], splat2(1.0f));        // erf(|x| / sqrt 2)
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c)                                                                        // Phi(x)
    pl[c] = __builtin_elementwise_fma(f32x2{copysignf(0.5f, v[2 * c]), copysignf(0.5f, v[2 * c + 1])}, pl[c], splat2(0.5f));
  GELU8_FENCE();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x2 y = x[c] * pl[c];
    o[c] = pack_bf16x2(y[0], y[1]);
  }
  GELU8_FENCE();
  if constexpr (CODE) {
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = x[c] * splat2(0.39894228040143267794f);
  GELU8_FENCE();
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = __builtin_elementwise_fma(t[c], e[c], pl[c]);                    // GELU' = Phi + x phi
  GELU8_FENCE();
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = __builtin_elementwise_fma(t[c], splat2(CLIPK_GD8_INV), splat2(CLIPK_GD8_OFF));
  GELU8_FENCE();
    // v_cvt_pk_u8_f32 ROUNDS to nearest and saturates to 0 .. 255 (a + 0.5 as for the truncating v_cvt_u32_f32 left every
    // code half a level high: the GPU test bounds the mean code error)
    unsigned w0 = 0u, w1 = 0u;
    w0 = __builtin_amdgcn_cvt_pk_u8_f32(t[0][0], 0, w0); w0 = __builtin_amdgcn_cvt_pk_u8_f32(t[0][1], 1, w0);
    w0 = __builtin_amdgcn_cvt_pk_u8_f32(t[1][0], 2, w0); w0 = __builtin_amdgcn_cvt_pk_u8_f32(t[1][1], 3, w0);
    w1 = __builtin_amdgcn_cvt_pk_u8_f32(t[2][0], 0, w1); w1 = __builtin_amdgcn_cvt_pk_u8_f32(t[2][1], 1, w1);
    w1 = __builtin_amdgcn_cvt_pk_u8_f32(t[3][0], 2, w1); w1 = __builtin_amdgcn_cvt_pk_u8_f32(t[3][1], 3, w1);
    q[0] = w0; q[1] = w1;
  }
#undef GELU8_FENCE
}

// second derivative of the activations the ICNN potential uses (2_icnn_core.py:121-127: CELU default, softplus): the
// training branch back-propagates through T(x) = dPsi/dx, i.e. through act'
__device__ __forceinline__ float act_grad2(float x, int act) {
  if (act == CLIPK_ACT_CELU) return x > 0.f ? 0.f : expf(x);
  if (act == CLIPK_ACT_SOFTPLUS) {
    if (x > 20.f) return 0.f;
    const float sg = 1.0f / (1.0f + expf(-x));
    return sg * (1.0f - sg);
  }
  return 0.f;                                             // none / relu (piecewise linear)
}

// ---- dropout: counter-based mask, keep(seed, element index) = hash32(...) >= thr with thr = p * 2^32.  No state, no
// stored mask: the backward recomputes it from the same (seed, index).  The index is the element's row-major position
// in the tensor the reference applies nn.Dropout to; the hash is the murmur3 32-bit finaliser over the folded 64-bit
// index (tests/ reproduce it in torch integer arithmetic to check every dropout site against masked references).
__device__ __forceinline__ bool drop_keep(unsigned seed, unsigned long long idx, unsigned thr) {
  unsigned h = (unsigned)idx * 0x9E3779B1u + (unsigned)(idx >> 32) * 0x85EBCA77u + seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h >= thr;
}
__device__ __forceinline__ float drop_mul(unsigned seed, unsigned long long idx, unsigned thr, float scale) {
  return drop_keep(seed, idx, thr) ? scale : 0.f;
}
// Dropout under hipGraph replay: a captured launch carries its seed as a constant, so every replay would draw the same
// mask.  clipk_set_dropout_epoch(ptr) registers ONE device word that every dropout site adds (times an odd constant) to
// its seed when the launch is made while the pointer is registered: the captured step increments the word once per replay
// (after the backward, which recomputes the masks) and each replay drops different elements.  nullptr (the default): the
// seed as given - eager behaviour, bit for bit.
__device__ __forceinline__ unsigned drop_seed_eff(unsigned seed, const unsigned* epoch) {
  return epoch ? seed + epoch[0] * 0x9E3779B9u : seed;
}
const unsigned* clipk_drop_epoch();                      // host side: the registered pointer or nullptr (core.hip)

// ---- pair-interleaved head order (round 4: RoPE of the hd-24 ESM heads in the qkv GEMM's epilogue).  The first il_rows
// output columns of a fused [q | k | v] projection (the q and k sections) are computed in an order in which the two
// partners of a rotate-half pair (d, d + hd/2) are NEIGHBOURS (2 j, 2 j + 1), so that a lane's 8 consecutive columns
// hold four complete pairs whatever the head dim (hd = 24 does not tile the epilogue's 64-column wave slices).  q . k is
// invariant under a common permutation of the head dim, so the attention kernels run unchanged on such q / k; only RoPE
// (forward: GEMM epilogue; backward: RoPE^T where the gradient rows are written) and the weight copies / weight-gradient
// rows know about the order.  il_src(r): the ORIGINAL row (output column) that permuted row r holds.
__host__ __device__ inline int il_src(int r, int hd, int il_rows) {
  if (r >= il_rows) return r;
  const int d = r % hd;
  return (r - d) + (d >> 1) + (d & 1) * (hd >> 1);
}

// wave-wide reductions over 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks b and b+8 share an XCD (observed round-robin
// dispatch, MI355X_MICROARCH.md §Workgroup dispatch), so give each XCD a contiguous chunk of the tile
// order.  Speed only — any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

static inline int clipk_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? CLIPK_OK : CLIPK_ERR_LAUNCH;
}
// hipFuncSetAttribute is a per-device setting: remember which devices have it (idempotent work, so a race between two
// host threads at most repeats it).
#include <atomic>
template <typename F>
static inline void clipk_once_per_device(std::atomic<uint64_t>& mask, F&& f) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(mask.load(std::memory_order_acquire) & bit)) {
    f();
    mask.fetch_or(bit, std::memory_order_release);
  }
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
