// common.h — shared device helpers for the gfx950 (CDNA4) kernels of libclipk.
// Wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/clipk.h"

#define CLIPK_ABI_VERSION 5
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
// Two elements at a time on the packed-f32 pipe (v_pk_fma_f32 / v_pk_mul_f32: 2 results per issue slot): 9 VALU + 2
// transcendentals per element instead of 15 + 2.  Same arithmetic as gelu_parts, element for element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float c) { return f32x2{c, c}; }
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2 ax, f32x2& erf_abs, f32x2& e) {
  const f32x2 d = __builtin_elementwise_fma(splat2(0.3275911f * 0.70710678118654752440f), ax, splat2(1.0f));
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 xx = (x * splat2(-0.72134752044448170368f)) * x;
  e = f32x2{__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  f32x2 poly = __builtin_elementwise_fma(splat2(1.061405429f), t, splat2(-1.453152027f));
  poly = __builtin_elementwise_fma(poly, t, splat2(1.421413741f));
  poly = __builtin_elementwise_fma(poly, t, splat2(-0.284496736f));
  poly = __builtin_elementwise_fma(poly, t, splat2(0.254829592f));
  erf_abs = __builtin_elementwise_fma(-(poly * t), e, splat2(1.0f));
}
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  f32x2 erf_abs, e;
  gelu_parts2(x, ax, erf_abs, e);
  return __builtin_elementwise_fma(ax * splat2(0.5f), erf_abs, x * splat2(0.5f));
}
__device__ __forceinline__ f32x2 gelu_erf_grad2(f32x2 x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  f32x2 erf_abs, e;
  gelu_parts2(x, ax, erf_abs, e);
  const f32x2 half_sgn = {copysignf(0.5f, x[0]), copysignf(0.5f, x[1])};
  const f32x2 cdf = __builtin_elementwise_fma(half_sgn, erf_abs, splat2(0.5f));
  return __builtin_elementwise_fma(x * splat2(0.39894228040143267794f), e, cdf);
}

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
// GELU' takes values in [-0.1290, 1.1290]; the code is the nearest of 256 levels over [-0.13, 1.13] (step 0.00494, error
// <= 0.0025 absolute - the size of a bf16 rounding of a value near 1): half the bytes of u in the two store-bound FFN
// epilogues, no erf in the backward one.  The forward value GELU(u) is untouched.
constexpr float CLIPK_GD8_MIN = -0.13f, CLIPK_GD8_STEP = 1.26f / 255.0f, CLIPK_GD8_INV = 255.0f / 1.26f;
__device__ __forceinline__ unsigned gelu_grad_code(float d) {                 // d = GELU'(u) -> 0 .. 255
  float t = fmaf(d, CLIPK_GD8_INV, -CLIPK_GD8_MIN * CLIPK_GD8_INV + 0.5f);    // + 0.5: v_cvt_u32_f32 truncates
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
// GELU(x) and the code of GELU'(x) for two elements from ONE evaluation of erf / exp, the code's scale and offset folded
// into the derivative's own two fmas: t = (Phi + x phi) * INV - MIN * INV, written into byte `b`, `b + 1` of `w` by
// v_cvt_pk_u8_f32, which ROUNDS to nearest and saturates to 0 .. 255 (measured: with the + 0.5 of the truncating
// v_cvt_u32_f32 path every code came out half a level high; the GPU test bounds the mean code error)
__device__ __forceinline__ f32x2 gelu_erf2_code(f32x2 x, unsigned& w, int b) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  f32x2 erf_abs, e;
  gelu_parts2(x, ax, erf_abs, e);
  constexpr float C0 = 0.5f * CLIPK_GD8_INV - CLIPK_GD8_MIN * CLIPK_GD8_INV;
  const f32x2 hs = {copysignf(0.5f * CLIPK_GD8_INV, x[0]), copysignf(0.5f * CLIPK_GD8_INV, x[1])};
  const f32x2 cdf = __builtin_elementwise_fma(hs, erf_abs, splat2(C0));
  const f32x2 t = __builtin_elementwise_fma(x * splat2(0.39894228040143267794f * CLIPK_GD8_INV), e, cdf);
  w = __builtin_amdgcn_cvt_pk_u8_f32(t[0], b, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(t[1], b + 1, w);
  return __builtin_elementwise_fma(ax * splat2(0.5f), erf_abs, x * splat2(0.5f));
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
