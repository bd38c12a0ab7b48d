// gemm_epilogue.h — epilogue shared by the clipk_gemm_nt kernels (gemm_nt_v2.hip, gemm_nt_v3.hip).
//
// Each wave owns a [16*NJ m][64 n] block of the output as accumulators acc[n-tile i][m-tile j] (swapped operand
// roles: lane holds rows n = 16i + 4g + r of column m = 16j + (lane & 15)).  One 16-row m-tile at a time goes through a
// wave-private LDS slab [16 m][64 n (+4)] f32 so that every lane then owns 8 consecutive columns of one row and all
// global accesses are 16 / 32 bytes per lane.
//
// The hot argument combinations are compile-time MODEs with a STRAIGHT-LINE body: on gfx9 loads and stores share
// the vmcnt counter, and with run-time `if (p.residual)` / `if (gm < M)` branches hipcc falls back to
// `s_waitcnt vmcnt(0)` in every half-iteration, i.e. each 8-row slice waits for the previous slice's stores to
// reach memory.  In the specialised modes the operand loads (residual, GELU' argument) are issued one slice ahead
// of their use with clamped addresses, only the stores are predicated, and the compiler's counted vmcnt(N) leaves
// the stores in flight.
#pragma once
#include "common.h"
#include <stdlib.h>

struct EpiArgs {
  void* C; long ldc; int c_f32;
  int M, N;
  const float* bias;
  int act;
  unsigned short* out_preact; long ldp;
  const unsigned short* dact_aux; long ldd; int dact;
  const void* residual; long ldr; int r_f32;
  float alpha;
  int aux_u8;       // out_preact / dact_aux hold 8-bit GELU' codes (clipk.h aux_dtype) instead of the bf16 pre-activation
  int nt;           // 1: non-temporal (streaming) output stores (epi_args_from decides)
  unsigned drop_thr, drop_seed; float drop_scale;   // dropout after the activation (generic epilogue only); thr 0 = off
  const unsigned* drop_epoch;                       // common.h drop_seed_eff (nullptr outside a captured step)
  const float* rope_cos; const float* rope_sin; int rope_L, rope_hd, rope_cols, rope_row0;   // EPI_ROPE only
};

enum {
  EPI_GENERIC = -1,   // everything decided at run time
  EPI_PLAIN = 0,      // bf16 out (+bias)
  EPI_RES32 = 1,      // f32 out = acc (+bias) + f32 residual
  EPI_GELU_PRE = 2,   // bf16 pre-activation out and bf16 GELU out (+bias)
  EPI_DGELU = 3,      // bf16 out = acc * GELU'(aux)
  EPI_RES16 = 4,      // f32 out = acc (+bias) + bf16 residual   (post-LN layers: the residual is the bf16 LayerNorm output)
  EPI_PRES16 = 5,     // bf16 out = acc (+bias) + bf16 residual  (their input gradients: bf16 residual-path gradient)
  EPI_ROPE = 6,       // bf16 out = rotate-half RoPE of (acc + bias) on the first rope_cols columns, plain on the rest
  EPI_GELU_D8 = 7,    // u8 GELU'(pre-activation) code out and bf16 GELU out (+bias)
  EPI_DGELU8 = 8,     // bf16 out = acc * decode(u8 aux)
  EPI_PLAIN_NB = 9,   // bf16 out, no bias (the plain input gradients)
  EPI_ROPE_IL = 10,   // bf16 out = RoPE of (acc + bias) on the first rope_cols columns, whose heads are PAIR-INTERLEAVED
                      // (common.h il_src): a pair is two neighbouring columns of one lane - any head dim % 8 == 0
};
// Every specialised mode requires alpha == 1 (no caller scales the product: one packed multiply per output pair gone from
// every epilogue), and the modes that only input-gradient GEMMs use - DGELU, DGELU8, PRES16, PLAIN_NB - take no bias
// (another packed add per pair): round 4, counted in the ISA (plain epilogue: 2 packed ops + 1 convert per pair -> 0 + 1).
constexpr bool epi_no_bias(int mode) {
  return mode == EPI_DGELU || mode == EPI_DGELU8 || mode == EPI_PRES16 || mode == EPI_PLAIN_NB;
}
constexpr int EPI_UNSUPPORTED = -2;   // epi_mode_for: the request cannot be honoured by any epilogue

// VMEM stores one wave issues in gemm_epilogue<MODE, NJ> (its loads are consumed inside): callers that keep LDS-DMA
// in flight across the epilogue count them in their s_waitcnt vmcnt(N)
constexpr int epi_stores(int mode, int nj) {
  return (mode == EPI_PLAIN || mode == EPI_PLAIN_NB || mode == EPI_DGELU || mode == EPI_PRES16 || mode == EPI_ROPE || mode == EPI_DGELU8 ||
          mode == EPI_ROPE_IL) ? 2 * nj
         : (mode == EPI_RES32 || mode == EPI_GELU_PRE || mode == EPI_RES16 || mode == EPI_GELU_D8) ? 4 * nj : -1;
}

// which specialised mode (if any) matches a request
static inline int epi_mode_for(const clipk_gemm_args* a) {
  const bool c_f32 = a->c_dtype == CLIPK_F32, has_res = a->residual != nullptr, has_aux = a->dact_aux != nullptr;
  const bool has_pre = a->out_preact != nullptr;
  const bool aux8 = a->aux_dtype == CLIPK_U8;
  const bool has_bias = a->bias != nullptr;
  if (aux8) {                                            // 8-bit GELU' codes: the FFN pair only
    if (has_pre && (a->act != CLIPK_ACT_GELU || (a->ldp & 7))) return EPI_UNSUPPORTED;
    if (has_aux && (a->dact != CLIPK_ACT_GELU || (a->ldd & 7))) return EPI_UNSUPPORTED;
  }
  const long lim = 0x7fffffffL;                          // buffer-descriptor stores: byte extents must stay < 2 GiB
  if (a->rope_cos && a->rope_interleaved) {              // pairs are neighbours: no tiling rule beyond 8-column chunks
    const int hd = a->rope_hd;
    const bool ok = a->rope_sin && hd >= 8 && (hd & 7) == 0 && a->rope_L > 0 && a->rope_cols > 0 &&
                    a->rope_cols % hd == 0 && a->rope_cols <= a->N && a->rope_row0 >= 0 && !c_f32 && !has_res &&
                    !has_aux && !has_pre && a->act == CLIPK_ACT_NONE && a->drop_p <= 0.f && a->alpha == 1.0f &&
                    ((long)(a->M - 1) * a->ldc + a->N) * 2 <= lim;
    return ok ? EPI_ROPE_IL : EPI_UNSUPPORTED;
  }
  if (a->rope_cos) {                                     // rotation exists in its straight-line mode only
    const int hd = a->rope_hd;
    const bool ok = a->rope_sin && (hd == 16 || hd == 32 || hd == 64) && a->rope_L > 0 && a->rope_cols > 0 &&
                    a->rope_cols % hd == 0 && a->rope_cols <= a->N && a->rope_row0 >= 0 && !c_f32 && !has_res &&
                    !has_aux && !has_pre && a->act == CLIPK_ACT_NONE && a->drop_p <= 0.f && a->alpha == 1.0f &&
                    ((long)(a->M - 1) * a->ldc + a->N) * 2 <= lim;
    return ok ? EPI_ROPE : EPI_UNSUPPORTED;
  }
  if (a->drop_p > 0.f) return EPI_GENERIC;               // dropout lives in the run-time epilogue
  if (a->alpha != 1.0f) return EPI_GENERIC;              // the straight-line modes do not scale the product
  if (((long)(a->M - 1) * a->ldc + a->N) * (c_f32 ? 4 : 2) > lim) return EPI_GENERIC;
  if (has_pre && ((long)(a->M - 1) * a->ldp + a->N) * (aux8 ? 1 : 2) > lim) return EPI_GENERIC;
  if (a->act == CLIPK_ACT_NONE && !has_aux && !has_res && !has_pre && !c_f32) return has_bias ? EPI_PLAIN : EPI_PLAIN_NB;
  if (a->act == CLIPK_ACT_NONE && !has_aux && has_res && a->r_dtype == CLIPK_F32 && !has_pre && c_f32) return EPI_RES32;
  if (a->act == CLIPK_ACT_NONE && !has_aux && has_res && a->r_dtype == CLIPK_BF16 && !has_pre) {
    if (c_f32) return EPI_RES16;
    return has_bias ? EPI_GENERIC : EPI_PRES16;          // bf16 out + bf16 residual: the post-LN input gradients (no bias)
  }
  // (without a pre-activation output — frozen encoders keep nothing for a backward — the same mode runs with a
  // zero-length descriptor for u: the hardware drops those stores)
  if (a->act == CLIPK_ACT_GELU && !has_aux && !has_res && !c_f32) return (aux8 && has_pre) ? EPI_GELU_D8 : EPI_GELU_PRE;
  if (a->act == CLIPK_ACT_NONE && has_aux && a->dact == CLIPK_ACT_GELU && !has_res && !has_pre && !c_f32 && !has_bias)
    return aux8 ? EPI_DGELU8 : EPI_DGELU;
  return EPI_GENERIC;
}

constexpr int EPI_LD = 68;                              // f32 per staged row (16 rows x 64 cols per wave + pad)

// float offset of the 16-B chunk `chunk` (0..15) of slab row `row` (0..15).  SWZ = false: rows padded to EPI_LD
// floats (4352 B per wave).  SWZ = true: 64-float rows, chunk index XOR row (4096 B per wave, conflict-free for the
// column-wise accumulator writes and the row-wise reads alike) — for kernels that have no LDS to spare for the pad.
template <bool SWZ>
__device__ __forceinline__ int slab_off(int row, int chunk) {
  return SWZ ? row * 64 + ((chunk ^ row) << 2) : row * EPI_LD + (chunk << 2);
}

// this lane's 8 bias values (columns gn .. gn+7), zeros without a bias.  Separate from gemm_epilogue so that a kernel
// with LDS-DMA prefetches in flight can issue these loads BEFORE the prefetch (vmcnt is in-order: a load issued
// after the prefetch cannot be waited for without waiting for the prefetch as well).
// Two steps so that the loads can be issued well ahead (epi_issue_bias) of the point where they must have landed
// (epi_retire_bias).
__device__ __forceinline__ void epi_issue_bias(const EpiArgs& p, int gn, f32x4& b0, f32x4& b1) {
  b0 = f32x4{0.f, 0.f, 0.f, 0.f}; b1 = b0;
  if (p.bias && gn < p.N) {
    b0 = *reinterpret_cast<const f32x4*>(p.bias + gn);
    b1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
  }
}
__device__ __forceinline__ void epi_retire_bias(const f32x4& b0, const f32x4& b1, float (&bv)[8]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) { bv[c] = b0[c]; bv[4 + c] = b1[c]; }
  // retire the loads here, so that none is pending on any path into the slice loop
#pragma unroll
  for (int c = 0; c < 8; ++c) asm volatile("" ::"v"(bv[c]));
}
__device__ __forceinline__ void epi_load_bias(const EpiArgs& p, int gn, float (&bv)[8]) {
  f32x4 b0, b1;
  epi_issue_bias(p, gn, b0, b1);
  epi_retire_bias(b0, b1, bv);
}

// output store through a buffer descriptor; nt = streaming hint (the aux bits must be immediates, hence the branch)
__device__ __forceinline__ void epi_store(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned off, int nt) {
  if (nt) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 2);
  else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 0);
}

// eb: wave-private slab; mbase: first output row of this wave; gn: this lane's first column; bv: epi_load_bias
template <int MODE, int NJ, bool SWZ = false>
__device__ __forceinline__ void gemm_epilogue(const EpiArgs& p, f32x4 (&acc)[4][NJ], float* eb, int lane, int mbase,
                                              int gn, const float (&bv)[8]) {
  const int M = p.M, N = p.N;
  const float alpha = p.alpha;
  const int g = lane >> 4, li = lane & 15;
  const int ecol = (lane & 7) * 8;

  if constexpr (MODE != EPI_GENERIC) {
    const bool col_ok = gn < N;
    const long gnc = col_ok ? gn : 0;
    // stores go through buffer descriptors: an out-of-range lane gets an offset past num_records and the hardware
    // drops its store, so the slice loop has no exec-masked blocks at all (epi_mode_for keeps the extents < 2 GiB)
    constexpr unsigned OOB = 0x80000000u;
    const int c_elt = (MODE == EPI_RES32 || MODE == EPI_RES16) ? 4 : 2;
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)(((long)(M - 1) * p.ldc + N) * c_elt), 0x00020000);
    constexpr bool HAS_U = MODE == EPI_GELU_PRE || MODE == EPI_GELU_D8;
    const auto u_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (HAS_U && p.out_preact) ? (void*)p.out_preact : p.C, 0,
        (HAS_U && p.out_preact) ? (int)(((long)(M - 1) * p.ldp + N) * (MODE == EPI_GELU_D8 ? 1 : 2)) : 0, 0x00020000);
    constexpr int S = 2 * NJ;
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};       // residual (RoPE: cosines) of the current slice
    f32x4 r2 = {0.f, 0.f, 0.f, 0.f}, r3 = {0.f, 0.f, 0.f, 0.f};       // RoPE: sines of the current slice
    u32x4 ax = {0u, 0u, 0u, 0u};                                       // GELU' argument of the current slice
    // RoPE: the wave's 64 columns hold whole heads (tiles start at multiples of 64 columns, hd divides 64), so a
    // lane's partner columns (+/- hd/2) are in the same slab row; all 8 columns of a lane sit on one side of the head
    int rp_chunk = 0, rp_tcol = 0;
    bool rp_on = false, rp_lo = false;
    float bvp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (MODE == EPI_ROPE_IL) {
      rp_tcol = (gn % p.rope_hd) >> 1;                                   // first of this lane's four pairs
      rp_on = col_ok && gn < p.rope_cols;
    }
    if constexpr (MODE == EPI_ROPE) {
      const int half = p.rope_hd >> 1, d0 = ecol & (p.rope_hd - 1);
      rp_lo = d0 < half;
      rp_tcol = rp_lo ? d0 : d0 - half;
      rp_chunk = (rp_lo ? ecol + half : ecol - half) >> 2;
      rp_on = col_ok && gn < p.rope_cols;
      if (p.bias && rp_on) {                                           // the partner columns' bias
        const float* bp = p.bias + (rp_lo ? gn + half : gn - half);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { bvp[c] = b0[c]; bvp[4 + c] = b1[c]; }
      }
    }
    auto row_of = [&](int s) { return mbase + (s >> 1) * 16 + (s & 1) * 8 + (lane >> 3); };
    auto fetch = [&](int s, f32x4& a0, f32x4& a1, u32x4& b, f32x4& a2, f32x4& a3) {
      int gm = row_of(s);
      gm = gm < M ? gm : M - 1;
      if constexpr (MODE == EPI_ROPE) {
        const int pos = (gm + p.rope_row0) % p.rope_L;
        const long t = (long)pos * (p.rope_hd >> 1) + rp_tcol;
        a0 = *reinterpret_cast<const f32x4*>(p.rope_cos + t);
        a1 = *reinterpret_cast<const f32x4*>(p.rope_cos + t + 4);
        a2 = *reinterpret_cast<const f32x4*>(p.rope_sin + t);
        a3 = *reinterpret_cast<const f32x4*>(p.rope_sin + t + 4);
      }
      if constexpr (MODE == EPI_ROPE_IL) {
        const int pos = (gm + p.rope_row0) % p.rope_L;
        const long t = (long)pos * (p.rope_hd >> 1) + rp_tcol;
        a0 = *reinterpret_cast<const f32x4*>(p.rope_cos + t);
        a2 = *reinterpret_cast<const f32x4*>(p.rope_sin + t);
      }
      if constexpr (MODE == EPI_RES32) {
        const float* r = reinterpret_cast<const float*>(p.residual) + (long)gm * p.ldr + gnc;
        a0 = *reinterpret_cast<const f32x4*>(r);
        a1 = *reinterpret_cast<const f32x4*>(r + 4);
      }
      if constexpr (MODE == EPI_DGELU) b = *reinterpret_cast<const u32x4*>(p.dact_aux + (long)gm * p.ldd + gnc);
      if constexpr (MODE == EPI_DGELU8) {                    // eight 1-byte codes
        const u32x2 q = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.dact_aux) + (long)gm * p.ldd + gnc);
        b = u32x4{q[0], q[1], 0u, 0u};
      }
      if constexpr (MODE == EPI_RES16 || MODE == EPI_PRES16)
        b = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.residual) + (long)gm * p.ldr + gnc);
    };
    // Operand prefetch ring.  vmcnt retires in issue order and counts stores too, so waiting for the operands of
    // slice s also waits for every store issued before their load: with the load one slice ahead that is the store of
    // slice s - 2, i.e. each wave moves two slices per store round trip.  The distance therefore grows as the
    // accumulators die (slice pair j frees 16 registers): 1, 2, 3, then RING - 1 slices ahead.
#ifndef CLIPK_EPI_RING
#define CLIPK_EPI_RING 4
#endif
    constexpr int RING = CLIPK_EPI_RING;
    f32x4 R0[RING], R1[RING], R2[RING], R3[RING];
    u32x4 AX[RING];
#pragma unroll
    for (int t = 0; t < RING; ++t) { R0[t] = r0; R1[t] = r1; R2[t] = r2; R3[t] = r3; AX[t] = ax; }
    auto hi = [](int s) {                                    // last slice whose operands are requested by slice s
      if (s < 0) return -1;
      int a = 1 + (s >> 1); a = a > RING - 1 ? RING - 1 : a;
      const int t = s + a;
      return t < S - 1 ? t : S - 1;
    };
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int j = s >> 1;
      if ((s & 1) == 0) {                                  // (alpha == 1 in every specialised mode: epi_mode_for)
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(eb + slab_off<SWZ>(li, i * 4 + g)) = acc[i][j];
      }
#pragma unroll
      for (int t = (s == 0 ? 0 : hi(s - 1) + 1); t <= hi(s); ++t)
        fetch(t, R0[t % RING], R1[t % RING], AX[t % RING], R2[t % RING], R3[t % RING]);
      r0 = R0[s % RING]; r1 = R1[s % RING]; r2 = R2[s % RING]; r3 = R3[s % RING]; ax = AX[s % RING];
      const int row = (s & 1) * 8 + (lane >> 3);
      const int gm = row_of(s);
      float v[8];
      {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(eb + slab_off<SWZ>(row, ecol >> 2));
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(eb + slab_off<SWZ>(row, (ecol >> 2) + 1));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = epi_no_bias(MODE) ? v0[c] : v0[c] + bv[c];
          v[4 + c] = epi_no_bias(MODE) ? v1[c] : v1[c] + bv[4 + c];
        }
      }
      const bool ok = col_ok && gm < M;
      if constexpr (MODE == EPI_ROPE) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(eb + slab_off<SWZ>(row, rp_chunk));
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(eb + slab_off<SWZ>(row, rp_chunk + 1));
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
          float y[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int k = c + e;
            const float vp = (k < 4 ? w0[k] : w1[k - 4]) + bvp[k];
            const float cs = rp_on ? (k < 4 ? r0[k] : r1[k - 4]) : 1.0f;
            const float sn = rp_on ? (k < 4 ? r2[k] : r3[k - 4]) : 0.0f;
            // explicit mul + fma, as rope_regs (attention.hip): x1' = fma(x1, c, -(x2 s)), x2' = fma(x2, c, x1 s)
            y[e] = fmaf(v[k], cs, rp_lo ? -(vp * sn) : vp * sn);
          }
          o[c >> 1] = pack_bf16x2(y[0], y[1]);
        }
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
        epi_store(o, c_rsrc, off, p.nt);
      } else if constexpr (MODE == EPI_ROPE_IL) {
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float cs = rp_on ? r0[c] : 1.0f, sn = rp_on ? r2[c] : 0.0f;
          // x1' = fma(x1, c, -(x2 s)), x2' = fma(x2, c, x1 s): the arithmetic of rope_regs (attention.hip), on the f32
          // value before its one bf16 rounding
          const float y1 = fmaf(v[2 * c], cs, -(v[2 * c + 1] * sn));
          const float y2 = fmaf(v[2 * c + 1], cs, v[2 * c] * sn);
          o[c] = pack_bf16x2(y1, y2);
        }
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
        epi_store(o, c_rsrc, off, p.nt);
      } else if constexpr (MODE == EPI_PLAIN || MODE == EPI_PLAIN_NB) {
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
        epi_store(o, c_rsrc, off, p.nt);
      } else if constexpr (MODE == EPI_RES32) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[c] += r0[c]; v[4 + c] += r1[c]; }
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 4) : OOB;
        epi_store(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]),
                                                     __float_as_uint(v[3])}, c_rsrc, off, p.nt);
        epi_store(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]),
                                                     __float_as_uint(v[7])}, c_rsrc, off + 16, p.nt);
      } else if constexpr (MODE == EPI_RES16 || MODE == EPI_PRES16) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[2 * c] += bf16_to_f32((unsigned short)(ax[c] & 0xffffu));
          v[2 * c + 1] += bf16_to_f32((unsigned short)(ax[c] >> 16));
        }
        if constexpr (MODE == EPI_RES16) {
          const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 4) : OOB;
          epi_store(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                    c_rsrc, off, p.nt);
          epi_store(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])},
                    c_rsrc, off + 16, p.nt);
        } else {
          u32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
          const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
          epi_store(o, c_rsrc, off, p.nt);
        }
      } else if constexpr (MODE == EPI_GELU_PRE) {
        u32x4 u, o;
        unsigned qq[2];
#pragma unroll
        for (int c = 0; c < 4; ++c) u[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
        gelu_erf8<false>(v, o, qq);                                            // packed-f32 pipe, stage by stage
        const unsigned offu = (ok && p.out_preact) ? (unsigned)(((long)gm * p.ldp + gn) * 2) : OOB;
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
        epi_store(u, u_rsrc, offu, p.nt);
        epi_store(o, c_rsrc, off, p.nt);
      } else if constexpr (MODE == EPI_GELU_D8) {
        u32x4 o;
        unsigned qq[2];
        gelu_erf8<true>(v, o, qq);
        const u32x2 q = {qq[0], qq[1]};
        const unsigned offu = (ok && p.out_preact) ? (unsigned)((long)gm * p.ldp + gn) : OOB;
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
        if (p.nt) __builtin_amdgcn_raw_buffer_store_b64(q, u_rsrc, offu, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b64(q, u_rsrc, offu, 0, 0);
        epi_store(o, c_rsrc, off, p.nt);
      } else if constexpr (MODE == EPI_DGELU8) {
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned w = ax[c >> 1] >> (16 * (c & 1));
          v[2 * c] *= gelu_grad_decode(w & 0xffu);
          v[2 * c + 1] *= gelu_grad_decode((w >> 8) & 0xffu);
          o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
        }
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
        epi_store(o, c_rsrc, off, p.nt);
      } else {  // EPI_DGELU
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[2 * c] *= gelu_erf_grad(bf16_to_f32((unsigned short)(ax[c] & 0xffffu)));
          v[2 * c + 1] *= gelu_erf_grad(bf16_to_f32((unsigned short)(ax[c] >> 16)));
          o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
        }
        const unsigned off = ok ? (unsigned)(((long)gm * p.ldc + gn) * 2) : OOB;
        epi_store(o, c_rsrc, off, p.nt);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(eb + slab_off<SWZ>(li, i * 4 + g)) = acc[i][j] * alpha;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int row = half * 8 + (lane >> 3);
        const int gm = mbase + j * 16 + row;
        float v[8];
        {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(eb + slab_off<SWZ>(row, ecol >> 2));
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(eb + slab_off<SWZ>(row, (ecol >> 2) + 1));
#pragma unroll
          for (int c = 0; c < 4; ++c) { v[c] = v0[c] + bv[c]; v[4 + c] = v1[c] + bv[4 + c]; }
        }
        if (gm < M && gn < N) {
          if (p.out_preact) {
            if (p.aux_u8) {
              float d[8];
#pragma unroll
              for (int c = 0; c < 8; ++c) d[c] = gelu_erf_grad(v[c]);
              *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.out_preact) + (long)gm * p.ldp + gn) = gelu_grad_pack8(d);
            } else {
              u32x4 o;
#pragma unroll
              for (int c = 0; c < 4; ++c) o[c] = pack_bf16x2(v[2 * c], v[2 * c + 1]);
              *reinterpret_cast<u32x4*>(p.out_preact + (long)gm * p.ldp + gn) = o;
            }
          }
          if (p.act != CLIPK_ACT_NONE) {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = act_apply(v[c], p.act);
          }
          if (p.drop_thr) {                                   // nn.Dropout on this tensor: index = m * N + n
            const unsigned long long base = (unsigned long long)gm * (unsigned)N + (unsigned)gn;
            const unsigned dseed = drop_seed_eff(p.drop_seed, p.drop_epoch);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] *= drop_mul(dseed, base + c, p.drop_thr, p.drop_scale);
          }
          if (p.dact_aux && p.aux_u8) {
            const u32x2 a = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.dact_aux) + (long)gm * p.ldd + gn);
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] *= gelu_grad_decode((a[c >> 2] >> (8 * (c & 3))) & 0xffu);
          } else if (p.dact_aux) {
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
              const f32x4 q0 = *reinterpret_cast<const f32x4*>(r);
              const f32x4 q1 = *reinterpret_cast<const f32x4*>(r + 4);
#pragma unroll
              for (int c = 0; c < 4; ++c) { v[c] += q0[c]; v[4 + c] += q1[c]; }
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
}

static inline EpiArgs epi_args_from(const clipk_gemm_args* a) {
  EpiArgs e;
  e.C = a->C; e.ldc = a->ldc; e.c_f32 = (a->c_dtype == CLIPK_F32);
  e.M = a->M; e.N = a->N;
  e.bias = a->bias; e.act = a->act;
  e.out_preact = (unsigned short*)a->out_preact; e.ldp = a->ldp;
  e.dact_aux = (const unsigned short*)a->dact_aux; e.ldd = a->ldd; e.dact = a->dact;
  e.residual = a->residual; e.ldr = a->ldr; e.r_f32 = (a->r_dtype == CLIPK_F32);
  e.alpha = a->alpha;
  e.aux_u8 = (a->aux_dtype == CLIPK_U8);
  // Streaming (non-temporal) output stores, option epi_nt = 1.  In the kernel microbenchmark they are worth 7 % on the
  // hot shapes (the output stream stops evicting operand panels from L2: esm qkv 256 -> 192 us; f32 residual outputs
  // get slower), in the training step nothing (96.8 vs 96.6 ms): there the consumer of the output runs next and finds
  // less of it in the Infinity Cache.  Off by default.
  e.nt = clipk_opt_get(OPT_EPI_NT);
  e.rope_cos = a->rope_cos; e.rope_sin = a->rope_sin; e.rope_L = a->rope_L; e.rope_hd = a->rope_hd;
  e.rope_cols = a->rope_cols; e.rope_row0 = a->rope_row0;
  e.drop_thr = 0; e.drop_seed = a->drop_seed; e.drop_scale = 1.f; e.drop_epoch = clipk_drop_epoch();
  if (a->drop_p > 0.f && a->drop_p < 1.f) {
    const double t = (double)a->drop_p * 4294967296.0;
    e.drop_thr = t < 1.0 ? 1u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
    e.drop_scale = 1.0f / (1.0f - a->drop_p);
  }
  return e;
}
