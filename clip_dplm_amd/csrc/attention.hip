// attention.hip — flash-style multi-head self-attention forward / backward for the two encoders.
//
// Reference arithmetic replaced: nn.MultiheadAttention inside nn.TransformerEncoderLayer
// (current/rna_clip_codes.ipynb:1915; key-padding mask -> -inf) and the third-party EsmSelfAttention
// (transformers modeling_esm.py:362-384: q *= hd^-1/2, rotate-half RoPE on q,k :48-52,74-79, softmax :306-314).
//
// gfx950 design:
//   * the L x L score matrix never leaves registers; 16x16x32 bf16 MFMA everywhere, f32 softmax (exp2, scale folded);
//   * "swapped" products: S^T = K·Q^T puts keys on accumulator rows and queries on lanes, so the online-softmax
//     row statistics are in-register reductions + two lane exchanges, and the P^T accumulator registers ARE the B
//     operand of O^T += V^T·P^T (no LDS round trip for P);
//   * V^T / K^T / Q^T / dO^T operands come from row-major LDS tiles through ds_read_b64_tr_b16 (CDNA4 transposed
//     read); tiles use a (2*DP+32)-byte row stride that tools/lds_conflicts.py shows conflict-free for both the b128
//     row reads and the transposed reads; head dims that are not a multiple of 32 (ESM-2-35M: 24) are zero-padded
//     to DP in LDS only;
//   * "row-owner" staging: one thread owns one token row of one operand (its D bf16 = D/8 16-byte chunks) in
//     registers, applies RoPE there (f32 math, static indices — the head dim is a template parameter on the RoPE
//     path) and writes the finished row to LDS: q/k are read once from HBM, no rotated copy is ever written and
//     no in-LDS rotation pass or extra barrier exists.  The same registers prefetch the next key (or query) block
//     while the MFMAs of the current one run;
//   * rows that need no rotation in registers (no RoPE, or q / k rotated once by clipk_rope_qk) are staged chunk per
//     lane instead: consecutive lanes on consecutive 16-byte chunks, whole 128-byte lines per wave instruction;
//   * backward, general shape = a dQ kernel (one workgroup per 128 queries, sweeping keys; it also produces
//     delta = rowsum(dO*O) from the rows it already holds) and a dK/dV kernel (one workgroup per 128 or 64 keys,
//     sweeping queries): 7 MFMA products instead of 5, but no float atomics and bitwise-reproducible gradients;
//   * backward, short heads (hd <= 32, 128 < L <= 256, the ESM-2 8M / 35M / 150M encoders) = ONE kernel, a whole
//     (batch, head) per persistent workgroup: 5 products, one softmax pass, dS transposed through LDS, dQ summed
//     over the key-owning waves in a rotated fixed order (attn_bwd_fused32_kernel below) - still no atomics.
#include "common.h"
#include <math.h>
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct AP {
  const unsigned short* qkv; const uint8_t* key_mask; const float* cosT; const float* sinT;
  unsigned short* out; float* lse;
  const unsigned short* dout; float* delta; unsigned short* dqkv;
  int B, L, H, D;
  float scale;
  int row_stores;   // whole-head kernels: 1 = gradient / rotated rows leave four lanes to a row from LDS, 0 = one thread per row
  int pre_rot;      // backward: q / k in `qkv` are already rotated (clipk_rope_qk): stage them as they are, the
                    // gradients still leave through RoPE^T
  // packed variable-length batches (current/rna_clip_codes.ipynb:1726-1736: sequences of 30..2542 tokens): sequence b
  // occupies token rows [cu[b], cu[b+1]) of the packed [T, ...] tensors, L is the LONGEST sequence (grid sizing only),
  // the per-row statistics (lse, delta) are [H][T].  cu == nullptr: the padded [B, L] layout.
  const int* cu; int T;
  // dropout on the attention probabilities (nn.MultiheadAttention(dropout=p) inside nn.TransformerEncoderLayer,
  // current/rna_clip_codes.ipynb:1915): P~ = P * keep / (1 - p) feeds P·V; the softmax normaliser uses P.  drop_thr = 0:
  // off.  Element index = ((token row of the query) * H + h) * L + key (L = p.L, the padded / longest length).
  unsigned drop_thr, drop_seed; float drop_scale;
  const unsigned* drop_epoch;       // common.h drop_seed_eff: read once per kernel (dseed), nullptr outside a captured step
};
__device__ __forceinline__ float attn_drop(const AP& p, unsigned dseed, long qrow, int h, int key) {
  return drop_mul(dseed, ((unsigned long long)qrow * p.H + h) * (unsigned long long)p.L + key, p.drop_thr, p.drop_scale);
}

// first token row and length of sequence b (L comes in as p.L)
__device__ __forceinline__ long seq_rows(const AP& p, int b, int& L) {
  if (p.cu) {
    const int s = p.cu[b];
    L = p.cu[b + 1] - s;
    return s;
  }
  return (long)b * L;
}
// index of row r of (sequence b, head h) in the lse / delta arrays
__device__ __forceinline__ long stat_at(const AP& p, int b, int h, int H, int L, long row0, int r) {
  return p.cu ? (long)h * p.T + row0 + r : ((long)b * H + h) * L + r;
}

constexpr float LOG2E = 1.4426950408889634f;

// v_exp_f32 directly: every argument here is <= 0 (or -inf), so the denormal-range fix-up of exp2f() is dead weight
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Work-item order: the G = (blocks per sequence) x H workgroups of one batch element read the same token rows
// (each head only D*2 bytes of a 3*H*D*2-byte row), so keep them on ONE XCD's L2: blocks b and b+8 share an XCD
// (round-robin dispatch), hence XCD x gets batch elements x, x+8, ...  Falls back to the plain order when B % 8 != 0.
__device__ __forceinline__ void work_item_at(int w, int nblk, int H, int B, int& blk, int& h, int& b) {
  const int G = nblk * H;
  if ((B & 7) == 0) {
    const int xcd = w & 7, slot = w >> 3;
    w = ((slot / G) * 8 + xcd) * G + (slot % G);
  }
  b = w / G;
  const int r = w - b * G;
  h = r / nblk;
  blk = r - h * nblk;
}
__device__ __forceinline__ void work_item(int nblk, int H, int B, int& blk, int& h, int& b) {
  work_item_at(blockIdx.x, nblk, H, B, blk, h, b);
}

template <int DP> struct Geo {
  static constexpr int RS = DP * 2 + 32;     // LDS row stride in bytes
  static constexpr int KS = DP / 32;         // contraction steps over the head dim
  static constexpr int DT = DP / 16;         // 16-wide d tiles
  static constexpr int NCH = DP / 8;         // 16-byte chunks per (padded) row
  static constexpr int KVB = (DP <= 96) ? 128 : 64;   // keys (or queries) per staged block
  // keys per dK/dV workgroup.  At DP = 96 a wave owning 32 keys needs 96 accumulator + 48 fragment + 48 prefetch
  // registers and hipcc kept part of the accumulators in scratch INSIDE the sweep, each reload waiting (vmcnt is in
  // order) for the whole query-block prefetch; with 16 keys per wave everything stays in registers
  static constexpr int KPB = (DP == 96) ? 64 : KVB;
  // minimum resident workgroups per CU asked of the register allocator.  Left alone hipcc spends 296 / 308 VGPRs on
  // the D = 96 backward kernels (one wave per SIMD, every LDS / HBM latency exposed); capped at 256 they spill 32 /
  // 72 registers and still run 1.3x faster (1330 -> 1010 us), D = 24: 915 -> 815 us backward, 380 -> 330 us forward
  // (Tried for the backward kernels and dropped: the element-wise part two scores at a time on the packed-f32 pipe.
  // D = 24: 816 -> 837 us, the pairs cost registers and moves; D = 96 dK/dV: 72 -> 147 spilled VGPRs.  The forward
  // kernel's packed softmax stays: 329 -> 315 us at D = 24, 323 -> 307 us at D = 96.)
  static constexpr int WG_FWD = (DP <= 32) ? 4 : (DP <= 96 ? 2 : 1);
  static constexpr int WG_BWD = (DP <= 32) ? 3 : (DP <= 96 ? 2 : 1);
};

template <int NCH> struct RowRegs { u32x4 c[NCH]; };

template <int NCH>
__device__ __forceinline__ void load_row(RowRegs<NCH>& r, const unsigned short* rowptr, int cpr) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    r.c[i] = u32x4{0u, 0u, 0u, 0u};
    if (i < cpr) r.c[i] = *reinterpret_cast<const u32x4*>(rowptr + 8 * i);
  }
}
// writes all NCH chunks: the pad columns D..DP-1 become zero in LDS
template <int NCH>
__device__ __forceinline__ void store_row(const RowRegs<NCH>& r, char* ldsrow) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) *reinterpret_cast<u32x4*>(ldsrow + 16 * i) = r.c[i];
}

// rotate-half RoPE on a register-resident row; D is compile time so every index is static.  dir=+1: forward.
template <int D, int NCH>
__device__ __forceinline__ void rope_regs(RowRegs<NCH>& r, const float* cosr, const float* sinr) {
  constexpr int H = D / 2;
#pragma unroll
  for (int j = 0; j < H; j += 2) {
    const int ja = j, jb = j + H;                         // element indices of the two words (H is even)
    unsigned int wa = r.c[ja >> 3][(ja & 7) >> 1], wb = r.c[jb >> 3][(jb & 7) >> 1];
    const float c0 = cosr[j], c1 = cosr[j + 1], s0 = sinr[j], s1 = sinr[j + 1];
    const float x1a = bf16_to_f32(wa & 0xffffu), x1b = bf16_to_f32(wa >> 16);
    const float x2a = bf16_to_f32(wb & 0xffffu), x2b = bf16_to_f32(wb >> 16);
    // explicit mul + fma: the same rounding in every kernel that rotates (contraction left to the compiler differs
    // from kernel to kernel by an ulp, and pre-rotated and staged-rotated paths must agree bit for bit)
    wa = pack_bf16x2(fmaf(x1a, c0, -(x2a * s0)), fmaf(x1b, c1, -(x2b * s1)));
    wb = pack_bf16x2(fmaf(x2a, c0, x1a * s0), fmaf(x2b, c1, x1b * s1));
    r.c[ja >> 3][(ja & 7) >> 1] = wa;
    r.c[jb >> 3][(jb & 7) >> 1] = wb;
  }
}

__device__ __forceinline__ bf16x8 row_frag(const char* tile, int RS, int row, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(tile + row * RS + ks * 64 + (lane >> 4) * 16);
}
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int RS, int row0, int colbyte) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(tile + row0 * RS + colbyte));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(tile + (row0 + 16) * RS + colbyte));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}
__device__ __forceinline__ bf16x8 pack_acc_pair(const f32x4 a, const f32x4 b) {
  const u32x4 w = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ float group_max(float v) {   // across the 4 lane groups sharing lane&15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// Row-owner staging of a K/V (or Q/dO) block pair.  With BLK rows per operand: threads [0,BLK) own operand-A rows,
// threads [BLK,2*BLK) own operand-B rows (BLK = 128 uses all 256 threads, BLK = 64 the first 128).
template <int DP, int DR, int BLK>
struct PairStager {
  static constexpr int NCH = Geo<DP>::NCH;
  RowRegs<NCH> r;
  int row, pos;
  bool is_a, active;

  __device__ __forceinline__ void init(int tid) {
    active = tid < 2 * BLK;
    is_a = tid < BLK;
    row = tid & (BLK - 1);
  }
  // a_base / b_base: row pointers base (token 0 of this batch element, head offset applied); strides in elements
  __device__ __forceinline__ void load(const unsigned short* a_base, long a_stride, const unsigned short* b_base,
                                       long b_stride, int pos0, int L, int cpr) {
    if (!active) return;
    pos = pos0 + row;
    const int pc = pos < L ? pos : L - 1;
    const unsigned short* rp = is_a ? a_base + (long)pc * a_stride : b_base + (long)pc * b_stride;
    load_row<NCH>(r, rp, cpr);
  }
  // rope_a: rotate operand-A rows (K or Q) before they reach LDS
  __device__ __forceinline__ void store(char* a_tile, char* b_tile, int RS, const AP& p, int L, bool rope_a) {
    if (!active) return;
    if (DR > 0) {
      if (rope_a && is_a && !p.pre_rot) {
        const int pc = pos < L ? pos : L - 1;
        rope_regs<(DR > 0 ? DR : 2), NCH>(r, p.cosT + (long)pc * (DR / 2), p.sinT + (long)pc * (DR / 2));
      }
    }
    store_row<NCH>(r, (is_a ? a_tile : b_tile) + row * RS);
  }
};

// a zero the optimiser cannot see through: values derived from `tid + opaque_zero()` inside a loop are recomputed
// there instead of being hoisted out and kept alive (and then spilled) across the register-tight sweep
__device__ __forceinline__ int opaque_zero() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

// Chunk-per-lane staging of the same block pair for the heads that need no rotation (DR == 0): consecutive lanes
// take consecutive 16-byte chunks of a row, so one wave instruction reads whole 128-byte lines.  The row-owner form
// above reads 16 bytes from each of 64 different lines per instruction; with 4 waves x 2 operands in flight the lines
// do not survive in the 32 KiB L1 until the row's next chunk is asked for, and every chunk pulls a full line from
// L2 again (8x the bytes through the L1 fill path, which is what bounded the hd = 96 kernels).
template <int DP, int BLK>
struct ChunkStager {
  static constexpr int NCH = Geo<DP>::NCH;
  static constexpr int PER_OP = BLK * NCH / 256;           // tasks per thread and operand (BLK * NCH % 256 == 0)
  static_assert((BLK * NCH) % 256 == 0, "chunk tasks must tile the workgroup");
  u32x4 c[2 * PER_OP];
  int tid_, cpr_;

  __device__ __forceinline__ void init(int tid) { tid_ = tid; }
  // 32-bit element offsets from the two (uniform) bases, recomputed at every call from an opaque copy of the thread
  // id: left to itself the compiler hoists the 2 * PER_OP 64-bit addresses out of the key / query loop and spills
  __device__ __forceinline__ void load(const unsigned short* a_base, long a_stride, const unsigned short* b_base,
                                       long b_stride, int pos0, int L, int cpr) {
    cpr_ = cpr;
    const int t = tid_ + opaque_zero();
#pragma unroll
    for (int it = 0; it < 2 * PER_OP; ++it) {
      const int idl = (it % PER_OP) * 256 + t;
      const int row = idl / NCH, ch = idl - row * NCH;
      int pos = pos0 + row; pos = pos < L ? pos : L - 1;
      const int chc = ch < cpr ? ch : cpr - 1;             // pad chunks re-read the last real one (dropped in store):
      const unsigned int off = (unsigned int)pos * (unsigned int)((it < PER_OP) ? a_stride : b_stride) + 8u * chc;
      c[it] = *reinterpret_cast<const u32x4*>(((it < PER_OP) ? a_base : b_base) + off);   // no branch around a load
    }
  }
  __device__ __forceinline__ void store(char* a_tile, char* b_tile, int RS, const AP&, int, bool) {
    const int t = tid_ + opaque_zero();
#pragma unroll
    for (int it = 0; it < 2 * PER_OP; ++it) {
      const int idl = (it % PER_OP) * 256 + t;
      const int row = idl / NCH, ch = idl - row * NCH;
      const u32x4 z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(((it < PER_OP) ? a_tile : b_tile) + row * RS + ch * 16) = ch < cpr_ ? c[it] : z;
    }
  }
};
// one operand, ROWS rows (the query-side prologues)
template <int DP, int ROWS>
struct ChunkRows {
  static constexpr int NCH = Geo<DP>::NCH;
  static constexpr int NT = ROWS * NCH / 256;
  static_assert((ROWS * NCH) % 256 == 0, "chunk tasks must tile the workgroup");
  u32x4 c[NT];

  __device__ __forceinline__ void load(const unsigned short* base, long stride, int pos0, int L, int cpr, int tid) {
    const int t = tid + opaque_zero();
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int idl = it * 256 + t;
      const int row = idl / NCH, ch = idl - row * NCH;
      int pos = pos0 + row; pos = pos < L ? pos : L - 1;
      c[it] = *reinterpret_cast<const u32x4*>(base + ((unsigned int)pos * (unsigned int)stride + 8u * (ch < cpr ? ch : cpr - 1)));
    }
  }
  __device__ __forceinline__ void store(char* tile, int RS, int cpr, int tid) const {
    const int t = tid + opaque_zero();
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int idl = it * 256 + t;
      const int row = idl / NCH, ch = idl - row * NCH;
      const u32x4 z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(tile + row * RS + ch * 16) = ch < cpr ? c[it] : z;
    }
  }
};
template <int DP, int DR, int BLK> struct StagerFor { typedef PairStager<DP, DR, BLK> type; };
template <int DP, int BLK> struct StagerFor<DP, 0, BLK> { typedef ChunkStager<DP, BLK> type; };

// =================================================================================================
// forward: one workgroup = 128 queries of one (batch, head); 4 waves x 32 queries; KVB-key staged blocks
// =================================================================================================
// DROP: dropout on the attention probabilities compiled in (a separate instantiation: the p = 0 kernels carry no
// trace of it)
template <int DP, int DR, int DX, bool DROP = false>
__global__ __launch_bounds__(256, Geo<DP>::WG_FWD) void attn_fwd_kernel(const AP p) {
  constexpr int RS = Geo<DP>::RS, KS = Geo<DP>::KS, DT = Geo<DP>::DT, NCH = Geo<DP>::NCH, KVB = Geo<DP>::KVB;
  constexpr int NSUB = KVB / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned dseed = DROP ? drop_seed_eff(p.drop_seed, p.drop_epoch) : 0u;   // one scalar load per kernel
  char* ktile = smem;
  char* vtile = smem + KVB * RS;
  unsigned char* mask_l = reinterpret_cast<unsigned char*>(smem + 2 * KVB * RS);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  // the head dim as a compile-time constant whenever the dispatcher knows it (DX): the `dt < dtv` guards of the
  // d-tile loops fold away - as run-time branches they pushed the dK / dV accumulators through scratch memory
  const int D = DX > 0 ? DX : p.D, H = p.H;
  int L = p.L;                                          // grid: the longest sequence; below: this sequence
  int qb, h, b;
  work_item((L + 127) / 128, H, p.B, qb, h, b);
  const long row0 = seq_rows(p, b, L);
  if (qb * 128 >= L) return;                              // packed batches: query block beyond this sequence
  const int cpr = D >> 3;
  const long tokstride = 3L * H * D;
  const unsigned short* qbase = p.qkv + row0 * tokstride + (long)h * D;
  const unsigned short* kbase = qbase + (long)H * D;
  const unsigned short* vbase = qbase + 2L * H * D;
  const int q0 = qb * 128;
  const int dtv = (D + 15) >> 4;
  const float c2 = p.scale * LOG2E;

  // ---- issue every first-use load up front: the Q rows and K/V block 0.  The 128 Q rows borrow the [K|V] region
  // before K/V land there
  typename StagerFor<DP, DR, KVB>::type kv;
  kv.init(tid);
  if constexpr (DR == 0) {
    ChunkRows<DP, 128> cq;
    cq.load(qbase, tokstride, q0, L, cpr, tid);
    kv.load(kbase, tokstride, vbase, tokstride, 0, L, cpr);
    cq.store(smem, RS, cpr, tid);
  } else {
    RowRegs<NCH> rq;                                       // row owner: RoPE in registers, then to LDS
    int pq = q0 + tid; pq = pq < L ? pq : L - 1;
    if (tid < 128) load_row<NCH>(rq, qbase + (long)pq * tokstride, cpr);
    kv.load(kbase, tokstride, vbase, tokstride, 0, L, cpr);
    if (tid < 128) {
      if (!p.pre_rot) rope_regs<(DR > 0 ? DR : 2), NCH>(rq, p.cosT + (long)pq * (DR / 2), p.sinT + (long)pq * (DR / 2));
      store_row<NCH>(rq, smem + tid * RS);
    }
  }
  __syncthreads();
  bf16x8 qf[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qt][ks] = row_frag(smem, RS, wid * 32 + qt * 16 + li, ks, lane);
  __syncthreads();
  kv.store(ktile, vtile, RS, p, L, true);
  if (tid < KVB) mask_l[tid] = (tid < L && (!p.key_mask || p.key_mask[row0 + tid])) ? 1 : 0;
  __syncthreads();

  f32x4 o[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { o[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[dt][1] = o[dt][0]; }
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  const int trow = 4 * g + (li >> 2), tcolb = 8 * (li & 3);

  const int nkb = (L + KVB - 1) / KVB;
  for (int kb = 0; kb < nkb; ++kb) {
    const bool more = (kb + 1) < nkb;
    if (more) kv.load(kbase, tokstride, vbase, tokstride, (kb + 1) * KVB, L, cpr);   // in flight during the MFMAs
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (kb * KVB + sub * 64 >= L) break;                       // wave-uniform: nothing valid in this sub-block
      const char* kt_ = ktile + sub * 64 * RS;
      const char* vt_ = vtile + sub * 64 * RS;
      f32x4 s[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) { s[kt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[kt][1] = s[kt][0]; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const bf16x8 kf = row_frag(kt_, RS, kt * 16 + li, ks, lane);
          s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[kt][0], 0, 0, 0);
          s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[kt][1], 0, 0, 0);
        }
      if (p.key_mask != nullptr || kb * KVB + sub * 64 + 64 > L) {     // wave-uniform: full, unmasked blocks skip this
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const unsigned int mk = *reinterpret_cast<const unsigned int*>(mask_l + sub * 64 + kt * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!((mk >> (8 * r)) & 0xffu)) { s[kt][0][r] = -INFINITY; s[kt][1][r] = -INFINITY; }
        }
      }
      bf16x8 pb[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
        mx = group_max(mx);
        const float m_new = fmaxf(m_run[qt], mx);
        const bool dead = (m_new == -INFINITY);
        const float alpha = dead ? 1.f : fast_exp2((m_run[qt] - m_new) * c2);
        // p = 2^(s c2 - m c2) as ONE fma per element; a dead row (everything masked so far) gets -inf as the
        // addend, and a masked score is -inf itself: -inf + -inf = -inf, 2^-inf = 0, no select per element.
        // Two elements per instruction on the packed-f32 pipe for the fma and the row sum.
        const f32x2 mc = splat2(dead ? -INFINITY : -m_new * c2);
        const f32x2 cc = splat2(c2);
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 a = __builtin_elementwise_fma(f32x2{s[kt][qt][r], s[kt][qt][r + 1]}, cc, mc);
            const f32x2 pv = {fast_exp2(a[0]), fast_exp2(a[1])};
            s[kt][qt][r] = pv[0]; s[kt][qt][r + 1] = pv[1];
            ls2 += pv;
          }
        const float ls = ls2[0] + ls2[1];
        l_run[qt] = l_run[qt] * alpha + ls;
        m_run[qt] = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][qt] *= alpha;
        if constexpr (DROP) {                                      // the normaliser above saw P itself
          const long qrow = row0 + q0 + wid * 32 + qt * 16 + li;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              s[kt][qt][r] *= attn_drop(p, dseed, qrow, h, kb * KVB + sub * 64 + kt * 16 + 4 * g + r);
        }
        pb[qt][0] = pack_acc_pair(s[0][qt], s[1][qt]);
        pb[qt][1] = pack_acc_pair(s[2][qt], s[3][qt]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          if (dt < dtv) {
            const bf16x8 vf = tr_frag(vt_, RS, 32 * s2 + trow, dt * 32 + tcolb);
            o[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[0][s2], o[dt][0], 0, 0, 0);
            o[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[1][s2], o[dt][1], 0, 0, 0);
          }
    }
    if (more) {
      __syncthreads();                                           // everyone done reading this block
      kv.store(ktile, vtile, RS, p, L, true);
      if (tid < KVB) {
        const int pos = (kb + 1) * KVB + tid;
        mask_l[tid] = (pos < L && (!p.key_mask || p.key_mask[row0 + pos])) ? 1 : 0;
      }
      __syncthreads();
    }
  }

#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float lt = group_sum(l_run[qt]);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    const int q = q0 + wid * 32 + qt * 16 + li;
    if (q < L) {
      if (g == 0) p.lse[stat_at(p, b, h, H, L, row0, q)] = lt > 0.f ? m_run[qt] * p.scale + logf(lt) : -INFINITY;
      unsigned short* orow = p.out + (row0 + q) * ((long)H * D) + (long)h * D;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + 4 * g;
        if (d < D) {
          u32x2 w;
          w[0] = pack_bf16x2(o[dt][qt][0] * inv, o[dt][qt][1] * inv);
          w[1] = pack_bf16x2(o[dt][qt][2] * inv, o[dt][qt][3] * inv);
          *reinterpret_cast<u32x2*>(orow + d) = w;
        }
      }
    }
  }
}

// write an f32 [rows][DP+4] LDS image (gradient w.r.t. rotated q/k) as bf16 rows of dqkv, applying the
// RoPE transpose on the way when ROPE
// il: the head dim is in pair-interleaved order (common.h il_src; the rows were rotated by clipk_gemm_nt's interleaved
// epilogue): the partner of column d is d ^ 1 and its angle index d >> 1
template <bool ROPE>
__device__ __forceinline__ void store_grad_rows(const float* img, int ILD, unsigned short* base, long tokstride,
                                                int pos0, int nrows, int L, int D, const float* cosT,
                                                const float* sinT, int tid, bool il = false) {
  const int cpr = D >> 3, half = D >> 1;
  for (int c = tid; c < nrows * cpr; c += 256) {
    const int r = c / cpr, ch = c - r * cpr;
    const int pos = pos0 + r;
    if (pos >= L) continue;
    const float* row = img + r * ILD;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = ch * 8 + e;
      float x = row[d];
      if (ROPE) {
        const bool lo = il ? !(d & 1) : d < half;
        const int j = il ? (d >> 1) : (lo ? d : d - half);
        const float cs = cosT[(long)pos * half + j], sn = sinT[(long)pos * half + j];
        const float other = row[il ? (d ^ 1) : (lo ? d + half : d - half)];
        x = lo ? (x * cs + other * sn) : (x * cs - other * sn);
      }
      v[e] = x;
    }
    u32x4 w;
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<u32x4*>(base + (long)pos * tokstride + ch * 8) = w;
  }
}

// Whole-head backward epilogue: one thread per token row.  The rotate-half tables of the row are loaded into
// registers BEFORE the next head's rows are requested: vmcnt retires in order, so a table load issued after that
// prefetch would have to wait for all of it.
template <int D> struct RopeRow { f32x4 cs[D / 8], sn[D / 8]; };

template <int D>
__device__ __forceinline__ void load_rope_row(RopeRow<D>& T, const float* cosT, const float* sinT, int pos) {
#pragma unroll
  for (int i = 0; i < D / 8; ++i) {
    T.cs[i] = *reinterpret_cast<const f32x4*>(cosT + pos * (D / 2) + 4 * i);
    T.sn[i] = *reinterpret_cast<const f32x4*>(sinT + pos * (D / 2) + 4 * i);
  }
}
// f32 image row (gradient w.r.t. the rotated q / k) -> RoPE^T -> bf16 row of dqkv
template <bool ROPE, int D>
__device__ __forceinline__ void store_grad_row(const float* row, unsigned short* dst, const RopeRow<D>& T, float scale,
                                               bool il = false) {
  constexpr int NG = D / 4;
  f32x4 x[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) x[i] = *reinterpret_cast<const f32x4*>(row + 4 * i) * scale;
  // `dst` may BE `row` (the whole-head backward converts in place): every read of the row before any write of it.  The
  // compiler sees a float row and an integer destination and assumes they cannot alias; in the rotate-half form the data
  // dependences (both halves feed every chunk) happen to enforce the order, in the interleaved form chunk c depends on
  // x[2 c], x[2 c + 1] only and nothing did.
  asm volatile("" ::: "memory");
  if (ROPE && il) {                                        // pair-interleaved head order: partners are neighbours
    // Four plain VALU instructions per pair, spelled out: left to hipcc this loop became v_pk_mul / v_pk_fma_f32 with
    // op_sel swizzles on half-overwritten register pairs, and the gradients of lanes 48 - 63 of every wave came out
    // different from run to run (a few hundred of 23.6 M elements, up to 0.03 absolute; tools/dbg_repro.py) - the
    // rotate-half form next to it, same kernel, same registers, is bit-reproducible.  The leading s_nop covers the wait
    // state a packed-f32 result needs before an instruction the hazard recogniser cannot see into.
#pragma unroll
    for (int gq = 0; gq < NG; ++gq)
#pragma unroll
      for (int q = 0; q < 4; q += 2) {
        const int j = 2 * gq + (q >> 1);                   // the pair's angle
        const float cs = T.cs[j >> 2][j & 3], sn = T.sn[j >> 2][j & 3];
        const float a = x[gq][q], b = x[gq][q + 1];
        float o0, o1, t0, t1;
        asm volatile("s_nop 1\n\t"
                     "v_mul_f32 %2, %5, %7\n\t"            // t0 = b sn
                     "v_mul_f32 %3, %4, %7\n\t"            // t1 = a sn
                     "v_fma_f32 %0, %4, %6, %2\n\t"        // a cs + b sn
                     "v_fma_f32 %1, %5, %6, -%3"            // b cs - a sn
                     : "=&v"(o0), "=&v"(o1), "=&v"(t0), "=&v"(t1)
                     : "v"(a), "v"(b), "v"(cs), "v"(sn));
        x[gq][q] = o0;
        x[gq][q + 1] = o1;
      }
  } else if (ROPE) {
#pragma unroll
    for (int i = 0; i < NG / 2; ++i) {
      const f32x4 lo = x[i], hi = x[i + NG / 2];
      x[i] = lo * T.cs[i] + hi * T.sn[i];
      x[i + NG / 2] = hi * T.cs[i] - lo * T.sn[i];
    }
  }
#pragma unroll
  for (int c = 0; c < D / 8; ++c) {
    u32x4 w;
    w[0] = pack_bf16x2(x[2 * c][0], x[2 * c][1]); w[1] = pack_bf16x2(x[2 * c][2], x[2 * c][3]);
    w[2] = pack_bf16x2(x[2 * c + 1][0], x[2 * c + 1][1]); w[3] = pack_bf16x2(x[2 * c + 1][2], x[2 * c + 1][3]);
    *reinterpret_cast<u32x4*>(dst + 8 * c) = w;
  }
}

// =================================================================================================
// backward dQ (+ delta): one workgroup = 128 queries, sweeps KVB-key blocks
// =================================================================================================
// DROP: dropout on the attention probabilities compiled in (a separate instantiation: the p = 0 kernels carry no
// trace of it)
template <int DP, int DR, int DX, bool DROP = false>
__global__ __launch_bounds__(256, Geo<DP>::WG_BWD) void attn_bwd_dq_kernel(const AP p) {
  constexpr int RS = Geo<DP>::RS, KS = Geo<DP>::KS, DT = Geo<DP>::DT, NCH = Geo<DP>::NCH, KVB = Geo<DP>::KVB;
  constexpr int NSUB = KVB / 64, ILD = DP + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned dseed = DROP ? drop_seed_eff(p.drop_seed, p.drop_epoch) : 0u;   // one scalar load per kernel
  char* ktile = smem;
  char* vtile = smem + KVB * RS;
  // rows [0,128) and [128,256) of smem stage Q and dO in the prologue (the region holds >= 256 rows)
  float* delta_l = reinterpret_cast<float*>(smem + 256 * RS);            // [128]
  unsigned char* mask_l = reinterpret_cast<unsigned char*>(delta_l + 128);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  // the head dim as a compile-time constant whenever the dispatcher knows it (DX): the `dt < dtv` guards of the
  // d-tile loops fold away - as run-time branches they pushed the dK / dV accumulators through scratch memory
  const int D = DX > 0 ? DX : p.D, H = p.H;
  int L = p.L;                                          // grid: the longest sequence; below: this sequence
  int qb, h, b;
  work_item((L + 127) / 128, H, p.B, qb, h, b);
  const long row0 = seq_rows(p, b, L);
  if (qb * 128 >= L) return;                              // packed batches: query block beyond this sequence
  const int cpr = D >> 3;
  const long tokstride = 3L * H * D, ostride = (long)H * D;
  const unsigned short* qbase = p.qkv + row0 * tokstride + (long)h * D;
  const unsigned short* kbase = qbase + (long)H * D;
  const unsigned short* vbase = qbase + 2L * H * D;
  const unsigned short* dobase = p.dout + row0 * ostride + (long)h * D;
  const unsigned short* obase = p.out + row0 * ostride + (long)h * D;
  const int q0 = qb * 128;
  const int dtv = (D + 15) >> 4;
  const float c2 = p.scale * LOG2E;

  // ---- prologue: Q and dO rows to LDS, delta = rowsum(dO * O) to delta_l and to global (the dK/dV kernel reads it)
  typename StagerFor<DP, DR, KVB>::type kv;
  kv.init(tid);
  if constexpr (DR == 0) {
    // chunk per lane: every (row, 16-byte chunk) task leaves its 8-term share of the row's dot product in LDS and
    // thread r adds the shares of row r in chunk order (deterministic)
    float* part = reinterpret_cast<float*>(mask_l + 256);              // [128 * NCH]
    ChunkRows<DP, 128> cq, cd, co;
    cq.load(qbase, tokstride, q0, L, cpr, tid);
    cd.load(dobase, ostride, q0, L, cpr, tid);
    co.load(obase, ostride, q0, L, cpr, tid);
    kv.load(kbase, tokstride, vbase, tokstride, 0, L, cpr);
    cq.store(smem, RS, cpr, tid);
    cd.store(smem + 128 * RS, RS, cpr, tid);
#pragma unroll
    for (int it = 0; it < ChunkRows<DP, 128>::NT; ++it) {
      const int idl = it * 256 + tid;
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += bf16_to_f32(co.c[it][e] & 0xffffu) * bf16_to_f32(cd.c[it][e] & 0xffffu);
        acc += bf16_to_f32(co.c[it][e] >> 16) * bf16_to_f32(cd.c[it][e] >> 16);
      }
      part[idl] = (idl % NCH) < cpr ? acc : 0.f;                        // pad chunks hold a re-read real chunk
    }
    __syncthreads();
    if (tid < 128) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc += part[tid * NCH + c];
      delta_l[tid] = acc;
      if (q0 + tid < L) p.delta[stat_at(p, b, h, H, L, row0, q0 + tid)] = acc;
    }
  } else {
    // row owner (rows are rotated in registers): Q rows on threads 0..127, dO rows on threads 128..255, which also
    // form delta
    PairStager<DP, DR, 128> qd;
    qd.init(tid);
    qd.load(qbase, tokstride, dobase, ostride, q0, L, cpr);
    if (!qd.is_a) {
      RowRegs<NCH> ro;
      const int pc = qd.pos < L ? qd.pos : L - 1;
      load_row<NCH>(ro, obase + (long)pc * ostride, cpr);
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc += bf16_to_f32(ro.c[i][e] & 0xffffu) * bf16_to_f32(qd.r.c[i][e] & 0xffffu);
          acc += bf16_to_f32(ro.c[i][e] >> 16) * bf16_to_f32(qd.r.c[i][e] >> 16);
        }
      delta_l[qd.row] = acc;
      if (qd.pos < L) p.delta[stat_at(p, b, h, H, L, row0, qd.pos)] = acc;
    }
    qd.store(smem, smem + 128 * RS, RS, p, L, true);
    kv.load(kbase, tokstride, vbase, tokstride, 0, L, cpr);
  }
  __syncthreads();
  bf16x8 qf[2][KS], dof[2][KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[qt][ks] = row_frag(smem, RS, wid * 32 + qt * 16 + li, ks, lane);
      dof[qt][ks] = row_frag(smem + 128 * RS, RS, wid * 32 + qt * 16 + li, ks, lane);
    }
  float lse2[2], dl[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int q = q0 + wid * 32 + qt * 16 + li; q = q < L ? q : L - 1;
    lse2[qt] = p.lse[stat_at(p, b, h, H, L, row0, q)] * LOG2E;
    dl[qt] = delta_l[wid * 32 + qt * 16 + li];
  }
  __syncthreads();
  kv.store(ktile, vtile, RS, p, L, true);
  if (tid < KVB) mask_l[tid] = (tid < L && (!p.key_mask || p.key_mask[row0 + tid])) ? 1 : 0;
  __syncthreads();

  f32x4 dq[DT][2];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dq[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[dt][1] = dq[dt][0]; }
  const int trow = 4 * g + (li >> 2), tcolb = 8 * (li & 3);

  const int nkb = (L + KVB - 1) / KVB;
  for (int kb = 0; kb < nkb; ++kb) {
    const bool more = (kb + 1) < nkb;
    if (more) kv.load(kbase, tokstride, vbase, tokstride, (kb + 1) * KVB, L, cpr);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (kb * KVB + sub * 64 >= L) break;
      const char* kt_ = ktile + sub * 64 * RS;
      const char* vt_ = vtile + sub * 64 * RS;
      f32x4 s[4][2], dp[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        // the S accumulators start at 0 / -inf for a valid / masked key (their rows are keys): the mask then costs
        // nothing per score, 2^(-inf) = 0 falls out of the exponential
        const unsigned int mk = *reinterpret_cast<const unsigned int*>(mask_l + sub * 64 + kt * 16 + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][0][r] = ((mk >> (8 * r)) & 0xffu) ? 0.f : -INFINITY;
        s[kt][1] = s[kt][0];
        // dP accumulators start at -delta of their query column: the MFMAs deliver dP - delta, no subtraction per score
        dp[kt][0] = f32x4{-dl[0], -dl[0], -dl[0], -dl[0]};
        dp[kt][1] = f32x4{-dl[1], -dl[1], -dl[1], -dl[1]};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const bf16x8 kf = row_frag(kt_, RS, kt * 16 + li, ks, lane);
          const bf16x8 vf = row_frag(vt_, RS, kt * 16 + li, ks, lane);
          s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[kt][0], 0, 0, 0);
          s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[kt][1], 0, 0, 0);
          dp[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[0][ks], dp[kt][0], 0, 0, 0);
          dp[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[1][ks], dp[kt][1], 0, 0, 0);
        }
      // dS^T = P (dP - delta), P = 2^(s c2 - lse + kb), kb = 0 / -inf for a valid / masked key (one select per key
      // row instead of one per element)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {
            const float pv = fast_exp2(s[kt][qt][r] * c2 - lse2[qt]);
            float dpv = dp[kt][qt][r];                             // dP - delta
            if constexpr (DROP) {                                  // dP passes through the dropout mask, delta does not
              const long qrow = row0 + q0 + wid * 32 + qt * 16 + li;
              const float ms = attn_drop(p, dseed, qrow, h, kb * KVB + sub * 64 + kt * 16 + 4 * g + r);
              dpv = (dpv + dl[qt]) * ms - dl[qt];
            }
            s[kt][qt][r] = pv * dpv;                               // dS^T (w.r.t. the scaled score)
          }
      bf16x8 db[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        db[qt][0] = pack_acc_pair(s[0][qt], s[1][qt]);
        db[qt][1] = pack_acc_pair(s[2][qt], s[3][qt]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          if (dt < dtv) {
            const bf16x8 kt_f = tr_frag(kt_, RS, 32 * s2 + trow, dt * 32 + tcolb);
            dq[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_f, db[0][s2], dq[dt][0], 0, 0, 0);
            dq[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_f, db[1][s2], dq[dt][1], 0, 0, 0);
          }
    }
    __syncthreads();
    if (more) {
      kv.store(ktile, vtile, RS, p, L, true);
      if (tid < KVB) {
        const int pos = (kb + 1) * KVB + tid;
        mask_l[tid] = (pos < L && (!p.key_mask || p.key_mask[row0 + pos])) ? 1 : 0;
      }
      __syncthreads();
    }
  }

  // dq~ (gradient w.r.t. the rotated q) -> f32 LDS image [128][DP+4] -> RoPE^T -> bf16 rows of dqkv
  float* img = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        img[(wid * 32 + qt * 16 + li) * ILD + dt * 16 + 4 * g + r] = dq[dt][qt][r] * p.scale;
  __syncthreads();
  store_grad_rows<(DR > 0)>(img, ILD, p.dqkv + row0 * tokstride + (long)h * D, tokstride, q0, 128, L, D,
                            p.cosT, p.sinT, tid, p.pre_rot == 2);
}

// =================================================================================================
// backward dK/dV: one workgroup = KVB keys (4 waves x KVB/4), sweeps KVB-query blocks
// =================================================================================================
// DROP: dropout on the attention probabilities compiled in (a separate instantiation: the p = 0 kernels carry no
// trace of it)
template <int DP, int DR, int DX, bool DROP = false>
__global__ __launch_bounds__(256, Geo<DP>::WG_BWD) void attn_bwd_dkv_kernel(const AP p) {
  constexpr int RS = Geo<DP>::RS, KS = Geo<DP>::KS, DT = Geo<DP>::DT, KVB = Geo<DP>::KVB, ILD = DP + 4;
  constexpr int KPB = Geo<DP>::KPB;                       // this workgroup's keys
  constexpr int KTW = KPB / 64;                           // 16-key tiles per wave
  constexpr int KPW = 16 * KTW, QB = KVB, NS2 = QB / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned dseed = DROP ? drop_seed_eff(p.drop_seed, p.drop_epoch) : 0u;   // one scalar load per kernel
  char* qtile = smem;                                     // [QB q][RS]   (first holds this workgroup's K rows)
  char* dotile = smem + QB * RS;                          // [QB q][RS]   (first holds this workgroup's V rows)
  float* lse_l = reinterpret_cast<float*>(smem + 2 * QB * RS);   // [QB]
  float* dl_l = lse_l + QB;                                      // [QB]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  // the head dim as a compile-time constant whenever the dispatcher knows it (DX): the `dt < dtv` guards of the
  // d-tile loops fold away - as run-time branches they pushed the dK / dV accumulators through scratch memory
  const int D = DX > 0 ? DX : p.D, H = p.H;
  int L = p.L;                                          // grid: the longest sequence; below: this sequence
  int kbk, h, b;
  work_item((L + KPB - 1) / KPB, H, p.B, kbk, h, b);
  const long row0 = seq_rows(p, b, L);
  if (kbk * KPB >= L) return;                             // packed batches: key block beyond this sequence
  const int cpr = D >> 3;
  const long tokstride = 3L * H * D, ostride = (long)H * D;
  const unsigned short* qbase = p.qkv + row0 * tokstride + (long)h * D;
  const unsigned short* kbase = qbase + (long)H * D;
  const unsigned short* vbase = qbase + 2L * H * D;
  const unsigned short* dobase = p.dout + row0 * ostride + (long)h * D;
  const int k0 = kbk * KPB;
  const int dtv = (D + 15) >> 4;
  const float c2 = p.scale * LOG2E;

  // ---- this workgroup's keys: K (rotated) and V as B fragments B[k = d][col = key], kept in registers
  typename StagerFor<DP, DR, QB>::type st;
  st.init(tid);
  {
    typename StagerFor<DP, DR, KPB>::type sk;
    sk.init(tid);
    sk.load(kbase, tokstride, vbase, tokstride, k0, L, cpr);
    st.load(qbase, tokstride, dobase, ostride, 0, L, cpr);          // first query block: in flight during the fragment reads
    sk.store(qtile, dotile, RS, p, L, true);
  }
  __syncthreads();
  bf16x8 kf[KTW][KS], vf[KTW][KS];
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[kt][ks] = row_frag(qtile, RS, wid * KPW + kt * 16 + li, ks, lane);
      vf[kt][ks] = row_frag(dotile, RS, wid * KPW + kt * 16 + li, ks, lane);
    }
  bool kvalid[KTW];
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt) {
    const int key = k0 + wid * KPW + kt * 16 + li;
    kvalid[kt] = key < L && (!p.key_mask || p.key_mask[row0 + key]);
  }
  float kbias[KTW];
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt) kbias[kt] = kvalid[kt] ? 0.f : -INFINITY;
  __syncthreads();

  f32x4 dk[DT][KTW], dv[DT][KTW];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) { dk[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = dk[dt][kt]; }
  const int trow = 4 * g + (li >> 2), tcolb = 8 * (li & 3);

  const int nqb = (L + QB - 1) / QB;
  for (int qb = 0; qb < nqb; ++qb) {
    st.store(qtile, dotile, RS, p, L, true);                         // Q rows rotated, dO rows as they are
    if (tid < QB) {
      const int q = qb * QB + tid;
      const bool ok = q < L;
      // queries past the end: lse = +inf makes p = exp2(-inf) = 0
      lse_l[tid] = ok ? p.lse[stat_at(p, b, h, H, L, row0, q)] * LOG2E : INFINITY;
      dl_l[tid] = ok ? -p.delta[stat_at(p, b, h, H, L, row0, q)] : 0.f;      // NEGATED: the dP accumulators start from it
    }
    __syncthreads();
    if (qb + 1 < nqb) st.load(qbase, tokstride, dobase, ostride, (qb + 1) * QB, L, cpr);   // prefetch under the MFMAs

#pragma unroll
    for (int s2 = 0; s2 < NS2; ++s2) {                     // 32 queries at a time
      if (qb * QB + s2 * 32 >= L) break;
      // one 16-query tile at a time: its P and dS leave the f32 accumulators as packed bf16 before the next tile's
      // accumulators are needed.  Rows of the accumulators are queries (4g + r), columns are this lane's key: S
      // starts at 0 / -inf for a valid / masked key, dP at -delta of its row (the MFMAs then deliver dP - delta)
      u32x2 pk[2][KTW], dsk[2][KTW];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int row = (2 * s2 + qq) * 16 + li;
        const f32x4 nd = *reinterpret_cast<const f32x4*>(dl_l + (2 * s2 + qq) * 16 + 4 * g);
        f32x4 s[KTW], dp[KTW];
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) { s[kt] = f32x4{kbias[kt], kbias[kt], kbias[kt], kbias[kt]}; dp[kt] = nd; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 qa = row_frag(qtile, RS, row, ks, lane);
          const bf16x8 da = row_frag(dotile, RS, row, ks, lane);
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) {
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], s[kt], 0, 0, 0);
            dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[kt], 0, 0, 0);
          }
        }
        const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_l + (2 * s2 + qq) * 16 + 4 * g);
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = fast_exp2(s[kt][r] * c2 - ls[r]);
            float pd = pv, dpv = dp[kt][r];
            if constexpr (DROP) {                               // dV sees the dropped-out P; dP passes through the mask
              const long qrow = row0 + qb * QB + (2 * s2 + qq) * 16 + 4 * g + r;
              const float ms = attn_drop(p, dseed, qrow, h, k0 + wid * KPW + kt * 16 + li);
              pd = pv * ms;
              dpv = (dpv - nd[r]) * ms + nd[r];                 // nd = -delta
            }
            s[kt][r] = pd;                                      // P (after dropout) for dV
            dp[kt][r] = pv * dpv;                               // dS = P (dP - delta)
          }
          pk[qq][kt] = u32x2{pack_bf16x2(s[kt][0], s[kt][1]), pack_bf16x2(s[kt][2], s[kt][3])};
          dsk[qq][kt] = u32x2{pack_bf16x2(dp[kt][0], dp[kt][1]), pack_bf16x2(dp[kt][2], dp[kt][3])};
        }
      }
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) {
        const bf16x8 pbf = __builtin_bit_cast(bf16x8, u32x4{pk[0][kt][0], pk[0][kt][1], pk[1][kt][0], pk[1][kt][1]});
        const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dsk[0][kt][0], dsk[0][kt][1], dsk[1][kt][0], dsk[1][kt][1]});
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          if (dt < dtv) {
            const bf16x8 dot_f = tr_frag(dotile, RS, 32 * s2 + trow, dt * 32 + tcolb);   // dO^T
            const bf16x8 qt_f = tr_frag(qtile, RS, 32 * s2 + trow, dt * 32 + tcolb);     // Q~^T
            dv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f, pbf, dv[dt][kt], 0, 0, 0);
            dk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsf, dk[dt][kt], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }

  // ---- dK~ / dV -> f32 LDS images -> (RoPE^T for dK) -> bf16 rows of dqkv
  float* img = reinterpret_cast<float*>(smem);
  unsigned short* dkbase = p.dqkv + row0 * tokstride + (long)H * D + (long)h * D;
  unsigned short* dvbase = p.dqkv + row0 * tokstride + 2L * H * D + (long)h * D;
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        img[(wid * KPW + kt * 16 + li) * ILD + dt * 16 + 4 * g + r] = dk[dt][kt][r] * p.scale;
  __syncthreads();
  store_grad_rows<(DR > 0)>(img, ILD, dkbase, tokstride, k0, KPB, L, D, p.cosT, p.sinT, tid, p.pre_rot == 2);
  __syncthreads();
#pragma unroll
  for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        img[(wid * KPW + kt * 16 + li) * ILD + dt * 16 + 4 * g + r] = dv[dt][kt][r];
  __syncthreads();
  store_grad_rows<false>(img, ILD, dvbase, tokstride, k0, KPB, L, D, p.cosT, p.sinT, tid);
}

// =================================================================================================
// backward, whole head in one workgroup (head dim <= 32, 128 < L <= 256): 5 MFMA products instead of 7 and the
// exponentials once instead of twice.
//
// The short-head backward is VALU-bound (one 16x16 score tile = 5 x 16 MFMA cycles against ~200 VALU cycles of
// softmax / dS arithmetic), so recomputing S and dP in a second kernel doubles the cost that matters.  Here one
// workgroup owns one (batch, head): all 256 Q and dO rows sit in LDS (64-byte rows, 16-byte chunks XOR-swizzled by
// (row >> 1) & 3: conflict-free for the b128 row reads and the transposed reads, tools/lds_conflicts.py), wave w owns
// keys [64w, 64w + 64) with their K / V fragments and the dK / dV accumulators in registers, and sweeps the eight
// 32-query blocks.  dS leaves the accumulators once more, transposed: written as bf16 [key][query] rows to a private
// 2 x 2 KiB LDS buffer and read back with ds_read_b64_tr_b16 as the B operand of dQ^T += K^T dS^T.  Each wave's dQ
// contribution covers its 64 keys only; the four waves walk the query blocks ROTATED (wave w takes block (s + w) & 7
// at step s), so at every step they add into four different rows of the f32 dQ image in LDS with plain
// read-add-write, one barrier per step, and every dQ element is summed in a fixed order: no float atomics, the
// gradients stay bitwise reproducible.
// =================================================================================================
__device__ __forceinline__ int swz64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

// transposed fragment from a swizzled 64-byte-row tile: `off` = this lane's byte offset for rows [0, 16) (rows
// 16..31 are 1024 bytes further, same swizzle)
__device__ __forceinline__ bf16x8 tr_frag_off(const char* tile, int off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off));
  const s16x4 hi =
      __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + off + 1024));
  bf16x8 f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

constexpr int FUSED_LMAX = 256;
#ifdef CLIPK_ATTN_TRACE
// experiment builds (tools/exp_attn_trace.py): shader-clock stamps (s_memtime) between the phases of the whole-head
// backward, summed over the heads of workgroup 0 (thread 0) -> cycles per phase and head.  No output depends on them.
__device__ unsigned long long* g_attn_trace = nullptr;
__device__ int g_attn_stagger = 0;       // experiment: the second workgroup of each CU starts this x ~3.9 us late
// (the sums live in 11 x 8 bytes of LDS behind the kernel's own allocation: the sweep has no registers to spare)
#define ATTN_STAMP(i)                                                                   \
  do {                                                                                  \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                         \
    if (tr_on) atomicAdd(&tr_lds[i], t_ - tr_t);                                        \
    tr_t = t_;                                                                          \
  } while (0)
#else
#define ATTN_STAMP(i) do { } while (0)
#endif
__host__ __device__ constexpr size_t lds_fused(int D) {
  // the bf16 dV image of the epilogue sits behind the dK image inside the dead Q / dO / dS^T region when both fit
  // (D <= 24), behind everything otherwise
  return (size_t)2 * FUSED_LMAX * 64 + 4 * 4096 + (size_t)FUSED_LMAX * (D + 4) * 4 + 2 * FUSED_LMAX * 4 +
         (D > 24 ? (size_t)FUSED_LMAX * 64 : 0);
}
#ifdef CLIPK_ATTN_TRACE
constexpr size_t FUSED_TRACE_LDS = 128;
#else
constexpr size_t FUSED_TRACE_LDS = 0;
#endif

// one head's rows as this thread holds them between the loads and the LDS staging: chunk (tid & 3) of rows
// (tid >> 2) + (threads / 4) * pass of K, V, Q, dO and O, this thread's lse and the mask bytes of its key columns
template <int NPS, int KTW>
struct HeadRegs {
  u32x4 k[NPS], v[NPS], q[NPS], d[NPS], o[NPS];
  float lse;
  unsigned char km[KTW];
};

// NW = waves per workgroup.  4 (the default): two workgroups per CU, a wave owns 64 keys, the next head's rows can
// only be requested after the sweep (the sweep needs 239 of the 256 registers).  8 (kept for A/B, slower - see
// launch_fused): one workgroup per CU, a wave owns 32 keys - half the accumulators and fragments, and the rows of a
// head spread over twice the threads (40 prefetch registers instead of 80) - so the next head is requested BEFORE
// the sweep and its gather runs under it.
template <bool ROPE, int D, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attn_bwd_fused32_kernel(const AP p) {
  constexpr int DT = 2, LQ = FUSED_LMAX;
  constexpr int NT = 64 * NW, KTW = 16 / NW, KW = 16 * KTW, NCK = KTW / 2, NPS = LQ * 4 / NT;
  constexpr int ILD = D + 4, cpr = D / 8;
  typedef HeadRegs<NPS, KTW> Regs;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* qtile = smem;                                     // [256 q][64 B]   (first: this head's K rows)
  char* dotile = smem + LQ * 64;                          // [256 q][64 B]   (first: this head's V rows)
  char* dst_all = smem + 2 * LQ * 64;                     // NW waves x NCK x [32 keys][32 q] bf16 = 16 KiB
  float* img = reinterpret_cast<float*>(dst_all + 4 * 4096);   // dQ image [256][D + 4] f32
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int H = p.H;                                      // (the sequence length L is per head: packed batches)
  float* lse_l = img + LQ * ILD;                          // [256], already times log2(e); +inf past the end
  float* dl_l = lse_l + LQ;                               // [256]
  const long tokstride = 3L * H * D, ostride = (long)H * D;
  const float c2 = p.scale * LOG2E;
  const int nheads = p.B * H;

  // Staging: four lanes per token row (one 16-byte chunk each, the pad chunk zero), 64 rows per pass: a wave
  // instruction covers 16 rows x D*2 contiguous bytes.  (One thread per row - the staging of the other kernels, needed
  // there to rotate rows in registers - touches 64 different 128-byte lines per instruction.)
  auto issue = [&](int w, Regs& R) {
    int blk_, h_, b_;
    work_item_at(w, 1, H, p.B, blk_, h_, b_);
    int L = p.L;
    long row0_ = seq_rows(p, b_, L);                       // packed batch: this sequence's first row and length
    if (L <= 0) { L = 1; row0_ = row0_ > 0 ? row0_ - 1 : 0; }   // empty sequence: in-range dummy row, never used
    const int t = tid + opaque_zero();
    const int ci = t & 3, r0 = t >> 2;
    // uniform bases (SGPR pairs) + 32-bit element offsets: one VGPR per address
    const unsigned short* qb = p.qkv + row0_ * tokstride + (long)h_ * D;
    const unsigned short* dob = p.dout + row0_ * ostride + (long)h_ * D;
    const unsigned short* ob = p.out + row0_ * ostride + (long)h_ * D;
    const unsigned int HD = (unsigned int)(H * D);
    // no branch around the loads (a lane whose chunk is the zero pad re-reads the last real chunk and drops it at
    // staging time): loads under a divergent branch make every later s_waitcnt assume they may not have been issued
    const unsigned int cc = 8u * (ci < cpr ? ci : cpr - 1);
#ifdef CLIPK_ATTN_HM_PROBE
    // TIMING-ONLY experiment (tools/exp_attn_headmajor.py): q, k, v and dO addressed as if they were stored head-major
    // ([B][H][3][L][D] / [B][H][L][D]: a head's rows contiguous).  The bytes read are not the head's: results garbage.
    const unsigned short* qh = p.qkv + (long)(b_ * H + h_) * 3 * p.L * D;
    const unsigned short* dh = p.dout + (long)(b_ * H + h_) * p.L * D;
    const unsigned int LD = (unsigned int)(p.L * D);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      int row = ps * (NT / 4) + r0; row = row < L ? row : L - 1;
      const unsigned int ho = (unsigned int)row * D + cc, oo = (unsigned int)row * HD + cc;
      R.q[ps] = *reinterpret_cast<const u32x4*>(qh + ho);
      R.k[ps] = *reinterpret_cast<const u32x4*>(qh + (ho + LD));
      R.v[ps] = *reinterpret_cast<const u32x4*>(qh + (ho + 2u * LD));
      R.d[ps] = *reinterpret_cast<const u32x4*>(dh + ho);
      R.o[ps] = *reinterpret_cast<const u32x4*>(ob + oo);
    }
    (void)qb; (void)dob;
#else
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      int row = ps * (NT / 4) + r0; row = row < L ? row : L - 1;
      const unsigned int qo = (unsigned int)row * 3u * HD + cc, oo = (unsigned int)row * HD + cc;
      R.q[ps] = *reinterpret_cast<const u32x4*>(qb + qo);
      R.k[ps] = *reinterpret_cast<const u32x4*>(qb + (qo + HD));
      R.v[ps] = *reinterpret_cast<const u32x4*>(qb + (qo + 2u * HD));
      R.d[ps] = *reinterpret_cast<const u32x4*>(dob + oo);
      R.o[ps] = *reinterpret_cast<const u32x4*>(ob + oo);
    }
#endif
    R.lse = p.lse[stat_at(p, b_, h_, H, L, row0_, t < L ? t : L - 1)];
    const int lane_ = t & 63, wid_ = t >> 6;
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) {
      const int key = wid_ * KW + kt * 16 + (lane_ & 15);
      R.km[kt] = p.key_mask ? p.key_mask[row0_ + (key < L ? key : L - 1)] : (unsigned char)1;
    }
  };

  // per-lane offsets into the swizzled tiles (row-block bases are multiples of 16, which the swizzle ignores)
  const int trow = 4 * g + (li >> 2);
  const int off_rf = swz64(li, g);                                            // row li, chunk g (8 of the 32 d)
  int off_tr[DT];                                                             // row trow, bytes dt*32 + 8*(li&3)
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) off_tr[dt] = swz64(trow, dt * 2 + ((li & 3) >> 1)) + 8 * (li & 1);
  char* dst = dst_all + wid * (NCK * 2048);
  // dS^T write: key row ktl*16 + li, queries qq*16 + 4g .. +3 -> 8 bytes at byte column qq*32 + 8g
  int off_dw[2];
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) off_dw[qq] = swz64(li, qq * 2 + (g >> 1)) + 8 * (g & 1);
  const bool dt1_live = 16 + 4 * g < D;                  // rows 16 + 4g .. of the second d tile are real head dims

  // Persistent: 2 workgroups per CU, each walking heads w, w + gridDim.x, ...; the rows of the NEXT head are
  // requested right after the sweep of the current one and land in registers while its gradients are written out.
  // (Tried and dropped: a start-up offset between the two workgroups of a CU so that one sweeps while the other moves
  // data.  Timed in interleaved rounds on a warm GPU it changes nothing - 450 us with or without; the gain first
  // measured came from the clock ramp of the first few hundred launches of a process.)
  Regs R;
  int w = blockIdx.x;
  if (w < nheads) issue(w, R);
#ifdef CLIPK_ATTN_TRACE
  const bool tr_on = blockIdx.x == 0 && tid == 0 && g_attn_trace != nullptr;
  unsigned long long* tr_lds = reinterpret_cast<unsigned long long*>(smem + lds_fused(D));
  if (tid < 16) tr_lds[tid] = 0;
  __syncthreads();
  if (g_attn_stagger > 0 && (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 0xF) != 0)     // HW_REG_HW_ID[3:0]: wave slot in the SIMD
    for (int i = 0; i < g_attn_stagger; ++i) __builtin_amdgcn_s_sleep(127);
  unsigned long long tr_t = __builtin_amdgcn_s_memtime();
  const unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime(), tr_c0 = tr_t;
#endif

  for (; w < nheads; w += gridDim.x) {
    int blk, h, b;
    work_item_at(w, 1, H, p.B, blk, h, b);
    int L = p.L;
    const long row0 = seq_rows(p, b, L);
    ATTN_STAMP(7);                                         // (loop bookkeeping)
    if (L <= 0) {                                          // (workgroup-uniform) empty sequence of a packed batch
      issue(w + (int)gridDim.x < nheads ? w + (int)gridDim.x : w, R);
      continue;
    }
    const int tq = tid + opaque_zero();
    const int ci = tq & 3, r0 = tq >> 2;
    // ---- K / V rows to LDS, lse and delta = rowsum(dO * O) to their arrays, dQ image to zero
    if (tid < LQ) lse_l[tid] = tid < L ? R.lse * LOG2E : INFINITY;       // +inf: p = 2^-inf = 0 past the end
    if (cpr < 4 && ci >= cpr) {                            // the pad chunk: zeros in LDS
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) R.k[ps] = R.v[ps] = R.q[ps] = R.d[ps] = R.o[ps] = z;
    }
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int row = ps * (NT / 4) + r0;
      *reinterpret_cast<u32x4*>(qtile + swz64(row, ci)) = R.k[ps];
      *reinterpret_cast<u32x4*>(dotile + swz64(row, ci)) = R.v[ps];
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += bf16_to_f32(R.o[ps][e] & 0xffffu) * bf16_to_f32(R.d[ps][e] & 0xffffu);
        acc += bf16_to_f32(R.o[ps][e] >> 16) * bf16_to_f32(R.d[ps][e] >> 16);
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (ci == 0) {
        const bool ok = row < L;
        dl_l[row] = ok ? -acc : 0.f;                       // NEGATED: the dP accumulators start from it
        if (ok) p.delta[stat_at(p, b, h, H, L, row0, row)] = acc;
      }
    }
    if (tid < LQ)
      for (int i = 0; i < ILD; i += 4) *reinterpret_cast<f32x4*>(img + tid * ILD + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    ATTN_STAMP(0);                                         // wait for the prefetched rows + K / V staging + delta + image zero

    bf16x8 kf[KTW], vf[KTW], ktf[DT][NCK];
    float kbias[KTW];
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) {
      kf[kt] = *reinterpret_cast<const bf16x8*>(qtile + (wid * KW + kt * 16) * 64 + off_rf);
      vf[kt] = *reinterpret_cast<const bf16x8*>(dotile + (wid * KW + kt * 16) * 64 + off_rf);
      kbias[kt] = (wid * KW + kt * 16 + li < L && R.km[kt]) ? 0.f : -INFINITY;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int c = 0; c < NCK; ++c) ktf[dt][c] = tr_frag_off(qtile + (wid * KW + c * 32) * 64, off_tr[dt]);   // K~^T
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      *reinterpret_cast<u32x4*>(qtile + swz64(ps * (NT / 4) + r0, ci)) = R.q[ps];
      *reinterpret_cast<u32x4*>(dotile + swz64(ps * (NT / 4) + r0, ci)) = R.d[ps];
    }
    // eight waves: R is free now and the sweep leaves room for it - request the next head here, its gather runs
    // under the sweep.  (Unconditional, see below; the last head of a workgroup re-requests its own rows.)
    if constexpr (NW == 8) issue(w + (int)gridDim.x < nheads ? w + (int)gridDim.x : w, R);
    __syncthreads();
    ATTN_STAMP(1);                                         // fragments + Q / dO staging

    f32x4 dk[DT][KTW], dv[DT][KTW];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) { dk[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = dk[dt][kt]; }

#pragma unroll 1
    for (int step = 0; step < LQ / 32; ++step) {
      const int j = (step + wid) & (LQ / 32 - 1);
      if (j * 32 < L && wid * KW < L) {                     // wave-uniform: a query block and keys of this wave exist
        const char* qt_ = qtile + j * 32 * 64;
        const char* dt_ = dotile + j * 32 * 64;
        // one 16-query tile at a time: its P and dS leave the f32 accumulators as packed bf16 (2 + 2 registers per
        // key tile) before the next tile's 32 accumulator registers are needed
        u32x2 pk[2][KTW], dsk[2][KTW];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qt_ + qq * 1024 + off_rf);
          const bf16x8 da = *reinterpret_cast<const bf16x8*>(dt_ + qq * 1024 + off_rf);
          // rows of the accumulators are queries (4g + r), columns are this lane's key.  dP starts at -delta of its
          // row, so the MFMA delivers dP - delta; S starts at 0 / -inf for a valid / masked key
          const f32x4 nd = *reinterpret_cast<const f32x4*>(dl_l + j * 32 + qq * 16 + 4 * g);
          f32x4 s[KTW], dp[KTW];
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) {
            s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                qa, kf[kt], f32x4{kbias[kt], kbias[kt], kbias[kt], kbias[kt]}, 0, 0, 0);
            dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt], nd, 0, 0, 0);
          }
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_l + j * 32 + qq * 16 + 4 * g);
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = fast_exp2(s[kt][r] * c2 - ls[r]);
              s[kt][r] = pv;                                      // P
              dp[kt][r] = pv * dp[kt][r];                         // dS = P (dP - delta)
            }
            pk[qq][kt] = u32x2{pack_bf16x2(s[kt][0], s[kt][1]), pack_bf16x2(s[kt][2], s[kt][3])};
            dsk[qq][kt] = u32x2{pack_bf16x2(dp[kt][0], dp[kt][1]), pack_bf16x2(dp[kt][2], dp[kt][3])};
          }
        }
        bf16x8 dot_f[DT], qt_f[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          dot_f[dt] = tr_frag_off(dt_, off_tr[dt]);               // dO^T
          qt_f[dt] = tr_frag_off(qt_, off_tr[dt]);                // Q~^T
        }
        f32x4 dq[DT][2];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dq[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[dt][1] = dq[dt][0]; }
#pragma unroll
        for (int c = 0; c < NCK; ++c) {
#pragma unroll
          for (int ktl = 0; ktl < 2; ++ktl) {
            const int kt = 2 * c + ktl;
            const bf16x8 pbf = __builtin_bit_cast(bf16x8, u32x4{pk[0][kt][0], pk[0][kt][1], pk[1][kt][0], pk[1][kt][1]});
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dsk[0][kt][0], dsk[0][kt][1], dsk[1][kt][0], dsk[1][kt][1]});
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) *reinterpret_cast<u32x2*>(dst + c * 2048 + ktl * 1024 + off_dw[qq]) = dsk[qq][kt];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              dv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f[dt], pbf, dv[dt][kt], 0, 0, 0);
              dk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f[dt], dsf, dk[dt][kt], 0, 0, 0);
            }
          }
          // dQ^T[d][q] += K~^T[d][32 keys of chunk c] dS^T[32 keys][q]
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const bf16x8 dsb = tr_frag_off(dst + c * 2048, swz64(trow, qq * 2 + ((li & 3) >> 1)) + 8 * (li & 1));
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
              dq[dt][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[dt][c], dsb, dq[dt][qq], 0, 0, 0);
          }
        }
        // this wave's share of dQ (its keys only) for query block j: plain read-add-write, no other wave is on block j now
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          float* row = img + (j * 32 + qq * 16 + li) * ILD + 4 * g;
          f32x4 a = *reinterpret_cast<f32x4*>(row);
          a += dq[0][qq];                                       // unscaled: store_grad_row applies q_scale once
          *reinterpret_cast<f32x4*>(row) = a;
          if (dt1_live) {
            f32x4 c1 = *reinterpret_cast<f32x4*>(row + 16);
            c1 += dq[1][qq];
            *reinterpret_cast<f32x4*>(row + 16) = c1;
          }
        }
      }
      __syncthreads();
    }

    // ---- this thread's rotate-half table row first, then the next head's rows (consumed at the top of the next
    // iteration).  The request is unconditional - the last head of a workgroup asks for its own rows again, L2 hits
    // nobody consumes - because a branch here makes the compiler wait for vmcnt(0) at the first use of the tables
    // (on the not-taken path they are the youngest loads), which waits for the whole prefetch as well.
    // (Eight waves: the rows were requested before the sweep and have landed; the tables simply come now.)
    ATTN_STAMP(2);                                         // the sweep
    RopeRow<D> T;
    if (ROPE) load_rope_row<D>(T, p.cosT, p.sinT, tq < L ? tq : L - 1);
    // (asking for the head in two instalments around the LDS work of the write-out changed nothing: 850 vs 853 us)
    if constexpr (NW == 4) issue(w + (int)gridDim.x < nheads ? w + (int)gridDim.x : w, R);
    ATTN_STAMP(3);                                         // issuing the next head's loads

    // ---- dQ rows from the image; dK~ (f32, RoPE^T wants f32 pairs) and dV (bf16) through images over the now dead
    // Q / dO / dS^T region
    float* img2 = reinterpret_cast<float*>(smem);
    char* dvimg = D <= 24 ? smem + LQ * ILD * 4 : reinterpret_cast<char*>(dl_l + LQ);   // bf16 [256][64 B]
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        if (dt == 0 || dt1_live) {
          const int key = wid * KW + kt * 16 + li, d = dt * 16 + 4 * g;
          *reinterpret_cast<f32x4*>(img2 + key * ILD + d) = dk[dt][kt];
          u32x2 wv;
          wv[0] = pack_bf16x2(dv[dt][kt][0], dv[dt][kt][1]);
          wv[1] = pack_bf16x2(dv[dt][kt][2], dv[dt][kt][3]);
          *reinterpret_cast<u32x2*>(dvimg + key * 64 + d * 2) = wv;
        }
    __syncthreads();
    ATTN_STAMP(4);                                         // dK / dV images + barrier
    // RoPE^T, scale and rounding by the row's owner, IN PLACE (the bf16 row over the start of its own f32 row; the
    // row is read whole before it is written) ...
    if (tq < L) {
      store_grad_row<ROPE, D>(img + tq * ILD, reinterpret_cast<unsigned short*>(img + tq * ILD), T, p.scale, p.pre_rot == 2);   // q_scale: once, here
      store_grad_row<ROPE, D>(img2 + tq * ILD, reinterpret_cast<unsigned short*>(img2 + tq * ILD), T, p.scale, p.pre_rot == 2);
    }
    __syncthreads();
    // ... and the rows leave four lanes to a row, as they came: a wave instruction covers 16 rows x 2 D bytes.  (One
    // thread per row - 64 lines per store instruction - kept the address path busy for 2 us per head and stood in
    // the way of the other workgroup's loads: 881 -> 863 us per layer at B = 1024, profiles/r03/attn_row_stores_ab_v1.txt.)
    {
      unsigned short* dqb = p.dqkv + row0 * tokstride + (long)h * D;
      if (ci < cpr) {
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
          const int row = ps * (NT / 4) + r0;
          if (row < L) {
            unsigned short* drow = dqb + (unsigned int)row * (unsigned int)tokstride + 8 * ci;
            *reinterpret_cast<u32x4*>(drow) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(img + row * ILD) + 16 * ci);
            *reinterpret_cast<u32x4*>(drow + H * D) =
                *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(img2 + row * ILD) + 16 * ci);
            *reinterpret_cast<u32x4*>(drow + 2 * H * D) = *reinterpret_cast<const u32x4*>(dvimg + row * 64 + 16 * ci);
          }
        }
      }
    }
    ATTN_STAMP(5);                                         // gradient rows: RoPE^T in place, barrier, row stores issued
    __syncthreads();                                       // the images are read; the next head may stage over them
    ATTN_STAMP(6);                                         // the closing barrier
#ifdef CLIPK_ATTN_TRACE
    if (tr_on) atomicAdd(&tr_lds[8], 1ull);
#endif
  }
#ifdef CLIPK_ATTN_TRACE
  if (tr_on) {
    for (int i = 0; i < 9; ++i) g_attn_trace[i] = tr_lds[i];
    g_attn_trace[9] = __builtin_amdgcn_s_memtime() - tr_c0;
    g_attn_trace[10] = __builtin_amdgcn_s_memrealtime() - tr_r0;
  }
#endif
}
#ifdef CLIPK_ATTN_TRACE
}  // namespace
extern "C" int clipk_attn_set_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
extern "C" int clipk_attn_set_stagger(int n) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stagger), &n, sizeof(n)) == hipSuccess ? 0 : -1;
}
namespace {
#endif

// =================================================================================================
// backward, whole head in one workgroup, head dim 96 (the 6 x 768 RNA encoder: 8 heads of 96), 128 < L <= 256, rows
// that need no rotation, no dropout.
//
// The dQ + dK/dV kernel pair reads q, k, v, dO twice (3.6 GB per layer at B = 1024 against 2.0 GB read once) and runs
// 7 products and the exponentials twice; its dK/dV half gives every 64-key workgroup a prologue and an epilogue as long
// as its sweep.  Here one workgroup (4 waves, ONE per SIMD: 512 VGPRs) owns a (batch, head): wave w holds the K / V
// fragments of keys [64w, 64w + 64) (loaded straight from HBM in fragment layout: 16-byte chunk g + 4 ks of row li),
// their K^T fragments (transposed LDS reads of the staged K rows) and the dK^T / dV^T accumulators in registers - 336
// of them - and all four waves sweep the eight 32-query blocks TOGETHER: Q / dO blocks are double-buffered in LDS
// (chunk per lane through registers, the next block requested before the sweep of the current one), delta =
// rowsum(dO * O) is formed from the staged chunks (shares in LDS, summed in chunk order), dS goes back to LDS transposed
// (private 4 KiB per wave) to become the B operand of dQ^T += K^T dS^T, and the four waves' dQ shares (their own keys)
// meet in LDS once per block: written side by side, summed in wave order by all 256 threads, scaled, rounded and
// stored.  5 products, one softmax pass, every operand read once, no atomics: bitwise reproducible.
// =================================================================================================
constexpr int F96_QB = 32;                                 // queries per step
__host__ __device__ constexpr size_t lds_fused96() {
  // K rows (then the dK / dV images) | the four dQ shares (4 x 32 x 100 f32) | 2 x (Q, dO) blocks | 4 x dS^T |
  // lse, -delta | delta shares
  return (size_t)FUSED_LMAX * Geo<96>::RS + 4 * F96_QB * 100 * 4 + 4 * F96_QB * Geo<96>::RS + 4 * 4096 +
         2 * FUSED_LMAX * 4 + F96_QB * 12 * 4;
}

__global__ __launch_bounds__(256, 1) void attn_bwd_fused96_kernel(const AP p) {
  constexpr int D = 96, RS = Geo<96>::RS, KS = 3, DT = 6, NCH = 12, LQ = FUSED_LMAX, QB = F96_QB, ILD = D + 4;
  constexpr int KTW = 4, KW = 64, NCK = 2;                 // per wave: four 16-key tiles = two 32-key chunks
  constexpr int NKP = LQ * NCH / 256;                      // K-row chunk tasks per thread (12)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ktile = smem;                                      // [256][RS] K rows (transposed fragments); later dK / dV images
  float* share = reinterpret_cast<float*>(smem + LQ * RS); // [4 waves][32 q][ILD] f32
  char* qd = smem + LQ * RS + 4 * QB * ILD * 4;            // [2 buffers][Q block | dO block][32][RS]
  char* dst_all = qd + 4 * QB * RS;                        // [4 waves][2 chunks][32 keys][64 B]
  float* lse_l = reinterpret_cast<float*>(dst_all + 4 * 4096);   // [256] lse * log2(e); +inf past the end
  float* dl_l = lse_l + LQ;                                // [256] -delta
  float* part = dl_l + LQ;                                 // [32][12] delta shares of the block being staged
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int H = p.H;
  const long tokstride = 3L * H * D, ostride = (long)H * D;
  const unsigned int HD = (unsigned int)(H * D);
  const float c2 = p.scale * LOG2E;
  const int nheads = p.B * H;

  // ---- the head whose rows are being LOADED (the next one, from the end of the current sweep on)
  const unsigned short *lq = nullptr, *ldo = nullptr, *lo = nullptr;
  int lL = 1;
  auto point_at = [&](int w, int& L_, long& row0_, int& b_, int& h_) {
    int blk_;
    work_item_at(w, 1, H, p.B, blk_, h_, b_);
    L_ = p.L;
    row0_ = seq_rows(p, b_, L_);
  };
  // Q / dO / O block staging: 32 rows x 12 chunks = 384 (row, chunk) tasks per tensor, two passes of 256 threads (the
  // second pass of threads >= 128 repeats a task of the first; its store is skipped)
  u32x4 cq[2], cd[2], co[2];
  auto issue_block = [&](int j) {
    const int t = tid + opaque_zero();
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      int idl = ps * 256 + t; idl = idl < QB * NCH ? idl : idl - 256;
      const int r = idl / NCH, ch = idl - r * NCH;
      int row = j * QB + r; row = row < lL ? row : lL - 1;
      cq[ps] = *reinterpret_cast<const u32x4*>(lq + ((unsigned int)row * 3u * HD + 8u * ch));
      cd[ps] = *reinterpret_cast<const u32x4*>(ldo + ((unsigned int)row * HD + 8u * ch));
      co[ps] = *reinterpret_cast<const u32x4*>(lo + ((unsigned int)row * HD + 8u * ch));
    }
  };
  auto store_block = [&](int j) {                          // -> buffer j & 1, delta shares -> part
    const int t = tid + opaque_zero();
    char* qt = qd + (j & 1) * (2 * QB * RS);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int idl = ps * 256 + t;
      if (idl < QB * NCH) {
        const int r = idl / NCH, ch = idl - r * NCH;
        *reinterpret_cast<u32x4*>(qt + r * RS + ch * 16) = cq[ps];
        *reinterpret_cast<u32x4*>(qt + QB * RS + r * RS + ch * 16) = cd[ps];
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc += bf16_to_f32(co[ps][e] & 0xffffu) * bf16_to_f32(cd[ps][e] & 0xffffu);
          acc += bf16_to_f32(co[ps][e] >> 16) * bf16_to_f32(cd[ps][e] >> 16);
        }
        part[idl] = acc;
      }
    }
  };
  // K / V fragments of this wave's keys straight from HBM in fragment layout (16-byte chunk g + 4 ks of row li), the
  // K rows for the transposed fragments (chunk per lane, through registers), lse of row tid, the key-mask bias
  bf16x8 kf[KTW][KS], vf[KTW][KS];
  float kbias[KTW], lse_r;
  auto issue_head = [&](int w) {                           // sets lq / ldo / lo / lL to head w and requests its rows
    int L_, b_, h_;
    long row0_;
    point_at(w, L_, row0_, b_, h_);
    if (L_ <= 0) { L_ = 1; row0_ = row0_ > 0 ? row0_ - 1 : 0; }   // empty sequence: an in-range dummy row, never used
    lL = L_;
    lq = p.qkv + row0_ * tokstride + (long)h_ * D;
    ldo = p.dout + row0_ * ostride + (long)h_ * D;
    lo = p.out + row0_ * ostride + (long)h_ * D;
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) {
      const int key = wid * KW + kt * 16 + li;
      const int row = key < L_ ? key : L_ - 1;
      kbias[kt] = (key < L_ && (!p.key_mask || p.key_mask[row0_ + row])) ? 0.f : -INFINITY;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const unsigned int off = (unsigned int)row * 3u * HD + (unsigned int)(ks * 32 + g * 8);
        kf[kt][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lq + (off + HD)));
        vf[kt][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lq + (off + 2u * HD)));
      }
    }
    lse_r = p.lse[stat_at(p, b_, h_, H, L_, row0_, tid < L_ ? tid : L_ - 1)];
    issue_block(0);
  };

  const int trow = 4 * g + (li >> 2), tcolb = 8 * (li & 3);
  char* dst = dst_all + wid * 4096;
  int off_dw[2];                                           // dS^T write: key row li of a 16-key tile, queries qq*16 + 4g ..
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) off_dw[qq] = swz64(li, qq * 2 + (g >> 1)) + 8 * (g & 1);

  // Persistent: one workgroup per CU walks heads w, w + gridDim.x, ...; the rows of the NEXT head are requested right
  // after the sweep of the current one (its K / V fragment registers are dead by then) and land while the dK / dV
  // images are written out.
  int w = blockIdx.x;
  if (w < nheads) issue_head(w);
  for (; w < nheads; w += gridDim.x) {
    int L, b, h;
    long row0;
    point_at(w, L, row0, b, h);
    const int wnext = w + (int)gridDim.x < nheads ? w + (int)gridDim.x : w;   // last head: re-request its own rows
    if (L <= 0) {                                          // (workgroup-uniform) empty sequence of a packed batch
      issue_head(wnext);
      continue;
    }
    const int nblk = (L + QB - 1) / QB;
    const auto finish_delta = [&](int j) {                 // after the barrier that follows store_block(j)
      if (tid < QB) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc += part[tid * NCH + c];
        const int q = j * QB + tid;
        const bool ok = q < L;
        dl_l[q] = ok ? -acc : 0.f;                         // NEGATED: the dP accumulators start from it
        if (ok) p.delta[stat_at(p, b, h, H, L, row0, q)] = acc;
      }
    };
    // ---- K rows -> LDS for the transposed fragments (chunk per lane through registers; L2-hot: the same rows came
    // through as fragments while the previous head was written out), block 0 -> buffer 0, lse
    {
      const int t = tid + opaque_zero();
      const unsigned short* kb_ = p.qkv + row0 * tokstride + (long)h * D + HD;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        u32x4 kr[NKP / 2];
#pragma unroll
        for (int ps = 0; ps < NKP / 2; ++ps) {
          const int idl = (half * (NKP / 2) + ps) * 256 + t;
          const int r = idl / NCH, ch = idl - r * NCH;
          kr[ps] = *reinterpret_cast<const u32x4*>(kb_ + ((unsigned int)(r < L ? r : L - 1) * 3u * HD + 8u * ch));
        }
#pragma unroll
        for (int ps = 0; ps < NKP / 2; ++ps) {
          const int idl = (half * (NKP / 2) + ps) * 256 + t;
          const int r = idl / NCH, ch = idl - r * NCH;
          *reinterpret_cast<u32x4*>(ktile + r * RS + ch * 16) = kr[ps];
        }
      }
    }
    lse_l[tid] = tid < L ? lse_r * LOG2E : INFINITY;       // p = 2^-inf = 0 past the end
    store_block(0);
    __syncthreads();
    finish_delta(0);
    __syncthreads();

    f32x4 dk[DT][KTW], dv[DT][KTW];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) { dk[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = dk[dt][kt]; }
    const bool wave_live = wid * KW < L;                   // this wave owns at least one real key
    unsigned short* dqb = p.dqkv + row0 * tokstride + (long)h * D;

#pragma unroll 1
    for (int j = 0; j < nblk; ++j) {
      if (j + 1 < nblk) issue_block(j + 1);                // (workgroup-uniform branch; the last step requests nothing)
      const char* qt_ = qd + (j & 1) * (2 * QB * RS);
      const char* dt_ = qt_ + QB * RS;
      if (wave_live) {
        u32x2 pk[2][KTW], dsk[2][KTW];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          // rows of the accumulators are queries (4g + r), columns are this lane's key; dP starts at -delta of its row,
          // S at 0 / -inf for a valid / masked key
          const f32x4 nd = *reinterpret_cast<const f32x4*>(dl_l + j * QB + qq * 16 + 4 * g);
          f32x4 s[KTW], dp[KTW];
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) { s[kt] = f32x4{kbias[kt], kbias[kt], kbias[kt], kbias[kt]}; dp[kt] = nd; }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qa = row_frag(qt_, RS, qq * 16 + li, ks, lane);
            const bf16x8 da = row_frag(dt_, RS, qq * 16 + li, ks, lane);
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
              s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kt][ks], s[kt], 0, 0, 0);
              dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[kt], 0, 0, 0);
            }
          }
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_l + j * QB + qq * 16 + 4 * g);
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = fast_exp2(s[kt][r] * c2 - ls[r]);
              s[kt][r] = pv;                               // P
              dp[kt][r] = pv * dp[kt][r];                  // dS = P (dP - delta)
            }
            pk[qq][kt] = u32x2{pack_bf16x2(s[kt][0], s[kt][1]), pack_bf16x2(s[kt][2], s[kt][3])};
            dsk[qq][kt] = u32x2{pack_bf16x2(dp[kt][0], dp[kt][1]), pack_bf16x2(dp[kt][2], dp[kt][3])};
          }
        }
        // dS^T -> the wave's private [key][query] tiles first: the transposed read-back has the dV / dK block to land
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
            *reinterpret_cast<u32x2*>(dst + (kt >> 1) * 2048 + (kt & 1) * 1024 + off_dw[qq]) = dsk[qq][kt];
        // dV^T[d][key] += dO^T[d][32 q] P[32 q][key], dK^T += Q^T dS: ONE transposed fragment pair per d tile for all
        // four key tiles (the fragments do not depend on the key tile)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bf16x8 dot_f = tr_frag(dt_, RS, trow, dt * 32 + tcolb);       // dO^T[d tile][32 q]
          const bf16x8 qt_f = tr_frag(qt_, RS, trow, dt * 32 + tcolb);        // Q^T
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) {
            const bf16x8 pbf = __builtin_bit_cast(bf16x8, u32x4{pk[0][kt][0], pk[0][kt][1], pk[1][kt][0], pk[1][kt][1]});
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dsk[0][kt][0], dsk[0][kt][1], dsk[1][kt][0], dsk[1][kt][1]});
            dv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f, pbf, dv[dt][kt], 0, 0, 0);
            dk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsf, dk[dt][kt], 0, 0, 0);
          }
        }
      }
      // dQ^T[d][q] += K^T[d][32 keys of chunk c] dS^T[32 keys][q], one 16-query tile at a time (24 accumulator registers
      // instead of 48; the wave's own dS^T writes: no barrier needed), then this wave's share of dQ (its keys only) for
      // block j as [q][d] f32 rows (zeros from a wave without keys)
      float* sh = share + wid * (QB * ILD);
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        f32x4 dq[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave_live) {
#pragma unroll
          for (int c = 0; c < NCK; ++c) {
            const bf16x8 dsb = tr_frag_off(dst + c * 2048, swz64(trow, qq * 2 + ((li & 3) >> 1)) + 8 * (li & 1));
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              // K^T[d tile][32 keys of chunk c]: re-read from the staged K rows (12 resident fragments would cost 48
              // registers the sweep does not have)
              const bf16x8 ktf = tr_frag(ktile, RS, wid * KW + c * 32 + trow, dt * 32 + tcolb);
              dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsb, dq[dt], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          *reinterpret_cast<f32x4*>(sh + (qq * 16 + li) * ILD + dt * 16 + 4 * g) = dq[dt];
      }
      if (j + 1 < nblk) store_block(j + 1);                // the other buffer
      __syncthreads();
      // ---- sum the four shares in wave order, scale, round, store: thread t -> query t / 8, 12 head dims
      {
        const int r = tid >> 3, c0 = (tid & 7) * 12;
        f32x4 a[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) a[i] = *reinterpret_cast<const f32x4*>(share + r * ILD + c0 + 4 * i);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww)
#pragma unroll
          for (int i = 0; i < 3; ++i) a[i] += *reinterpret_cast<const f32x4*>(share + ww * (QB * ILD) + r * ILD + c0 + 4 * i);
        const int q = j * QB + r;
        if (q < L) {
          unsigned short* dqrow = dqb + (unsigned int)q * 3u * HD + c0;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            u32x2 w2;
            w2[0] = pack_bf16x2(a[i][0] * p.scale, a[i][1] * p.scale);
            w2[1] = pack_bf16x2(a[i][2] * p.scale, a[i][3] * p.scale);
            *reinterpret_cast<u32x2*>(dqrow + 4 * i) = w2;
          }
        }
      }
      if (j + 1 < nblk) finish_delta(j + 1);
      __syncthreads();
    }

    // ---- the next head's rows: K / V fragment registers are free now; they land under the write-out below
    issue_head(wnext);

    // ---- dK^T (x scale) and dV^T accumulators -> bf16 [key][d] images (one after the other, over the shares) -> rows
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const f32x4 v = which == 0 ? dk[dt][kt] * p.scale : dv[dt][kt];
          u32x2 wv;
          wv[0] = pack_bf16x2(v[0], v[1]);
          wv[1] = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<u32x2*>(ktile + (wid * KW + kt * 16 + li) * RS + (dt * 16 + 4 * g) * 2) = wv;
        }
      __syncthreads();
      {
        const int t = tid + opaque_zero();
#pragma unroll
        for (int ps = 0; ps < NKP; ++ps) {
          const int idl = ps * 256 + t;
          const int r = idl / NCH, ch = idl - r * NCH;
          if (r < L)
            *reinterpret_cast<u32x4*>(dqb + ((unsigned int)r * 3u * HD + (unsigned int)(which + 1) * HD + 8u * ch)) =
                *reinterpret_cast<const u32x4*>(ktile + r * RS + ch * 16);
        }
      }
      __syncthreads();
    }
  }
}

// =================================================================================================
// The same backward with EIGHT waves (two per SIMD, 256 registers each): a wave owns 32 keys - half the fragments and
// accumulators (144 resident registers instead of 288) - so that the second wave of a SIMD fills the first one's
// dependency stalls (the 4-wave kernel keeps a SIMD 0.41 busy: one wave, every LDS / MFMA latency exposed).  Every wave
// reads the Q / dO fragments of the whole 32-query block for its 32 keys (twice the LDS reads per key of the 4-wave
// kernel), so dQ is NOT summed from per-wave shares here: after a barrier the waves compute complete (d tile, query tile)
// outputs over all keys from the eight dS^T tiles (12 tiles: two each for waves 0-3, one each for waves 4-7 = three per
// SIMD).  Same LDS layout as the 4-wave kernel (the shares' region holds one 32-row bf16 image), three barriers per
// step, bitwise reproducible.  Everything else (staging, prefetch of the next head, write-out) as above.
// The default for hd 96 (option attn_fused_waves = 4 keeps the 4-wave kernel): 948 against 1132 us per layer at B = 1024 -
// with per-wave dQ shares added pairwise it was 1173 (profiles/r03/attn96_eight_waves_ab_v1.txt).
// =================================================================================================
__global__ __launch_bounds__(512, 1) void attn_bwd_fused96w8_kernel(const AP p) {
  constexpr int D = 96, RS = Geo<96>::RS, KS = 3, DT = 6, NCH = 12, LQ = FUSED_LMAX, QB = F96_QB, ILD = D + 4;
  constexpr int NT = 512, KTW = 2, KW = 32;                // per wave: two 16-key tiles = one 32-key chunk
  constexpr int NKP = LQ * NCH / NT;                       // K-row chunk tasks per thread (6)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ktile = smem;                                      // [256][RS] K rows (transposed fragments); later dK / dV images
  char* dqimg = smem + LQ * RS;                            // [32 q][RS] bf16 dQ rows of the step (where the 4-wave kernel keeps its shares)
  char* qd = smem + LQ * RS + 4 * QB * ILD * 4;            // [2 buffers][Q block | dO block][32][RS]
  char* dst_all = qd + 4 * QB * RS;                        // [8 waves][32 keys][64 B]
  float* lse_l = reinterpret_cast<float*>(dst_all + 4 * 4096);   // [256] lse * log2(e); +inf past the end
  float* dl_l = lse_l + LQ;                                // [256] -delta
  float* part = dl_l + LQ;                                 // [32][12] delta shares of the block being staged
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int H = p.H;
  const long tokstride = 3L * H * D, ostride = (long)H * D;
  const unsigned int HD = (unsigned int)(H * D);
  const float c2 = p.scale * LOG2E;
  const int nheads = p.B * H;

  const unsigned short *lq = nullptr, *ldo = nullptr, *lo = nullptr;
  int lL = 1;
  auto point_at = [&](int w, int& L_, long& row0_, int& b_, int& h_) {
    int blk_;
    work_item_at(w, 1, H, p.B, blk_, h_, b_);
    L_ = p.L;
    row0_ = seq_rows(p, b_, L_);
  };
  // Q / dO / O block staging: 32 rows x 12 chunks = 384 (row, chunk) tasks per tensor, one per thread (threads >= 384
  // repeat a task; their stores are skipped)
  u32x4 cq, cd, co;
  auto issue_block = [&](int j) {
    const int t = tid + opaque_zero();
    const int idl = t < QB * NCH ? t : t - 256;
    const int r = idl / NCH, ch = idl - r * NCH;
    int row = j * QB + r; row = row < lL ? row : lL - 1;
    cq = *reinterpret_cast<const u32x4*>(lq + ((unsigned int)row * 3u * HD + 8u * ch));
    cd = *reinterpret_cast<const u32x4*>(ldo + ((unsigned int)row * HD + 8u * ch));
    co = *reinterpret_cast<const u32x4*>(lo + ((unsigned int)row * HD + 8u * ch));
  };
  auto store_block = [&](int j) {                          // -> buffer j & 1, delta shares -> part
    const int t = tid + opaque_zero();
    char* qt = qd + (j & 1) * (2 * QB * RS);
    if (t < QB * NCH) {
      const int r = t / NCH, ch = t - r * NCH;
      *reinterpret_cast<u32x4*>(qt + r * RS + ch * 16) = cq;
      *reinterpret_cast<u32x4*>(qt + QB * RS + r * RS + ch * 16) = cd;
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += bf16_to_f32(co[e] & 0xffffu) * bf16_to_f32(cd[e] & 0xffffu);
        acc += bf16_to_f32(co[e] >> 16) * bf16_to_f32(cd[e] >> 16);
      }
      part[t] = acc;
    }
  };
  bf16x8 vf[KTW][KS];
  u32x4 kr[NKP];                                           // the head's K rows, chunk per lane, on their way to LDS
  float kbias[KTW], lse_r;
  auto issue_head = [&](int w) {                           // sets lq / ldo / lo / lL to head w and requests its rows
    int L_, b_, h_;
    long row0_;
    point_at(w, L_, row0_, b_, h_);
    if (L_ <= 0) { L_ = 1; row0_ = row0_ > 0 ? row0_ - 1 : 0; }   // empty sequence: an in-range dummy row, never used
    lL = L_;
    lq = p.qkv + row0_ * tokstride + (long)h_ * D;
    ldo = p.dout + row0_ * ostride + (long)h_ * D;
    lo = p.out + row0_ * ostride + (long)h_ * D;
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) {
      const int key = wid * KW + kt * 16 + li;
      const int row = key < L_ ? key : L_ - 1;
      kbias[kt] = (key < L_ && (!p.key_mask || p.key_mask[row0_ + row])) ? 0.f : -INFINITY;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const unsigned int off = (unsigned int)row * 3u * HD + (unsigned int)(ks * 32 + g * 8);
        vf[kt][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lq + (off + 2u * HD)));
      }
    }
    const int lr = tid < LQ ? tid : LQ - 1;
    lse_r = p.lse[stat_at(p, b_, h_, H, L_, row0_, lr < L_ ? lr : L_ - 1)];
    // the K rows for the transposed (and, in this kernel, the row) fragments: requested HERE, with the rest of the head,
    // so that they land under the previous head's write-out instead of at the top of this one
    {
      const int t = tid + opaque_zero();
#pragma unroll
      for (int ps = 0; ps < NKP; ++ps) {
        const int idl = ps * NT + t;
        const int r = idl / NCH, ch = idl - r * NCH;
        kr[ps] = *reinterpret_cast<const u32x4*>(lq + HD + ((unsigned int)(r < L_ ? r : L_ - 1) * 3u * HD + 8u * ch));
      }
    }
    issue_block(0);
  };

  const int trow = 4 * g + (li >> 2), tcolb = 8 * (li & 3);
  char* dst = dst_all + wid * 2048;
  int off_dw[2];                                           // dS^T write: key row li of a 16-key tile, queries qq*16 + 4g ..
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) off_dw[qq] = swz64(li, qq * 2 + (g >> 1)) + 8 * (g & 1);

  int w = blockIdx.x;
  if (w < nheads) issue_head(w);
#ifdef CLIPK_ATTN_TRACE
  const bool tr_on = blockIdx.x == 0 && tid == 0 && g_attn_trace != nullptr;
  unsigned long long* tr_lds = reinterpret_cast<unsigned long long*>(smem + lds_fused96());
  if (tid < 16) tr_lds[tid] = 0;
  __syncthreads();
  unsigned long long tr_t = __builtin_amdgcn_s_memtime();
  const unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime(), tr_c0 = tr_t;
#endif
  for (; w < nheads; w += gridDim.x) {
    int L, b, h;
    long row0;
    point_at(w, L, row0, b, h);
    ATTN_STAMP(7);                                         // (loop bookkeeping)
    const int wnext = w + (int)gridDim.x < nheads ? w + (int)gridDim.x : w;   // last head: re-request its own rows
    if (L <= 0) {                                          // (workgroup-uniform) empty sequence of a packed batch
      issue_head(wnext);
      continue;
    }
    const int nblk = (L + QB - 1) / QB;
    const auto finish_delta = [&](int j) {                 // after the barrier that follows store_block(j)
      if (tid < QB) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc += part[tid * NCH + c];
        const int q = j * QB + tid;
        const bool ok = q < L;
        dl_l[q] = ok ? -acc : 0.f;                         // NEGATED: the dP accumulators start from it
        if (ok) p.delta[stat_at(p, b, h, H, L, row0, q)] = acc;
      }
    };
    // ---- K rows -> LDS for the transposed fragments, block 0 -> buffer 0, lse
    {
      const int t = tid + opaque_zero();
#pragma unroll
      for (int ps = 0; ps < NKP; ++ps) {
        const int idl = ps * NT + t;
        const int r = idl / NCH, ch = idl - r * NCH;
        *reinterpret_cast<u32x4*>(ktile + r * RS + ch * 16) = kr[ps];
      }
    }
    if (tid < LQ) lse_l[tid] = tid < L ? lse_r * LOG2E : INFINITY;       // p = 2^-inf = 0 past the end
    store_block(0);
    __syncthreads();
    finish_delta(0);
    __syncthreads();
    ATTN_STAMP(0);                                         // wait for the head's rows, K rows -> LDS, block 0, delta (2 barriers)

    f32x4 dk[DT][KTW], dv[DT][KTW];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt) { dk[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt][kt] = dk[dt][kt]; }
    const bool wave_live = wid * KW < L;                   // this wave owns at least one real key
    unsigned short* dqb = p.dqkv + row0 * tokstride + (long)h * D;

#pragma unroll 1
    for (int j = 0; j < nblk; ++j) {
      // UNCONDITIONAL (the last step re-requests its own block and stages it into the idle buffer): with `if (j + 1 <
      // nblk)` around the request and around the staging hipcc cannot pair the two branches, assumes the loads may
      // still be in flight at the next request and waits vmcnt(0) there - for the block it has just asked for
      issue_block(j + 1 < nblk ? j + 1 : j);
      const char* qt_ = qd + (j & 1) * (2 * QB * RS);
      const char* dt_ = qt_ + QB * RS;
      u32x2 pk[2][KTW], dsk[2][KTW];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) { pk[qq][kt] = u32x2{0u, 0u}; dsk[qq][kt] = pk[qq][kt]; }
      if (wave_live) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const f32x4 nd = *reinterpret_cast<const f32x4*>(dl_l + j * QB + qq * 16 + 4 * g);
          f32x4 s[KTW], dp[KTW];
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) { s[kt] = f32x4{kbias[kt], kbias[kt], kbias[kt], kbias[kt]}; dp[kt] = nd; }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 qa = row_frag(qt_, RS, qq * 16 + li, ks, lane);
            const bf16x8 da = row_frag(dt_, RS, qq * 16 + li, ks, lane);
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
              s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, row_frag(ktile, RS, wid * KW + kt * 16 + li, ks, lane), s[kt], 0, 0, 0);
              dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf[kt][ks], dp[kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);             // (fragments of one k-step at a time: the SIMD's other wave hides the reads)
          }
          const f32x4 ls = *reinterpret_cast<const f32x4*>(lse_l + j * QB + qq * 16 + 4 * g);
#pragma unroll
          for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pv = fast_exp2(s[kt][r] * c2 - ls[r]);
              s[kt][r] = pv;                               // P
              dp[kt][r] = pv * dp[kt][r];                  // dS = P (dP - delta)
            }
            pk[qq][kt] = u32x2{pack_bf16x2(s[kt][0], s[kt][1]), pack_bf16x2(s[kt][2], s[kt][3])};
            dsk[qq][kt] = u32x2{pack_bf16x2(dp[kt][0], dp[kt][1]), pack_bf16x2(dp[kt][2], dp[kt][3])};
          }
          __builtin_amdgcn_sched_barrier(0);               // one 16-query tile at a time: the second one's 16 accumulator
        }                                                  // registers reuse the first one's
      }
      // dS^T -> the wave's private [32 keys][32 queries] tile (zeros from a wave without keys: the dQ tiles below read
      // all eight tiles)
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
          *reinterpret_cast<u32x2*>(dst + kt * 1024 + off_dw[qq]) = dsk[qq][kt];
      // dV^T[d][key] += dO^T[d][32 q] P[32 q][key], dK^T += Q^T dS
      const auto dvdk = [&]() {
        if (wave_live) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const bf16x8 dot_f = tr_frag(dt_, RS, trow, dt * 32 + tcolb);     // dO^T[d tile][32 q]
            const bf16x8 qt_f = tr_frag(qt_, RS, trow, dt * 32 + tcolb);      // Q^T
#pragma unroll
            for (int kt = 0; kt < KTW; ++kt) {
              const bf16x8 pbf = __builtin_bit_cast(bf16x8, u32x4{pk[0][kt][0], pk[0][kt][1], pk[1][kt][0], pk[1][kt][1]});
              const bf16x8 dsf = __builtin_bit_cast(bf16x8, u32x4{dsk[0][kt][0], dsk[0][kt][1], dsk[1][kt][0], dsk[1][kt][1]});
              dv[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_f, pbf, dv[dt][kt], 0, 0, 0);
              dk[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_f, dsf, dk[dt][kt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
      __builtin_amdgcn_sched_barrier(0);
      dvdk();
      __builtin_amdgcn_sched_barrier(0);
      store_block(j + 1);                                  // the other buffer
      ATTN_STAMP(1);                                       // step: S / dP, softmax, dS^T, dV / dK, staging of the next block
      __syncthreads();                                     // every wave's dS^T tile is written
      ATTN_STAMP(4);                                       // step: the barrier after it
      // dQ^T = K^T dS^T as COMPLETE (d tile, 16-query tile) outputs over all keys: waves 0-3 take d tiles 0-3 (both query
      // tiles), waves 4-7 d tiles 4 / 5 (one query tile each) - three tiles per SIMD - reading the K^T fragments of the
      // staged K rows and the dS^T tile of the wave that owns each 32-key chunk.  No shares, no sums over waves: the
      // accumulator IS the gradient; it goes (scaled, bf16) into a [32 q][RS] image and leaves as whole rows.
      {
        const int dt0 = wid < 4 ? wid : 4 + ((wid - 4) >> 1);
        const int nq = wid < 4 ? 2 : 1, q0t = wid < 4 ? 0 : (wid & 1);
        f32x4 dq[2];
        dq[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[1] = dq[0];
        // all eight 32-key chunks, unrolled (a chunk past the end multiplies clamped K rows by a zero dS^T tile): as a
        // run-time loop over the live chunks every fragment read waited out its own LDS latency - 3160 cycles per step
#pragma unroll
        for (int c = 0; c < LQ / 32; ++c) {
          const bf16x8 ktf = tr_frag(ktile, RS, c * 32 + trow, dt0 * 32 + tcolb);
          const char* dsc = dst_all + c * 2048;
          const bf16x8 dsb0 = tr_frag_off(dsc, swz64(trow, q0t * 2 + ((li & 3) >> 1)) + 8 * (li & 1));
          dq[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsb0, dq[0], 0, 0, 0);
          if (nq == 2) {                                   // (wave-uniform)
            const bf16x8 dsb1 = tr_frag_off(dsc, swz64(trow, 2 + ((li & 3) >> 1)) + 8 * (li & 1));
            dq[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsb1, dq[1], 0, 0, 0);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (t < nq) {
            const f32x4 v = dq[t] * p.scale;
            u32x2 wv;
            wv[0] = pack_bf16x2(v[0], v[1]);
            wv[1] = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<u32x2*>(dqimg + ((q0t + t) * 16 + li) * RS + (dt0 * 16 + 4 * g) * 2) = wv;
          }
      }
      __syncthreads();
      // ---- dQ rows: thread t < 384 -> query t / 12, one 16-byte chunk
      if (tid < QB * NCH) {
        const int r = tid / NCH, ch = tid - r * NCH;
        const int q = j * QB + r;
        if (q < L)
          *reinterpret_cast<u32x4*>(dqb + (unsigned int)q * 3u * HD + 8 * ch) = *reinterpret_cast<const u32x4*>(dqimg + r * RS + ch * 16);
      }
      if (j + 1 < nblk) finish_delta(j + 1);
      __syncthreads();
      ATTN_STAMP(2);                                       // step: dQ tiles, barrier, dQ rows, delta, barrier
    }

    // ---- the next head's rows: K / V fragment registers are free now; they land under the write-out below
    issue_head(wnext);
    ATTN_STAMP(3);                                         // issuing the next head's loads

    // ---- dK^T (x scale) and dV^T accumulators -> bf16 [key][d] images (one after the other, over the K rows) -> rows
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const f32x4 v = which == 0 ? dk[dt][kt] * p.scale : dv[dt][kt];
          u32x2 wv;
          wv[0] = pack_bf16x2(v[0], v[1]);
          wv[1] = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<u32x2*>(ktile + (wid * KW + kt * 16 + li) * RS + (dt * 16 + 4 * g) * 2) = wv;
        }
      __syncthreads();
      {
        const int t = tid + opaque_zero();
#pragma unroll
        for (int ps = 0; ps < NKP; ++ps) {
          const int idl = ps * NT + t;
          const int r = idl / NCH, ch = idl - r * NCH;
          if (r < L)
            *reinterpret_cast<u32x4*>(dqb + ((unsigned int)r * 3u * HD + (unsigned int)(which + 1) * HD + 8u * ch)) =
                *reinterpret_cast<const u32x4*>(ktile + r * RS + ch * 16);
        }
      }
      __syncthreads();
    }
    ATTN_STAMP(5);                                         // dK / dV images, barriers, row stores
#ifdef CLIPK_ATTN_TRACE
    if (tr_on) atomicAdd(&tr_lds[8], 1ull);
#endif
  }
#ifdef CLIPK_ATTN_TRACE
  if (tr_on) {
    for (int i = 0; i < 9; ++i) g_attn_trace[i] = tr_lds[i];
    g_attn_trace[9] = __builtin_amdgcn_s_memtime() - tr_c0;
    g_attn_trace[10] = __builtin_amdgcn_s_memrealtime() - tr_r0;
  }
#endif
}

// =================================================================================================
// forward, whole head in one workgroup (head dim <= 32, 128 < L <= 256, rows that need no rotation): all 256 K and V
// rows are staged ONCE per head (the general kernel's two 128-query workgroups each stage all of them) and stay in
// LDS while the four waves take two passes of 32 queries each.  Same tiles, same 64-key sub-block order and the
// same arithmetic per query as attn_fwd_kernel, so outputs and LSE are bit-identical to it.
// =================================================================================================
// ROT: q / k arrive UNROTATED together with the RoPE tables; the kernel rotates them (one thread per row, in
// registers, the same rope_regs as clipk_rope_qk: bit-identical values), uses them, and writes the rotated rows back
// in place - one workgroup owns every row of its head, so nobody else ever reads them unrotated.  That is
// clipk_rope_qk + clipk_attn_fwd in one pass over q / k (clipk_attn_fwd_rot).
template <int D, bool ROT>
__global__ __launch_bounds__(256, ROT ? 3 : 4) void attn_fwd_whole32_kernel(const AP p) {
  constexpr int DT = 2, LQ = FUSED_LMAX, cpr = D / 8, dtv = (D + 15) >> 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ktile = smem;                                     // [256][64 B], 16-byte chunks swizzled by (row >> 1) & 3
  char* vtile = smem + LQ * 64;
  unsigned char* mask_l = reinterpret_cast<unsigned char*>(smem + 2 * LQ * 64);   // [256]
  char* qtile = smem + 2 * LQ * 64 + 256;                 // ROT only: [256][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int H = p.H;
  int blk, h, b;
  work_item(1, H, p.B, blk, h, b);
  int L = p.L;
  const long row0 = seq_rows(p, b, L);                     // packed batch: this sequence's rows [row0, row0 + L)
  if (L <= 0) return;                                      // (workgroup-uniform)
  const float c2 = p.scale * LOG2E;
  u32x4 qfr[2][2];
  {
    // four lanes per token row, one 16-byte chunk each (the pad chunk re-reads the last real one and is dropped)
    const int ci = tid & 3, r0 = tid >> 2;
    const unsigned int HD = (unsigned int)(H * D), cc = 8u * (ci < cpr ? ci : cpr - 1);
    const unsigned short* qb = p.qkv + row0 * 3 * (long)HD + (long)h * D;
    u32x4 ck[4], cv[4];
#ifdef CLIPK_ATTN_HM_PROBE
    const unsigned short* qhm = p.qkv + (long)(b * H + h) * 3 * p.L * D;      // timing only, see attn_bwd_fused32_kernel
    const unsigned int LDm = (unsigned int)(p.L * D);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      int row = ps * 64 + r0; row = row < L ? row : L - 1;
      const unsigned int ho = (unsigned int)row * D + cc;
      if (!ROT) ck[ps] = *reinterpret_cast<const u32x4*>(qhm + (ho + LDm));
      cv[ps] = *reinterpret_cast<const u32x4*>(qhm + (ho + 2u * LDm));
    }
#else
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      int row = ps * 64 + r0; row = row < L ? row : L - 1;
      const unsigned int qo = (unsigned int)row * 3u * HD + cc;
      if (!ROT) ck[ps] = *reinterpret_cast<const u32x4*>(qb + (qo + HD));
      cv[ps] = *reinterpret_cast<const u32x4*>(qb + (qo + 2u * HD));
    }
#endif
    RowRegs<4> rk, rq;                                     // ROT: thread t owns row t of K and of Q
    const int pc = tid < L ? tid : L - 1;
    if (ROT) {
      load_row<4>(rk, qb + (unsigned int)pc * 3u * HD + HD, cpr);
      load_row<4>(rq, qb + (unsigned int)pc * 3u * HD, cpr);
    } else {
      // Q never touches LDS: lane (li, g) reads chunk g of query row li of its tile - the MFMA fragment as it is
#pragma unroll
      for (int qh = 0; qh < 2; ++qh)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          int row = qh * 128 + wid * 32 + qt * 16 + li; row = row < L ? row : L - 1;
#ifdef CLIPK_ATTN_HM_PROBE
          if (!ROT) qfr[qh][qt] = *reinterpret_cast<const u32x4*>(qhm + ((unsigned int)row * D + 8u * (g < cpr ? g : cpr - 1)));
          else
#endif
          qfr[qh][qt] = *reinterpret_cast<const u32x4*>(qb + ((unsigned int)row * 3u * HD + 8u * (g < cpr ? g : cpr - 1)));
        }
    }
    mask_l[tid] = (tid < L && (!p.key_mask || p.key_mask[row0 + tid])) ? 1 : 0;
    const bool pad = cpr < 4 && ci >= cpr;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int off = swz64(ps * 64 + r0, ci);
      if (!ROT) *reinterpret_cast<u32x4*>(ktile + off) = pad ? z : ck[ps];
      *reinterpret_cast<u32x4*>(vtile + off) = pad ? z : cv[ps];
    }
    if (ROT) {
      rope_regs<D, 4>(rk, p.cosT + (long)pc * (D / 2), p.sinT + (long)pc * (D / 2));
      rope_regs<D, 4>(rq, p.cosT + (long)pc * (D / 2), p.sinT + (long)pc * (D / 2));
#pragma unroll
      for (int i = 0; i < 4; ++i) {                        // chunks cpr..3 are the zero pad (load_row)
        *reinterpret_cast<u32x4*>(ktile + swz64(tid, i)) = rk.c[i];
        *reinterpret_cast<u32x4*>(qtile + swz64(tid, i)) = rq.c[i];
      }
      if (!p.row_stores && tid < L) {                      // (kept for A/B: option attn_row_stores = 0)
        unsigned short* wq = const_cast<unsigned short*>(qb) + (unsigned int)tid * 3u * HD;
#pragma unroll
        for (int i = 0; i < cpr; ++i) {
          *reinterpret_cast<u32x4*>(wq + 8 * i) = rq.c[i];
          *reinterpret_cast<u32x4*>(wq + HD + 8 * i) = rk.c[i];
        }
      }
    } else if (cpr < 4 && g >= cpr) {
#pragma unroll
      for (int qh = 0; qh < 2; ++qh) { qfr[qh][0] = z; qfr[qh][1] = z; }
    }
  }
  __syncthreads();
  if (ROT && p.row_stores) {
    // (option attn_row_stores & 2, off by default) the rotated rows go back in place from the LDS tiles, four lanes to a
    // row.  Measured SLOWER than the owner's stores before the barrier (526 vs 498 us at B = 1024): those overlap the rest
    // of the staging, these delay the sweep
    const int ci = tid & 3, r0 = tid >> 2;
    if (ci < cpr) {
      unsigned short* qw = const_cast<unsigned short*>(p.qkv) + row0 * 3 * (long)(H * D) + (long)h * D;
      const unsigned int HD = (unsigned int)(H * D);
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int row = ps * 64 + r0;
        if (row < L) {
          *reinterpret_cast<u32x4*>(qw + ((unsigned int)row * 3u * HD + 8u * ci)) = *reinterpret_cast<const u32x4*>(qtile + swz64(row, ci));
          *reinterpret_cast<u32x4*>(qw + ((unsigned int)row * 3u * HD + HD + 8u * ci)) = *reinterpret_cast<const u32x4*>(ktile + swz64(row, ci));
        }
      }
    }
  }

  const int trow = 4 * g + (li >> 2);
  const int off_rf = swz64(li, g);
  int off_tr[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) off_tr[dt] = swz64(trow, dt * 2 + ((li & 3) >> 1)) + 8 * (li & 1);

#pragma unroll 1
  for (int qh = 0; qh < 2; ++qh) {
    const int q0 = qh * 128;
    if (q0 >= L) break;
    bf16x8 qf[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (ROT) qf[qt] = *reinterpret_cast<const bf16x8*>(qtile + (q0 + wid * 32 + qt * 16) * 64 + swz64(li, g));
      else qf[qt] = __builtin_bit_cast(bf16x8, qh == 0 ? qfr[0][qt] : qfr[1][qt]);
    }
    f32x4 o[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { o[dt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[dt][1] = o[dt][0]; }
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll 1
    for (int sub = 0; sub < LQ / 64; ++sub) {
      if (sub * 64 >= L) break;                                  // wave-uniform: nothing valid in this sub-block
      const char* kt_ = ktile + sub * 64 * 64;
      const char* vt_ = vtile + sub * 64 * 64;
      f32x4 s[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kt_ + kt * 16 * 64 + off_rf);
        s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
      if (p.key_mask != nullptr || sub * 64 + 64 > L) {          // wave-uniform: full, unmasked blocks skip this
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const unsigned int mk = *reinterpret_cast<const unsigned int*>(mask_l + sub * 64 + kt * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!((mk >> (8 * r)) & 0xffu)) { s[kt][0][r] = -INFINITY; s[kt][1][r] = -INFINITY; }
        }
      }
      bf16x8 pb[2][2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
        mx = group_max(mx);
        const float m_new = fmaxf(m_run[qt], mx);
        const bool dead = (m_new == -INFINITY);
        const float alpha = dead ? 1.f : fast_exp2((m_run[qt] - m_new) * c2);
        const f32x2 mc = splat2(dead ? -INFINITY : -m_new * c2);          // see attn_fwd_kernel
        const f32x2 cc = splat2(c2);
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 a = __builtin_elementwise_fma(f32x2{s[kt][qt][r], s[kt][qt][r + 1]}, cc, mc);
            const f32x2 pv = {fast_exp2(a[0]), fast_exp2(a[1])};
            s[kt][qt][r] = pv[0]; s[kt][qt][r + 1] = pv[1];
            ls2 += pv;
          }
        const float ls = ls2[0] + ls2[1];
        l_run[qt] = l_run[qt] * alpha + ls;
        m_run[qt] = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt][qt] *= alpha;
        pb[qt][0] = pack_acc_pair(s[0][qt], s[1][qt]);
        pb[qt][1] = pack_acc_pair(s[2][qt], s[3][qt]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
          if (dt < dtv) {
            const bf16x8 vf = tr_frag_off(vt_ + s2 * 32 * 64, off_tr[dt]);
            o[dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[0][s2], o[dt][0], 0, 0, 0);
            o[dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[1][s2], o[dt][1], 0, 0, 0);
          }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float lt = group_sum(l_run[qt]);
      const float inv = lt > 0.f ? 1.0f / lt : 0.f;
      const int q = q0 + wid * 32 + qt * 16 + li;
      if (q < L) {
        if (g == 0) p.lse[stat_at(p, b, h, H, L, row0, q)] = lt > 0.f ? m_run[qt] * p.scale + logf(lt) : -INFINITY;
        unsigned short* orow = p.out + (row0 + q) * ((long)H * D) + (long)h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d = dt * 16 + 4 * g;
          if (d < D) {
            u32x2 w;
            w[0] = pack_bf16x2(o[dt][qt][0] * inv, o[dt][qt][1] * inv);
            w[1] = pack_bf16x2(o[dt][qt][2] * inv, o[dt][qt][3] * inv);
            *reinterpret_cast<u32x2*>(orow + d) = w;
          }
        }
      }
    }
  }
}

int attn_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int DP> constexpr size_t lds_fwd() { return (size_t)2 * Geo<DP>::KVB * Geo<DP>::RS + 256; }
template <int DP> constexpr size_t lds_dq() {
  const size_t rows = 256 * (size_t)Geo<DP>::RS + 128 * 4 + 256 + 128 * (size_t)Geo<DP>::NCH * 4;   // + delta shares
  const size_t img = 128 * (size_t)(DP + 4) * 4;
  return rows > img ? rows : img;
}
template <int DP> constexpr size_t lds_dkv() {
  const size_t rows = (size_t)2 * Geo<DP>::KVB * Geo<DP>::RS + 2 * Geo<DP>::KVB * 4;
  const size_t img = (size_t)Geo<DP>::KVB * (DP + 4) * 4;
  return rows > img ? rows : img;
}

template <int D, bool ROT>
void launch_fwd_whole(const AP& p, hipStream_t st) {
  constexpr int lds = (ROT ? 3 : 2) * FUSED_LMAX * 64 + 256;
  AP q = p;
  q.row_stores = (clipk_opt_get(OPT_ATTN_ROW_STORES) & 2) != 0;   // forward: measured slower (526 vs 498 us), off by default
  hipLaunchKernelGGL((attn_fwd_whole32_kernel<D, ROT>), dim3(p.H * p.B), dim3(256), lds, st, q);
}
inline bool whole_fwd_applies(int L, int D) {
  return clipk_opt_get(OPT_ATTN_WHOLE_FWD) != 0 && L > 128 && L <= FUSED_LMAX && (D == 16 || D == 24 || D == 32);
}

template <int DP, int DR, int DX>
int launch_fwd(const AP& p, hipStream_t st) {
  if constexpr (DP == 32 && DR == 0) {
    // short heads whose rows need no rotation: whole-head kernel (option attn_whole_fwd = 0: the general one)
    if (!p.drop_thr && whole_fwd_applies(p.L, p.D)) {        // (packed batches too: per-sequence rows from cu_seqlens)
      switch (p.D) {
        case 16: launch_fwd_whole<16, false>(p, st); break;
        case 24: launch_fwd_whole<24, false>(p, st); break;
        default: launch_fwd_whole<32, false>(p, st); break;
      }
      return clipk_check_launch();
    }
  }
  constexpr size_t lds = lds_fwd<DP>();
  dim3 grid(((p.L + 127) / 128) * p.H * p.B);
  if (p.drop_thr) {
    if constexpr (DR == 0) {                                 // dropout belongs to the post-LN (no RoPE) encoder layers
      if (lds > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<DP, DR, DX, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((attn_fwd_kernel<DP, DR, DX, true>), grid, dim3(256), lds, st, p);
      return clipk_check_launch();
    } else {
      return CLIPK_ERR_UNSUPPORTED;
    }
  }
  if (lds > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<DP, DR, DX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((attn_fwd_kernel<DP, DR, DX>), grid, dim3(256), lds, st, p);
  return clipk_check_launch();
}

template <bool ROPE, int D, int NW>
void launch_fused_nw(const AP& p, hipStream_t st) {
  constexpr size_t lds = lds_fused(D) + FUSED_TRACE_LDS;
  static std::atomic<uint64_t> attr_set{0};
  clipk_once_per_device(attr_set, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused32_kernel<ROPE, D, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  int nwg = (NW == 4 ? 2 : 1) * attn_cu_count();          // persistent: the resident workgroups walk the heads
  if (nwg > p.H * p.B) nwg = p.H * p.B;
  hipLaunchKernelGGL((attn_bwd_fused32_kernel<ROPE, D, NW>), dim3(nwg), dim3(64 * NW), lds, st, p);
}
template <bool ROPE, int D>
void launch_fused(const AP& p, hipStream_t st) {
  // 4 waves (default) or 8 (option attn_fused_waves = 8).  Measured, ESM-2-35M shape, warm, interleaved rounds: 435 us
  // with 4 waves, 585 us with 8 - the gather does run under the sweep, but with ONE workgroup per CU every barrier
  // (eight per head in the sweep, six around it) stalls the whole CU and the steps are half as long.
  if (clipk_opt_get(OPT_ATTN_FUSED_WAVES) == 8) launch_fused_nw<ROPE, D, 8>(p, st);   // (0 = auto: four waves here)
  else launch_fused_nw<ROPE, D, 4>(p, st);
}

inline void launch_fused96(const AP& p, hipStream_t st) {
  constexpr size_t lds = lds_fused96();
  static_assert(lds <= 160 * 1024, "fused96 LDS budget");
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused96_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int nheads = p.B * p.H, cus = attn_cu_count();
  if (clipk_opt_get(OPT_ATTN_FUSED_WAVES) != 4) {             // default: eight waves of 32 keys (948 vs 1132 us per layer at B = 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused96w8_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + FUSED_TRACE_LDS));
    hipLaunchKernelGGL(attn_bwd_fused96w8_kernel, dim3(nheads < cus ? nheads : cus), dim3(512), lds + FUSED_TRACE_LDS, st, p);
    return;
  }
  hipLaunchKernelGGL(attn_bwd_fused96_kernel, dim3(nheads < cus ? nheads : cus), dim3(256), lds, st, p);
}

template <int DP, int DR, int DX>
int launch_bwd(const AP& p, hipStream_t st) {
  if constexpr (DP == 96 && DR == 0 && DX == 96) {
    // whole-head kernel for the 6 x 768 RNA encoder's heads (8 x 96) at L <= 256; option attn_fused_bwd = 0 keeps the
    // dQ + dK/dV pair (tests compare the two)
    if (!p.drop_thr && clipk_opt_get(OPT_ATTN_FUSED_BWD) != 0 && p.L > 128 && p.L <= FUSED_LMAX) {
      launch_fused96(p, st);
      return clipk_check_launch();
    }
  }
  if constexpr (DP == 32) {
    // whole-head kernel for the short-head encoders (ESM-2 8M / 35M / 150M at L <= 256); option attn_fused_bwd = 0
    // keeps the two-kernel path (tests compare the two)
    const bool fused_on = clipk_opt_get(OPT_ATTN_FUSED_BWD) != 0;
    // q / k must arrive rotated (clipk_rope_qk) or unrotated-by-design: the chunk-per-lane staging cannot rotate
    // (packed batches too: the kernel takes each head's rows and length from cu_seqlens; p.L = the longest sequence)
    if (!p.drop_thr && fused_on && p.L > 128 && p.L <= FUSED_LMAX && p.D >= 16 && (DR == 0 || p.pre_rot)) {
      switch (p.D) {
        case 16: launch_fused<(DR > 0), 16>(p, st); break;
        case 24: launch_fused<(DR > 0), 24>(p, st); break;
        default: launch_fused<(DR > 0), 32>(p, st); break;
      }
      return clipk_check_launch();
    }
  }
  constexpr size_t l1 = lds_dq<DP>(), l2 = lds_dkv<DP>();
  constexpr int KPB = Geo<DP>::KPB;
  dim3 gq(((p.L + 127) / 128) * p.H * p.B);
  dim3 gk(((p.L + KPB - 1) / KPB) * p.H * p.B);
  if (p.drop_thr) {
    if constexpr (DR == 0) {
      if (l1 > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<DP, DR, DX, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
      if (l2 > 65536)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<DP, DR, DX, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
      hipLaunchKernelGGL((attn_bwd_dq_kernel<DP, DR, DX, true>), gq, dim3(256), l1, st, p);
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP, DR, DX, true>), gk, dim3(256), l2, st, p);
      return clipk_check_launch();
    } else {
      return CLIPK_ERR_UNSUPPORTED;
    }
  }
  if (l1 > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<DP, DR, DX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
  if (l2 > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<DP, DR, DX>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<DP, DR, DX>), gq, dim3(256), l1, st, p);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP, DR, DX>), gk, dim3(256), l2, st, p);
  return clipk_check_launch();
}

// In-place rotate-half RoPE of the q and k sections of qkv [B*L, 3*H*D]: one thread per (token, q|k, head) row.
// The three attention kernels stage every K row 5 times and every Q row 4 times per layer (two query-block
// workgroups per head in the forward and dQ kernels, two key-block workgroups in the dK/dV kernel) and rotated it
// each time: ~160 VALU per row next to a ~150-instruction inner loop per 64 keys.  Rotating once after the qkv
// GEMM makes those stagings plain copies.
template <int D>
__global__ __launch_bounds__(256) void rope_qk_kernel(unsigned short* qkv, const float* cosT, const float* sinT,
                                                      long rows, int L, int H) {
  constexpr int NCH = D / 8;
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const long t = i / (2 * H);
  const int r = (int)(i - t * 2 * H);                      // r < H: q head r;  r >= H: k head r - H (adjacent in memory)
  unsigned short* rowp = qkv + t * 3L * H * D + (long)r * D;
  const int pos = (int)(t % L);
  RowRegs<NCH> rr;
  load_row<NCH>(rr, rowp, NCH);
  rope_regs<D, NCH>(rr, cosT + (long)pos * (D / 2), sinT + (long)pos * (D / 2));
#pragma unroll
  for (int c = 0; c < NCH; ++c) *reinterpret_cast<u32x4*>(rowp + 8 * c) = rr.c[c];
}

int check_common(const void* qkv, int B, int L, int H, int D, bool rope) {
  if (!qkv || B <= 0 || L <= 0 || H <= 0 || D <= 0) return CLIPK_ERR_BAD_ARG;
  if ((D & 7) || D > 160) return CLIPK_ERR_UNSUPPORTED;
  // RoPE runs on register-resident rows with compile-time indices: the ESM-2 head dims
  if (rope && !(D == 16 || D == 24 || D == 32 || D == 64 || D == 128)) return CLIPK_ERR_UNSUPPORTED;
  if (!aligned16(qkv)) return CLIPK_ERR_BAD_ARG;
  return CLIPK_OK;
}

}  // namespace

static inline int set_dropout(AP& p, float dropout_p, unsigned seed) {
  if (!(dropout_p >= 0.f) || dropout_p >= 1.f) return CLIPK_ERR_BAD_ARG;
  p.drop_epoch = clipk_drop_epoch();
  if (dropout_p == 0.f) { p.drop_thr = 0; p.drop_seed = 0; p.drop_scale = 1.f; return CLIPK_OK; }
  double t = (double)dropout_p * 4294967296.0;
  p.drop_thr = t < 1.0 ? 1u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
  p.drop_seed = seed;
  p.drop_scale = 1.0f / (1.0f - dropout_p);
  return CLIPK_OK;
}

// DX = the head dim when it is one the dispatcher can name at compile time (every RoPE case; multiples of 32
// otherwise), 0 = run time
#define ATTN_DISPATCH(FN, D, ROPE, P, ST)                                   \
  do {                                                                      \
    if (ROPE) {                                                             \
      switch (D) {                                                          \
        case 16: return FN<32, 16, 16>(P, ST);                              \
        case 24: return FN<32, 24, 24>(P, ST);                              \
        case 32: return FN<32, 32, 32>(P, ST);                              \
        case 64: return FN<64, 64, 64>(P, ST);                              \
        default: return FN<128, 128, 128>(P, ST);                           \
      }                                                                     \
    }                                                                       \
    switch (D) {                                                            \
      case 24: return FN<32, 0, 24>(P, ST);                                 \
      case 32: return FN<32, 0, 32>(P, ST);                                 \
      case 64: return FN<64, 0, 64>(P, ST);                                 \
      case 96: return FN<96, 0, 96>(P, ST);                                 \
      case 128: return FN<128, 0, 128>(P, ST);                              \
      case 160: return FN<160, 0, 160>(P, ST);                              \
      default: break;                                                       \
    }                                                                       \
    if ((D) <= 32) return FN<32, 0, 0>(P, ST);                              \
    if ((D) <= 64) return FN<64, 0, 0>(P, ST);                              \
    if ((D) <= 96) return FN<96, 0, 0>(P, ST);                              \
    if ((D) <= 128) return FN<128, 0, 0>(P, ST);                            \
    return FN<160, 0, 0>(P, ST);                                            \
  } while (0)

extern "C" int clipk_attn_fwd(const void* qkv, const uint8_t* key_mask, const float* rope_cos, const float* rope_sin,
                              void* out, float* lse, int B, int L, int H, int D, float q_scale, float dropout_p,
                              uint32_t dropout_seed, void* stream) {
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return CLIPK_ERR_BAD_ARG;
  const bool rope = rope_cos != nullptr;
  int rc = check_common(qkv, B, L, H, D, rope);
  if (rc) return rc;
  if (!out || !lse || !aligned16(out)) return CLIPK_ERR_BAD_ARG;
  AP p{};
  p.qkv = (const unsigned short*)qkv; p.key_mask = key_mask; p.cosT = rope_cos; p.sinT = rope_sin;
  p.out = (unsigned short*)out; p.lse = lse; p.B = B; p.L = L; p.H = H; p.D = D; p.scale = q_scale;
  rc = set_dropout(p, dropout_p, dropout_seed);
  if (rc) return rc;
  ATTN_DISPATCH(launch_fwd, D, rope, p, (hipStream_t)stream);
}

extern "C" int clipk_rope_qk(void* qkv, const float* rope_cos, const float* rope_sin, int B, int L, int H, int D,
                             void* stream) {
  if (!rope_cos || !rope_sin) return CLIPK_ERR_BAD_ARG;
  int rc = check_common(qkv, B, L, H, D, true);
  if (rc) return rc;
  const long rows = (long)B * L * 2 * H;
  const dim3 grid((unsigned)((rows + 255) / 256)), blk(256);
  unsigned short* q = (unsigned short*)qkv;
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 16: hipLaunchKernelGGL(rope_qk_kernel<16>, grid, blk, 0, st, q, rope_cos, rope_sin, rows, L, H); break;
    case 24: hipLaunchKernelGGL(rope_qk_kernel<24>, grid, blk, 0, st, q, rope_cos, rope_sin, rows, L, H); break;
    case 32: hipLaunchKernelGGL(rope_qk_kernel<32>, grid, blk, 0, st, q, rope_cos, rope_sin, rows, L, H); break;
    case 64: hipLaunchKernelGGL(rope_qk_kernel<64>, grid, blk, 0, st, q, rope_cos, rope_sin, rows, L, H); break;
    default: hipLaunchKernelGGL(rope_qk_kernel<128>, grid, blk, 0, st, q, rope_cos, rope_sin, rows, L, H); break;
  }
  return clipk_check_launch();
}

extern "C" int clipk_attn_fwd_rot(void* qkv, const uint8_t* key_mask, const float* rope_cos, const float* rope_sin,
                                  void* out, float* lse, int B, int L, int H, int D, float q_scale, void* stream) {
  if (!rope_cos || !rope_sin) return CLIPK_ERR_BAD_ARG;
  int rc = check_common(qkv, B, L, H, D, true);
  if (rc) return rc;
  if (!out || !lse || !aligned16(out)) return CLIPK_ERR_BAD_ARG;
  if (!whole_fwd_applies(L, D)) {                           // every other shape: the two calls it stands for
    rc = clipk_rope_qk(qkv, rope_cos, rope_sin, B, L, H, D, stream);
    if (rc) return rc;
    return clipk_attn_fwd(qkv, key_mask, nullptr, nullptr, out, lse, B, L, H, D, q_scale, 0.f, 0u, stream);
  }
  AP p{};
  p.qkv = (const unsigned short*)qkv; p.key_mask = key_mask; p.cosT = rope_cos; p.sinT = rope_sin;
  p.out = (unsigned short*)out; p.lse = lse; p.B = B; p.L = L; p.H = H; p.D = D; p.scale = q_scale;
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 16: launch_fwd_whole<16, true>(p, st); break;
    case 24: launch_fwd_whole<24, true>(p, st); break;
    default: launch_fwd_whole<32, true>(p, st); break;
  }
  return clipk_check_launch();
}

extern "C" int clipk_attn_bwd(const void* qkv, const uint8_t* key_mask, const float* rope_cos, const float* rope_sin,
                              const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                              int B, int L, int H, int D, float q_scale, int prerotated, float dropout_p,
                              uint32_t dropout_seed, void* stream) {
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return CLIPK_ERR_BAD_ARG;
  const bool rope = rope_cos != nullptr;
  int rc = check_common(qkv, B, L, H, D, rope);
  if (rc) return rc;
  if (!out || !dout || !lse || !delta || !dqkv) return CLIPK_ERR_BAD_ARG;
  if (!aligned16(out) || !aligned16(dout) || !aligned16(dqkv)) return CLIPK_ERR_BAD_ARG;
  AP p{};
  p.qkv = (const unsigned short*)qkv; p.key_mask = key_mask; p.cosT = rope_cos; p.sinT = rope_sin;
  p.out = (unsigned short*)out; p.lse = const_cast<float*>(lse);
  p.dout = (const unsigned short*)dout; p.delta = delta; p.dqkv = (unsigned short*)dqkv;
  p.B = B; p.L = L; p.H = H; p.D = D; p.scale = q_scale;
  p.pre_rot = (rope && prerotated) ? (prerotated == 2 ? 2 : 1) : 0;
  rc = set_dropout(p, dropout_p, dropout_seed);
  if (rc) return rc;
  ATTN_DISPATCH(launch_bwd, D, rope, p, (hipStream_t)stream);
}

// ---- packed variable-length batches: no padded token ever exists, so the Linear / LayerNorm kernels see only real rows
// ([T, ...] with T = sum of lengths) and attention runs per sequence on rows [cu[b], cu[b+1]).  Replaces the padded
// batches + key-padding masks of current/rna_clip_codes.ipynb:1824-1857,1936-1946 (sequence lengths 30..2542).
// Same kernels as the padded entry points (general flash kernels; the whole-head short-sequence variants need one
// length).  RoPE tables, if given, are indexed by the position INSIDE the sequence and must cover max_len rows.
extern "C" int clipk_attn_varlen_fwd(const void* qkv, const int* cu_seqlens, const float* rope_cos, const float* rope_sin,
                                     void* out, float* lse, int B, int T, int max_len, int H, int D, float q_scale,
                                     float dropout_p, uint32_t dropout_seed, void* stream) {
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return CLIPK_ERR_BAD_ARG;
  const bool rope = rope_cos != nullptr;
  int rc = check_common(qkv, B, max_len, H, D, rope);
  if (rc) return rc;
  if (!cu_seqlens || T <= 0 || max_len > T || !out || !lse || !aligned16(out)) return CLIPK_ERR_BAD_ARG;
  AP p{};
  p.qkv = (const unsigned short*)qkv; p.key_mask = nullptr; p.cosT = rope_cos; p.sinT = rope_sin;
  p.out = (unsigned short*)out; p.lse = lse; p.B = B; p.L = max_len; p.H = H; p.D = D; p.scale = q_scale;
  p.cu = cu_seqlens; p.T = T;
  rc = set_dropout(p, dropout_p, dropout_seed);
  if (rc) return rc;
  ATTN_DISPATCH(launch_fwd, D, rope, p, (hipStream_t)stream);
}

// clipk_attn_fwd_rot for a packed batch: q / k of every sequence rotated IN PLACE (positions count from the sequence's
// first row) by the whole-head kernel while it stages them; only where that kernel applies (D in {16, 24, 32},
// 128 < max_len <= 256) - CLIPK_ERR_UNSUPPORTED otherwise (callers then use clipk_attn_varlen_fwd, which rotates at
// every staging and leaves qkv alone).  Backward: clipk_attn_varlen_bwd(..., prerotated = 1).
extern "C" int clipk_attn_varlen_fwd_rot(void* qkv, const int* cu_seqlens, const float* rope_cos, const float* rope_sin,
                                         void* out, float* lse, int B, int T, int max_len, int H, int D, float q_scale,
                                         void* stream) {
  if (!rope_cos || !rope_sin) return CLIPK_ERR_BAD_ARG;
  int rc = check_common(qkv, B, max_len, H, D, true);
  if (rc) return rc;
  if (!cu_seqlens || T <= 0 || max_len > T || !out || !lse || !aligned16(out)) return CLIPK_ERR_BAD_ARG;
  if (!whole_fwd_applies(max_len, D)) return CLIPK_ERR_UNSUPPORTED;
  AP p{};
  p.qkv = (const unsigned short*)qkv; p.key_mask = nullptr; p.cosT = rope_cos; p.sinT = rope_sin;
  p.out = (unsigned short*)out; p.lse = lse; p.B = B; p.L = max_len; p.H = H; p.D = D; p.scale = q_scale;
  p.cu = cu_seqlens; p.T = T;
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 16: launch_fwd_whole<16, true>(p, st); break;
    case 24: launch_fwd_whole<24, true>(p, st); break;
    default: launch_fwd_whole<32, true>(p, st); break;
  }
  return clipk_check_launch();
}

extern "C" int clipk_attn_varlen_bwd(const void* qkv, const int* cu_seqlens, const float* rope_cos, const float* rope_sin,
                                     const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                                     int B, int T, int max_len, int H, int D, float q_scale, int prerotated,
                                     float dropout_p, uint32_t dropout_seed, void* stream) {
  if ((rope_cos == nullptr) != (rope_sin == nullptr)) return CLIPK_ERR_BAD_ARG;
  const bool rope = rope_cos != nullptr;
  int rc = check_common(qkv, B, max_len, H, D, rope);
  if (rc) return rc;
  if (!cu_seqlens || T <= 0 || max_len > T || !out || !dout || !lse || !delta || !dqkv) return CLIPK_ERR_BAD_ARG;
  if (!aligned16(out) || !aligned16(dout) || !aligned16(dqkv)) return CLIPK_ERR_BAD_ARG;
  AP p{};
  p.qkv = (const unsigned short*)qkv; p.key_mask = nullptr; p.cosT = rope_cos; p.sinT = rope_sin;
  p.out = (unsigned short*)out; p.lse = const_cast<float*>(lse);
  p.dout = (const unsigned short*)dout; p.delta = delta; p.dqkv = (unsigned short*)dqkv;
  p.B = B; p.L = max_len; p.H = H; p.D = D; p.scale = q_scale;
  p.cu = cu_seqlens; p.T = T;
  p.pre_rot = (rope && prerotated) ? (prerotated == 2 ? 2 : 1) : 0;
  rc = set_dropout(p, dropout_p, dropout_seed);
  if (rc) return rc;
  ATTN_DISPATCH(launch_bwd, D, rope, p, (hipStream_t)stream);
}
