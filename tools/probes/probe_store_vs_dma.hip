// Does it help to give the epilogue's stores and the main loop's LDS-DMA loads to DIFFERENT waves?
//
// gemm_nt_v3 model: one 8-wave workgroup per CU walks output tiles; per tile KT "K-steps" (each: 64 KiB of operand
// tiles by LDS-DMA from an L2-resident panel, two steps in flight, a compute phase emulated by real MFMAs), then an
// epilogue that stores 128 KiB (a 256 x 256 bf16 tile) to fresh HBM addresses.  vmcnt retires in order and is shared by
// loads and stores, so in the kernel as built (variant A) a wave's loads of K-step >= 2 of the next tile cannot be
// confirmed before its own stores of the previous tile have drained.
//   A  every wave: 8 DMA pieces per step + 16 stores per tile, next tile's steps 0 / 1 prefetched before the stores,
//      counted waits that tolerate the stores for those two steps (what gemm_nt_v3 does);
//   B  waves 0-3: all 16 DMA pieces per step, no stores;  waves 4-7: all 32 stores per tile, no loads, no vmcnt waits;
//   C  A without the stores (lower bound);   D  B without the stores;
//   E  every wave: 8 DMA pieces + 2 stores per step (the previous tile's output trickling out under the main loop:
//      what a kernel could do if it had room to keep the finished tile in LDS).
// Output: time per tile for each variant.  If B ~ C, wave specialisation hides the store drain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int KT = 8;             // K-steps per tile (K = 480 -> 7.5)
constexpr int STEP_BYTES = 65536; // operand bytes per K-step per workgroup
constexpr int TILE_OUT = 131072;  // bf16 256 x 256

template <int VARIANT>
__global__ __launch_bounds__(512, 1) void k(const char* src, long src_bytes, char* out, int tiles_per_wg, float* sink, int nmfma) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 64 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool SPLIT = (VARIANT == 1 || VARIANT == 3);
  constexpr bool STORES = (VARIANT == 0 || VARIANT == 1);
  constexpr bool TRICKLE = (VARIANT == 4);
  const bool loader = !SPLIT || wid < 4;
  const bool storer = !SPLIT || wid >= 4;
  constexpr int PPS = SPLIT ? 16 : 8;                           // DMA pieces (1 KiB) per loader wave and step
  constexpr int NS = SPLIT ? 32 : 16;                           // stores per storer wave and tile
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
  const int lw = wid;                                           // loader index (0-7, or 0-3 when split)
  auto stage = [&](int gstep) {                                 // K-step number since kernel start (buffer = gstep & 1)
    char* dst = smem + (gstep & 1) * STEP_BYTES + lw * PPS * 1024;
    const unsigned soff = (unsigned)(((long)gstep * STEP_BYTES) % (src_bytes - STEP_BYTES));
#pragma unroll
    for (int i = 0; i < PPS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                               (int)(lw * PPS * 1024 + i * 1024 + lane * 16), (int)soff, 0, 0);
  };
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 3, 4, 5, 6, 7, 8, 9};
  auto compute = [&]() {
    for (int i = 0; i < nmfma; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  };
  int g = 0;                                                    // global K-step counter
  if (loader) { stage(0); stage(1); }
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (int t = 0; t < tiles_per_wg; ++t) {
    const bool first = (t == 0);
    for (int s = 0; s < KT; ++s, ++g) {
      // wait for step g (in flight: step g+1, plus - for steps 0 and 1 of a tile after the first - the NS stores)
      if (loader) {
        const bool tol = STORES && !SPLIT && !first && s < 2;    // stores are younger than these loads
        if (s == KT - 1 && t == tiles_per_wg - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (tol) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // 8 younger loads + 16 stores may stay
        else if (SPLIT) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (TRICKLE) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");   // 8 loads + 2 stores of the next step
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      compute();
      __builtin_amdgcn_s_barrier();                              // buffer g & 1 is free again
      const bool more = !(t == tiles_per_wg - 1 && s >= KT - 2);
      if (loader && more) stage(g + 2);                          // the last two steps of a tile prefetch the next tile
      if (TRICKLE && t > 0) {                                    // 2 of the previous tile's 16 stores per step
        char* o = out + ((long)blockIdx.x * tiles_per_wg + t - 1) * TILE_OUT + wid * 16 * 1024 + s * 2048;
        *(u32x4*)(o + lane * 16) = v;
        *(u32x4*)(o + 1024 + lane * 16) = v;
      }
    }
    // epilogue: the two prefetches of the next tile were issued in the last two steps (they precede the stores)
    if (STORES && storer) {
      char* o = out + ((long)blockIdx.x * tiles_per_wg + t) * TILE_OUT + (SPLIT ? (wid - 4) * NS * 1024 : wid * NS * 1024);
#pragma unroll
      for (int i = 0; i < NS; ++i) *(u32x4*)(o + i * 1024 + lane * 16) = v;
    }
  }
  float r = 0;
  for (int j = 0; j < 4; ++j) r += acc[j][0];
  if (r == 1.2345f) sink[0] = r;
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? atoi(argv[1]) : 12;
  const int nmfma = argc > 2 ? atoi(argv[2]) : 16;   // 4 * nmfma MFMAs (16 cycles each) per wave and K-step; 16 -> ~1024 cycles
  int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
  const int ncu = prop.multiProcessorCount;
  const long src_bytes = 4L << 20;
  char *src, *out; float* sink;
  hipMalloc(&src, src_bytes); hipMemset(src, 1, src_bytes);
  hipMalloc(&out, (size_t)ncu * tiles * TILE_OUT); hipMalloc(&sink, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](auto kern, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STEP_BYTES);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(ncu), dim3(512), 2 * STEP_BYTES, 0, src, src_bytes, out, tiles, sink, nmfma);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(a);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(ncu), dim3(512), 2 * STEP_BYTES, 0, src, src_bytes, out, tiles, sink, nmfma);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("%-42s %8.2f us per tile   (%.1f us per launch)\n", name, best / 5 * 1e3 / tiles, best / 5 * 1e3);
  };
  printf("CUs %d, %d tiles per workgroup, %d K-steps per tile, %d MFMAs per wave and step\n", ncu, tiles, KT, 4 * nmfma);
  for (int round = 0; round < 2; ++round) {
    run(k<0>, "A  all waves load + store (v3 scheme)");
    run(k<1>, "B  waves 0-3 load, waves 4-7 store");
    run(k<2>, "C  A without stores");
    run(k<3>, "D  B without stores");
    run(k<4>, "E  2 stores per wave and step (trickle)");
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
