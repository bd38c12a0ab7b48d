// probe_mfma_valu.hip — do a matrix-instruction stream and a vector-instruction stream of two DIFFERENT waves on the
// same SIMD overlap on gfx950?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) issue independent
// v_mfma_f32_16x16x32_bf16, waves 4-7 (their SIMD partners) issue independent v_pk_fma_f32 (or v_exp_f32).
// Prints the time of each stream alone and of both together: "max" = they overlap, "sum" = they do not.
//   make tools/probes/probe_mfma_valu && tools/probes/probe_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// mode bit 0: waves 0-3 run the MFMA stream; bit 1: waves 4-7 run packed fma; bit 2: waves 4-7 run v_exp_f32
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, int mode) {
  const int wid = threadIdx.x >> 6;
  if (wid < 4) {
    if (!(mode & 1)) return;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f - threadIdx.x * 0.002f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e-30f) out[threadIdx.x] = s;
  } else if (mode & 2) {
    f32x2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f32x2{threadIdx.x * 1e-3f + i, 1.f};
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);       // 64 v_pk_fma_f32 per iteration
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i][0] + v[i][1];
    if (s == 1.2345e-30f) out[threadIdx.x] = s;
  } else if (mode & 4) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = -(threadIdx.x * 1e-3f + i) * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0f;          // 16 v_exp_f32 + 16 v_add per iteration
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    if (s == 1.2345e-30f) out[threadIdx.x] = s;
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int modes[] = {1, 2, 3, 4, 5};
  const char* names[] = {"MFMA stream alone (waves 0-3: 16 x 16x16x32 bf16 per iteration)", "packed-fma stream alone (waves 4-7: 64 v_pk_fma_f32 per iteration)",
                         "MFMA + packed fma together", "v_exp stream alone (waves 4-7: 16 v_exp_f32 + 16 v_add per iteration)", "MFMA + v_exp together"};
  for (int rep = 0; rep < 2; ++rep)
    for (int k = 0; k < 5; ++k) {
      hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, out, iters, modes[k]);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, out, iters, modes[k]);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) printf("%-75s %8.3f ms  (%.1f ns per iteration)\n", names[k], ms, ms * 1e6 / iters);
    }
  return 0;
}
