// probe_layouts.hip — hardware probes for the gfx950 lane maps the kernels rely on.
// Run on the GPU box (`make probes && tools/probes/probe_layouts`); prints PASS/FAIL per map.
// Maps under test are the ones documented in /opt/skills/guides/cdna_hip_programming.md §3 and T10.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __host__ inline unsigned short f2bf(float f) { unsigned int u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

// ---- 1. mfma_f32_16x16x32_bf16 : A[16][32], B[32][16] -> C[16][16]
__global__ void k_mfma16(const float* A, const float* B, float* Cout) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)f2bf(A[(l & 15) * 32 + 8 * (l >> 4) + j]);
    b[j] = (short)f2bf(B[(8 * (l >> 4) + j) * 16 + (l & 15)]);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) Cout[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// ---- 2. mfma_f32_32x32x2f32 : A[32][2], B[2][32] -> C[32][32]
__global__ void k_mfma32f(const float* A, const float* B, float* Cout) {
  const int l = threadIdx.x;
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) Cout[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// ---- 3. ds_read_b64_tr_b16: T[32][STRIDE] 16-bit, value = row*64+col
constexpr int TSTRIDE = 48;  // elements (96 B rows)
__global__ void k_tr(short* out /*[64][4]*/) {
  __shared__ __attribute__((aligned(16))) short T[32 * TSTRIDE];
  for (int i = threadIdx.x; i < 32 * TSTRIDE; i += 64) T[i] = (short)((i / TSTRIDE) * 64 + (i % TSTRIDE));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(T + (4 * g + q) * TSTRIDE + 16 + 4 * p));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
// ---- 4. attention PV chain: O^T[d][q] = sum_key V[key][d] P[key][q]; keys=32, d=16, q=16.
// P^T given in the accumulator layout of two 16x16 S^T tiles; V via transposed LDS reads.
__global__ void k_pv(const float* V /*[32][16]*/, const float* P /*[32][16]*/, float* OT /*[16][16]*/) {
  __shared__ __attribute__((aligned(16))) short Vl[32 * TSTRIDE];
  for (int i = threadIdx.x; i < 32 * 16; i += 64) Vl[(i / 16) * TSTRIDE + (i % 16)] = (short)f2bf(V[i]);
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  f32x4 p0, p1;   // accumulator-layout S^T tiles: rows = keys, col = query (l&15)
  for (int r = 0; r < 4; ++r) { p0[r] = P[(4 * g + r) * 16 + i]; p1[r] = P[(16 + 4 * g + r) * 16 + i]; }
  bf16x8 b;
  for (int r = 0; r < 4; ++r) { b[r] = (short)f2bf(p0[r]); b[4 + r] = (short)f2bf(p1[r]); }
  s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(Vl + (4 * g + q) * TSTRIDE + 4 * p));
  s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(Vl + (16 + 4 * g + q) * TSTRIDE + 4 * p));
  bf16x8 a;
  for (int r = 0; r < 4; ++r) { a[r] = a0[r]; a[4 + r] = a1[r]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) OT[(4 * g + r) * 16 + i] = c[r];
}
// ---- 5. wgrad: dW[n][k] = sum_m dY[m][n] X[m][k]; m=32, n=16, k=16; both operands by tr reads
__global__ void k_wgrad(const float* dY /*[32][16]*/, const float* X /*[32][16]*/, float* dW /*[16][16]*/) {
  __shared__ __attribute__((aligned(16))) short Yl[32 * TSTRIDE];
  __shared__ __attribute__((aligned(16))) short Xl[32 * TSTRIDE];
  for (int i = threadIdx.x; i < 32 * 16; i += 64) {
    Yl[(i / 16) * TSTRIDE + (i % 16)] = (short)f2bf(dY[i]);
    Xl[(i / 16) * TSTRIDE + (i % 16)] = (short)f2bf(X[i]);
  }
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  bf16x8 a, b;
  {
    s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Yl + (4 * g + q) * TSTRIDE + 4 * p));
    s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Yl + (16 + 4 * g + q) * TSTRIDE + 4 * p));
    for (int r = 0; r < 4; ++r) { a[r] = t0[r]; a[4 + r] = t1[r]; }
    t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Xl + (4 * g + q) * TSTRIDE + 4 * p));
    t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Xl + (16 + 4 * g + q) * TSTRIDE + 4 * p));
    for (int r = 0; r < 4; ++r) { b[r] = t0[r]; b[4 + r] = t1[r]; }
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) dW[(4 * g + r) * 16 + i] = c[r];
}

static std::vector<float> rnd(int n, unsigned seed) {
  std::vector<float> v(n); srand(seed);
  for (auto& x : v) x = (float)((rand() % 17) - 8);
  return v;
}
template <class T> static T* dev(const std::vector<T>& h) {
  T* d; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d;
}
static int report(const char* name, const std::vector<float>& got, const std::vector<float>& ref) {
  double e = 0; for (size_t i = 0; i < ref.size(); ++i) e = fmax(e, fabs(got[i] - ref[i]));
  printf("%-44s %s (max err %g)\n", name, e == 0 ? "PASS" : "FAIL", e);
  return e == 0 ? 0 : 1;
}

int main() {
  int fails = 0;
  {  // 1
    auto A = rnd(16 * 32, 1), B = rnd(32 * 16, 2); std::vector<float> C(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += A[i * 32 + k] * B[k * 16 + j];
    float *dA = dev(A), *dB = dev(B), *dC = dev(C);
    hipLaunchKernelGGL(k_mfma16, 1, 64, 0, 0, dA, dB, dC); hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
    fails += report("mfma_f32_16x16x32_bf16 A/B/C lane maps", C, R);
  }
  {  // 2
    auto A = rnd(64, 3), B = rnd(64, 4); std::vector<float> C(1024), R(1024, 0.f);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 2; ++k) R[i * 32 + j] += A[i * 2 + k] * B[k * 32 + j];
    float *dA = dev(A), *dB = dev(B), *dC = dev(C);
    hipLaunchKernelGGL(k_mfma32f, 1, 64, 0, 0, dA, dB, dC); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    fails += report("mfma_f32_32x32x2f32 A/B/C lane maps", C, R);
  }
  {  // 3
    std::vector<short> o(256); short* d = dev(o);
    hipLaunchKernelGGL(k_tr, 1, 64, 0, 0, d); hipMemcpy(o.data(), d, 512, hipMemcpyDeviceToHost);
    std::vector<float> got(256), ref(256);
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { got[l * 4 + e] = o[l * 4 + e]; ref[l * 4 + e] = (float)((4 * (l >> 4) + e) * 64 + 16 + (l & 15)); }
    fails += report("ds_read_b64_tr_b16 block semantics", got, ref);
  }
  {  // 4
    auto V = rnd(32 * 16, 5), P = rnd(32 * 16, 6); std::vector<float> O(256), R(256, 0.f);
    for (int d = 0; d < 16; ++d) for (int q = 0; q < 16; ++q) for (int k = 0; k < 32; ++k) R[d * 16 + q] += V[k * 16 + d] * P[k * 16 + q];
    float *dV = dev(V), *dP = dev(P), *dO = dev(O);
    hipLaunchKernelGGL(k_pv, 1, 64, 0, 0, dV, dP, dO); hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
    fails += report("acc-as-B chain O^T = V^T P^T (tr reads)", O, R);
  }
  {  // 5
    auto Y = rnd(32 * 16, 7), X = rnd(32 * 16, 8); std::vector<float> W(256), R(256, 0.f);
    for (int n = 0; n < 16; ++n) for (int k = 0; k < 16; ++k) for (int m = 0; m < 32; ++m) R[n * 16 + k] += Y[m * 16 + n] * X[m * 16 + k];
    float *dY = dev(Y), *dX = dev(X), *dW = dev(W);
    hipLaunchKernelGGL(k_wgrad, 1, 64, 0, 0, dY, dX, dW); hipMemcpy(W.data(), dW, 1024, hipMemcpyDeviceToHost);
    fails += report("wgrad dW = dY^T X (both tr reads)", W, R);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 2; }
  printf("%s\n", fails ? "PROBES FAILED" : "ALL PROBES PASS");
  return fails ? 1 : 0;
}
