// Store-shape microbenchmark: how much does the shape of a wave's 1-KiB store instruction matter for a large
// streamed bf16 output?  A: 8 rows x 128 B (what the GEMM epilogue's per-wave slab gives), B: 2 rows x 512 B,
// C: 1 row x 1 KiB.  Output [M][N] bf16, every byte written once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int ROWS>   // rows per wave instruction; 1024 / ROWS bytes per row
__global__ __launch_bounds__(256) void k(unsigned short* out, int M, int N) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
  constexpr int BPR = 1024 / ROWS, LPR = BPR / 16;          // bytes, lanes per row
  const int segs = (N * 2) / BPR;                            // segments per row
  const long units = (long)(M / ROWS) * segs;                // one unit = ROWS rows x BPR bytes
  const u32x4 v = {1u, 2u, 3u, 4u};
  for (long u = wave; u < units; u += nwaves) {
    const long rb = u / segs; const int sg = (int)(u - rb * segs);
    const int r = lane / LPR, c = lane % LPR;
    char* p = (char*)out + ((rb * ROWS + r) * (long)N * 2) + (long)sg * BPR + c * 16;
    *(u32x4*)p = v;
  }
}
int main() {
  const int M = 131072, N = 2048;
  unsigned short* d; hipMalloc(&d, (size_t)M * N * 2);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](auto kern, const char* name) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, M, N);
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, M, N);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s: %.1f us  %.2f TB/s\n", name, ms * 100, (double)M * N * 2 / (ms / 10 * 1e-3) / 1e12);
  };
  run(k<8>, "8 rows x 128 B");
  run(k<2>, "2 rows x 512 B");
  run(k<1>, "1 row  x 1 KiB");
  return 0;
}
