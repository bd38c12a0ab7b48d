// probe_gather.hip — how fast can the chip gather the attention operands' head slices?
// A (batch, head) slice of qkv [B*L, 3*H*D] bf16 is L pieces of D*2 bytes at a stride of 3*H*D*2 bytes (hd 24:
// 48-byte pieces every 2880 bytes).  Persistent workgroups walk the heads exactly like attn_bwd_fused32_kernel
// (four lanes per row, 16 bytes per lane, 20 loads per lane and head, XCD-grouped head order) and only XOR the
// data together: no LDS, no barriers, nothing but the loads.  Compared with the same bytes read contiguously.
//   hipcc --offload-arch=gfx950 -O3 -o probe_gather probe_gather.hip && ./probe_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int depth>
__global__ __launch_bounds__(256, 2) void gather_kernel(const unsigned short* qkv, const unsigned short* dout,
                                                        const unsigned short* out, unsigned int* sink, int B, int L, int H,
                                                        int D) {
  const int tid = threadIdx.x, ci = tid & 3, r0 = tid >> 2, cpr = D >> 3;
  const int nheads = B * H;
  const unsigned int HD = H * D, cc = 8u * (ci < cpr ? ci : cpr - 1);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int w0 = blockIdx.x; w0 < nheads; w0 += gridDim.x * depth) {
    u32x4 v[depth][20];
#pragma unroll
    for (int dd = 0; dd < depth; ++dd) {
      int w = w0 + dd * gridDim.x; w = w < nheads ? w : nheads - 1;
      const int xcd = w & 7, slot = w >> 3;
      const int wi = ((slot / H) * 8 + xcd) * H + (slot % H);       // all heads of a batch element on one XCD
      const int b = wi / H, h = wi - b * H;
      const unsigned short* qb = qkv + (long)b * L * 3 * HD + (long)h * D;
      const unsigned short* db = dout + (long)b * L * HD + (long)h * D;
      const unsigned short* ob = out + (long)b * L * HD + (long)h * D;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const unsigned int row = ps * 64 + r0;
        const unsigned int qo = row * 3u * HD + cc, oo = row * HD + cc;
        v[dd][5 * ps + 0] = *reinterpret_cast<const u32x4*>(qb + qo);
        v[dd][5 * ps + 1] = *reinterpret_cast<const u32x4*>(qb + qo + HD);
        v[dd][5 * ps + 2] = *reinterpret_cast<const u32x4*>(qb + qo + 2u * HD);
        v[dd][5 * ps + 3] = *reinterpret_cast<const u32x4*>(db + oo);
        v[dd][5 * ps + 4] = *reinterpret_cast<const u32x4*>(ob + oo);
      }
    }
#pragma unroll
    for (int dd = 0; dd < depth; ++dd)
#pragma unroll
      for (int i = 0; i < 20; ++i) acc ^= v[dd][i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

__global__ __launch_bounds__(256) void stream_kernel(const u32x4* x, long n, unsigned int* sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
    u32x4 a = x[i], b = i + stride < n ? x[i + stride] : a, c = i + 2 * stride < n ? x[i + 2 * stride] : a,
          d = i + 3 * stride < n ? x[i + 3 * stride] : a;
    acc ^= a ^ b ^ c ^ d;
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main() {
  const int B = 512, L = 256;
  struct Shape { int H, D; } shapes[] = {{20, 24}, {20, 32}};     // four 16-byte lanes per row: head dims <= 32
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned int* sink; (void)hipMalloc(&sink, 4);
  for (auto sh : shapes) {
    const long T = (long)B * L, nq = T * 3 * sh.H * sh.D, no = T * sh.H * sh.D;
    unsigned short *qkv, *dout, *out;
    (void)hipMalloc(&qkv, nq * 2); (void)hipMalloc(&dout, no * 2); (void)hipMalloc(&out, no * 2);
    (void)hipMemset(qkv, 1, nq * 2); (void)hipMemset(dout, 2, no * 2); (void)hipMemset(out, 3, no * 2);
    const double bytes = (double)(nq + 2 * no) * 2;
    for (int it = 0; it < 300; ++it) hipLaunchKernelGGL(gather_kernel<1>, dim3(512), dim3(256), 0, 0, qkv, dout, out, sink, B, L, sh.H, sh.D);   // clock ramp
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int wgs : {512, 1024}) {
      for (int depth : {1, 2}) {
        auto launch = [&]() {
          if (depth == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(wgs), dim3(256), 0, 0, qkv, dout, out, sink, B, L, sh.H, sh.D);
          else hipLaunchKernelGGL(gather_kernel<2>, dim3(wgs), dim3(256), 0, 0, qkv, dout, out, sink, B, L, sh.H, sh.D);
        };
        for (int it = 0; it < 3; ++it) launch();
        (void)hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("H=%d D=%d gather  wgs=%4d heads-in-flight/wg=%d : %7.1f us  %5.2f TB/s\n", sh.H, sh.D, wgs, depth, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12);
      }
    }
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, (const u32x4*)qkv, nq / 8, sink);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, (const u32x4*)qkv, nq / 8, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("H=%d D=%d stream of qkv only                  : %7.1f us  %5.2f TB/s\n", sh.H, sh.D, ms * 100, (double)nq * 2 / (ms / 10 * 1e-3) / 1e12);
    (void)hipFree(qkv); (void)hipFree(dout); (void)hipFree(out);
  }
  return 0;
}
