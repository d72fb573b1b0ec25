// Micro-benchmark (tuning aid): fp32 MFMA issue rate on gfx950 under the operand-feeding patterns of the conv kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));

// mode 0: register operands, NACC independent accumulators; mode 1: operands from LDS (2 ds_read_b32 per MFMA)
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.001f * (i & 31);
  __syncthreads();
  floatx16 acc[NACC];
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
  float av = threadIdx.x * 0.001f, bv = 0.5f;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* ap = lds + (lane >> 5) * 32 + (lane & 31) + wave * 64;
  const float* bp = lds + 4096 + (lane >> 5) * 132 + (lane & 31);
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    } else {
      float x[4], y[4][NACC];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        x[u] = ap[((it * 4 + u) & 31) * 64];
#pragma unroll
        for (int a = 0; a < NACC; a++) y[u][a] = bp[((it * 4 + u) & 15) * 264 + a * 32];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u][a], acc[a], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
  if (s == 12345.678f) out[0] = s;
}

template <int NACC, int MODE>
void run(int blocks_per_cu, int iters) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 4 * NACC * 2.0 * 32 * 32 * 2;
  printf("mode %d nacc %d blocks/CU %d (waves/SIMD %d): %.3f ms  %.1f TFLOP/s\n", MODE, NACC, blocks_per_cu, blocks_per_cu, ms,
         flops / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int b : {1, 2, 4}) { run<1, 0>(b, 20000); run<2, 0>(b, 10000); run<4, 0>(b, 5000); }
  for (int b : {1, 2, 4}) { run<1, 1>(b, 20000); run<2, 1>(b, 10000); run<4, 1>(b, 5000); }
  return 0;
}
