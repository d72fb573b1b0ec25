// How do fp32 MFMAs (v_mfma_f32_16x16x4_f32, 8 passes = 32 cycles) and plain VALU work share a SIMD on gfx950?  The loops of the
// minimal-filtering conv kernels issue, per ring slot and wave, 2 (KW + 1) MFMAs and 40-85 VALU instructions (PReLU, edge
// selects, input transform); with two waves per SIMD their loops run at ~1.6x the matrix-pipe time although neither the VALU
// issue port (4 cycles per wave64 instruction) nor memory is saturated on paper.  This measures cycles per "slot" of 12
// independent MFMAs + NV independent VALU FMAs, for one / two waves per SIMD and two instruction orders:
//   block:       NV VALU, then 12 MFMAs          interleaved: after every MFMA, NV / 12 VALU
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_mix.hip -o /tmp/mix && /tmp/mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>
__global__ __launch_bounds__(512) void mix_kernel(float* sink, long long* out, int iters) {
  f32x4 acc[12];
#pragma unroll
  for (int i = 0; i < 12; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 0.001f + i;
  float a = threadIdx.x * 0.5f, b = 1.0001f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < NV; k++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k & 7]) : "v"(a), "v"(b));
#pragma unroll
      for (int m = 0; m < 12; m++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
    } else {
#pragma unroll
      for (int m = 0; m < 12; m++) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < NV / 12; k++) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[(m + k) & 7]) : "v"(a), "v"(b));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 12; i++) s += acc[i].x + acc[i].w;
#pragma unroll
  for (int i = 0; i < 8; i++) s += v[i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NV, int MODE>
static void run(float* sink, long long* out, int threads) {
  const int iters = 2000, blocks = 256;
  hipLaunchKernelGGL((mix_kernel<NV, MODE>), dim3(blocks), dim3(threads), 0, 0, sink, out, 10);
  hipLaunchKernelGGL((mix_kernel<NV, MODE>), dim3(blocks), dim3(threads), 0, 0, sink, out, iters);
  CHECK(hipDeviceSynchronize());
  long long h[256];
  CHECK(hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (int i = 0; i < blocks; i++) avg += h[i];
  avg /= blocks;
  const int wps = threads / 256;
  printf("  %2d VALU + 12 MFMA per slot, %-11s %d wave(s) per SIMD: %7.1f cycles per slot per wave -> %6.1f cycles per slot of all waves of a SIMD "
         "(matrix pipe alone: %d, VALU port alone: %d)\n", NV, MODE == 0 ? "block," : "interleaved,", wps, avg / iters, avg / iters,
         12 * 32 * wps, (NV + 12) * 4 * wps);
}

int main() {
  float* sink; long long* out;
  CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&out, 8192));
  for (int threads : {256, 512}) {
    run<0, 0>(sink, out, threads);
    run<24, 0>(sink, out, threads); run<24, 1>(sink, out, threads);
    run<48, 0>(sink, out, threads); run<48, 1>(sink, out, threads);
    run<84, 0>(sink, out, threads); run<84, 1>(sink, out, threads);
    run<120, 0>(sink, out, threads); run<120, 1>(sink, out, threads);
  }
  return 0;
}
