// How fast does a freshly launched kernel pull a deep-level conv's WEIGHT operand through a cold L2?  (L2 is per XCD and is
// invalidated at kernel boundaries; the weights of one layer are used by one launch per score pass.)  The access pattern of
// conv_direct4w_kernel's A loads for a 512 x 512 k5 layer at 401 frames -- 224 blocks (32 row groups of 16 rows x 7 column tiles),
// 8 waves per block = 8 slices of the channels, 16 ring slots per wave, ring depth 4, per slot one dwordx4 + one dwordx2 per lane
// from [Cin][Mp][8] floats (lane (row, kk): row m0 + l15 of channel 4 J + kk) -- against the same bytes laid out contiguously per
// row group ([row group][Cin][16][8]), with the buffer warm in the memory-side cache (same buffer every launch) or cold (a
// rotation of buffers larger than the 256 MB infinity cache).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/cold_stream.hip -o /tmp/cold && /tmp/cold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CIN = 512, MP = 512, KWP = 8, NRG = 32, NCT = 7;

template <int LAYOUT, int D, int X2>
__global__ __launch_bounds__(512) void fill_kernel(const float* w, float* sink, long long* out) {
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kk = lane >> 4;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = blockIdx.x, q8 = L >> 3;
  const int rg = (q8 % (NRG / 8)) * 8 + (L & 7);
  u32x4 rsrc;
  const unsigned long long a = (unsigned long long)w;
  rsrc.x = (unsigned)a; rsrc.y = (unsigned)(a >> 32) & 0xFFFFu; rsrc.z = CIN * MP * KWP * 4u; rsrc.w = 0x00020000u;
  // LAYOUT 0: [Cin][Mp][KWP] (the packer's);  1: [row group][Cin][16][KWP]
  const int avo = LAYOUT == 0 ? (kk * MP + rg * 16 + l15) * KWP * 4 : ((rg * CIN + kk) * 16 + l15) * KWP * 4;
  f32x4 a4[D]; f32x2 a2[D];
  float acc = 0.f;
  const long long t0 = __builtin_readcyclecounter();
#define ISSUE(g, d) { const int c4 = (wk + 8 * (g)) * 4; const int aso = LAYOUT == 0 ? c4 * MP * KWP * 4 : c4 * 16 * KWP * 4; \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a4[d]) : "v"(avo), "s"(rsrc), "s"(aso)); \
    if (X2) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:16" : "=v"(a2[d]) : "v"(avo), "s"(rsrc), "s"(aso)); }
#define USE(d, n) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((n) * (1 + X2))); asm volatile("" : "+v"(a4[d])); if (X2) asm volatile("" : "+v"(a2[d])); \
    acc += a4[d].x + a4[d].w + (X2 ? a2[d].y : 0.f); }
  constexpr int NS = CIN / 4 / 8;
#pragma unroll
  for (int d = 0; d < D; d++) ISSUE(d, d);
#pragma unroll
  for (int g = 0; g < NS; g++) {
    USE(g % D, (NS - 1 - g) < (D - 1) ? (NS - 1 - g) : (D - 1));
    if (g + D < NS) ISSUE(g + D, g % D);
  }
  const long long t1 = __builtin_readcyclecounter();
  if (acc == 12345.678f) sink[tid] = acc;
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int LAYOUT, int D, int X2>
static void run(float** bufs, int nbuf, float* sink, long long* out, const char* what) {
  const int blocks = NRG * NCT, reps = 40;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 4; i++) hipLaunchKernelGGL((fill_kernel<LAYOUT, D, X2>), dim3(blocks), dim3(512), 0, 0, bufs[i % nbuf], sink, out);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((fill_kernel<LAYOUT, D, X2>), dim3(blocks), dim3(512), 0, 0, bufs[i % nbuf], sink, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  long long h[1024];
  CHECK(hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0, mx = 0;
  for (int i = 0; i < blocks; i++) { avg += h[i]; if (h[i] > mx) mx = h[i]; }
  avg /= blocks;
  const double bytes = (double)CIN * MP * (X2 ? 24 : 16);  // unique bytes the launch uses
  printf("  %-34s %-6s D=%d %s: %6.2f us per launch, in-kernel %6.0f cycles per block (max %6.0f) = %5.2f us at 2.4 GHz | unique %4.1f MB -> %5.2f TB/s\n",
         LAYOUT == 0 ? "[Cin][Mp][8] (strided 512 B pieces)" : "[row group][Cin][16][8] (contiguous)", what, D, X2 ? "x4+x2" : "x4   ",
         ms * 1e3 / reps, avg, mx, avg / 2400.0, bytes / 1e6, bytes / (avg / 2.4e9) / 1e12);
}

int main() {
  const size_t n = (size_t)CIN * MP * KWP;
  const int NB = 40;  // 40 x 8.4 MB = 336 MB > the 256 MB memory-side cache
  float* bufs[NB];
  for (int i = 0; i < NB; i++) { CHECK(hipMalloc(&bufs[i], n * 4)); CHECK(hipMemset(bufs[i], 0, n * 4)); }
  float* sink; long long* out;
  CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&out, 8192));
  for (int pass = 0; pass < 2; pass++) {
    const int nbuf = pass == 0 ? 1 : NB;
    const char* what = pass == 0 ? "warm" : "cold";
    run<0, 4, 1>(bufs, nbuf, sink, out, what);
    run<1, 4, 1>(bufs, nbuf, sink, out, what);
    run<0, 8, 1>(bufs, nbuf, sink, out, what);
    run<1, 8, 1>(bufs, nbuf, sink, out, what);
    run<0, 4, 0>(bufs, nbuf, sink, out, what);
    run<1, 4, 0>(bufs, nbuf, sink, out, what);
  }
  return 0;
}
