// VMEM issue / L1 throughput on MI355X for the operand loads of conv_direct_kernel: how many cycles a CU spends per
// buffer_load_dword / dwordx2 / dwordx4 wave instruction when 16 waves stream register operands from L2, with the
// kernel's address pattern (half wave 0 -> row r, half wave 1 -> row r + 1, consecutive lanes = consecutive elements).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/vmem_issue.hip -o /tmp/vmem && /tmp/vmem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// W = dwords per lane per load.  Each wave walks its own rows (like a K slice): row pair p of wave w at
// base + ((w * NP + p) * 2 + half) * rowlen.  NL loads in flight per batch, REP batches.
template <int W>
__global__ __launch_bounds__(512) void stream_kernel(const float* base, int rowlen, int np, int rep, float* sink, long long* out) {
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* blk = base + (size_t)blockIdx.x * 8 * np * 2 * rowlen;
  u32x4 rsrc;
  const unsigned long long a = (unsigned long long)blk;
  rsrc.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  rsrc.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  rsrc.z = __builtin_amdgcn_readfirstlane(8u * np * 2 * rowlen * 4u);
  rsrc.w = 0x00020000u;
  const int voff = (half * rowlen + l31 * W) * 4;
  float acc = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rep; r++) {
    for (int p = 0; p < np; p += 8) {
      float v[8][W];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int soff = ((wave * np + p + i) * 2 * rowlen) * 4;
        if constexpr (W == 1) asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(v[i][0]) : "v"(voff), "s"(rsrc), "s"(soff));
        if constexpr (W == 2) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(*(f32x2*)v[i]) : "v"(voff), "s"(rsrc), "s"(soff));
        if constexpr (W == 4) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(*(f32x4*)v[i]) : "v"(voff), "s"(rsrc), "s"(soff));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int k = 0; k < W; k++) { asm volatile("" : "+v"(v[i][k])); acc += v[i][k]; }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (acc == 12345.678f) sink[tid] = acc;
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int W>
static void run(const float* buf, float* sink, long long* out, int blocks, int np, int rowlen) {
  const int rep = 20;
  hipLaunchKernelGGL((stream_kernel<W>), dim3(blocks), dim3(512), 0, 0, buf, rowlen, np, 2, sink, out);  // warm L2
  hipLaunchKernelGGL((stream_kernel<W>), dim3(blocks), dim3(512), 0, 0, buf, rowlen, np, rep, sink, out);
  CHECK(hipDeviceSynchronize());
  long long h[1024];
  CHECK(hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (int i = 0; i < blocks; i++) avg += h[i];
  avg /= blocks;
  const double instr_per_wave = (double)rep * np;
  const double waves_per_cu = 8.0 * blocks / 256.0;
  const double cyc_per_instr_cu = avg / (instr_per_wave * waves_per_cu);
  printf("  dwordx%d  %d blocks (%.0f waves/CU), %3d row pairs/wave: %7.0f cycles per wave-instr, %5.1f cycles per instr per CU, %5.1f B/clk/CU\n",
         W, blocks, waves_per_cu, np, avg / instr_per_wave, cyc_per_instr_cu, 256.0 * W / cyc_per_instr_cu);
}

int main() {
  const int rowlen = 256;  // floats per row; a lane reads l31 * W .. + W of its half wave's row
  float* buf; float* sink; long long* out;
  const size_t n = (size_t)512 * 8 * 64 * 2 * rowlen;
  CHECK(hipMalloc(&buf, n * 4));
  CHECK(hipMemset(buf, 0, n * 4));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMalloc(&out, 8192));
  for (int blocks : {256, 512}) {
    for (int np : {8, 24}) {  // 8 KB .. 48 KB per wave: the block's working set comes from L2 every pass
      run<1>(buf, sink, out, blocks, np, rowlen);
      run<2>(buf, sink, out, blocks, np, rowlen);
      run<4>(buf, sink, out, blocks, np, rowlen);
    }
  }
  // Is the ~14-20 B/clk/CU of the L2-streaming case a per-CU limit or an aggregate one?  The same stream from fewer blocks
  // (one 8-wave block per CU while blocks <= 256): bytes per clock per BLOCK and the aggregate.
  for (int blocks : {16, 32, 64, 128, 192, 256}) {
    const int np = 24, rep = 20;
    hipLaunchKernelGGL((stream_kernel<4>), dim3(blocks), dim3(512), 0, 0, buf, rowlen, np, 2, sink, out);
    hipLaunchKernelGGL((stream_kernel<4>), dim3(blocks), dim3(512), 0, 0, buf, rowlen, np, rep, sink, out);
    CHECK(hipDeviceSynchronize());
    long long h[1024];
    CHECK(hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < blocks; i++) avg += h[i];
    avg /= blocks;
    const double bytes = (double)rep * np * 8 * 1024.0;
    printf("  dwordx4 L2 stream, %3d blocks of 8 waves: %6.1f B/clk per block, aggregate %7.0f B/clk\n", blocks, bytes / avg,
           bytes / avg * blocks);
  }
  return 0;
}
