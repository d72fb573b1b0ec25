// Per-dispatch cost of back-to-back dependent kernels on one stream (MI355X): how much of a ~12 us conv launch is the
// dispatch itself.   hipcc --offload-arch=gfx950 -O3 tools/ubench/launch_overhead.hip -o /tmp/lo && /tmp/lo
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 1000) p[0] = 1.f; }
// touch: every thread reads and writes one float (the next kernel depends on it through memory)
__global__ void touch_kernel(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
// spin: every wave busy for `cycles` shader clocks
__global__ void spin_kernel(float* p, int cycles) {
  long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (p && threadIdx.x == 1000) p[0] = 1.f;
}

template <class F>
static double timed(F f, int n, hipStream_t st) {
  for (int i = 0; i < 50; i++) f();
  CHECK(hipStreamSynchronize(st));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipEventRecord(a, st));
  for (int i = 0; i < n; i++) f();
  CHECK(hipEventRecord(b, st));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / n;
}

int main() {
  float* buf;
  CHECK(hipMalloc(&buf, 1 << 24));
  CHECK(hipMemset(buf, 0, 1 << 24));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  const int N = 2000;
  printf("per-kernel time of %d back-to-back launches on one stream (us)\n", N);
  printf("  empty <<<1, 64>>>            %6.2f\n", timed([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, buf); }, N, st));
  printf("  empty <<<512, 512>>>         %6.2f\n", timed([&] { hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(512), 0, st, buf); }, N, st));
  printf("  empty <<<512, 512>>> 40K LDS %6.2f\n", timed([&] { hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(512), 40960, st, buf); }, N, st));
  printf("  touch 256K floats            %6.2f\n", timed([&] { hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, st, buf, 262144); }, N, st));
  for (int us : {2, 5, 10})
    printf("  spin %2d us <<<512, 512>>>    %6.2f\n", us, timed([&] { hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(512), 0, st, buf, us * 2400); }, N, st));
  // the same chain captured in a graph
  for (int variant = 0; variant < 2; variant++) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; i++) {
      if (variant == 0) hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(512), 0, st, buf);
      else hipLaunchKernelGGL(spin_kernel, dim3(512), dim3(512), 0, st, buf, 5 * 2400);
    }
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double t = timed([&] { CHECK(hipGraphLaunch(ge, st)); }, 20, st) / 200;
    printf("  graph of 200 x %s %6.2f\n", variant == 0 ? "empty <<<512, 512>>>  " : "spin 5 us <<<512,512>>>", t);
  }
  return 0;
}
