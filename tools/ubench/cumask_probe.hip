// Where do the workgroups of a launch on a CU-MASKED stream (hipExtStreamCreateWithCUMask) run?
// Question behind it (review item 8): can every lane of a multi-lane process be given its own pair of XCDs
// -- its convs off the other lanes' recurrences -- and does the GRU cluster kernel's placement rule (blocks congruent mod 8
// share an XCD, proved by a rendezvous on HW_REG_XCC_ID) survive under such a mask?
//   * which mask bits select which XCC (first 64 bits vs every eighth bit ...),
//   * block id -> XCC under each mask (is it still "round robin over the enabled XCCs"?),
//   * distinct (XCC, SE, SH, CU) slots a co-resident grid occupies,
//   * four disjoint masked streams at once: do their kernels run side by side?
// hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip && ./cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CHK(x)                                                                                  \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); }         \
  } while (0)

__global__ void where_kernel(unsigned* out, unsigned long long hold_ticks) {
  if (threadIdx.x != 0) return;
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
  out[blockIdx.x * 2] = x & 0xF;
  out[blockIdx.x * 2 + 1] = h;
}

__global__ void spin_kernel(unsigned long long* stamps, unsigned long long hold_ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    atomicMin(stamps, t0);
    atomicMax(stamps + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
  }
}

static void probe(const char* name, const std::vector<uint32_t>& mask, int nblk, unsigned* d) {
  hipStream_t st;
  hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%-34s hipExtStreamCreateWithCUMask -> %s\n", name, hipGetErrorString(e)); return; }
  CHK(hipMemsetAsync(d, 0xFF, nblk * 2 * sizeof(unsigned), st));
  hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(256), 0, st, d, 20000ull);  // hold 0.2 ms: the whole grid is co-resident
  CHK(hipStreamSynchronize(st));
  std::vector<unsigned> h(nblk * 2);
  CHK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
  int per_xcc[16] = {0};
  std::set<unsigned> slots;
  int rr = 0;       // blocks whose XCC equals that of block (b % k), k = number of XCCs seen -- i.e. a k-periodic deal
  std::vector<int> first;
  for (int b = 0; b < nblk; b++) {
    per_xcc[h[2 * b] & 15]++;
    slots.insert((h[2 * b] << 16) | ((h[2 * b + 1] >> 8) & 0xFFu));  // XCC | SE_ID SH_ID CU_ID
  }
  int nx = 0;
  for (int i = 0; i < 16; i++) nx += per_xcc[i] != 0;
  for (int b = 0; b < nblk; b++) rr += h[2 * b] == h[2 * (b % nx)];
  int same8 = 0;    // blocks whose XCC equals that of block b % 8: the cluster kernel's assumption
  for (int b = 0; b < nblk; b++) same8 += h[2 * b] == h[2 * (b % 8)];
  printf("%-34s blocks %4d  XCCs seen %d  per XCC:", name, nblk, nx);
  for (int i = 0; i < 8; i++) printf(" %3d", per_xcc[i]);
  printf("  distinct CU slots %3zu  period-%d deal %4d/%d  same XCC as block %% 8: %4d/%d\n    first 24 blocks ->", slots.size(), nx,
         rr, nblk, same8, nblk);
  for (int b = 0; b < 24 && b < nblk; b++) printf(" %u", h[2 * b]);
  printf("\n");
  CHK(hipStreamDestroy(st));
}

int main() {
  hipDeviceProp_t pr;
  CHK(hipGetDeviceProperties(&pr, 0));
  printf("%s, %d CUs\n", pr.name, pr.multiProcessorCount);
  const int NW = 8;  // 256 mask bits
  unsigned* d;
  CHK(hipMalloc(&d, 1024 * 2 * sizeof(unsigned)));
  auto mk = [&](auto pred) {
    std::vector<uint32_t> m(NW, 0u);
    for (int i = 0; i < 32 * NW; i++)
      if (pred(i)) m[i / 32] |= 1u << (i % 32);
    return m;
  };
  probe("all 256 bits", mk([](int) { return true; }), 256, d);
  probe("bits 0..63", mk([](int i) { return i < 64; }), 64, d);
  probe("bits 0..63 (256 blocks)", mk([](int i) { return i < 64; }), 256, d);
  probe("bits 0..31", mk([](int i) { return i < 32; }), 64, d);
  probe("bits i%8 < 2", mk([](int i) { return i % 8 < 2; }), 64, d);
  probe("bits i%8 < 2 (256 blocks)", mk([](int i) { return i % 8 < 2; }), 256, d);
  probe("bits i%8 == 0", mk([](int i) { return i % 8 == 0; }), 64, d);
  probe("bits i%8 in {2,3}", mk([](int i) { return i % 8 == 2 || i % 8 == 3; }), 64, d);
  probe("bits i%8 in {6,7}", mk([](int i) { return i % 8 >= 6; }), 64, d);
  probe("bits i%8 < 4", mk([](int i) { return i % 8 < 4; }), 128, d);
  probe("bits 64..127", mk([](int i) { return i >= 64 && i < 128; }), 64, d);

  // four disjoint partitions (by the interleaved rule AND by the contiguous rule), one 64-block 2-ms kernel each: side by side?
  for (int rule = 0; rule < 2; rule++) {
    hipStream_t st[4];
    bool ok = true;
    for (int l = 0; l < 4; l++) {
      auto m = rule == 0 ? mk([l](int i) { return i % 8 == 2 * l || i % 8 == 2 * l + 1; }) : mk([l](int i) { return i / 64 == l; });
      if (hipExtStreamCreateWithCUMask(&st[l], NW, m.data()) != hipSuccess) ok = false;
    }
    if (!ok) { printf("partition rule %d: stream creation failed\n", rule); continue; }
    unsigned long long* stamps;
    CHK(hipMalloc(&stamps, 4 * 2 * sizeof(unsigned long long)));
    std::vector<unsigned long long> init(8);
    for (int l = 0; l < 4; l++) { init[2 * l] = ~0ull; init[2 * l + 1] = 0ull; }
    for (int rep = 0; rep < 2; rep++) {
      CHK(hipMemcpy(stamps, init.data(), 64, hipMemcpyHostToDevice));
      for (int l = 0; l < 4; l++) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, st[l], stamps + 2 * l, 200000ull);
      CHK(hipDeviceSynchronize());
      std::vector<unsigned long long> s(8);
      CHK(hipMemcpy(s.data(), stamps, 64, hipMemcpyDeviceToHost));
      unsigned long long lo = ~0ull, hi = 0;
      for (int l = 0; l < 4; l++) { lo = s[2 * l] < lo ? s[2 * l] : lo; hi = s[2 * l + 1] > hi ? s[2 * l + 1] : hi; }
      printf("partition rule %s, rep %d: four 2.0-ms kernels on four masked streams took %.2f ms start to end (starts:", rule == 0 ? "i%8 pairs" : "i/64", rep,
             (hi - lo) / 1e5);
      for (int l = 0; l < 4; l++) printf(" +%.2f", (s[2 * l] - lo) / 1e5);
      printf(" ms)\n");
    }
    for (int l = 0; l < 4; l++) CHK(hipStreamDestroy(st[l]));
    CHK(hipFree(stamps));
  }
  return 0;
}
