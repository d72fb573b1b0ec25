// Cross-CU hand-off latency on MI355X: two workgroups play ping-pong through one {value, tag} granule each, with the
// store / load cache policies the GRU exchange could use.  Decides which instructions gru_ring_kernel publishes and polls
// with, and what a time step's exchange can cost at best.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xchg_latency.hip -o /tmp/xchg && /tmp/xchg
// Every wait is bounded (a policy that never observes the other side's store reports "no progress" instead of hanging).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

enum { LD_SC1 = 0, LD_SC0SC1, LD_SC0, LD_NT, LD_SC1NT, LD_ATOMIC, LD_INV_PLAIN, LD_INV0_PLAIN, NLD };
enum { ST_PLAIN = 0, ST_SC1, ST_SC0SC1, ST_NT, ST_SC0, ST_WB, NST };
static const char* kLd[] = {"ld sc1", "ld sc0 sc1", "ld sc0", "ld nt", "ld sc1 nt", "atomic_or(0) rtn", "buffer_inv sc1 + ld", "buffer_inv sc0 + ld"};
static const char* kSt[] = {"st plain", "st sc1", "st sc0 sc1", "st nt", "st sc0", "st + wb sc1"};

template <int LD> struct Ld;
#define DEF_LD(ID, PRE, SUF)                                                                                             \
  template <> struct Ld<ID> {                                                                                            \
    static __device__ __forceinline__ u32x2 one(const unsigned long long* p) {                                           \
      u32x2 v;                                                                                                           \
      asm volatile(PRE "global_load_dwordx2 %0, %1, off " SUF "\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");   \
      return v;                                                                                                          \
    }                                                                                                                    \
    static __device__ __forceinline__ void four(const unsigned long long* p, u32x4& a, u32x4& b) { /* 4 granules */      \
      asm volatile(PRE "global_load_dwordx4 %0, %2, off " SUF "\n global_load_dwordx4 %1, %2, off offset:16 " SUF       \
                       "\n s_waitcnt vmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(p) : "memory");                               \
    }                                                                                                                    \
  };
DEF_LD(LD_SC1, "", "sc1")
DEF_LD(LD_SC0SC1, "", "sc0 sc1")
DEF_LD(LD_SC0, "", "sc0")
DEF_LD(LD_NT, "", "nt")
DEF_LD(LD_SC1NT, "", "sc1 nt")
DEF_LD(LD_INV_PLAIN, "buffer_inv sc1\n ", "")
DEF_LD(LD_INV0_PLAIN, "buffer_inv sc0\n ", "")
template <> struct Ld<LD_ATOMIC> {
  static __device__ __forceinline__ u32x2 one(const unsigned long long* p) {
    u32x2 v, z = {0u, 0u};
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
    return v;
  }
  static __device__ __forceinline__ void four(const unsigned long long* p, u32x4& a, u32x4& b) {
    u32x2 v0, v1, v2, v3, z = {0u, 0u};
    asm volatile("global_atomic_or_x2 %0, %4, %5, off sc0\n global_atomic_or_x2 %1, %4, %5, off offset:8 sc0\n"
                 "global_atomic_or_x2 %2, %4, %5, off offset:16 sc0\n global_atomic_or_x2 %3, %4, %5, off offset:24 sc0\n"
                 "s_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p), "v"(z) : "memory");
    a = u32x4{v0.x, v0.y, v1.x, v1.y};
    b = u32x4{v2.x, v2.y, v3.x, v3.y};
  }
};
template <int LD>
__device__ __forceinline__ u32x2 poll_load(const unsigned long long* p) { return Ld<LD>::one(p); }
template <int ST>
__device__ __forceinline__ void publish(unsigned long long* p, u32x2 v) {
  if constexpr (ST == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if constexpr (ST == ST_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (ST == ST_SC0SC1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if constexpr (ST == ST_NT) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if constexpr (ST == ST_SC0) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if constexpr (ST == ST_WB) asm volatile("global_store_dwordx2 %0, %1, off\n buffer_wbl2 sc1" ::"v"(p), "v"(v) : "memory");
}

// blocks a and b play; every other block exits.  mine / theirs: 128-byte-separated granules.  out[0] = cycles per round
// trip (0 = no progress), out[1] = xcc ids (a | b << 8), out[2] = polls per hand-off x 16
template <int LD, int ST>
__global__ void pingpong(unsigned long long* area, int a, int b, int rounds, unsigned base, long long* out) {
  const int bid = blockIdx.x;
  if (bid != a && bid != b) return;
  if (threadIdx.x != 0) return;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 0xF;
  const bool first = bid == a;
  unsigned long long* mine = area + (first ? 0 : 16);
  const unsigned long long* theirs = area + (first ? 16 : 0);
  long long polls = 0;
  const long long t0 = __builtin_readcyclecounter();
  bool ok = true;
  for (int r = 1; r <= rounds && ok; r++) {
    const unsigned tag = base + (unsigned)r;
    if (first) publish<ST>(mine, u32x2{(unsigned)r, tag});
    unsigned spins = 0;
    while (true) {
      const u32x2 v = poll_load<LD>(theirs);
      polls++;
      if (v.y == tag) break;
      if (++spins > 200000u) { ok = false; break; }
    }
    if (!first && ok) publish<ST>(mine, u32x2{(unsigned)r, tag});
  }
  const long long t1 = __builtin_readcyclecounter();
  if (first) {
    out[0] = ok ? (t1 - t0) / rounds : 0;
    out[2] = polls * 16 / rounds;
    atomicOr((unsigned long long*)&out[1], (unsigned long long)xcc);
  } else {
    atomicOr((unsigned long long*)&out[1], (unsigned long long)xcc << 8);
    if (!ok) out[3] = 1;
  }
}

// "gather" flavour: NW publisher blocks (one granule each per step) and every block's wave 0 polls ALL NW*? granules with
// 16-byte loads, as gru_ring_kernel does: H = 256 granules, lane l reads granules 4l .. 4l+3 (two loads).  All blocks
// step in lock-step; out[0] = cycles per step.
template <int LD, int ST>
__global__ void allgather(unsigned long long* area, int nwg, int stride8, int steps, unsigned base, long long* out) {
  // participating blocks: bid % 8 == 0 (one XCD) when stride8, else the first nwg blocks (spread over XCDs)
  const int bid = blockIdx.x;
  int g;
  if (stride8) { if (bid & 7) return; g = bid >> 3; } else g = bid;
  if (g >= nwg) return;
  const int lane = threadIdx.x;  // 64 threads
  const int H = 256, upw = H / nwg;  // this block publishes granules g*upw .. +upw (lanes < upw)
  bool ok = true;
  const long long t0 = __builtin_readcyclecounter();
  for (int s = 1; s <= steps && ok; s++) {
    const unsigned tag = base + (unsigned)s;
    unsigned long long* buf = area + (size_t)(s & 1) * H;
    if (lane < upw) publish<ST>(buf + g * upw + lane, u32x2{(unsigned)s, tag});
    unsigned spins = 0;
    while (true) {
      u32x4 va, vb;
      Ld<LD>::four(buf + lane * 4, va, vb);
      const bool all = va.y == tag && va.w == tag && vb.y == tag && vb.w == tag;
      if (__builtin_amdgcn_ballot_w64(!all) == 0ull) break;
      if (++spins > 100000u) { ok = false; break; }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (g == 0 && lane == 0) out[0] = ok ? (t1 - t0) / steps : 0;
  if (!ok && lane == 0) out[3] = 1;
}

template <int LD, int ST>
static void run(unsigned long long* area, long long* out, unsigned& base) {
  const int rounds = 2000;
  long long h[3][4];
  const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};  // same XCD (block i -> XCD i % 8), neighbouring XCD, far XCD
  for (int k = 0; k < 3; k++) {
    CHECK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL((pingpong<LD, ST>), dim3(16), dim3(64), 0, 0, area, pairs[k][0], pairs[k][1], rounds, base, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h[k], out, 32, hipMemcpyDeviceToHost));
    base += rounds + 8;
  }
  long long g[2][4];
  for (int k = 0; k < 2; k++) {
    CHECK(hipMemset(out, 0, 64));
    hipLaunchKernelGGL((allgather<LD, ST>), dim3(16 * 8), dim3(64), 0, 0, area + 64, 16, k == 0 ? 1 : 0, 1000, base, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(g[k], out, 32, hipMemcpyDeviceToHost));
    base += 1008;
  }
  printf("%-22s %-12s | round trip cycles: sameXCD %5lld (xcc %lld/%lld, %4.1f polls)  nextXCD %5lld (xcc %lld/%lld)  farXCD %5lld"
         " | 16-block all-gather step: oneXCD %5lld  spread %5lld\n",
         kLd[LD], kSt[ST], h[0][0], h[0][1] & 0xff, h[0][1] >> 8, h[0][2] / 16.0, h[1][0], h[1][1] & 0xff, h[1][1] >> 8,
         h[2][0], g[0][0], g[1][0]);
  fflush(stdout);
}

int main() {
  unsigned long long* area;
  long long* out;
  CHECK(hipMalloc(&area, 1 << 16));
  CHECK(hipMemset(area, 0, 1 << 16));
  CHECK(hipMalloc(&out, 64));
  unsigned base = 1;
  int clk = 0;
  CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  printf("shader clock %d kHz; cycles below are shader clocks (s_memtime)\n", clk);
  run<LD_SC1, ST_PLAIN>(area, out, base);
  run<LD_SC1, ST_SC1>(area, out, base);
  run<LD_SC1, ST_SC0SC1>(area, out, base);
  run<LD_SC1, ST_NT>(area, out, base);
  run<LD_SC0SC1, ST_PLAIN>(area, out, base);
  run<LD_SC0SC1, ST_SC0SC1>(area, out, base);
  run<LD_SC1NT, ST_PLAIN>(area, out, base);
  run<LD_NT, ST_PLAIN>(area, out, base);
  run<LD_NT, ST_NT>(area, out, base);
  run<LD_ATOMIC, ST_PLAIN>(area, out, base);
  run<LD_INV_PLAIN, ST_PLAIN>(area, out, base);
  run<LD_INV_PLAIN, ST_WB>(area, out, base);
  run<LD_INV0_PLAIN, ST_PLAIN>(area, out, base);
  run<LD_SC0, ST_PLAIN>(area, out, base);
  run<LD_SC0, ST_SC0>(area, out, base);
  return 0;
}
