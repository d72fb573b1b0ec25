// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: the no-split-K throughput kernels (conv_direct3_kernel,
// conv_direct3s_kernel): one (16 TM) x 64 tile per wave over the whole reduction, for launches with many output columns.
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ou {

// ---------------------------------------------------------------------------------------------------------
// conv_direct3_kernel: stride-1 k3 / k5 convs with MANY output columns (batch x length in the hundreds of thousands) --
// the throughput regime.  No split-K, no LDS, no barrier, no cross-wave reduction: every WAVE owns a (16 TM) x 64 output
// tile over the whole reduction and stores it straight from its accumulators.
//   v_mfma_f32_16x16x4_f32 (same 64 FLOP/clk/SIMD as 32x32x2, 16-row granularity: the 48 / 96 / 192-channel levels of
//   UNIVERSE++ 24 kHz tile exactly):  A lane (m, kk) = W[m0 + 16 i + m][4 J + kk][tap],  B lane (n, kk) = x[4 J + kk][..],
//   D lane (n, q) reg r = out[m0 + 16 i + 4 q + r][n0 + 4 n + j]  -- output columns are interleaved over the TN = 4
//   accumulator tiles (column n0 + 4 n + j), so that
//     * ONE 16-byte load (+ an 8- / 16-byte one) gives a lane the 4 + KW - 1 consecutive samples it needs for all taps of
//       its four columns (as in conv_direct2_kernel), one 16-byte load from the taps-innermost weight copy all taps of a row;
//     * the epilogue stores 16 bytes per lane and row: four adjacent samples, 256 contiguous bytes per 16 lanes.
//   Per ring slot (4 input channels): TM (k3) / 2 TM (k5) + 2 load instructions for 4 KW TM MFMAs (48 / 80 at TM = 4):
//   0.13 loads per MFMA, ~16 B/clk/CU of L1 traffic -- the kernel is bound by the matrix pipe, 2 waves per SIMD.
//   Block = 4 waves = 4 adjacent column tiles; blocks of one column chunk (all row groups) run on ONE XCD back to back
//   (the activations are fetched into one L2, once), weights are L2-resident everywhere.
//   Summation order per output: channel groups ascending, taps ascending, the 4 channels of a group in MFMA order -- fixed,
//   but different from the split-K kernels (results agree to fp32 rounding).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4acc __attribute__((ext_vector_type(4)));
template <int KW, int TM, int D, bool PRE>
__global__ __launch_bounds__(256, 2) void conv_direct3_kernel(ConvArgs p) {
  constexpr int TN = 4, W = KW + TN - 1, KWP = KW == 3 ? 4 : 8, PAD = (KW - 1) / 2;
  constexpr int A2 = KW == 5 ? 1 : 0;       // second A load per row tile (tap 4)
  constexpr int LPS = TM * (1 + A2) + 2;    // load instructions per ring slot
  static_assert(KW == 3 || KW == 5, "k3 / k5");
  static_assert(D * LPS <= 60, "loads in flight must fit vmcnt");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (column chunk, row group): blocks L, L + 8, L + 16, ... (one XCD) walk the row groups of one chunk
  // (batch element, chunk) pairs are numbered through -- a short signal has only a chunk or two, and eight of those pairs,
  // not eight chunks of one element, are what is spread over the XCDs
  const int L = blockIdx.x, q8 = L >> 3, rg = q8 % p.grid_m, cidx = (q8 / p.grid_m) * 8 + (L & 7);
  const int b = cidx / p.grid_n, chunk = cidx - b * p.grid_n;
  const int n0 = (chunk * 4 + wv) * 64, m0 = rg * (16 * TM);
  if (b >= p.B || n0 >= p.Nq) return;  // (whole waves: nothing in this kernel synchronises)
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int l15 = lane & 15, kk = lane >> 4;
  const int Tin = p.Tin, Mp = p.Mp;
  const float alpha = p.act ? p.alpha_val : 1.0f;
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin, (unsigned)p.Cin * (unsigned)Tin * 4u);
  const u32x4 rw = direct_desc(p.wd, (unsigned)p.Cin * (unsigned)Mp * (unsigned)KWP * 4u);
  const int avo = (kk * Mp + m0 + l15) * KWP * 4;
  // this lane's window: samples t0 .. t0 + W - 1 of channel 4 J + kk; `sh` = samples cut off in front of the row
  const int t0 = n0 + TN * l15 - PAD;
  const int sh = t0 < 0 ? -t0 : 0;
  const int bvo = (t0 + sh < Tin) ? (kk * Tin + t0 + sh) * 4 : (int)0x80000000;
  const bool edge = __builtin_amdgcn_readfirstlane((n0 < PAD || n0 + 64 + KW - 1 - PAD > Tin) ? 1 : 0) != 0;
  unsigned vmask = 0;  // bit i: window element i is inside the row
#pragma unroll
  for (int i = 0; i < W; i++) vmask |= (t0 + i >= 0 && t0 + i < Tin) ? (1u << i) : 0u;

  const int NG = p.Cin >> 2;  // ring slots (launcher: a multiple of D)
  f32x4 a4[D][TM], b4[D], b4b[D];
  float a1[D][TM];
  f32x2 b2[D];
#pragma unroll
  for (int d0 = 0; d0 < D; d0++) {
#pragma unroll
    for (int i = 0; i < TM; i++) { a4[d0][i] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[d0][i] = 0.f; }
    b4[d0] = f32x4{0.f, 0.f, 0.f, 0.f}; b4b[d0] = f32x4{0.f, 0.f, 0.f, 0.f}; b2[d0] = f32x2{0.f, 0.f};
  }
  f32x4acc acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

#define OU_ISSUE(g_, d)                                                                                               \
  {                                                                                                                   \
    const int aso = (g_) * 4 * Mp * KWP * 4, xso = (g_) * 4 * Tin * 4;                                                \
    _Pragma("unroll") for (int i = 0; i < TM; i++) {                                                                  \
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"                                               \
                   : "+v"(a4[d][i]) : "v"(avo), "s"(rw), "s"(aso), "n"(i * 16 * KWP * 4));                            \
      if constexpr (A2 == 1)                                                                                          \
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4"                                               \
                     : "+v"(a1[d][i]) : "v"(avo), "s"(rw), "s"(aso), "n"(i * 16 * KWP * 4 + 16));                     \
    }                                                                                                                 \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(b4[d]) : "v"(bvo), "s"(rx), "s"(xso));             \
    if constexpr (KW == 3)                                                                                            \
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:16" : "+v"(b2[d]) : "v"(bvo), "s"(rx), "s"(xso)); \
    else                                                                                                              \
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "+v"(b4b[d]) : "v"(bvo), "s"(rx), "s"(xso)); \
  }
#define OU_MMA(d, out) OU_MMAX(d, out, 0)
#define OU_MMAX(d, out, extra)                                                                                        \
  {                                                                                                                   \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPS + (extra)));                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; i++) {                                                                  \
      asm volatile("" : "+v"(a4[d][i]));                                                                              \
      if constexpr (A2 == 1) asm volatile("" : "+v"(a1[d][i]));                                                       \
    }                                                                                                                 \
    asm volatile("" : "+v"(b4[d]));                                                                                   \
    if constexpr (KW == 3) asm volatile("" : "+v"(b2[d]));                                                            \
    else asm volatile("" : "+v"(b4b[d]));                                                                             \
    const float Lw[8] = {b4[d].x, b4[d].y, b4[d].z, b4[d].w, KW == 3 ? b2[d].x : b4b[d].x, KW == 3 ? b2[d].y : b4b[d].y, \
                         b4b[d].z, b4b[d].w};                                                                         \
    float X[W];                                                                                                       \
    if (edge) {                                                                                                       \
      _Pragma("unroll") for (int i = 0; i < W; i++) {                                                                 \
        float v = Lw[i];                                                                                              \
        _Pragma("unroll") for (int s2 = 1; s2 <= PAD; s2++) v = sh == s2 ? (i - s2 >= 0 ? Lw[i - s2 >= 0 ? i - s2 : 0] : 0.f) : v; \
        X[i] = ((vmask >> i) & 1u) ? v : 0.f;                                                                         \
      }                                                                                                               \
    } else {                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < W; i++) X[i] = Lw[i];                                                     \
    }                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < W; i++) X[i] = X[i] >= 0.f ? X[i] : alpha * X[i];                           \
    _Pragma("unroll") for (int k = 0; k < KW; k++)                                                                    \
      _Pragma("unroll") for (int i = 0; i < TM; i++) {                                                                \
        const float av = k == 0 ? a4[d][i].x : (k == 1 ? a4[d][i].y : (k == 2 ? a4[d][i].z : (k == 3 ? a4[d][i].w : a1[d][i]))); \
        _Pragma("unroll") for (int j = 0; j < TN; j++)                                                                \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, X[j + k], acc[i][j], 0, 0, 0);                         \
      }                                                                                                               \
  }
  static_assert(D == 4, "ring depth");
  // The epilogue's tensor operand (the residual, or the cond add when there is no residual) is as large as the output: read
  // after the main loop its 16 KB per tile are pure exposed latency / bandwidth (the `.v` layers ran 5-20 us behind their
  // residual-free twins).  It is PREFETCHED into a wave-private LDS slab with LDS-DMA -- no registers, LDS is otherwise unused
  // here -- right before the last four ring slots, i.e. under 4 KW TM 4 = 192-320 MFMAs; every lane fetches exactly the
  // 4 TM quads it will consume (instruction (i, r): row m0 + 16 i + 4 kk + r, columns c0 .. c0 + 3 -> LDS slab (4 i + r) KB +
  // 16 lane), so the read-back is conflict-free and needs no barrier.  The DMA loads count in vmcnt like any load: the
  // counted waits of the drain carry them (NDMA younger loads still in flight).
  extern __shared__ __attribute__((aligned(16))) float smem3[];
  constexpr int NDMA = 4 * TM;
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
  const int c0 = n0 + TN * l15;
  int ncol = p.Nq - c0;
  if (ncol > 4) ncol = 4;
  const bool vec4 = ncol == 4;  // (16-byte accesses at dword alignment: rows of 2005 frames too)
  const int rlen = ragged_len(p.lens, b);  // ragged batch: this row's own length (stride 1: Tout grid == column grid)
  // PRE (chosen by the launcher: an operand exists, rows are 16-byte multiples -- then every lane has a whole quad or none --
  // and the LDS was provided).  A template parameter, not a branch: a branch here would split the control flow while ring
  // loads are in flight, and the copies the compiler places at the join read registers whose data has not landed.
  const float* pre = p.res ? p.res : p.add;  // the operand that is prefetched
  constexpr bool pre_on = PRE;
  float* const slab = smem3 + wv * (NDMA * 256);
  {
    OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
    const int NR = NG / 4;
    for (int r = 0; r + 1 < NR; r++) {
      const int g = r * 4;
      OU_MMA(0, 3); OU_ISSUE(g + 4, 0);
      OU_MMA(1, 3); OU_ISSUE(g + 5, 1);
      OU_MMA(2, 3); OU_ISSUE(g + 6, 2);
      OU_MMA(3, 3); OU_ISSUE(g + 7, 3);
    }
    if constexpr (PRE) {
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(pre + ybase, (unsigned)p.Cout * (unsigned)p.Tout * 4u);
      const int pvo = ncol > 0 ? ((m0 + 4 * kk) * p.Tout + c0) * 4 : (int)0x80000000;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          dma_b128(rp, slab + (4 * i + r) * 256, pvo, (16 * i + r) * p.Tout * 4);
      asm volatile("" ::: "memory");
      OU_MMAX(0, 3, NDMA); OU_MMAX(1, 2, NDMA); OU_MMAX(2, 1, NDMA); OU_MMAX(3, 0, NDMA);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      OU_MMA(0, 3); OU_MMA(1, 2); OU_MMA(2, 1); OU_MMA(3, 0);
    }
  }
#undef OU_ISSUE
#undef OU_MMA
#undef OU_MMAX

  // ---- epilogue: bias, cond add, FiLM, residual -- straight from the accumulators, 16 bytes per lane and row
  if (ncol > 0) {
#pragma unroll
    for (int i = 0; i < TM; i++) {
      f32x4 ad[4], rs[4];
      float bi[4], ga[4], be[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + 16 * i + 4 * kk + r;
        const bool on = row < p.M;
        const size_t idx = ybase + (size_t)(on ? row : 0) * p.Tout + c0;
        ad[r] = f32x4{0.f, 0.f, 0.f, 0.f}; rs[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        bi[r] = on ? p.bias[row] : 0.f;
        ga[r] = 1.f; be[r] = 0.f;
        if (on && filmb) { ga[r] = filmb[row]; be[r] = filmb[p.Cout + row]; }
        if (on && vec4) {
          const f32x4 pq = pre_on ? *reinterpret_cast<const f32x4*>(slab + (4 * i + r) * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.add) ad[r] = (pre_on && !p.res) ? pq : f32x4(*reinterpret_cast<const f32x4u*>(p.add + idx));
          if (p.res) rs[r] = pre_on ? pq : f32x4(*reinterpret_cast<const f32x4u*>(p.res + idx));
        } else if (on) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (p.add && j < ncol) ad[r][j] = p.add[idx + j];
            if (p.res && j < ncol) rs[r][j] = p.res[idx + j];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + 16 * i + 4 * kk + r;
        if (row >= p.M) continue;
        const size_t idx = ybase + (size_t)row * p.Tout + c0;
        f32x4 v = f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
        if (p.in_scale) v *= insc;
        v += bi[r];
        if (p.add) v = (v + ad[r]) * p.add_scale;
        if (filmb) v = ga[r] * v + be[r];
        if (p.res) v = (v + rs[r]) * p.res_scale;
        if (p.out_act) {
#pragma unroll
          for (int oa = 0; oa < 4; oa++) v[oa] = v[oa] >= 0.f ? v[oa] : p.out_alpha * v[oa];
        }
        if (p.lens) v = ragged_mask4(v, c0, rlen);
        if (vec4) {
          *reinterpret_cast<f32x4u*>(p.y + idx) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (j < ncol) p.y[idx + j] = v[j];
        }
      }
    }
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// ---------------------------------------------------------------------------------------------------------
// conv_direct3w_kernel: conv_direct3_kernel with MINIMAL FILTERING F(2, KW) (round 5; see conv_direct2w_kernel for the scheme
// and its numerics).  A lane's four adjacent output columns are two tile positions p = 0, 1; its window of KW + 3 samples --
// the very loads of conv_direct3_kernel -- holds both input tiles d_p = window[2 p .. 2 p + KW]; A lane (m, kk) loads the KW + 1
// Winograd-domain values U_x of (row m0 + 16 i + m, channel 4 J + kk) from the third weight copy (16 bytes for k3 -- what the
// taps took --, 16 + 8 for k5).  Per ring slot and 32 x 64 wave tile: 2 x 2 x (KW + 1) MFMAs instead of 2 x 4 x KW (16 / 24, was
// 24 / 40) on 2 x 2 x (KW + 1) independent 16 x 16 accumulators; A^T is applied in the epilogue, which then finishes quads of
// four adjacent samples exactly as before.  32-row tiles only (TM = 2): the accumulators of a 64-row tile would not leave room
// for the ring (k3: 64 + 56 ring registers at TM = 2, 128 + 88 at TM = 4).
// ---------------------------------------------------------------------------------------------------------
// (EDGE / ACT: whole-function variants behind the kernel's wave-uniform branch, see direct2w_tile)
template <int KW, int TM, bool PRE, bool EDGE, bool ACT>
__device__ __forceinline__ void direct3w_body(const ConvArgs& p) {
  constexpr int D = 4, TN = 4, NX = KW + 1, W = KW + TN - 1, KWP = KW == 3 ? 4 : 8, PAD = (KW - 1) / 2;
  constexpr int A2 = KW == 5 ? 1 : 0;       // second A load per row tile (U_4, U_5)
  constexpr int LPS = TM * (1 + A2) + 2;    // load instructions per ring slot
  static_assert(KW == 3 || KW == 5, "k3 / k5");
  static_assert(D * LPS <= 60, "loads in flight must fit vmcnt");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = blockIdx.x, q8 = L >> 3, rg = q8 % p.grid_m, cidx = (q8 / p.grid_m) * 8 + (L & 7);
  const int b = cidx / p.grid_n, chunk = cidx - b * p.grid_n;
  const int n0 = (chunk * 4 + wv) * 64, m0 = rg * (16 * TM);
  if (b >= p.B || n0 >= p.Nq) return;  // (whole waves: nothing in this kernel synchronises)
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int l15 = lane & 15, kk = lane >> 4;
  const int Tin = p.Tin, Mp = p.Mp;
  const float alpha = p.act ? p.alpha_val : 1.0f;
  // (the descriptor starts PAD samples in front of the tensor -- workspace memory, never the first bytes of an allocation --: the
  // window of the first lane of the first tile, which begins at t = -PAD, is an in-range load whose leading elements are masked)
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin - PAD, ((unsigned)p.Cin * (unsigned)Tin + PAD) * 4u);
  const u32x4 rw = direct_desc(p.wu, (unsigned)p.Cin * (unsigned)Mp * (unsigned)KWP * 4u);
  const int avo = (kk * Mp + m0 + l15) * KWP * 4;
  const int t0 = n0 + TN * l15 - PAD;
  const int bvo = (t0 < Tin) ? (kk * Tin + t0 + PAD) * 4 : (int)0x80000000;
  unsigned vmask = 0;  // bit i: window element i is inside the row
#pragma unroll
  for (int i = 0; i < W; i++) vmask |= (t0 + i >= 0 && t0 + i < Tin) ? (1u << i) : 0u;

  const int NG = p.Cin >> 2;  // ring slots (launcher: a multiple of D)
  f32x4 a4[D][TM], b4[D], b4b[D];
  f32x2 a2[D][TM], b2[D];
#pragma unroll
  for (int d0 = 0; d0 < D; d0++) {
#pragma unroll
    for (int i = 0; i < TM; i++) { a4[d0][i] = f32x4{0.f, 0.f, 0.f, 0.f}; a2[d0][i] = f32x2{0.f, 0.f}; }
    b4[d0] = f32x4{0.f, 0.f, 0.f, 0.f}; b4b[d0] = f32x4{0.f, 0.f, 0.f, 0.f}; b2[d0] = f32x2{0.f, 0.f};
  }
  f32x4acc acc[TM][2][NX];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int x = 0; x < NX; x++) acc[i][q][x] = f32x4acc{0.f, 0.f, 0.f, 0.f};

#define OU_ISSUE(g_, d)                                                                                               \
  {                                                                                                                   \
    const int aso = (g_) * 4 * Mp * KWP * 4, xso = (g_) * 4 * Tin * 4;                                                \
    _Pragma("unroll") for (int i = 0; i < TM; i++) {                                                                  \
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"                                               \
                   : "+v"(a4[d][i]) : "v"(avo), "s"(rw), "s"(aso), "n"(i * 16 * KWP * 4));                            \
      if constexpr (A2 == 1)                                                                                          \
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4"                                             \
                     : "+v"(a2[d][i]) : "v"(avo), "s"(rw), "s"(aso), "n"(i * 16 * KWP * 4 + 16));                     \
    }                                                                                                                 \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(b4[d]) : "v"(bvo), "s"(rx), "s"(xso));             \
    if constexpr (KW == 3)                                                                                            \
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:16" : "+v"(b2[d]) : "v"(bvo), "s"(rx), "s"(xso)); \
    else                                                                                                              \
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "+v"(b4b[d]) : "v"(bvo), "s"(rx), "s"(xso)); \
  }
#define OU_MMA(d, out) OU_MMAX(d, out, 0)
#define OU_MMAX(d, out, extra)                                                                                        \
  {                                                                                                                   \
    if (ts_on) { const long long ta = __builtin_readcyclecounter();                                                  \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPS + (extra)) : "memory");                                    \
      twait += (unsigned)(__builtin_readcyclecounter() - ta); }                                                       \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPS + (extra)));                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; i++) {                                                                  \
      asm volatile("" : "+v"(a4[d][i]));                                                                              \
      if constexpr (A2 == 1) asm volatile("" : "+v"(a2[d][i]));                                                       \
    }                                                                                                                 \
    asm volatile("" : "+v"(b4[d]));                                                                                   \
    if constexpr (KW == 3) asm volatile("" : "+v"(b2[d]));                                                            \
    else asm volatile("" : "+v"(b4b[d]));                                                                             \
    const float Lw[8] = {b4[d].x, b4[d].y, b4[d].z, b4[d].w, KW == 3 ? b2[d].x : b4b[d].x, KW == 3 ? b2[d].y : b4b[d].y, \
                         b4b[d].z, b4b[d].w};                                                                         \
    float X[W];                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < W; i++) {                                                                   \
      X[i] = Lw[i];                                                                                                   \
      if constexpr (EDGE) X[i] = ((vmask >> i) & 1u) ? X[i] : 0.f;                                                    \
      if constexpr (ACT) X[i] = X[i] >= 0.f ? X[i] : alpha * X[i];                                                    \
    }                                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 2; q++) {                                                                   \
      float V[NX];                                                                                                    \
      wino_bt<KW>(X + 2 * q, V);                                                                                      \
      _Pragma("unroll") for (int i = 0; i < TM; i++)                                                                  \
        _Pragma("unroll") for (int x = 0; x < NX; x++) {                                                              \
          const float av = x == 0 ? a4[d][i].x : (x == 1 ? a4[d][i].y : (x == 2 ? a4[d][i].z : (x == 3 ? a4[d][i].w : (x == 4 ? a2[d][i].x : a2[d][i].y)))); \
          acc[i][q][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, V[x], acc[i][q][x], 0, 0, 0);                       \
        }                                                                                                             \
    }                                                                                                                 \
  }
  extern __shared__ __attribute__((aligned(16))) float smem3[];
  constexpr int NDMA = 4 * TM;
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
  const int c0 = n0 + TN * l15;
  int ncol = p.Nq - c0;
  if (ncol > 4) ncol = 4;
  const bool vec4 = ncol == 4;
  const int rlen = ragged_len(p.lens, b);  // ragged batch: this row's own length
  const float* pre = p.res ? p.res : p.add;  // the operand that is prefetched (PRE: see conv_direct3_kernel)
  constexpr bool pre_on = PRE;
  float* const slab = smem3 + wv * (NDMA * 256);
  // tuning only (OU_TS, tools/d3_ts.py): cycles in the loop and inside its s_waitcnt vmcnt
  const bool ts_on = p.tstamps != nullptr;
  unsigned twait = 0;
  long long tl0 = 0, tr0 = 0;
  if (ts_on) { tr0 = (long long)__builtin_amdgcn_s_memrealtime(); tl0 = __builtin_readcyclecounter(); }
  {
    OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
    const int NR = NG / 4;
    for (int r = 0; r + 1 < NR; r++) {
      const int g = r * 4;
      OU_MMA(0, 3); OU_ISSUE(g + 4, 0);
      OU_MMA(1, 3); OU_ISSUE(g + 5, 1);
      OU_MMA(2, 3); OU_ISSUE(g + 6, 2);
      OU_MMA(3, 3); OU_ISSUE(g + 7, 3);
    }
    if constexpr (PRE) {
      const __amdgpu_buffer_rsrc_t rp = make_rsrc(pre + ybase, (unsigned)p.Cout * (unsigned)p.Tout * 4u);
      const int pvo = ncol > 0 ? ((m0 + 4 * kk) * p.Tout + c0) * 4 : (int)0x80000000;
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          dma_b128(rp, slab + (4 * i + r) * 256, pvo, (16 * i + r) * p.Tout * 4);
      asm volatile("" ::: "memory");
      OU_MMAX(0, 3, NDMA); OU_MMAX(1, 2, NDMA); OU_MMAX(2, 1, NDMA); OU_MMAX(3, 0, NDMA);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      OU_MMA(0, 3); OU_MMA(1, 2); OU_MMA(2, 1); OU_MMA(3, 0);
    }
  }
#undef OU_ISSUE
#undef OU_MMA
#undef OU_MMAX
  const long long tl1 = ts_on ? __builtin_readcyclecounter() : 0;

  // ---- epilogue: A^T, then bias, cond add, FiLM, residual -- straight from the accumulators, 16 bytes per lane and row
  if (ncol > 0) {
#pragma unroll
    for (int i = 0; i < TM; i++) {
      f32x4 ad[4], rs[4];
      float bi[4], ga[4], be[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + 16 * i + 4 * kk + r;
        const bool on = row < p.M;
        const size_t idx = ybase + (size_t)(on ? row : 0) * p.Tout + c0;
        ad[r] = f32x4{0.f, 0.f, 0.f, 0.f}; rs[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        bi[r] = on ? p.bias[row] : 0.f;
        ga[r] = 1.f; be[r] = 0.f;
        if (on && filmb) { ga[r] = filmb[row]; be[r] = filmb[p.Cout + row]; }
        if (on && vec4) {
          const f32x4 pq = pre_on ? *reinterpret_cast<const f32x4*>(slab + (4 * i + r) * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (p.add) ad[r] = (pre_on && !p.res) ? pq : f32x4(*reinterpret_cast<const f32x4u*>(p.add + idx));
          if (p.res) rs[r] = pre_on ? pq : f32x4(*reinterpret_cast<const f32x4u*>(p.res + idx));
        } else if (on) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (p.add && j < ncol) ad[r][j] = p.add[idx + j];
            if (p.res && j < ncol) rs[r][j] = p.res[idx + j];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + 16 * i + 4 * kk + r;
        if (row >= p.M) continue;
        const size_t idx = ybase + (size_t)row * p.Tout + c0;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if constexpr (KW == 3) {
            v[2 * q] = acc[i][q][0][r] + acc[i][q][1][r] + acc[i][q][2][r];
            v[2 * q + 1] = acc[i][q][1][r] - acc[i][q][2][r] - acc[i][q][3][r];
          } else {
            v[2 * q] = acc[i][q][0][r] + acc[i][q][1][r] + acc[i][q][2][r] + acc[i][q][3][r] + acc[i][q][4][r];
            v[2 * q + 1] = acc[i][q][1][r] - acc[i][q][2][r] + 0.5f * acc[i][q][3][r] - 2.0f * acc[i][q][4][r] + acc[i][q][5][r];
          }
        }
        if (p.in_scale) v *= insc;
        v += bi[r];
        if (p.add) v = (v + ad[r]) * p.add_scale;
        if (filmb) v = ga[r] * v + be[r];
        if (p.res) v = (v + rs[r]) * p.res_scale;
        if (p.out_act) {
#pragma unroll
          for (int oa = 0; oa < 4; oa++) v[oa] = v[oa] >= 0.f ? v[oa] : p.out_alpha * v[oa];
        }
        if (p.lens) v = ragged_mask4(v, c0, rlen);
        if (vec4) {
          *reinterpret_cast<f32x4u*>(p.y + idx) = v;
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (j < ncol) p.y[idx + j] = v[j];
        }
      }
    }
  }
  if (ts_on && lane == 0 && blockIdx.x < 2048) {  // {start ticks, loop cycles, cycles in s_waitcnt vmcnt, epilogue cycles, -, -, -, end ticks}
    long long* o = p.tstamps + ((size_t)blockIdx.x * 4 + wv) * 8;
    o[0] = tr0; o[1] = tl1 - tl0; o[2] = twait; o[3] = __builtin_readcyclecounter() - tl1; o[4] = 0; o[5] = 0; o[6] = 0;
    o[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

template <int KW, int TM, bool PRE>
__global__ __launch_bounds__(256, 2) void conv_direct3w_kernel(ConvArgs p) {
  constexpr int PAD = (KW - 1) / 2;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q8 = blockIdx.x >> 3, cidx = (q8 / p.grid_m) * 8 + (blockIdx.x & 7);
  const int chunk = cidx - (cidx / p.grid_n) * p.grid_n;
  const int n0 = (chunk * 4 + wv) * 64;
  const bool edge = n0 < PAD || n0 + 64 + KW - 1 - PAD > p.Tin;  // (wave-uniform)
  if (edge) {
    if (p.act) direct3w_body<KW, TM, PRE, true, true>(p);
    else direct3w_body<KW, TM, PRE, true, false>(p);
  } else {
    if (p.act) direct3w_body<KW, TM, PRE, false, true>(p);
    else direct3w_body<KW, TM, PRE, false, false>(p);
  }
}

// ---------------------------------------------------------------------------------------------------------
// conv_direct3s_kernel<R>: the same per-wave scheme for the layers WITHOUT a taps-innermost weight copy, many columns:
//   R = 1: 1x1 convs and transposed convs as `up` phase GEMMs (row m = co * up + phase);
//   R > 1: rate-change (down) convs, k = s = R, whole frames (Tin = Nq * R).
// Operands from the tap-major packed weights [Cin/CK][R][CK][Mp] and the activations:
//   A (tap k, 4 channels 4 J + kk): lane (m, kk) loads FOUR ADJACENT ROWS m0 + 4 m .. + 3 of weight row (channel, tap) with
//     one 16-byte load -- the four 16-row accumulator tiles are row-INTERLEAVED (tile i holds rows m0 + 4 m + i), so one load
//     feeds all four;
//   B: lane (n, kk) loads the 4 R consecutive samples of its four adjacent output columns (R 16-byte loads); column j, tap k
//     is window element j R + k.
//   D tile (i, j): lane (n, q) reg r = out[m0 + 4 (4 q + r) + i][n0 + 4 n + j].  For a phase GEMM with up = 4 that is
//   channel (m0 / 4 + 4 q + r), phase i, frame n0 + 4 n + j: the lane's 16 values of one channel are 16 CONSECUTIVE output
//   samples (64-byte stores); up = 2 / 8 likewise in runs of 8 / 32; other rates store sample by sample.
//   2 R load instructions per 16 R MFMAs.  The up path's anti-alias FIR stays a separate pass behind this kernel.
// ---------------------------------------------------------------------------------------------------------
template <int R, int D>
__global__ __launch_bounds__(256, 2) void conv_direct3s_kernel(ConvArgs p) {
  constexpr int TM = 4, TN = 4, LPS = 2 * R;
  static_assert(D * LPS <= 60, "loads in flight must fit vmcnt");
  static_assert(D == 2 || D == 4, "ring depth");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = blockIdx.x, q8 = L >> 3, rg = q8 % p.grid_m, cidx = (q8 / p.grid_m) * 8 + (L & 7);
  const int b = cidx / p.grid_n, chunk = cidx - b * p.grid_n;  // (batch element, chunk) pairs numbered through
  const int n0 = (chunk * 4 + wv) * 64, m0 = rg * 64;
  if (b >= p.B || n0 >= p.Nq) return;
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int l15 = lane & 15, kk = lane >> 4;
  const int Tin = p.Tin, Mp = p.Mp, CK = p.CK, lck = 31 - __clz(CK);
  const float alpha = p.act ? p.alpha_val : 1.0f;
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin, (unsigned)p.Cin * (unsigned)Tin * 4u);
  const u32x4 rw = direct_desc(p.w, (unsigned)p.Cin * (unsigned)R * (unsigned)Mp * 4u);
  const int avo = (kk * Mp + m0 + 4 * l15) * 4;
  const int c0 = n0 + TN * l15;  // this lane's first output column
  const int bvo = c0 < p.Nq ? (kk * Tin + c0 * R) * 4 : (int)0x80000000;

  const int NG = p.Cin >> 2;  // ring slots = groups of 4 channels (launcher: a multiple of D, CK % 4 == 0)
  f32x4 a4[D][R], b4[D][R];
#pragma unroll
  for (int d0 = 0; d0 < D; d0++)
#pragma unroll
    for (int k = 0; k < R; k++) { a4[d0][k] = f32x4{0.f, 0.f, 0.f, 0.f}; b4[d0][k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4acc acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

#define OU_ISSUE(g_, d)                                                                                                  \
  {                                                                                                                      \
    const int c4 = (g_) * 4;                                                                                             \
    const int wrow = ((c4 >> lck) * R) * CK + (c4 & (CK - 1)); /* packed row of (channel 4 J, tap 0) */                  \
    const int xso = c4 * Tin * 4;                                                                                        \
    _Pragma("unroll") for (int k = 0; k < R; k++)                                                                        \
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(a4[d][k]) : "v"(avo), "s"(rw), "s"((wrow + k * CK) * Mp * 4)); \
    _Pragma("unroll") for (int k = 0; k < R; k++)                                                                        \
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(b4[d][k]) : "v"(bvo), "s"(rx), "s"(xso), "n"(16 * k)); \
  }
#define OU_MMA(d, out)                                                                                                   \
  {                                                                                                                      \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPS));                                                              \
    _Pragma("unroll") for (int k = 0; k < R; k++) { asm volatile("" : "+v"(a4[d][k])); asm volatile("" : "+v"(b4[d][k])); } \
    float X[4 * R];                                                                                                      \
    _Pragma("unroll") for (int k = 0; k < R; k++) {                                                                      \
      X[4 * k + 0] = b4[d][k].x; X[4 * k + 1] = b4[d][k].y; X[4 * k + 2] = b4[d][k].z; X[4 * k + 3] = b4[d][k].w;        \
    }                                                                                                                    \
    _Pragma("unroll") for (int e = 0; e < 4 * R; e++) X[e] = X[e] >= 0.f ? X[e] : alpha * X[e];                          \
    _Pragma("unroll") for (int k = 0; k < R; k++)                                                                        \
      _Pragma("unroll") for (int i = 0; i < TM; i++) {                                                                   \
        const float av = i == 0 ? a4[d][k].x : (i == 1 ? a4[d][k].y : (i == 2 ? a4[d][k].z : a4[d][k].w));              \
        _Pragma("unroll") for (int j = 0; j < TN; j++)                                                                   \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, X[j * R + k], acc[i][j], 0, 0, 0);                        \
      }                                                                                                                  \
  }
  if constexpr (D == 4) {
    OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
    const int NR = NG / 4;
    for (int r = 0; r + 1 < NR; r++) {
      const int g = r * 4;
      OU_MMA(0, 3); OU_ISSUE(g + 4, 0);
      OU_MMA(1, 3); OU_ISSUE(g + 5, 1);
      OU_MMA(2, 3); OU_ISSUE(g + 6, 2);
      OU_MMA(3, 3); OU_ISSUE(g + 7, 3);
    }
    OU_MMA(0, 3); OU_MMA(1, 2); OU_MMA(2, 1); OU_MMA(3, 0);
  } else {
    OU_ISSUE(0, 0); OU_ISSUE(1, 1);
    const int NR = NG / 2;
    for (int r = 0; r + 1 < NR; r++) {
      const int g = r * 2;
      OU_MMA(0, 1); OU_ISSUE(g + 2, 0);
      OU_MMA(1, 1); OU_ISSUE(g + 3, 1);
    }
    OU_MMA(0, 1); OU_MMA(1, 0);
  }
#undef OU_ISSUE
#undef OU_MMA

  // ---- epilogue: value (i, j, r) = row m0 + 4 (4 kk + r) + i, column c0 + j
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
  const int up = p.up;
  int ncol = p.Nq - c0;
  if (ncol > 4) ncol = 4;
  if (ncol <= 0) return;
  const int rlen = ragged_len(p.lens, b);  // ragged batch: valid output samples of this row
  auto finish = [&](f32x4 v, int co, size_t idx, bool full) {  // 4 consecutive output samples of channel co at idx
    if (p.in_scale) v *= insc;
    v += p.bias[co];
    if (p.add) {
      f32x4 ad;
      if (full) ad = *reinterpret_cast<const f32x4u*>(p.add + idx);
      else { ad = f32x4{0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 4; e++) if (e < ncol) ad[e] = p.add[idx + e]; }
      v = (v + ad) * p.add_scale;
    }
    if (filmb) v = filmb[co] * v + filmb[p.Cout + co];
    if (p.res) {
      f32x4 rs;
      if (full) rs = *reinterpret_cast<const f32x4u*>(p.res + idx);
      else { rs = f32x4{0.f, 0.f, 0.f, 0.f}; for (int e = 0; e < 4; e++) if (e < ncol) rs[e] = p.res[idx + e]; }
      v = (v + rs) * p.res_scale;
    }
    if (p.lens) v = ragged_mask4(v, (int)(idx - (ybase + (size_t)co * p.Tout)), rlen);
    if (full) *reinterpret_cast<f32x4u*>(p.y + idx) = v;
    else for (int e = 0; e < 4; e++) if (e < ncol) p.y[idx + e] = v[e];
  };
  const bool al4 = true;  // (16-byte accesses at dword alignment)
  if (up == 1) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int row = m0 + 4 * (4 * kk + r) + i;
        if (row >= p.M) continue;
        finish(f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]}, row, ybase + (size_t)row * p.Tout + c0,
               al4 && ncol == 4);
      }
  } else if (up == 4 && ncol == 4) {  // channel co: phases i = 0..3 of frames c0 + j -> 16 consecutive samples
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int co = (m0 >> 2) + 4 * kk + r;
      if (co * 4 >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; j++)
        finish(f32x4{acc[0][j][r], acc[1][j][r], acc[2][j][r], acc[3][j][r]}, co,
               ybase + (size_t)co * p.Tout + (size_t)(c0 + j) * 4, true);
    }
  } else if (up == 2 && ncol == 4) {  // rows 4 (4 kk + r) + {0, 1} = channel a (phases 0, 1), + {2, 3} = channel a + 1
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        const int co = (m0 >> 1) + 2 * (4 * kk + r) + h2;
        if (co * 2 >= p.M) continue;
        const size_t idx = ybase + (size_t)co * p.Tout + (size_t)c0 * 2;
        finish(f32x4{acc[2 * h2][0][r], acc[2 * h2 + 1][0][r], acc[2 * h2][1][r], acc[2 * h2 + 1][1][r]}, co, idx, true);
        finish(f32x4{acc[2 * h2][2][r], acc[2 * h2 + 1][2][r], acc[2 * h2][3][r], acc[2 * h2 + 1][3][r]}, co, idx + 4, true);
      }
  } else if (up == 8 && ncol == 4) {  // rows 4 (4 kk + r) + i: channel 2 kk + (r >> 1), phase 4 (r & 1) + i
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int co = (m0 >> 3) + 2 * kk + (r >> 1);
      if (co * 8 >= p.M) continue;
#pragma unroll
      for (int j = 0; j < TN; j++)
        finish(f32x4{acc[0][j][r], acc[1][j][r], acc[2][j][r], acc[3][j][r]}, co,
               ybase + (size_t)co * p.Tout + (size_t)(c0 + j) * 8 + 4 * (r & 1), true);
    }
  } else {  // any rate: sample by sample
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < TM; i++) {
        const int m = m0 + 4 * (4 * kk + r) + i;
        if (m >= p.M) continue;
        const int co = (int)__umulhi((unsigned)m, p.magic_up), ph = m - co * up;
#pragma unroll
        for (int j = 0; j < TN; j++) {
          if (j >= ncol) continue;
          const size_t idx = ybase + (size_t)co * p.Tout + (size_t)(c0 + j) * up + ph;
          float v = acc[i][j][r];
          if (p.in_scale) v *= insc;
          v += p.bias[co];
          if (p.add) v = (v + p.add[idx]) * p.add_scale;
          if (filmb) v = filmb[co] * v + filmb[p.Cout + co];
          if (p.res) v = (v + p.res[idx]) * p.res_scale;
          if ((c0 + j) * up + ph >= rlen) v = 0.f;
          p.y[idx] = v;
        }
      }
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

struct Direct3Cfg {
  int KW, TM, D;
  void (*kern)(ConvArgs);      // no tensor operand in the epilogue (or rows that are not 16-byte multiples)
  void (*kern_pre)(ConvArgs);  // residual / cond add prefetched into LDS under the last ring slots
};
#define OU_D3(KW, TM, D) {KW, TM, D, conv_direct3_kernel<KW, TM, D, false>, conv_direct3_kernel<KW, TM, D, true>}
static const Direct3Cfg kDirect3Cfgs[] = {
    // ring depth 4 only: the depth-2 instantiations come out of the compiler with MORE registers (240-256, spills)
    OU_D3(3, 2, 4), OU_D3(3, 3, 4), OU_D3(3, 4, 4), OU_D3(5, 2, 4), OU_D3(5, 3, 4), OU_D3(5, 4, 4),
};
struct Direct3wCfg {
  int KW, TM;
  void (*kern)(ConvArgs);
  void (*kern_pre)(ConvArgs);
};
#define OU_D3W(KW, TM) {KW, TM, conv_direct3w_kernel<KW, TM, false>, conv_direct3w_kernel<KW, TM, true>}
// (k3 at 64 rows and k5 at 48 rows do not fit the register file: 256 VGPRs + scratch)
static const Direct3wCfg kDirect3wCfgs[] = {OU_D3W(3, 1), OU_D3W(3, 2), OU_D3W(3, 3), OU_D3W(5, 1), OU_D3W(5, 2)};
hipError_t init_direct3_kernels() {
  for (const Direct3Cfg& c : kDirect3Cfgs) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.kern_pre), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
  }
  for (const Direct3wCfg& c : kDirect3wCfgs) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.kern_pre), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
// rows per wave tile (in units of 16) for a layer with M output channels: exact tiling where 16-row granularity allows it
static int direct3_tm(int M) {
  if (M <= 32) return 2;
  if (M % 64 != 0 && M % 48 == 0) return 3;  // 48, 96, 144: no padding rows
  return 4;
}
// wave tiles per SIMD the throughput kernel would get for a stride-1 k3 / k5 layer of M rows (what launch_conv's choice and
// the ConvBlock fusion plan are based on)
double direct3_tiles_per_simd(int M, int Nq, int B, int num_cu) {
  int tm = direct3_tm(M);
  double t = (double)((M + 16 * tm - 1) / (16 * tm)) * ((Nq + 63) / 64) * B / (4.0 * num_cu);
  if (tm > 2 && M % 32 == 0 && t < 3.0) t = (double)((M + 31) / 32) * ((Nq + 63) / 64) * B / (4.0 * num_cu);
  return t;
}
struct Direct3sCfg {
  int R;
  void (*kern)(ConvArgs);
};
static const Direct3sCfg kDirect3sCfgs[] = {
    {1, conv_direct3s_kernel<1, 4>}, {2, conv_direct3s_kernel<2, 4>}, {3, conv_direct3s_kernel<3, 2>},
    {4, conv_direct3s_kernel<4, 2>}, {5, conv_direct3s_kernel<5, 2>},
};
// Launches the throughput kernel when the layer fits it AND supplies enough wave tiles to fill the machine without
// splitting K; hipErrorInvalidConfiguration = "use the other kernels".
static hipError_t launch_conv_direct3s(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out, double tile_min) {
  // 1x1 / phase GEMMs (KW = 1, any up) and k = s = R rate-change convs on whole frames
  const int R = a.stride > 1 ? a.stride : 1;
  if (a.KW != R || a.pad != 0 || (a.stride > 1 && (a.up != 1 || a.Tin != a.Nq * R)) || a.Cin % 16 || a.CK % 4 || a.fir ||
      (a.in_scale != nullptr && a.act))
    return hipErrorInvalidConfiguration;
  if ((long)a.Cin * a.Tin * 4 >= (1L << 31) || (long)a.Cin * a.KW * a.Mp * 4 >= (1L << 31)) return hipErrorInvalidConfiguration;
  void (*kern)(ConvArgs) = nullptr;
  for (const Direct3sCfg& c : kDirect3sCfgs)
    if (c.R == R) { kern = c.kern; break; }
  if (!kern) return hipErrorInvalidConfiguration;
  const long gy = (a.M + 63) / 64, ct = (a.Nq + 63) / 64;
  const double per_simd = (double)gy * ct * a.B / (4.0 * num_cu);
  if (a.force_cfg < 200 && per_simd < tile_min) return hipErrorInvalidConfiguration;
  if (a.out_act) return hipErrorNotSupported;  // (this kernel would take the layer, but has no activating epilogue)
  ConvArgs aa = a;
  aa.grid_m = (int)gy;
  aa.magic_up = a.up == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)a.up) + 1u;
  const long chunks = (ct + 3) / 4, total8 = (chunks * a.B + 7) / 8 * 8;
  aa.grid_n = (int)chunks;
  if (cfg_out) *cfg_out = 260 + R;
  hipLaunchKernelGGL(kern, dim3((unsigned)(total8 * gy)), dim3(256), 0, stream, aa);
  return hipGetLastError();
}
hipError_t launch_conv_direct3(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  const double tile_min_s = a.tile_min >= 0 ? a.tile_min : 1.2;
  if (a.KW == 1 || a.stride > 1) {
    if (tile_min_s > 0 && a.Nq < 1024 && a.force_cfg < 200) return hipErrorInvalidConfiguration;

    // (with the up-path FIR requested as a fused epilogue: refuse, so that the caller runs conv + FIR pass -- unless the layer
    // is too small for this kernel anyway, then the split-K kernel with its fused FIR gets its chance)
    if (a.fir) {
      ConvArgs probe = a;
      probe.fir = nullptr;
      const int R = 1;
      const double per_simd = (double)((a.M + 63) / 64) * ((a.Nq + 63) / 64) * a.B / (4.0 * num_cu);
      if (a.KW == R && a.stride == 1 && a.pad == 0 && a.Cin % 16 == 0 && a.CK % 4 == 0 && per_simd >= tile_min_s &&
          !(a.in_scale != nullptr && a.act) && a.force_cfg < 0)
        return hipErrorNotSupported;
      return hipErrorInvalidConfiguration;
    }
    return launch_conv_direct3s(a, num_cu, stream, cfg_out, tile_min_s);
  }
  if (!a.wd || a.stride != 1 || a.up != 1 || (a.KW != 3 && a.KW != 5) || a.pad != (a.KW - 1) / 2 || a.fir || a.Cin % 16 ||
      (a.in_scale != nullptr && a.act))
    return hipErrorInvalidConfiguration;
  if ((long)a.Cin * a.Tin * 4 >= (1L << 31) || (long)a.Cin * a.Mp * 8 * 4 >= (1L << 31)) return hipErrorInvalidConfiguration;
  // wave tiles per SIMD below which the split-K kernels are ahead (measured, PP16 / PP24 at B = 1 .. 16: break-even at
  // ~1 tile per SIMD, +8 .. +60 % from 1.5 up, 2-3x slower at 0.25; OU_TILE_MIN: tuning / tests, 0 = wherever it fits)
  const double tile_min = a.tile_min >= 0 ? a.tile_min : 1.2;
  // short signals (the T / 160 level: 401 frames = 6.3 column tiles per element) waste the last tile and supply few
  // chunks; the split-K kernels keep them whatever the batch (B = 8: 54 vs 107 us on the latent k3 convs) -- except the k5
  // layers in their minimal-filtering form from ~0.85 wave tiles per SIMD (PP16 B = 8: 76.5 vs 89.1 us; k3 50.5 vs 54.4 there
  // but 116 vs 109 us on PP24's 768 channels: k5 only)
  const bool wino_ok = a.wino && a.direct >= 5 && a.wu && a.M % 16 == 0;
  const bool short_k5 = wino_ok && a.KW == 5 && a.M % 32 == 0 &&
                        (double)(a.M / 32) * ((a.Nq + 63) / 64) * a.B / (4.0 * num_cu) >= 0.85;
  if (tile_min > 0 && a.Nq < 1024 && a.force_cfg < 200 && !short_k5) return hipErrorInvalidConfiguration;
  int tm = direct3_tm(a.M);
  const long ct = (a.Nq + 63) / 64;
  // 32-row tiles where the preferred ones leave fewer than ~3 wave tiles per SIMD (and M tiles by 32): twice the waves, half the
  // registers (4 waves per SIMD instead of 2), for 4 instead of 6 loads per 24 instead of 48 MFMAs.  Measured (tile_sweep):
  // PP24 C = 384 at B = 8 (2.4 -> 4.8 tiles per SIMD) 355 / 222 -> 305 / 190 us, PP16 C = 64 at B = 8 (3.9 -> 7.8) 109 / 68 -> 104 / 64;
  // even at 5.9 tiles per SIMD (PP24 C = 192) the two are equal.
  if (tm > 2 && a.M % 32 == 0 && (double)((a.M + 16 * tm - 1) / (16 * tm)) * ct * a.B / (4.0 * num_cu) < 3.0) tm = 2;
  // minimal filtering (conv_direct3w_kernel: 32-row tiles) wherever the rows tile by 32: 2/3 (k3) / 3/5 (k5) of the MFMAs
  // minimal filtering: 32-row tiles (16-row tiles where the rows only tile by 16: PP24's 48 channels, 165 vs 184 us)
  bool wino = wino_ok;
  if (wino) tm = a.M % 32 == 0 ? 2 : 1;
  if (a.force_cfg >= 200) { tm = (a.force_cfg / 10) % 10; wino = false; }
  if (a.force_cfg >= 500 && a.force_cfg < 600 && a.wu) wino = true;  // 500 + 10 TM + KW: the minimal-filtering form (tests / sweeps)
  if (tm < (wino ? 1 : 2) || tm > 4) return hipErrorInvalidConfiguration;
  const long gy = (a.M + 16 * tm - 1) / (16 * tm);
  const double per_simd = (double)gy * ct * a.B / (4.0 * num_cu);
  if (a.force_cfg < 200 && per_simd < (short_k5 && a.Nq < 1024 ? 0.85 : tile_min)) return hipErrorInvalidConfiguration;
  // 4 waves x 4 TM KB of LDS for the prefetched epilogue operand (OU_TILE_PREFETCH=0 switches it off)
  const bool prefetch = (a.res || a.add) && (a.Tout & 3) == 0 && a.tile_prefetch != 0;
  void (*kern)(ConvArgs) = nullptr;
  if (wino) {
    for (const Direct3wCfg& c : kDirect3wCfgs)
      if (c.KW == a.KW && c.TM == tm) { kern = prefetch ? c.kern_pre : c.kern; break; }
  } else {
    for (const Direct3Cfg& c : kDirect3Cfgs)
      if (c.KW == a.KW && c.TM == tm) { kern = prefetch ? c.kern_pre : c.kern; break; }
  }
  if (!kern) return hipErrorInvalidConfiguration;
  ConvArgs aa = a;
  aa.grid_m = (int)gy;
  const long chunks = (ct + 3) / 4, total8 = (chunks * a.B + 7) / 8 * 8;
  aa.grid_n = (int)chunks;
  if (cfg_out) *cfg_out = (wino ? 500 : 200) + 10 * tm + a.KW;
  const size_t smem = prefetch ? (size_t)4 * 4 * tm * 1024 : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)(total8 * gy)), dim3(256), smem, stream, aa);
  return hipGetLastError();
}


}  // namespace ou
