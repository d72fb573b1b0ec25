// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: conv_direct4_kernel -- the 1x1 convs, the transposed
// convs as phase GEMMs and the k = s = r rate-change convs at FEW output columns (batch 1 .. 4), with 16-byte operand loads.
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ou {

// ---------------------------------------------------------------------------------------------------------
// conv_direct4_kernel<R, TM, WN, WK, D>
// What it replaces: conv_direct_kernel<1, ..> / conv_direct_strided_kernel (first-generation register-direct split-K kernels:
// ONE DWORD per lane and MFMA operand, 1.5 load instructions per MFMA -- bound by the CU's vector-memory issue rate, 25 % of the
// fp32 MFMA peak at batch 1) for the layers whose reduction is plain channels (x R taps of whole frames):
//   reference ops: blocks.py:264-283 (rate-change Conv1d / ConvTranspose1d), score.py:189-194 (signal_cond_proj 1x1),
//   condition.py:33-65 (st convs after space-to-depth), the GRU input projections (score.py:83-89).
// Operand scheme = conv_direct3s_kernel's (v_mfma_f32_16x16x4_f32, four channels per MFMA):
//   A (tap k, channels 4J + kk): lane (m, kk) loads TM ADJACENT ROWS m0 + TM m .. of packed weight row (channel, tap) with one
//     16-byte (TM = 4) / 8-byte (TM = 2) load; the TM 16-row accumulator tiles are row-interleaved (tile i = rows m0 + TM m + i);
//   B: lane (n, kk) loads the 4 R consecutive samples of its four adjacent output columns (R 16-byte loads); the four column
//     tiles are column-interleaved (tile j = columns n0 + 4 n + j), column j / tap k is window element j R + k.
//   2 R load instructions per 4 TM R MFMAs (first generation: 3 R dword loads per 2 R MFMAs of twice the size).
// What is new against conv_direct3s_kernel (which needs >= 1.2 wave tiles per SIMD, i.e. batch >= 4):
//   * the reduction can be SPLIT over the WK waves of a block (slot g of wave wk = channel group wk + WK g), and the tile can
//     be 32 rows (TM = 2): a 512 x 401 layer gets 8 x 7 x WK = 448 waves instead of 56;
//   * ONE epilogue for every (up, R, WK): the accumulators go through LDS (16-byte writes, one slab per wave), are reduced over
//     the K slices on read and leave as 16-byte stores of four CONSECUTIVE output samples -- for the phase GEMMs too (up = 2,
//     4, 5, 8: sample t = q up + phase is row co up + phase, column q of the tile), where the first generation stored sample by
//     sample.  With WK = 1 every wave reads back only its own slab: no barrier.
// Summation order per output: channel groups wk, wk + WK, .. ascending inside a wave (4 channels per MFMA, taps ascending),
// then the WK partial sums in wave order -- fixed, but not the first generation's (results agree to fp32 rounding).
// ---------------------------------------------------------------------------------------------------------
typedef float f32x4acc __attribute__((ext_vector_type(4)));

template <int R, int TM, int WN, int WK, int D>
__global__ __launch_bounds__(64 * WN * WK) void conv_direct4_kernel(ConvArgs p) {
  constexpr int TN = 4, LPS = 2 * R, BM = 16 * TM, NW = WN * WK, EP = 68;
  static_assert(WN == 1 || WK == 1, "a block is either WN column tiles or WK slices of the reduction");
  static_assert(TM >= 1 && TM <= 4, "16 .. 64-row tiles");
  static_assert(D * LPS <= 60, "loads in flight must fit vmcnt");
  static_assert(D == 2 || D == 4, "ring depth");
  // A fragment of one (tap, 4 channels): TM adjacent rows per lane -- one dword / dwordx2 / dwordx3 / dwordx4 load
  typedef float f32x3 __attribute__((ext_vector_type(3)));
  typedef typename std::conditional<TM == 4, f32x4, typename std::conditional<TM == 3, f32x3,
          typename std::conditional<TM == 2, f32x2, float>::type>::type>::type avec;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = WK == 1 ? wv : 0, wk = WK == 1 ? 0 : wv;
  // Which operand an XCD's L2 owns (block L runs on XCD L % 8, private 4 MB L2):
  //   xcd_map 2 (activations >= weights): blocks L, L + 8, .. walk the row groups of one (batch element, column chunk) pair --
  //     a chunk's activations are fetched into ONE L2, every XCD reads all the weights; the pairs are numbered through (a
  //     short signal has only a chunk or two per element);
  //   xcd_map 1 (weights > activations: the 401-frame levels, the K = 5120 st convs): row groups are dealt to the XCDs, the
  //     blocks of an XCD walk the column chunks of ITS row groups -- every XCD reads an eighth of the weights and all the
  //     activations (3 + 8 x 0.8 MB instead of 8 x 3 + 0.8 MB of fabric traffic for the GRU input projection).
  const int L = blockIdx.x, q8 = L >> 3;
  int rg, cidx;
  if (p.xcd_map == 1) {
    const int gm8 = (p.grid_m + 7) >> 3;
    rg = (q8 % gm8) * 8 + (L & 7);
    cidx = q8 / gm8;
  } else {
    rg = q8 % p.grid_m;
    cidx = (q8 / p.grid_m) * 8 + (L & 7);
  }
  const int b = cidx / p.grid_n, chunk = cidx - b * p.grid_n;
  const int n0 = (chunk * WN + wn) * 64, m0 = rg * BM;
  if (b >= p.B || rg >= p.grid_m) return;  // (block-uniform)
  if (WK == 1 && n0 >= p.Nq) return;     // (wave-uniform; no barrier in the WK = 1 form)
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int l15 = lane & 15, kk = lane >> 4;
  const int Tin = p.Tin, Mp = p.Mp, CK = p.CK, lck = 31 - __clz(CK);
  const float alpha = p.act ? p.alpha_val : 1.0f;
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin, (unsigned)p.Cin * (unsigned)Tin * 4u);
  const u32x4 rw = direct_desc(p.w, (unsigned)p.Cin * (unsigned)R * (unsigned)Mp * 4u);
  const int avo = (kk * Mp + m0 + TM * l15) * 4;
  const int c0 = n0 + TN * l15;  // this lane's first output column
  const int bvo = c0 < p.Nq ? (kk * Tin + c0 * R) * 4 : (int)0x80000000;  // past the buffer: reads as 0

  const int NS = (p.Cin >> 2) / WK;  // ring slots of this wave = groups of 4 channels (launcher: a multiple of D)
  avec a4[D][R];
  f32x4 b4[D][R];
#pragma unroll
  for (int d0 = 0; d0 < D; d0++)
#pragma unroll
    for (int k = 0; k < R; k++) {
      if constexpr (TM == 4) a4[d0][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      else if constexpr (TM == 3) a4[d0][k] = f32x3{0.f, 0.f, 0.f};
      else if constexpr (TM == 2) a4[d0][k] = f32x2{0.f, 0.f};
      else a4[d0][k] = 0.f;
      b4[d0][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  f32x4acc acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};

  // Loads and counted waits are inline asm (see conv_direct_kernel: left to the compiler a register ring carried around a loop
  // resolves to vmcnt(0) at the loop header); ring registers are tied "+v" operands, values are used only after the empty "+v"
  // asm that follows the wait; tools/check_isa.py verifies the generated code.
#define OU_ISSUE(g_, d)                                                                                                  \
  {                                                                                                                      \
    const int c4 = (wk + WK * (g_)) * 4;                                                                                 \
    const int wrow = ((c4 >> lck) * R) * CK + (c4 & (CK - 1)); /* packed row of (channel 4 J, tap 0) */                  \
    const int xso = c4 * Tin * 4;                                                                                        \
    _Pragma("unroll") for (int k = 0; k < R; k++) {                                                                      \
      if constexpr (TM == 4)                                                                                             \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(a4[d][k]) : "v"(avo), "s"(rw), "s"((wrow + k * CK) * Mp * 4)); \
      else if constexpr (TM == 3)                                                                                        \
        asm volatile("buffer_load_dwordx3 %0, %1, %2, %3 offen" : "+v"(a4[d][k]) : "v"(avo), "s"(rw), "s"((wrow + k * CK) * Mp * 4)); \
      else if constexpr (TM == 2)                                                                                        \
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(a4[d][k]) : "v"(avo), "s"(rw), "s"((wrow + k * CK) * Mp * 4)); \
      else                                                                                                               \
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(a4[d][k]) : "v"(avo), "s"(rw), "s"((wrow + k * CK) * Mp * 4)); \
    }                                                                                                                    \
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
        float av;                                                                                                        \
        if constexpr (TM == 1) av = a4[d][k]; else av = a4[d][k][i];                                                     \
        _Pragma("unroll") for (int j = 0; j < TN; j++)                                                                   \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, X[j * R + k], acc[i][j], 0, 0, 0);                        \
      }                                                                                                                  \
  }
  const bool ts_on = p.tstamps != nullptr;
  long long tc0 = 0, tc1 = 0, tc2 = 0, tr0 = 0;
  if (ts_on) { tr0 = (long long)__builtin_amdgcn_s_memrealtime(); tc0 = __builtin_readcyclecounter(); }
  if constexpr (D == 4) {
    OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
    if (ts_on) tc1 = __builtin_readcyclecounter();
    const int NR = NS / 4;
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
    if (ts_on) tc1 = __builtin_readcyclecounter();
    const int NR = NS / 2;
    for (int r = 0; r + 1 < NR; r++) {
      const int g = r * 2;
      OU_MMA(0, 1); OU_ISSUE(g + 2, 0);
      OU_MMA(1, 1); OU_ISSUE(g + 3, 1);
    }
    OU_MMA(0, 1); OU_MMA(1, 0);
  }
  if (ts_on) tc2 = __builtin_readcyclecounter();
#undef OU_ISSUE
#undef OU_MMA

  // ---- epilogue.  Accumulator (i, j) of lane (n = l15, q = kk), register r = row TM (4 q + r) + i, column 4 n + j of the tile:
  // one 16-byte LDS write per (i, r) into this wave's slab [BM][EP].
  float* Ew = smem + (size_t)wv * BM * EP;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      *reinterpret_cast<f32x4*>(&Ew[(TM * (4 * kk + r) + i) * EP + 4 * l15]) =
          f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
  if constexpr (WK > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (LDS operations of one wave complete in order)

  // readers: with WK > 1 the whole block shares the tile (slab k = K slice k, summed on read); with WK = 1 every wave reads
  // back its own slab
  const int rt = WK == 1 ? lane : tid;
  constexpr int RNT = WK == 1 ? 64 : 64 * WK;
  const float* Er = WK == 1 ? Ew : smem;
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
  const int nvalid = p.Nq - n0;  // columns of this tile inside the signal (>= 1)
  const int rlen = ragged_len(p.lens, b);  // ragged batch: valid output samples of this row
  if (p.up == 1) {
    // quad e = (row, four adjacent columns): 16 consecutive lanes = 256 contiguous bytes of one output row
    constexpr int NQ = BM * 16, QPT = (NQ + RNT - 1) / RNT;
#pragma unroll
    for (int u = 0; u < QPT; u++) {
      const int e = rt + u * RNT;
      if (NQ % RNT != 0 && e >= NQ) break;
      const int row = e >> 4, cq = (e & 15) * 4;
      const int m = m0 + row;
      if (m >= p.M || cq >= nvalid) continue;
      const size_t idx = ybase + (size_t)m * p.Tout + n0 + cq;
      const bool full = cq + 4 <= nvalid;
      f32x4 ad = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
      if (full) {
        if (p.add) ad = *reinterpret_cast<const f32x4u*>(p.add + idx);
        if (p.res) rs = *reinterpret_cast<const f32x4u*>(p.res + idx);
      } else {
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if (p.add && cq + s < nvalid) ad[s] = p.add[idx + s];
          if (p.res && cq + s < nvalid) rs[s] = p.res[idx + s];
        }
      }
      f32x4 v = *reinterpret_cast<const f32x4*>(&Er[row * EP + cq]);
#pragma unroll
      for (int k = 1; k < WK; k++) v += *reinterpret_cast<const f32x4*>(&Er[(k * BM + row) * EP + cq]);
      if (p.in_scale) v *= insc;
      v += p.bias[m];
      if (p.add) v = (v + ad) * p.add_scale;
      if (filmb) v = filmb[m] * v + filmb[p.Cout + m];
      if (p.res) v = (v + rs) * p.res_scale;
      if (p.lens) v = ragged_mask4(v, n0 + cq, rlen);
      if (full) {
        *reinterpret_cast<f32x4u*>(p.y + idx) = v;
      } else {
#pragma unroll
        for (int s = 0; s < 4; s++)
          if (cq + s < nvalid) p.y[idx + s] = v[s];
      }
    }
  } else {
    // transposed conv as `up` phase GEMMs: row m = co up + phase, column q -> sample t = q up + phase.  Quad e = (channel, four
    // CONSECUTIVE samples); a channel whose phases straddle two row tiles is completed by both (disjoint samples).
    const int up = p.up;
    const int m_hi = (m0 + BM - 1 < p.M - 1) ? m0 + BM - 1 : p.M - 1;
    const int co_lo = (int)__umulhi((unsigned)m0, p.magic_up);
    const int nco = (int)__umulhi((unsigned)m_hi, p.magic_up) - co_lo + 1;
    const int qpc = 16 * up;  // quads per channel: 64 frames x up samples / 4
    const int total = nco * qpc;
    for (int e = rt; e < total; e += RNT) {
      const int cl = (int)__umulhi((unsigned)(e >> 4), p.magic_up);  // e / (16 up)
      const int tl = (e - cl * qpc) * 4;                              // first local sample of the quad, 0 .. 64 up - 4
      const int co = co_lo + cl;
      const int q0 = (int)__umulhi((unsigned)tl, p.magic_up);         // tl / up
      int ph = tl - q0 * up, q = q0;
      f32x4 v;
      bool ok[4];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int row = co * up + ph - m0;
        ok[s] = row >= 0 && row < BM && m0 + row <= m_hi && q < nvalid;
        float a = 0.f;
        if (ok[s]) {
          a = Er[row * EP + q];
#pragma unroll
          for (int k = 1; k < WK; k++) a += Er[(k * BM + row) * EP + q];
        }
        v[s] = a;
        if (++ph == up) { ph = 0; q++; }
      }
      const size_t idx = ybase + (size_t)co * p.Tout + (size_t)n0 * up + tl;
      const bool full = ok[0] && ok[1] && ok[2] && ok[3];
      if (!(ok[0] || ok[1] || ok[2] || ok[3])) continue;
      f32x4 ad = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
      if (full) {
        if (p.add) ad = *reinterpret_cast<const f32x4u*>(p.add + idx);
        if (p.res) rs = *reinterpret_cast<const f32x4u*>(p.res + idx);
      } else {
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if (p.add && ok[s]) ad[s] = p.add[idx + s];
          if (p.res && ok[s]) rs[s] = p.res[idx + s];
        }
      }
      if (p.in_scale) v *= insc;
      v += p.bias[co];
      if (p.add) v = (v + ad) * p.add_scale;
      if (filmb) v = filmb[co] * v + filmb[p.Cout + co];
      if (p.res) v = (v + rs) * p.res_scale;
      if (p.lens) v = ragged_mask4(v, n0 * up + tl, rlen);
      if (full) {
        *reinterpret_cast<f32x4u*>(p.y + idx) = v;
      } else {
#pragma unroll
        for (int s = 0; s < 4; s++)
          if (ok[s]) p.y[idx + s] = v[s];
      }
    }
  }
  if (ts_on && lane == 0) {  // tuning only (OU_TS): {start ticks, cycles: prologue, main loop, epilogue, -, -, -, end ticks}
    const long long tc3 = __builtin_readcyclecounter();
    long long* o = p.tstamps + ((size_t)blockIdx.x * NW + wv) * 8;
    o[0] = tr0; o[1] = tc1 - tc0; o[2] = tc2 - tc1; o[3] = tc3 - tc2; o[4] = 0; o[5] = 0; o[6] = 0;
    o[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// ---------------------------------------------------------------------------------------------------------
// conv_direct4w_kernel<KW, TM, WK>: the stride-1 k3 / k5 layers with FEW output columns (the 401-frame levels at batch 1) in
// minimal-filtering form, F(2, KW) (round 5; scheme and numerics: conv_direct2w_kernel).  conv_direct2w_kernel needs 64-column
// tiles of 32 rows -- 112 blocks for a 512 x 401 layer, less than half the chip; its plain form with 32-column tiles gets 208
// and runs its loop at 77 % of the matrix pipe on 2 KB of L1 traffic per three MFMAs.  Here: conv_direct3w_kernel's operands
// (v_mfma_f32_16x16x4_f32: a lane's four adjacent columns are two tile positions, A lane (m, kk) loads the KW + 1 values U_x of
// (row m0 + 16 i + m, channel 4 J + kk), B lane (n, kk) the KW + 3 samples of its window) on (16 TM) x 64 wave tiles, the
// reduction split over the WK waves of a block (slot g of wave wk = channel group wk + WK g) and conv_direct4_kernel's epilogue
// (A^T in registers, 16-byte LDS writes into a per-wave slab, K slices summed on read, 16-byte stores).  16-row tiles: 32 x 7 =
// 224 blocks for 512 x 401, one per CU; per ring slot 2 (KW + 1) MFMAs of 32 cycles on 2.5 / 3.5 KB of operands -- 2/3 (3/5) of
// the matrix-pipe time and 5/8 of the L1 traffic of the kernel it replaces.
// ---------------------------------------------------------------------------------------------------------
template <int KW, int TM, int WK>
__global__ __launch_bounds__(64 * WK) void conv_direct4w_kernel(ConvArgs p) {
  constexpr int D = 4, TN = 4, NX = KW + 1, W = KW + TN - 1, KWP = KW == 3 ? 4 : 8, PAD = (KW - 1) / 2, BM = 16 * TM, EP = 68;
  constexpr int A2 = KW == 5 ? 1 : 0;
  constexpr int LPS = TM * (1 + A2) + 2;
  static_assert(KW == 3 || KW == 5, "k3 / k5");
  static_assert(D * LPS <= 60, "loads in flight must fit vmcnt");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> tile: row groups are dealt to the XCDs (the weights are the larger operand on these levels), the blocks of an XCD
  // walk the column tiles of ITS row groups (xcd_map 1 of conv_direct4_kernel)
  const int L = blockIdx.x, q8 = L >> 3;
  const int gm8 = (p.grid_m + 7) >> 3;
  const int rg = (q8 % gm8) * 8 + (L & 7), cidx = q8 / gm8;
  const int b = cidx / p.grid_n, chunk = cidx - b * p.grid_n;
  const int n0 = chunk * 64, m0 = rg * BM;
  if (b >= p.B || rg >= p.grid_m) return;  // (block-uniform)
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int l15 = lane & 15, kk = lane >> 4;
  const int Tin = p.Tin, Mp = p.Mp;
  const float alpha = p.act ? p.alpha_val : 1.0f;
  // (the descriptor starts PAD samples in front of the tensor -- workspace memory, never the start of an allocation -- so that the
  // window of the first lane of the first tile, which begins at t = -PAD, is an ordinary in-range load whose first elements are
  // masked like those behind the end of a row: one select per window element in the two edge tiles, none elsewhere)
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin - PAD, ((unsigned)p.Cin * (unsigned)Tin + PAD) * 4u);
  const u32x4 rw = direct_desc(p.wu, (unsigned)p.Cin * (unsigned)Mp * (unsigned)KWP * 4u);
  const int avo = (kk * Mp + m0 + l15) * KWP * 4;
  const int t0 = n0 + TN * l15 - PAD;
  const int bvo = (t0 < Tin) ? (kk * Tin + t0 + PAD) * 4 : (int)0x80000000;
  const bool edge = __builtin_amdgcn_readfirstlane((n0 < PAD || n0 + 64 + KW - 1 - PAD > Tin) ? 1 : 0) != 0;
  unsigned vmask = 0;  // bit i: window element i is inside the row
#pragma unroll
  for (int i = 0; i < W; i++) vmask |= (t0 + i >= 0 && t0 + i < Tin) ? (1u << i) : 0u;

  const int NS = (p.Cin >> 2) / WK;  // ring slots of this wave (launcher: a multiple of D)
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
    const int c4 = (wk + WK * (g_)) * 4;                                                                              \
    const int aso = c4 * Mp * KWP * 4, xso = c4 * Tin * 4;                                                            \
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
#define OU_MMA(d, out)                                                                                                \
  {                                                                                                                   \
    if (ts_on) { const long long ta = __builtin_readcyclecounter();                                                  \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPS) : "memory");                                              \
      twait += (unsigned)(__builtin_readcyclecounter() - ta); }                                                                               \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPS));                                                           \
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
  // tuning only (OU_TS, tools/d4_ts.py): per-wave phase stamps
  const bool ts_on = p.tstamps != nullptr;
  long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, tc4 = 0, tr0 = 0;
  unsigned twait = 0;  // cycles inside the loop's s_waitcnt vmcnt
  if (ts_on) { tr0 = (long long)__builtin_amdgcn_s_memrealtime(); tc0 = __builtin_readcyclecounter(); }
  // four copies of the loop behind block-uniform branches: with / without the edge masks, with / without the PReLU of the operand
  // path (act = 0: the producer's epilogue stored activated values, ConvArgs::out_act)
  auto run = [&](auto edge_c, auto act_c) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_c)::value, ACT = decltype(act_c)::value;
    OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
    if (ts_on) tc1 = __builtin_readcyclecounter();
    const int NR = NS / 4;
    for (int r = 0; r + 1 < NR; r++) {
      const int g = r * 4;
      OU_MMA(0, 3); OU_ISSUE(g + 4, 0);
      OU_MMA(1, 3); OU_ISSUE(g + 5, 1);
      OU_MMA(2, 3); OU_ISSUE(g + 6, 2);
      OU_MMA(3, 3); OU_ISSUE(g + 7, 3);
    }
    OU_MMA(0, 3); OU_MMA(1, 2); OU_MMA(2, 1); OU_MMA(3, 0);
  };
  const bool act_on = p.act != 0;
  if (edge) {
    if (act_on) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
  } else {
    if (act_on) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
  }
  if (ts_on) tc2 = __builtin_readcyclecounter();
#undef OU_ISSUE
#undef OU_MMA

  // ---- epilogue: A^T -> this wave's slab [BM][EP] (lane (n = l15, q = kk), register r = row 16 i + 4 q + r, columns 4 n .. + 3)
  float* Ew = smem + (size_t)wk * BM * EP;
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
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
      *reinterpret_cast<f32x4*>(&Ew[(16 * i + 4 * kk + r) * EP + 4 * l15]) = v;
    }
  if (ts_on) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tc3 = __builtin_readcyclecounter(); }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (ts_on) tc4 = __builtin_readcyclecounter();
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
  const int nvalid = p.Nq - n0;  // columns of this tile inside the signal (>= 1)
  constexpr int RNT = 64 * WK, NQ = BM * 16, QPT = (NQ + RNT - 1) / RNT;
#pragma unroll
  for (int u = 0; u < QPT; u++) {
    const int e = tid + u * RNT;
    if (NQ % RNT != 0 && e >= NQ) break;
    const int row = e >> 4, cq = (e & 15) * 4;
    const int m = m0 + row;
    if (m >= p.M || cq >= nvalid) continue;
    const size_t idx = ybase + (size_t)m * p.Tout + n0 + cq;
    const bool full = cq + 4 <= nvalid;
    f32x4 ad = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
    if (full) {
      if (p.add) ad = *reinterpret_cast<const f32x4u*>(p.add + idx);
      if (p.res) rs = *reinterpret_cast<const f32x4u*>(p.res + idx);
    } else {
#pragma unroll
      for (int s = 0; s < 4; s++) {
        if (p.add && cq + s < nvalid) ad[s] = p.add[idx + s];
        if (p.res && cq + s < nvalid) rs[s] = p.res[idx + s];
      }
    }
    f32x4 v = *reinterpret_cast<const f32x4*>(&smem[row * EP + cq]);
#pragma unroll
    for (int k = 1; k < WK; k++) v += *reinterpret_cast<const f32x4*>(&smem[(k * BM + row) * EP + cq]);
    if (p.in_scale) v *= insc;
    v += p.bias[m];
    if (p.add) v = (v + ad) * p.add_scale;
    if (filmb) v = filmb[m] * v + filmb[p.Cout + m];
    if (p.res) v = (v + rs) * p.res_scale;
    if (p.out_act) {
#pragma unroll
      for (int s = 0; s < 4; s++) v[s] = v[s] >= 0.f ? v[s] : p.out_alpha * v[s];
    }
    if (p.lens) v = ragged_mask4(v, n0 + cq, ragged_len(p.lens, b));
    if (full) {
      *reinterpret_cast<f32x4u*>(p.y + idx) = v;
    } else {
#pragma unroll
      for (int s = 0; s < 4; s++)
        if (cq + s < nvalid) p.y[idx + s] = v[s];
    }
  }
  if (ts_on && lane == 0) {  // {start ticks, cycles: prologue, loop, A^T + slab write, barrier wait, reduce + store, -, end ticks}
    const long long tc5 = __builtin_readcyclecounter();
    long long* o = p.tstamps + ((size_t)blockIdx.x * WK + wk) * 8;
    o[0] = tr0; o[1] = tc1 - tc0; o[2] = tc2 - tc1; o[3] = tc3 - tc2; o[4] = tc4 - tc3; o[5] = tc5 - tc4; o[6] = twait;
    o[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

struct Direct4wCfg {
  int KW, TM, WK;
  void (*kern)(ConvArgs);
};
#define OU_D4W(KW, TM, WK) {KW, TM, WK, conv_direct4w_kernel<KW, TM, WK>}
static const Direct4wCfg kDirect4wCfgs[] = {OU_D4W(3, 1, 8), OU_D4W(3, 2, 8), OU_D4W(5, 1, 8), OU_D4W(5, 2, 8),
                                            OU_D4W(3, 1, 4), OU_D4W(3, 2, 4), OU_D4W(5, 1, 4), OU_D4W(5, 2, 4)};
// The k3 / k5 layers of the 401-frame levels at small batch.  Tile height for an even fill of the CUs (cost = rounds x rows, as
// direct4_pick); hipErrorInvalidConfiguration = "not a layer for it" (the caller goes on to the other kernels).
// force_cfg 600 + 10 TM + KW (+ 100: four K slices instead of eight): tuning.
hipError_t launch_conv_direct4w(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (!a.wu || a.stride != 1 || a.up != 1 || (a.KW != 3 && a.KW != 5) || a.pad != (a.KW - 1) / 2 || a.fir || a.Cin % 16 ||
      (a.in_scale != nullptr && a.act))
    return hipErrorInvalidConfiguration;
  if ((long)a.Cin * a.Tin * 4 >= (1L << 31) || (long)a.Cin * a.Mp * 8 * 4 >= (1L << 31)) return hipErrorInvalidConfiguration;
  const bool forced = a.force_cfg >= 600 && a.force_cfg < 800;
  if (!forced && !(a.wino && a.direct >= 5)) return hipErrorInvalidConfiguration;
  const int slots = a.Cin / 4;
  const long ct = (a.Nq + 63) / 64;
  int tm = 0, wk = 8;
  if (forced) {
    tm = ((a.force_cfg % 100) / 10);
    wk = a.force_cfg >= 700 ? 4 : 8;
  } else {
    // short layers only, and only where conv_direct2w_kernel's 64-column x 32-row tiles leave CUs idle
    const long b64 = (long)((a.M + 31) / 32) * ct * a.B;
    if (a.Nq >= 1024 || b64 * 10 >= (long)num_cu * 8) return hipErrorInvalidConfiguration;
    long best = 0;
    for (int t = 1; t <= 2; t++) {
      const long blocks = (long)((a.M + 16 * t - 1) / (16 * t)) * ct * a.B;
      const long cost = (blocks + num_cu - 1) / num_cu * t;
      if (!tm || cost <= best) { tm = t; best = cost; }  // (ties: the taller tile -- PP24's 768 x 401: 27.0 vs 32.4 us)
    }
  }
  if (tm < 1 || tm > 2 || slots % (wk * 4) != 0) return hipErrorInvalidConfiguration;
  const Direct4wCfg* c = nullptr;
  for (const Direct4wCfg& k : kDirect4wCfgs)
    if (k.KW == a.KW && k.TM == tm && k.WK == wk) { c = &k; break; }
  if (!c) return hipErrorInvalidConfiguration;
  ConvArgs aa = a;
  const long gy = (a.M + 16 * tm - 1) / (16 * tm);
  aa.grid_m = (int)gy;
  aa.grid_n = (int)ct;
  const long nblocks = 8L * ((gy + 7) / 8) * ct * a.B;
  if (cfg_out) *cfg_out = (wk == 4 ? 700 : 600) + 10 * tm + a.KW;
  hipLaunchKernelGGL(c->kern, dim3((unsigned)nblocks), dim3(64 * wk), (size_t)wk * 16 * tm * 68 * 4, stream, aa);
  return hipGetLastError();
}

// ---- dispatch -----------------------------------------------------------------------------------------------
struct Direct4Cfg {
  int R, TM, WN, WK, D;
  void (*kern)(ConvArgs);
};
#define OU_D4(R, TM, WN, WK, D) {R, TM, WN, WK, D, conv_direct4_kernel<R, TM, WN, WK, D>}
// ring depth: 4 slots of 2 R loads while that fits the 6-bit vmcnt and the register file (R <= 2), else 2
// (64-row tiles -- TM = 4 -- are never the rule's choice, see direct4_pick: in `make EXPERIMENTS=1` builds only, for the sweep
// tool; 16- and 48-row tiles -- TM = 1 / 3 -- exist for the short layers, where the tile height decides how evenly the blocks
// fill 256 CUs)
#ifdef OU_EXPERIMENTS
#define OU_D4_SHAPES(R, D)                                                                                    \
  OU_D4(R, 4, 4, 1, D), OU_D4(R, 2, 4, 1, D), OU_D4(R, 4, 1, 2, D), OU_D4(R, 2, 1, 2, D), OU_D4(R, 4, 1, 4, D), \
  OU_D4(R, 2, 1, 4, D), OU_D4(R, 4, 1, 8, D), OU_D4(R, 2, 1, 8, D), OU_D4(R, 1, 1, 4, D), OU_D4(R, 1, 1, 8, D), \
  OU_D4(R, 3, 1, 4, D), OU_D4(R, 3, 1, 8, D)
#else
#define OU_D4_SHAPES(R, D)                                                                            \
  OU_D4(R, 2, 4, 1, D), OU_D4(R, 2, 1, 2, D), OU_D4(R, 2, 1, 4, D), OU_D4(R, 2, 1, 8, D), OU_D4(R, 1, 1, 4, D), \
  OU_D4(R, 1, 1, 8, D), OU_D4(R, 3, 1, 4, D), OU_D4(R, 3, 1, 8, D)
#endif
static const Direct4Cfg kDirect4Cfgs[] = {
    OU_D4_SHAPES(1, 4), OU_D4_SHAPES(2, 4), OU_D4_SHAPES(3, 2), OU_D4_SHAPES(4, 2), OU_D4_SHAPES(5, 2), OU_D4_SHAPES(8, 2),
};
static size_t direct4_smem(const Direct4Cfg& c) { return (size_t)c.WN * c.WK * 16 * c.TM * 68 * 4; }
hipError_t init_direct4_kernels() {
  for (const Direct4Cfg& c : kDirect4Cfgs) {
    if (direct4_smem(c) <= 64 * 1024) continue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)direct4_smem(c));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Which (TM, WK) a layer gets.  Measured (tools/d4_sweep.py, PP16 at B = 1 / 4 / 8, profiles/r04_d4_sweep_*): 32-row tiles
// with the reduction split 4 or 8 ways are within ~5 % of the best shape on every layer -- a wave's LDS slab (16 TM x 68
// floats) caps a CU at 8 waves with 64-row tiles and at 16 with 32-row tiles, and one or two waves per SIMD run their MFMA loop
// at 55-77 % (operand latency is longer than the ring) -- as long as a wave keeps >= 64 MFMAs of its own and the launch stays
// under ~12 000 waves.  (An estimate of cycles per launch picked worse shapes than this rule on a third of the layers: block
// counts just above a multiple of the CU count, LDS-limited residency and the single-wave MFMA rate all enter.)
// SHORT layers (a few hundred frames, batch 1 - 2: the 401-frame levels): a few hundred tiles whatever the shape, so the launch
// lasts as long as its fullest CU -- the tile height (16 / 32 / 48 rows) is chosen for the block count that fills 256 CUs most
// evenly: cost = ceil(blocks / CUs) x rows; 1536 x 401 (GRU input projection): 48 rows -> 224 blocks, one per CU (32 rows: 336
// blocks = two on 80 CUs); 512 x 401: 16 rows -> 224 blocks (32 rows: 112).  Widest K split the layer admits.
static bool direct4_pick(int M, long ct_b, int slots, int R, int D, bool short_layer, int num_cu, int* tm_out, int* wk_out) {
  if (short_layer) {
    int wk = 8;
    while (wk > 4 && slots % (wk * D) != 0) wk >>= 1;
    if (slots % (wk * D)) return false;
    int best_tm = 0;
    long best_cost = 0;
    for (int tm = 1; tm <= 3; tm++) {
      const long blocks = (long)((M + 16 * tm - 1) / (16 * tm)) * ct_b;
      const long cost = (blocks + num_cu - 1) / num_cu * tm;
      if (!best_tm || cost <= best_cost) { best_tm = tm; best_cost = cost; }  // (ties: the taller tile)
    }
    *tm_out = best_tm; *wk_out = wk;
    return true;
  }
  const long tiles = (long)((M + 31) / 32) * ct_b;
  int wk = 8;
  while (wk > 1 && (slots % (wk * D) != 0 || (slots / wk) * 8 * R < 64 || tiles * wk > 12000)) wk >>= 1;
  if (slots % (wk * D)) return false;
  *tm_out = 2; *wk_out = wk;
  return true;
}

// force_cfg 300 + 10 * TM + log2(WK): tuning (ou_bench_conv / tools/direct_sweep.py)
hipError_t launch_conv_direct4(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out, bool probe) {
  const int R = a.stride > 1 ? a.stride : 1;
  if (a.KW != R || a.pad != 0 || (a.stride > 1 && (a.up != 1 || a.Tin != a.Nq * R)) || a.Cin % 16 || a.CK % 4 || a.fir ||
      (a.in_scale != nullptr && a.act))
    return hipErrorInvalidConfiguration;
  if ((long)a.Cin * a.Tin * 4 >= (1L << 31) || (long)a.Cin * a.KW * a.Mp * 4 >= (1L << 31)) return hipErrorInvalidConfiguration;
  if (a.up > 1 && (a.Cout * a.up != a.M || a.Tout != a.Nq * a.up)) return hipErrorInvalidConfiguration;
  const int slots = a.Cin / 4;
  const long ct = (a.Nq + 63) / 64;
  // The 401-frame levels (7 column tiles per element, M = 512 .. 1536: a few hundred tiles whatever the shape).  At batch 1
  // the tile height is picked for an even fill of the CUs (direct4_pick): 10.3 vs 15.6 us on the k = s = 5 rate-change conv
  // (224 blocks of 16 rows instead of 112 of 32), 11.0 vs 11.9 on the GRU input projection (224 of 48 rows), the rest equal to
  // the first generation (profiles/r04_final_d4_sweep_PP16_B1.txt).  A transposed conv whose FIR the first generation fuses
  // stays there (conv + FIR pass: 11.8 + 5.3 vs 17.1 us fused), and so do batch 2 - 3 (12 .. 20 us either way); from batch 4
  // this kernel is 3-8 % ahead on these levels too.
  const bool short_layer = a.Nq < 1024 && a.B < 4;
  if (a.force_cfg < 300 && a.d4_force == 0 && short_layer && (!a.d4_short || a.B > 1 || (probe && a.up > 1)))
    return hipErrorInvalidConfiguration;
  const Direct4Cfg* best = nullptr;
  auto code = [](const Direct4Cfg& c) { return 10 * c.TM + (c.WK == 1 ? 0 : c.WK == 2 ? 1 : c.WK == 4 ? 2 : 3); };
  const int D = R <= 2 ? 4 : 2;
  int want = a.force_cfg >= 300 ? a.force_cfg - 300 : a.d4_force;
  for (int pass = 0; pass < 2 && !best; pass++) {
    // pass 0: the shape asked for (force_cfg / OU_D4_FORCE) where the layer admits it; pass 1: the rule
    if (pass == 1) {
      if (a.force_cfg >= 300) break;
      int tm = 0, wk = 0;
      if (!direct4_pick(a.M, ct * a.B, slots, R, D, short_layer, num_cu, &tm, &wk)) break;
      want = 10 * tm + (wk == 1 ? 0 : wk == 2 ? 1 : wk == 4 ? 2 : 3);
    } else if (want == 0) {
      continue;
    }
    for (const Direct4Cfg& c : kDirect4Cfgs)
      if (c.R == R && slots % (c.WK * c.D) == 0 && code(c) == want) { best = &c; break; }
  }
  if (!best) return hipErrorInvalidConfiguration;
  if (a.out_act) return hipErrorNotSupported;  // (this family would take the layer, but has no activating epilogue)
  if (probe) return hipSuccess;
  const Direct4Cfg& c = *best;
  ConvArgs aa = a;
  const long gy = (a.M + 16 * c.TM - 1) / (16 * c.TM);
  aa.grid_m = (int)gy;
  aa.magic_up = a.up == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)a.up) + 1u;
  const long chunks = (ct + c.WN - 1) / c.WN;
  aa.grid_n = (int)chunks;
  // which operand an XCD's L2 owns (see the kernel): the larger one
  const double xb = (double)a.Cin * a.Tin * a.B, wb = (double)a.M * a.Cin * a.KW;
  aa.xcd_map = wb > xb ? 1 : 2;
  if (a.force_xcd_map == 1 || a.force_xcd_map == 2) aa.xcd_map = a.force_xcd_map;
  const long nblocks = aa.xcd_map == 1 ? 8L * ((gy + 7) / 8) * chunks * a.B : (chunks * a.B + 7) / 8 * 8 * gy;
  if (cfg_out) *cfg_out = 300 + code(c);
  hipLaunchKernelGGL(c.kern, dim3((unsigned)nblocks), dim3(64 * c.WN * c.WK), direct4_smem(c), stream, aa);
  return hipGetLastError();
}

}  // namespace ou
