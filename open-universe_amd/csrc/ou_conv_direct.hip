// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: register-direct conv kernels (split-K direct / direct2 / strided, no-split-K direct3 / direct3s, fused deep ConvBlock)
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ou {

// =========================================================================================================
// Split-K Conv1d with register-direct operands ("direct" kernel) -- the deep levels (T <= ~8000, K in the hundreds to
// thousands), stride 1, any tap count, transposed convs as phase GEMMs.
//   In the 8-wave split-K configurations of conv_mfma_kernel every wave consumes its own K slice of both operands:
//   nothing staged in LDS is ever shared between waves, LDS is only an asynchronous landing buffer -- paid for with a
//   DMA -> wait -> ds_read -> wait -> MFMA chain per pipeline stage, ~1.6 LDS reads and a dozen scalar instructions per
//   MFMA, and a 3 k-cycle prologue of index arithmetic.  Here the MFMA operands are loaded from L2 / L1 straight into
//   the registers the MFMA reads:
//     A fragment (tap, channel pair I): lane (m, half) <- w[row(2I + half, tap)][m0 + m]      2 x 128 B, streamed once
//     B fragment (tap, pair I, tile j): lane (n, half) <- x[2I + half][n0 + 32 j + n + tap - pad]
//                                       2 x 128 B; the KW shifted reads of a row hit the same L1 lines
//   as a 4-deep ring of register groups (one group = GP channel pairs x KW taps): the loads of group g + 4 are issued
//   right after the MFMAs of group g, so ~40-60 loads are in flight per wave at any time.  Loads and their counted
//   s_waitcnt vmcnt(N) are inline asm (see conv_direct_kernel; tools/check_isa.py verifies the generated code).  Zero
//   padding = per-lane offsets past the buffer bounds (computed once per block).  No LDS, no barrier and ~1 scalar
//   instruction per load in the main loop; LDS only for the cross-wave reduction of the epilogue (bias, cond add, FiLM,
//   residual, optionally the up-path FIR).
//   Same K order per output element as conv_mfma_kernel's split-K configs (pairs kw, kw + 8, ... tap-inner vs tap-outer
//   differs) -- results agree to fp32 rounding, not bit-wise.
//   Family: conv_direct_kernel (this scheme; now the 1x1 / phase-GEMM layers), conv_direct2_kernel (k3 / k5: one 16-byte
//   load per operand feeds all taps), conv_direct_strided_kernel (rate-change convs).
// =========================================================================================================

// Fused epilogue of the direct kernels: accumulators of the 8 K-slice waves -> LDS -> reduced on read -> bias, cond
// add, FiLM, residual -> store.
//   up == 1: each thread owns four consecutive samples of one output row (16-byte accesses when rows are 16-byte
//            multiples, scalar otherwise: the deep levels have T = 401, 2005);
//   up  > 1: transposed conv as `up` phase GEMMs (row m = co*up + ph -> sample t = q*up + ph): element e of the tile's
//            (co, t) range, consecutive threads = consecutive samples.
template <int TN, bool IL = false>  // IL: accumulator j holds columns TN n + j (conv_direct2_kernel), else 32 j + n
struct DirectEpilogue {
  static constexpr int WK = 8, NT = 512, BM = 32, BN = 32 * TN, EP = BN + 4, C4 = BN / 4;

  // Everything -- index arithmetic and the loads of bias / cond / FiLM / residual -- happens AFTER the main loop.
  // Prefetching these operands before the ring (tried: inline-asm loads issued first, consumed here) hides one memory
  // latency per launch but keeps 11-36 more registers live across the main loop: the 64-column kernels went from
  // 97-125 to 136-165 VGPRs, i.e. from two resident workgroups per CU to one, and the 504-block latent layers got 25-30 %
  // slower.  Occupancy wins.
  // Up path with its anti-alias FIR (blocks.py:217-225): y = FIR_{2R+1}(u) + bias, R = up, u = convT output.  The tile
  // holds MB = (32 / R) * R rows = whole output channels (all R phases) and BN frames of which the outer two are halo:
  // an output sample needs u up to R samples = one frame to either side.  Same summation order as the separate
  // launch_fir pass (8 K slices in order, taps in order), so results are bit-identical to it.
  template <int R>
  static __device__ __forceinline__ void fir_up(const ConvArgs& p, float* Es, int tid, int m0, int n0, size_t ybase,
                                                float insc) {
    constexpr int LBN = (BN == 64) ? 6 : 5, MB = (32 / R) * R, NTAP = 2 * R + 1;
    constexpr int SPAN = (BN - 2) * R;             // output samples per channel of this tile
    constexpr int TOTAL = (MB / R) * SPAN, EPT = (TOTAL + NT - 1) / NT;
    const int rlen = ragged_len(p.lens, (int)(ybase / ((size_t)p.Cout * p.Tout)));  // ragged batch: this row's valid samples
    float f[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; j++) f[j] = p.fir[j];
    // the residual / bias operands of this thread's outputs first: their latency overlaps the reduction below
    float rs[EPT], bi[EPT];
    size_t idx[EPT];
    int tl_[EPT], cl_[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      const int e = tid + k * NT;
      const int cl = e / SPAN, tl = e - cl * SPAN + R;  // local channel, local sample (frames 1 .. BN - 2)
      const int co = m0 / R + cl;
      const long t = (long)n0 * R + tl;
      const bool on = e < TOTAL && co < p.Cout && t >= 0 && t < p.Tout;
      cl_[k] = on ? cl : -1; tl_[k] = tl;
      idx[k] = on ? ybase + (size_t)co * p.Tout + (size_t)t : 0;
      bi[k] = on ? p.bias[co] : 0.f;
      rs[k] = (on && p.res) ? p.res[idx[k]] : 0.f;
    }
    for (int e = tid; e < BM * BN; e += NT) {  // reduce the K slices in place; zero outside the signal ('same' padding)
      const int row = e >> LBN, q = e & (BN - 1);
      float v = Es[row * EP + q];
#pragma unroll
      for (int k = 1; k < WK; k++) v += Es[(k * BM + row) * EP + q];
      if (p.in_scale) v *= insc;
      const int fr = n0 + q;
      if (row >= MB || m0 + row >= p.M || fr < 0 || fr >= p.Nq) v = 0.f;
      Es[row * EP + q] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      if (cl_[k] < 0) continue;
      const int tau = tl_[k] - R;  // >= 0
      int fq = tau / R, ph = tau - fq * R;
      const float* zrow = Es + (cl_[k] * R) * EP;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < NTAP; j++) {
        acc = fmaf(f[j], zrow[ph * EP + fq], acc);
        if (++ph == R) { ph = 0; fq++; }
      }
      acc += bi[k];
      if (p.res) acc = (acc + rs[k]) * p.res_scale;
      if (p.lens && (long)n0 * R + tl_[k] >= rlen) acc = 0.f;
      p.y[idx[k]] = acc;
    }
  }

  // [c_lo, c_hi): output columns this tile may STORE (plain up == 1 path only; the fused ConvBlock kernel computes halo
  // columns that belong to a neighbouring tile group)
  static __device__ __forceinline__ void run(const ConvArgs& p, const floatx16 (&acc)[TN], float* Es, int tid, int kw,
                                             int b, int m0, int n0, int c_lo = 0, int c_hi = 0x7fffffff) {
    const int lane = tid & 63, lhalf = lane >> 5, l31 = lane & 31;
    const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
    int m_hi = m0 + BM - 1;
    if (m_hi > p.M - 1) m_hi = p.M - 1;
    const size_t ybase = (size_t)b * p.Cout * p.Tout;
    const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
    // up == 1: this thread's output quad and its global operands -- the loads go out before the accumulators are
    // staged, so their latency overlaps the LDS traffic and the barrier
    const bool plain = p.up == 1 && !p.fir;
    const int eq = (tid % C4) * 4, er = tid / C4;
    const bool e_on = plain && er < BM && m0 + er <= m_hi && n0 + eq < p.Nq && n0 + eq + 4 > c_lo && n0 + eq < c_hi;
    const int m = m0 + er;
    const size_t eidx = ybase + (size_t)m * p.Tout + n0 + eq;
    int e_n = p.Nq - (n0 + eq);
    if (e_n > 4) e_n = 4;
    if (e_n > c_hi - (n0 + eq)) e_n = c_hi - (n0 + eq);
    const int e_0 = c_lo - (n0 + eq) > 0 ? c_lo - (n0 + eq) : 0;  // first element of the quad inside the store range
    // 16-byte accesses wherever the whole quad is stored (dwordx4 at dword alignment: the 401- / 2005-frame levels too)
    const bool vec4 = e_0 == 0 && e_n == 4;
    f32x4 ad = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
    float bi = 0.f, ga = 1.f, be = 0.f;
    if (e_on) {
      if (vec4) {
        if (p.add) ad = *reinterpret_cast<const f32x4u*>(p.add + eidx);
        if (p.res) rs = *reinterpret_cast<const f32x4u*>(p.res + eidx);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (p.add && j >= e_0 && j < e_n) ad[j] = p.add[eidx + j];
          if (p.res && j >= e_0 && j < e_n) rs[j] = p.res[eidx + j];
        }
      }
      bi = p.bias[m];
      if (filmb) { ga = filmb[m]; be = filmb[p.Cout + m]; }
    }
    if constexpr (IL && TN == 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        *reinterpret_cast<f32x2*>(&Es[(kw * BM + row) * EP + 2 * l31]) = f32x2{acc[0][r], acc[1][r]};
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
          Es[(kw * BM + row) * EP + 32 * j + l31] = acc[j][r];
        }
    }
    // LDS-only hand-over: wait for the ds_writes, not for the global loads above (__syncthreads would)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (p.fir) {
      switch (p.up) {
        case 2: fir_up<2>(p, Es, tid, m0, n0, ybase, insc); break;
        case 3: fir_up<3>(p, Es, tid, m0, n0, ybase, insc); break;
        case 4: fir_up<4>(p, Es, tid, m0, n0, ybase, insc); break;
        case 5: fir_up<5>(p, Es, tid, m0, n0, ybase, insc); break;
        default: fir_up<8>(p, Es, tid, m0, n0, ybase, insc); break;
      }
      return;
    }
    if (p.up == 1) {
      if (!e_on) return;
      f32x4 v = *reinterpret_cast<const f32x4*>(&Es[er * EP + eq]);
#pragma unroll
      for (int kk = 1; kk < WK; kk++) v += *reinterpret_cast<const f32x4*>(&Es[(kk * BM + er) * EP + eq]);
      if (p.in_scale) v *= insc;
      v += bi;
      if (p.add) v = (v + ad) * p.add_scale;
      if (filmb) v = ga * v + be;
      if (p.res) v = (v + rs) * p.res_scale;
      if (p.lens) v = ragged_mask4(v, n0 + eq, ragged_len(p.lens, b));
      if (vec4) {
        *reinterpret_cast<f32x4u*>(p.y + eidx) = v;
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (j >= e_0 && j < e_n) p.y[eidx + j] = v[j];
      }
      return;
    }
    // up > 1: transposed conv as `up` phase GEMMs (row m = co*up + ph -> sample t = q*up + ph): element e of the tile's
    // (co, t) range, consecutive threads = consecutive samples
    constexpr int LBN = (BN == 64) ? 6 : 5;
    const int up = p.up;
    const int co_first = (int)__umulhi((unsigned)m0, p.magic_up);
    const int nco = (int)__umulhi((unsigned)m_hi, p.magic_up) - co_first + 1;
    const int total = nco * BN * up;
    for (int e = tid; e < total; e += NT) {
      const int rest = (int)__umulhi((unsigned)e, p.magic_up);  // e / up
      const int ph = e - rest * up;
      const int q = rest & (BN - 1);
      const int co = co_first + (rest >> LBN);
      const int m = co * up + ph;
      const int t = (n0 + q) * up + ph;
      if (m < m0 || m > m_hi || (n0 + q) >= p.Nq || t >= p.Tout) continue;
      const int lds = (m - m0) * EP + q;
      const size_t idx = ybase + (size_t)co * p.Tout + t;
      float v = Es[lds];
#pragma unroll
      for (int k = 1; k < WK; k++) v += Es[k * BM * EP + lds];
      if (p.in_scale) v *= insc;
      v += p.bias[co];
      if (p.add) v = (v + p.add[idx]) * p.add_scale;
      if (filmb) v = filmb[co] * v + filmb[p.Cout + co];
      if (p.res) v = (v + p.res[idx]) * p.res_scale;
      if (p.lens && t >= ragged_len(p.lens, b)) v = 0.f;
      p.y[idx] = v;
    }
  }
};

// One ring slot of the direct kernel: GP channel pairs x KW taps.  Loads and waits are inline asm (see the kernel).
template <int KW, int TN, int GP>
__device__ __forceinline__ void direct_issue(float (&av)[GP * KW], float (&bv)[GP * KW * TN], int g, int kw, int Tin,
                                             int Mp, int CK, int lck, int avo, const int (&bvo)[KW][TN], u32x4 rx,
                                             u32x4 rw) {
#pragma unroll
  for (int q = 0; q < GP; q++) {
    const int ci = 2 * (kw + 8 * (g * GP + q));                          // first channel of the pair
    const int xso = ci * Tin * 4;
    const int wrow = ((ci >> lck) * KW) * CK + (ci & (CK - 1));          // packed row of (ci, tap 0)
#pragma unroll
    for (int k = 0; k < KW; k++) {
      // "+v": the destination is the SAME register as the slot's previous value -- one live range around the loop, so
      // the allocator has no phi to resolve with a copy (a copy of a register whose load is still in flight reads
      // garbage; tools/check_isa.py verifies the generated code)
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen"
                   : "+v"(av[q * KW + k]) : "v"(avo), "s"(rw), "s"((wrow + k * CK) * Mp * 4));
#pragma unroll
      for (int j = 0; j < TN; j++)
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen"
                     : "+v"(bv[(q * KW + k) * TN + j]) : "v"(bvo[k][j]), "s"(rx), "s"(xso));
    }
  }
}
// wait until at most OUT groups issued after this slot's are still in flight (loads return in order), then the MFMAs
template <int NA, int NB, int TN, int OUT>
__device__ __forceinline__ void direct_mma(float (&av)[NA], float (&bv)[NB], floatx16 (&acc)[TN], float alpha) {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OUT * (NA + NB)));
#pragma unroll
  for (int u = 0; u < NA; u++) asm volatile("" : "+v"(av[u]));  // the registers are valid only past the wait
#pragma unroll
  for (int u = 0; u < NB; u++) asm volatile("" : "+v"(bv[u]));
#pragma unroll
  for (int u = 0; u < NA; u++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const float x = bv[u * TN + j];
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], x >= 0.f ? x : alpha * x, acc[j], 0, 0, 0);
    }
}

template <int KW, int TN, int GP>
__global__ __launch_bounds__(512) void conv_direct_kernel(ConvArgs p) {
  constexpr int D = 4;
  constexpr int NA = GP * KW, NB = GP * KW * TN;  // A / B dwords per group
  static_assert(D * (NA + NB) <= 60, "loads in flight must fit vmcnt");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int kw = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_m, tile_n;
  if (!direct_tile(p, tile_m, tile_n)) return;
  // (tile_bn / tile_bm / tile_halo: BN, 32, 0 -- except with the fused up-path FIR, whose tiles overlap by a frame on
  // either side and hold whole output channels only, see DirectEpilogue)
  const int n0 = tile_n * p.tile_bn - p.tile_halo, m0 = tile_m * p.tile_bm, b = blockIdx.z;
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());

  const int lhalf = lane >> 5, l31 = lane & 31;
  const int CK = p.CK, lck = 31 - __clz(CK);
  const int Tin = p.Tin, Mp = p.Mp;
  const float alpha = p.act ? p.alpha_val : 1.0f;
  // The loads and their vmcnt waits are inline asm: the compiler's own wait-count insertion resolves a register ring
  // carried around a loop to vmcnt(0) at the loop header, which is exactly the serialisation this kernel exists to avoid.
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin, (unsigned)p.Cin * (unsigned)Tin * 4u);
  const u32x4 rw = direct_desc(p.w, (unsigned)p.Cin * (unsigned)KW * (unsigned)Mp * 4u);
  // per-lane byte offsets: A = (half row, m); B = (half row, t) per (tap, tile) with the zero padding folded in
  const int avo = (lhalf * Mp + m0 + l31) * 4;
  int bvo[KW][TN];
#pragma unroll
  for (int k = 0; k < KW; k++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int t = n0 + 32 * j + l31 + k - p.pad;
      bvo[k][j] = (t >= 0 && t < Tin) ? (lhalf * Tin + t) * 4 : (int)0x80000000;  // past the buffer: reads as 0
    }
  const int NG = (p.Cin >> 4) / GP;  // groups per wave (launcher: a multiple of D)
  float av[D][NA], bv[D][NB];
#pragma unroll
  for (int d0 = 0; d0 < D; d0++) {
#pragma unroll
    for (int u = 0; u < NA; u++) av[d0][u] = 0.f;
#pragma unroll
    for (int u = 0; u < NB; u++) bv[d0][u] = 0.f;
  }
  floatx16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
#define OU_ISSUE(g, d) direct_issue<KW, TN, GP>(av[d], bv[d], (g), kw, Tin, Mp, CK, lck, avo, bvo, rx, rw)
#define OU_MMA(d, out) direct_mma<NA, NB, TN, out>(av[d], bv[d], acc, alpha)
  // tuning only (OU_TS): per-wave phase stamps -- {start (10 ns ticks), cycles: prologue, first data, main loop, drain,
  // epilogue, -, end (ticks)}
  const bool ts_on = p.tstamps != nullptr;
  long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, r0 = 0;
  if (ts_on) { r0 = (long long)__builtin_amdgcn_s_memrealtime(); c0 = __builtin_readcyclecounter(); }
  OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
  if (ts_on) c1 = __builtin_readcyclecounter();

  // ---- main loop: rounds of D groups; the last round issues nothing
  const int NR = NG / D;
  for (int r = 0; r + 1 < NR; r++) {
    const int g = r * D;
    OU_MMA(0, 3);
    if (ts_on && r == 0) c2 = __builtin_readcyclecounter();
    OU_ISSUE(g + 4, 0);
    OU_MMA(1, 3); OU_ISSUE(g + 5, 1);
    OU_MMA(2, 3); OU_ISSUE(g + 6, 2);
    OU_MMA(3, 3); OU_ISSUE(g + 7, 3);
  }
  if (ts_on) c3 = __builtin_readcyclecounter();
  OU_MMA(0, 3); OU_MMA(1, 2); OU_MMA(2, 1); OU_MMA(3, 0);
  if (ts_on) c4 = __builtin_readcyclecounter();
#undef OU_ISSUE
#undef OU_MMA

  DirectEpilogue<TN>::run(p, acc, smem, tid, kw, b, m0, n0);
  if (ts_on && lane == 0) {
    const long long c5 = __builtin_readcyclecounter();
    long long* o = p.tstamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + kw) * 8;
    o[0] = r0; o[1] = c1 - c0; o[2] = (NR > 1 ? c2 : c4) - c1; o[3] = NR > 1 ? c3 - c2 : 0; o[4] = c4 - c3; o[5] = c5 - c4;
    o[6] = 0; o[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// Epilogue of conv_direct2_kernel (stride 1, up == 1, no fused FIR): the accumulators of the WK K-slice waves -> LDS -> reduced
// on read -> bias, cond add, FiLM, residual -> 16-byte stores.  Each thread owns QPT quads of four consecutive samples of one
// output row.  Its global operands (cond add, residual, bias, FiLM row) are fetched by `issue()` BEFORE THE DRAIN of the
// operand ring -- the last D slots of MFMAs, about one memory latency long, during which no ring load is issued any more --
// with inline-asm buffer loads that join the ring's in-order vmcnt accounting (NLOAD more loads outstanding behind every ring
// slot still in flight).  The first generation fetched them after the main loop and paid one exposed L2 / Infinity-Cache
// round trip per launch; fetching them before the main loop kept 11-36 registers live across it and cost occupancy (see
// DirectEpilogue).  Here their registers are live across the drain only: the allocator takes them from what the main loop's
// address arithmetic has freed.  Absent operands use a zero-length descriptor (reads as 0, no memory access); quads that cross
// the end of a row read the neighbouring row's samples into elements that are never used.
template <int TN, int WK>
struct Direct2Epilogue {
  static constexpr int NT = 64 * WK, BM = 32, BN = 32 * TN, EP = BN + 4, C4 = BN / 4, NQ = BM * C4;
  static constexpr int QPT = (NQ + NT - 1) / NT;  // quads per thread: 1 (half the threads idle at TN = 1, WK = 8) or 2
  static constexpr int NLOAD = 5 * QPT;           // buffer loads issue() puts in flight
  struct Pre {
    f32x4 ad[QPT], rs[QPT];
    float bi[QPT], ga[QPT], be[QPT];
  };
  static __device__ __forceinline__ void issue(const ConvArgs& p, Pre& q, int tid, int b, int m0, int n0) {
    const unsigned ybytes = (unsigned)p.Cout * (unsigned)p.Tout * 4u;
    const size_t ybase = (size_t)b * p.Cout * p.Tout;
    const u32x4 ra = direct_desc(p.add ? p.add + ybase : p.y, p.add ? ybytes : 0u);
    const u32x4 rr = direct_desc(p.res ? p.res + ybase : p.y, p.res ? ybytes : 0u);
    const u32x4 rb = direct_desc(p.bias, (unsigned)p.Cout * 4u);
    const u32x4 rf = direct_desc(p.film ? p.film + (size_t)b * p.film_bstride : p.bias, p.film ? 2u * (unsigned)p.Cout * 4u : 0u);
#pragma unroll
    for (int j = 0; j < QPT; j++) {
      const int i = tid + j * NT, er = i / C4, eq = (i % C4) * 4, m = m0 + er;
      const bool on = i < NQ && m < p.M && n0 + eq < p.Nq;
      const int vo = on ? (m * p.Tout + n0 + eq) * 4 : (int)0x80000000;
      const int vb = on ? m * 4 : (int)0x80000000;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q.ad[j]) : "v"(vo), "s"(ra));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(q.rs[j]) : "v"(vo), "s"(rr));
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(q.bi[j]) : "v"(vb), "s"(rb));
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(q.ga[j]) : "v"(vb), "s"(rf));
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(q.be[j]) : "v"(vb), "s"(rf), "s"(p.Cout * 4));
    }
  }
  // [c_lo, c_hi): output columns this tile may STORE (the fused ConvBlock kernel computes halo columns that belong to a
  // neighbouring tile group)
  static __device__ __forceinline__ void run(const ConvArgs& p, const floatx16 (&acc)[TN], Pre& q, float* Es, int tid, int kw,
                                             int b, int m0, int n0, int c_lo, int c_hi) {
    const int lane = tid & 63, lhalf = lane >> 5, l31 = lane & 31;
    if constexpr (TN == 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        *reinterpret_cast<f32x2*>(&Es[(kw * BM + row) * EP + 2 * l31]) = f32x2{acc[0][r], acc[1][r]};
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        Es[(kw * BM + row) * EP + l31] = acc[0][r];
      }
    }
    // the prefetched operands have landed long ago (they were issued before the drain); the registers are valid only past
    // this wait and the empty asm statements that follow it
    asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
    for (int j = 0; j < QPT; j++) {
      asm volatile("" : "+v"(q.ad[j]));
      asm volatile("" : "+v"(q.rs[j]));
      asm volatile("" : "+v"(q.bi[j]));
      asm volatile("" : "+v"(q.ga[j]));
      asm volatile("" : "+v"(q.be[j]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const size_t ybase = (size_t)b * p.Cout * p.Tout;
    const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
#pragma unroll
    for (int j = 0; j < QPT; j++) {
      const int i = tid + j * NT, er = i / C4, eq = (i % C4) * 4, m = m0 + er;
      if (!(i < NQ && m < p.M && n0 + eq < p.Nq && n0 + eq + 4 > c_lo && n0 + eq < c_hi)) continue;
      const size_t eidx = ybase + (size_t)m * p.Tout + n0 + eq;
      int e_n = p.Nq - (n0 + eq);
      if (e_n > 4) e_n = 4;
      if (e_n > c_hi - (n0 + eq)) e_n = c_hi - (n0 + eq);
      const int e_0 = c_lo - (n0 + eq) > 0 ? c_lo - (n0 + eq) : 0;  // first element of the quad inside the store range
      f32x4 v = *reinterpret_cast<const f32x4*>(&Es[er * EP + eq]);
#pragma unroll
      for (int kk = 1; kk < WK; kk++) v += *reinterpret_cast<const f32x4*>(&Es[(kk * BM + er) * EP + eq]);
      if (p.in_scale) v *= insc;
      v += q.bi[j];
      if (p.add) v = (v + q.ad[j]) * p.add_scale;
      if (p.film) v = q.ga[j] * v + q.be[j];
      if (p.res) v = (v + q.rs[j]) * p.res_scale;
      if (p.out_act) {
#pragma unroll
        for (int oa = 0; oa < 4; oa++) v[oa] = v[oa] >= 0.f ? v[oa] : p.out_alpha * v[oa];
      }
      if (p.lens) v = ragged_mask4(v, n0 + eq, ragged_len(p.lens, b));
      if (e_0 == 0 && e_n == 4) {  // 16-byte store at dword alignment (the 401- / 2005-frame levels too)
        *reinterpret_cast<f32x4u*>(p.y + eidx) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (e >= e_0 && e < e_n) p.y[eidx + e] = v[e];
      }
    }
  }
};

// One ring slot of conv_direct2_kernel after its wait: window -> (edge fix-up) -> PReLU -> KW x TN MFMAs.
template <int KW, int TN>
__device__ __forceinline__ void direct2_mma(const f32x4& a4, float a1, const f32x4& b4, float b1, const f32x2& b2,
                                            floatx16 (&acc)[TN], float alpha, bool edge, int sh, unsigned vmask) {
  constexpr int W = KW + TN - 1, PAD = (KW - 1) / 2;
  const float L[6] = {b4.x, b4.y, b4.z, b4.w, W == 5 ? b1 : b2.x, b2.y};
  float X[W];
  if (edge) {  // block-uniform: first / last column tiles only
#pragma unroll
    for (int i = 0; i < W; i++) {
      float v = L[i];  // sh == 0
#pragma unroll
      for (int s = 1; s <= PAD; s++) v = sh == s ? (i - s >= 0 ? L[i - s >= 0 ? i - s : 0] : 0.f) : v;
      X[i] = ((vmask >> i) & 1u) ? v : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < W; i++) X[i] = L[i];
  }
#pragma unroll
  for (int i = 0; i < W; i++) X[i] = X[i] >= 0.f ? X[i] : alpha * X[i];
  const float A[5] = {a4.x, a4.y, a4.z, a4.w, a1};
#pragma unroll
  for (int k = 0; k < KW; k++)
#pragma unroll
    for (int q = 0; q < TN; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[k], X[q + k], acc[q], 0, 0, 0);
}

// One ring slot of conv_direct2w_kernel after its wait: window -> (edge fix-up) -> PReLU -> B^T -> KW + 1 MFMAs on independent
// accumulators.  B^T of F(2, 3): points 0, 1, -1, inf; of F(2, 5): points 0, 1, -1, 1/2, -2, inf, rows scaled to small integers
// (the scale is in G, ou_model.cpp; tests/test_packing.py holds U, B^T and A^T against the convolution they must reproduce).
template <int KW, bool EDGE, bool ACT>
__device__ __forceinline__ void direct2w_mma(const f32x4& a4, const f32x2& a2, const f32x4& b4, const f32x2& b2,
                                             floatx16 (&acc)[KW + 1], float alpha, unsigned vmask) {
  constexpr int W = KW + 1;
  const float L[6] = {b4.x, b4.y, b4.z, b4.w, b2.x, b2.y};
  float X[W];
#pragma unroll
  for (int i = 0; i < W; i++) {
    X[i] = L[i];
    if constexpr (EDGE) X[i] = ((vmask >> i) & 1u) ? X[i] : 0.f;  // first / last column tiles only (the window starts in front of the row
                                                                 // or ends behind it: neighbouring rows' samples, not zeros)
    if constexpr (ACT) X[i] = X[i] >= 0.f ? X[i] : alpha * X[i];  // (ACT = false: the producer's epilogue stored activated values)
  }
  float V[W];
  wino_bt<KW>(X, V);
  const float A[6] = {a4.x, a4.y, a4.z, a4.w, a2.x, a2.y};
#pragma unroll
  for (int x = 0; x < W; x++) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[x], V[x], acc[x], 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// conv_direct2_kernel: the stride-1 k3 / k5 direct kernel with WIDE operand loads.
// The first direct kernel is bound by vector-memory instruction issue, not by MFMA or bytes: a CU retires one
// buffer_load_dword wave instruction per ~8.4 cycles whatever the data (tools/ubench/vmem_issue.hip: dword and dwordx2
// 31-43 B/clk/CU, dwordx4 58-75 B/clk/CU), and with 1.5 load instructions per MFMA (64-column tiles) the 16 waves of a
// CU spend ~9.7 k cycles issuing loads next to 12.3 k cycles of MFMAs -- the 36 prologue loads alone hold every wave for
// 4.8 k cycles before its first MFMA (tools/direct_ts.py).  Here one 16-byte load per operand feeds all taps:
//   A: a second copy of the weights with the taps innermost ([ci][m][KWP], KWP = 4 / 8): lane (m, half) gets all KW
//      taps of channel 2I + half with one dwordx4 (+ one dword for k5);
//   B: lane (n, half) loads the KW + TN - 1 consecutive samples x[2I + half][n0 + TN n - pad ...] it needs for ALL taps
//      of its TN adjacent output columns (column n0 + TN n + q reads window element q + k for tap k): one dwordx4
//      (+ dword / dwordx2 for k5).  Output columns are interleaved over the TN accumulators instead of blocked -- a
//      permutation the epilogue undoes for free (8-byte LDS writes).
//   k3, 64 columns: 2 load instructions per 6 MFMAs (was 9); k5: 4 per 10 (was 15).
// Windows that leave [0, Tin) (first / last column tiles only, block-uniform branch): lanes that would start before the
// row load from its start instead and shift their elements; elements outside the row are zeroed -- a row-crossing
// 16-byte load returns the neighbouring row's samples, not zeros.  Same K order per output element as
// conv_direct_kernel (pairs in ring order, taps ascending): bit-identical results.
// ---------------------------------------------------------------------------------------------------------
// (the tile body is a device function: conv_direct2_kernel runs it once per block, conv_block3_kernel three times with a
// group barrier in between; [c_lo, c_hi) = columns the tile may store)
template <int KW, int TN, int WK = 8>
__device__ __forceinline__ void direct2_tile(const ConvArgs& p, float* smem, int b, int m0, int n0, int c_lo, int c_hi) {
  constexpr int D = 4, W = KW + TN - 1, KWP = KW == 3 ? 4 : 8, PAD = (KW - 1) / 2;
  constexpr int B2 = W - 4;                    // elements in the second B load: 0 (none), 1 (dword), 2 (dwordx2)
  constexpr int A2 = KW - 4 > 0 ? KW - 4 : 0;  // elements in the second A load: 0 / 1
  constexpr int LPS = 1 + (A2 ? 1 : 0) + 1 + (B2 > 0 ? 1 : 0);  // load instructions per ring slot (= channel pair)
  using Epi = Direct2Epilogue<TN, WK>;
  constexpr int NE = Epi::NLOAD;               // epilogue operand loads that join the queue before the drain
  static_assert(KW == 3 || KW == 5, "k3 / k5");
  static_assert(B2 >= -1 && B2 <= 2 && D * LPS + NE <= 60, "window / vmcnt");
  constexpr int BN = 32 * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int kw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int Tin = p.Tin, Mp = p.Mp;
  const float alpha = p.act ? p.alpha_val : 1.0f;
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin, (unsigned)p.Cin * (unsigned)Tin * 4u);
  const u32x4 rw = direct_desc(p.wd, (unsigned)p.Cin * (unsigned)Mp * (unsigned)KWP * 4u);
  const int avo = ((lhalf * Mp) + m0 + l31) * KWP * 4;
  // this lane's window: samples t0 .. t0 + W - 1 of row 2I + half; `sh` = samples cut off in front of the row
  const int t0 = n0 + TN * l31 - PAD;
  const int sh = t0 < 0 ? -t0 : 0;
  const int bvo = (t0 + sh < Tin) ? (lhalf * Tin + t0 + sh) * 4 : (int)0x80000000;
  const bool edge = __builtin_amdgcn_readfirstlane((n0 < PAD || n0 + BN + KW - 1 - PAD > Tin) ? 1 : 0) != 0;
  unsigned vmask = 0;  // bit i: window element i is inside the row
#pragma unroll
  for (int i = 0; i < W; i++) vmask |= (t0 + i >= 0 && t0 + i < Tin) ? (1u << i) : 0u;

  const int NG = p.Cin / (2 * WK);  // channel pairs per wave (launcher: a multiple of D)
  f32x4 a4[D], b4[D];
  float a1[D];   // k5: tap 4
  float b1[D];   // window element 4 (5-element windows)
  f32x2 b2[D];   // window elements 4, 5 (6-element windows)
#pragma unroll
  for (int d0 = 0; d0 < D; d0++) {
    a4[d0] = f32x4{0.f, 0.f, 0.f, 0.f}; b4[d0] = f32x4{0.f, 0.f, 0.f, 0.f};
    a1[d0] = 0.f; b1[d0] = 0.f; b2[d0] = f32x2{0.f, 0.f};
  }
  floatx16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

#define OU_ISSUE(g_, d)                                                                                              \
  {                                                                                                                  \
    const int ci = 2 * (kw + WK * (g_));                                                                             \
    const int aso = ci * Mp * KWP * 4, xso = ci * Tin * 4;                                                           \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(a4[d]) : "v"(avo), "s"(rw), "s"(aso));            \
    if constexpr (A2 == 1)                                                                                           \
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:16" : "+v"(a1[d]) : "v"(avo), "s"(rw), "s"(aso));  \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(b4[d]) : "v"(bvo), "s"(rx), "s"(xso));            \
    if constexpr (B2 == 1)                                                                                           \
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:16" : "+v"(b1[d]) : "v"(bvo), "s"(rx), "s"(xso));   \
    if constexpr (B2 == 2)                                                                                           \
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:16" : "+v"(b2[d]) : "v"(bvo), "s"(rx), "s"(xso)); \
  }
// `out`: loads issued after this slot's that may still be in flight (whole ring slots x LPS, plus the epilogue operands)
#define OU_MMA(d, out)                                                                                               \
  {                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(out));                                                                  \
    asm volatile("" : "+v"(a4[d]));                                                                                  \
    asm volatile("" : "+v"(b4[d]));                                                                                  \
    if constexpr (A2 == 1) asm volatile("" : "+v"(a1[d]));                                                           \
    if constexpr (B2 == 1) asm volatile("" : "+v"(b1[d]));                                                           \
    if constexpr (B2 == 2) asm volatile("" : "+v"(b2[d]));                                                           \
    direct2_mma<KW, TN>(a4[d], a1[d], b4[d], b1[d], b2[d], acc, alpha, edge, sh, vmask);                             \
  }
  const bool ts_on = p.tstamps != nullptr;
  long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, r0 = 0;
  if (ts_on) { r0 = (long long)__builtin_amdgcn_s_memrealtime(); c0 = __builtin_readcyclecounter(); }
  OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
  if (ts_on) c1 = __builtin_readcyclecounter();
  const int NR = NG / D;
  for (int r = 0; r + 1 < NR; r++) {
    const int g = r * D;
    OU_MMA(0, 3 * LPS);
    if (ts_on && r == 0) c2 = __builtin_readcyclecounter();
    OU_ISSUE(g + 4, 0);
    OU_MMA(1, 3 * LPS); OU_ISSUE(g + 5, 1);
    OU_MMA(2, 3 * LPS); OU_ISSUE(g + 6, 2);
    OU_MMA(3, 3 * LPS); OU_ISSUE(g + 7, 3);
  }
  if (ts_on) c3 = __builtin_readcyclecounter();
  typename Epi::Pre pre;
  Epi::issue(p, pre, tid, b, m0, n0);  // the epilogue's global operands travel under the drain
  OU_MMA(0, 3 * LPS + NE); OU_MMA(1, 2 * LPS + NE); OU_MMA(2, LPS + NE); OU_MMA(3, NE);
  if (ts_on) c4 = __builtin_readcyclecounter();
#undef OU_ISSUE
#undef OU_MMA
  Epi::run(p, acc, pre, smem, tid, kw, b, m0, n0, c_lo, c_hi);
  if (ts_on && lane == 0) {
    const long long c5 = __builtin_readcyclecounter();
    long long* o = p.tstamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + kw) * 8;
    o[0] = r0; o[1] = c1 - c0; o[2] = (NR > 1 ? c2 : c4) - c1; o[3] = NR > 1 ? c3 - c2 : 0; o[4] = c4 - c3; o[5] = c5 - c4;
    o[6] = 0; o[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}
// WK = 8: 512 threads, the reduction split over eight waves (round 2).  WK = 4 (round 5): 256 threads, four K slices -- twice
// the K loop per wave for the same prologue and drain, half the LDS read traffic of the cross-wave reduction, one wave per
// SIMD and block (the waves of a block advance together: no barrier skew between two waves that share a matrix pipe), and
// twice as many blocks on a CU whose epilogues and prologues no longer coincide.
template <int KW, int TN, int WK>
__global__ __launch_bounds__(64 * WK) void conv_direct2_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int tile_m, tile_n;
  if (!direct_tile(p, tile_m, tile_n)) return;
  if (p.prof && threadIdx.x == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  direct2_tile<KW, TN, WK>(p, smem, blockIdx.z, tile_m * 32, tile_n * 32 * TN, 0, 0x7fffffff);
  if (p.prof && threadIdx.x == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

// ---------------------------------------------------------------------------------------------------------
// conv_direct2w_kernel: conv_direct2_kernel with MINIMAL FILTERING (Winograd / Cook-Toom F(2, KW)) -- round 5.
// The split-K direct kernels are bound by the fp32 matrix pipe wherever their loop runs (85-93 % of its rate), and exact
// fp32 has no faster MFMA: what is left is to issue fewer of them.  F(2, KW) computes the two adjacent outputs of a tile
// position from KW + 1 products instead of 2 KW:
//     y[2p + j] = sum_x AT[j][x] * ( sum_ci U_x[co][ci] * V_x[ci][p] ),   U = G w (packer, in double),  V = B^T d (here)
// i.e. KW + 1 GEMMs [M x Cin] x [Cin x T/2] instead of KW of [M x Cin] x [Cin x T]: 4 instead of 6 MFMAs per channel pair and
// 32 x 64 output tile for k3, 6 instead of 10 for k5.  Everything else is conv_direct2_kernel's: lane (n, half) already
// loads the KW + 1 consecutive samples d[2 n - PAD ..] of row 2 I + half that its two adjacent output columns need (ONE
// 16-byte load [+ 8 bytes]), lane (m, half) the KW + 1 values U_x of (row m, channel 2 I + half) with one 16-byte load [+ 8]
// from the third weight copy -- k3: the same bytes as before (the fourth float of the slot is no longer padding), k5: 24
// instead of 20 -- the ring, the counted waits, the prefetched epilogue operands.  PReLU is applied to the samples, then
// B^T (4 VALU operations for k3, ~20 for k5: small integers only), KW + 1 independent accumulators (no dependent MFMA chain);
// A^T is applied to the accumulators once per wave before the cross-wave reduction, which then sees two accumulators as ever.
// Numerics: fp32 throughout; against a double evaluation F(2, 3) loses 2 dB to the plain summation (129-132 vs 131-133 dB per
// layer), F(2, 5) with the points 0, +-1, 1/2, -2 about 9 (122-125 dB) -- two and a half orders of magnitude inside the 60 dB
// tolerance of the path, see DESIGN.md 5.  64-column tiles only (32 tile positions = one MFMA's columns): the 401-frame
// levels at batch 1 (32-column tiles) stay on conv_direct2_kernel.  OU_WINO=0 switches the variant off.
// ---------------------------------------------------------------------------------------------------------
// EDGE: this tile's windows leave the row (first / last column tile); ACT: PReLU in the operand path (false: the producer's
// epilogue stored activated values, ConvArgs::out_act).  Whole-function variants behind the kernel's block-uniform branch: each
// has its own register allocation (loop-level variants inside one function made the compiler spill the ring).
template <int KW, int WK, bool EDGE, bool ACT>
__device__ __forceinline__ void direct2w_tile(const ConvArgs& p, float* smem, int b, int m0, int n0) {
  constexpr int D = 4, TN = 2, NX = KW + 1, W = NX, KWP = KW == 3 ? 4 : 8, PAD = (KW - 1) / 2;
  constexpr int X2 = NX - 4;                  // elements in the second load of either operand: 0 (k3) / 2 (k5)
  constexpr int LPS = X2 ? 4 : 2;             // load instructions per ring slot (= channel pair)
  using Epi = Direct2Epilogue<TN, WK>;
  constexpr int NE = Epi::NLOAD;
  static_assert(KW == 3 || KW == 5, "k3 / k5");
  static_assert(D * LPS + NE <= 60, "vmcnt");
  const int tid = threadIdx.x, lane = tid & 63;
  const int kw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int Tin = p.Tin, Mp = p.Mp;
  const float alpha = p.act ? p.alpha_val : 1.0f;
  // (the descriptor starts PAD samples in front of the tensor -- workspace memory, never the first bytes of an allocation --: the
  // window of the first lane of the first tile, which begins at t = -PAD, is an in-range load whose leading elements are masked)
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin - PAD, ((unsigned)p.Cin * (unsigned)Tin + PAD) * 4u);
  const u32x4 rw = direct_desc(p.wu, (unsigned)p.Cin * (unsigned)Mp * (unsigned)KWP * 4u);
  const int avo = ((lhalf * Mp) + m0 + l31) * KWP * 4;
  // this lane's window: samples t0 .. t0 + W - 1 of row 2I + half
  const int t0 = n0 + TN * l31 - PAD;
  const int bvo = (t0 < Tin) ? (lhalf * Tin + t0 + PAD) * 4 : (int)0x80000000;
  unsigned vmask = 0;  // bit i: window element i is inside the row
#pragma unroll
  for (int i = 0; i < W; i++) vmask |= (t0 + i >= 0 && t0 + i < Tin) ? (1u << i) : 0u;

  const int NG = p.Cin / (2 * WK);  // channel pairs per wave (launcher: a multiple of D)
  f32x4 a4[D], b4[D];
  f32x2 a2[D], b2[D];  // k5: U_4, U_5 / window elements 4, 5
#pragma unroll
  for (int d0 = 0; d0 < D; d0++) {
    a4[d0] = f32x4{0.f, 0.f, 0.f, 0.f}; b4[d0] = f32x4{0.f, 0.f, 0.f, 0.f};
    a2[d0] = f32x2{0.f, 0.f}; b2[d0] = f32x2{0.f, 0.f};
  }
  floatx16 acc[NX];
#pragma unroll
  for (int j = 0; j < NX; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

#define OU_ISSUE(g_, d)                                                                                              \
  {                                                                                                                  \
    const int ci = 2 * (kw + WK * (g_));                                                                             \
    const int aso = ci * Mp * KWP * 4, xso = ci * Tin * 4;                                                           \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(a4[d]) : "v"(avo), "s"(rw), "s"(aso));            \
    if constexpr (X2 == 2)                                                                                           \
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:16" : "+v"(a2[d]) : "v"(avo), "s"(rw), "s"(aso)); \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(b4[d]) : "v"(bvo), "s"(rx), "s"(xso));            \
    if constexpr (X2 == 2)                                                                                           \
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:16" : "+v"(b2[d]) : "v"(bvo), "s"(rx), "s"(xso)); \
  }
#define OU_MMA(d, out)                                                                                               \
  {                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(out));                                                                  \
    asm volatile("" : "+v"(a4[d]));                                                                                  \
    asm volatile("" : "+v"(b4[d]));                                                                                  \
    if constexpr (X2 == 2) asm volatile("" : "+v"(a2[d]));                                                           \
    if constexpr (X2 == 2) asm volatile("" : "+v"(b2[d]));                                                           \
    direct2w_mma<KW, EDGE, ACT>(a4[d], a2[d], b4[d], b2[d], acc, alpha, vmask);                                      \
  }
  // tuning only (OU_TS): per-wave phase stamps, as in conv_direct2_kernel
  const bool ts_on = p.tstamps != nullptr;
  long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, r0 = 0;
  if (ts_on) { r0 = (long long)__builtin_amdgcn_s_memrealtime(); c0 = __builtin_readcyclecounter(); }
  OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
  if (ts_on) c1 = __builtin_readcyclecounter();
  const int NR = NG / D;
  for (int r = 0; r + 1 < NR; r++) {
    const int g = r * D;
    OU_MMA(0, 3 * LPS);
    if (ts_on && r == 0) c2 = __builtin_readcyclecounter();
    OU_ISSUE(g + 4, 0);
    OU_MMA(1, 3 * LPS); OU_ISSUE(g + 5, 1);
    OU_MMA(2, 3 * LPS); OU_ISSUE(g + 6, 2);
    OU_MMA(3, 3 * LPS); OU_ISSUE(g + 7, 3);
  }
  if (ts_on) c3 = __builtin_readcyclecounter();
  typename Epi::Pre pre;
  Epi::issue(p, pre, tid, b, m0, n0);  // the epilogue's global operands travel under the drain
  OU_MMA(0, 3 * LPS + NE); OU_MMA(1, 2 * LPS + NE); OU_MMA(2, LPS + NE); OU_MMA(3, NE);
  if (ts_on) c4 = __builtin_readcyclecounter();
#undef OU_ISSUE
#undef OU_MMA
  // A^T: the wave's KW + 1 partial GEMM results -> its two output accumulators (even / odd columns of the tile)
  floatx16 out[TN];
  if constexpr (KW == 3) {
    out[0] = acc[0] + acc[1] + acc[2];
    out[1] = acc[1] - acc[2] - acc[3];
  } else {
    out[0] = acc[0] + acc[1] + acc[2] + acc[3] + acc[4];
    out[1] = acc[1] - acc[2] + 0.5f * acc[3] - 2.0f * acc[4] + acc[5];
  }
  Epi::run(p, out, pre, smem, tid, kw, b, m0, n0, 0, 0x7fffffff);
  if (ts_on && lane == 0) {
    const long long c5 = __builtin_readcyclecounter();
    long long* o = p.tstamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + kw) * 8;
    o[0] = r0; o[1] = c1 - c0; o[2] = (NR > 1 ? c2 : c4) - c1; o[3] = NR > 1 ? c3 - c2 : 0; o[4] = c4 - c3; o[5] = c5 - c4;
    o[6] = 0; o[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}
template <int KW, int WK>
__global__ __launch_bounds__(64 * WK) void conv_direct2w_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int tile_m, tile_n;
  if (!direct_tile(p, tile_m, tile_n)) return;
  if (p.prof && threadIdx.x == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int n0 = tile_n * 64;
  constexpr int PAD = (KW - 1) / 2;
  const bool edge = n0 < PAD || n0 + 64 + KW - 1 - PAD > p.Tin;  // (block-uniform)
  if (edge) {
    if (p.act) direct2w_tile<KW, WK, true, true>(p, smem, blockIdx.z, tile_m * 32, n0);
    else direct2w_tile<KW, WK, true, false>(p, smem, blockIdx.z, tile_m * 32, n0);
  } else {
    if (p.act) direct2w_tile<KW, WK, false, true>(p, smem, blockIdx.z, tile_m * 32, n0);
    else direct2w_tile<KW, WK, false, false>(p, smem, blockIdx.z, tile_m * 32, n0);
  }
  if (p.prof && threadIdx.x == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

#ifdef OU_EXPERIMENTS  // built with `make EXPERIMENTS=1` only: measured, bit-identical, and slower end to end
// ---------------------------------------------------------------------------------------------------------
// conv_block3_kernel: the three body convs of a deep-level ConvBlock (k5 -> k3 -> k3, C >= 256, a few hundred to a few
// thousand frames, batch 1) in ONE launch.  The time axis is cut into eight windows, one per XCD: the 32 workgroups that
// the dispatcher places on XCD x (block ids congruent to x mod 8) compute ALL output channels of window x for all three
// convs, so everything a conv reads from its predecessor was written on the same XCD and is served by that XCD's L2 --
// plain stores, plain loads, no write-back, and only a 32-member barrier between the convs.  The halo (2 + 1 columns either
// side) is recomputed inside the window: it fits the columns the 32-column tiles waste today (401 frames = 8 x 51, window 55,
// two tiles = 64; 2005 = 8 x 251, window 255, four 64-column tiles = 256), and a window stores, per conv, only the columns
// that are valid there (conv1: the whole window -- it depends on the block input alone; conv2: own range +- 1; conv3: own
// range); overlapping stores of neighbouring windows carry bit-identical values.  Same tile body, same K order, same
// epilogues as three conv_direct2_kernel launches: bit-identical results.
// The placement is checked, not assumed: every workgroup adds its XCC id to its group's mask; a group that spans XCDs is
// counted in status word 34 (diagnostics; never observed) and hands over with the agent-scope release / acquire form, which is
// correct under any placement; spins are bounded (status bit 16).  In `make EXPERIMENTS=1` builds only, behind OU_BLOCK3=1: no
// per-layer tensors names / profile records for the three convs of a fused launch.
// ---------------------------------------------------------------------------------------------------------
struct Block3Args {
  ConvArgs cv[3];
  unsigned long long* bar;  // per XCD: 32 tag slots, epoch, XCC mask (B3_STRIDE x 8 bytes), zero-initialised
  unsigned* err;            // sticky status word
  int cpx;                  // columns per window = ceil(T / 8)
  int ncolt;                // column tiles per window
  int nrow;                 // 32-row tiles
};
// Barrier among the workgroups of one window group, the way the GRU clusters hand over h (4.4): every member owns one 8-byte
// slot and stores its tag there (sc1: visible under any placement), wave 0 of every member polls all slots with ONE load per
// lane until none is behind.  Tags only grow (epoch + 1, epoch + 2; the epoch word is advanced by member 0 after the second
// barrier), so a slot that is already one barrier ahead passes too.  An arrival counter -- one atomic word per group, 32
// arrivals serialised in one L2 channel under 31 pollers -- cost 4.8 us per barrier.
constexpr int B3_STRIDE = 40;  // u64 per group: 32 slots, epoch, XCC mask, spare
__device__ __forceinline__ void block3_sync(unsigned long long* grp, int slot, unsigned nact, unsigned long long want,
                                             unsigned* err, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores are in L2
  __syncthreads();
  if (tid < 64) {
    if (tid == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(grp + slot), "v"(want) : "memory");
    const unsigned long long* src = grp + (tid < (int)nact ? tid : 0);
    unsigned spins = 0;
    while (true) {
      unsigned long long v;
      asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
      if (__builtin_amdgcn_ballot_w64(v < want) == 0ull) break;
      if (++spins > 4000000u) { if (tid == 0) atomicOr(err, 16u); break; }
    }
  }
  __syncthreads();
}
template <int TN>
__global__ __launch_bounds__(512) void conv_block3_kernel(Block3Args a) {
  constexpr int BN = 32 * TN;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const unsigned nact = (unsigned)(a.nrow * a.ncolt);
  const int T = a.cv[0].Nq;
  const int own_lo = xcd * a.cpx, own_hi = own_lo + a.cpx < T ? own_lo + a.cpx : T;
  if ((unsigned)slot >= nact || own_lo >= T) return;  // (whole groups: the members of a group agree on both)
  const int tile_m = slot / a.ncolt, jt = slot - tile_m * a.ncolt;
  const int win0 = xcd == 0 ? 0 : own_lo - 2;
  const int m0 = tile_m * 32, n0 = win0 + BN * jt;
  unsigned long long* grp = a.bar + B3_STRIDE * xcd;
  unsigned* mask = reinterpret_cast<unsigned*>(grp + 33);
  unsigned long long epoch;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(epoch) : "v"(grp + 32) : "memory");
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __hip_atomic_fetch_or(mask, 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int dbg = a.cv[0].dbg;  // timing experiments (results invalid): 128 no barriers, 256 conv1 only
  direct2_tile<5, TN>(a.cv[0], smem, 0, m0, n0, 0, 0x7fffffff);
  if (dbg & 256) return;
  if (dbg & 128) {
    __syncthreads();
    direct2_tile<3, TN>(a.cv[1], smem, 0, m0, n0, xcd == 0 ? 0 : own_lo - 1, own_hi + 1);
    __syncthreads();
    direct2_tile<3, TN>(a.cv[2], smem, 0, m0, n0, own_lo, own_hi);
    return;
  }
  block3_sync(grp, slot, nact, epoch + 1ull, a.err, tid);
  __shared__ int sh_cross;
  if (tid == 0) {
    const unsigned mk = __hip_atomic_load(mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a group that spans XCDs (never observed; OU_DBG 64 forces the path for the tests): its plain stores are not visible
    // to all members -- status word 34 counts, and the hand-overs below become agent-scope release / acquire pairs
    sh_cross = (__popc(mk) != 1 || (dbg & 64)) ? 1 : 0;
    if (sh_cross && slot == 0) atomicAdd(a.err + 34, 1u);
  }
  __syncthreads();
  const bool cross = sh_cross != 0;
  unsigned long long want = epoch + 2ull;
  if (cross) {  // conv1's stores were plain: write the L2 back, meet again, drop what this L2 / L1 hold of other XCDs' lines
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    block3_sync(grp, slot, nact, want, a.err, tid);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    want += 1ull;
  }
  direct2_tile<3, TN>(a.cv[1], smem, 0, m0, n0, xcd == 0 ? 0 : own_lo - 1, own_hi + 1);
  if (cross) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  block3_sync(grp, slot, nact, want, a.err, tid);
  if (cross) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (slot == 0 && tid == 0) {
    // everybody has read the epoch (at entry) and the mask (after the first barrier): next launch's values
    __hip_atomic_store(mask, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(grp + 32), "v"(want) : "memory");
  }
  direct2_tile<3, TN>(a.cv[2], smem, 0, m0, n0, own_lo, own_hi);
}
hipError_t init_block3_kernels() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_block3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_block3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
}
// cv[0..2] = conv1 (k5), conv2 (k3), conv3 (k3) of one ConvBlock as conv() would launch them.  hipErrorInvalidConfiguration:
// not a shape for this kernel (the caller launches the three convs separately).
hipError_t launch_conv_block3(const ConvArgs* cv, unsigned long long* bar, unsigned* err, int num_cu, hipStream_t st,
                              int* cfg_out) {
  if (num_cu != 256 || !bar || !err) return hipErrorInvalidConfiguration;
  const int C = cv[0].Cin, T = cv[0].Nq;
  const int kws[3] = {5, 3, 3};
  for (int s = 0; s < 3; s++) {
    const ConvArgs& a = cv[s];
    if (a.B != 1 || a.KW != kws[s] || a.stride != 1 || a.up != 1 || a.pad != (a.KW - 1) / 2 || !a.wd || a.fir || a.in_scale ||
        a.Cin != C || a.M != C || a.Cout != C || a.Nq != T || a.Tin != T || a.Tout != T || a.force_cfg >= 0 || a.prof ||
        a.tstamps)
      return hipErrorInvalidConfiguration;
    if ((long)a.Cin * a.Tin * 4 >= (1L << 31) || (long)a.Cin * a.Mp * 8 * 4 >= (1L << 31)) return hipErrorInvalidConfiguration;
  }
  if (C % 64 || C < 256 || T < 64) return hipErrorInvalidConfiguration;
  if (cv[1].x != cv[0].y || cv[2].x != cv[1].y) return hipErrorInvalidConfiguration;
  const int nrow = C / 32, cpx = (T + 7) / 8, win = cpx + 4;
  int tn = 0, ncolt = 0;
  for (int t = 1; t <= 2; t++) {
    const int n = (win + 32 * t - 1) / (32 * t);
    if (nrow * n <= 32) { tn = t; ncolt = n; break; }
  }
  if (!tn) return hipErrorInvalidConfiguration;
  Block3Args ba;
  for (int s = 0; s < 3; s++) ba.cv[s] = cv[s];
  ba.bar = bar; ba.err = err; ba.cpx = cpx; ba.ncolt = ncolt; ba.nrow = nrow;
  const size_t smem = (size_t)8 * 32 * (32 * tn + 4) * 4;
  if (cfg_out) *cfg_out = 300 + tn;
  if (tn == 1) hipLaunchKernelGGL(conv_block3_kernel<1>, dim3(256), dim3(512), smem, st, ba);
  else hipLaunchKernelGGL(conv_block3_kernel<2>, dim3(256), dim3(512), smem, st, ba);
  return hipGetLastError();
}
#else
hipError_t init_block3_kernels() { return hipSuccess; }
hipError_t launch_conv_block3(const ConvArgs*, unsigned long long*, unsigned*, int, hipStream_t, int*) {
  return hipErrorInvalidConfiguration;  // "not a shape for it": the caller launches the three convs separately
}
#endif  // OU_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------
// Strided variant of the direct kernel: Conv1d with stride R and KW = G*R taps, pad = (G - 1)/2 * R -- the rate-change
// (down) convs: k = s = r (G = 1), or 3r taps with the binomial anti-alias FIR folded into the weights (G = 3, see the
// packer).  Output column q reads x[(q + g - pad/R) R + j], j < R, for its tap block g: R CONSECUTIVE samples, so a
// lane fetches a whole tap block with one or two wide loads (dwordx2 / x4 [+ dword]) that are contiguous across the
// lanes of a half wave -- a lane-strided dword per tap would cost R times the cache-line traffic.  One ring slot =
// (channel pair, tap block): R A dwords + the wide B loads, R x TN MFMAs.
// ---------------------------------------------------------------------------------------------------------
template <int R, int G, int TN>
__global__ __launch_bounds__(512) void conv_direct_strided_kernel(ConvArgs p) {
  constexpr int WK = 8, D = 4, BM = 32, BN = 32 * TN, KW = R * G;
  constexpr int N4 = R / 4, N2 = (R % 4) / 2, N1 = R % 2;  // a run of R samples as 16 / 8 / 4-byte loads
  constexpr int NLD = N4 + N2 + N1;                          // load instructions per run
  constexpr int LPG = R + TN * NLD;                          // ... per ring slot
  static_assert(D * LPG <= 60, "loads in flight must fit vmcnt");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int kw = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_m, tile_n;
  if (!direct_tile(p, tile_m, tile_n)) return;
  const int n0 = tile_n * BN, m0 = tile_m * BM, b = blockIdx.z;
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int CK = p.CK, lck = 31 - __clz(CK);
  const int Tin = p.Tin, Mp = p.Mp;
  const float alpha = p.act ? p.alpha_val : 1.0f;
  const u32x4 rx = direct_desc(p.x + (size_t)b * p.Cin * Tin, (unsigned)p.Cin * (unsigned)Tin * 4u);
  const u32x4 rw = direct_desc(p.w, (unsigned)p.Cin * (unsigned)KW * (unsigned)Mp * 4u);
  const int avo = (lhalf * Mp + m0 + l31) * 4;
  // B offsets per (tap block, tile): frame (n0 + 32 j + n) + g - pad/R of R samples; whole frames are inside or outside
  // the signal (the launcher checks Tin == Nq * R), outside -> past the buffer bounds -> zeros
  int bvo[G][TN];
#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int j = 0; j < TN; j++) {
      const int fr = n0 + 32 * j + l31 + g - p.pad / R;
      bvo[g][j] = (fr >= 0 && fr < p.Nq) ? (lhalf * Tin + fr * R) * 4 : (int)0x80000000;
    }
  const int NS = (p.Cin >> 4) * G;  // ring slots per wave: (pair, tap block), tap block fastest (launcher: multiple of D)
  float av[D][R];
  f32x4 bq[D][TN][N4 ? N4 : 1];
  f32x2 bd[D][TN];
  float bs[D][TN];
#pragma unroll
  for (int d0 = 0; d0 < D; d0++) {
#pragma unroll
    for (int k = 0; k < R; k++) av[d0][k] = 0.f;
#pragma unroll
    for (int j = 0; j < TN; j++) {
#pragma unroll
      for (int i = 0; i < (N4 ? N4 : 1); i++) bq[d0][j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      bd[d0][j] = f32x2{0.f, 0.f};
      bs[d0][j] = 0.f;
    }
  }
  floatx16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
#define OU_ISSUE(s_, d)                                                                                     \
  {                                                                                                         \
    const int s = (s_);                                                                                     \
    const int pr = G == 1 ? s : s / G, g = s - pr * G;                                                      \
    const int ci = 2 * (kw + WK * pr);                                                                      \
    const int wrow = ((ci >> lck) * KW + g * R) * CK + (ci & (CK - 1));                                     \
    _Pragma("unroll") for (int k = 0; k < R; k++)                                                           \
      asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(av[d][k]) : "v"(avo), "s"(rw), "s"((wrow + k * CK) * Mp * 4)); \
    _Pragma("unroll") for (int j = 0; j < TN; j++) {                                                        \
      int bsel = bvo[0][j];                                                                                 \
      if (G > 1 && g == 1) bsel = bvo[G > 1 ? 1 : 0][j];                                                    \
      if (G > 2 && g == 2) bsel = bvo[G > 2 ? 2 : 0][j];                                                    \
      const int xso = ci * Tin * 4;                                                                         \
      _Pragma("unroll") for (int i = 0; i < N4; i++)                                                        \
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "+v"(bq[d][j][i]) : "v"(bsel), "s"(rx), "s"(xso), "n"(16 * i)); \
      if constexpr (N2 == 1)                                                                                \
        asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen offset:%4" : "+v"(bd[d][j]) : "v"(bsel), "s"(rx), "s"(xso), "n"(16 * N4)); \
      if constexpr (N1 == 1)                                                                                \
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "+v"(bs[d][j]) : "v"(bsel), "s"(rx), "s"(xso), "n"(16 * N4 + 8 * N2)); \
    }                                                                                                       \
  }
#define OU_MMA(d, out)                                                                                      \
  {                                                                                                         \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPG));                                                 \
    _Pragma("unroll") for (int k = 0; k < R; k++) asm volatile("" : "+v"(av[d][k]));                        \
    _Pragma("unroll") for (int j = 0; j < TN; j++) {                                                        \
      _Pragma("unroll") for (int i = 0; i < N4; i++) asm volatile("" : "+v"(bq[d][j][i]));                  \
      if constexpr (N2 == 1) asm volatile("" : "+v"(bd[d][j]));                                             \
      if constexpr (N1 == 1) asm volatile("" : "+v"(bs[d][j]));                                             \
    }                                                                                                       \
    _Pragma("unroll") for (int k = 0; k < R; k++)                                                           \
      _Pragma("unroll") for (int j = 0; j < TN; j++) {                                                      \
        float x;                                                                                            \
        if (k < 4 * N4) x = bq[d][j][k / 4 < N4 ? k / 4 : 0][k % 4];                                        \
        else if (k < 4 * N4 + 2 * N2) x = bd[d][j][(k - 4 * N4) % 2];                                       \
        else x = bs[d][j];                                                                                  \
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d][k], x >= 0.f ? x : alpha * x, acc[j], 0, 0, 0); \
      }                                                                                                     \
  }
  OU_ISSUE(0, 0); OU_ISSUE(1, 1); OU_ISSUE(2, 2); OU_ISSUE(3, 3);
  const int NR = NS / D;
  for (int r = 0; r + 1 < NR; r++) {
    const int s0 = r * D;
    OU_MMA(0, 3); OU_ISSUE(s0 + 4, 0);
    OU_MMA(1, 3); OU_ISSUE(s0 + 5, 1);
    OU_MMA(2, 3); OU_ISSUE(s0 + 6, 2);
    OU_MMA(3, 3); OU_ISSUE(s0 + 7, 3);
  }
  OU_MMA(0, 3); OU_MMA(1, 2); OU_MMA(2, 1); OU_MMA(3, 0);
#undef OU_ISSUE
#undef OU_MMA
  DirectEpilogue<TN>::run(p, acc, smem, tid, kw, b, m0, n0);
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

struct DirectCfg {
  int KW, TN, GP;
  void (*kern)(ConvArgs);
};
static const DirectCfg kDirectCfgs[] = {
    {1, 1, 4, conv_direct_kernel<1, 1, 4>}, {1, 1, 2, conv_direct_kernel<1, 1, 2>},
    {1, 2, 4, conv_direct_kernel<1, 2, 4>}, {1, 2, 2, conv_direct_kernel<1, 2, 2>},
    {3, 1, 2, conv_direct_kernel<3, 1, 2>}, {3, 1, 1, conv_direct_kernel<3, 1, 1>},
    {3, 2, 1, conv_direct_kernel<3, 2, 1>},
    {5, 1, 1, conv_direct_kernel<5, 1, 1>}, {5, 2, 1, conv_direct_kernel<5, 2, 1>},
};

struct StridedCfg {
  int R, G, TN;
  void (*kern)(ConvArgs);
};
// 64-column tiles only: with the same source, the 32-column instantiations come out of the register allocator with phi
// copies of ring registers whose loads are still in flight (tools/check_isa.py) -- and a 32 x 64 tile per wave has the
// same MFMA time per CU as two waves with 32 x 32 tiles, with half the A traffic.
#define OU_STRIDED(R, G) {R, G, 2, conv_direct_strided_kernel<R, G, 2>}
// (The G = 3 instantiations -- 3r taps, anti-alias FIR folded into the weights, OU_FIR_FOLD -- are not built: the compiler
// gives each of them a 32-byte private segment (10-14 scratch instructions around the ring), and the folded form lost to the
// separate FIR pass on every level anyway; with OU_FIR_FOLD those layers run on conv_mfma_kernel.  Every kernel that IS
// dispatched has private_segment_fixed_size 0 -- `make check` verifies it.)
static const StridedCfg kStridedCfgs[] = {
    OU_STRIDED(2, 1), OU_STRIDED(3, 1), OU_STRIDED(4, 1), OU_STRIDED(5, 1), OU_STRIDED(8, 1),
};

// Tile width, K split and arithmetic of the wide-load kernel for one layer.  force_cfg 105 / 106: eight slices, 64 / 32
// columns (round 2); 107 / 108: four slices; 109 / 110: minimal filtering (conv_direct2w_kernel), eight / four slices.
static void direct2_pick(const ConvArgs& a, int num_cu, int& tn, int& wk, bool& wino) {
  const long gm = (a.M + 31) / 32;
  const long b64 = gm * ((a.Nq + 63) / 64) * a.B;
  wk = 8;  // tn: the caller's block-round rule
  const bool wk4_ok = (a.Cin / 8) % 4 == 0;  // channel pairs per wave with four slices: whole rounds of the ring
  if (a.d2_wk == 4 && wk4_ok) wk = 4;
  // Minimal filtering wherever the block-round rule takes 64-column tiles.  Residency: the k5 form holds six accumulators
  // (~170 VGPRs: two waves per SIMD), so a launch that needs two 8-wave blocks per CU to run in one round splits the
  // reduction over four waves instead (two 4-wave blocks per CU: the same two waves per SIMD).
  wino = a.wino && a.direct >= 5 && a.wu && tn == 2;
  if (wino && a.KW == 5 && b64 > num_cu && a.d2_wk != 8) {
    if (wk4_ok) wk = 4; else wino = false;
  }
  if (a.force_cfg == 105) { tn = 2; wk = 8; wino = false; }
  if (a.force_cfg == 106) { tn = 1; wk = 8; wino = false; }
  if (a.force_cfg == 107 && wk4_ok) { tn = 2; wk = 4; wino = false; }
  if (a.force_cfg == 108 && wk4_ok) { tn = 1; wk = 4; wino = false; }
  if (a.force_cfg == 109 && a.wu) { tn = 2; wk = 8; wino = true; }
  if (a.force_cfg == 110 && a.wu && wk4_ok) { tn = 2; wk = 4; wino = true; }
}

// Launches a direct kernel when the layer fits one; hipErrorInvalidConfiguration = "use conv_mfma_kernel".
hipError_t launch_conv_direct(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (a.Cin % 16 || (a.in_scale != nullptr && a.act)) return hipErrorInvalidConfiguration;
  if ((long)a.Cin * a.Tin * 4 >= (1L << 31) || (long)a.Cin * a.KW * a.Mp * 4 >= (1L << 31)) return hipErrorInvalidConfiguration;
  const int npw = a.Cin / 16;  // channel pairs per wave
  const long gm = (a.M + 31) / 32;
  // 64- or 32-column tiles: all blocks of these launches start together, so a launch takes about ceil(blocks / CUs)
  // block times, and a 64-column block costs two 32-column ones.  Ties go to 64 columns (half the A traffic).
  // Measured (PP16, B = 1): 504 / 256 blocks of 64 columns beat 1008 / 504 of 32 by 5-10 %, 336 (GRU input projection) and
  // 280 (first up conv) lose to 624 / 520 by 20 %.
  const long b64 = gm * ((a.Nq + 63) / 64) * a.B, b32 = gm * ((a.Nq + 31) / 32) * a.B;
  int tn = 2 * ((b64 + num_cu - 1) / num_cu) <= (b32 + num_cu - 1) / num_cu ? 2 : 1;
  // Round 6: only the 64-column tiles have the minimal-filtering form (conv_direct2w_kernel: (KW + 1) / (2 KW) of the MFMAs), so a
  // 64-column round costs 4/3 (k3) / 6/5 (k5) of a 32-column one there, not 2 -- the 512-channel k3 convs of the 401-frame level
  // at batch 8 (896 blocks of 64 columns = 4 rounds against 1 664 of 32 = 7) ran 54 us on the plain 32-column kernel where the
  // minimal-filtering form takes 40-41 (profiles/r06_d2_tile_rule_ab.txt).
  if (a.d2_tile_rule && tn == 1 && a.wino && a.direct >= 5 && a.wu && !a.fir && a.stride == 1 && a.up == 1 && (a.KW == 3 || a.KW == 5)) {
    const double c64 = 2.0 * (a.KW + 1) / (2.0 * a.KW) * (double)((b64 + num_cu - 1) / num_cu);
    if (c64 <= (double)((b32 + num_cu - 1) / num_cu)) tn = 2;
  }
  if (a.force_cfg == 105) tn = 2;
  if (a.force_cfg == 106) tn = 1;
  void (*kern)(ConvArgs) = nullptr;
  int variant = 0;
  int bm_step = 32, halo = 0;
  long gm_fir = gm;
  if (a.fir) {  // fused up-path FIR: whole output channels per tile, one halo frame either side
    if ((a.up != 2 && a.up != 3 && a.up != 4 && a.up != 5 && a.up != 8) || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.fir_len != 2 * a.up + 1 || a.add || a.film)
      return hipErrorNotSupported;
    bm_step = (32 / a.up) * a.up;
    halo = 1;
    gm_fir = (a.M + bm_step - 1) / bm_step;
    const long b62 = gm_fir * ((a.Nq + 61) / 62) * a.B, b30 = gm_fir * ((a.Nq + 29) / 30) * a.B;
    tn = 2 * ((b62 + num_cu - 1) / num_cu) <= (b30 + num_cu - 1) / num_cu ? 2 : 1;
    if (a.force_cfg == 105) tn = 2;
    if (a.force_cfg == 106) tn = 1;
    // The halo costs tiles (62 of 64 / 30 of 32 columns, whole channels only).  All blocks of these launches start
    // together -- 2 (64-column) or 4 (32-column) resident per CU -- so a launch takes ceil(blocks / slots) rounds, and
    // one more round costs more than the separate FIR pass saves (measured, PP16 B = 1: 512 -> 528 blocks at the
    // T/32 level: 10.8 + 6.3 us unfused, 17.9 us fused).  Fuse only when the round count stays.
    const long slots = (long)num_cu * (tn == 2 ? 2 : 4);
    const long fused = gm_fir * ((a.Nq + 32 * tn - 3) / (32 * tn - 2)) * a.B;
    const long plain = gm * ((a.Nq + 32 * tn - 1) / (32 * tn)) * a.B;
    if (a.force_cfg < 0 && (fused + slots - 1) / slots > (plain + slots - 1) / slots) return hipErrorNotSupported;
  }
  int wk = 8;
  const bool wide = a.stride == 1 && a.wd && a.direct >= 2 && !a.fir && a.up == 1 && (a.KW == 3 || a.KW == 5) && npw % 4 == 0 &&
                    a.pad == (a.KW - 1) / 2 && (long)a.Cin * a.Mp * 8 * 4 < (1L << 31) && (long)a.Cout * a.Tout * 4 < (1L << 31);
  if (wide) {
    // wide-load variant (taps-innermost weight copy); the reduction split over 8 or 4 waves (direct2_pick)
    bool wino = false;
    direct2_pick(a, num_cu, tn, wk, wino);
    if (wino) {
      if (wk == 8) kern = a.KW == 3 ? conv_direct2w_kernel<3, 8> : conv_direct2w_kernel<5, 8>;
      else kern = a.KW == 3 ? conv_direct2w_kernel<3, 4> : conv_direct2w_kernel<5, 4>;
      variant = 400 + 10 * wk + a.KW;  // minimal-filtering variants (64-column tiles): 483 / 485 (eight slices), 443 / 445
    } else {
      if (wk == 8)
        kern = a.KW == 3 ? (tn == 2 ? conv_direct2_kernel<3, 2, 8> : conv_direct2_kernel<3, 1, 8>)
                         : (tn == 2 ? conv_direct2_kernel<5, 2, 8> : conv_direct2_kernel<5, 1, 8>);
      else
        kern = a.KW == 3 ? (tn == 2 ? conv_direct2_kernel<3, 2, 4> : conv_direct2_kernel<3, 1, 4>)
                         : (tn == 2 ? conv_direct2_kernel<5, 2, 4> : conv_direct2_kernel<5, 1, 4>);
      variant = (wk == 8 ? 56 : 57) + 10 * tn;  // 66 / 76 (eight K slices), 67 / 77 (four)
    }
  } else if (a.stride == 1) {
    if (a.KW != 1 && a.KW != 3 && a.KW != 5) return hipErrorInvalidConfiguration;
    for (const DirectCfg& c : kDirectCfgs) {
      if (c.KW != a.KW || c.TN != tn || npw % (c.GP * 4)) continue;
      kern = c.kern;
      variant = 50 + 10 * tn + c.GP;  // 6x / 7x: stride-1 direct variants (profile records)
      break;
    }
  } else {
    // k = s = r, or 3r taps with the anti-alias FIR folded in; whole frames only
    const int R = a.stride, G = a.KW / R;
    if (a.up != 1 || a.KW != G * R || (G != 1 && G != 3) || a.pad != (G - 1) / 2 * R || a.Tin != a.Nq * R)
      return hipErrorInvalidConfiguration;
    if ((npw * G) % 4) return hipErrorInvalidConfiguration;
    tn = 2;
    for (const StridedCfg& c : kStridedCfgs) {
      if (c.R != R || c.G != G || c.TN != tn) continue;
      kern = c.kern;
      variant = 80 + 10 * (tn - 1) + G;  // 8x / 9x: strided direct variants
      break;
    }
  }
  if (!kern) return a.fir ? hipErrorNotSupported : hipErrorInvalidConfiguration;
  if (a.out_act && !wide) return hipErrorNotSupported;  // (only Direct2Epilogue -- the wide-load kernels -- has the activating form)
  ConvArgs aa = a;
  const int BN = 32 * tn;
  aa.magic_up = a.up == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)a.up) + 1u;
  aa.tile_bm = bm_step; aa.tile_bn = BN - 2 * halo; aa.tile_halo = halo;
  aa.grid_n = (a.Nq + aa.tile_bn - 1) / aa.tile_bn;
  aa.grid_m = (int)gm_fir;
  const bool is_d2 = variant == 66 || variant == 76 || variant == 67 || variant == 77 || (variant >= 400 && variant < 500);
  {
    const double xb = (double)a.Cin * a.Nq * a.stride, wb = (double)a.M * a.Cin * a.KW;
    aa.xcd_map = 0;
    if (aa.grid_m % 8 == 0 && wb >= xb) aa.xcd_map = 1;
    else if (aa.grid_n >= 8) aa.xcd_map = 2;
    if (is_d2 && aa.xcd_map == 2 && aa.grid_m % 2 == 0) {
      // 2-D ownership instead of "column tiles x mod 8 per XCD": an XCD then owns a CONTIGUOUS quarter of the time axis and
      // half of the rows.  Measured (tools/d2_sweep.py, PP16 B = 1): 19.3 -> 18.3 us (64 channels, T = 32 080), 16.5 -> 15.5 /
      // 11.2 -> 10.9 us (128 channels), no difference at 256 channels; (4 x 2) the same where it fits and a disaster where the
      // row groups do not divide (two row tiles: half the blocks are padding).  A per-XCD byte count (W/2 + X/4 against
      // W + X/8) predicts the opposite for the 128-channel level: with interleaved ownership every XCD also pulls the halo
      // lines of its neighbours' tiles, and the weights are the smaller operand there anyway.
      if (direct_grid_blocks(3, aa.grid_m, aa.grid_n) * 16 <= (long)aa.grid_m * ((aa.grid_n + 7) / 8 * 8) * 17) aa.xcd_map = 3;
    }
    if (is_d2 && a.d2_map >= 0) aa.xcd_map = a.d2_map;
    if (a.force_xcd_map >= 0) aa.xcd_map = a.force_xcd_map;
    if (aa.xcd_map == 1 && aa.grid_m % 8) aa.xcd_map = 0;
    if ((aa.xcd_map == 3 || aa.xcd_map == 4) && !is_d2) aa.xcd_map = 0;
  }
  const int nblocks = direct_grid_blocks(aa.xcd_map, aa.grid_m, aa.grid_n);
  const size_t smem = (size_t)wk * 32 * (BN + 4) * 4;
  if (cfg_out) *cfg_out = variant;
  hipLaunchKernelGGL(kern, dim3(nblocks, 1, a.B), dim3(64 * wk), smem, stream, aa);
  return hipGetLastError();
}


}  // namespace ou
