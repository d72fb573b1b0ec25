// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: LDS-tiled implicit-GEMM Conv1d (conv_mfma_kernel) and the conv dispatcher (launch_conv)
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ou {

// =========================================================================================================
// Generic Conv1d as an fp32-MFMA implicit GEMM
//   GEMM view: rows m (output channel x phase), columns q (time), reduction (ci, tap).
//   A = packed weights [chunk][tap][ci_local][Mp]  (K-major: an LDS tile row is BM consecutive floats)
//   B = activations, staged as CK contiguous rows of `span` samples (receptive-field halo included);
//       the fragment for (ci, tap) is the same LDS row shifted by `tap` -> every sample is fetched from HBM
//       once per block and re-used KW times from LDS.
//   v_mfma_f32_32x32x2_f32: A lane l = A[l&31][l>>5], B lane l = B[l>>5][l&31]; a K-pair is two adjacent
//   input channels at the same tap.  D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
//   Block = 4 waves arranged WM x WN x WK (WK = intra-block split of the reduction for the small-T levels).
//   Register-prefetched double-buffered LDS staging: one barrier per channel chunk.
// =========================================================================================================
constexpr int CONV_XCAP = 4096;  // X-tile floats per stage (SC*CK*span), spread over the block's threads
constexpr int CONV_XCAP_BIG = 8192;  // ... for the 64x64 split-K config (2x2 accumulator tiles per wave)

// Pipeline stage = SC consecutive packed chunks = SCK = SC*CK input channels.
// LDS images of a stage:
//   Xs[SCK][span]            activations incl. halo (PReLU / input scale applied while staging)
//   Ws[KW][SCK][BM]          weights, re-ordered tap-major while staging (global order is [chunk][tap][CK])
// so that for a fixed tap both MFMA operands advance by a constant stride from one channel pair to the next:
//   A(tap, I) = Ws[(tap*SCK + 2I + half)*BM + m],  B(tap, I) = Xs[(2I + half)*span + n*stride + tap]
// The k-loop is tap-outer / channel-pair-inner; the WK waves of a split-K block take pairs I = kw, kw+WK, ...
// Fragment groups of U steps are software-pipelined (reads of group g+1 issued before the MFMAs of group g).
template <int TM, int TN, int WM, int WN, int WK, int CONV_MAXW, int U, bool EXACT>
__global__ __launch_bounds__(64 * WM * WN * WK) void conv_mfma_kernel(ConvArgs p) {
  constexpr int CONV_NT = 64 * WM * WN * WK;      // 4 or 8 waves
  constexpr int XCAP = (WK == 8 && TM * TN == 4) ? CONV_XCAP_BIG : CONV_XCAP;
  constexpr int CONV_MAXX = XCAP / CONV_NT;  // X-tile floats per thread
  static_assert(WM * WN * WK == 4 || WM * WN * WK == 8, "4 or 8 waves per block");
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar step math
  const int kw = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
  // XCD-aware tile mapping: block i runs on XCD i % 8 (observed; speed only) and every XCD has its own L2, so the
  // operand shared by the blocks of different XCDs is fetched from memory once per XCD.  The launcher picks which
  // operand is owned (p.xcd_map): 1 = weight slabs (all time tiles of slab m on XCD m % 8; deep levels, W >> X),
  // 2 = time tiles (all slabs of time tile n on XCD n % 8; wide levels, X >> W), 0 = plain row-major.
  int tile_m, tile_n;
  {
    const int L = blockIdx.x, gx = p.grid_n, gy = p.grid_m;
    if (p.xcd_map == 1) {
      const int q = L >> 3;
      const int mg = q / gx;
      tile_n = q - mg * gx;
      tile_m = mg * 8 + (L & 7);
    } else if (p.xcd_map == 2) {
      const int q = L >> 3;
      const int ng = q / gy;
      tile_m = q - ng * gy;
      tile_n = ng * 8 + (L & 7);
      if (tile_n >= gx) return;  // grid padded to whole groups of 8 time tiles
    } else {
      tile_m = L / gx;
      tile_n = L - tile_m * gx;
    }
  }
  const int n0 = tile_n * BN, m0 = tile_m * BM, b = blockIdx.z;

  const int KW = p.KW, CK = p.CK, stride = p.stride, SC = p.SC;
  const int span = (BN - 1) * stride + KW;
  const int SCK = SC * CK;                // input channels per stage (power of two)
  const int lck = 31 - __clz(CK), lsck = 31 - __clz(SCK);
  const int xt = SCK * span;              // X tile elements
  const int xt_al = (xt + 3) & ~3;        // keep the W tile 16-B aligned
  const int KCs = SCK * KW;               // weight rows per stage
  const int wt4 = KCs * (BM / 4);         // W tile float4 count
  float* Xs = smem;                       // [2][xt_al]
  float* Ws = smem + 2 * xt_al;           // [2][KCs*BM]
  float* Zs = Ws + 2 * (size_t)KCs * BM;  // [2*BM] zeros: the A operand of k-steps past the end
  for (int i = tid; i < 2 * BM; i += CONV_NT) Zs[i] = 0.f;
  const int nstages = p.Cin / SCK;
  const int nI_ = SCK >> 1;               // channel pairs per stage

  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  long long tsv[8];
  const bool ts_on = p.tstamps != nullptr;
  if (ts_on) tsv[0] = __builtin_readcyclecounter();
  long long t_mma = 0, t_wait = 0;

  const float* xb = p.x + (size_t)b * p.Cin * p.Tin;
  const bool act = p.act != 0;
  const float alpha = act ? p.alpha_val : 1.0f;  // applied to every B operand read (1: identity)
  // input scale (mel front-end only): the conv is linear in its input and that layer has no PReLU prologue, so the
  // scale is applied to the accumulators in the epilogue
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;

  // Staging is direct global -> LDS (LDS-DMA buffer loads: no staging registers, no ds_write pass, the copy of stage
  // c+1 runs under the MFMAs of stage c).  The LDS destination of such a load is wave-uniform base + lane * size, so
  // the tile images are filled in thread order: element e = tid + i*NT of the X tile (one dword per lane) and float4
  // f = tid + i*NT of the W tile; WHICH global word lands there is the per-lane byte offset computed once here --
  // the per-stage part of the address is a scalar offset.  Zero padding: lanes whose sample lies outside the signal
  // never load, their LDS words are zeroed once below (the positions are the same in every stage).  The PReLU
  // prologue is applied where the B operand is read from LDS (a copy cannot transform).
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)p.Cin * (unsigned)p.Tin * 4u);
  const __amdgpu_buffer_rsrc_t rwt = make_rsrc(p.w, (unsigned)p.Cin * (unsigned)KW * (unsigned)p.Mp * 4u);
  constexpr bool PRIV_ = (WK == 8) && EXACT;  // wave-private pipeline (below): the block-wide images are not used
  int xvo[CONV_MAXX];  // byte offset inside a stage's SCK input rows, -1: zero padding / past the tile
#pragma unroll
  for (int i = 0; i < CONV_MAXX; i++) {
    int e = tid + i * CONV_NT;
    int g = -1;
    if (!PRIV_ && e < xt) {
      int l = (int)__umulhi((unsigned)e, p.magic_span[BN == 128 ? 0 : (BN == 64 ? 1 : 2)]);  // e / span
      int j = e - l * span;
      int t = n0 * stride - p.pad + j;
      if (t >= 0 && t < p.Tin) g = (l * p.Tin + t) * 4;
      else { Xs[e] = 0.f; Xs[xt_al + e] = 0.f; }
    }
    xvo[i] = g;
  }
  // ... and of its W-tile float4s: LDS row (tap, cl) <- packed row (sub*KW + tap)*CK + l,  cl = sub*CK + l
  int wvo[CONV_MAXW];
#pragma unroll
  for (int i = 0; i < CONV_MAXW; i++) {
    int f = tid + i * CONV_NT;
    int row = f / (BM / 4), c4 = f % (BM / 4);
    int tap = row >> lsck, cl = row & (SCK - 1);
    int sub = cl >> lck, l = cl & (CK - 1);
    wvo[i] = (!PRIV_ && f < wt4) ? (((sub * KW + tap) * CK + l) * p.Mp + c4 * 4 + m0) * 4 : -1;
  }
  auto dma_stage = [&](int c, int buf) {
    const int xso = c * SCK * p.Tin * 4, wso = c * KCs * p.Mp * 4;
    float* xd = Xs + buf * xt_al + wave * 64;
    float* wd = Ws + (size_t)buf * KCs * BM + wave * 256;
#pragma unroll
    for (int i = 0; i < CONV_MAXX; i++)
      if (xvo[i] >= 0) dma_b32(rx, xd + i * CONV_NT, xvo[i], xso);
#pragma unroll
    for (int i = 0; i < CONV_MAXW; i++)
      if (wvo[i] >= 0) dma_b128(rwt, wd + i * CONV_NT * 4, wvo[i], wso);
  };
  // ---- split-K configs (exact variants): wave-PRIVATE stage pipeline -------------------------------------------
  // Wave kw of a split-K block only ever reads the channel pairs I = kw, kw + WK, ... of a stage -- 1/WK of the X
  // and W tiles.  So every wave copies exactly the rows it consumes into its own slice of LDS and runs its own
  // double-buffered pipeline, ordered by its own vmcnt: no workgroup barrier in the main loop.  With a barrier per
  // stage both waves of a SIMD stop together at every stage boundary (copy issue, first LDS round trip, barrier) and
  // the MFMA pipe idles for about as long as a stage's 12-24 MFMAs keep it busy; unsynchronised, one wave's
  // boundary hides under the other's MFMAs.
  //   slice of wave kw, buffer b:  Xw[2*ppw][span] (rows: local pair i, half -> channel 2*(kw + WK*i) + half)
  //                                Ww[KW][2*ppw][BM]
  constexpr bool PRIV = (WK == 8) && EXACT;
  constexpr int PMAXX = PRIV ? XCAP / WK / 64 + 1 : 1;
  const int ppw = nI_ / WK;                       // channel pairs per wave and stage (power of two)
  const int lp2 = 31 - __clz(2 * ppw);
  const int xw = 2 * ppw * span, xw_al = (xw + 3) & ~3;
  const int ww = KW * 2 * ppw * BM;
  const int wsz = xw_al + ww;                     // floats per wave and buffer
  float* const pbase = smem + (size_t)wave * 2 * wsz;
  int pxvo[PMAXX], pwvo[PRIV ? CONV_MAXW : 1];
  int Kw = 0;                                     // copies this wave issues per stage
  if constexpr (PRIV) {
#pragma unroll
    for (int i = 0; i < PMAXX; i++) {
      const int e = lane + 64 * i;
      int g = -1;
      if (e < xw) {
        const int r = (int)__umulhi((unsigned)e, p.magic_span[BN == 128 ? 0 : (BN == 64 ? 1 : 2)]);  // e / span
        const int jx = e - r * span;
        const int cl = 2 * (kw + WK * (r >> 1)) + (r & 1);
        const int t = n0 * stride - p.pad + jx;
        g = (t >= 0 && t < p.Tin) ? (cl * p.Tin + t) * 4 : (int)0x80000000;  // past the buffer: reads as 0
      }
      pxvo[i] = g;
      Kw += (64 * i < xw) ? 1 : 0;
    }
    const int wt4p = ww / 4;
#pragma unroll
    for (int i = 0; i < CONV_MAXW; i++) {
      const int f = lane + 64 * i;
      const int R = f / (BM / 4), c4 = f % (BM / 4);
      const int tap = R >> lp2, r = R & (2 * ppw - 1);
      const int cl = 2 * (kw + WK * (r >> 1)) + (r & 1);
      const int sub = cl >> lck, l = cl & (CK - 1);
      pwvo[i] = f < wt4p ? (((sub * KW + tap) * CK + l) * p.Mp + c4 * 4 + m0) * 4 : -1;
      Kw += (64 * i < wt4p) ? 1 : 0;
    }
  }
  auto dma_private = [&](int c, int buf) {
    const int xso = c * SCK * p.Tin * 4, wso = c * KCs * p.Mp * 4;
    float* xd = pbase + buf * wsz;
    float* wd = xd + xw_al;
#pragma unroll
    for (int i = 0; i < PMAXX; i++)
      if (pxvo[i] != -1) dma_b32(rx, xd + 64 * i, pxvo[i], xso);
#pragma unroll
    for (int i = 0; i < (PRIV ? CONV_MAXW : 1); i++)
      if (pwvo[i] != -1) dma_b128(rwt, wd + 256 * i, pwvo[i], wso);
  };
  if (!(p.dbg & 1)) {
    if constexpr (PRIV) dma_private(0, 0);
    else dma_stage(0, 0);
  }

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int lhalf = lane >> 5, l31 = lane & 31;
  const int a_col = wm * (32 * TM) + l31;
  const int b_col = (wn * (32 * TN) + l31) * stride;
  const int nI = nI_;
  const int my_steps = nI > kw ? (nI - kw + WK - 1) / WK : 0;  // pairs of this wave per tap
  const int gpt = (my_steps + U - 1) / U;              // fragment groups per tap
  const int ngroups = gpt * KW;
  // operand strides between consecutive channel pairs of a wave, and between taps
  constexpr int a_step = PRIV ? 2 * BM : 2 * WK * BM;
  const int b_step = PRIV ? 2 * span : 2 * WK * span;
  const int tap_step = PRIV ? 2 * ppw * BM : SCK * BM;
  const float* zrow = Zs + lhalf * BM + a_col;

  // group cursor (tap, jg) advanced by every load_group call, in program order
  int cur_tap = 0, cur_jg = 0;
  // EXACT: every wave has a whole number of groups per tap (the launcher guarantees it) -> no guards, the A reads
  // are immediate offsets from one base register
  auto load_group = [&](const float* wsb, const float* xsb, float (&av)[U][TM], float (&bv)[U][TN]) {
    const float* wt = wsb + cur_tap * tap_step + cur_jg * (U * a_step);
    const float* xq = xsb + cur_tap + cur_jg * (U * b_step);
    const int j0 = cur_jg * U;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool ok = EXACT || (j0 + u < my_steps);
      const float* wrow = ok ? wt + u * a_step : zrow;
      const float* xrow = ok ? xq + u * b_step : xq;
#pragma unroll
      for (int i = 0; i < TM; i++) av[u][i] = wrow[32 * i];
#pragma unroll
      for (int j = 0; j < TN; j++) bv[u][j] = xrow[32 * j * stride];
    }
    if (++cur_jg == gpt) { cur_jg = 0; ++cur_tap; }
  };
  auto mma_group = [&](float (&av)[U][TM], float (&bv)[U][TN]) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      // PReLU prologue of the layer, applied to the B fragment at its point of use (the reads of this group were
      // issued a whole group of MFMAs ago, so nothing waits on LDS here)
      float bt[TN];
#pragma unroll
      for (int j = 0; j < TN; j++) bt[j] = bv[u][j] >= 0.f ? bv[u][j] : alpha * bv[u][j];
#pragma unroll
      for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bt[j], acc[i][j], 0, 0, 0);
    }
  };

  if (ts_on) tsv[1] = __builtin_readcyclecounter();
  if constexpr (!PRIV) __syncthreads();  // (waits for the stage-0 copies: an LDS-DMA in flight counts on vmcnt)
  if (ts_on) tsv[2] = __builtin_readcyclecounter();
  // Epilogue operands of the fast path (bias, FiLM, cond add, residual): fetched now, so that their latency hides
  // behind the whole main loop (kept in registers; only for tiles with <= 4 epilogue passes per thread)
  constexpr int C4e = BN / 4, RPPe = CONV_NT / C4e, NPe = (BM + RPPe - 1) / RPPe;
  constexpr bool EARLY = NPe <= 4;
  const bool fast_epi = p.up == 1 && (p.Tout & 3) == 0;
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  int m_hi = m0 + BM - 1;
  if (m_hi > p.M - 1) m_hi = p.M - 1;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  f32x4 addv[NPe], resv[NPe];
  float bi[NPe], ga[NPe], be[NPe];
  auto fetch_epi = [&]() {
    const int q = (tid % C4e) * 4, r0 = tid / C4e;
    if (n0 + q < p.Nq) {
#pragma unroll
      for (int k = 0; k < NPe; k++) {
        const int m = m0 + r0 + k * RPPe;
        const bool ok = m <= m_hi && r0 + k * RPPe < BM;
        const int mm = ok ? m : m0;
        const size_t idx = ybase + (size_t)mm * p.Tout + n0 + q;
        bi[k] = p.bias[mm];
        if (p.add) addv[k] = *reinterpret_cast<const f32x4*>(p.add + idx);
        if (p.res) resv[k] = *reinterpret_cast<const f32x4*>(p.res + idx);
        if (filmb) { ga[k] = filmb[mm]; be[k] = filmb[p.Cout + mm]; }
      }
    }
  };
  if (EARLY && fast_epi) fetch_epi();
  auto compute_from = [&](const float* wsb, const float* xsb) {
    cur_tap = 0;
    cur_jg = 0;
    // software-pipelined: the LDS reads of group g+1 are issued before the MFMAs of group g
    float a0[U][TM], b0[U][TN], a1[U][TM], b1[U][TN];
    if (ngroups > 0) load_group(wsb, xsb, a0, b0);
    for (int g = 0; g < ngroups; g += 2) {
      if (g + 1 < ngroups) load_group(wsb, xsb, a1, b1);
      mma_group(a0, b0);
      if (g + 1 < ngroups) {
        if (g + 2 < ngroups) load_group(wsb, xsb, a0, b0);
        mma_group(a1, b1);
      }
    }
  };
  if constexpr (PRIV) {
    for (int c = 0; c < nstages; c++) {
      const int buf = c & 1;
      long long ta = 0, tb = 0;
      if (ts_on) ta = __builtin_readcyclecounter();
      const bool more = c + 1 < nstages;
      if (more && !(p.dbg & 1)) dma_private(c + 1, buf ^ 1);
      wait_vmcnt(more ? Kw : 0);  // this wave's copies of stage c have landed; those of stage c+1 stay in flight
      if (ts_on) { tb = __builtin_readcyclecounter(); t_wait += tb - ta; }
      if (!(p.dbg & 2)) {
        const float* xd = pbase + buf * wsz;
        compute_from(xd + xw_al + lhalf * BM + a_col, xd + lhalf * span + b_col);
      }
      if (ts_on) t_mma += __builtin_readcyclecounter() - tb;
    }
    __syncthreads();  // the epilogue re-uses the stage buffers of all waves
  } else {
    for (int c = 0; c < nstages; c++) {
      const int buf = c & 1;
      long long ta = 0;
      if (ts_on) ta = __builtin_readcyclecounter();
      if (c + 1 < nstages && !(p.dbg & 1)) dma_stage(c + 1, buf ^ 1);
      if (!(p.dbg & 2))
        compute_from(Ws + ((size_t)buf * KCs + 2 * kw + lhalf) * BM + a_col,
                     Xs + buf * xt_al + (2 * kw + lhalf) * span + b_col);
      long long tb = 0;
      if (ts_on) { tb = __builtin_readcyclecounter(); t_mma += tb - ta; }
      __syncthreads();
      if (ts_on) t_wait += __builtin_readcyclecounter() - tb;
    }
  }
  if (ts_on) tsv[3] = __builtin_readcyclecounter();
  if (p.dbg & 4) { if (acc[0][0][0] == 12345.f) p.y[0] = 1.f; return; }

  // ---- epilogue: accumulators -> LDS (sum over the WK split on read) -> coalesced fused store ----------
  constexpr int EP = BN + 4;  // keeps rows 16-B aligned for the float4 read-back
  float* Es = smem;           // [WK][BM][EP]
#pragma unroll
  for (int i = 0; i < TM; i++)
#pragma unroll
    for (int j = 0; j < TN; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int row = wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        int col = wn * (32 * TN) + 32 * j + l31;
        Es[(kw * BM + row) * EP + col] = acc[i][j][r];
      }
  __syncthreads();
  if (ts_on) tsv[4] = __builtin_readcyclecounter();

  const int up = p.up, Cout = p.Cout, Tout = p.Tout;
  if (fast_epi) {
    // fast path: one float4 of consecutive time samples per thread and pass, shift-only indexing
    constexpr int C4 = C4e, RPP = RPPe, NP = NPe;
    const int c4 = tid % C4, q = c4 * 4, r0 = tid / C4;
    if (!EARLY) fetch_epi();
    if (n0 + q < p.Nq) {
#pragma unroll
      for (int k = 0; k < NP; k++) {
        const int row = r0 + k * RPP, m = m0 + row;
        if (m > m_hi || row >= BM) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(&Es[row * EP + q]);
#pragma unroll
        for (int kk = 1; kk < WK; kk++) v += *reinterpret_cast<const f32x4*>(&Es[(kk * BM + row) * EP + q]);
        if (p.in_scale) v *= insc;
        v += bi[k];
        if (p.add) v = (v + addv[k]) * p.add_scale;
        if (filmb) v = ga[k] * v + be[k];
        if (p.res) v = (v + resv[k]) * p.res_scale;
        if (p.lens) v = ragged_mask4(v, n0 + q, ragged_len(p.lens, (int)blockIdx.z));
        *reinterpret_cast<f32x4*>(p.y + ybase + (size_t)m * Tout + n0 + q) = v;
      }
    }
  } else {
    // general path (transposed-conv phase interleave, or rows that are not 16-B aligned):
    //   e -> (co, q, ph) with the output sample t = (n0 + q)*up + ph fastest across threads
    constexpr int LBN = (BN == 128) ? 7 : (BN == 64 ? 6 : 5);
    const int co_first = up == 1 ? m0 : (int)__umulhi((unsigned)m0, p.magic_up);
    const int nco = (up == 1 ? m_hi : (int)__umulhi((unsigned)m_hi, p.magic_up)) - co_first + 1;
    const int total = nco * BN * up;
    for (int e = tid; e < total; e += CONV_NT) {
      const int rest = up == 1 ? e : (int)__umulhi((unsigned)e, p.magic_up);  // e / up
      const int ph = e - rest * up;
      const int q = rest & (BN - 1);
      const int co = co_first + (rest >> LBN);
      const int m = co * up + ph;
      const int t = (n0 + q) * up + ph;
      if (m < m0 || m > m_hi || (n0 + q) >= p.Nq || t >= Tout) continue;
      float v = Es[(m - m0) * EP + q];
#pragma unroll
      for (int k = 1; k < WK; k++) v += Es[(k * BM + (m - m0)) * EP + q];
      if (p.in_scale) v *= insc;
      v += p.bias[co];
      const size_t idx = ybase + (size_t)co * Tout + t;
      if (p.add) v = (v + p.add[idx]) * p.add_scale;
      if (filmb) v = filmb[co] * v + filmb[Cout + co];
      if (p.res) v = (v + p.res[idx]) * p.res_scale;
      if (p.lens && t >= ragged_len(p.lens, (int)blockIdx.z)) v = 0.f;
      p.y[idx] = v;
    }
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
  if (ts_on && lane == 0) {
    long long* o = p.tstamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * (CONV_NT / 64) + wave) * 8;
    o[0] = tsv[1] - tsv[0]; o[1] = tsv[2] - tsv[1]; o[2] = tsv[3] - tsv[2]; o[3] = tsv[4] - tsv[3];
    o[4] = __builtin_readcyclecounter() - tsv[4]; o[5] = t_mma; o[6] = t_wait; o[7] = tsv[0];
  }
}

struct ConvCfg {
  int BM, BN, WK, MAXW, NT, XCAP;
  void (*kern4)(ConvArgs);  // fragment groups of 4 k-steps, exact
  void (*kern2)(ConvArgs);  // ... of 2, exact (few channel pairs per wave and tap)
  void (*kern_g)(ConvArgs); // groups of 2 with guards (odd pair counts: tiny test models only)
};
#define OU_CONV_CFG(BM, BN, WK, MAXW, NT, TM, TN, WM, WN)                                             \
  {BM, BN, WK, MAXW, NT, (WK == 8 && TM * TN == 4) ? CONV_XCAP_BIG : CONV_XCAP,                       \
   conv_mfma_kernel<TM, TN, WM, WN, WK, MAXW, 4, true>,                          \
   conv_mfma_kernel<TM, TN, WM, WN, WK, MAXW, 2, true>, conv_mfma_kernel<TM, TN, WM, WN, WK, MAXW, 2, false>}
// a table slot whose kernels are not in this build (the slot keeps its index: profile records and tools name configs by it)
#define OU_CONV_CFG_OFF(BM, BN, WK, MAXW, NT) {BM, BN, WK, MAXW, NT, CONV_XCAP, nullptr, nullptr, nullptr}
static const ConvCfg kConvCfgs[] = {
    OU_CONV_CFG(64, 128, 1, 6, 256, 1, 2, 2, 2),
    OU_CONV_CFG(32, 128, 1, 6, 256, 1, 1, 1, 4),
    OU_CONV_CFG(64, 64, 1, 6, 256, 1, 1, 2, 2),
    // small-T levels: reduction split over 4 waves, up to 4 packed chunks per pipeline stage -- never chosen by launch_conv
    // (superseded by the 8-wave variants below; tools/conv_sweep.py can still force them in an EXPERIMENTS build)
#ifdef OU_EXPERIMENTS
    OU_CONV_CFG(32, 64, 4, 12, 256, 1, 2, 1, 1),
    OU_CONV_CFG(32, 32, 4, 12, 256, 1, 1, 1, 1),
#else
    OU_CONV_CFG_OFF(32, 64, 4, 12, 256), OU_CONV_CFG_OFF(32, 32, 4, 12, 256),
#endif
    // 8 waves (two per SIMD), reduction split 8 ways
    OU_CONV_CFG(32, 64, 8, 6, 512, 1, 2, 1, 1),
    OU_CONV_CFG(32, 32, 8, 6, 512, 1, 1, 1, 1),
    // 64x64, reduction split 8 ways, 2x2 accumulator tiles per wave: one LDS read per MFMA (tuning only: needs a cross-CU split-K)
#ifdef OU_EXPERIMENTS
    OU_CONV_CFG(64, 64, 8, 12, 512, 2, 2, 1, 1),
#else
    OU_CONV_CFG_OFF(64, 64, 8, 12, 512),
#endif
};
constexpr int kNumConvCfgs = sizeof(kConvCfgs) / sizeof(kConvCfgs[0]);

static size_t conv_smem_bytes(const ConvCfg& c, const ConvArgs& a) {
  int span = (c.BN - 1) * a.stride + a.KW;
  size_t xt_al = ((size_t)a.SC * a.CK * span + 3) & ~size_t(3);
  size_t stage = 2 * (xt_al + (size_t)a.SC * a.CK * a.KW * c.BM) + 2 * c.BM;
  if (c.WK == 8) stage += 2 * 4 * c.WK;  // wave-private slices: per-wave 16-B alignment of the X image
  size_t epi = (size_t)c.WK * c.BM * (c.BN + 4);
  return 4 * (stage > epi ? stage : epi);
}

hipError_t init_conv_kernels() {
  hipError_t e_d3 = hipSuccess;
  for (int i = 0; i < kNumConvCfgs; i++) {
    if (!kConvCfgs[i].kern4) continue;  // not in this build
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kConvCfgs[i].kern4),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kConvCfgs[i].kern2),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kConvCfgs[i].kern_g),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  e_d3 = init_direct3_kernels();
  if (e_d3 != hipSuccess) return e_d3;
  e_d3 = init_block3_kernels();
  if (e_d3 != hipSuccess) return e_d3;
  e_d3 = init_direct4_kernels();
  if (e_d3 != hipSuccess) return e_d3;
  e_d3 = init_split_kernels();
  if (e_d3 != hipSuccess) return e_d3;
  return init_chain_kernels();
}

// Every conv family but the fused ConvBlock bodies (variants 100 .. 199: their conv1 / conv2 tiles live in LDS, where nothing
// zeroes them behind a row's end -- the runner does not fuse ragged batches) keeps ConvArgs::lens in its epilogue.
bool conv_masks_rows(int cfg) { return cfg >= 0 && !(cfg >= 100 && cfg < 200); }

// ---- the kernel families of a Conv1d launch, in the order they are asked ------------------------------------------------------
// Each `try_*` answers for ONE family: hipSuccess / a launch error = it took the layer; hipErrorInvalidConfiguration = not a
// layer for this family (ask the next one); hipErrorNotSupported = the layer would be its, but the requested epilogue (fused
// FIR, activating store) is not -- the caller's fallback.  `force_cfg` (ou_bench_conv, tests) names a variant code; a family
// is asked only when the code lies in its range [lo, hi), and then its answer is final.
namespace {
// conv_split_kernel (8xx / 9xx): stride-1 k3 / k5 layers with enough work per launch to feed the BF16 matrix pipe.  Rule from
// tools/ubench/split_conv.hip against the per-layer tables of the fp32 kernels and from A / B runs of the product (profiles/
// r05_split_*, r05_late_*; re-derived by forcing in round 6, r06_ab_split_rule_*): rows a multiple of 256 (four waves stacked
// along the rows, each weight fragment fetched once per block) and a whole device's worth of 256 x 128 output blocks, half a
// device's worth for the k5 layers and on short rows (the 401-frame level, on the split-K fp32 kernels otherwise) -- PP16 /
// OR16: the 256- and 512-channel levels from B = 16, the 256-channel k5 convs from B = 8.
hipError_t try_split(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (!a.wsplit || a.split == 0) return hipErrorInvalidConfiguration;
  const double tiles = (double)(a.M / 64) * ((a.Nq + 127) / 128) * a.B / 4.0;  // 256 x 128 blocks' worth of output
  const bool rule = a.M >= 256 && a.M % 256 == 0 && tiles >= ((a.Nq < 1024 || a.KW == 5) ? 0.45 : 0.9) * num_cu;
  if (!(a.split == 1 || a.force_cfg >= 800 || rule)) return hipErrorInvalidConfiguration;
  return launch_conv_split(a, num_cu, stream, cfg_out);
}
// conv_direct3(w)(s)_kernel (2xx / 5xx): the no-split-K throughput kernels for launches with many columns
hipError_t try_direct3(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (a.direct < 3) return hipErrorInvalidConfiguration;
  return launch_conv_direct3(a, num_cu, stream, cfg_out);
}
// conv_direct4_kernel (3xx): 1x1 convs, phase GEMMs and k = s = r rate-change convs with too few columns for the family above.
// It has no fused up-path FIR: a layer it would take runs as conv + FIR pass (the caller's fallback on hipErrorNotSupported)
// unless d4_fir_unfused is off.
hipError_t try_direct4(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (a.direct < 4) return hipErrorInvalidConfiguration;
  if (a.fir) {
    if (a.d4_fir_unfused && a.force_cfg < 0) {
      ConvArgs probe = a;
      probe.fir = nullptr; probe.fir_len = 0;
      if (launch_conv_direct4(probe, num_cu, stream, nullptr, true) == hipSuccess) return hipErrorNotSupported;
    }
    return a.force_cfg >= 0 ? hipErrorNotSupported : hipErrorInvalidConfiguration;  // (forced onto a family without the FIR epilogue)
  }
  return launch_conv_direct4(a, num_cu, stream, cfg_out, false);
}
// conv_direct4w_kernel (6xx / 7xx): stride-1 k3 / k5 layers with few columns (the 401-frame levels at batch 1), minimal filtering
hipError_t try_direct4w(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (a.direct < 5) return hipErrorInvalidConfiguration;
  return launch_conv_direct4w(a, num_cu, stream, cfg_out);
}
// conv_direct_kernel / conv_direct2(w)_kernel / conv_direct_strided_kernel (1xx as a request; 50 .. 99 and 4xx as answers): the
// split-K kernels of the deep levels.  Up to deep_factor blocks of 64 x 128 per CU (measured: PP16 B = 8 33.3 -> 32.3 ms, OR16
// B = 16 63.4 -> 60.2 ms when the limit goes from 3 to 6-12; beyond that nothing moves); longer rows only for the wide-load
// variant (a 64-channel k5 conv at T = 32 080 runs 19 vs 24 us on it).
hipError_t try_direct(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (a.direct == 0) return hipErrorInvalidConfiguration;
  const long wide = (long)((a.M + 63) / 64) * ((a.Nq + 127) / 128) * a.B;
  const bool wide_ok = a.wd && a.direct >= 2 && a.stride == 1 && a.up == 1;
  const bool deep = ((a.Nq <= 16384 || wide_ok) && wide < (long)a.deep_factor * num_cu) || a.force_cfg >= 100;
  if (!deep) return hipErrorInvalidConfiguration;
  return launch_conv_direct(a, num_cu, stream, cfg_out);
}
struct ConvFamily {
  int lo, hi;          // force_cfg codes that name this family
  bool final_if_forced;  // a forced request ends with this family's answer (the first-generation family falls through to the
                         // LDS kernel even when forced: force_cfg 100 .. 199 also covers layers only that kernel takes)
  hipError_t (*attempt)(const ConvArgs&, int, hipStream_t, int*);
};
const ConvFamily kConvFamilies[] = {
    {800, 1100, true, try_split},
    {200, 300, true, try_direct3},   // (and 500 .. 599: see forced_into)
    {300, 500, true, try_direct4},
    {600, 800, true, try_direct4w},
    {100, 200, false, try_direct},
};
bool forced_into(const ConvFamily& f, int cfg) {
  if (cfg >= f.lo && cfg < f.hi) return true;
  return f.attempt == try_direct3 && cfg >= 500 && cfg < 600;
}
}  // namespace

hipError_t launch_conv(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (a.Cin % a.CK || a.CK < 2 || (a.CK & (a.CK - 1)) || a.Mp % 64 || a.Nq <= 0) return hipErrorInvalidValue;
  // Deep levels and the throughput regime: the register-direct families, most specific first.  What none of them takes (wide
  // levels at small batch, layers without a direct form) stays on the LDS-tiled configs below.
  for (const ConvFamily& f : kConvFamilies) {
    const bool forced = a.force_cfg >= 0 && forced_into(f, a.force_cfg);
    if (a.force_cfg >= 0 && !forced) continue;
    const hipError_t e = f.attempt(a, num_cu, stream, cfg_out);
    if (e != hipErrorInvalidConfiguration) return e;
    if (forced && f.final_if_forced) return e;
  }
  if (a.fir) return hipErrorNotSupported;  // only the direct kernel has the fused FIR epilogue
  int pick = -1;
  for (int i = 0; i < kNumConvCfgs; i++) {
    const ConvCfg& c = kConvCfgs[i];
    if (a.force_cfg >= 0 && i != a.force_cfg) continue;
    if (!c.kern4) continue;  // (an EXPERIMENTS-only config)
    if (c.BM == 64 && a.M <= 32) continue;
    int span = (c.BN - 1) * a.stride + a.KW;
    if ((long)a.CK * span > c.XCAP) continue;
    if ((long)a.CK * a.KW * c.BM > (long)c.MAXW * c.NT * 4) continue;
    if (a.force_cfg < 0 && c.WK == 4) continue;
    if (a.force_cfg < 0 && c.BM == 64 && c.WK == 8) continue;  // tuning only (ou_bench_conv): needs a cross-CU split-K
    // latent-level k3 / k5 layers (a few hundred frames, K in the thousands): the barrier-free split-K pipelines
    // beat the one-tile-per-wave configs even when the batch supplies enough blocks (B = 8: 134 -> 114 us)
    if (a.force_cfg < 0 && c.WK == 1 && a.KW > 1 && a.Nq < 1024 && a.Cin * a.KW >= 1024) continue;
    // strided convs stage `stride` input samples per output column: with fewer than two blocks per CU the 64-column
    // tile loses to the 32-column one (enc2 rate-change conv: 22 -> 16 us)
    if (a.force_cfg < 0 && c.WK == 8 && c.BN == 64 && a.stride >= 4 &&
        (long)((a.M + c.BM - 1) / c.BM) * ((a.Nq + c.BN - 1) / c.BN) * a.B < 2L * num_cu)
      continue;  // superseded by the 8-wave split-K variants (tools/conv_sweep.py)
    if (a.force_cfg < 0 && c.WK == 8 && c.BN == 64 && a.KW == 1 && a.Nq < 1024) continue;  // 1x1, tiny T: 32x32 wins
    pick = i;
    // measured on MI355X (tools/conv_sweep.py): the one-tile-per-wave configs want >= 1.5 blocks per CU before
    // they beat the next smaller tile; the 32x64 split-K config is still ahead of 32x32 at one block per CU
    const long want = c.WK == 1 ? (long)num_cu * 3 / 2 : (long)num_cu * 15 / 16;
    long blocks = (long)((a.M + c.BM - 1) / c.BM) * ((a.Nq + c.BN - 1) / c.BN) * a.B;
    if (blocks >= want) break;
  }
  if (pick < 0) return hipErrorInvalidConfiguration;
  if (a.out_act) return hipErrorNotSupported;  // (the LDS-tiled kernels have no activating epilogue)
  const ConvCfg& c = kConvCfgs[pick];
  if (cfg_out) *cfg_out = pick;
  ConvArgs aa = a;
  {  // chunks per pipeline stage: as many as the per-thread staging registers and 128 KB of LDS allow
    const int span = (c.BN - 1) * a.stride + a.KW;
    const int nch = a.Cin / a.CK;
    int sc = 1;
    for (int cand = 4; cand >= 2; cand >>= 1) {
      if (a.force_sc > 0 && cand > a.force_sc) continue;
      if (nch % cand) continue;
      if ((long)cand * a.CK * span > c.XCAP) continue;
      if ((long)cand * a.CK * a.KW * c.BM > (long)c.MAXW * c.NT * 4) continue;
      aa.SC = cand;
      if (conv_smem_bytes(c, aa) > 160 * 1024) continue;
      sc = cand;
      break;
    }
    aa.SC = sc;
  }
  // exact for the index ranges used (e < 2^13, divisor < 2^11): floor(e/d) == umulhi(e, 2^32/d + 1)
  const int bns[3] = {128, 64, 32};
  for (int i = 0; i < 3; i++) aa.magic_span[i] = (unsigned)(0x100000000ull / (unsigned)((bns[i] - 1) * a.stride + a.KW)) + 1u;
  aa.magic_up = a.up == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)a.up) + 1u;
  aa.grid_n = (a.Nq + c.BN - 1) / c.BN;
  aa.grid_m = (a.M + c.BM - 1) / c.BM;
  {  // which operand an XCD's L2 owns: memory-side bytes ~ 8 X + W (slabs) vs X + 8 W (time tiles)
    const double xb = (double)a.Cin * a.Nq * a.stride, wb = (double)a.M * a.Cin * a.KW;
    aa.xcd_map = 0;
    if (aa.grid_m % 8 == 0 && wb >= xb) aa.xcd_map = 1;
    else if (aa.grid_n >= 8) aa.xcd_map = 2;
    if (a.force_xcd_map >= 0) aa.xcd_map = a.force_xcd_map;
    if (aa.xcd_map == 1 && aa.grid_m % 8) aa.xcd_map = 0;
  }
  const int gn_pad = aa.xcd_map == 2 ? (aa.grid_n + 7) / 8 * 8 : aa.grid_n;
  dim3 grid(gn_pad * aa.grid_m, 1, a.B);
  size_t smem = conv_smem_bytes(c, aa);
  // channel pairs per wave and tap: exact groups of 4 or 2 when every wave gets the same whole number of them
  const int pairs = aa.SC * a.CK / 2;
  const bool even = pairs % c.WK == 0;
  const int per_wave = pairs / c.WK;
  auto kern = (even && per_wave % 4 == 0) ? c.kern4 : ((even && per_wave % 2 == 0) ? c.kern2 : c.kern_g);
  hipLaunchKernelGGL(kern, grid, dim3(c.NT), smem, stream, aa);
  return hipGetLastError();
}


}  // namespace ou
