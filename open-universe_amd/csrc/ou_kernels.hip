// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path.  No portability layer: wave = 64,
// fp32 MFMA (v_mfma_f32_32x32x2_f32), DPP row reductions, agent-scope granule hand-offs.
#include "ou_kernels.h"

#include <cmath>
#include <type_traits>

namespace ou {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// a float4 at any 4-byte boundary: global dwordx4 accesses need dword alignment only (rows of 401 / 2005 frames)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// global -> LDS copies (LDS-DMA): the destination is wave-uniform `lds` + lane * size.  Kept in non-template device
// functions: the generic -> LDS address-space cast must not be instantiated on the host side of a kernel template.
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma_b32(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void dma_b128(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, soff, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): vmcnt = bits [3:0] | [15:14],
// expcnt / lgkmcnt left at their maxima.
#define OU_VMCNT_CASE(n) case n: __builtin_amdgcn_s_waitcnt(((n) & 0xF) | (((n) >> 4) << 14) | 0x0F70); break;
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    OU_VMCNT_CASE(1) OU_VMCNT_CASE(2) OU_VMCNT_CASE(3) OU_VMCNT_CASE(4) OU_VMCNT_CASE(5) OU_VMCNT_CASE(6)
    OU_VMCNT_CASE(7) OU_VMCNT_CASE(8) OU_VMCNT_CASE(9) OU_VMCNT_CASE(10) OU_VMCNT_CASE(11) OU_VMCNT_CASE(12)
    OU_VMCNT_CASE(13) OU_VMCNT_CASE(14) OU_VMCNT_CASE(15) OU_VMCNT_CASE(16)
    default: __builtin_amdgcn_s_waitcnt(0x0F70); break;  // 0, or out of table: wait for everything (always safe)
  }
}

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
static const ConvCfg kConvCfgs[] = {
    OU_CONV_CFG(64, 128, 1, 6, 256, 1, 2, 2, 2),
    OU_CONV_CFG(32, 128, 1, 6, 256, 1, 1, 1, 4),
    OU_CONV_CFG(64, 64, 1, 6, 256, 1, 1, 2, 2),
    // small-T levels: reduction split over the waves, up to 4 packed chunks per pipeline stage
    OU_CONV_CFG(32, 64, 4, 12, 256, 1, 2, 1, 1),
    OU_CONV_CFG(32, 32, 4, 12, 256, 1, 1, 1, 1),
    // 8 waves (two per SIMD), reduction split 8 ways
    OU_CONV_CFG(32, 64, 8, 6, 512, 1, 2, 1, 1),
    OU_CONV_CFG(32, 32, 8, 6, 512, 1, 1, 1, 1),
    // 64x64, reduction split 8 ways, 2x2 accumulator tiles per wave: one LDS read per MFMA
    OU_CONV_CFG(64, 64, 8, 12, 512, 2, 2, 1, 1),
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

static hipError_t init_chain_kernels();
static hipError_t init_direct3_kernels();
static hipError_t init_block3_kernels();
hipError_t init_conv_kernels() {
  hipError_t e_d3 = hipSuccess;
  for (int i = 0; i < kNumConvCfgs; i++) {
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
  return init_chain_kernels();
}

static hipError_t launch_conv_direct(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out);
static hipError_t launch_conv_direct3(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out);
hipError_t launch_conv(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (a.Cin % a.CK || a.CK < 2 || (a.CK & (a.CK - 1)) || a.Mp % 64 || a.Nq <= 0) return hipErrorInvalidValue;
  // Deep levels (what the 8-wave split-K configs below were built for): the register-direct kernels.  Wide levels
  // (many blocks of 64 x 128 per CU without splitting K) stay on the LDS-tiled configs.
  if (a.direct >= 3 && (a.force_cfg < 0 || a.force_cfg >= 200)) {
    hipError_t e = launch_conv_direct3(a, num_cu, stream, cfg_out);
    if (e != hipErrorInvalidConfiguration) return e;
    if (a.force_cfg >= 200) return e;
  }
  if (a.direct != 0 && (a.force_cfg < 0 || (a.force_cfg >= 100 && a.force_cfg < 200))) {
    const long wide = (long)((a.M + 63) / 64) * ((a.Nq + 127) / 128) * a.B;
    // (longer rows only for the wide-load variant: a 64-channel k5 conv at T = 32 080 runs 19 vs 24 us on it)
    const bool wide_ok = a.wd && a.direct >= 2 && a.stride == 1 && a.up == 1;
    // up to 8 blocks of 64 x 128 per CU (measured: PP16 B = 8 33.3 -> 32.3 ms, OR16 B = 16 63.4 -> 60.2 ms when the limit
    // goes from 3 to 6-12; beyond that nothing moves: those layers are not direct-capable anyway)
    static const long deep_factor = [] { const char* e = getenv("OU_DEEP_FACTOR"); return e ? atol(e) : 8L; }();
    const bool deep = ((a.Nq <= 16384 || wide_ok) && wide < deep_factor * num_cu) || a.force_cfg >= 100;
    if (deep) {
      hipError_t e = launch_conv_direct(a, num_cu, stream, cfg_out);
      if (e != hipErrorInvalidConfiguration) return e;
    }
  }
  if (a.fir) return hipErrorNotSupported;  // only the direct kernel has the fused FIR epilogue
  int pick = -1;
  for (int i = 0; i < kNumConvCfgs; i++) {
    const ConvCfg& c = kConvCfgs[i];
    if (a.force_cfg >= 0 && i != a.force_cfg) continue;
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
// Block -> output tile of the direct kernels (same XCD-aware mappings as conv_mfma_kernel).  false: padding block.
__device__ __forceinline__ bool direct_tile(const ConvArgs& p, int& tile_m, int& tile_n) {
  const int L = blockIdx.x, gx = p.grid_n, gy = p.grid_m;
  if (p.xcd_map == 1) {
    const int q = L >> 3, mg = q / gx;
    tile_n = q - mg * gx;
    tile_m = mg * 8 + (L & 7);
  } else if (p.xcd_map == 2) {
    const int q = L >> 3, ng = q / gy;
    tile_m = q - ng * gy;
    tile_n = ng * 8 + (L & 7);
    if (tile_n >= gx) return false;
  } else {
    tile_m = L / gx;
    tile_n = L - tile_m * gx;
  }
  return true;
}
__device__ __forceinline__ u32x4 direct_desc(const void* base, unsigned bytes) {
  // buffer descriptor as a plain SGPR quad (base, bounds, raw-dword format) for the inline-asm loads
  const unsigned long long a = (unsigned long long)base;
  u32x4 d;
  d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  d.z = __builtin_amdgcn_readfirstlane(bytes);
  d.w = 0x00020000u;
  return d;
}

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
template <int KW, int TN>
__device__ __forceinline__ void direct2_tile(const ConvArgs& p, float* smem, int b, int m0, int n0, int c_lo, int c_hi) {
  constexpr int D = 4, W = KW + TN - 1, KWP = KW == 3 ? 4 : 8, PAD = (KW - 1) / 2;
  constexpr int B2 = W - 4;                    // elements in the second B load: 0 (none), 1 (dword), 2 (dwordx2)
  constexpr int A2 = KW - 4 > 0 ? KW - 4 : 0;  // elements in the second A load: 0 / 1
  constexpr int LPS = 1 + (A2 ? 1 : 0) + 1 + (B2 > 0 ? 1 : 0);  // load instructions per ring slot (= channel pair)
  static_assert(KW == 3 || KW == 5, "k3 / k5");
  static_assert(B2 >= -1 && B2 <= 2 && D * LPS <= 60, "window / vmcnt");
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

  const int NG = p.Cin >> 4;  // channel pairs per wave (launcher: a multiple of D)
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
    const int ci = 2 * (kw + 8 * (g_));                                                                              \
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
#define OU_MMA(d, out)                                                                                               \
  {                                                                                                                  \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((out) * LPS));                                                          \
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
  DirectEpilogue<TN, true>::run(p, acc, smem, tid, kw, b, m0, n0, c_lo, c_hi);
  if (ts_on && lane == 0) {
    const long long c5 = __builtin_readcyclecounter();
    long long* o = p.tstamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + kw) * 8;
    o[0] = r0; o[1] = c1 - c0; o[2] = (NR > 1 ? c2 : c4) - c1; o[3] = NR > 1 ? c3 - c2 : 0; o[4] = c4 - c3; o[5] = c5 - c4;
    o[6] = 0; o[7] = (long long)__builtin_amdgcn_s_memrealtime();
  }
}
template <int KW, int TN>
__global__ __launch_bounds__(512) void conv_direct2_kernel(ConvArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int tile_m, tile_n;
  if (!direct_tile(p, tile_m, tile_n)) return;
  if (p.prof && threadIdx.x == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  direct2_tile<KW, TN>(p, smem, blockIdx.z, tile_m * 32, tile_n * 32 * TN, 0, 0x7fffffff);
  if (p.prof && threadIdx.x == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

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
// The placement is checked, not assumed: every workgroup adds its XCC id to its group's mask, a group that spans XCDs
// raises status bit 32 (the host falls back to separate launches for good); the barrier itself uses agent-scope atomics and
// is correct under any placement; spins are bounded (status bit 16).
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
static hipError_t init_block3_kernels() {
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
static hipError_t init_direct3_kernels() {
  for (const Direct3Cfg& c : kDirect3Cfgs) {
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
  ConvArgs aa = a;
  aa.grid_m = (int)gy;
  aa.magic_up = a.up == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)a.up) + 1u;
  const long chunks = (ct + 3) / 4, total8 = (chunks * a.B + 7) / 8 * 8;
  aa.grid_n = (int)chunks;
  if (cfg_out) *cfg_out = 260 + R;
  hipLaunchKernelGGL(kern, dim3((unsigned)(total8 * gy)), dim3(256), 0, stream, aa);
  return hipGetLastError();
}
static hipError_t launch_conv_direct3(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
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
  // chunks; the split-K kernels keep them whatever the batch (B = 8: 54 vs 107 us on the latent k3 convs)
  if (tile_min > 0 && a.Nq < 1024 && a.force_cfg < 200) return hipErrorInvalidConfiguration;
  int tm = direct3_tm(a.M);
  const long ct = (a.Nq + 63) / 64;
  // 32-row tiles where the preferred ones leave fewer than ~3 wave tiles per SIMD (and M tiles by 32): twice the waves, half the
  // registers (4 waves per SIMD instead of 2), for 4 instead of 6 loads per 24 instead of 48 MFMAs.  Measured (tile_sweep):
  // PP24 C = 384 at B = 8 (2.4 -> 4.8 tiles per SIMD) 355 / 222 -> 305 / 190 us, PP16 C = 64 at B = 8 (3.9 -> 7.8) 109 / 68 -> 104 / 64;
  // even at 5.9 tiles per SIMD (PP24 C = 192) the two are equal.
  if (tm > 2 && a.M % 32 == 0 && (double)((a.M + 16 * tm - 1) / (16 * tm)) * ct * a.B / (4.0 * num_cu) < 3.0) tm = 2;
  if (a.force_cfg >= 200) tm = (a.force_cfg / 10) % 10;
  if (tm < 2 || tm > 4) return hipErrorInvalidConfiguration;
  const long gy = (a.M + 16 * tm - 1) / (16 * tm);
  const double per_simd = (double)gy * ct * a.B / (4.0 * num_cu);
  if (a.force_cfg < 200 && per_simd < tile_min) return hipErrorInvalidConfiguration;
  // 4 waves x 4 TM KB of LDS for the prefetched epilogue operand (OU_TILE_PREFETCH=0 switches it off)
  const bool prefetch = (a.res || a.add) && (a.Tout & 3) == 0 && a.tile_prefetch != 0;
  void (*kern)(ConvArgs) = nullptr;
  for (const Direct3Cfg& c : kDirect3Cfgs)
    if (c.KW == a.KW && c.TM == tm) { kern = prefetch ? c.kern_pre : c.kern; break; }
  if (!kern) return hipErrorInvalidConfiguration;
  ConvArgs aa = a;
  aa.grid_m = (int)gy;
  const long chunks = (ct + 3) / 4, total8 = (chunks * a.B + 7) / 8 * 8;
  aa.grid_n = (int)chunks;
  if (cfg_out) *cfg_out = 200 + 10 * tm + a.KW;
  const size_t smem = prefetch ? (size_t)4 * 4 * tm * 1024 : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)(total8 * gy)), dim3(256), smem, stream, aa);
  return hipGetLastError();
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

// Launches a direct kernel when the layer fits one; hipErrorInvalidConfiguration = "use conv_mfma_kernel".
static hipError_t launch_conv_direct(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
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
  if (a.stride == 1 && a.wd && a.direct >= 2 && !a.fir && a.up == 1 && (a.KW == 3 || a.KW == 5) && npw % 4 == 0 &&
      a.pad == (a.KW - 1) / 2 && (long)a.Cin * a.Mp * 8 * 4 < (1L << 31)) {
    // wide-load variant (taps-innermost weight copy)
    kern = a.KW == 3 ? (tn == 2 ? conv_direct2_kernel<3, 2> : conv_direct2_kernel<3, 1>)
                     : (tn == 2 ? conv_direct2_kernel<5, 2> : conv_direct2_kernel<5, 1>);
    variant = 56 + 10 * tn;  // 66 / 76
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
  ConvArgs aa = a;
  const int BN = 32 * tn;
  aa.magic_up = a.up == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)a.up) + 1u;
  aa.tile_bm = bm_step; aa.tile_bn = BN - 2 * halo; aa.tile_halo = halo;
  aa.grid_n = (a.Nq + aa.tile_bn - 1) / aa.tile_bn;
  aa.grid_m = (int)gm_fir;
  {
    const double xb = (double)a.Cin * a.Nq * a.stride, wb = (double)a.M * a.Cin * a.KW;
    aa.xcd_map = 0;
    if (aa.grid_m % 8 == 0 && wb >= xb) aa.xcd_map = 1;
    else if (aa.grid_n >= 8) aa.xcd_map = 2;
    if (a.force_xcd_map >= 0) aa.xcd_map = a.force_xcd_map;
    if (aa.xcd_map == 1 && aa.grid_m % 8) aa.xcd_map = 0;
  }
  const int gn_pad = aa.xcd_map == 2 ? (aa.grid_n + 7) / 8 * 8 : aa.grid_n;
  const size_t smem = (size_t)8 * 32 * (BN + 4) * 4;
  if (cfg_out) *cfg_out = variant;
  hipLaunchKernelGGL(kern, dim3(gn_pad * aa.grid_m, 1, a.B), dim3(512), smem, stream, aa);
  return hipGetLastError();
}

// =========================================================================================================
// Fused ConvBlock body for the wide, shallow levels (C = 32 / 64 channels, tens of thousands of samples)
//   conv1 (k5) -> (+cond)/sqrt2 -> FiLM -> conv2 (k3) -> conv3 (k3) -> (+h)/sqrt2        (blocks.py:377-399)
// As separate launches each of these is one short pipeline stage per block: load everything, ~80 MFMAs per wave,
// store everything, with the intermediate (B, C, T) tensors going out to L2/HBM and back.  Here a block owns TN output
// samples and walks the whole chain on LDS-resident tiles: stage s computes NC = 32 * (waves / MT) columns
// (time t0 - R_s + u, R_s = halo still needed downstream), masks columns outside [0, T) to the zero padding the
// next conv must see, applies that conv's PReLU and leaves the tile in LDS; only the last stage writes HBM.
// The halo (2 + 1 + 1 samples each side for depth 3) is recomputed per tile: TN = NC - 2 R_0.
//   * weights are streamed: one packed chunk ([KW][CK][C], <= 96 rows) per pipeline slot, register-prefetched one
//     slot ahead into a 2-slot LDS ring, one barrier per slot -- the slot sequence runs straight through the conv
//     boundaries, so the next conv's first chunk is already in flight while the previous epilogue runs.
//   * every wave owns one 32 x 32 output tile of every stage (no split-K, no cross-wave reduction); the k-loop is
//     the generic kernel's: tap-outer, channel-pair-inner, fragment groups of 4 software-pipelined.
//   * depth 2 (conv2, conv3 only; conv1 stays a generic launch) exists because of tile quantisation at B = 1:
//     T = 32000 over 256 CUs is 125 samples per CU -- 126-sample tiles fit one round, 124-sample tiles do not.
// =========================================================================================================
// Addressing: every global access is a buffer instruction -- the per-lane byte offset is computed once, the row
// (channel) part of the address is a wave-uniform SGPR offset -- so the prologue / epilogues spend their VALU
// cycles on the arithmetic only (they are a third of a block's time at C = 32).
template <int MT, int NWV>
__global__ __launch_bounds__(64 * NWV) void conv_chain_kernel(ChainArgs p, int TN, int ntiles) {
  constexpr int C = 32 * MT, NTN = NWV / MT, NC = 32 * NTN, XS = NC + 4, NTH = 64 * NWV;
  constexpr int WSLOT = 96 * C;  // floats per weight slot (KW * CK <= 96 rows of C)
  constexpr int MAXW = (WSLOT / 4 + NTH - 1) / NTH;
  constexpr int RPW = C / NWV, NCG = NC / 64;  // input-tile rows per wave, full 64-column groups per row
  constexpr int U = 4;
  static_assert(NC % 64 == 0 && RPW * 4 <= 64 && C % NWV == 0, "input tile mapping");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bufA = smem;            // [C][XS]
  float* bufB = bufA + C * XS;   // [C][XS]
  float* Wb = bufB + C * XS;     // [2][WSLOT]
  float* prm = Wb + 2 * WSLOT;   // bias[3][C], gamma[C], beta[C]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % MT, wn = wave / MT;
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int tile = blockIdx.x % ntiles, b = blockIdx.x / ntiles;
  const int D = p.depth, T = p.T;
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const bool ts_on = p.tstamps != nullptr;
  long long tsv[3] = {0, 0, 0}, t_mma = 0, t_epi = 0, t_bar = 0, t_iss = 0, t_ws = 0;
  if (ts_on) tsv[0] = __builtin_readcyclecounter();

  // halo still needed downstream of stage s (the conv index is a compile-time constant everywhere below: a
  // run-time index into p.cv[] would be re-read from the kernarg segment at every use)
  const int h1 = (p.cv[1].KW - 1) / 2, h2 = D == 3 ? (p.cv[2].KW - 1) / 2 : 0;
  const int R0 = h1 + h2, R1 = h2;
  const int t0 = tile * TN;
  const size_t rowbase = (size_t)b * C * T;
  const unsigned plane = (unsigned)C * (unsigned)T * 4u;  // bytes of one batch element (launcher: < 2^31)
  const int Tb = T * 4;

  // weight slot (conv s, chunk c) -> registers; rows past the chunk read as 0 (buffer bounds)
  int wvoff[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; i++) {
    const int f = tid + i * NTH;
    wvoff[i] = ((f / (C / 4)) * p.Mp + (f % (C / 4)) * 4) * 4;
  }
  u32x4 wr[MAXW];
  auto load_slot = [&](auto SC, int c) {
    constexpr int s = decltype(SC)::value;
    const int rows = p.cv[s].KW * p.cv[s].CK;
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.cv[s].w + (size_t)c * rows * p.Mp, (unsigned)(rows * p.Mp * 4));
#pragma unroll
    for (int i = 0; i < MAXW; i++)
      if ((i + 1) * NTH <= WSLOT / 4 || tid + i * NTH < WSLOT / 4)
        wr[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[i], 0, 0);
  };
  auto store_slot = [&](int buf) {
    u32x4* wd = reinterpret_cast<u32x4*>(Wb + buf * WSLOT);
#pragma unroll
    for (int i = 0; i < MAXW; i++)
      if ((i + 1) * NTH <= WSLOT / 4 || tid + i * NTH < WSLOT / 4) wd[tid + i * NTH] = wr[i];
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;

  load_slot(S0{}, 0);
  {  // input tile: column j <-> time t0 - R_0 - h_0 + j, PReLU of the first conv applied while staging.
     // wave w stages rows w, w + NWV, ...: NCG full 64-column groups per row + one 4-column tail per row
    const int tin0 = t0 - R0 - (p.cv[0].KW - 1) / 2;
    const float a0 = p.cv[0].alpha;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + rowbase, plane);
    float xr[RPW][NCG], xt = 0.f;
    int voff[NCG];
    bool okc[NCG];
#pragma unroll
    for (int g = 0; g < NCG; g++) {
      const int t = tin0 + g * 64 + lane;
      okc[g] = t >= 0 && t < T;
      voff[g] = t * 4;
    }
#pragma unroll
    for (int k = 0; k < RPW; k++) {
      const int soff = (wave + NWV * k) * Tb;
#pragma unroll
      for (int g = 0; g < NCG; g++) xr[k][g] = okc[g] ? buf_load(rx, voff[g], soff) : 0.f;
    }
    const int trow = wave + NWV * (lane >> 2), tt = tin0 + NC + (lane & 3);
    if (lane < 4 * RPW && tt >= 0 && tt < T) xt = buf_load(rx, (trow * T + tt) * 4, 0);
    float* dst = bufA + wave * XS + lane;
#pragma unroll
    for (int k = 0; k < RPW; k++)
#pragma unroll
      for (int g = 0; g < NCG; g++) {
        const float v = xr[k][g];
        dst[k * NWV * XS + g * 64] = v >= 0.f ? v : a0 * v;
      }
    if (lane < 4 * RPW) bufA[trow * XS + NC + (lane & 3)] = xt >= 0.f ? xt : a0 * xt;
  }
  for (int i = tid; i < C; i += NTH) {
    prm[i] = p.cv[0].bias[i];
    prm[C + i] = p.cv[1].bias[i];
    prm[2 * C + i] = D == 3 ? p.cv[2].bias[i] : 0.f;
    prm[3 * C + i] = p.film ? p.film[(size_t)b * p.film_bstride + i] : 1.f;
    prm[4 * C + i] = p.film ? p.film[(size_t)b * p.film_bstride + C + i] : 0.f;
  }
  for (int i = tid; i < C * 4; i += NTH) bufB[(i >> 2) * XS + NC + (i & 3)] = 0.f;  // never-written pad columns
  if (ts_on) tsv[1] = __builtin_readcyclecounter();
  store_slot(0);
  __syncthreads();
  if (ts_on) tsv[2] = __builtin_readcyclecounter();

  const int u = wn * 32 + l31;                       // this lane's column in every stage
  const int lrow = wm * 32 + 4 * lhalf;              // lane part of the accumulator row: row(r) = lrow + KR(r)
  const float* prm_l = prm + lrow;
  int q = 0;

  auto run_stage = [&](auto SC) {
    constexpr int s = decltype(SC)::value;
    using SN = std::integral_constant<int, (s < 2 ? s + 1 : 2)>;
    const int KW = p.cv[s].KW, CK = p.cv[s].CK, nch = C / CK;
    const float* inb = (s & 1) ? bufB : bufA;
    float* outb = (s & 1) ? bufA : bufB;
    const bool last = s == D - 1;
    const bool first3 = s == 0 && D == 3;
    const int t = t0 - (s == 0 ? R0 : (s == 1 ? R1 : 0)) + u;
    const bool inside = t >= 0 && t < T;
    const int evoff = (lrow * T + t) * 4;
    floatx16 acc;
    float ev[16];  // epilogue operand (cond add of conv1 / residual of the last conv), fetched one slot early
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[r] = 0.f; ev[r] = 0.f; }
    for (int c = 0; c < nch; c++, q++) {
      const bool more = c + 1 < nch;
      const bool has_next = more || !last;
      long long ta = 0, tb = 0, tc = 0, td = 0, te = 0;
      if (ts_on) ta = __builtin_readcyclecounter();
      if (more) load_slot(SC, c + 1);
      else if (!last) load_slot(SN{}, 0);
      if (!more) {
        const float* src = last ? p.res : (first3 ? p.add : nullptr);
        if (src && inside) {
          const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + rowbase, plane);
#pragma unroll
          for (int r = 0; r < 16; r++) ev[r] = buf_load(rs, evoff, ((r & 3) + 8 * (r >> 2)) * Tb);
        }
      }
      if (ts_on) td = __builtin_readcyclecounter();
      {
        const float* wsb = Wb + (q & 1) * WSLOT + lhalf * C + wm * 32 + l31;
        const float* xsb = inb + (c * CK + lhalf) * XS + u;
        const int gpt = CK / (2 * U);  // fragment groups per tap
        const int ngroups = gpt * KW;
        int cur_tap = 0, cur_jg = 0;
        auto load_group = [&](float (&av)[U], float (&bv)[U]) {
          const float* wt = wsb + (cur_tap * CK + cur_jg * (2 * U)) * C;
          const float* xq = xsb + cur_jg * (2 * U) * XS + cur_tap;
#pragma unroll
          for (int k = 0; k < U; k++) { av[k] = wt[k * 2 * C]; bv[k] = xq[k * 2 * XS]; }
          if (++cur_jg == gpt) { cur_jg = 0; ++cur_tap; }
        };
        auto mma_group = [&](float (&av)[U], float (&bv)[U]) {
#pragma unroll
          for (int k = 0; k < U; k++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k], bv[k], acc, 0, 0, 0);
        };
        float a0[U], b0[U], a1[U], b1[U];
        load_group(a0, b0);
        for (int g = 0; g < ngroups; g += 2) {
          if (g + 1 < ngroups) load_group(a1, b1);
          mma_group(a0, b0);
          if (g + 1 < ngroups) {
            if (g + 2 < ngroups) load_group(a0, b0);
            mma_group(a1, b1);
          }
        }
      }
      if (ts_on) tb = __builtin_readcyclecounter();
      if (!more) {
        const float* bias_l = prm_l + s * C;
        if (!last) {
          const float an = p.cv[s < 2 ? s + 1 : 2].alpha;
          const bool own = inside && t >= t0 && t < t0 + TN;
          const bool all_in = __builtin_amdgcn_ballot_w64(!inside) == 0ull;  // wave-uniform: no masking needed
          float* out_l = outb + lrow * XS + u;
          float v[16];
#pragma unroll
          for (int r = 0; r < 16; r++) v[r] = acc[r] + bias_l[(r & 3) + 8 * (r >> 2)];
          if (first3) {
            if (p.add) {
#pragma unroll
              for (int r = 0; r < 16; r++) v[r] = (v[r] + ev[r]) * p.add_scale;
            }
            if (p.film) {
#pragma unroll
              for (int r = 0; r < 16; r++) {
                const int kr = (r & 3) + 8 * (r >> 2);
                v[r] = prm_l[3 * C + kr] * v[r] + prm_l[4 * C + kr];
              }
            }
            if (p.c1_out && own) {
              const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.c1_out + rowbase, plane);
#pragma unroll
              for (int r = 0; r < 16; r++) buf_store(v[r], rc, evoff, ((r & 3) + 8 * (r >> 2)) * Tb);
            }
          }
          if (!all_in) {  // the zero padding the next conv sees outside the signal
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = inside ? v[r] : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const float x = v[r];
            out_l[((r & 3) + 8 * (r >> 2)) * XS] = x >= 0.f ? x : an * x;
          }
        } else if (u < TN && inside) {
          const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + rowbase, plane);
#pragma unroll
          for (int r = 0; r < 16; r++) {
            float x = acc[r] + bias_l[(r & 3) + 8 * (r >> 2)];
            if (p.res) x = (x + ev[r]) * p.res_scale;
            buf_store(x, ry, evoff, ((r & 3) + 8 * (r >> 2)) * Tb);
          }
        }
      }
      if (ts_on) te = __builtin_readcyclecounter();
      if (has_next) store_slot((q + 1) & 1);
      if (ts_on) tc = __builtin_readcyclecounter();
      __syncthreads();
      if (ts_on) { t_mma += tb - td; t_iss += td - ta; t_epi += te - tb; t_ws += tc - te; t_bar += __builtin_readcyclecounter() - tc; }
    }
  };
  run_stage(S0{});
  run_stage(S1{});
  if (D == 3) run_stage(S2{});

  if (ts_on && lane == 0) {
    long long* o = p.tstamps + ((size_t)blockIdx.x * NWV + wave) * 8;
    o[0] = tsv[1] - tsv[0]; o[1] = tsv[2] - tsv[1]; o[2] = t_mma; o[3] = t_epi; o[4] = t_bar;
    o[5] = __builtin_readcyclecounter() - tsv[0]; o[6] = t_iss; o[7] = t_ws;
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

struct ChainVariant {
  int C, NC, NWV;
  void (*kern)(ChainArgs, int, int);
};
static const ChainVariant kChainVariants[] = {
    {32, 128, 4, conv_chain_kernel<1, 4>},
    {32, 256, 8, conv_chain_kernel<1, 8>},
    {64, 128, 8, conv_chain_kernel<2, 8>},
};
constexpr int kNumChainVariants = sizeof(kChainVariants) / sizeof(kChainVariants[0]);
static size_t chain_smem_bytes(const ChainVariant& v) {
  return 4 * ((size_t)2 * v.C * (v.NC + 4) + 2 * 96 * v.C + 5 * v.C);
}

static bool chain_shape_ok(const ChainArgs& a) {
  if (a.depth != 2 && a.depth != 3) return false;
  if (a.C != 32 && a.C != 64) return false;
  if (a.Mp < a.C || a.Mp % 4 || a.T < 1 || (long)a.C * a.T * 4 >= (1L << 31)) return false;
  for (int s = 0; s < a.depth; s++) {
    const ChainConv& c = a.cv[s];
    if ((c.KW != 3 && c.KW != 5) || c.CK < 8 || (c.CK & (c.CK - 1)) || a.C % c.CK || c.KW * c.CK > 96) return false;
  }
  return true;
}

// Estimated duration in core cycles of variant v: rounds x (MFMA time of one block at its SIMD sharing + fixed part)
static double chain_variant_cost(const ChainArgs& a, const ChainVariant& v, int num_cu) {
  int R0 = 0, mf = 0;
  for (int s = 0; s < a.depth; s++) {
    if (s) R0 += (a.cv[s].KW - 1) / 2;
    mf += a.C * a.cv[s].KW / 2;  // MFMAs per wave
  }
  const int TN = v.NC - 2 * R0;
  const long blocks = (long)a.B * ((a.T + TN - 1) / TN);
  const int occ = (int)(160 * 1024 / chain_smem_bytes(v));
  const long slots = (long)num_cu * (occ < 1 ? 1 : occ);
  const long rounds = (blocks + slots - 1) / slots;
  const int resident = (int)((blocks < slots ? blocks : slots) + num_cu - 1) / num_cu;  // blocks sharing a CU
  const double per_block = mf * 64.0 * (v.NWV / 4.0) * resident / 0.8 + 8000.0;
  return rounds * per_block;
}

double chain_cost(const ChainArgs& a, int num_cu, int* nc_out) {
  if (!chain_shape_ok(a)) return -1.0;
  double best = -1.0;
  for (int i = 0; i < kNumChainVariants; i++) {
    const ChainVariant& v = kChainVariants[i];
    if (v.C != a.C || (a.force_nc && v.NC != a.force_nc)) continue;
    const double c = chain_variant_cost(a, v, num_cu);
    if (best < 0 || c < best) { best = c; if (nc_out) *nc_out = v.NC; }
  }
  return best;
}

static hipError_t init_chain_kernels() {
  for (int i = 0; i < kNumChainVariants; i++) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kChainVariants[i].kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_smem_bytes(kChainVariants[i]));
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_chain(const ChainArgs& a, int num_cu, hipStream_t st, int* variant) {
  int nc = 0;
  if (chain_cost(a, num_cu, &nc) < 0) return hipErrorInvalidConfiguration;
  for (int i = 0; i < kNumChainVariants; i++) {
    const ChainVariant& v = kChainVariants[i];
    if (v.C != a.C || v.NC != nc) continue;
    int R0 = 0;
    for (int s = 1; s < a.depth; s++) R0 += (a.cv[s].KW - 1) / 2;
    const int TN = v.NC - 2 * R0;
    const int ntiles = (a.T + TN - 1) / TN;
    if (variant) *variant = 100 + 10 * a.depth + i;
    hipLaunchKernelGGL(v.kern, dim3(ntiles * a.B), dim3(64 * v.NWV), chain_smem_bytes(v), st, a, TN, ntiles);
    return hipGetLastError();
  }
  return hipErrorInvalidConfiguration;
}

// =========================================================================================================
// Small-K rate-change convs of the wide levels, with the anti-alias FIR fused (blocks.py:205-227)
//   down:  y = conv_{k=s=R}(FIR_{2R+1}(prelu(x))) + bias        K = Cin R <= 96, M = Cout <= 96
// At T = 64 160 the first rate-change conv is a 0.26 GFLOP GEMM with K = 64: bandwidth- and latency-sized work that used
// to take a FIR pass (6.5 us) + a generic conv launch (11.4 us) + a dispatch gap; here 11.6 us in one launch.  (The next
// level -- K = 256, M = 128, 251 workgroups with one wave per SIMD -- came out at 19.6 us against 10.6 + 6.5: its phases
// run back to back with nothing to overlap them, so it stays on the FIR pass + the strided direct kernel.)
// One workgroup owns BQ output frames and ALL output channels, so nothing is split or reduced across waves:
//   1. every thread loads runs of the input, applies PReLU and the FIR in registers (same tap order as fir_kernel) and
//      writes the filtered tile to LDS once;
//   2. the whole weight matrix of the layer sits in registers (K/2 A operands per lane, issued before step 1);
//   3. K/2 MFMAs per wave on B operands read from LDS; bias in the epilogue, stores straight from the accumulators.
// =========================================================================================================
template <int R, int MT, int NWN, int K2>
__global__ __launch_bounds__(64 * MT * NWN) void rate_down_kernel(ConvArgs p) {
  constexpr int NW = MT * NWN, NT = 64 * NW, BQ = 32 * NWN, SP = BQ * R, ROW = SP + 4, NCH = SP / 4;
  static_assert(SP % 4 == 0, "runs of 4 samples");
  extern __shared__ __attribute__((aligned(16))) float smem[];  // xf[Cin][ROW]
  const int tid = threadIdx.x, lane = tid & 63, lhalf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % MT, wn = wave / MT;
  const int q0 = blockIdx.x * BQ, b = blockIdx.y;
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int Cin = p.Cin, Tin = p.Tin, Mp = p.Mp, CK = p.CK, lck = 31 - __clz(CK);
  // ---- filtered input tile -> LDS.  Work item = (channel, run of 4 output samples); the raw windows of IB items are
  // loaded first (wide loads where the window is inside the row), then filtered: with one wave per SIMD nothing else
  // hides a dependent load -> FIR -> store chain per item.
  const bool fir = p.fir != nullptr;
  float f[2 * R + 1];
#pragma unroll
  for (int j = 0; j <= 2 * R; j++) f[j] = fir ? p.fir[j] : 0.f;
  const float alpha = p.alpha_val;
  const bool act = p.act != 0;
  const float* xb = p.x + (size_t)b * Cin * Tin;
  constexpr int CIN = 2 * K2 / R, ITEMS = CIN * NCH / NT, IB = ITEMS < 4 ? ITEMS : 4, WIN = 4 + 2 * R;
  static_assert(CIN * NCH % NT == 0 && ITEMS % IB == 0, "items per thread");
  // (the K/2 A operands -- the layer's whole weight matrix -- are requested right after the first batch of input
  // windows: loads return in order, so the filter only waits for the windows and the weights land behind it)
  float a[K2];
#pragma unroll 1
  for (int i0 = 0; i0 < ITEMS; i0 += IB) {
    float v[IB][WIN];
#pragma unroll
    for (int u = 0; u < IB; u++) {
      const int item = tid + (i0 + u) * NT;
      const int ci = item / NCH, c = item - ci * NCH;
      const int ts = q0 * R + 4 * c - (fir ? R : 0);  // first sample of the window (a multiple of 2 / of 4 for R = 4)
      const int nwin = fir ? WIN : 4;
      const float* xr = xb + (size_t)ci * Tin;
      if (ts >= 0 && ts + nwin <= Tin) {
        if (!fir) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(xr + ts);
          v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < WIN; i += 2) {
            const f32x2 q = *reinterpret_cast<const f32x2*>(xr + ts + i);
            v[u][i] = q.x; v[u][i + 1] = q.y;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < WIN; i++) {
          const int t = ts + i;
          v[u][i] = (i < nwin && t >= 0 && t < Tin) ? xr[t] : 0.f;
        }
      }
    }
    if (i0 == 0) {
      // K index 2 ks + half = (ci, tap), packed row ((ci / CK) R + tap) CK + ci % CK
#pragma unroll
      for (int ks = 0; ks < K2; ks++) {
        const int idx = 2 * ks + lhalf, ci = idx / R, tap = idx - ci * R;
        const int row = (((ci >> lck) * R + tap) << lck) + (ci & (CK - 1));
        a[ks] = p.w[(size_t)row * Mp + 32 * wm + l31];
      }
    }
#pragma unroll
    for (int u = 0; u < IB; u++) {
      const int item = tid + (i0 + u) * NT;
      const int ci = item / NCH, c = item - ci * NCH;
#pragma unroll
      for (int i = 0; i < WIN; i++) v[u][i] = (act && v[u][i] < 0.f) ? alpha * v[u][i] : v[u][i];  // blocks.py:213
      f32x4 o;
      if (fir) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float acc = 0.f;
#pragma unroll
          for (int i = 0; i <= 2 * R; i++) acc = fmaf(f[i], v[u][j + i], acc);
          o[j] = acc;
        }
      } else {
        o = f32x4{v[u][0], v[u][1], v[u][2], v[u][3]};
      }
      *reinterpret_cast<f32x4*>(&smem[ci * ROW + 4 * c]) = o;
    }
  }
  __syncthreads();
  // ---- K/2 MFMAs per wave
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  const float* bs = smem + (32 * wn + l31) * R;
#pragma unroll
  for (int ks = 0; ks < K2; ks++) {
    const int idx = 2 * ks + lhalf, ci = idx / R, tap = idx - ci * R;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], bs[ci * ROW + tap], acc, 0, 0, 0);
  }
  // ---- epilogue
  const int q = q0 + 32 * wn + l31;
  if (q < p.Nq) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      if (m < p.M) p.y[((size_t)b * p.Cout + m) * p.Nq + q] = acc[r] + p.bias[m];
    }
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

struct RateDownCfg {
  int R, M, Cin;
  void (*kern)(ConvArgs);
  int threads, bq;
};
static const RateDownCfg kRateDownCfgs[] = {
    {2, 64, 32, rate_down_kernel<2, 2, 2, 32>, 256, 64},    // PP16 / OR16: 32 -> 64 channels, T -> T/2
    {2, 96, 48, rate_down_kernel<2, 3, 1, 48>, 192, 32},    // PP24: 48 -> 96
};
bool rate_down_supported(const ConvArgs& a) {
  if (a.up != 1 || a.stride != a.KW || a.pad != 0 || a.Tin != a.Nq * a.stride || a.add || a.film || a.res || a.in_scale)
    return false;
  if (a.fir && a.fir_len != 2 * a.stride + 1) return false;
  for (const RateDownCfg& c : kRateDownCfgs)
    if (c.R == a.stride && c.M == a.M && c.Cin == a.Cin) return true;
  return false;
}
hipError_t launch_rate_down(const ConvArgs& a, hipStream_t st, int* cfg_out) {
  if (!rate_down_supported(a)) return hipErrorNotSupported;
  for (const RateDownCfg& c : kRateDownCfgs) {
    if (c.R != a.stride || c.M != a.M || c.Cin != a.Cin) continue;
    const size_t smem = (size_t)a.Cin * (c.bq * c.R + 4) * 4;
    if (cfg_out) *cfg_out = 40 + c.R;
    hipLaunchKernelGGL(c.kern, dim3((a.Nq + c.bq - 1) / c.bq, a.B), dim3(c.threads), smem, st, a);
    return hipGetLastError();
  }
  return hipErrorNotSupported;
}

//   up:    y = FIR_{2R+1}(convT_{k=s=R}(prelu(x))) + bias ; y = res ? (y + res) res_scale : y      K = Cin <= 96
// The last up conv (64 -> 32 channels x 2 phases, K = 64, T/2 -> T): the same idea the other way round.  A workgroup owns
// BF input frames (the outer two are halo when there is a FIR) and all M = Cout R phase rows: A (the whole weight matrix)
// and B (this lane's K/2 input samples, PReLU applied) go straight from global memory to registers, K/2 MFMAs per wave,
// the phase-GEMM result is laid out in LDS as [row = co R + phase][frame] and filtered from there -- same tap order as
// fir_kernel -- with bias and residual on the way out.  Replaces a generic launch (14.2 us) + a FIR pass (6.5 us).
template <int R, int MT, int NWN, int K2>
__global__ __launch_bounds__(64 * MT * NWN) void rate_up_kernel(ConvArgs p) {
  constexpr int NW = MT * NWN, NT = 64 * NW, BF = 32 * NWN, UP = BF + 1;  // UP: LDS row pitch (M = 32 MT rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];  // U[M][UP]
  const int tid = threadIdx.x, lane = tid & 63, lhalf = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % MT, wn = wave / MT;
  const bool fir = p.fir != nullptr;
  const int H = fir ? 1 : 0, BV = BF - 2 * H;  // halo frames, frames this block completes
  const int q0 = blockIdx.x * BV - H, b = blockIdx.y;
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int Tin = p.Tin, Mp = p.Mp;
  // ---- operands: B first (older loads return first), then A
  const int fq = q0 + 32 * wn + l31;
  const bool inside = fq >= 0 && fq < Tin;
  const float* xb = p.x + ((size_t)b * p.Cin + lhalf) * Tin + (inside ? fq : 0);
  float bx[K2], a[K2];
#pragma unroll
  for (int ks = 0; ks < K2; ks++) bx[ks] = inside ? xb[(size_t)2 * ks * Tin] : 0.f;
#pragma unroll
  for (int ks = 0; ks < K2; ks++) a[ks] = p.w[(size_t)(2 * ks + lhalf) * Mp + 32 * wm + l31];
  const float alpha = p.alpha_val;
  const bool act = p.act != 0;
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < K2; ks++) {
    const float x = bx[ks];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], (act && x < 0.f) ? alpha * x : x, acc, 0, 0, 0);
  }
  // ---- phase-GEMM tile -> LDS (frames outside the signal are zero: the 'same' padding of the FIR)
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int row = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
    smem[row * UP + 32 * wn + l31] = inside ? acc[r] : 0.f;
  }
  __syncthreads();
  float f[2 * R + 1];
#pragma unroll
  for (int j = 0; j <= 2 * R; j++) f[j] = fir ? p.fir[j] : 0.f;
  const int Cout = p.Cout;
  const int span = BV * R;  // output samples per channel
  const int total = Cout * span;
  // Rows that are 16-byte multiples: four consecutive samples per thread -- one float4 of the residual in, one float4 out
  // (a tile's samples of a channel start at a multiple of span, itself a multiple of 4).  Same arithmetic per sample as the
  // scalar loop below: bit-identical.  (B = 8: 198 MB per launch at 2.7 TB/s with 4-byte accesses.)
  if ((p.Tout & 3) == 0 && ((BF * R) & 3) == 0 && ((2 * R) & 3) == 0) {
    constexpr int QPT = (32 * MT / R * BF * R / 4 + NT - 1) / NT;
    f32x4 rs4[QPT];
    size_t idx4[QPT];
    int co4[QPT], tl4[QPT];
#pragma unroll
    for (int u = 0; u < QPT; u++) {
      const int e = 4 * (tid + u * NT);
      const int co = e / span, tl = e - co * span + H * R;
      const long t = (long)q0 * R + tl;
      const bool on = e < total && t < p.Tout;
      co4[u] = on ? co : -1; tl4[u] = tl;
      idx4[u] = on ? ((size_t)b * Cout + co) * p.Tout + (size_t)t : 0;
      rs4[u] = (on && p.res) ? *reinterpret_cast<const f32x4*>(p.res + idx4[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < QPT; u++) {
      if (co4[u] < 0) continue;
      const float* urow = smem + (co4[u] * R) * UP;
      const float bi = p.bias[co4[u]];
      f32x4 o;
#pragma unroll
      for (int s4 = 0; s4 < 4; s4++) {
        const int tl = tl4[u] + s4;
        float v;
        if (fir) {
          int tau = tl - R, qf = tau / R, ph = tau - qf * R;
          v = 0.f;
#pragma unroll
          for (int j = 0; j <= 2 * R; j++) {
            v = fmaf(f[j], urow[ph * UP + qf], v);
            if (++ph == R) { ph = 0; qf++; }
          }
        } else {
          const int qf = tl / R, ph = tl - qf * R;
          v = urow[ph * UP + qf];
        }
        v += bi;
        if (p.res) v = (v + rs4[u][s4]) * p.res_scale;
        o[s4] = v;
      }
      *reinterpret_cast<f32x4*>(p.y + idx4[u]) = o;
    }
    if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
    return;
  }
  // outputs in batches of EB per thread: the residual loads of a batch go out together, ahead of the filter
  constexpr int EPT = (32 * MT / R * BF * R + NT - 1) / NT, EB = 8;
  static_assert(EPT % EB == 0, "output batches");
#pragma unroll 1
  for (int e0 = 0; e0 < EPT; e0 += EB) {
    float rs[EB];
    size_t idx[EB];
    int co_[EB], tl_[EB];
#pragma unroll
    for (int u = 0; u < EB; u++) {
      const int e = tid + (e0 + u) * NT;
      const int co = e / span, tl = e - co * span + H * R;  // sample index inside the tile (frame tl / R, phase tl % R)
      const long t = (long)q0 * R + tl;
      const bool on = e < total && t < p.Tout;
      co_[u] = on ? co : -1; tl_[u] = tl;
      idx[u] = on ? ((size_t)b * Cout + co) * p.Tout + (size_t)t : 0;
      rs[u] = (on && p.res) ? p.res[idx[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < EB; u++) {
      if (co_[u] < 0) continue;
      const float* urow = smem + (co_[u] * R) * UP;
      float v;
      if (fir) {
        int tau = tl_[u] - R, qf = tau / R, ph = tau - qf * R;  // tau >= 0: one halo frame in front
        v = 0.f;
#pragma unroll
        for (int j = 0; j <= 2 * R; j++) {
          v = fmaf(f[j], urow[ph * UP + qf], v);
          if (++ph == R) { ph = 0; qf++; }
        }
      } else {
        const int qf = tl_[u] / R, ph = tl_[u] - qf * R;
        v = urow[ph * UP + qf];
      }
      v += p.bias[co_[u]];
      if (p.res) v = (v + rs[u]) * p.res_scale;
      p.y[idx[u]] = v;
    }
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

struct RateUpCfg {
  int R, M, Cin;
  void (*kern)(ConvArgs);
  int threads, bf;
};
static const RateUpCfg kRateUpCfgs[] = {
    {2, 64, 64, rate_up_kernel<2, 2, 2, 32>, 256, 64},  // PP16 / OR16: 64 -> 32 channels x 2 phases, T/2 -> T
    {2, 96, 96, rate_up_kernel<2, 3, 1, 48>, 192, 32},  // PP24: 96 -> 48 x 2
};
bool rate_up_supported(const ConvArgs& a) {
  if (a.up < 2 || a.stride != 1 || a.KW != 1 || a.pad != 0 || a.add || a.film || a.in_scale || a.Nq != a.Tin ||
      a.Tout != a.Tin * a.up || a.M != a.Cout * a.up)
    return false;
  if (a.fir && a.fir_len != 2 * a.up + 1) return false;
  for (const RateUpCfg& c : kRateUpCfgs)
    if (c.R == a.up && c.M == a.M && c.Cin == a.Cin) return true;
  return false;
}
hipError_t launch_rate_up(const ConvArgs& a, hipStream_t st, int* cfg_out) {
  if (!rate_up_supported(a)) return hipErrorNotSupported;
  for (const RateUpCfg& c : kRateUpCfgs) {
    if (c.R != a.up || c.M != a.M || c.Cin != a.Cin) continue;
    const int bv = c.bf - (a.fir ? 2 : 0);
    const size_t smem = (size_t)c.M * (c.bf + 1) * 4;
    if (cfg_out) *cfg_out = 45 + c.R;
    hipLaunchKernelGGL(c.kern, dim3((a.Tin + bv - 1) / bv, a.B), dim3(c.threads), smem, st, a);
    return hipGetLastError();
  }
  return hipErrorNotSupported;
}

// =========================================================================================================
// small VALU kernels
// =========================================================================================================
__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.f ? v : a * v; }

// grid = (time tiles of 256, channel groups of IN_CONV_CG, batch): at batch 1 a (T/256)-block grid is one block per CU
// with 32 dependent stores per thread; splitting the channels gives the dispatcher 4x the blocks for the same traffic
constexpr int IN_CONV_CG = 8;
__global__ __launch_bounds__(256) void in_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const StepCoef* coef,
                                                      int coef_bstride, float* __restrict__ y, int C, int T, int KW) {
  const int b = blockIdx.z;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float sc = coef ? coef[(size_t)b * coef_bstride].w_in : 1.f;  // universe.py:199,202
  float xv[7];
  const int pad = (KW - 1) / 2;
#pragma unroll
  for (int k = 0; k < 7; k++) {
    int tt = t + k - pad;
    xv[k] = (k < KW && tt >= 0 && tt < T) ? x[(size_t)b * T + tt] * sc : 0.f;
  }
  const int c0 = blockIdx.y * IN_CONV_CG, c1 = min(C, c0 + IN_CONV_CG);
  for (int c = c0; c < c1; c++) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; k++)
      if (k < KW) acc = fmaf(w[c * KW + k], xv[k], acc);
    y[((size_t)b * C + c) * T + t] = acc + bias[c];
  }
}

hipError_t launch_in_conv(const float* x, const float* w, const float* bias, const StepCoef* coef, int coef_bstride,
                          float* y, int B, int C, int T, int KW, hipStream_t s) {
  if (KW > 7) return hipErrorInvalidValue;
  hipLaunchKernelGGL(in_conv_kernel, dim3((T + 255) / 256, (C + IN_CONV_CG - 1) / IN_CONV_CG, B), dim3(256), 0, s, x, w,
                     bias, coef, coef_bstride, y, C, T, KW);
  return hipGetLastError();
}

__global__ __launch_bounds__(512) void out_conv_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ alphas,
                                                       const float* x, const float* noise, float* out,
                                                       const StepCoef* coef, int coef_bstride, int edm, int mode, int C,
                                                       int T, int KW) {
  // block = 64 time quads x 8 channel groups (one wave each: at batch 1 the grid is one block per CU, and a wave's
  // channels are a serial chain of loads); 4 consecutive output samples per thread from one aligned float4 + the halo
  // scalars per channel row; the 8 partial sums meet in LDS
  __shared__ float part[8][64][4];
  const int b = blockIdx.y;
  const int tq = threadIdx.x & 63, cgp = threadIdx.x >> 6;
  const int t0 = (blockIdx.x * 64 + tq) * 4;
  const float a1 = alphas[0], a2 = alphas[1];
  const int pad = (KW - 1) / 2;  // <= 3
  const bool vec = (T & 3) == 0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int cpg = (C + 7) / 8;
  if (t0 < T) {
#pragma unroll 4
    for (int cc = 0; cc < cpg; cc++) {
      const int c = cgp * cpg + cc;
      if (c >= C) break;
      const float* sr = s + ((size_t)b * C + c) * T;
      float v[10];  // samples t0-3 .. t0+6
#pragma unroll
      for (int i = 0; i < 10; i++) v[i] = 0.f;
      if (vec) {
        const f32x4 m = *reinterpret_cast<const f32x4*>(sr + t0);
        v[3] = m.x; v[4] = m.y; v[5] = m.z; v[6] = m.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) v[3 + i] = (t0 + i < T) ? sr[t0 + i] : 0.f;
      }
#pragma unroll
      for (int i = 1; i <= 3; i++) {
        if (i <= pad) {
          v[3 - i] = (t0 - i >= 0) ? sr[t0 - i] : 0.f;
          v[6 + i] = (t0 + 3 + i < T) ? sr[t0 + 3 + i] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 10; i++) v[i] = prelu(prelu(v[i], a1), a2);
#pragma unroll
      for (int k = 0; k < 7; k++) {
        if (k < KW) {
          const float wk = w[c * KW + k];
#pragma unroll
          for (int j = 0; j < 4; j++) acc[j] = fmaf(wk, v[3 + j + k - pad], acc[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) part[cgp][tq][j] = acc[j];
  __syncthreads();
  // thread (tq, j = cgp < 4) finishes sample t0 + j
  const int j = cgp, t = t0 + j;
  if (j >= 4 || t >= T) return;
  const float net = (((part[0][tq][j] + part[1][tq][j]) + (part[2][tq][j] + part[3][tq][j])) +
                     ((part[4][tq][j] + part[5][tq][j]) + (part[6][tq][j] + part[7][tq][j]))) + bias[0];
  const StepCoef cf = coef[(size_t)b * coef_bstride];
  const size_t i = (size_t)b * T + t;
  const float xv = x ? x[i] : 0.f;
  float score = net;
  if (edm) {
    float est = cf.w_skip * xv + cf.w_out * net;  // universe.py:203
    score = (est - xv) / cf.sig2;                 // universe.py:204
  }
  if (mode == OUT_SCORE) {
    out[i] = score;
  } else {
    float r = xv + cf.c1 * score;  // universe.py:339 / :343
    if (noise) r = r + cf.beta * (noise[i] * cf.s_next);
    out[i] = r;
  }
}

hipError_t launch_out_conv(const float* s, const float* w, const float* bias, const float* alphas, const float* x,
                           const float* noise, float* out, const StepCoef* coef, int coef_bstride, int edm, int mode,
                           int B, int C, int T, int KW, hipStream_t st) {
  if (KW > 7) return hipErrorInvalidValue;
  hipLaunchKernelGGL(out_conv_kernel, dim3((T + 255) / 256, B), dim3(512), 0, st, s, w, bias, alphas, x, noise, out,
                     coef, coef_bstride, edm, mode, C, T, KW);
  return hipGetLastError();
}

// ---- noise-level embedding ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sigma_embed_kernel(const StepCoef* coef, const float* __restrict__ prm,
                                                          int simple, int n_rff, int D, float* __restrict__ g) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const float ls = log10f(coef[s].sigma_net);  // score.py:283
  const float two_pi = 6.283185307179586f;
  if (simple) {
    // sigma_block.py:73-78  f = 0.5*sigmoid(w*ls + b); p = (2pi*f)*k; g = [sin p, cos p]
    const float f = 0.5f * (1.0f / (1.0f + expf(-(prm[0] * ls + prm[1]))));
    const float tf = two_pi * f;
    for (int k = tid; k < D / 2; k += 256) {
      float ph = tf * (float)k;
      g[(size_t)s * D + k] = (float)sin((double)ph);
      g[(size_t)s * D + D / 2 + k] = (float)cos((double)ph);
    }
    return;
  }
  // sigma_block.py:50-57 random Fourier features + 3 x (Linear -> PReLU)
  __shared__ float bufA[1024], bufB[1024];
  for (int k = tid; k < n_rff; k += 256) {
    float ph = (two_pi * prm[k]) * ls;
    bufA[k] = (float)sin((double)ph);
    bufA[n_rff + k] = (float)cos((double)ph);
  }
  __syncthreads();
  const float* q = prm + n_rff;
  int din = 2 * n_rff;
  float* in = bufA;
  float* outb = bufB;
  for (int layer = 0; layer < 3; layer++) {
    int dout = layer == 2 ? D : 2 * din;
    const float al = q[0];
    const float* W = q + 1;
    const float* bb = W + (size_t)dout * din;
    for (int o = tid; o < dout; o += 256) {
      float acc = 0.f;
      for (int i = 0; i < din; i++) acc = fmaf(W[(size_t)o * din + i], in[i], acc);
      acc += bb[o];
      acc = prelu(acc, al);
      if (layer == 2) g[(size_t)s * D + o] = acc; else outb[o] = acc;
    }
    __syncthreads();
    q = bb + dout;
    din = dout;
    float* tmp = in; in = outb; outb = tmp;
  }
}

hipError_t launch_sigma_embed(const StepCoef* coef, int S, const float* params, int simple, int n_rff, int D, float* g,
                              hipStream_t st) {
  if (D > 1024 || 8 * n_rff > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sigma_embed_kernel, dim3(S), dim3(256), 0, st, coef, params, simple, n_rff, D, g);
  return hipGetLastError();
}

// one wave per output row; the row of W stays in registers across the S columns
__global__ __launch_bounds__(256) void film_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ film, int S,
                                                   int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float wv[16];
  const int per = D / 64;  // <= 16
#pragma unroll
  for (int i = 0; i < 16; i++) wv[i] = (i < per) ? W[(size_t)row * D + i * 64 + lane] : 0.f;
  const float bb = bias[row];
  for (int s = 0; s < S; s++) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++)
      if (i < per) acc = fmaf(wv[i], g[(size_t)s * D + i * 64 + lane], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) film[(size_t)s * rows + row] = acc + bb;
  }
}

hipError_t launch_film(const float* g, const float* W, const float* b, float* film, int S, int rows, int D,
                       hipStream_t st) {
  if (D % 64 || D > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(film_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, g, W, b, film, S, rows, D);
  return hipGetLastError();
}

__global__ void upload_coef_kernel(StepCoef* dst, CoefBlock blk, int n) {
  int i = threadIdx.x;
  if (i < n) dst[i] = blk.c[i];
}
hipError_t launch_upload_coef(StepCoef* dst, const CoefBlock& blk, int n, hipStream_t st) {
  hipLaunchKernelGGL(upload_coef_kernel, dim3(1), dim3(64), 0, st, dst, blk, n);
  return hipGetLastError();
}

// ---- block reductions --------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
template <typename T>
__device__ T block_sum(T v, T* sh) {  // blockDim multiple of 64, <= 1024
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); i++) r += sh[i];
  return r;
}
__device__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = fmaxf(r, sh[i]);
  return r;
}

// pad + zero-mean + unit-level normalisation, one block per batch element
__global__ __launch_bounds__(1024) void pad_normalize_kernel(const float* __restrict__ mix, float* __restrict__ y,
                                                             float* __restrict__ stats, int T_raw, int T_pad,
                                                             int pad_left, float level) {
  __shared__ double shd[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = mix + (size_t)b * T_raw;
  // (one CU per utterance and three dependent passes: the loops are unrolled by hand so that 8 loads are in flight per
  // thread -- same elements per thread, same order of the double sums as the plain loop)
  constexpr int U = 8, NREG = 64;
  float* yb = y + (size_t)b * T_pad;
  if (T_raw <= NREG * 1024) {
    // utterances of up to 65 536 samples (4 s at 16 kHz): every thread keeps its <= 64 samples in registers -- ONE trip to
    // memory with all loads in flight instead of three dependent passes (26 -> ~6 us at batch 1, where this kernel is the
    // first link of the chain).  Same elements per thread and the same order of the double sums as the loops below.
    // (buffer instructions: one VGPR offset for all 64 accesses, the k-th sample 4096 k bytes further in the SGPR offset;
    // samples past the end of the row read as 0 and stores past the end of the padded row are dropped by the bounds check)
    float v[NREG];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)T_raw * 4u);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(yb, (unsigned)T_pad * 4u);
    const int nk = (T_raw - tid + 1023) >> 10;  // this thread's samples: tid + 1024 k, k < nk
#pragma unroll
    for (int k = 0; k < NREG; k++) v[k] = buf_load(rx, tid * 4, k * 4096);
    double s = 0, sq = 0;
#pragma unroll
    for (int k = 0; k < NREG; k++) {  // (+ 0.0 past the end: exact)
      // (the empty asm ties sample k to the running sums: without it the scheduler converts all 64 samples to double
      // first -- 128 more live registers, spills)
      asm volatile("" : "+v"(v[k]), "+v"(s), "+v"(sq));
      const double d = v[k]; s += d; sq += d * d;
    }
    s = block_sum(s, shd);
    sq = block_sum(sq, shd);
    const float mean = (float)(s / T_pad);  // norm.py:62  (mean over the padded signal)
    double ss = 0;
#pragma unroll
    for (int k = 0; k < NREG; k++) {
      asm volatile("" : "+v"(v[k]), "+v"(ss));
      const double d = k < nk ? (double)(v[k] - mean) : 0.0; ss += d * d;
    }
    ss = block_sum(ss, shd);
    ss += (double)(T_pad - T_raw) * (double)(0.f - mean) * (double)(0.f - mean);
    float sd = (float)sqrt(ss / (double)(T_pad - 1));  // unbiased std, norm.py:22-23
    sd = fmaxf(sd, 1e-5f);
    const float gain = level / sd;
    // a sample past the end is 0 here, i.e. exactly the padding value (0 - mean) * gain of the position it lands on
#pragma unroll
    for (int k = 0; k < NREG; k++) buf_store((v[k] - mean) * gain, ry, (pad_left + tid) * 4, k * 4096);
    const float pv = (0.f - mean) * gain;  // the rest of the padding
    for (int t = tid; t < pad_left; t += 1024) yb[t] = pv;
    for (int t = pad_left + NREG * 1024 + tid; t < T_pad; t += 1024) yb[t] = pv;
    if (tid == 0) {
      stats[b * 4 + 0] = mean;
      stats[b * 4 + 1] = gain;
      stats[b * 4 + 2] = (float)sqrt(sq / (double)T_raw);
      stats[b * 4 + 3] = 0.f;
    }
    return;
  }
  double s = 0, sq = 0;
  {
    int t = tid;
    for (; t + (U - 1) * 1024 < T_raw; t += U * 1024) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = xb[t + u * 1024];
#pragma unroll
      for (int u = 0; u < U; u++) { const double d = v[u]; s += d; sq += d * d; }
    }
    for (; t < T_raw; t += 1024) { const double d = xb[t]; s += d; sq += d * d; }
  }
  s = block_sum(s, shd);
  sq = block_sum(sq, shd);
  const float mean = (float)(s / T_pad);  // norm.py:62  (mean over the padded signal)
  double ss = 0;
  {
    int t = tid;
    for (; t + (U - 1) * 1024 < T_raw; t += U * 1024) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = xb[t + u * 1024];
#pragma unroll
      for (int u = 0; u < U; u++) { const double d = (double)(v[u] - mean); ss += d * d; }
    }
    for (; t < T_raw; t += 1024) { const double d = (double)(xb[t] - mean); ss += d * d; }
  }
  ss = block_sum(ss, shd);
  ss += (double)(T_pad - T_raw) * (double)(0.f - mean) * (double)(0.f - mean);
  float sd = (float)sqrt(ss / (double)(T_pad - 1));  // unbiased std, norm.py:22-23
  sd = fmaxf(sd, 1e-5f);
  const float gain = level / sd;
#pragma unroll 8
  for (int t = tid; t < T_pad; t += 1024) {
    int tr = t - pad_left;
    float v = (tr >= 0 && tr < T_raw) ? xb[tr] : 0.f;
    yb[t] = (v - mean) * gain;
  }
  if (tid == 0) {
    stats[b * 4 + 0] = mean;
    stats[b * 4 + 1] = gain;
    stats[b * 4 + 2] = (float)sqrt(sq / T_raw);  // mix_rms, universe.py:259
    stats[b * 4 + 3] = 0.f;
  }
}
hipError_t launch_pad_normalize(const float* mix, float* y, float* stats, int B, int T_raw, int T_pad, int pad_left,
                                float level, hipStream_t st) {
  hipLaunchKernelGGL(pad_normalize_kernel, dim3(B), dim3(1024), 0, st, mix, y, stats, T_raw, T_pad, pad_left, level);
  return hipGetLastError();
}

// post_kernel for utterances of up to 65 536 samples: as in pad_normalize_kernel the thread's samples stay in registers -- one
// trip to memory with all loads in flight instead of three dependent passes (25 -> ~6 us at batch 1, where this kernel is the
// last link of the chain).  Same elements per thread, same order of the double sum.  (A kernel of its own: sharing a
// function with the general loops below costs SGPR spills.)
constexpr int POST_NREG = 64;
__global__ __launch_bounds__(1024) void post_reg_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                        float* __restrict__ out, int T_raw, int T_pad, int pad_left,
                                                        int keep_rms, int peak_guard) {
  constexpr int NREG = POST_NREG;
  __shared__ double shd[16];
  __shared__ float shf[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * T_pad + pad_left;
  float g = 1.f;
  {
    float v[NREG];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(xb, (unsigned)T_raw * 4u);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)b * T_raw, (unsigned)T_raw * 4u);
    int so = 0;  // the k-th access 4096 k bytes further: ONE scalar offset stepped by asm (64 constants would spill SGPRs)
#pragma unroll
    for (int k = 0; k < NREG; k++) {
      v[k] = buf_load(rx, tid * 4, so);  // 0 past the end
      asm volatile("s_add_u32 %0, %0, 0x1000" : "+s"(so) : : "scc");
    }
    if (keep_rms) {  // universe.py:352-354
      double sq = 0;
#pragma unroll
      for (int k = 0; k < NREG; k++) {
        // the conversion as asm that also "touches" the running sum: one sample at a time (see pad_normalize_kernel); the
        // sample registers themselves stay untouched -- redefining them inside this branch costs 64 phi copies at its end
        double d;
        asm volatile("v_cvt_f64_f32 %0, %2" : "=v"(d), "+v"(sq) : "v"(v[k]));
        sq += d * d;
      }
      sq = block_sum(sq, shd);
      const float x_rms = fmaxf((float)sqrt(sq / T_raw), 1e-5f);
      g = stats[b * 4 + 2] / x_rms;
    }
    // scaled in place once (g = 1 without keep_rms: x * 1 is x): the same product feeds the peak and the output
#pragma unroll
    for (int k = 0; k < NREG; k++) v[k] = v[k] * g;
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < NREG; k++) mx = fmaxf(mx, fabsf(v[k]));
    mx = block_max(mx, shf);
    const bool div = peak_guard && mx > 1.0f;  // universe.py:356-357
    if (div) {
      so = 0;
#pragma unroll
      for (int k = 0; k < NREG; k++) {
        buf_store(v[k] / mx, ro, tid * 4, so);  // (dropped past the end)
        asm volatile("s_add_u32 %0, %0, 0x1000" : "+s"(so) : : "scc");
        __builtin_amdgcn_sched_barrier(0);      // one division's worth of temporaries at a time
      }
    } else {
      so = 0;
#pragma unroll
      for (int k = 0; k < NREG; k++) {
        buf_store(v[k], ro, tid * 4, so);
        asm volatile("s_add_u32 %0, %0, 0x1000" : "+s"(so) : : "scc");
      }
    }
  }
}
__global__ __launch_bounds__(1024) void post_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                    float* __restrict__ out, int T_raw, int T_pad, int pad_left,
                                                    int keep_rms, int peak_guard) {
  __shared__ double shd[16];
  __shared__ float shf[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (size_t)b * T_pad + pad_left;
  float g = 1.f;
  if (keep_rms) {  // universe.py:352-354
    double sq = 0;
    for (int t = tid; t < T_raw; t += 1024) {
      double v = xb[t];
      sq += v * v;
    }
    sq = block_sum(sq, shd);
    float x_rms = fmaxf((float)sqrt(sq / T_raw), 1e-5f);
    g = stats[b * 4 + 2] / x_rms;
  }
  float mx = 0.f;
#pragma unroll 8
  for (int t = tid; t < T_raw; t += 1024) mx = fmaxf(mx, fabsf(xb[t] * g));
  mx = block_max(mx, shf);
  const bool div = peak_guard && mx > 1.0f;  // universe.py:356-357
#pragma unroll 8
  for (int t = tid; t < T_raw; t += 1024) {
    float v = xb[t];
    if (keep_rms) v = v * g;
    if (div) v = v / mx;
    out[(size_t)b * T_raw + t] = v;
  }
}
hipError_t launch_post(const float* x, const float* stats, float* out, int B, int T_raw, int T_pad, int pad_left,
                       int keep_rms, int peak_guard, hipStream_t st) {
  if (T_raw <= POST_NREG * 1024)
    hipLaunchKernelGGL(post_reg_kernel, dim3(B), dim3(1024), 0, st, x, stats, out, T_raw, T_pad, pad_left, keep_rms,
                       peak_guard);
  else
    hipLaunchKernelGGL(post_kernel, dim3(B), dim3(1024), 0, st, x, stats, out, T_raw, T_pad, pad_left, keep_rms,
                       peak_guard);
  return hipGetLastError();
}

__global__ void init_x_kernel(const float* __restrict__ noise, const float* base, float sigma, float* __restrict__ x,
                              size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = noise[i] * sigma;  // universe.py:39-41
  x[i] = base ? base[i] + v : v;
}
hipError_t launch_init_x(const float* noise, const float* base, float sigma, float* x, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(init_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, noise, base, sigma, x, n);
  return hipGetLastError();
}

__global__ void sampler_step_kernel(float* __restrict__ x, const float* __restrict__ score, const float* z, float c1,
                                    float c2, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float r = x[i] + c1 * score[i];  // universe.py:339 / :343, same association as the fused update in out_conv_kernel
  if (z) r = r + c2 * z[i];
  x[i] = r;
}
hipError_t launch_sampler_step(float* x, const float* score, const float* z, float c1, float c2, size_t n,
                               hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, score, z, c1, c2, n);
  return hipGetLastError();
}

// ---- CompressedMagSTFT (layers/dyn_range_comp.py:51-225) --------------------------------------------------------
// Signal pre-conditioning transform of the non-shipped STFT-domain configs: STFT (center=True, zero padding,
// onesided) -> magnitude compression -> (real | imag) stacked as channels; and its inverse (expansion -> iSTFT =
// windowed overlap-add / window envelope, torch.istft semantics).  n_fft is arbitrary (510 in the reference's
// experiments): direct DFT with an exact (k*n mod N) twiddle table in LDS, one block per (frame, batch).
__device__ __forceinline__ void stft_twiddles(float* tc, float* ts, int n_fft) {
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    float sn, cs;
    sincospif(2.0f * (float)n / (float)n_fft, &sn, &cs);
    tc[n] = cs;
    ts[n] = sn;
  }
}
// (re, im) -> compressed (re, im).  dyn_range_comp.py:117-131
__device__ __forceinline__ void spec_compress(float& re, float& im, int type, float e, float factor) {
  if (type == 1) {        // "exponent": (1e-7 + |s|)^(e - 1) * s * factor
    if (e != 1.0f) {
      const float g = powf(1e-7f + sqrtf(re * re + im * im), e - 1.0f);
      re *= g; im *= g;
    }
    re *= factor; im *= factor;
  } else if (type == 2) {  // "log": log(1 + |s|) * sgn(s) * factor
    const float mag = sqrtf(re * re + im * im);
    const float g = mag > 0.f ? log1pf(mag) / mag : 0.f;
    re *= g * factor; im *= g * factor;
  }
}
// dyn_range_comp.py:133-145
__device__ __forceinline__ void spec_expand(float& re, float& im, int type, float e, float factor) {
  if (type == 1) {
    re /= factor; im /= factor;
    if (e != 1.0f) {
      const float g = powf(1e-7f + sqrtf(re * re + im * im), 1.0f / e - 1.0f);
      re *= g; im *= g;
    }
  } else if (type == 2) {
    re /= factor; im /= factor;
    const float mag = sqrtf(re * re + im * im);
    const float g = mag > 0.f ? expm1f(mag) / mag : 0.f;
    re *= g; im *= g;
  }
}

__global__ __launch_bounds__(256) void stft_forward_kernel(const float* __restrict__ x, const float* __restrict__ win,
                                                           float* __restrict__ out, int T, int n_fft, int hop, int F,
                                                           int n_frames, int type, float e, float factor) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sx = sm;            // [n_fft] windowed frame
  float* tc = sx + n_fft;    // [n_fft]
  float* ts = tc + n_fft;    // [n_fft]
  const int f = blockIdx.x, b = blockIdx.y;
  const int pad = n_fft / 2;  // center=True, pad_mode="constant"
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    const int t = f * hop + n - pad;
    sx[n] = ((t >= 0 && t < T) ? x[(size_t)b * T + t] : 0.f) * win[n];
  }
  stft_twiddles(tc, ts, n_fft);
  __syncthreads();
  for (int k = threadIdx.x; k < F; k += blockDim.x) {
    float re = 0.f, im = 0.f;
    int idx = 0;
    for (int n = 0; n < n_fft; n++) {
      const float v = sx[n];
      re = fmaf(v, tc[idx], re);
      im = fmaf(-v, ts[idx], im);
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
    }
    spec_compress(re, im, type, e, factor);
    out[((size_t)b * 2 * F + k) * n_frames + f] = re;        // (batch, real/imag, freq, frame), dyn_range_comp.py:91-95
    out[((size_t)b * 2 * F + F + k) * n_frames + f] = im;
  }
}

// expansion + inverse real DFT + synthesis window of one frame -> frames[b][f][n]
__global__ __launch_bounds__(256) void stft_inverse_frames_kernel(const float* __restrict__ spec,
                                                                  const float* __restrict__ win,
                                                                  float* __restrict__ frames, int n_fft, int F,
                                                                  int n_frames, int type, float e, float factor) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sr = sm;          // [F]
  float* si = sr + F;      // [F]
  float* tc = si + F;      // [n_fft]
  float* ts = tc + n_fft;  // [n_fft]
  const int f = blockIdx.x, b = blockIdx.y;
  for (int k = threadIdx.x; k < F; k += blockDim.x) {
    float re = spec[((size_t)b * 2 * F + k) * n_frames + f];
    float im = spec[((size_t)b * 2 * F + F + k) * n_frames + f];
    spec_expand(re, im, type, e, factor);
    sr[k] = re;
    si[k] = im;
  }
  stft_twiddles(tc, ts, n_fft);
  __syncthreads();
  const bool even = (n_fft & 1) == 0;
  const int kmax = even ? F - 1 : F;  // bins 1 .. kmax-1 appear twice (conjugate symmetry)
  const float inv_n = 1.0f / (float)n_fft;
  for (int n = threadIdx.x; n < n_fft; n += blockDim.x) {
    float acc = sr[0];  // the imaginary parts of the DC and Nyquist bins are ignored (c2r transform)
    if (even) acc += (n & 1) ? -sr[F - 1] : sr[F - 1];
    float s2 = 0.f;
    int idx = n;  // (k * n) mod N for k = 1, 2, ...
    for (int k = 1; k < kmax; k++) {
      s2 = fmaf(sr[k], tc[idx], s2);
      s2 = fmaf(-si[k], ts[idx], s2);
      idx += n;
      if (idx >= n_fft) idx -= n_fft;
    }
    frames[((size_t)b * n_frames + f) * n_fft + n] = (acc + 2.0f * s2) * inv_n * win[n];
  }
}
// overlap-add / window envelope, trimmed like torch.istft(center=True, length=...)
__global__ __launch_bounds__(256) void stft_overlap_add_kernel(const float* __restrict__ frames,
                                                               const float* __restrict__ win, float* __restrict__ y,
                                                               int n_fft, int hop, int n_frames, int length) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= length) return;
  const int tp = t + n_fft / 2;  // position in the centred (padded) signal
  int f_hi = tp / hop;
  if (f_hi > n_frames - 1) f_hi = n_frames - 1;
  int f_lo = (tp - n_fft + hop) / hop;  // ceil((tp - n_fft + 1) / hop)
  if (tp - n_fft + 1 <= 0) f_lo = 0;
  float acc = 0.f, env = 0.f;
  for (int f = f_lo; f <= f_hi; f++) {
    const int n = tp - f * hop;
    if (n < 0 || n >= n_fft) continue;
    acc += frames[((size_t)b * n_frames + f) * n_fft + n];
    const float w = win[n];
    env = fmaf(w, w, env);
  }
  y[(size_t)b * length + t] = env > 1e-11f ? acc / env : 0.f;
}

hipError_t launch_stft_forward(const float* x, const float* win, float* out, int B, int T, int n_fft, int hop,
                               int type, float e, float factor, hipStream_t st) {
  if (n_fft < 2 || n_fft > 8192 || hop < 1 || T < 1) return hipErrorInvalidValue;
  const int F = n_fft / 2 + 1, n_frames = 1 + (T + 2 * (n_fft / 2) - n_fft) / hop;
  hipLaunchKernelGGL(stft_forward_kernel, dim3(n_frames, B), dim3(256), (size_t)3 * n_fft * 4, st, x, win, out, T, n_fft,
                     hop, F, n_frames, type, e, factor);
  return hipGetLastError();
}
hipError_t launch_stft_inverse(const float* spec, const float* win, float* frames, float* y, int B, int n_frames,
                               int n_fft, int hop, int type, float e, float factor, int length, hipStream_t st) {
  if (n_fft < 2 || n_fft > 8192 || hop < 1 || n_frames < 1 || length < 1) return hipErrorInvalidValue;
  const int F = n_fft / 2 + 1;
  hipLaunchKernelGGL(stft_inverse_frames_kernel, dim3(n_frames, B), dim3(256), (size_t)(2 * F + 2 * n_fft) * 4, st, spec,
                     win, frames, n_fft, F, n_frames, type, e, factor);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(stft_overlap_add_kernel, dim3((length + 255) / 256, B), dim3(256), 0, st, frames, win, y, n_fft, hop,
                     n_frames, length);
  return hipGetLastError();
}

// ---- mel front-end -------------------------------------------------------------------------------------------
// One block per (frame, batch).  n_fft is 640 / 960 (not a power of two): direct DFT with an exact
// (k*n mod N) twiddle table in LDS; 0.33 GFLOP per utterance, once per enhance call.
__global__ __launch_bounds__(512) void mel_kernel(const float* __restrict__ x, const float* __restrict__ win,
                                                  const float* __restrict__ tw, const float* __restrict__ fb,
                                                  float* __restrict__ mel, float* __restrict__ esum, int T, int n_fft,
                                                  int hop, int pad_left, int n_freq, int n_mels, int L) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sx = sm;                // [n_fft] windowed frame
  float* tc = sx + n_fft;        // [n_fft] cos
  float* ts = tc + n_fft;        // [n_fft] sin
  float* pw = ts + n_fft;        // [n_freq]
  __shared__ float shf[8];
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  for (int n = tid; n < n_fft; n += 512) {
    int t = f * hop + n - pad_left;  // condition.py:98 padding
    float v = (t >= 0 && t < T) ? x[(size_t)b * T + t] : 0.f;
    sx[n] = v * win[n];
    tc[n] = tw[n];
    ts[n] = tw[n_fft + n];
  }
  __syncthreads();
  for (int k = tid; k < n_freq; k += 512) {
    float re = 0.f, im = 0.f;
    int idx = 0;
    for (int n = 0; n < n_fft; n++) {
      float v = sx[n];
      re = fmaf(v, tc[idx], re);
      im = fmaf(-v, ts[idx], im);
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
    }
    pw[k] = re * re + im * im;  // power spectrogram
  }
  __syncthreads();
  float e = 0.f;
  for (int m = tid; m < n_mels; m += 512) {
    float acc = 0.f;
    for (int k = 0; k < n_freq; k++) acc = fmaf(pw[k], fb[(size_t)k * n_mels + m], acc);
    mel[((size_t)b * n_mels + m) * L + f] = acc;
    e += acc * acc;
  }
  e = block_sum(e, shf);
  if (tid == 0) esum[(size_t)b * L + f] = e;
}
hipError_t launch_mel(const float* x, const float* win, const float* tw, const float* fb, float* mel, float* esum,
                      int B, int T, int n_fft, int hop, int pad_left, int n_freq, int n_mels, int L, hipStream_t st) {
  size_t smem = (size_t)(3 * n_fft + n_freq) * 4;
  hipLaunchKernelGGL(mel_kernel, dim3(L, B), dim3(512), smem, st, x, win, tw, fb, mel, esum, T, n_fft, hop, pad_left,
                     n_freq, n_mels, L);
  return hipGetLastError();
}
// condition.py:105-106: scale = 1 / max(sqrt(mean_frames(sum_mel mel^2)), 1e-5)
__global__ __launch_bounds__(256) void mel_scale_kernel(const float* __restrict__ esum, float* scale, int L) {
  __shared__ double shd[4];
  const int b = blockIdx.x;
  double s = 0;
  for (int f = threadIdx.x; f < L; f += 256) s += esum[(size_t)b * L + f];
  s = block_sum(s, shd);
  if (threadIdx.x == 0) scale[b] = 1.0f / fmaxf((float)sqrt(s / L), 1e-5f);
}
hipError_t launch_mel_scale(const float* esum, float* scale, int B, int L, hipStream_t st) {
  hipLaunchKernelGGL(mel_scale_kernel, dim3(B), dim3(256), 0, st, esum, scale, L);
  return hipGetLastError();
}

// ---- space-to-depth + PReLU ------------------------------------------------------------------------------------
constexpr int S2D_QB = 32;
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, const float* alpha,
                                                  float* __restrict__ y, int C, int T, int R) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int Nq = T / R;
  const int q0 = blockIdx.x * S2D_QB, ci = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int nq = min(S2D_QB, Nq - q0);
  const float a = *alpha;
  const float* xr = x + ((size_t)b * C + ci) * T + (size_t)q0 * R;
  const int n = nq * R;
  for (int j = tid; j < n; j += 256) sm[j + (j >> 5)] = prelu(xr[j], a);
  __syncthreads();
  float* yb = y + ((size_t)b * C * R + (size_t)ci * R) * Nq + q0;
  for (int e = tid; e < R * S2D_QB; e += 256) {
    int k = e / S2D_QB, q = e % S2D_QB;
    if (q < nq) {
      int j = q * R + k;
      yb[(size_t)k * Nq + q] = sm[j + (j >> 5)];
    }
  }
}
hipError_t launch_s2d(const float* x, const float* alpha, float* y, int B, int C, int T, int R, hipStream_t st) {
  if (T % R) return hipErrorInvalidValue;
  int Nq = T / R;
  size_t n = (size_t)S2D_QB * R;
  size_t smem = (n + (n >> 5) + 1) * 4;
  hipLaunchKernelGGL(s2d_kernel, dim3((Nq + S2D_QB - 1) / S2D_QB, C, B), dim3(256), smem, st, x, alpha, y, C, T, R);
  return hipGetLastError();
}

// ---- binomial anti-alias FIR ------------------------------------------------------------------------------------
constexpr int FIR_TILE = 1024;
__global__ __launch_bounds__(256) void fir_kernel(const float* __restrict__ x, const float* __restrict__ taps, int ntaps,
                                                  float alpha, int act, const float* __restrict__ bias,
                                                  const float* res, float res_scale, float* __restrict__ y, int C,
                                                  int T) {
  __shared__ float tile[FIR_TILE + 40];
  __shared__ float tp[40];
  const int c = blockIdx.y, b = blockIdx.z, t0 = blockIdx.x * FIR_TILE, tid = threadIdx.x;
  const int r = ntaps >> 1;
  const size_t row = ((size_t)b * C + c) * T;
  if (tid < ntaps) tp[tid] = taps[tid];
  for (int i = tid; i < FIR_TILE + 2 * r; i += 256) {
    int t = t0 + i - r;
    float v = (t >= 0 && t < T) ? x[row + t] : 0.f;
    if (act) v = v >= 0.f ? v : alpha * v;
    tile[i] = v;
  }
  __syncthreads();
  const float bb = bias ? bias[c] : 0.f;
#pragma unroll
  for (int k = 0; k < FIR_TILE / 256; k++) {
    const int i = tid + k * 256, t = t0 + i;
    if (t >= T) break;
    float acc = 0.f;
    for (int j = 0; j < ntaps; j++) acc = fmaf(tp[j], tile[i + j], acc);
    acc += bb;
    if (res) acc = (acc + res[row + t]) * res_scale;
    y[row + t] = acc;
  }
}
// The same pass with 16-byte accesses (<= 17 taps; global dwordx4 needs dword alignment only, so any row length): a thread loads one float4 of the tile,
// filters FOUR consecutive outputs from a register window read from LDS with aligned 16-byte reads, and stores one float4
// (+ one float4 of the residual).  Same tap order per output: bit-identical to fir_kernel.  At batch 8 the scalar form moves
// 130 MB per launch at 4.0 TB/s.
template <int NT>
__global__ __launch_bounds__(256) void fir4_kernel(const float* __restrict__ x, const float* __restrict__ taps, float alpha,
                                                   int act, const float* __restrict__ bias, const float* res,
                                                   float res_scale, float* __restrict__ y, int C, int T) {
  constexpr int R = NT >> 1, WIN = 4 + NT - 1, NW4 = (WIN + 3) / 4;
  __shared__ __attribute__((aligned(16))) float tile[FIR_TILE + 2 * R + 8];
  const int c = blockIdx.y, b = blockIdx.z, t0 = blockIdx.x * FIR_TILE, tid = threadIdx.x;
  const size_t row = ((size_t)b * C + c) * T;
  float tp[NT];
#pragma unroll
  for (int j = 0; j < NT; j++) tp[j] = taps[j];
  // tile[i] = prelu(x[t0 + i - R]); the main part with one float4 per thread, the 2 R halo samples by the first threads
  {
    const int t = t0 + 4 * tid;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t + 3 < T) v = *reinterpret_cast<const f32x4u*>(x + row + t);
    else {
#pragma unroll
      for (int e = 0; e < 4; e++) if (t + e < T) v[e] = x[row + t + e];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float u = v[e];
      if (act) u = u >= 0.f ? u : alpha * u;
      tile[R + 4 * tid + e] = u;
    }
    if (tid < 2 * R) {
      const int i = tid < R ? tid : FIR_TILE + tid;  // tile index: R samples in front, R behind
      const int th = t0 + i - R;
      float u = (th >= 0 && th < T) ? x[row + th] : 0.f;
      if (act) u = u >= 0.f ? u : alpha * u;
      tile[i] = u;
    }
  }
  __syncthreads();
  const int t = t0 + 4 * tid;
  if (t >= T) return;
  f32x4 w4[NW4];
#pragma unroll
  for (int q = 0; q < NW4; q++) w4[q] = *reinterpret_cast<const f32x4*>(&tile[4 * tid + 4 * q]);
  const float bb = bias ? bias[c] : 0.f;
  f32x4 rs = {0.f, 0.f, 0.f, 0.f};
  const bool full = t + 3 < T;
  if (res) {
    if (full) rs = *reinterpret_cast<const f32x4u*>(res + row + t);
    else {
#pragma unroll
      for (int e = 0; e < 4; e++) if (t + e < T) rs[e] = res[row + t + e];
    }
  }
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < NT; j++) acc = fmaf(tp[j], w4[(e + j) >> 2][(e + j) & 3], acc);
    acc += bb;
    if (res) acc = (acc + rs[e]) * res_scale;
    o[e] = acc;
  }
  if (full) *reinterpret_cast<f32x4u*>(y + row + t) = o;
  else {
#pragma unroll
    for (int e = 0; e < 4; e++) if (t + e < T) y[row + t + e] = o[e];
  }
}
hipError_t launch_fir(const float* x, const float* taps, int ntaps, float alpha, int act, const float* bias,
                      const float* res, float res_scale, float* y, int B, int C, int T, hipStream_t st) {
  if (ntaps > 39 || !(ntaps & 1)) return hipErrorInvalidValue;
  const dim3 grid((T + FIR_TILE - 1) / FIR_TILE, C, B);
  static const bool wide = [] { const char* e = getenv("OU_FIR_WIDE"); return !e || atoi(e) != 0; }();
  if (wide) {  // (dwordx4 accesses at dword alignment: any T)
    void (*k)(const float*, const float*, float, int, const float*, const float*, float, float*, int, int) = nullptr;
    switch (ntaps) {
      case 5: k = fir4_kernel<5>; break;
      case 7: k = fir4_kernel<7>; break;
      case 9: k = fir4_kernel<9>; break;
      case 11: k = fir4_kernel<11>; break;
      case 17: k = fir4_kernel<17>; break;
      default: break;
    }
    if (k) {
      hipLaunchKernelGGL(k, grid, dim3(256), 0, st, x, taps, alpha, act, bias, res, res_scale, y, C, T);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(fir_kernel, grid, dim3(256), 0, st, x, taps, ntaps, alpha, act, bias, res, res_scale, y, C, T);
  return hipGetLastError();
}

__global__ void sum_kernel(const float* a, const float* b, const float* c, const float* d, const float* e, float scale,
                           float* __restrict__ y, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = a[i];
  if (b) v += b[i];
  if (c) v += c[i];
  if (d) v += d[i];
  if (e) v += e[i];
  y[i] = v * scale;
}
hipError_t launch_sum(const float* a, const float* b, const float* c, const float* d, const float* e, float scale,
                      float* y, size_t n, hipStream_t st) {
  hipLaunchKernelGGL(sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, c, d, e, scale, y, n);
  return hipGetLastError();
}

// =========================================================================================================
// GRU recurrence (torch.nn.GRU semantics; score.py:83-89,116 / condition.py:173-179,212)
//   W_hh of one direction is 3H x H fp32 (786 KB at H = 256): larger than one CU's LDS and VGPR file, so a
//   cluster of HB = H/64 workgroups (512 threads each, one per CU) keeps it resident in registers:
//   workgroup g owns hidden units [64g, 64g+64) = 192 gate rows; thread (rg, cg) holds rows of units
//   {2rg, 2rg+1} x columns {4cg + 64i + 0..3}.  Per time step: 96..144 FMAs per thread, a 16-lane DPP
//   row reduction, the gate math, then the 64 new hidden values are published to the other workgroups as
//   8-byte {step tag, value} granules with relaxed agent-scope stores (write-through to L2) and gathered
//   by one polling wave -- no fence, no flag (the tag is the flag).  Spins are bounded; a timeout raises
//   the status word instead of hanging the GPU.
// =========================================================================================================
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(t);
}
// sum over the 8 lanes of an aligned 8-lane group (every lane ends with the total)
__device__ __forceinline__ float row8_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror: lane i <-> 7-i within each 8
  return v;
}
// sum over the 16 lanes of a DPP row (every lane ends with the total)
__device__ __forceinline__ float row16_sum(float v) {
  v = row8_sum(v);
  v = dpp_add<0x140>(v);  // row_mirror: lane i <-> 15-i
  return v;
}
// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): the gate math is
// a serial chain on the critical path of every GRU time step, libm-grade expf/tanhf/division cost ~100 dependent
// instructions there.  Absolute error ~2e-7, two orders of magnitude inside the parity gate.
__device__ __forceinline__ float sigmoidf_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
  // tanh(x) = 1 - 2/(1 + e^{2x});  e^{2x} -> inf gives 1, -> 0 gives -1
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x));
}
constexpr unsigned GRU_SPIN_LIMIT = 4000000u;

// Thread mapping: a direction's H hidden units are split over NWG = H/UPW workgroups of NT threads;
// LPU = NT/UPW lanes share one unit: thread (u, cg) = (tid/LPU, tid%LPU) holds, for unit UPW*g + u, the three gate
// rows (r, z, n) x columns {4cg + 4*LPU*i + 0..3, i < NI = H/(4*LPU)} in registers (gathered from the canonical
// row-major W_hh at kernel start).  Per step: NI ds_read_b128 of h, 6*NI v_pk_fma_f32, a DPP reduction of the 3 gate
// sums over the LPU lanes of the unit, the gate math in lane cg == 0, publish.
// Variants (chosen by launch_gru from the batch size): <64 units, 512 thr> = 4 workgroups per direction at H = 256
// (throughput: 8 CUs per utterance), <32, 512> = 8, <16, 256> = 16 (latency: one wave per SIMD, least work per step).
template <int HB, int UPW, int NT>
__global__ __launch_bounds__(NT) void gru_cluster_kernel(GruArgs p, int nclusters) {
  constexpr int H = 64 * HB, LPU = NT / UPW, NI = H / (4 * LPU), NR = 12 * NI, NWG = H / UPW;
  static_assert(LPU == 8 || LPU == 16, "8 or 16 lanes per hidden unit");
  static_assert(H % (4 * LPU) == 0, "column blocks");
  __shared__ __attribute__((aligned(16))) float hbuf[2][H];
  __shared__ int abort_flag;
  const int tid = threadIdx.x, lane = tid & 63;
  // Workgroup -> (cluster, member): the dispatcher places block i on XCD i % 8 (observed, speed only), so the
  // NWG members of a cluster are given ids that are congruent mod 8 and share one L2.  Correctness does not
  // depend on it: the exchange below is agent-scope.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int cluster = xcd + 8 * (slot / NWG);
  const int g = slot % NWG;
  if (cluster >= nclusters) return;
  const int dir = cluster & 1, b = cluster >> 1;
  const int ul = tid / LPU, cg = tid % LPU;
  const int unit = g * UPW + ul;
  const int T = p.T;

  f32x2 w[NR / 2];
  {
    const float* wd = p.whh + (size_t)dir * 3 * H * H;
#pragma unroll
    for (int gt = 0; gt < 3; gt++)
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const float4 v = *reinterpret_cast<const float4*>(wd + (size_t)(gt * H + unit) * H + cg * 4 + 4 * LPU * i);
        w[(gt * NI + i) * 2] = f32x2{v.x, v.y};
        w[(gt * NI + i) * 2 + 1] = f32x2{v.z, v.w};
      }
  }
  for (int i = tid; i < 2 * H; i += NT) (&hbuf[0][0])[i] = 0.f;
  if (tid == 0) abort_flag = 0;

  const bool fin = cg == 0;
  const float bhn = p.bhn[dir * H + unit];
  const float* gxb = p.gx + ((size_t)b * 6 * H + (size_t)dir * 3 * H) * T;
  const float* gx_r = gxb + (size_t)unit * T;
  const float* gx_z = gxb + (size_t)(H + unit) * T;
  const float* gx_n = gxb + (size_t)(2 * H + unit) * T;
  const size_t orow = ((size_t)b * 2 * H + (size_t)dir * H + unit) * T;
  unsigned long long* xq = p.xchg + ((size_t)(b * 2 + dir) * 2) * H;
  const bool has_res = p.res != nullptr;

  // One-time rendezvous: every member posts the id of the XCD it runs on (granule g*UPW of buffer 0, which is not
  // written again before all members have passed step 0).  When the whole cluster shares one XCD -- the normal case,
  // see the block mapping above -- the per-step publishes can be ordinary stores: the vector L1 is write-through, so
  // they land in the L2 that serves every poller's sc1 (agent-scope) load, without the write-through to the memory
  // side that an agent-scope store adds (measured: -6 % per GRU launch, and finer splits stop losing to store
  // traffic).  Anything else keeps agent-scope stores.
  __shared__ int plain_flag;
  if (NWG > 1 && tid < 64) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xFu;
    constexpr unsigned RTAG = 0x80000000u;
    if (lane == 0)
      __hip_atomic_store(xq + (size_t)g * UPW, ((unsigned long long)RTAG << 32) | xcc, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    bool same = true, fail = false;
    if (lane < NWG) {
      unsigned spins = 0;
      unsigned long long v;
      while (true) {
        v = __hip_atomic_load(xq + (size_t)lane * UPW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 32) == RTAG) break;
        if (++spins > GRU_SPIN_LIMIT) { fail = true; break; }
      }
      same = !fail && (unsigned)v == xcc;
    }
    const bool all_same = __builtin_amdgcn_ballot_w64(!same) == 0ull;
    if (lane == 0) plain_flag = (all_same && !p.agent_stores) ? 1 : 0;
    if (fail) atomicOr(p.err, 1u);
  }

  int t = dir ? T - 1 : 0;
  const int dt = dir ? -1 : 1;
  float xr = 0.f, xz = 0.f, xn = 0.f, rs = 0.f;
  if (fin) {
    xr = gx_r[t]; xz = gx_z[t]; xn = gx_n[t];
    if (has_res) rs = p.res[orow + t];
  }
  // Everything loaded so far (weights, bhn, first-step inputs) is first USED inside the loop; without this the
  // compiler places a vmcnt(0) wait at that first use -- in every iteration, right behind the prefetch loads
  // issued there, which exposes a full memory latency per time step.  vmcnt(0), expcnt/lgkmcnt untouched:
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  const bool plain = NWG > 1 && __builtin_amdgcn_readfirstlane(plain_flag) != 0;

  long long c_comp = 0, c_poll = 0, c_bar = 0, c_mv = 0, c_red = 0, c_gate = 0;
  const bool ts_on = p.tstamps != nullptr;
  for (int step = 0; step < T; step++, t += dt) {
    const int cur = step & 1;
    long long q0 = 0, q1 = 0, q2 = 0;
    if (ts_on) q0 = __builtin_readcyclecounter();
    const float hp = hbuf[cur][unit];  // read with the matvec operands, off the gate chain
    f32x2 acc[3][2];
#pragma unroll
    for (int gt = 0; gt < 3; gt++) { acc[gt][0] = 0.f; acc[gt][1] = 0.f; }
#pragma unroll
    for (int i = 0; i < NI; i++) {
      const float4 hv = *reinterpret_cast<const float4*>(&hbuf[cur][cg * 4 + 4 * LPU * i]);
      const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
#pragma unroll
      for (int gt = 0; gt < 3; gt++) {
        const int r = (gt * NI + i) * 2;
        acc[gt][0] = __builtin_elementwise_fma(w[r], h01, acc[gt][0]);
        acc[gt][1] = __builtin_elementwise_fma(w[r + 1], h23, acc[gt][1]);
      }
    }
    long long qa = 0, qb = 0;
    if (ts_on) qa = __builtin_readcyclecounter();
    float hs[3];
#pragma unroll
    for (int gt = 0; gt < 3; gt++) {
      const float part = (acc[gt][0].x + acc[gt][0].y) + (acc[gt][1].x + acc[gt][1].y);
      hs[gt] = LPU == 8 ? row8_sum(part) : row16_sum(part);
    }

    if (ts_on) qb = __builtin_readcyclecounter();
    // next step's input-projection / residual values: issued now, consumed one iteration later, so that no
    // global-load latency ever sits between the gate math and the publish below
    float nxr = 0.f, nxz = 0.f, nxn = 0.f, nrs = 0.f;
    if (fin && step + 1 < T) {
      nxr = gx_r[t + dt]; nxz = gx_z[t + dt]; nxn = gx_n[t + dt];
      if (has_res) nrs = p.res[orow + t + dt];
    }

    if (fin) {
      const float r = sigmoidf_(xr + hs[0]);
      const float z = sigmoidf_(xz + hs[1]);
      const float n = tanhf_(xn + r * (hs[2] + bhn));
      const float hnew = (hp - n) * z + n;
      if (NWG > 1) {  // publish first: the other workgroups are waiting on this
        unsigned long long gran = ((unsigned long long)(unsigned)(step + 1) << 32) | (unsigned)__float_as_int(hnew);
        unsigned long long* dst = xq + (size_t)(cur ^ 1) * H + unit;
        if (plain) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(gran) : "memory");
        else __hip_atomic_store(dst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      hbuf[cur ^ 1][unit] = hnew;
      p.out[orow + t] = has_res ? (hnew + rs) * p.res_scale : hnew;
    }
    xr = nxr; xz = nxz; xn = nxn; rs = nrs;
    if (ts_on) { q1 = __builtin_readcyclecounter(); c_mv += qa - q0; c_red += qb - qa; c_gate += q1 - qb; }

    if (NWG > 1 && tid < 64) {
      // gather the other workgroups' slices: lane l polls granules l, l+64, ...; all polls in flight together
      const unsigned tag = (unsigned)(step + 1);
      unsigned long long* src = xq + (size_t)(cur ^ 1) * H + lane;
      unsigned long long v[HB];
      unsigned spins = 0;
      if (p.poll_backoff > 0) __builtin_amdgcn_s_sleep(8);   // ~512 cycles: nothing can have arrived yet
      if (p.poll_backoff > 1) __builtin_amdgcn_s_sleep(8);
      while (true) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < HB; k++) {
          const bool own = (k * 64 + lane) / UPW == g;
          v[k] = own ? ((unsigned long long)tag << 32)
                     : __hip_atomic_load(src + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((unsigned)(v[k] >> 32) == tag);
        }
        if (ok) break;
        if (++spins > GRU_SPIN_LIMIT) { abort_flag = 1; break; }
        // safety net as in gru_ring_kernel: a plain publish has no visibility deadline; after a long wait (~1 ms; the gate
        // lanes have written this step's values to LDS long before) repeat this workgroup's own granules as system-scope
        // write-through stores
        if (plain && (spins & 1023u) == 1023u) {
#pragma unroll
          for (int k = 0; k < HB; k++)
            if ((k * 64 + lane) / UPW == g) {
              const float hv = *reinterpret_cast<volatile float*>(&hbuf[cur ^ 1][k * 64 + lane]);
              const unsigned long long gran = ((unsigned long long)tag << 32) | (unsigned)__float_as_int(hv);
              unsigned long long* dst = src + k * 64;
              asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(gran) : "memory");
            }
        }
      }
#pragma unroll
      for (int k = 0; k < HB; k++)
        if ((k * 64 + lane) / UPW != g) hbuf[cur ^ 1][k * 64 + lane] = __int_as_float((int)(unsigned)v[k]);
    }
    if (ts_on) q2 = __builtin_readcyclecounter();
    __syncthreads();
    if (ts_on) { long long q3 = __builtin_readcyclecounter(); c_comp += q1 - q0; c_poll += q2 - q1; c_bar += q3 - q2; }
    if (NWG > 1 && abort_flag) {
      if (tid == 0) atomicOr(p.err, 1u);
      break;
    }
  }
  if (ts_on && lane == 0) {
    long long* o = p.tstamps + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
    o[0] = c_comp; o[1] = c_poll; o[2] = c_bar; o[3] = T; o[4] = c_mv; o[5] = c_red; o[6] = c_gate;
  }
}

// ---------------------------------------------------------------------------------------------------------
// GRU recurrence, second generation ("ring"): same cluster decomposition, different exchange.
//   * every WAVE gathers the h columns its lanes need straight from L2 into registers (volatile agent-scope 16-byte
//     buffer loads = two {value, tag} granules each): no polling wave, no LDS hop, no workgroup barrier -- the four
//     waves of a workgroup run unsynchronised, each ordered only by the tags it reads;
//   * tags never repeat: tag(step s of this launch) = epoch + s with a device-side epoch that the last block of a
//     launch advances by T + 1 (graph-replay safe, no per-launch memset).  Stale granules therefore always carry
//     SMALLER tags, so "all tags arrived" is one v_min3 tree + one compare;
//   * granule = {h, tag}: the matvec multiplies a (W_r, W_z) row pair by the granule's low half with one v_pk_fma_f32
//     (op_sel_hi = 0 on the h operand) and W_n by a scalar FMA -- the loaded registers are the FMA operands, no
//     repacking;
//   * the one-time XCD rendezvous uses the cluster's own rendezvous granules with tag = epoch.
// Double buffering by step parity is enough without barriers: h_{s+2} overwrites h_s only after its writer has read all of
// h_{s+1}, and every wave publishes its part of h_{s+1} only after it has finished reading h_s.
// ---------------------------------------------------------------------------------------------------------
// s_waitcnt vmcnt(0) that the register allocator sees as the producer of the gathered registers (the loads themselves are
// inline asm, invisible to the compiler's own wait-count insertion)
template <int N>
__device__ __forceinline__ void gather_wait(u32x4 (&hv)[N]) {
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]));
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]));
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]));
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]));
  else if constexpr (N == 8)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]),
                 "+v"(hv[6]), "+v"(hv[7]));
  else if constexpr (N == 12)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]),
                 "+v"(hv[6]), "+v"(hv[7]), "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]));
  else if constexpr (N == 16)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]),
                 "+v"(hv[6]), "+v"(hv[7]), "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]), "+v"(hv[12]),
                 "+v"(hv[13]), "+v"(hv[14]), "+v"(hv[15]));
  else static_assert(N == 2, "gather_wait: unsupported register count");
}

// First-recovery record, out of line so that the hot loop does not change: when a hand-off has not arrived after 256 poll
// rounds, look at the first stale granule of this lane's columns once more with three kinds of loads and leave what they
// return -- and where this wave runs now vs. at the rendezvous -- in status words 21..29 (first event of a workspace only).
// Reading the record: sc1 == want            -> the publish was only late (a member was not scheduled / not resident);
//                     sc1 != want, atomic == want (or sc0 sc1 == want) -> the line sits where an L2-served agent-scope load of
//                                               THIS CU does not see it: the writer or the reader is not on the cluster's XCD
//                                               any more (xcc now != xcc at the rendezvous), e.g. after a context save / restore;
//                     nothing == want         -> the writer has not stored it: look at that member's own wait record.
__device__ __attribute__((noinline)) void gru_stale_probe(const unsigned long long* buf, int col0, int ncol, int lstride,
                                                          unsigned want, unsigned* err, unsigned who, unsigned xcc_then,
                                                          unsigned step, int grp) {
  for (int i = 0; i < ncol; i++) {
    const int goff = col0 + (i / grp) * lstride + (i % grp);
    const unsigned long long* g = buf + goff;
    u32x2 a, b, c;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(g) : "memory");
    if (a.y == want) continue;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(b) : "v"(g) : "memory");
    const unsigned long long v = __hip_atomic_fetch_or(const_cast<unsigned long long*>(g), 0ull, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
    c = u32x2{(unsigned)v, (unsigned)(v >> 32)};
    if (atomicAdd(err + 21, 1u) == 0u) {
      unsigned now;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(now));
      err[22] = who; err[23] = (unsigned)goff; err[24] = want;
      err[25] = a.y; err[26] = b.y; err[27] = c.y;
      asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(g) : "memory");
      err[28] = a.y;
      err[29] = (step << 16) | ((xcc_then & 0xFFu) << 8) | (now & 0xFFu);
    }
    return;
  }
}

// Rare-path bookkeeping of the ring kernel, out of line and with few arguments: inlined, the diagnostics below cost the hot
// loop 10 SGPRs (106 -> spills to VGPR lanes, 17 v_readlane in the step loop) and 20 VGPRs, +25 us per 401-frame pass.
// A workgroup that has been waiting for ~2 ms leaves its position in its rendezvous slot: {epoch, step << 8 | XCC now << 4 | XCC
// at the rendezvous} -- the tag stays the epoch, late members still pass the rendezvous.
__device__ __attribute__((noinline)) void gru_note_long_wait(unsigned long long* slot, unsigned epoch, unsigned xcc_then,
                                                             unsigned step) {
  unsigned now;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(now));
  const unsigned long long pos = ((unsigned long long)epoch << 32) | ((now & 0xFu) << 4) | (xcc_then & 0xFu) | (step << 8);
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(slot), "v"(pos) : "memory");
}
// Time-out report (first reporter of a workspace only): who waited for what, and where every member of the cluster is.
//   ids = cluster << 16 | member << 8 | plain-publish flag, pos = step << 8 | XCC at the rendezvous
__device__ __attribute__((noinline)) void gru_timeout_report(unsigned* err, const unsigned long long* slots, int nwg,
                                                             unsigned epoch, unsigned ids, unsigned pos, unsigned m,
                                                             unsigned mx, unsigned want) {
  if ((atomicOr(err, 4u) & 4u) != 0u) return;
  unsigned now;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(now));
  err[11] = 0x100u | (now & 0xFu);  // has the workgroup been moved since the rendezvous? (context save / restore)
  err[12] = ids >> 16; err[13] = (ids >> 8) & 0xFFu; err[14] = pos >> 8; err[15] = m; err[16] = want;
  err[17] = pos & 0xFFu; err[18] = ids & 1u; err[19] = (unsigned)blockIdx.x;
  err[32] = mx;
  for (int i = 0; i < nwg && i < 24; i++) {  // step << 8 | xcc of every member that ever waited ~2 ms
    const unsigned long long v = __hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    err[36 + i] = (unsigned)(v >> 32) == epoch ? (unsigned)v : 0xFFFFFFFFu;
  }
}

constexpr unsigned GRU_EPOCH_WRAP = 0x7F000000u;  // past this the last block of a launch clears the exchange area

// WIDE: the gather layout with no redundant loads.  In the default layout a unit's 16 lanes hold all H columns, so the four
// units of a wave each fetch the same 2 KB (H = 256) per poll round: 8 dwordx4 loads per lane, 32 wave-loads of 16 TA cycles
// per workgroup and round -- the round is bound by the CU's address path (~500 cycles), not by the L2 latency (~200).  WIDE:
// lane l holds columns 2l, 2l + 1 (+ 128 i) for ALL four units of its wave: HB / 2 loads per lane and round; the 12 (unit,
// gate) partial sums are folded 64 -> 16 lanes by two swap levels (v_permlane32_swap / v_permlane16_swap halve the value
// count as they halve the lane count: 6 + 3 swaps), after which row r of the wave holds unit r's sums exactly as in the
// default layout.
// (Tried on top of WIDE and dropped: the four waves of a workgroup sharing the gather -- wave w polls granules [w H / 4,
// (w + 1) H / 4), 32 instead of 128 requests per line and step, and passes them on through a tag-checked LDS copy of the
// exchange buffer.  Gather 1 225 instead of 738 cycles per step: the L2 request rate is not what a poll round waits for.
// Measured for the record (OU_GRU_BACKOFF=10..14, tools/gru_ts.py): the first poll round succeeds on 92-95 % of the steps; a
// wave's two stores are acknowledged after ~230 cycles; one isolated 8-byte load, sc1 or plain, quiet or just-written line,
// takes ~320 cycles; every cycle a wave spends between its publish and its poll comes back one-to-one in everybody's step
// time -- the clusters run in lock step, the step is compute + one store latency + one load latency + the skew of 64-128 waves.)
template <int HB, int UPW, bool WIDE = false>
__global__ __launch_bounds__(256) void gru_ring_kernel(GruArgs p, int nclusters) {
  constexpr int H = 64 * HB, NT = 256, LPU = NT / UPW, NC = WIDE ? HB : H / LPU, NI = WIDE ? HB / 2 : NC / 4, NWG = H / UPW;
  constexpr int WU = UPW / 4;  // units per wave
  static_assert(!WIDE || ((UPW == 16 || UPW == 8) && HB % 2 == 0), "wide gather: 4 / 2 units per wave, whole granule pairs");
  constexpr int CSTRIDE = 2 * H + 64;  // granules per cluster: two parity buffers + rendezvous slots
  static_assert(LPU == 8 || LPU == 16 || LPU == 32, "8, 16 or 32 lanes per hidden unit");
  static_assert((WIDE || NC % 4 == 0) && NWG > 1 && NWG < 64, "column blocks / rendezvous slots (slot 63 = the mode flag)");
  const int tid = threadIdx.x, lane = tid & 63;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int cluster = xcd + 8 * (slot / NWG);
  const int g = slot % NWG;
  // tags of this launch: epoch (rendezvous), epoch + s (h after s steps).  The stored counter starts at 0 in a freshly
  // cleared workspace, whose granules carry tag 0 -> + 1
  const unsigned epoch = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int T = p.T;
  const bool ts_on = p.tstamps != nullptr;
  long long c_poll = 0, c_comp = 0, r_start = 0, r_loop = 0;
  unsigned c_rounds = 0;
  long long c_ack = 0;
  if (ts_on) r_start = (long long)__builtin_amdgcn_s_memrealtime();
  if (cluster < nclusters) {
    const int dir = cluster & 1, b = cluster >> 1;
    const int ul = tid / LPU, cg = tid % LPU;
    const int unit = g * UPW + ul;

    // this lane's weights: (W_r, W_z) row pairs and W_n for columns 4cg + 4 LPU i + {0..3}
    // (WIDE: for the wave's four units u and columns 2 lane + 128 i + {0, 1}: index u * HB + 2 i + {0, 1})
    constexpr int NW = WIDE ? WU * HB : NC;
    f32x2 wrz[NW];
    float wn[NW];
    if constexpr (WIDE) {
#pragma unroll
      for (int u = 0; u < WU; u++) {
        const float* wd = p.whh + (size_t)dir * 3 * H * H + (size_t)(g * UPW + (tid >> 6) * WU + u) * H + 2 * lane;
#pragma unroll
        for (int i = 0; i < NI; i++) {
          const float2 vr = *reinterpret_cast<const float2*>(wd + 128 * i);
          const float2 vz = *reinterpret_cast<const float2*>(wd + (size_t)H * H + 128 * i);
          const float2 vn = *reinterpret_cast<const float2*>(wd + (size_t)2 * H * H + 128 * i);
          wrz[u * HB + 2 * i] = f32x2{vr.x, vz.x}; wrz[u * HB + 2 * i + 1] = f32x2{vr.y, vz.y};
          wn[u * HB + 2 * i] = vn.x; wn[u * HB + 2 * i + 1] = vn.y;
        }
      }
    } else {
      const float* wd = p.whh + (size_t)dir * 3 * H * H + (size_t)unit * H + cg * 4;
#pragma unroll
      for (int i = 0; i < NI; i++) {
        const float4 vr = *reinterpret_cast<const float4*>(wd + 4 * LPU * i);
        const float4 vz = *reinterpret_cast<const float4*>(wd + (size_t)H * H + 4 * LPU * i);
        const float4 vn = *reinterpret_cast<const float4*>(wd + (size_t)2 * H * H + 4 * LPU * i);
        wrz[4 * i + 0] = f32x2{vr.x, vz.x}; wrz[4 * i + 1] = f32x2{vr.y, vz.y};
        wrz[4 * i + 2] = f32x2{vr.z, vz.z}; wrz[4 * i + 3] = f32x2{vr.w, vz.w};
        wn[4 * i + 0] = vn.x; wn[4 * i + 1] = vn.y; wn[4 * i + 2] = vn.z; wn[4 * i + 3] = vn.w;
      }
    }
    // the lane that ends up with the unit's gate sums: any lane of a 8 / 16-lane group (all-reduce), the upper row of a
    // 32-lane group (row_bcast:15 adds the lower row's total into the upper row only)
    const bool fin = cg == (LPU == 32 ? 16 : 0);
    const float bhn = p.bhn[dir * H + unit];
    const float* gxb = p.gx + ((size_t)b * 6 * H + (size_t)dir * 3 * H) * T;
    const size_t orow = ((size_t)b * 2 * H + (size_t)dir * H + unit) * T;
    unsigned long long* xq = p.xchg + (size_t)cluster * CSTRIDE;
    const bool has_res = p.res != nullptr;

    // one-time rendezvous (also proves that every member of the cluster is resident): member g posts {xcc, epoch}
    bool sysmode = (p.dbg & 2) != 0;  // system-scope publishes for the rest of this launch, see the safety net
    bool plain = false;
    unsigned xcc;
    {
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 0xFu;
      if (tid == 0)
        __hip_atomic_store(xq + 2 * H + g, ((unsigned long long)epoch << 32) | xcc, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      bool same = true, fail = false;
      if (lane < NWG) {
        unsigned spins = 0;
        unsigned long long v;
        while (true) {
          v = __hip_atomic_load(xq + 2 * H + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(v >> 32) == epoch) break;
          if (++spins > GRU_SPIN_LIMIT) { fail = true; break; }
        }
        same = !fail && (unsigned)v == xcc;
      }
      plain = __builtin_amdgcn_ballot_w64(!same) == 0ull && !p.agent_stores;
      if (!plain && tid == 0) atomicAdd(p.err + 30, 1u);  // diagnostics: workgroups whose cluster spans XCDs
      if (fail) { atomicOr(p.err, 2u); p.err[8] = (unsigned)cluster; p.err[9] = (unsigned)g; p.err[10] = epoch; }
    }

    // Input projections (gx rows r, z, n of this wave's units) and the residual row are staged through a wave-private
    // LDS ring in chunks of CH time steps: the global loads of chunk c + 2 are issued at the start of chunk c (coalesced
    // along time, a whole chunk of steps to land) and parked in registers, moved to LDS one chunk later and read from
    // there by the gate lanes -- the per-step loop touches global memory only for the exchange and the output store.
    constexpr int CH = 32, UW = UPW / 4;           // steps per chunk, units per wave
    constexpr int ROWS = 4 * UW, PER = ROWS * CH / 64;  // staged rows per wave (unit x {r, z, n, res}), floats per lane
    static_assert(ROWS * CH % 64 == 0 && 64 % ROWS == 0, "staging map");
    constexpr int LPR = 64 / ROWS;                 // lanes per staged row
    __shared__ float stage[4][2][ROWS][CH];
    const int wv = tid >> 6;
    const int srow = lane / LPR, sq = lane % LPR;  // this lane loads steps sq * PER .. + PER of staged row srow
    const int s_unit = g * UPW + wv * UW + (srow >> 2), s_kind = srow & 3;
    const float* s_src = s_kind == 3 ? (has_res ? p.res + ((size_t)b * 2 * H + (size_t)dir * H + s_unit) * T : nullptr)
                                     : gxb + (size_t)(s_kind * H + s_unit) * T;
    float park[PER];
    auto fetch_chunk = [&](int c) {  // -> park[]: steps c*CH + sq*PER + j
#pragma unroll
      for (int j = 0; j < PER; j++) {
        const int sidx = c * CH + sq * PER + j;
        const int tt = dir ? T - 1 - sidx : sidx;
        park[j] = (s_src && sidx < T) ? s_src[tt] : 0.f;
      }
    };
    auto park_to_lds = [&](int c) {
#pragma unroll
      for (int j = 0; j < PER; j++) stage[wv][c & 1][srow][sq * PER + j] = park[j];
    };
    fetch_chunk(0);
    park_to_lds(0);
    fetch_chunk(1);
    const int ulw = ul - wv * UW;                  // this lane's unit within its wave
    float hprev = 0.f;
    int t = dir ? T - 1 : 0;
    const int dt = dir ? -1 : 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // weights landed: no vmcnt(0) inside the loop on their account

    if (ts_on) r_loop = (long long)__builtin_amdgcn_s_memrealtime();
    bool fast_pub = plain && !sysmode;                                   // plain-store publishes
    const int inject_step = ((p.dbg & 4) && g == 1) ? 50 : -1;           // fault injection (tests)
    for (int step = 0; step < T; step++, t += dt) {
      long long q0 = 0, q1 = 0;
      if (ts_on) q0 = __builtin_readcyclecounter();
      const int cidx = step / CH, soff = step % CH;
      if (soff == 0 && step > 0) {  // chunk boundary: park -> LDS (chunk cidx), start fetching chunk cidx + 1
        park_to_lds(cidx);
        fetch_chunk(cidx + 1);
      }
      float xr = 0.f, xz = 0.f, xn = 0.f, rs = 0.f;
      if (fin) {
        const float* sp = &stage[wv][cidx & 1][ulw * 4][soff];
        xr = sp[0]; xz = sp[CH]; xn = sp[2 * CH]; rs = sp[3 * CH];
      }
      // ---- h_step: zero at step 0, else gathered from the parity buffer (all granules must carry tag epoch + step)
      u32x4 hv[NC / 2];
      if (step == 0) {
#pragma unroll
        for (int k = 0; k < NC / 2; k++) hv[k] = u32x4{0u, 0u, 0u, 0u};
      } else {
        const unsigned want = epoch + (unsigned)step;
        const unsigned long long* src = xq + (size_t)(step & 1) * H + (WIDE ? 2 * lane : cg * 4);
        unsigned spins = 0;
        // experiment (OU_GRU_BACKOFF = 6..9): nothing can have arrived right after this wave's own publish -- the first poll
        // rounds only load the L2 channels that the other members' stores have to get through
        if (p.poll_backoff >= 6) {
          if (p.poll_backoff == 6) __builtin_amdgcn_s_sleep(1);
          else if (p.poll_backoff == 7) __builtin_amdgcn_s_sleep(2);
          else if (p.poll_backoff == 8) __builtin_amdgcn_s_sleep(3);
          else __builtin_amdgcn_s_sleep(5);
        }
        while (true) {
          // 16-byte loads = two granules each; asm: the compiler must neither cache the values nor pick the scope.
          // sc1 = agent scope.  (sc0 -- workgroup scope -- polls were tried for clusters that share an XCD: they never
          // observe the other CUs' publishes; kept behind OU_GRU_BACKOFF=3 for the record.)
          if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < NI; i++)
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(hv[i]) : "v"(src), "n"(1024 * i) : "memory");
          } else if (plain && p.poll_backoff == 3) {
#pragma unroll
            for (int i = 0; i < NI; i++) {
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0"
                           : "=v"(hv[2 * i]) : "v"(src), "n"(4 * LPU * i * 8) : "memory");
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0"
                           : "=v"(hv[2 * i + 1]) : "v"(src), "n"(4 * LPU * i * 8 + 16) : "memory");
            }
          } else if (p.poll_backoff == 4) {  // experiment: system-scope polls
#pragma unroll
            for (int i = 0; i < NI; i++) {
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1"
                           : "=v"(hv[2 * i]) : "v"(src), "n"(4 * LPU * i * 8) : "memory");
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc0 sc1"
                           : "=v"(hv[2 * i + 1]) : "v"(src), "n"(4 * LPU * i * 8 + 16) : "memory");
            }
          } else {
            if (p.poll_backoff == 5 && (spins & 15u) == 15u) asm volatile("buffer_inv sc1" ::: "memory");  // experiment
#pragma unroll
            for (int i = 0; i < NI; i++) {
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1"
                           : "=v"(hv[2 * i]) : "v"(src), "n"(4 * LPU * i * 8) : "memory");
              asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1"
                           : "=v"(hv[2 * i + 1]) : "v"(src), "n"(4 * LPU * i * 8 + 16) : "memory");
            }
          }
          gather_wait(hv);
          // all tags arrived <=> the smallest one is the wanted one: stale granules always carry SMALLER tags (tags are
          // monotonic per exchange area; a larger one could only come from a buffer that ou_workspace_init has not prepared,
          // which the C ABI refuses, and an epoch wrap clears the whole area).  An explicit min == max == want test costs a
          // second reduction tree on the critical path of every step: +50 cycles per step, measured.
          unsigned m = hv[0].y < hv[0].w ? hv[0].y : hv[0].w;
#pragma unroll
          for (int k = 1; k < NC / 2; k++) {
            const unsigned a = hv[k].y < hv[k].w ? hv[k].y : hv[k].w;
            m = a < m ? a : m;
          }
          if (__builtin_amdgcn_ballot_w64(m != want) == 0ull) break;  // wave-uniform: the wave needs all H values anyway
          // every wave polls all H granules: 32 line requests per wave and round -- a few per cent of the L2 request
          // rate for the two clusters of a batch-1 call; with dozens of clusters the polling-wave kernel (one poller per
          // workgroup) is ahead again, see the version rule in ou_api.cpp.  Back off if a wait gets long.
          ++spins;
          if ((spins & 63u) == 0u) __builtin_amdgcn_s_sleep(4);
          // Safety net: a plain store carries no visibility deadline.  Under load -- a second process on the device
          // (tests/test_gpu_distributed.py: 1 run in 4 timed out), or dozens of clusters (OR16, B = 16: multi-second stalls)
          // -- a publish was seen to stay invisible to the other CUs for good.  Everybody ends up waiting then, the wave
          // whose store is missing too: after ~0.1 ms of waiting (256 poll rounds; a healthy wait is 2-4) every wave repeats
          // its last publish (tag epoch + step) as a system-scope write-through store.  Never taken in a healthy run.
          if (__builtin_expect((spins & 15u) != 15u, 1)) continue;
          // a wait that long is unusual (a healthy one takes 2-4 rounds): has somebody raised the cluster's mode flag?
          bool flagged = false;
          if (!sysmode && !(p.dbg & 1)) {
            u32x2 mflag;
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=v"(mflag) : "v"(xq + 2 * H + 63) : "memory");
            flagged = __builtin_amdgcn_readfirstlane(mflag.y) == epoch;
          }
          if (((spins & 255u) == 255u || flagged) && !(p.dbg & 1)) {
            if (!flagged && !sysmode)  // first trigger of this wave: leave a record of what the stale granule looks like
              gru_stale_probe(xq + (size_t)(step & 1) * H, WIDE ? 2 * lane : cg * 4, NC, WIDE ? 128 : 4 * LPU, want, p.err,
                              ((unsigned)cluster << 16) | ((unsigned)g << 8) | (unsigned)(tid >> 6), xcc, (unsigned)step,
                              WIDE ? 2 : 4);
            if (fin) {
              const unsigned long long gran = ((unsigned long long)want << 32) | (unsigned)__float_as_int(hprev);
              unsigned long long* dst = xq + (size_t)(step & 1) * H + unit;
              asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(gran) : "memory");
            }
            // ... and for the rest of this launch the wave publishes that way in the first place: when the effect shows
            // up it lasts (one GRU pass out of 574 in a profiled run needed a recovery on almost every step: 46 ms
            // instead of 0.28), whereas a system-scope publish costs about one more hop per step (0.39 ms per pass).
            // The cluster's flag granule (slot 63 of the rendezvous area, = this launch's epoch) makes every other wave
            // -- they are all waiting, and look at the flag every 16 rounds -- switch right away instead of after a
            // 256-round wait of its own (64 waves x 0.1 ms otherwise).  Status word 31 counts the triggers.
            if (!sysmode) {
              sysmode = true;
              fast_pub = false;
              if (lane == 0 && !flagged) {  // tell the rest of the cluster: flag granule = this launch's epoch
                const unsigned long long fl = ((unsigned long long)epoch << 32) | 1u;
                asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(xq + 2 * H + 63), "v"(fl) : "memory");
                atomicAdd(p.err + 31, 1u);
              }
            }
            // status word 20 counts the recoveries (one per wave and event)
            if (lane == 0) atomicAdd(p.err + 20, 1u);
          }
          // a long wait (~2 ms): leave this workgroup's position in its rendezvous slot for whoever reports a time-out
          if (spins == 4096u && tid == 0) gru_note_long_wait(xq + 2 * H + g, epoch, xcc, (unsigned)step);
          if (spins > GRU_SPIN_LIMIT) {
            if (lane == 0)
              gru_timeout_report(p.err, xq + 2 * H, NWG, epoch, ((unsigned)cluster << 16) | ((unsigned)g << 8) | (plain ? 1u : 0u),
                                 ((unsigned)step << 8) | (xcc & 0xFFu), m, 0u, want);
            // (which granule is stale: the first-event record of gru_stale_probe, written at the first recovery)
            step = T;
            break;
          }
        }
        if (step >= T) break;
        if (ts_on) c_rounds += spins + 1u;
      }
      if (ts_on) q1 = __builtin_readcyclecounter();
      // ---- matvec: (r, z) as packed row pairs against the granule's value half, n as scalar FMAs
      float hs[3];
      if constexpr (WIDE) {
        // 3 WU partial sums (unit u of the wave x gate) over this lane's HB columns
        f32x2 arz[WU];
        float an[WU];
#pragma unroll
        for (int u = 0; u < WU; u++) { arz[u] = f32x2{0.f, 0.f}; an[u] = 0.f; }
#pragma unroll
        for (int k = 0; k < NC / 2; k++) {
          const float h0 = __uint_as_float(hv[k].x), h1 = __uint_as_float(hv[k].z);
#pragma unroll
          for (int u = 0; u < WU; u++) {
            arz[u] = __builtin_elementwise_fma(wrz[u * HB + 2 * k], f32x2{h0, h0}, arz[u]);
            an[u] = fmaf(wn[u * HB + 2 * k], h0, an[u]);
            arz[u] = __builtin_elementwise_fma(wrz[u * HB + 2 * k + 1], f32x2{h1, h1}, arz[u]);
            an[u] = fmaf(wn[u * HB + 2 * k + 1], h1, an[u]);
          }
        }
        // fold 64 -> 32 lanes: units u and u + WU / 2 trade halves; lanes < 32 keep the lower units, lanes >= 32 the upper ones
        constexpr int HU = WU / 2;
        f32x2 rz01[HU];
        float n01[HU];
#pragma unroll
        for (int u = 0; u < HU; u++) {
          const auto sr = __builtin_amdgcn_permlane32_swap(__float_as_uint(arz[u].x), __float_as_uint(arz[u + HU].x), false, false);
          const auto sz = __builtin_amdgcn_permlane32_swap(__float_as_uint(arz[u].y), __float_as_uint(arz[u + HU].y), false, false);
          const auto sn = __builtin_amdgcn_permlane32_swap(__float_as_uint(an[u]), __float_as_uint(an[u + HU]), false, false);
          rz01[u] = f32x2{__uint_as_float(sr[0]), __uint_as_float(sz[0])} + f32x2{__uint_as_float(sr[1]), __uint_as_float(sz[1])};
          n01[u] = __uint_as_float(sn[0]) + __uint_as_float(sn[1]);
        }
        if constexpr (WU == 4) {
          // fold 32 -> 16 lanes: the two remaining units trade rows; row r of the wave ends up with unit r
          const auto sr = __builtin_amdgcn_permlane16_swap(__float_as_uint(rz01[0].x), __float_as_uint(rz01[1].x), false, false);
          const auto sz = __builtin_amdgcn_permlane16_swap(__float_as_uint(rz01[0].y), __float_as_uint(rz01[1].y), false, false);
          const auto sn = __builtin_amdgcn_permlane16_swap(__float_as_uint(n01[0]), __float_as_uint(n01[1]), false, false);
          const f32x2 rz = f32x2{__uint_as_float(sr[0]), __uint_as_float(sz[0])} + f32x2{__uint_as_float(sr[1]), __uint_as_float(sz[1])};
          hs[0] = rz.x; hs[1] = rz.y;
          hs[2] = __uint_as_float(sn[0]) + __uint_as_float(sn[1]);
        } else {  // two units per wave: each half of the wave goes on as one 32-lane group (row sums + row_bcast:15 below)
          hs[0] = rz01[0].x; hs[1] = rz01[0].y; hs[2] = n01[0];
        }
      } else {
        f32x2 arz0 = {0.f, 0.f}, arz1 = {0.f, 0.f};
        float an0 = 0.f, an1 = 0.f;
#pragma unroll
        for (int k = 0; k < NC / 2; k++) {
          const float h0 = __uint_as_float(hv[k].x), h1 = __uint_as_float(hv[k].z);
          arz0 = __builtin_elementwise_fma(wrz[2 * k], f32x2{h0, h0}, arz0);
          arz1 = __builtin_elementwise_fma(wrz[2 * k + 1], f32x2{h1, h1}, arz1);
          an0 = fmaf(wn[2 * k], h0, an0);
          an1 = fmaf(wn[2 * k + 1], h1, an1);
        }
        hs[0] = arz0.x + arz1.x; hs[1] = arz0.y + arz1.y; hs[2] = an0 + an1;
      }
#pragma unroll
      for (int gt = 0; gt < 3; gt++) {
        hs[gt] = LPU == 8 ? row8_sum(hs[gt]) : row16_sum(hs[gt]);
        if (LPU == 32) {  // rows 1 / 3 += total of rows 0 / 2
          const int lo = __builtin_amdgcn_update_dpp(0, __float_as_int(hs[gt]), 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
          hs[gt] += __int_as_float(lo);
        }
      }

      if (fin) {
        const float r = sigmoidf_(xr + hs[0]);
        const float z = sigmoidf_(xz + hs[1]);
        const float n = tanhf_(xn + r * (hs[2] + bhn));
        const float hnew = (hprev - n) * z + n;
        hprev = hnew;
        {  // publish first: everybody is waiting on this
          const unsigned long long gran =
              ((unsigned long long)(epoch + (unsigned)step + 1u) << 32) | (unsigned)__float_as_int(hnew);
          unsigned long long* dst = xq + (size_t)((step + 1) & 1) * H + unit;
          // (one branch on the fast path: this store is on the critical path of every step.)  Plain store = the cluster
          // shares one XCD (proved by the rendezvous): the line stays in the L2 that every poller's sc1 load is served from.
          // Otherwise ONE agent-scope (sc1, write-through) 8-byte store per granule, the documented form of a data-tagged
          // hand-off on gfx950 -- it drops the line from the XCD's L2 (+0.11 ms per pass), see DESIGN.md 4.4.
          if (__builtin_expect(fast_pub && step != inject_step, 1)) {
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(gran) : "memory");
          } else if (step == inject_step) {
            // fault injection (tests): workgroup 1 "loses" its publishes of step 50 -- the safety net has to bring them back
          } else if (sysmode) {
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(dst), "v"(gran) : "memory");
          } else {
            asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(gran) : "memory");
          }
        }
        p.out[orow + t] = has_res ? (hnew + rs) * p.res_scale : hnew;
      }
      if (ts_on) {
        const long long q2 = __builtin_readcyclecounter();
        c_poll += q1 - q0; c_comp += q2 - q1;
        if (p.poll_backoff == 10) {  // experiment: how long until this wave's two stores (publish, output) are acknowledged?
          __builtin_amdgcn_s_waitcnt(0x0F70);
          c_ack += __builtin_readcyclecounter() - q2;
        } else if (p.poll_backoff >= 11 && p.poll_backoff <= 14) {
          // experiment: latency of ONE 8-byte load per lane once this wave's stores are acknowledged --
          // 11: sc1, quiet lines (the rendezvous slots), 12: sc1, the buffer that was gathered in this step (nobody writes
          // it now), 13: the same without sc1, 14: sc1, the buffer everybody is publishing into right now
          __builtin_amdgcn_s_waitcnt(0x0F70);
          const unsigned long long* a = p.poll_backoff == 11 ? xq + 2 * H + (lane & 31)
                                        : xq + (size_t)((step + (p.poll_backoff == 14 ? 1 : 0)) & 1) * H + (tid >> 6) * (H / 4) + lane % (H / 4);
          u32x2 d;
          const long long q3 = __builtin_readcyclecounter();
          if (p.poll_backoff == 13) asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(a) : "memory");
          else asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(d) : "v"(a) : "memory");
          c_ack += __builtin_readcyclecounter() - q3;
        }
      }
    }
    if (ts_on && lane == 0) {
      long long* o = p.tstamps + ((size_t)blockIdx.x * 8 + (tid >> 6)) * 8;
      // 10 ns ticks: kernel entry -> first step (weights, rendezvous, first chunks), the T steps
      o[0] = c_comp; o[1] = c_poll; o[2] = c_rounds; o[3] = T; o[4] = r_loop - r_start;
      o[5] = (long long)__builtin_amdgcn_s_memrealtime() - r_loop; o[6] = r_start; o[7] = c_ack;
    }
  }
  // ---- epoch hand-over: the last block to finish advances the epoch for the next launch on this exchange area
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    const unsigned done = atomicAdd(p.epoch + 1, 1u);
    if (done == gridDim.x - 1) {
      p.epoch[1] = 0u;
      unsigned next = epoch + (unsigned)T;  // stored counter = last tag used
      if (next >= GRU_EPOCH_WRAP) {  // tags must stay monotonic: clear the area and restart (every block is done)
        // the whole area of this GRU layer, not just this launch's clusters: the sub-launches of a chunked batch share
        // it, and a stale high tag left behind a smaller remainder launch would outlive the restart
        const size_t n = p.xchg_granules ? p.xchg_granules : (size_t)nclusters * CSTRIDE;
        for (size_t i = 0; i < n; i++) p.xchg_base[i] = 0ull;
        next = 0u;
      }
      __threadfence();
      __hip_atomic_store(p.epoch, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int HB>
static void (*gru_ring_entry(int upw, int wide))(GruArgs, int) {
  if constexpr (HB <= 4) {
    if (upw == 32) return gru_ring_kernel<HB, 32>;
  }
  if constexpr (HB % 2 == 0) {  // 8 units per workgroup: half the matvec per wave, twice the workgroups
    if (upw == 8 && wide) return gru_ring_kernel<HB, 8, true>;
  }
  if constexpr (HB >= 2 && HB <= 4) {
    if (upw == 8) return gru_ring_kernel<HB, 8>;
  }
  if constexpr (HB % 2 == 0) {
    if (wide) return gru_ring_kernel<HB, 16, true>;
  }
  return gru_ring_kernel<HB, 16>;
}
// Workgroups of the ring kernel that can be resident per CU (every member of a cluster spins on the others: the whole
// grid has to be on the machine at once).  The occupancy query can be one block high where SGPRs are the limit (guide:
// admitted = min(API, 8, 800 / (ceil(sgpr / 16) * 16 + 16)): API 8 -> 7 at 81-96 SGPRs, 7 -> 6 at 97-112), which only
// concerns answers >= 7: one block of margin is taken off those.  At most TWO per CU are relied upon (two workgroups = two
// waves per SIMD, each waiting on its gather most of the time).
template <int HB>
static int gru_ring_resident_per_cu(int upw, int wide) {
  static int cache[5] = {0, 0, 0, 0, 0};
  const int slot = upw == 32 ? 2 : (upw == 8 ? (wide ? 4 : 0) : (wide ? 3 : 1));
  if (cache[slot] == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(gru_ring_entry<HB>(upw, wide)), 256,
                                                     0) != hipSuccess)
      nb = 1;
    if (nb >= 7) nb -= 1;
    cache[slot] = nb >= 2 ? 2 : 1;
  }
  return cache[slot];
}
// OU_GRU_UPW (force_upw): 0 auto | 16 / 8 units per workgroup, wide layout | 17 / 9 the same in the default (round-2)
// layout | 32 units, default layout
static int gru_ring_wide(int H, int force_upw) {
  return ((H / 64) % 2 == 0 && force_upw != 17 && force_upw != 32 && force_upw != 9) ? 1 : 0;
}
// hidden units per workgroup.  16 by default; 8 (wide layout: two units per wave -- half the matvec and one fold level less
// on the critical path of every step, 252 -> 240 us per 401-frame pass) while every workgroup of the launch still gets a CU
// of its own: measured 7.59 -> 7.43 ms per enhance at B = 1, 16.80 -> 16.71 at B = 4, but 29.0 -> 29.5 at B = 8 (two
// workgroups per CU).  B = 0: the choice for the smallest batch.
static int gru_ring_upw(int H, int force_upw, int B, int num_cu) {
  if (force_upw == 32 && H <= 256) return 32;
  if (force_upw == 9 && H >= 128 && H <= 256) return 8;
  if (force_upw == 8 && (H / 64) % 2 == 0 && H <= 384) return 8;
  if (force_upw == 0 && (H / 64) % 2 == 0 && H <= 256 && 2 * (B > 0 ? B : 1) * (H / 8) <= num_cu) return 8;
  return 16;
}
// utterances one ring-kernel launch may carry: whole groups of 8 clusters (one per XCD), two clusters per utterance,
// the whole grid resident -- on HALF the machine when another GRU layer may run beside it (`shared`)
int gru_ring_batch_cap(int H, int num_cu, int shared, int force_upw, int B) {
  if (H % 64) return 0;
  const int upw = gru_ring_upw(H, force_upw, B, num_cu), nwg = H / upw;
  const int wide = gru_ring_wide(H, force_upw);
  int per_cu = 1;
  switch (H / 64) {
    case 1: per_cu = gru_ring_resident_per_cu<1>(upw, wide); break;
    case 2: per_cu = gru_ring_resident_per_cu<2>(upw, wide); break;
    case 4: per_cu = gru_ring_resident_per_cu<4>(upw, wide); break;
    case 6: per_cu = gru_ring_resident_per_cu<6>(upw, wide); break;
    default: return 0;
  }
  const int wg_cap = num_cu * per_cu / (shared ? 2 : 1);
  return (wg_cap / (8 * nwg)) * 8 / 2;
}
template <int HB>
static hipError_t launch_gru_ring(const GruArgs& c, int upw, int nclusters, hipStream_t st) {
  constexpr int H = 64 * HB;
  const int nwg = H / upw;
  dim3 grid(8 * nwg * ((nclusters + 7) / 8));
  hipLaunchKernelGGL(gru_ring_entry<HB>(upw, gru_ring_wide(H, c.force_upw)), grid, dim3(256), 0, st, c, nclusters);
  return hipGetLastError();
}

template <int HB>
static hipError_t launch_gru_variant(const GruArgs& c, int upw, int nclusters, hipStream_t st) {
  constexpr int H = 64 * HB;
  const int nwg = H / upw;
  dim3 grid(8 * nwg * ((nclusters + 7) / 8));
  if (upw == 64) hipLaunchKernelGGL((gru_cluster_kernel<HB, 64, 512>), grid, dim3(512), 0, st, c, nclusters);
  else if (upw == 32) hipLaunchKernelGGL((gru_cluster_kernel<HB, 32, 512>), grid, dim3(512), 0, st, c, nclusters);
  else hipLaunchKernelGGL((gru_cluster_kernel<HB, 16, 256>), grid, dim3(256), 0, st, c, nclusters);
  return hipGetLastError();
}

hipError_t launch_gru(const GruArgs& a, int num_cu, hipStream_t st) {
  if (a.H % 64) return hipErrorInvalidValue;
  const int HB = a.H / 64;
  // Every workgroup of a cluster has to be resident at once; clusters are dealt to XCDs in groups of 8 and a launch
  // is sized to at most half the CUs (the conditioner's and the score net's GRUs may overlap).  Take the finest split
  // -- least work per time step; measured 12.1 / 12.6 / 13.1 ms per PP16 enhance for 16 / 32 / 64 units per
  // workgroup, the same order at B = 2, 4, 8 -- that still runs the whole batch in ONE launch.
  auto batch_cap = [&](int u) { return (num_cu / (8 * (a.H / u))) * 8 / 2; };
  int upw = 64;
  if (a.force_upw) upw = a.force_upw;
  else if (a.H % 16 == 0 && batch_cap(16) >= a.B) upw = 16;
  else if (a.H % 32 == 0 && batch_cap(32) >= a.B) upw = 32;
  // the ring kernel runs 256-thread workgroups of 16 units (32 on request, H <= 256)
  if (a.version == 2) upw = gru_ring_upw(a.H, a.force_upw, a.B, num_cu);
  const int nwg = a.H / upw;
  int bmax = batch_cap(upw);
  // residency of the ring kernel: every member of a cluster spins on the others, so a launch is sized to what can be on
  // the machine at once (half of it when a second GRU layer may run beside this one: conditioner / first score pass);
  // a batch that does not fit is split into sub-launches, never enqueued oversized
  if (a.version == 2) bmax = gru_ring_batch_cap(a.H, num_cu, a.shared, a.force_upw, a.B);
  if (a.force_bmax > 0 && a.force_bmax < bmax) bmax = a.force_bmax;
  if (bmax < 1) return hipErrorInvalidConfiguration;
  for (int b0 = 0; b0 < a.B; b0 += bmax) {
    GruArgs c = a;
    c.B = (a.B - b0 < bmax) ? a.B - b0 : bmax;
    c.gx = a.gx + (size_t)b0 * 6 * a.H * a.T;
    c.out = a.out + (size_t)b0 * 2 * a.H * a.T;
    if (a.res) c.res = a.res + (size_t)b0 * 2 * a.H * a.T;
    if (a.version == 2) {
      if (!a.epoch) return hipErrorInvalidValue;
      c.xchg_base = a.xchg;
      c.xchg_granules = gru_granules(a.B, a.H);
      hipError_t e;
      switch (HB) {
        case 1: e = launch_gru_ring<1>(c, upw, 2 * c.B, st); break;
        case 2: e = launch_gru_ring<2>(c, upw, 2 * c.B, st); break;
        case 4: e = launch_gru_ring<4>(c, upw, 2 * c.B, st); break;
        case 6: e = launch_gru_ring<6>(c, upw, 2 * c.B, st); break;
        default: return hipErrorInvalidConfiguration;
      }
      if (e != hipSuccess) return e;
      continue;
    }
    if (nwg > 1) {
      hipError_t e = hipMemsetAsync(c.xchg, 0, (size_t)c.B * 4 * a.H * sizeof(unsigned long long), st);
      if (e != hipSuccess) return e;
    }
    hipError_t e;
    switch (HB) {
      case 1: e = launch_gru_variant<1>(c, upw, 2 * c.B, st); break;
      case 2: e = launch_gru_variant<2>(c, upw, 2 * c.B, st); break;
      case 4: e = launch_gru_variant<4>(c, upw, 2 * c.B, st); break;
      case 6: e = launch_gru_variant<6>(c, upw, 2 * c.B, st); break;
      default: return hipErrorInvalidConfiguration;
    }
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// =========================================================================================================
// signal decoupling layer (cold branch): 2x sinc upsample -> Snake -> 2x downsample -> Conv1d(C -> 1, k3)
//   torchaudio Resample(1->2): pad (7, 8), conv with the 2 phase kernels (15 taps), interleave, crop to 2T.
//   Resample(2->1): pad (13, 15), 28-tap kernel, stride 2.
// =========================================================================================================
__global__ __launch_bounds__(256) void snake_up_kernel(const float* __restrict__ aux, const float* __restrict__ alpha_exp,
                                                       const float* __restrict__ up_k, float* __restrict__ u, int C,
                                                       int T) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;  // index in the 2T up-sampled signal
  if (i >= 2 * T) return;
  const float* xr = aux + ((size_t)b * C + c) * T;
  const int q = i >> 1, ph = i & 1;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 15; k++) {
    int t = q + k - 7;
    float v = (t >= 0 && t < T) ? xr[t] : 0.f;
    acc = fmaf(up_k[ph * 15 + k], v, acc);
  }
  const float a = alpha_exp[c];
  const float sn = sinf(acc * a);
  u[((size_t)b * C + c) * 2 * T + i] = acc + (1.0f / (a + 1e-9f)) * (sn * sn);  // snake.py:59-62
}
__global__ __launch_bounds__(256) void snake_down_conv_kernel(const float* __restrict__ u, const float* __restrict__ down_k,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ out, int C, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float acc = 0.f;
  for (int c = 0; c < C; c++) {
    const float* ur = u + ((size_t)b * C + c) * 2 * T;
#pragma unroll
    for (int k3 = 0; k3 < 3; k3++) {
      int tt = t + k3 - 1;
      if (tt < 0 || tt >= T) continue;
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < 28; k++) {
        int j = 2 * tt + k - 13;
        float v = (j >= 0 && j < 2 * T) ? ur[j] : 0.f;
        d = fmaf(down_k[k], v, d);
      }
      acc = fmaf(w[c * 3 + k3], d, acc);
    }
  }
  out[(size_t)b * T + t] = acc + bias[0];
}
hipError_t launch_decoupling(const float* aux, const float* alpha_exp, const float* up_k, const float* down_k,
                             const float* w, const float* bias, float* tmp_up, float* out, int B, int C, int T,
                             hipStream_t st) {
  hipLaunchKernelGGL(snake_up_kernel, dim3((2 * T + 255) / 256, C, B), dim3(256), 0, st, aux, alpha_exp, up_k, tmp_up, C, T);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(snake_down_conv_kernel, dim3((T + 255) / 256, B), dim3(256), 0, st, tmp_up, down_k, w, bias, out, C, T);
  return hipGetLastError();
}

}  // namespace ou
