// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: fused ConvBlock body of the wide levels (conv_chain_kernel) and the small-K rate-change kernels
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ou {

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
#ifdef OU_EXPERIMENTS  // round 3's fused body (37-40 spilled SGPRs, no default rule selects it since conv_chainw_kernel: round 6 retires it to `make EXPERIMENTS=1`)
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

#endif  // OU_EXPERIMENTS

// =========================================================================================================
// conv_chainw_kernel: the fused ConvBlock body of the 32-channel level in MINIMAL-FILTERING form (round 5)
//   conv1 (k5, F(2, 5)) -> (+cond)/sqrt2 -> FiLM -> conv2 (k3, F(2, 3)) -> conv3 (k3) -> (+h)/sqrt2      (blocks.py:377-399)
// conv_chain_kernel spends 55 % of a block's life in its MFMA phases (two waves per SIMD, 84 % of the pipe there) and the rest
// staging, in epilogues and at one barrier per weight slot (tools/chain_ts.py: 27 k of 48.5 k cycles).  Here:
//   * F(2, KW) (see conv_direct2w_kernel): a wave owns 32 rows x 32 tile positions = 64 columns of every stage; per channel
//     pair KW + 1 MFMAs on independent accumulators instead of 2 KW -- 224 MFMAs per wave and block instead of 352;
//   * ALL weights of the three convs sit in LDS in the Winograd domain for the whole block (C = 32: 24 + 16 + 16 KB, loaded
//     once beside the input tile): no slot ring, no per-slot barrier -- three barriers per block in all;
//   * four waves, one per SIMD, 256 columns per block (252 finished: the halo of 2 + 1 + 1 is recomputed, as before);
//     operands come from LDS with 16- / 8-byte reads (A: the KW + 1 values U_x of (channel, row) are adjacent; B: the window of
//     a tile position starts at an even column of the activation tile, whatever the stage -- stage s writes its output u
//     where stage s + 1 reads input u), ~3-5 LDS instructions per 4-6 MFMAs;
//   * the global operands of the epilogues (cond add, residual) are requested before the channel loop of their stage.
// Same masking of columns outside the signal as conv_chain_kernel (the next conv must see zero padding there).  Results
// differ from the plain fused kernel by the rounding of the transforms (tests: > 100 dB against it and the oracle).
// =========================================================================================================
// MT: 32-row tiles (C = 32 MT).  D3: depth 3 = conv1 (k5), conv2, conv3 (k3); else depth 2 = conv2, conv3 (conv1 was a launch of
// its own -- the 64-channel level: a depth-3 tile of 128 columns finishes 124 of them, 259 tiles for T = 32 080 on 256 CUs).
//   C = 32, depth 3: four waves = four column groups of 64; all weights resident (U1a | U1b | U2 | U3), two activation tiles.
//   C = 64, depth 2: four waves = two row tiles x two column groups, 128 columns per block (126 finished); ONE weight region
//     (64 KB) that holds U2 during stage 0 and U3 -- parked in registers since kernel entry -- from the barrier behind it; one
//     activation tile.
template <int MT, bool D3>
__global__ __launch_bounds__(256) void conv_chainw_kernel(ChainArgs p, int TN, int ntiles) {
  constexpr int C = 32 * MT, NW = 4, NWN = NW / MT, NC = 64 * NWN, XS = NC + 8, NTH = 64 * NW, NS = D3 ? 3 : 2;
  constexpr int WI = C * C / NTH;  // (ci, m) items per thread and conv
  static_assert(!D3 || MT == 1, "depth 3: the 32-channel level");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bufB = smem;                          // [C][XS]  output of stage 0
  float* bufA = bufB + C * XS;                 // [C][XS]  output of stage 1 (depth 3 only)
  float* Ua0 = bufA + (D3 ? C * XS : 0);       // [C ci][C m][4]  stage 0: U_0 .. U_3
  float* Ub0 = Ua0 + C * C * 4;                // [C ci][C m][2]  stage 0 of depth 3: U_4, U_5
  float* U2 = Ub0 + (D3 ? C * C * 2 : 0);      // depth 3: conv2, conv3
  float* U3 = U2 + (D3 ? C * C * 4 : 0);
  float* prm = D3 ? U3 + C * C * 4 : Ub0;      // bias[3][C], gamma[C], beta[C]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % MT, wn = wave / MT;
  const int lhalf = lane >> 5, l31 = lane & 31;
  const int tile = blockIdx.x % ntiles, b = blockIdx.x / ntiles;
  const int T = p.T, Mp = p.Mp;
  // ragged batch: every stage's tile is zero from the row's own end on (what 'same' padding is for a row alone); x, add and res
  // are zero there already
  const int rlen = ragged_len(p.lens, b), Tl = rlen < T ? rlen : T;
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int t0 = tile * TN;
  constexpr int KW0 = D3 ? 5 : 3, PAD0 = (KW0 - 1) / 2, R0 = D3 ? 2 : 1;  // halo still needed downstream of stage 0
  const bool ts_on = p.tstamps != nullptr;  // tuning (OU_CHAIN_TS): cycles {loads issued, first barrier, NS x (channel loop, epilogue + barrier)}
  long long tsv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (ts_on) tsv[0] = __builtin_readcyclecounter();
  const size_t rowbase = (size_t)b * C * T;
  const unsigned plane = (unsigned)C * (unsigned)T * 4u;
  const int Tb = T * 4;

  // ---- weights -> registers (item = (ci, m)); the first stage's are staged to LDS below, the others parked
  f32x4 wa[WI], wb[WI], wc[D3 ? WI : 1];
  f32x2 wa2[D3 ? WI : 1];
#pragma unroll
  for (int k = 0; k < WI; k++) {
    const int it = tid + k * NTH, ci = it / C, m = it % C;
    if constexpr (D3) {
      const float* s1 = p.cv[0].wu + ((size_t)ci * Mp + m) * 8;
      wa[k] = *reinterpret_cast<const f32x4*>(s1);
      wa2[k] = *reinterpret_cast<const f32x2*>(s1 + 4);
      wb[k] = *reinterpret_cast<const f32x4*>(p.cv[1].wu + ((size_t)ci * Mp + m) * 4);
      wc[k] = *reinterpret_cast<const f32x4*>(p.cv[2].wu + ((size_t)ci * Mp + m) * 4);
    } else {
      wa[k] = *reinterpret_cast<const f32x4*>(p.cv[0].wu + ((size_t)ci * Mp + m) * 4);
    }
  }
  // ---- stage 0's B operands come STRAIGHT FROM GLOBAL MEMORY into registers (as in conv_direct2w_kernel): lane (position, half)
  // loads the KW + 1 samples of its window of channel 2 I + half for all C / 2 channel pairs, issued here, consumed in order by
  // the stage-0 loop -- the first conv starts as soon as the first rows and its weights are there instead of after the whole
  // input tile has been staged through LDS and a barrier (7 k + 2 k of 42 k cycles per block in the first version).
  const int pcol = 64 * wn + 2 * l31;     // this lane's tile position: outputs u = pcol, pcol + 1 of every stage
  const int tw = t0 - R0 - PAD0 + pcol;   // time of window element 0 (even)
  const int sh = tw < 0 ? -tw : 0;        // samples cut off in front of the row (first tile: 4 / 2)
  unsigned wmask = 0;                     // bit i: window element i is inside the signal
#pragma unroll
  for (int i = 0; i <= KW0; i++) wmask |= (tw + i >= 0 && tw + i < T) ? (1u << i) : 0u;
  const bool edge = __builtin_amdgcn_readfirstlane((t0 - R0 - PAD0 < 0 || t0 - R0 - PAD0 + NC + KW0 - 1 > T) ? 1 : 0) != 0;
  f32x4 gw4[C / 2];
  f32x2 gw2[D3 ? C / 2 : 1];
  {
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + rowbase, plane);
    const int wvo = wmask ? (lhalf * T + tw + sh) * 4 : (int)0x80000000;  // (windows wholly outside: out of range, reads 0)
#pragma unroll
    for (int I = 0; I < C / 2; I++) {
      gw4[I] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, wvo, 2 * I * Tb, 0));
      if constexpr (D3) gw2[I] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, wvo + 16, 2 * I * Tb, 0));
    }
  }
  if constexpr (!D3) {  // the second conv's weights: parked in registers until the weight region is free (behind stage 0)
#pragma unroll
    for (int k = 0; k < WI; k++) {
      const int it = tid + k * NTH, ci = it / C, m = it % C;
      wb[k] = *reinterpret_cast<const f32x4*>(p.cv[1].wu + ((size_t)ci * Mp + m) * 4);
    }
  }
  if (ts_on) tsv[1] = __builtin_readcyclecounter();
  for (int i = tid; i < C; i += NTH) {
    prm[i] = p.cv[0].bias[i];
    prm[C + i] = p.cv[1].bias[i];
    prm[2 * C + i] = D3 ? p.cv[2].bias[i] : 0.f;
    prm[3 * C + i] = (D3 && p.film) ? p.film[(size_t)b * p.film_bstride + i] : 1.f;
    prm[4 * C + i] = (D3 && p.film) ? p.film[(size_t)b * p.film_bstride + C + i] : 0.f;
  }
  for (int i = tid; i < C * 8; i += NTH) {  // columns NC .. NC + 7 of the tiles: read by the last windows, never written
    bufB[(i >> 3) * XS + NC + (i & 7)] = 0.f;
    if constexpr (D3) bufA[(i >> 3) * XS + NC + (i & 7)] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < WI; k++) {
    const int it = tid + k * NTH;
    *reinterpret_cast<f32x4*>(&Ua0[it * 4]) = wa[k];
    if constexpr (D3) *reinterpret_cast<f32x2*>(&Ub0[it * 2]) = wa2[k];
  }
  __syncthreads();
  if (ts_on) tsv[2] = __builtin_readcyclecounter();

  const int lrow = 32 * wm + 4 * lhalf;   // lane part of the accumulator row: row(r) = lrow + (r & 3) + 8 (r >> 2)

  auto run_stage = [&](auto SC) {
    constexpr int s = decltype(SC)::value;
    constexpr int KW = (D3 && s == 0) ? 5 : 3, NX = KW + 1;
    constexpr bool last = s == NS - 1;
    constexpr bool first3 = D3 && s == 0;          // conv1: cond add + FiLM + c1_out
    const float* inb = (s == 1) ? bufB : bufA;     // (stage 0 reads registers)
    float* outb = (s == 0) ? bufB : bufA;
    const float* Ua = D3 ? (s == 0 ? Ua0 : (s == 1 ? U2 : U3)) : Ua0;
    constexpr int Rs = NS - 1 - s;                  // halo still needed downstream of this stage (k3 convs follow)
    const int t = t0 - Rs + pcol;                   // time of output u = pcol (even)
    const bool in0 = t >= 0 && t < Tl, in1 = t + 1 >= 0 && t + 1 < Tl;   // live samples of this row
    const bool on0 = t >= 0 && t < T, on1 = t + 1 >= 0 && t + 1 < T;     // samples of the tensor
    // epilogue operand (cond add of conv1 / residual of the last conv), requested before the channel loop
    f32x2 ev[16];
    const float* src = last ? p.res : (first3 ? p.add : nullptr);
    if (src) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + rowbase, plane);
      const int vo = (on0 || on1) ? (lrow * T + t) * 4 : (int)0x80000000;  // (t even, T even: whole pairs inside or outside)
#pragma unroll
      for (int r = 0; r < 16; r++)
        ev[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, ((r & 3) + 8 * (r >> 2)) * Tb, 0));
    }
    floatx16 acc[NX];
#pragma unroll
    for (int x = 0; x < NX; x++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[x][r] = 0.f;
    const float* ap = Ua + (lhalf * C + 32 * wm + l31) * 4;  // U_0..3 of (ci = half, m); + 2 I C 4 per channel pair
    const float* ap2 = Ub0 + (lhalf * C + 32 * wm + l31) * 2;
    const float* bp = inb + lhalf * XS + pcol;               // window of (ci = half, position); + 2 I XS per channel pair
    // ONE wave per SIMD: nothing but this wave's own instruction stream can fill the matrix pipe's 64 cycles per MFMA.  The
    // loop is software-pipelined by hand -- the MFMAs of channel pair I are interleaved with B^T of pair I + 1 and with the LDS
    // reads of pair I + 4 -- and the interleaving is pinned with sched_group_barrier (one MFMA, then a few VALU / one LDS
    // read), or the scheduler clusters the MFMAs and the wave sits in their issue queue with the transform still to do.
    struct Ops { f32x4 a4; f32x2 a2, d01, d23; };
    auto fetch = [&](int I) {
      Ops o;
      o.a4 = *reinterpret_cast<const f32x4*>(ap + I * (2 * C * 4));
      o.a2 = f32x2{0.f, 0.f}; o.d01 = f32x2{0.f, 0.f}; o.d23 = f32x2{0.f, 0.f};
      if constexpr (KW == 5) o.a2 = *reinterpret_cast<const f32x2*>(ap2 + I * (2 * C * 2));
      if constexpr (s > 0) {
        const float* bq = bp + I * (2 * XS);
        o.d01 = *reinterpret_cast<const f32x2*>(bq);
        o.d23 = *reinterpret_cast<const f32x2*>(bq + 2);
      }
      return o;
    };
    // V = B^T d of channel pair I: later stages from the LDS tile (PReLU applied by the producer), stage 0 from the window
    // registers -- edge fix-up (first / last tile only: shift what was loaded from the row start, zero what is outside the
    // signal), then the first conv's PReLU
    auto transform = [&](const Ops& o, int I, float (&V)[NX]) {
      float X[6] = {o.d01.x, o.d01.y, o.d23.x, o.d23.y, 0.f, 0.f};
      if constexpr (s == 0) {
        const float L[6] = {gw4[I].x, gw4[I].y, gw4[I].z, gw4[I].w, D3 ? gw2[D3 ? I : 0].x : 0.f, D3 ? gw2[D3 ? I : 0].y : 0.f};
        const float a0 = p.cv[0].alpha;
        if (edge) {
#pragma unroll
          for (int i = 0; i < NX; i++) {
            float v = L[i];
            v = sh == 1 ? (i >= 1 ? L[i >= 1 ? i - 1 : 0] : 0.f) : v;
            v = sh == 2 ? (i >= 2 ? L[i >= 2 ? i - 2 : 0] : 0.f) : v;
            v = sh == 4 ? (i >= 4 ? L[i >= 4 ? i - 4 : 0] : 0.f) : v;
            X[i] = ((wmask >> i) & 1u) ? v : 0.f;
          }
        } else {
#pragma unroll
          for (int i = 0; i < NX; i++) X[i] = L[i];
        }
#pragma unroll
        for (int i = 0; i < NX; i++) X[i] = X[i] >= 0.f ? X[i] : a0 * X[i];
      }
      wino_bt<KW>(X, V);
    };
    // (LDS read latency under four waves' traffic is ~300 cycles -- more than the 256 MFMA cycles of a k3 channel pair: with
    // the reads one pair ahead the loop ran at 55 % of the pipe.  Ring of four pairs: reads three pairs ahead.)
    constexpr int NP = C / 2, RD = 4;
    static_assert(NP % RD == 0, "ring");
    Ops ring[RD];
#pragma unroll
    for (int k = 0; k < RD; k++) ring[k] = fetch(k);
    float Vc[NX];
    transform(ring[0], 0, Vc);
#pragma unroll
    for (int I0 = 0; I0 < NP; I0 += RD) {   // (fully unrolled: the window registers of stage 0 are indexed by the pair)
#pragma unroll
      for (int k = 0; k < RD; k++) {
        const int I = I0 + k;
        float Vn[NX];
        transform(ring[(k + 1) % RD], I + 1 < NP ? I + 1 : NP - 1, Vn);   // pair I + 1
        const Ops cur = ring[k];
        const float A[6] = {cur.a4.x, cur.a4.y, cur.a4.z, cur.a4.w, cur.a2.x, cur.a2.y};
#pragma unroll
        for (int x = 0; x < NX; x++) acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[x], Vc[x], acc[x], 0, 0, 0);
        ring[k] = fetch(I + RD < NP ? I + RD : NP - 1);     // pair I + 4 into the slot that has just been consumed
#pragma unroll
        for (int x = 0; x < NX; x++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, KW == 5 ? 6 : 2, 0);  // a slice of the next pair's transform
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // one LDS read
        }
#pragma unroll
        for (int x = 0; x < NX; x++) Vc[x] = Vn[x];
      }
    }
    if (ts_on) tsv[3 + 2 * s] = __builtin_readcyclecounter();
    // A^T -> outputs u = pcol (o0), pcol + 1 (o1)
    floatx16 o0, o1;
    if constexpr (KW == 3) {
      o0 = acc[0] + acc[1] + acc[2];
      o1 = acc[1] - acc[2] - acc[3];
    } else {
      o0 = acc[0] + acc[1] + acc[2] + acc[3] + acc[4];
      o1 = acc[1] - acc[2] + 0.5f * acc[3] - 2.0f * acc[4] + acc[5];
    }
    const float* bias_l = prm + s * C + lrow;
    if constexpr (!last) {
      const float an = p.cv[s + 1].alpha;
      float* out_l = outb + lrow * XS + pcol;
      const __amdgpu_buffer_rsrc_t rc = make_rsrc((p.c1_out ? p.c1_out : p.y) + rowbase, plane);
      const bool own0 = on0 && t >= t0 && t < t0 + TN, own1 = on1 && t + 1 >= t0 && t + 1 < t0 + TN;
      float v0[16], v1[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kr = (r & 3) + 8 * (r >> 2);
        v0[r] = o0[r] + bias_l[kr]; v1[r] = o1[r] + bias_l[kr];
      }
      if constexpr (first3) {
        if (p.add) {
#pragma unroll
          for (int r = 0; r < 16; r++) { v0[r] = (v0[r] + ev[r].x) * p.add_scale; v1[r] = (v1[r] + ev[r].y) * p.add_scale; }
        }
        if (p.film) {
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int kr = (r & 3) + 8 * (r >> 2);
            const float ga = prm[3 * C + lrow + kr], be = prm[4 * C + lrow + kr];
            v0[r] = ga * v0[r] + be; v1[r] = ga * v1[r] + be;
          }
        }
        if (p.c1_out) {
          const int vo0 = own0 ? (lrow * T + t) * 4 : (int)0x80000000, vo1 = own1 ? (lrow * T + t + 1) * 4 : (int)0x80000000;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int kr = (r & 3) + 8 * (r >> 2);
            buf_store(in0 ? v0[r] : 0.f, rc, vo0, kr * Tb);
            buf_store(in1 ? v1[r] : 0.f, rc, vo1, kr * Tb);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kr = (r & 3) + 8 * (r >> 2);
        float a = in0 ? v0[r] : 0.f;  // the zero padding the next conv sees outside the signal
        float c = in1 ? v1[r] : 0.f;
        a = a >= 0.f ? a : an * a;
        c = c >= 0.f ? c : an * c;
        *reinterpret_cast<f32x2*>(&out_l[kr * XS]) = f32x2{a, c};
      }
    } else {
      const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y + rowbase, plane);
      const bool st0 = on0 && pcol < TN, st1 = on1 && pcol + 1 < TN;
      // (pairs are whole: t, T and TN are even; a store past the signal or the tile goes to an out-of-range offset = dropped)
      const int vo = (st0 && st1) ? (lrow * T + t) * 4 : (int)0x80000000;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kr = (r & 3) + 8 * (r >> 2);
        float v0 = o0[r] + bias_l[kr], v1 = o1[r] + bias_l[kr];
        if (p.res) { v0 = (v0 + ev[r].x) * p.res_scale; v1 = (v1 + ev[r].y) * p.res_scale; }
        v0 = in0 ? v0 : 0.f; v1 = in1 ? v1 : 0.f;
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{v0, v1}), ry, vo, kr * Tb, 0);
      }
    }
    if constexpr (D3 && s == 0) {  // conv2 / conv3 weights: their loads have landed under this stage's MFMAs
#pragma unroll
      for (int k = 0; k < WI; k++) {
        const int it = tid + k * NTH;
        *reinterpret_cast<f32x4*>(&U2[it * 4]) = wb[k];
        *reinterpret_cast<f32x4*>(&U3[it * 4]) = wc[D3 ? k : 0];
      }
    }
    if constexpr (!D3 && s == 0) {  // the weight region is free once every wave has left the loop: U3 takes U2's place
      __syncthreads();
#pragma unroll
      for (int k = 0; k < WI; k++) *reinterpret_cast<f32x4*>(&Ua0[(tid + k * NTH) * 4]) = wb[k];
    }
    if constexpr (!last) __syncthreads();
    if (ts_on) tsv[4 + 2 * s] = __builtin_readcyclecounter();
  };
  run_stage(std::integral_constant<int, 0>{});
  run_stage(std::integral_constant<int, 1>{});
  if constexpr (D3) run_stage(std::integral_constant<int, 2>{});
  if (ts_on && lane == 0) {
    long long* o = p.tstamps + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = tsv[i + 1] - tsv[i];
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}
static constexpr size_t kChainwSmem32 = 4 * ((size_t)2 * 32 * (256 + 8) + 32 * 32 * (4 + 2 + 4 + 4) + 5 * 32);
static constexpr size_t kChainwSmem64 = 4 * ((size_t)64 * (128 + 8) + 64 * 64 * 4 + 5 * 64);
// 0: not a shape for the minimal-filtering form; 1: 32 channels, depth 3; 2: 64 channels, depth 2
static int chainw_kind(const ChainArgs& a) {
  if (!a.wino || a.T % 4 || (long)a.C * a.T * 4 >= (1L << 31) || a.force_nc != 0) return 0;
  if (a.C == 32 && a.depth == 3 && a.cv[0].KW == 5 && a.cv[1].KW == 3 && a.cv[2].KW == 3 && a.cv[0].wu && a.cv[1].wu && a.cv[2].wu &&
      a.T >= 252)
    return 1;
  if (a.C == 64 && a.depth == 2 && a.cv[0].KW == 3 && a.cv[1].KW == 3 && a.cv[0].wu && a.cv[1].wu && a.T >= 126 && !a.add &&
      !a.film && !a.c1_out)
    return 2;
  return 0;
}

struct ChainVariant {
  int C, NC, NWV;
  void (*kern)(ChainArgs, int, int);
};
#ifdef OU_EXPERIMENTS
static const ChainVariant kChainVariants[] = {
    {32, 128, 4, conv_chain_kernel<1, 4>},
    {32, 256, 8, conv_chain_kernel<1, 8>},
    {64, 128, 8, conv_chain_kernel<2, 8>},
};
constexpr int kNumChainVariants = sizeof(kChainVariants) / sizeof(kChainVariants[0]);
#else  // the default library fuses with conv_chainw_kernel only (32 channels depth 3, 64 channels depth 2, T % 4 == 0)
static const ChainVariant kChainVariants[1] = {{0, 0, 0, nullptr}};
constexpr int kNumChainVariants = 0;
#endif
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
  if (const int kind = chainw_kind(a)) {  // one wave per SIMD, 224 / 256 MFMAs per wave at ~70 %, ~12 k cycles of everything else
    const int TN = kind == 1 ? 252 : 126;
    const long blocks = (long)a.B * ((a.T + TN - 1) / TN);
    if (nc_out) *nc_out = kind == 1 ? 256 : 128;
    return (double)((blocks + num_cu - 1) / num_cu) * ((kind == 1 ? 224 : 256) * 64.0 / 0.7 + 12000.0);
  }
  if (a.lens) return -1.0;  // (only conv_chainw_kernel zeroes its stages behind a row's own end)
  double best = -1.0;
  for (int i = 0; i < kNumChainVariants; i++) {
    const ChainVariant& v = kChainVariants[i];
    if (v.C != a.C || (a.force_nc && v.NC != a.force_nc)) continue;
    const double c = chain_variant_cost(a, v, num_cu);
    if (best < 0 || c < best) { best = c; if (nc_out) *nc_out = v.NC; }
  }
  return best;
}

hipError_t init_chain_kernels() {
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_chainw_kernel<1, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainwSmem32);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_chainw_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kChainwSmem64);
    if (e != hipSuccess) return e;
  }
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
  if (const int kind = chainw_kind(a)) {  // the minimal-filtering forms: 252 (C = 32, depth 3) / 126 (C = 64, depth 2) finished columns
    const int TN = kind == 1 ? 252 : 126, ntiles = (a.T + TN - 1) / TN;
    if (variant) *variant = 190 + a.depth;
    if (kind == 1) hipLaunchKernelGGL((conv_chainw_kernel<1, true>), dim3(ntiles * a.B), dim3(256), kChainwSmem32, st, a, TN, ntiles);
    else hipLaunchKernelGGL((conv_chainw_kernel<2, false>), dim3(ntiles * a.B), dim3(256), kChainwSmem64, st, a, TN, ntiles);
    return hipGetLastError();
  }
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
    const bool live = q < ragged_len(p.lens, b);  // (ragged batch: zero behind the row's own end)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
      if (m < p.M) p.y[((size_t)b * p.Cout + m) * p.Nq + q] = live ? acc[r] + p.bias[m] : 0.f;
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
  const int rlen = ragged_len(p.lens, b);  // ragged batch: valid output samples of this row
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
      if (p.lens) o = ragged_mask4(o, q0 * R + tl4[u], rlen);
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
      if (q0 * R + tl_[u] >= rlen) v = 0.f;
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
    // (round 6, review item 6: {4, 256, 128, rate_up_kernel<4, 8, 1, 64>, 512, 32} -- the level below, 128 -> 64 channels x 4
    //  phases with its 9-tap FIR and the residual in one launch instead of conv_direct4_kernel 14.5 us + FIR pass 6.5 us --
    //  was built, parity-green, and measured: 8 launches fewer per enhance, the enhance 0.03-0.09 ms SLOWER at batch 1 and 1-2 %
    //  slower at batch 8 (profiles/r06_rate_up_deeper_level_ab.txt): the fused block's phases run back to back, as on the down path)
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


}  // namespace ou
