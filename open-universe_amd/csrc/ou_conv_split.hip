// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: conv_split_kernel -- stride-1 k3 / k5 convs of the
// throughput regime on the BF16 matrix pipe with fp32-faithful operands (round 5, SURVEY.md "Roofline honesty").
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
//
// Why: the f32-input MFMAs run at 64 FLOP/clk/SIMD (157 TFLOP/s), v_mfma_f32_32x32x16_bf16 at 1024 (2.5 PFLOP/s).  An fp32
// value is EXACTLY the sum of three bf16 values (8 + 8 + 8 significand bits: hi = bf16(x), mid = bf16(x - hi), lo =
// bf16(x - hi - mid)), so
//     a * b = (ah + am + al)(bh + bm + bl) = ah bh + [ah bm + am bh] + [ah bl + am bm + al bh] + O(2^-24 |a b|)
// -- six bf16 products per fp32 product, each exact in the pipe's fp32 accumulator, the dropped terms (am bl, al bm, al bl) of
// the size of ONE fp32 rounding of the product.  16 / 6 = 2.67x the f32 MFMA rate at fp32-class accuracy (measured per layer
// against a double evaluation in tools/ubench/split_conv.hip, end to end in the parity tests); fewer accumulator roundings than
// the f32 MFMA's fmaf chain too (one per 16-deep instruction instead of one per product).
//
// Data flow (no change to any tensor layout: x, y stay fp32 (B, C, T)):
//   * weights: a fourth packed copy, split on the host (ou_model.cpp) and laid out as MFMA A fragments:
//       [Cin / 16][KW][Mp / 32][3 pieces][64 lanes][8 bf16]   lane l: row 32 mt + (l & 31), channels 16 cc + 8 (l >> 5) + 0..7
//     one 16-byte load per lane and fragment, 1 KB contiguous per wave -- straight to registers, every wave of a block owns
//     different rows (no LDS for weights);
//   * activations: a block stages 16 channels x (BN + KW - 1) samples per K chunk: 4-byte global loads (coalesced along time),
//     PReLU, the 3-way split on the VALU (v_cvt_pk_bf16_f32: 5.5 instructions per element, once per BLOCK and chunk -- every
//     element then feeds BM x KW x 6 MACs), written to LDS as [piece][K half][sample][8 channels] bf16: the B fragment of tap k is the
//     16-byte read at row (column + k) of the lane's K half -- the taps are row offsets into the same image, no im2col, and the
//     32 lanes of a half read 512 contiguous bytes (no bank conflicts);
//   * per wave a 64 x (32 TNW) output tile: 2 x TNW x KW x 6 MFMAs per chunk (k5, TNW = 4: 240 = 7 680 cycles) against one
//     barrier, 6 KW weight fragments and 3 TNW KW fragment reads from LDS: the loop is bound by the matrix pipe by construction
//     (measured, DESIGN.md 4.1f: an MFMA every 39-46 cycles instead of 32, the card power-limited at 2.0 GHz under this kernel,
//     a fifth of a launch in the epilogue: 1.1-1.4x the fp32 kernels on the layers the launcher's rule gives it, not 2.67x).
//   Summation order per output: channel chunks ascending, taps ascending, the six piece products hi.hi first, 16 channels per
//   instruction in the pipe's own order -- fixed, different from every other family (results agree to fp32 rounding).
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cstdlib>
#include <type_traits>

// scheduling of a step (tuning: the microbenchmark is built with other values; the product with 0)
#ifndef OU_SPLIT_SCHED
#define OU_SPLIT_SCHED 0
#endif

namespace ou {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> three dwords of packed bf16 pairs (element 0 in the low half).  Packed conversion FIRST, the fp32 value of a
// piece re-read from the packed dword (a shift / a mask): 11 instructions per pair -- written through `__bf16` scalars the compiler
// converts every element twice (once alone for the residual, once in the pair): 17, and every VALU instruction beside the MFMAs
// costs about four cycles of the matrix pipe.
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, bf16x2{(__bf16)a, (__bf16)b});  // v_cvt_pk_bf16_f32 (round to nearest even)
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
  H = pack_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(H << 16), r1 = x1 - __uint_as_float(H & 0xFFFF0000u);  // exact
  M = pack_bf16(r0, r1);
  const float q0 = r0 - __uint_as_float(M << 16), q1 = r1 - __uint_as_float(M & 0xFFFF0000u);  // exact
  L = pack_bf16(q0, q1);
}

// (TNW = 2: two blocks per CU -- the other block's MFMAs cover this one's barriers, fragment latencies, staging and epilogue)
// (ACT: the PReLU of the operand path, 6 VALU instructions per staged pair -- most layers of a ConvBlock get their input already
// activated by the producer's epilogue, out_act, and run the variant without it)
template <int KW, int WM, int TNW, bool ACT>
__global__ __launch_bounds__(256, (TNW == 2 && WM >= 2) ? 2 : 1) void conv_split_kernel(ConvArgs p) {
  constexpr int WN = 4 / WM, WTN = 32 * TNW, BN = WN * WTN, PAD = (KW - 1) / 2;
  constexpr int R = BN + KW - 1;          // staged samples per channel and chunk
  // LDS image of one piece: two planes by K half, [half][row][8 channels = 16 bytes] -- the 32 lanes of a fragment half read 512
  // contiguous bytes (a [row][16 channels] image puts lanes i and i + 8 on the same banks: 2-way conflicts on every fragment read,
  // 47 % of the LDS cycles in SQ_LDS_BANK_CONFLICT).  ROWS16: R + 1 rows (one row nobody reads: the halo item of the threads that
  // have none is written there -- no branch in the staging code), padded so that the two planes start on opposite bank halves (the
  // staging writes of a wave go to 8 rows of both planes at once).
  constexpr int ROWS16 = ((R + 1 + 7) / 16) * 16 + 8 >= R + 1 ? ((R + 1 + 7) / 16) * 16 + 8 : ((R + 1 + 7) / 16) * 16 + 24;
  static_assert(ROWS16 >= R + 1 && ROWS16 % 16 == 8, "plane rows");
  constexpr int HALF = ROWS16 * 16;       // bytes of one plane
  constexpr int PIECE = 2 * HALF;         // bytes of one piece
  constexpr int BUF = 3 * PIECE;          // one stage (hi, mid, lo)
  constexpr int NMAIN = BN / 32;          // staged (sample, channel pair) items per thread; + (KW - 1) * 8 halo items on threads 0..
  static_assert(KW == 3 || KW == 5, "k3 / k5");
  static_assert(WM == 1 || WM == 2 || WM == 4, "waves along the rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  // block -> (batch element, column tile, row tile): blocks L, L + 8, ... (one XCD) walk the row tiles of one column tile
  const int L = blockIdx.x, q8 = L >> 3, rg = q8 % p.grid_m, cidx = (q8 / p.grid_m) * 8 + (L & 7);
  const int b = cidx / p.grid_n, ct = cidx - b * p.grid_n;
  if (b >= p.B) return;  // (whole blocks)
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
#ifdef OU_SPLIT_TUNING  // (tools/ubench/split_conv.hip only: the product kernel has no switch that changes its results or timing)
  // stagger experiment: every second block of a CU starts (dbg >> 8) us late, so that one block's epilogue (memory-bound) would
  // run under the other's main loop (matrix-bound) -- measured: no gain (profiles/r05_split_ubench_stagger_experiment.txt)
  if (((p.dbg & 16) && ((L >> 8) & 1)) || ((p.dbg & 32) && ((L >> 3) & 1))) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(p.dbg >> 8) * 100ull) __builtin_amdgcn_s_sleep(32);
  }
#endif
#if OU_SPLIT_SCHED == 4
  __builtin_amdgcn_s_setprio(2);
#endif
  const int n0 = ct * BN, m0 = rg * (64 * WM) + wm * 64;
  const int Tin = p.Tin, Cin = p.Cin;
  const int NCH = Cin >> 4, MT = p.Mp >> 5;
  const float alpha = p.alpha_val;
  const float* xb = p.x + (size_t)b * Cin * Tin;

  // ---- staging of the activation tile
  const int cp = tid & 7;                          // channel pair 2 cp, 2 cp + 1 of the chunk
  const int row0 = tid >> 3;                       // sample rows row0 + 32 j
  float sx[NMAIN + 1][2];
  auto stage_load = [&](int cc) {
    const float* s0 = xb + (size_t)(cc * 16 + 2 * cp) * Tin;
#pragma unroll
    for (int j = 0; j < NMAIN; j++) {
      const int t = n0 - PAD + row0 + 32 * j;
      const bool ok = t >= 0 && t < Tin;
      const int tc = t < 0 ? 0 : (t < Tin ? t : Tin - 1);  // (clamped address + select: no branch)
      const float v0 = s0[tc], v1 = s0[Tin + tc];
      sx[j][0] = ok ? v0 : 0.f;
      sx[j][1] = ok ? v1 : 0.f;
    }
    {  // halo rows BN .. BN + KW - 2: threads 0 .. 8 (KW - 1) - 1
      const int t = n0 - PAD + BN + row0;
      const bool ok = tid < 8 * (KW - 1) && t >= 0 && t < Tin;
      const int tc = t < 0 ? 0 : (t < Tin ? t : Tin - 1);
      const float v0 = s0[tc], v1 = s0[Tin + tc];
      sx[NMAIN][0] = ok ? v0 : 0.f;
      sx[NMAIN][1] = ok ? v1 : 0.f;
    }
  };
  auto stage_store = [&](int buf) {
    unsigned char* base = smem_split + buf * BUF + (cp >> 2) * HALF + (cp & 3) * 4;
#pragma unroll
    for (int j = 0; j <= NMAIN; j++) {
      const int row = j < NMAIN ? row0 + 32 * j : (tid < 8 * (KW - 1) ? BN + row0 : R);
      unsigned H, M, Lo;
      if constexpr (ACT) split_pair(prelu(sx[j][0], alpha), prelu(sx[j][1], alpha), H, M, Lo);
      else split_pair(sx[j][0], sx[j][1], H, M, Lo);
      *reinterpret_cast<unsigned*>(base + row * 16) = H;
      *reinterpret_cast<unsigned*>(base + PIECE + row * 16) = M;
      *reinterpret_cast<unsigned*>(base + 2 * PIECE + row * 16) = Lo;
    }
  };

  // ---- weight fragments: [cc][tap][mt][piece][lane] x 16 bytes
  const u32x4* wsp = reinterpret_cast<const u32x4*>(p.wsplit) + lane;
  const int mt0 = m0 >> 5;
  // fragment registers: two sets, used alternately by consecutive steps (compile-time parity: no copies between steps)
  u32x4 A[2][2][3];
  auto load_a = [&](int step, u32x4 (&a)[2][3]) {  // step = cc * KW + tap
    const u32x4* s = wsp + ((size_t)step * MT + mt0) * 3 * 64;
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
      for (int pc = 0; pc < 3; pc++) a[tm][pc] = s[(tm * 3 + pc) * 64];
  };

  floatx16 acc[2][TNW];
#pragma unroll
  for (int tm = 0; tm < 2; tm++)
#pragma unroll
    for (int tn = 0; tn < TNW; tn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.f;

  const int boff = (lane >> 5) * HALF + (wn * WTN + (lane & 31)) * 16;  // this lane's plane (K half) and row inside a piece
  bf16x8 Bf[2][TNW][3];
  auto read_b = [&](int buf, int tap, bf16x8 (&bf)[TNW][3]) {
    const unsigned char* bb = smem_split + buf * BUF + boff + tap * 16;
#pragma unroll
    for (int tn = 0; tn < TNW; tn++)
#pragma unroll
      for (int pc = 0; pc < 3; pc++) bf[tn][pc] = *reinterpret_cast<const bf16x8*>(bb + pc * PIECE + tn * 32 * 16);
  };

  load_a(0, A[0]);
  stage_load(0);
  stage_store(0);
  __syncthreads();
  read_b(0, 0, Bf[0]);

  // One step = (chunk, tap): 2 x TNW x 6 MFMAs on fragments that were fetched DURING the previous step (weights from L2,
  // activations from LDS) into the other register set.  The scheduling groups spread the step's 6 weight loads and 3 TNW
  // fragment reads over its first MFMAs (left alone the compiler either sinks every load to its first use -- an L2 round trip per
  // step with the matrix pipe idle -- or, behind a plain barrier, issues them in one block while the pipe drains: 700 of 2 300
  // cycles per step, measured with the phase stamps); the split + LDS write of the next chunk's activation tile rides in
  // the last tap's step the same way.
  const bool ts_on = p.tstamps != nullptr;
  long long c_step = 0, c_bar = 0, c_t0 = 0;
  if (ts_on) c_t0 = __builtin_readcyclecounter();
  const long long c_begin = c_t0;
  auto chunk = [&](auto P0c, int cc) {
    constexpr int P0 = decltype(P0c)::value;
    // (no run-time branches inside a chunk: the scheduling groups work within one basic block.  The last chunk stages itself
    // once more into the buffer nobody reads and fetches the last step's weights again.)
    const int ccn = cc + 1 < NCH ? cc + 1 : cc;
    stage_load(ccn);
#pragma unroll
    for (int tap = 0; tap < KW; tap++) {
      constexpr int NMMA = 2 * TNW * 6;
      const int P = (P0 + tap) & 1;
      const bool last = tap == KW - 1;
      u32x4 (&ac)[2][3] = A[P];
      bf16x8 (&bc)[TNW][3] = Bf[P];
      load_a(last ? ccn * KW + (ccn == cc ? tap : 0) : cc * KW + tap + 1, A[P ^ 1]);
      if (!last) read_b(cc & 1, tap + 1, Bf[P ^ 1]);
      if (last) stage_store((cc + 1) & 1);
#if OU_SPLIT_SCHED == 1
      if (!last) __builtin_amdgcn_sched_barrier(0);  // (experiment: the step's loads in one block in front of its MFMAs)
#elif OU_SPLIT_SCHED == 2
      __builtin_amdgcn_sched_barrier(0);             // (experiment: loads AND the staging work in front of the MFMAs)
#endif
      // hi.hi | hi.mid, mid.hi | hi.lo, mid.mid, lo.hi -- two independent accumulators alternate (per product all 2 TNW
      // accumulators in turn instead: the same within 2 %, profiles/r05_split_ubench_mfma_order.txt)
#pragma unroll
      for (int tn = 0; tn < TNW; tn++) {
#pragma unroll
        for (int q = 0; q < 6; q++) {
          constexpr int QA[6] = {0, 0, 1, 0, 1, 2}, QB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
          for (int tm = 0; tm < 2; tm++)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[tm][QA[q]]), bc[tn][QB[q]],
                                                                  acc[tm][tn], 0, 0, 0);
        }
      }
      // issue order of the step: 2 MFMAs, 6 x (weight load, MFMA), 3 TNW x (fragment read, MFMA), the rest of the MFMAs with the
      // staging work of the last tap (VALU + LDS writes) in their shadow
#if OU_SPLIT_SCHED == 0 || OU_SPLIT_SCHED == 4 || (OU_SPLIT_SCHED == 1)
      if (OU_SPLIT_SCHED != 1 || last) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
        for (int i = 0; i < 6; i++) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (!last) {
#pragma unroll
          for (int i = 0; i < 3 * TNW; i++) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, NMMA - 8 - 3 * TNW, 0);
        } else {
          // (NMAIN + 1) items x ~22 VALU (+ 3 LDS writes each, wherever they fall) in the shadow of the remaining MFMAs
#pragma unroll
          for (int i = 0; i < NMMA - 8; i++) {
            __builtin_amdgcn_sched_group_barrier(0x002, ((NMAIN + 1) * 22 + NMMA - 9) / (NMMA - 8), 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (ts_on) {
        const long long now_ = __builtin_readcyclecounter();
        c_step += now_ - c_t0;
        c_t0 = now_;
      }
    }
    __syncthreads();
    read_b((cc + 1) & 1, 0, Bf[(P0 + KW) & 1]);
    if (ts_on) {
      __builtin_amdgcn_sched_barrier(0);
      const long long now_ = __builtin_readcyclecounter();
      c_bar += now_ - c_t0;
      c_t0 = now_;
    }
  };
  {
    int cc = 0;
    for (; cc + 1 < NCH; cc += 2) {
      chunk(std::integral_constant<int, 0>{}, cc);
      chunk(std::integral_constant<int, KW & 1>{}, cc + 1);
    }
    if (cc < NCH) chunk(std::integral_constant<int, 0>{}, cc);
  }
  if (ts_on && lane == 0) {
    long long* o = p.tstamps + ((size_t)blockIdx.x * 4 + wv) * 8;
    o[0] = 0; o[1] = 0; o[2] = c_step; o[3] = c_bar; o[4] = __builtin_readcyclecounter() - c_begin; o[5] = NCH * KW;
  }
#ifdef OU_SPLIT_TUNING
  if (p.dbg & 1) return;  // (main loop alone)
#endif

  // ---- epilogue: in_scale, bias, cond add, FiLM, residual, PReLU of the next layer -- straight from the accumulators
  // (C / D layout of the 32 x 32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
  const int tcol = n0 + wn * WTN + (lane & 31);
  const int rlen = ragged_len(p.lens, b);
  // (all operand loads of a 32-row tile first, with clamped addresses instead of branches: one memory round trip per tile, not
  // one per element -- the first version of this epilogue waited out 128 dependent loads per lane, 95 of its 120 us)
#pragma unroll
  for (int tm = 0; tm < 2; tm++) {
    float bi[16], ga[16], be[16], rs[TNW][16], ad[TNW][16];
    size_t rbase[16];
    int tc[TNW];
#pragma unroll
    for (int tn = 0; tn < TNW; tn++) tc[tn] = tcol + tn * 32 < p.Nq ? tcol + tn * 32 : p.Nq - 1;
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int rc = row < p.M ? row : p.M - 1;
      rbase[r] = ybase + (size_t)rc * p.Tout;
      bi[r] = p.bias[rc];
      ga[r] = 1.f; be[r] = 0.f;
      if (filmb) { ga[r] = filmb[rc]; be[r] = filmb[p.Cout + rc]; }
    }
    if (p.res) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int tn = 0; tn < TNW; tn++) rs[tn][r] = p.res[rbase[r] + tc[tn]];
    }
    if (p.add) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int tn = 0; tn < TNW; tn++) ad[tn][r] = p.add[rbase[r] + tc[tn]];
    }
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
      for (int tn = 0; tn < TNW; tn++) {
        float v = acc[tm][tn][r];
        if (p.in_scale) v *= insc;
        v += bi[r];
        if (p.add) v = (v + ad[tn][r]) * p.add_scale;
        if (filmb) v = ga[r] * v + be[r];
        if (p.res) v = (v + rs[tn][r]) * p.res_scale;
        if (p.out_act) v = v >= 0.f ? v : p.out_alpha * v;
        if (tcol + tn * 32 >= rlen) v = 0.f;  // (ragged batch: behind the row's own end)
        if (row < p.M && tcol + tn * 32 < p.Nq) p.y[rbase[r] + tcol + tn * 32] = v;
      }
    }
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}

#ifdef OU_EXPERIMENTS
// =========================================================================================================
// conv_splitw_kernel<WM, ACT>: F(2, 3) minimal filtering ON the bf16 pipe (round 6; review item 4a) -- `make EXPERIMENTS=1`
// builds only: MEASURED SLOWER than conv_split_kernel on the layers it was built for (256-channel k3 convs, PP16 batch 16, 1 486
// launches of a profiled bench run: 104.5 vs 89.3 us per launch; end to end 368.6 / 369.1 vs 372.6 / 366.3 utt/s, profiles/
// r06_splitw_*).  Two thirds of the piece MFMAs -- but 4/3 of the weight-fragment bytes per chunk for 2/3 of the MFMA time
// (7.8 instead of 3.9 bytes per clock and wave from L2, straight into registers, nothing shared between the waves of a block)
// and twice the staging work: what bounds the split kernels on these layers is the operand stream, not the matrix pipe.
// conv_split_kernel issues KW x 6 piece MFMAs per 16 channels and 32 output columns; it is power- and issue-bound on exactly
// those.  Here a pair of adjacent outputs (y[2p], y[2p + 1]) comes from the KW + 1 = 4 element-wise products of the
// Winograd / Cook-Toom domain,  y = A^T [ (G w) . (B^T d) ]  (same matrices as conv_direct2w_kernel, ou_dev.h / ou_model.cpp):
//   * weights: U = G w, computed by the packer in double, rounded ONCE to fp32 (the values the fp32 minimal-filtering kernels
//     multiply with), then split into three bf16 pieces and laid out as A fragments with the 4 transformed taps in the place of the
//     3 taps (ConvArgs::wsplitw);
//   * activations: the block stages PAIRS: d = x[2p - 1 .. 2p + 2] (PReLU first), V = B^T d in fp32 -- four values per pair
//     and channel --, every V split into three pieces and written to ONE LDS IMAGE PER TRANSFORMED TAP, rows = pairs: the B
//     fragment of tap x is the 16-byte read at the lane's pair row of image x (no row offsets between taps any more);
//   * per 16 channels and 32 PAIRS (64 output columns) 4 x 6 piece MFMAs instead of 2 x 3 x 6: two thirds of the matrix work,
//     for twice the staging work per sample (4 V values per 2 samples), which rides under the MFMAs as before;
//   * the four accumulators of a pair meet in the epilogue: y[2p] = M0 + M1 + M2, y[2p + 1] = M1 - M2 - M3 -- the lane holds
//     both samples: 8-byte stores, 256 contiguous bytes per row and half wave (the plain kernel stores 4 bytes per lane).
// Wave tile 64 rows x 64 pairs (= 128 columns): 2 x 2 x 4 accumulators of 16 registers = 256, the accumulation registers of a
// wave; WM = 4 / 2 waves along the rows (M % 256 / % 128), 1 / 2 along the pairs.  Accuracy: the split is exact for U and V as
// it is for w and x; what is added is the rounding of the fp32 transforms themselves, i.e. what conv_direct2w_kernel has
// (tests: >= 100 dB against the plain split kernel and the oracle).
// =========================================================================================================
template <int WM, bool ACT>
__global__ __launch_bounds__(256, 1) void conv_splitw_kernel(ConvArgs p) {
  constexpr int NTAU = 4, TNP = 2, WN = 4 / WM, WTP = 32 * TNP, BP = WN * WTP;  // pairs per wave / per block
  constexpr int ROWS16 = ((BP + 1 + 7) / 16) * 16 + 8;  // rows of one plane: BP pairs + the row nobody reads, % 16 == 8
  static_assert(ROWS16 >= BP + 1 && ROWS16 % 16 == 8, "plane rows");
  constexpr int HALF = ROWS16 * 16, PIECE = 2 * HALF, IMG = 3 * PIECE, BUF = NTAU * IMG;
  constexpr int NITEM = BP / 32;  // staged (pair, channel pair) items per thread
  static_assert(WM == 2 || WM == 4, "waves along the rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_split[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WN, wn = wv % WN;
  const int L = blockIdx.x, q8 = L >> 3, rg = q8 % p.grid_m, cidx = (q8 / p.grid_m) * 8 + (L & 7);
  const int b = cidx / p.grid_n, ct = cidx - b * p.grid_n;
  if (b >= p.B) return;  // (whole blocks)
  if (p.prof && tid == 0) atomicMin(p.prof + (blockIdx.x & 15), (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const int p0 = ct * BP, m0 = rg * (64 * WM) + wm * 64;
  const int Tin = p.Tin, Cin = p.Cin;
  const int NCH = Cin >> 4, MT = p.Mp >> 5;
  const float alpha = p.alpha_val;
  const float* xb = p.x + (size_t)b * Cin * Tin;

  // ---- staging: thread = (channel pair cp, pair rows row0 + 32 j); d = x[2 P - 1 .. 2 P + 2] of both channels
  const int cp = tid & 7, row0 = tid >> 3;
  float sx[NITEM][2][4];
  auto stage_load = [&](int cc) {
    const float* s0 = xb + (size_t)(cc * 16 + 2 * cp) * Tin;
#pragma unroll
    for (int j = 0; j < NITEM; j++) {
      const int t0 = 2 * (p0 + row0 + 32 * j) - 1;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int t = t0 + e;
        const bool ok = t >= 0 && t < Tin;
        const int tc = t < 0 ? 0 : (t < Tin ? t : Tin - 1);  // (clamped address + select: no branch)
        const float v0 = s0[tc], v1 = s0[Tin + tc];
        sx[j][0][e] = ok ? v0 : 0.f;
        sx[j][1][e] = ok ? v1 : 0.f;
      }
    }
  };
  auto stage_store = [&](int buf) {
    unsigned char* base = smem_split + buf * BUF + (cp >> 2) * HALF + (cp & 3) * 4;
#pragma unroll
    for (int j = 0; j < NITEM; j++) {
      const int row = row0 + 32 * j;
      float d0[4], d1[4], V0[4], V1[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        d0[e] = ACT ? prelu(sx[j][0][e], alpha) : sx[j][0][e];
        d1[e] = ACT ? prelu(sx[j][1][e], alpha) : sx[j][1][e];
      }
      wino_bt<3>(d0, V0);
      wino_bt<3>(d1, V1);
#pragma unroll
      for (int x = 0; x < NTAU; x++) {
        unsigned H, M, Lo;
        split_pair(V0[x], V1[x], H, M, Lo);
        *reinterpret_cast<unsigned*>(base + x * IMG + row * 16) = H;
        *reinterpret_cast<unsigned*>(base + x * IMG + PIECE + row * 16) = M;
        *reinterpret_cast<unsigned*>(base + x * IMG + 2 * PIECE + row * 16) = Lo;
      }
    }
  };

  // ---- weight fragments: [cc][transformed tap][mt][piece][lane] x 16 bytes
  const u32x4* wsp = reinterpret_cast<const u32x4*>(p.wsplitw) + lane;
  const int mt0 = m0 >> 5;
  u32x4 A[2][2][3];
  auto load_a = [&](int step, u32x4 (&a)[2][3]) {  // step = cc * 4 + x
    const u32x4* s = wsp + ((size_t)step * MT + mt0) * 3 * 64;
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
      for (int pc = 0; pc < 3; pc++) a[tm][pc] = s[(tm * 3 + pc) * 64];
  };

  floatx16 acc[2][TNP][NTAU];
#pragma unroll
  for (int tm = 0; tm < 2; tm++)
#pragma unroll
    for (int tn = 0; tn < TNP; tn++)
#pragma unroll
      for (int x = 0; x < NTAU; x++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][tn][x][r] = 0.f;

  const int boff = (lane >> 5) * HALF + (wn * WTP + (lane & 31)) * 16;  // this lane's plane (K half) and pair row inside a piece
  bf16x8 Bf[2][TNP][3];
  auto read_b = [&](int buf, int x, bf16x8 (&bf)[TNP][3]) {
    const unsigned char* bb = smem_split + buf * BUF + x * IMG + boff;
#pragma unroll
    for (int tn = 0; tn < TNP; tn++)
#pragma unroll
      for (int pc = 0; pc < 3; pc++) bf[tn][pc] = *reinterpret_cast<const bf16x8*>(bb + pc * PIECE + tn * 32 * 16);
  };

  load_a(0, A[0]);
  stage_load(0);
  stage_store(0);
  __syncthreads();
  read_b(0, 0, Bf[0]);

  // one step = (chunk, transformed tap): 2 x TNP x 6 MFMAs on fragments fetched during the previous step (see conv_split_kernel)
  auto chunk = [&](int cc) {
    const int ccn = cc + 1 < NCH ? cc + 1 : cc;
    stage_load(ccn);
#pragma unroll
    for (int x = 0; x < NTAU; x++) {
      constexpr int NMMA = 2 * TNP * 6;
      const int P = x & 1;
      const bool last = x == NTAU - 1;
      u32x4 (&ac)[2][3] = A[P];
      bf16x8 (&bc)[TNP][3] = Bf[P];
      load_a(last ? ccn * NTAU + (ccn == cc ? x : 0) : cc * NTAU + x + 1, A[P ^ 1]);
      if (!last) read_b(cc & 1, x + 1, Bf[P ^ 1]);
      if (last) stage_store((cc + 1) & 1);
#pragma unroll
      for (int tn = 0; tn < TNP; tn++) {
#pragma unroll
        for (int q = 0; q < 6; q++) {
          constexpr int QA[6] = {0, 0, 1, 0, 1, 2}, QB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
          for (int tm = 0; tm < 2; tm++)
            acc[tm][tn][x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ac[tm][QA[q]]), bc[tn][QB[q]],
                                                                     acc[tm][tn][x], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
      for (int i = 0; i < 6; i++) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if (!last) {
#pragma unroll
        for (int i = 0; i < 3 * TNP; i++) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMMA - 8 - 3 * TNP, 0);
      } else {
        // NITEM items x (8 transform + 4 x 11 split [+ 16 PReLU] VALU, 12 LDS writes) in the shadow of the remaining MFMAs
#pragma unroll
        for (int i = 0; i < NMMA - 8; i++) {
          __builtin_amdgcn_sched_group_barrier(0x002, (NITEM * (ACT ? 70 : 54) + NMMA - 9) / (NMMA - 8), 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    read_b((cc + 1) & 1, 0, Bf[0]);
  };
  for (int cc = 0; cc < NCH; cc++) chunk(cc);

  // ---- epilogue: A^T, then in_scale, bias, cond add, FiLM, residual, PReLU of the next layer -- 8 bytes per lane and row
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
  const float* filmb = p.film ? p.film + (size_t)b * p.film_bstride : nullptr;
  const size_t ybase = (size_t)b * p.Cout * p.Tout;
  const float insc = p.in_scale ? p.in_scale[b] : 1.0f;
  const int rlen = ragged_len(p.lens, b);
  // (half a 32 x 32-pair tile at a time -- 8 accumulator registers = 8 rows: its operand loads first, one memory round trip
  //  per half tile, 32-bit row offsets; the 256 accumulation registers leave the epilogue half the register file)
#pragma unroll
  for (int tm = 0; tm < 2; tm++) {
#pragma unroll
    for (int tn = 0; tn < TNP; tn++) {
      const int t = 2 * (p0 + wn * WTP + tn * 32 + (lane & 31));
      const bool two = t + 1 < p.Nq;
#pragma unroll
      for (int rh = 0; rh < 2; rh++) {
        float bi[8], ga[8], be[8];
        f32x2 rs[8], ad[8];
        int roff[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int r = 8 * rh + q;
          const int row = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int rc = row < p.M ? row : p.M - 1;
          roff[q] = rc * p.Tout;
          bi[q] = p.bias[rc];
          ga[q] = 1.f; be[q] = 0.f;
          if (filmb) { ga[q] = filmb[rc]; be[q] = filmb[p.Cout + rc]; }
        }
        // (a row of odd length ends in half a pair: its last lane takes the single sample t = Nq - 1 as component 0)
        const int t1 = t < p.Nq ? t : p.Nq - 1;
        if (p.res) {
#pragma unroll
          for (int q = 0; q < 8; q++)
            rs[q] = two ? f32x2(*reinterpret_cast<const f32x2u*>(p.res + ybase + roff[q] + t)) : f32x2{p.res[ybase + roff[q] + t1], 0.f};
        }
        if (p.add) {
#pragma unroll
          for (int q = 0; q < 8; q++)
            ad[q] = two ? f32x2(*reinterpret_cast<const f32x2u*>(p.add + ybase + roff[q] + t)) : f32x2{p.add[ybase + roff[q] + t1], 0.f};
        }
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int r = 8 * rh + q;
          const int row = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const float a0 = acc[tm][tn][0][r], a1 = acc[tm][tn][1][r], a2 = acc[tm][tn][2][r], a3 = acc[tm][tn][3][r];
          f32x2 v = f32x2{(a0 + a1) + a2, (a1 - a2) - a3};
          if (p.in_scale) v *= insc;
          v += bi[q];
          if (p.add) v = (v + ad[q]) * p.add_scale;
          if (filmb) v = ga[q] * v + be[q];
          if (p.res) v = (v + rs[q]) * p.res_scale;
          if (p.out_act) { v[0] = v[0] >= 0.f ? v[0] : p.out_alpha * v[0]; v[1] = v[1] >= 0.f ? v[1] : p.out_alpha * v[1]; }
          if (t >= rlen) v[0] = 0.f;      // (ragged batch: behind the row's own end)
          if (t + 1 >= rlen) v[1] = 0.f;
          if (row < p.M) {
            if (two) *reinterpret_cast<f32x2u*>(p.y + ybase + roff[q] + t) = v;
            else if (t < p.Nq) p.y[ybase + roff[q] + t] = v[0];
          }
        }
      }
    }
  }
  if (p.prof && tid == 0) atomicMin(p.prof + 16 + (blockIdx.x & 15), ~(unsigned long long)__builtin_amdgcn_s_memrealtime());
}
#endif  // OU_EXPERIMENTS

namespace {
struct SplitCfg {
  int KW, WM, TNW;
  void (*kern)(ConvArgs);      // operand path without PReLU
  void (*kern_act)(ConvArgs);  // ... with it
  size_t lds;
};
template <int KW, int WM, int TNW>
constexpr SplitCfg split_cfg() {
  // two staging buffers of three pieces, two K-half planes each (>= the kernel's 2 BUF; see ROWS16)
  return {KW, WM, TNW, conv_split_kernel<KW, WM, TNW, false>, conv_split_kernel<KW, WM, TNW, true>,
          (size_t)2 * 3 * 2 * (((((4 / WM) * 32 * TNW + KW) + 7) / 16) * 16 + 24) * 16};
}
const SplitCfg kSplitCfgs[] = {
    split_cfg<3, 1, 4>(), split_cfg<3, 2, 4>(), split_cfg<3, 4, 4>(), split_cfg<3, 1, 2>(), split_cfg<3, 2, 2>(), split_cfg<3, 4, 2>(),
    split_cfg<5, 1, 4>(), split_cfg<5, 2, 4>(), split_cfg<5, 4, 4>(), split_cfg<5, 1, 2>(), split_cfg<5, 2, 2>(), split_cfg<5, 4, 2>(),
};
}  // namespace

constexpr size_t splitw_lds(int wm) {  // two stages x 4 images x 3 pieces x 2 planes
  return (size_t)2 * 4 * 3 * 2 * ((((4 / wm) * 64 + 1 + 7) / 16) * 16 + 8) * 16;
}
hipError_t init_split_kernels() {
#ifdef OU_EXPERIMENTS
  {
    const void* ks[4] = {reinterpret_cast<const void*>(conv_splitw_kernel<4, false>), reinterpret_cast<const void*>(conv_splitw_kernel<4, true>),
                         reinterpret_cast<const void*>(conv_splitw_kernel<2, false>), reinterpret_cast<const void*>(conv_splitw_kernel<2, true>)};
    for (int i = 0; i < 4; i++) {
      hipError_t e = hipFuncSetAttribute(ks[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)splitw_lds(i < 2 ? 4 : 2));
      if (e != hipSuccess) return e;
    }
  }
#endif
  for (const SplitCfg& c : kSplitCfgs) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(c.kern_act), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.lds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// cfg_out: 800 + 100 (TNW == 2) + 10 log2(WM) + KW
hipError_t launch_conv_split(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out) {
  if (!a.wsplit || a.stride != 1 || a.up != 1 || (a.KW != 3 && a.KW != 5) || a.pad != (a.KW - 1) / 2 || a.fir || a.Cin % 16 ||
      a.M % 64 || (a.in_scale != nullptr && a.act))
    return hipErrorInvalidConfiguration;
#ifdef OU_EXPERIMENTS
  // minimal filtering on the bf16 pipe where the layer has the Winograd-domain split copy (k3, rows a multiple of 128) and
  // the 128-column wave tiles still fill the device: variant 850 + 10 log2(WM) + KW
  if (a.wsplitw && a.split_wino && a.KW == 3 && a.M % 128 == 0 && (a.force_cfg < 0 || (a.force_cfg >= 850 && a.force_cfg < 900))) {
    int wmw = a.M % 256 == 0 ? 4 : 2;
    if (a.force_cfg >= 850) wmw = 1 << ((a.force_cfg - 850) / 10);
    if ((wmw == 4 || wmw == 2) && a.M % (64 * wmw) == 0) {
      const long bn = (4 / wmw) * 128L;
      const long blocks = (a.M / (64 * wmw)) * ((a.Nq + bn - 1) / bn) * a.B;
      if (blocks >= num_cu || a.force_cfg >= 850 || a.split == 1) {
        ConvArgs aa = a;
        aa.grid_m = a.M / (64 * wmw);
        aa.grid_n = (int)((a.Nq + bn - 1) / bn);
        const long total8 = ((long)aa.grid_n * a.B + 7) / 8 * 8;
        if (cfg_out) *cfg_out = 850 + 10 * (wmw == 4 ? 2 : 1) + a.KW;
        auto kern = wmw == 4 ? (a.act ? conv_splitw_kernel<4, true> : conv_splitw_kernel<4, false>)
                             : (a.act ? conv_splitw_kernel<2, true> : conv_splitw_kernel<2, false>);
        hipLaunchKernelGGL(kern, dim3((unsigned)(total8 * aa.grid_m)), dim3(256), splitw_lds(wmw), stream, aa);
        return hipGetLastError();
      }
    }
    if (a.force_cfg >= 850) return hipErrorInvalidConfiguration;
  }
#endif
  int wm = a.M % 256 == 0 ? 4 : (a.M % 128 == 0 ? 2 : 1);
  int tnw = 4;
  auto blocks = [&](int wm_, int tnw_) {
    const long bn = (4 / wm_) * 32L * tnw_;
    return ((a.M + 64 * wm_ - 1) / (64 * wm_)) * ((a.Nq + bn - 1) / bn) * a.B;
  };
  // fill the device: narrower wave tiles, then fewer rows per block, while there are fewer blocks than CUs
  if (blocks(wm, tnw) < num_cu) tnw = 2;
  if (a.force_cfg >= 800 && a.force_cfg < 1100 && !(a.force_cfg >= 850 && a.force_cfg < 900)) {
    const int f = a.force_cfg - 800;
    tnw = f >= 100 ? 2 : 4;
    wm = 1 << ((f % 100) / 10);
  }
  const SplitCfg* c = nullptr;
  for (const SplitCfg& k : kSplitCfgs)
    if (k.KW == a.KW && k.WM == wm && k.TNW == tnw) c = &k;
  if (!c || a.M % (64 * wm)) return hipErrorInvalidConfiguration;
  ConvArgs aa = a;
  const long bn = (4 / wm) * 32L * tnw;
  aa.grid_m = a.M / (64 * wm);
  aa.grid_n = (int)((a.Nq + bn - 1) / bn);
  const long total8 = ((long)aa.grid_n * a.B + 7) / 8 * 8;
  if (cfg_out) *cfg_out = 800 + (tnw == 2 ? 100 : 0) + 10 * (wm == 4 ? 2 : (wm == 2 ? 1 : 0)) + a.KW;
  hipLaunchKernelGGL(a.act ? c->kern_act : c->kern, dim3((unsigned)(total8 * aa.grid_m)), dim3(256), c->lds, stream, aa);
  return hipGetLastError();
}

}  // namespace ou
