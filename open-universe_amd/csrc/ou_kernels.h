// Launch wrappers of the gfx950 kernels (ou_kernels.hip).  Host-callable, everything enqueued on `stream`.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ou {

// ---- generic fp32-MFMA implicit-GEMM Conv1d -------------------------------------------------------------
//   y[b][co][q*up + p] = epi( bias[co] + sum_{ci,k} W[co*up+p][ci][k] * act(in_scale[b] * x[b][ci][q*stride + k - pad]) )
//   epi(v): v = (v + add)*add_scale ; v = gamma*v + beta (FiLM) ; v = (v + res)*res_scale     (each optional)
struct ConvArgs {
  const float* x = nullptr;       // (B, Cin, Tin)
  const float* w = nullptr;       // packed [Cin/CK][KW][CK][Mp]
  const float* wd = nullptr;      // second copy, taps innermost: [Cin][Mp][4 (k3) / 8 (k5)] (conv_direct2_kernel) or null
  const float* wu = nullptr;      // third copy, Winograd domain U = G w: [Cin][Mp][4 (F(2,3)) / 8 (F(2,5): 6 used)] or null
  const void* wsplit = nullptr;   // fourth copy, three bf16 pieces per weight as MFMA A fragments: [Cin/16][KW][Mp/32][3][64][8 bf16]
                                  // (conv_split_kernel) or null
  const void* wsplitw = nullptr;  // fifth copy: the Winograd-domain weights U = G w the same way, [Cin/16][KW + 1][Mp/32][3][64][8 bf16]
                                  // (conv_splitw_kernel) or null
  int split_wino = 1;             // option split_wino: 0 = conv_split_kernel also where conv_splitw_kernel could take the layer
  const float* bias = nullptr;    // [Cout]
  float* y = nullptr;             // (B, Cout, Tout)
  const float* in_scale = nullptr;  // [B] or null
  int act = 0;                      // PReLU prologue on/off
  float alpha_val = 0.f;            // its slope (by value: no dependent scalar load at kernel entry)
  const float* add = nullptr;       // (B, Cout, Tout) or null
  const float* film = nullptr;      // gamma at film[b*film_bstride + co], beta at [.. + Cout + co]
  const float* res = nullptr;       // (B, Cout, Tout) or null
  float add_scale = 1.f, res_scale = 1.f;
  int film_bstride = 0;
  int B = 1, Cin = 0, Tin = 0, Cout = 0, M = 0, Mp = 0, KW = 1, stride = 1, pad = 0, up = 1, CK = 2;
  int Nq = 0;    // GEMM columns (output positions per row)
  int Tout = 0;  // output length (<= Nq*up)
  unsigned magic_span[3] = {0, 0, 0};  // filled by launch_conv: 2^32/span + 1 for BN = 128 / 64 / 32
  unsigned magic_up = 0;               // 2^32/up + 1 (0 when up == 1)
  int SC = 1;                          // filled by launch_conv: packed chunks per pipeline stage
  int grid_n = 1, grid_m = 1;          // filled by launch_conv: tile counts (1-D grid, XCD-aware mapping)
  int xcd_map = 0;                     // filled by launch_conv: block -> tile mapping (see the kernel)
  int force_cfg = -1, force_sc = 0;    // tuning overrides (ou_bench_conv)
  int force_xcd_map = -1;              // tuning: 0 / 1 / 2
  double tile_min = -1.0;              // OU_TILE_MIN (< 0: the launcher's default of 1.2 wave tiles per SIMD)
  int tile_prefetch = 1;               // OU_TILE_PREFETCH: LDS prefetch of the epilogue operand in conv_direct3_kernel
  int direct = 5;                      // OU_CONV_DIRECT: 0 = never use the register-direct kernels, 1 = only the first
                                       // generation (dword loads), 2 = + wide-load split-K variant where a layer has `wd`,
                                       // 3 = + the no-split-K throughput kernel (conv_direct3_kernel) for many-column launches,
                                       // 4 = + the wide-load split-K kernel for 1x1 / phase-GEMM / rate-change layers
                                       //     (conv_direct4_kernel),
                                       // 5 = + minimal filtering F(2, 3) / F(2, 5) for the k3 / k5 layers on 64-column tiles
                                       //     (conv_direct2w_kernel; default)
  int d2_map = -1;                     // OU_D2_MAP: block -> tile mapping of the wide-load split-K kernels only (tuning)
  int split = -1;                      // OU_SPLIT: -1 = the bf16-split kernel (conv_split_kernel) where the launcher's rule says so,
                                       // 0 = never, 1 = wherever a layer has the split copy (tests / tuning)
  int wino = 1;                        // OU_WINO=0: never use the minimal-filtering variants (conv_direct2w_kernel, ...)
  int d2_wk = 0;                       // OU_D2_WK = 4 / 8: K slices (waves per block) of conv_direct2_kernel (0: the launcher's rule)
  int d4_fir_unfused = 1;              // OU_D4_FIR=0: up convs with a fusable FIR stay on the first-generation fused kernel
  int d4_short = 1;                    // OU_D4_SHORT=0: the 401-frame levels at batch 1 stay on the first-generation kernels
  int d4_force = 0;                    // OU_D4_FORCE = 10 TM + log2(WK): that tile shape wherever a layer admits it (tests / tuning)
  int d2_tile_rule = 1;                // option d2_tile_rule: 0 = round 5's tile-width rule of the wide-load split-K kernels (A / B)
  int deep_factor = 8;                 // option deep_factor: up to this many blocks of 64 x 128 per CU a layer is "deep" (split-K kernels)
  // Anti-alias FIR of the up path fused into the epilogue (direct kernel, up > 1, KW == 1 only; launch_conv returns
  // hipErrorNotSupported otherwise and the caller runs launch_fir after a plain launch):
  //   y = FIR_{2 up + 1}(u) + bias ; y = res ? (y + res) * res_scale : y,   u = the transposed conv's output WITHOUT bias
  // Activation moved from the consumer's operand path to the producer's epilogue (round 5): with out_act the kernel stores
  // prelu(y; out_alpha) INSTEAD of y -- for tensors whose only reader is the next PReLU_Conv (conv1 -> conv2 -> conv3 inside a
  // ConvBlock, blocks.py:395-399); that conv then runs with act = 0 and its loop has no PReLU (every input element used to be
  // activated again by every row group and every overlapping window, and each VALU instruction beside fp32 MFMAs costs ~3 cycles
  // of the matrix pipe: tools/ubench/mfma_valu_mix.hip).  Same arithmetic on the same values: bit-identical.
  // Kernels without the epilogue form return hipErrorNotSupported (the caller stores y and keeps act = 1 downstream).
  int out_act = 0;
  float out_alpha = 1.f;
  // Ragged batch (ou_enhance_var): valid OUTPUT samples per batch row, [B] on the device, or null (all rows whole).  The kernel
  // families listed in conv_masks_rows() store 0 from lens[b] on themselves; after the others the runner launches
  // launch_mask_tail.  (Inputs need nothing: they are zero behind their rows by the same invariant.)
  const int* lens = nullptr;
  const float* fir = nullptr;
  int fir_len = 0;
  int tile_bm = 32, tile_bn = 0, tile_halo = 0;  // filled by launch_conv_direct: rows / columns a tile advances by, left halo
  int dbg = 0;                         // phase ablation switches (tuning only)
  long long* tstamps = nullptr;        // per-wave phase cycle counts (tuning only)
  unsigned long long* prof = nullptr;  // measurement: 16 x min block start, 16 x ~max block end (slot = block id & 15: 500
                                       // same-address atomics in half a microsecond stall an L2 channel), 10 ns ticks
};
// returns hipSuccess or the launch error; `cfg_out` (optional) receives the tile configuration index used
hipError_t launch_conv(const ConvArgs& a, int num_cu, hipStream_t stream, int* cfg_out = nullptr);
// does the kernel behind variant code `cfg` (launch_conv's cfg_out) honour ConvArgs::lens in its epilogue?
bool conv_masks_rows(int cfg);
hipError_t init_conv_kernels();  // raises the dynamic-LDS limit of every instantiation
double direct3_tiles_per_simd(int M, int Nq, int B, int num_cu);  // wave tiles per SIMD of the no-split-K throughput kernel
// Small-K rate-change conv of the wide levels with the anti-alias FIR fused: y = conv_{k=s=r}(FIR(prelu(x))) + bias
// (a.fir = taps or null, a.fir_len = 2r + 1; a.act / a.alpha_val = the PReLU).  hipErrorNotSupported = use launch_fir +
// launch_conv.
bool rate_down_supported(const ConvArgs& a);
hipError_t launch_rate_down(const ConvArgs& a, hipStream_t st, int* cfg_out = nullptr);
// ... and the last up conv: y = FIR(convT_{k=s=r}(prelu(x))) + bias, then the residual (a.fir = taps or null, a.bias = the
// bias added after the filter)
bool rate_up_supported(const ConvArgs& a);
hipError_t launch_rate_up(const ConvArgs& a, hipStream_t st, int* cfg_out = nullptr);

// ---- small VALU kernels ------------------------------------------------------------------------------------
// Per-step scalars of the sampler / EDM wrapper (universe.py:175-209, 333-343), one row per batch element
// (or one shared row when bstride == 0).  All fp32, computed on the host in the reference's op order.
struct StepCoef {
  float w_in, sigma_net, w_skip, w_out, sig2, c1, s_next, beta;
};
enum OutMode { OUT_SCORE = 0, OUT_UPDATE = 1 };
// Conv1d(1 -> C, k) 'same' on w_in[b]*x (w_in from the coefficient row, 1 when coef == null)
//   score.py:243-245,284 ; condition.py:295-300,360
hipError_t launch_in_conv(const float* x, const float* w, const float* bias, const StepCoef* coef, int coef_bstride,
                          float* y, int B, int C, int T, int KW, hipStream_t s, const int* lens = nullptr);
// PReLU -> PReLU -> Conv1d(C -> 1, k) fused with the EDM score and the sampler update:
//   net -> score = edm ? (w_skip*x + w_out*net - x)/sig2 : net
//   OUT_SCORE : out = score ;  OUT_UPDATE : out = x + c1*score + beta*(noise*s_next)   (noise may be null)
hipError_t launch_out_conv(const float* s, const float* w, const float* bias, const float* alphas, const float* x,
                           const float* noise, float* out, const StepCoef* coef, int coef_bstride, int edm,
                           int mode, int B, int C, int T, int KW, hipStream_t st, const int* lens = nullptr);

// Noise-level embedding for S sigma rows (sigma_block.py) -> g (S, D)
hipError_t launch_sigma_embed(const StepCoef* coef, int S, const float* params, int simple, int n_rff, int D,
                              float* g, hipStream_t st);
// All FiLM projections at once: film[s][r] = W[r][:] . g[s][:] + b[r]
hipError_t launch_film(const float* g, const float* W, const float* b, float* film, int S, int rows, int D,
                       hipStream_t st);
// Upload up to 128 StepCoef rows passed by value (graph-capture safe, no host memory involved)
struct CoefBlock { StepCoef c[64]; };
hipError_t launch_upload_coef(StepCoef* dst, const CoefBlock& blk, int n, hipStream_t st);

// ---- batches whose rows have lengths of their own ("ragged": ou_enhance_var) --------------------------------------------
// Row b of the batch is the utterance it would be in a call of its own: its own pad() split (universe.py:219-223), its own
// statistics, and 'same' zero padding right behind ITS last sample on every level.  The invariant that carries this through
// the network: every activation tensor is ZERO from the row's own length on (len_l[b] = t_pad[b] * T_l / T on the level of
// length T_l), so that a halo read past a row's end sees what the reference's conv padding would supply.  Kernels that take a
// `lens` pointer keep the invariant themselves (null = all rows have the tensor's length); after the others the runner
// launches launch_mask_tail.
struct RowInfo { int t_raw, pad_left, t_pad, _r; };
constexpr int kMaxLenLevels = 8;
struct LevelSpec { int n = 0; int num[kMaxLenLevels]; int den[kMaxLenLevels]; };  // len_l[b] = t_pad[b] * num / den
struct RowBlock { int t_raw[64]; };
// rows[off .. off + n) <- {t_raw, pad_left, t_pad} (universe.py:219-223) ; lens[l * B + off + i] <- t_pad * num_l / den_l
hipError_t launch_upload_rows(RowInfo* rows, int* lens, const RowBlock& blk, int n, int off, int B, int tot_ds,
                              const LevelSpec& lv, hipStream_t st);
// y[b][c][t] = 0 for t >= lens[b]   (y: (B, C, T))
hipError_t launch_mask_tail(float* y, const int* lens, int B, int C, int T, hipStream_t st);
// GRU input projections gx (B, 6H, T) of a ragged batch, t >= lens[b]: 0 in the r / n rows and +1e4 in the z rows of both
// directions -- z = sigmoid(1e4 + ..) is exactly 1, so h' = (h - n) z + n holds the state: the backward pass reaches the row's
// last frame with h = 0 exactly, as the reference's pass over that row alone starts (the forward pass's frames past the end
// are masked afterwards).  The recurrence kernels need no per-row length.
hipError_t launch_gru_tail_fill(float* gx, const int* lens, int B, int H, int T, hipStream_t st);
// pad + normalize / unpad + post with per-row geometry (rows: RowInfo[B]); mix and out are (B, T_raw_max), y / x (B, T_pad_max)
hipError_t launch_pad_normalize_var(const float* mix, float* y, float* stats, const RowInfo* rows, int B, int T_raw_max,
                                    int T_pad_max, float level, hipStream_t st);
hipError_t launch_post_var(const float* x, const float* stats, float* out, const RowInfo* rows, int B, int T_raw_max,
                           int T_pad_max, int keep_rms, int peak_guard, hipStream_t st);

// pad (universe.py:219-223) + normalize_batch (utils/norm.py:47-87).  stats[b] = {mean, gain, mix_rms, 0}
hipError_t launch_pad_normalize(const float* mix, float* y, float* stats, int B, int T_raw, int T_pad, int pad_left,
                                float level, hipStream_t st);
// unpad + keep_rms + peak guard (universe.py:349-357)
hipError_t launch_post(const float* x, const float* stats, float* out, int B, int T_raw, int T_pad, int pad_left,
                       int keep_rms, int peak_guard, hipStream_t st);
// x0 = sigma0 * noise (universe.py:326) ; optionally x0 = base + sigma*noise (warm start :330)
hipError_t launch_init_x(const float* noise, const float* base, float sigma, float* x, size_t n, hipStream_t st);

// x += c1*score + c2*z  (universe.py:339,343; z may be null) -- the stand-alone form of the update that ou_enhance fuses
// into the output conv
hipError_t launch_sampler_step(float* x, const float* score, const float* z, float c1, float c2, size_t n,
                               hipStream_t st);

// CompressedMagSTFT forward / inverse (layers/dyn_range_comp.py:51-225).  type: 0 none, 1 exponent, 2 log.
//   forward: x (B, T) -> out (B, 2F, n_frames), F = n_fft/2 + 1, n_frames = 1 + T/hop-ish (center=True)
//   inverse: spec (B, 2F, n_frames) -> y (B, length); `frames` = scratch of B * n_frames * n_fft floats
hipError_t launch_stft_forward(const float* x, const float* win, float* out, int B, int T, int n_fft, int hop,
                               int type, float e, float factor, hipStream_t st);
hipError_t launch_stft_inverse(const float* spec, const float* win, float* frames, float* y, int B, int n_frames,
                               int n_fft, int hop, int type, float e, float factor, int length, hipStream_t st);

// mel front-end (condition.py:92-108): power STFT -> mel fb ; esum[b][frame] = sum_mel mel^2
hipError_t launch_mel(const float* x, const float* win, const float* tw, const float* fb, float* mel, float* esum,
                      int B, int T, int n_fft, int hop, int pad_left, int n_freq, int n_mels, int L, hipStream_t st);
// (`lens`: frames per row of a ragged batch or null -- the mean runs over the row's own frames)
hipError_t launch_mel_scale(const float* esum, float* scale, int B, int L, hipStream_t st, const int* lens = nullptr);

// space-to-depth + PReLU for the conditioner's strided "st" convs: y[b][ci*R + k][q] = prelu(x[b][ci][q*R + k])
hipError_t launch_s2d(const float* x, const float* alpha, float* y, int B, int C, int T, int R, hipStream_t st);
// (`lens` of launch_in_conv / launch_out_conv / launch_fir: per-row valid lengths of a ragged batch or null, see ConvArgs::lens)
// Binomial anti-alias FIR (blocks.py:119-130, depthwise 'same' conv with 2r+1 taps), bandwidth-bound:
//   pre  (down path): y = FIR(prelu(x))                       blocks.py:211-215
//   post (up path)  : y = FIR(u) + bias[c] ; y = res ? (y + res)*res_scale : y     blocks.py:219-225, 374-376
hipError_t launch_fir(const float* x, const float* taps, int ntaps, float alpha, int act, const float* bias,
                      const float* res, float res_scale, float* y, int B, int C, int T, hipStream_t st,
                      const int* lens = nullptr);
// y = (a + b + c + d + e) * scale   (nulls skipped)   condition.py:202-206
hipError_t launch_sum(const float* a, const float* b, const float* c, const float* d, const float* e, float scale,
                      float* y, size_t n, hipStream_t st);

// Bidirectional GRU recurrence on a cluster of H/64 workgroups per (batch, direction).
//   gx : (B, 6H, T) input projection incl. biases ; out: (B, 2H, T) ; out = res ? (h + res)*scale : h
// Fused body of a ConvBlock (blocks.py:377-399) for the wide, shallow levels (C = 32 / 64): two or three stride-1
// 'same' convs chained through LDS-resident activation tiles, see conv_chain_kernel.
struct ChainConv {
  const float* wu = nullptr;    // Winograd-domain copy [C][Mp][4 | 8] (conv_chainw_kernel) or null
  const float* w = nullptr;     // packed generic-conv weights [C/CK][KW][CK][Mp]
  const float* bias = nullptr;  // [C]
  float alpha = 0.f;            // PReLU slope applied to this conv's INPUT
  int KW = 3, CK = 32;
};
struct ChainArgs {
  const float* x = nullptr;    // (B, C, T) input of the first fused conv
  float* y = nullptr;          // (B, C, T) block output
  const float* res = nullptr;  // residual added after the last conv: y = (conv + bias + res) * res_scale
  float res_scale = 1.f;
  const float* add = nullptr;  // depth 3 only: y1 = (conv1 + bias + add) * add_scale, then FiLM
  float add_scale = 1.f;
  const float* film = nullptr; // depth 3 only: [B][film_bstride] = gamma[C] | beta[C]
  int film_bstride = 0;
  float* c1_out = nullptr;     // depth 3 only: raw conv1 result, when a caller needs it
  int B = 1, C = 32, T = 0, Mp = 64;
  int depth = 3;               // number of fused convs: 3 = conv1..conv3, 2 = conv2, conv3
  ChainConv cv[3];
  int force_nc = 0;            // tuning: 128 / 256 columns per tile (0: chosen by the launcher)
  int wino = 1;                // minimal-filtering form where there is one (32 channels, depth 3); OU_WINO / OU_CONV_DIRECT < 5: 0
  const int* lens = nullptr;   // ragged batch: valid samples per row at this level (conv_chainw_kernel zeroes every stage behind them)
  unsigned long long* prof = nullptr;
  long long* tstamps = nullptr;  // tuning: per-wave phase cycle counts
};
// Shape check + cost estimate (in cycles) of the best variant; <0 when the shape is not supported.
double chain_cost(const ChainArgs& a, int num_cu, int* nc_out);
hipError_t launch_chain(const ChainArgs& a, int num_cu, hipStream_t st, int* variant);
// the three body convs of a deep-level ConvBlock (k5, k3, k3; batch 1) in one launch; hipErrorInvalidConfiguration = not a
// shape for it.  `bar`: 8 x 40 x 8 zero-initialised bytes that only this kernel touches, `err`: the status words
hipError_t launch_conv_block3(const ConvArgs* cv, unsigned long long* bar, unsigned* err, int num_cu, hipStream_t st,
                              int* cfg_out);

struct GruArgs {
  const float* gx = nullptr;
  const float* whh = nullptr;  // canonical [dir][3H][H] row-major
  const float* bhn = nullptr;
  float* out = nullptr;
  const float* res = nullptr;
  float res_scale = 1.f;
  unsigned long long* xchg = nullptr;  // exchange area: 2B clusters x (2H + 64) granules (see gru_granules())
  unsigned long long* xchg_base = nullptr;  // ring kernel: start and size of the WHOLE area of this GRU layer (filled by
  size_t xchg_granules = 0;                 // launch_gru; a chunked batch advances `xchg` per sub-launch)
  unsigned* epoch = nullptr;           // version 2: {tag epoch, finished-block count} of this exchange area (device)
  int version = 2;                     // 1: polling-wave kernel (memset per launch), 2: ring kernel (epoch tags)
  unsigned* err = nullptr;             // device status word
  long long* tstamps = nullptr;        // per-wave cycle breakdown (tuning only)
  int B = 1, T = 0, H = 0;
  int poll_backoff = 0;  // tuning: 0/1/2 x ~512 cycles of sleep before the first poll
  int agent_stores = 1;  // 1 (default): publish with agent-scope (sc1) stores; 0: plain stores when the cluster shares one XCD
  int dbg = 0;           // experiments (OU_GRU_DBG): bit 0 = no republish safety net, bit 1 = system-scope publishes from the start,
                         // bit 2 = fault injection: one workgroup drops its publishes of step 50
  int share = 1;       // GRU launches that may be resident on one XCD at the same time, this one included (2: another layer on a
                       // side stream; more: other lanes of the process, ou_set_lanes): a launch takes 1 / share of an XCD
  int lanes = 1;       // enhance calls in flight side by side in this process
  int xcd_rot = 0;     // cluster c of the launch is dealt to XCD (c + xcd_rot) % 8
  unsigned long long* prof = nullptr;  // measurement: slot [0] <- ticks before the pass, [16] <- ~ticks behind it (stamp kernels)
  int force_bmax = 0;  // testing: cap the utterances per launch (forces the chunked path at small batches)
  int force_upw = 0;  // tuning: 16 / 32 / 64 hidden units per workgroup (0: chosen from the batch size)
};
hipError_t launch_gru(const GruArgs& a, int num_cu, hipStream_t st);
// utterances one ring-kernel launch can carry with its whole grid resident (0: hidden size not supported)
int gru_ring_batch_cap(int H, int num_cu, int share, int force_upw, int B, int lanes);
// 8-byte exchange granules needed for a batch of B sequences with hidden size H (both kernel generations)
inline size_t gru_granules(int B, int H) { return (size_t)2 * B * (2 * H + 64); }

// Alias-free Snake + Conv1d(C -> 1, k3)  (universe_gan.py:117-126,145-149)
// (`lens` / `lens2`: per-row lengths of a ragged batch on the T and 2T grids, or null)
hipError_t launch_decoupling(const float* aux, const float* alpha_exp, const float* up_k, const float* down_k,
                             const float* w, const float* bias, float* tmp_up, float* out, int B, int C, int T,
                             hipStream_t st, const int* lens = nullptr, const int* lens2 = nullptr);

}  // namespace ou
