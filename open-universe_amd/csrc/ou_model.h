// Host-side model description: walks the UNIVERSE / UNIVERSE++ architecture
// (reference: networks/universe/{score,condition,blocks,sigma_block}.py) and assigns every layer a slot
// in one packed fp32 weight blob laid out for the gfx950 kernels.  Pure host C++ (no HIP).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/ouniverse.h"

namespace ou {

// A dense conv lowered to the generic MFMA implicit-GEMM kernel:
//   y[co*up + p? -> see below][q] = bias[co] + sum_{ci,k} W[m][ci][k] * act(x[ci][q*stride + k - pad])
//   rows m in [0, M), M = Cout*up, m = co*up + p, output sample t = q*up + p  (up > 1: transposed conv)
struct ConvL {
  std::string name;       // reference state-dict prefix (e.g. "_edm_model.encoder.ds_modules.0.conv1")
  int kind = 0;           // see enum below (how the packer derives W from the checkpoint tensors)
  int Cin = 0, Cout = 0;  // logical channels (Cin after space-to-depth for ST convs)
  int KW = 1, stride = 1, pad = 0, up = 1;
  int M = 0, Mp = 0;      // GEMM rows, padded to a multiple of 64
  int CK = 2;             // input-channel chunk (power of two, divides Cin)
  int rate = 1;           // original rate-change factor (down/up/st convs)
  int act = 0;            // PReLU prologue
  size_t wd_off = 0;      // stride-1 k3 / k5 layers: second copy of the weights with the taps innermost, [Cin][Mp][KWP]
  int KWP = 0;            // ... 4 (k3) / 8 (k5); 0 = no such copy
  size_t wu_off = 0;      // ... and a third copy in the Winograd / Cook-Toom domain, U = G w: F(2, 3) -> 4 floats, F(2, 5) -> 6 (of
                          // 8) floats per (row, channel), same [Cin][Mp][KWP] slots (conv_direct2w_kernel); valid when KWP != 0
  size_t ws_off = 0;      // ... and a fourth copy for the bf16 matrix pipe: every weight as three bf16 pieces, laid out as MFMA A
                          // fragments [Cin/16][KW][Mp/32][3][64 lanes][8 bf16] (conv_split_kernel, ou_split_pack.h); valid when ws_on
  int ws_on = 0;
  size_t wsw_off = 0;     // ... and (round 6) the same for the Winograd-domain weights U = G w: [Cin/16][KW + 1][Mp/32][3][64][8 bf16]
                          // (conv_splitw_kernel: minimal filtering ON the bf16 pipe -- KW + 1 instead of 2 KW piece-product sets per
                          // pair of outputs); valid when wsw_on
  int wsw_on = 0;
  size_t w_off = 0;       // float offsets into the blob
  size_t b_off = 0;       // bias[Cout]
  size_t a_off = 0;       // prelu slope (1 float) when act
  int fir_mode = 0;       // anti-alias FIR (blocks.py:213-221): 0 none; 1 / 2 = separate FIR pass before (down) / after
                          // (up) the conv; folded into the conv weights by the packer (OU_FIR_FOLD, see make_conv):
                          // 3 = down (KW = 3r, stride r, pad r), 4 = up (3-tap phase GEMMs)
  int fir_len = 0;        // 2*rate + 1 taps
  size_t fir_off = 0;     // taps
  size_t fbias_off = 0;   // fir_mode 2: the manual bias added after the FIR (the conv itself has none)
  size_t w_floats() const { return (size_t)Cin * KW * Mp; }
};
enum ConvKind {
  CK_CONV = 0,      // Conv1d stride 1 'same' (conv1/2/3, mel conv, 1x1 signal-cond proj)
  CK_DOWN = 1,      // strided Conv1d k=s=r (+ folded binomial FIR when anti-aliased)
  CK_UP = 2,        // ConvTranspose1d k=s=r (+ folded binomial FIR)
  CK_ST = 3,        // st_conv: PReLU -> Conv1d k=s=R, lowered to space-to-depth + 1x1
  CK_GRU_PROJ = 4,  // GRU input projection, both directions stacked (rows: dir*3H + gate*H + j)
};

struct BlockL {  // ConvBlock (blocks.py:230-412)
  std::string name;
  int C = 0;     // n_channels of conv1..3
  int dir = 0;   // 0 none, 1 down, 2 up
  int rate = 1;
  ConvL rc, c1, c2, c3;
};

struct GruL {  // one bidirectional GRU layer
  std::string name;  // e.g. "_edm_model.encoder.gru"
  int layer = 0;
  int I = 0, H = 0;
  ConvL proj;          // (2*3H, I) input projection incl. b_ih (+ b_hr, b_hz folded)
  size_t whh_off = 0;  // packed recurrent weights, see pack layout in ou_model.cpp
  size_t bhn_off = 0;  // [2][H]
};

struct SmallConvL {  // VALU conv (Cin==1 input conv, Cout==1 output conv)
  std::string name;
  int Cin = 0, Cout = 0, KW = 3;
  size_t w_off = 0;  // in-conv: [Cout][KW]; out-conv: [Cin][KW]
  size_t b_off = 0;
  size_t a_off = 0;  // out-conv: two PReLU slopes (score.prelu, output_conv.prelu)
};

struct FilmL {  // the 2*n_blocks FiLM projections (score.py:58-79,159-188) concatenated
  int D = 512;
  int rows = 0;
  std::vector<int> enc_off, dec_off;  // row offset of each block's (gamma|beta) pair
  size_t w_off = 0;                   // [rows][D]
  size_t b_off = 0;                   // [rows]
};

struct SigmaL {  // sigma_block.py
  int simple = 1;
  int D = 512, n_rff = 32;
  size_t p_off = 0;  // simple: {weight, bias}; rff: freq[n_rff], then 3 x {alpha, W[out][in], b[out]}
};

struct MelL {  // condition.py:68-108
  int n_fft = 640, hop = 160, n_freq = 321, n_mels = 80, pad_left = 240;
  size_t win_off = 0;  // window[n_fft]
  size_t fb_off = 0;   // fb[n_freq][n_mels]
  size_t tw_off = 0;   // twiddle cos[n_fft], sin[n_fft]
};

struct DecouplingL {  // universe_gan.py:117-126 + bigvgan/snake.py, alias_free_act.py
  int present = 0, act = 0, C = 0;
  size_t alpha_off = 0;  // exp(alpha)[C] (snake, log-scale)
  size_t up_off = 0;     // (2,15)
  size_t down_off = 0;   // (28)
  size_t prelu_off = 0;
  SmallConvL conv;       // C -> 1, k3
};

struct Model {
  ou_config cfg;
  int tot_ds = 160, n_levels = 5, n_blocks = 5, C0 = 32, OC = 512;
  std::string score_prefix;
  // score network
  SigmaL sigma;
  SmallConvL s_in, s_out;
  std::vector<BlockL> s_enc, s_dec;
  std::vector<ConvL> s_sig;  // signal_cond_proj 1x1
  FilmL film;
  GruL s_gru;
  // conditioner
  MelL mel;
  ConvL c_melconv;
  BlockL c_melblock;
  SmallConvL c_in;
  std::vector<BlockL> c_enc, c_dec;
  std::vector<ConvL> c_st;
  BlockL c_cb1, c_cb2, c_decin;
  GruL c_gru0, c_gru1;
  DecouplingL dec;
  size_t total_floats = 0;
  std::string json;  // plan description
};

// Builds the layer list and blob layout.  Returns "" or an error message.
std::string build_model(const ou_config& cfg, Model& m);

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};
// Fills `blob` (m.total_floats) from checkpoint tensors.  Returns "" or an error; `code` gets OU_E*.
std::string pack_weights(const Model& m, const std::map<std::string, HostTensor>& sd, std::vector<float>& blob,
                         int& code);

}  // namespace ou
