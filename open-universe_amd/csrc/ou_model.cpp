// Architecture walk + weight packing (host).  See ou_model.h.
#include "ou_model.h"
#include "ou_split_pack.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace ou {

static thread_local int g_no_split = 0;  // ou_config.no_split_copy of the model being built (build_model sets it)


namespace {

struct Alloc {
  size_t n = 0;
  size_t take(size_t floats) {
    size_t off = n;
    n += (floats + 63) & ~size_t(63);  // 256-byte aligned slots
    return off;
  }
};

int choose_ck(int Cin, int KW, int stride) {
  for (int c = 32; c >= 2; c >>= 1) {
    if (Cin % c) continue;
    if (c * KW > 96) continue;
    if ((long)c * (127L * stride + KW) > 4096) continue;
    return c;
  }
  return 0;
}

std::string finish_conv(ConvL& L, Alloc& a) {
  L.M = L.Cout * L.up;
  L.Mp = (L.M + 63) / 64 * 64;
  L.CK = choose_ck(L.Cin, L.KW, L.stride);
  if (!L.CK) return "layer " + L.name + ": unsupported channel count " + std::to_string(L.Cin);
  L.w_off = a.take(L.w_floats());
  L.b_off = a.take(L.Cout);
  L.a_off = a.take(1);
  if (L.fir_mode == 1 || L.fir_mode == 2) { L.fir_off = a.take(L.fir_len); L.fbias_off = a.take(L.Cout); }
  // layers the wide-load direct kernel can take (deep levels at small batch): taps-innermost copy of the weights
  // (Cin % 64: what conv_direct2_kernel's 8-way split-K needs; Cin % 16: the no-split-K conv_direct3_kernel)
  if (L.stride == 1 && L.up == 1 && (L.KW == 3 || L.KW == 5) && L.Cin % 16 == 0 && L.pad == (L.KW - 1) / 2) {
    L.KWP = L.KW == 3 ? 4 : 8;
    L.wd_off = a.take((size_t)L.Cin * L.Mp * L.KWP);
    L.wu_off = a.take((size_t)L.Cin * L.Mp * L.KWP);
    // the bf16-split copy (conv_split_kernel: 64-row tiles -- the 48-channel level of UNIVERSE++ 24 kHz stays on the fp32 kernels)
    if (L.M % 64 == 0 && !g_no_split) { L.ws_on = 1; L.ws_off = a.take(split_floats(L.Cin, L.KW, L.Mp)); }
#ifdef OU_EXPERIMENTS
    // ... and of the Winograd-domain weights, for the layers conv_splitw_kernel takes (k3, rows a multiple of 128): the
    // experiments build only -- the kernel measured slower (ou_conv_split.hip), the default blob does not carry its copy
    if (L.M % 128 == 0 && L.KW == 3) { L.wsw_on = 1; L.wsw_off = a.take(split_floats(L.Cin, L.KW + 1, L.Mp)); }
#endif
  }
  return "";
}

// ou_config.fir_fold (bit 0: down path, bit 1: up path; until ABI 4 the environment variable OU_FIR_FOLD): fold the anti-alias FIRs into the rate-change conv weights.  One
// launch and one bandwidth pass less per rate change, for 3x that conv's FLOPs.  Measured on MI355X (PP16, B = 1) the
// separate 6.5 us FIR pass is cheaper on every level (down: 17-24 vs 21-42 us, up: 17-22 vs 24-38 us), so the default
// is 0; the plan and the packed blob depend on it, so it has to be the same when packing and when creating the model.
static thread_local int g_fir_fold = 0;  // ou_config.fir_fold of the model being built (build_model sets it)
static int fir_fold() { return g_fir_fold; }

std::string make_conv(ConvL& L, Alloc& a, const std::string& name, int kind, int cin, int cout, int k, int rate,
                      bool aa, bool act) {
  L = ConvL();
  L.name = name;
  L.kind = kind;
  L.act = act ? 1 : 0;
  L.rate = rate;
  switch (kind) {
    case CK_CONV:
      L.Cin = cin; L.Cout = cout; L.KW = k; L.stride = 1; L.pad = (k - 1) / 2; L.up = 1;
      if (k % 2 == 0) return "even 'same' kernel not supported: " + name;
      break;
    case CK_DOWN:
      L.Cin = cin; L.Cout = cout; L.stride = rate; L.up = 1; L.KW = rate; L.pad = 0;
      // anti-aliased (blocks.py:211-215): FIR_{2r+1} as its own bandwidth pass before the conv, or (fold) ONE strided
      // conv with 3r taps and left pad r
      if (aa) {
        L.fir_len = 2 * rate + 1;
        if (fir_fold() & 1) { L.fir_mode = 3; L.KW = 3 * rate; L.pad = rate; }
        else L.fir_mode = 1;
      }
      break;
    case CK_UP:
      L.Cin = cin; L.Cout = cout; L.stride = 1; L.up = rate; L.KW = 1; L.pad = 0;
      // anti-aliased (blocks.py:217-221): FIR_{2r+1} as a bandwidth pass after the phase GEMMs, or (fold) r phase GEMMs
      // with 3 taps each -- FIR(convT_{k=s=r}(.)) reaches one input frame to either side of an output frame
      if (aa) {
        L.fir_len = 2 * rate + 1;
        if (fir_fold() & 2) { L.fir_mode = 4; L.KW = 3; L.pad = 1; }
        else L.fir_mode = 2;
      }
      break;
    case CK_ST:
      L.Cin = cin * rate; L.Cout = cout; L.KW = 1; L.stride = 1; L.pad = 0; L.up = 1;
      break;
    case CK_GRU_PROJ:
      L.Cin = cin; L.Cout = cout; L.KW = 1; L.stride = 1; L.pad = 0; L.up = 1;
      break;
  }
  return finish_conv(L, a);
}

std::string make_block(BlockL& B, Alloc& a, const std::string& name, int C, int dir, int rate, bool aa) {
  B = BlockL();
  B.name = name; B.C = C; B.dir = dir; B.rate = rate;
  std::string e;
  if (dir == 1) e = make_conv(B.rc, a, name + ".rate_change_conv", CK_DOWN, C, 2 * C, rate, rate, aa, true);
  if (dir == 2) e = make_conv(B.rc, a, name + ".rate_change_conv", CK_UP, 2 * C, C, rate, rate, aa, true);
  if (!e.empty()) return e;
  if (!(e = make_conv(B.c1, a, name + ".conv1", CK_CONV, C, C, 5, 1, false, true)).empty()) return e;
  if (!(e = make_conv(B.c2, a, name + ".conv2", CK_CONV, C, C, 3, 1, false, true)).empty()) return e;
  if (!(e = make_conv(B.c3, a, name + ".conv3", CK_CONV, C, C, 3, 1, false, true)).empty()) return e;
  return "";
}

std::string make_gru(GruL& G, Alloc& a, const std::string& name, int layer, int I, int H) {
  G = GruL();
  G.name = name; G.layer = layer; G.I = I; G.H = H;
  if (H % 64) return "GRU hidden size must be a multiple of 64 (got " + std::to_string(H) + ")";
  std::string e = make_conv(G.proj, a, name + "#l" + std::to_string(layer), CK_GRU_PROJ, I, 6 * H, 1, 1, false, false);
  if (!e.empty()) return e;
  G.whh_off = a.take((size_t)6 * H * H);
  G.bhn_off = a.take((size_t)2 * H);
  return "";
}

void json_conv(std::ostringstream& os, const ConvL& L, bool& first) {
  if (!first) os << ",\n";
  first = false;
  os << "  {\"name\":\"" << L.name << "\",\"kind\":" << L.kind << ",\"Cin\":" << L.Cin << ",\"Cout\":" << L.Cout
     << ",\"KW\":" << L.KW << ",\"stride\":" << L.stride << ",\"pad\":" << L.pad << ",\"up\":" << L.up
     << ",\"M\":" << L.M << ",\"Mp\":" << L.Mp << ",\"CK\":" << L.CK << ",\"rate\":" << L.rate << ",\"act\":" << L.act
     << ",\"w_off\":" << L.w_off << ",\"b_off\":" << L.b_off << ",\"a_off\":" << L.a_off
     << ",\"fir_mode\":" << L.fir_mode << ",\"fir_len\":" << L.fir_len << ",\"fir_off\":" << L.fir_off
     << ",\"fbias_off\":" << L.fbias_off << ",\"wd_off\":" << L.wd_off << ",\"wu_off\":" << L.wu_off << ",\"KWP\":" << L.KWP << ",\"ws_on\":" << L.ws_on << ",\"ws_off\":" << L.ws_off << ",\"wsw_on\":" << L.wsw_on << ",\"wsw_off\":" << L.wsw_off << "}";
}
void json_block(std::ostringstream& os, const BlockL& B, bool& first) {
  if (B.dir) json_conv(os, B.rc, first);
  json_conv(os, B.c1, first);
  json_conv(os, B.c2, first);
  json_conv(os, B.c3, first);
}

}  // namespace

std::string build_model(const ou_config& cfg, Model& m) {
  m = Model();
  m.cfg = cfg;
  if (cfg.abi_version != OU_ABI_VERSION) return "ou_config.abi_version mismatch";
  if (cfg.fir_fold < 0 || cfg.fir_fold > 3) return "ou_config.fir_fold must be 0 .. 3";
  g_fir_fold = cfg.fir_fold;
  g_no_split = cfg.no_split_copy != 0;
  const ou_net_config& s = cfg.score;
  const ou_net_config& c = cfg.cond;
  if (s.n_rates < 1 || s.n_rates > OU_MAX_RATES) return "bad n_rates";
  if (c.n_rates != s.n_rates || c.n_channels != s.n_channels) return "score/cond topology mismatch";
  for (int i = 0; i < s.n_rates; i++)
    if (c.rate_factors[i] != s.rate_factors[i] || s.rate_factors[i] < 2 || s.rate_factors[i] > 8)
      return "rate factors must match between score/cond and lie in [2,8]";
  if (s.fb_kernel_size % 2 == 0 || s.fb_kernel_size > 7) return "fb_kernel_size must be odd and <= 7";
  if (c.fb_kernel_size != s.fb_kernel_size) return "fb_kernel_size mismatch";
  if (s.n_channels % 2 || s.n_channels > 64) return "n_channels must be even and <= 64";
  if (s.noise_cond_dim % 64 || s.noise_cond_dim > 1024) return "noise_cond_dim must be a multiple of 64";
  if (!s.time_embedding_simple && (s.n_rff < 1 || s.n_rff > 64)) return "n_rff out of range";
  if (s.extra_conv_block != c.extra_conv_block) return "extra_conv_block mismatch";

  const int n = s.n_rates, C0 = s.n_channels;
  int tot = 1;
  for (int i = 0; i < n; i++) tot *= s.rate_factors[i];
  m.tot_ds = tot;
  m.n_levels = n + 1;
  m.n_blocks = n + (s.extra_conv_block ? 1 : 0);
  m.C0 = C0;
  m.OC = C0 << n;
  m.score_prefix = cfg.has_edm ? "_edm_model" : "score_model";
  const std::string sp = m.score_prefix, cp = "condition_model";
  const bool aa = s.use_antialiasing != 0;
  const int OC = m.OC, D = s.noise_cond_dim;
  Alloc a;
  std::string e;

  // ---- score network ------------------------------------------------------------------------------
  m.sigma.simple = s.time_embedding_simple;
  m.sigma.D = D;
  m.sigma.n_rff = s.n_rff;
  if (m.sigma.simple) {
    m.sigma.p_off = a.take(2);
  } else {
    size_t nfl = s.n_rff;
    int dims[4] = {2 * s.n_rff, 4 * s.n_rff, 8 * s.n_rff, D};
    for (int i = 0; i < 3; i++) nfl += 1 + (size_t)dims[i + 1] * dims[i] + dims[i + 1];
    m.sigma.p_off = a.take(nfl);
  }
  m.s_in = SmallConvL();
  m.s_in.name = sp + ".input_conv"; m.s_in.Cin = 1; m.s_in.Cout = C0; m.s_in.KW = s.fb_kernel_size;
  m.s_in.w_off = a.take((size_t)C0 * m.s_in.KW); m.s_in.b_off = a.take(C0);

  m.s_enc.resize(m.n_blocks);
  m.film.D = D;
  int frow = 0;
  for (int i = 0; i < m.n_blocks; i++) {
    int C = C0 << (i < n ? i : n);
    if (i < n) e = make_block(m.s_enc[i], a, sp + ".encoder.ds_modules." + std::to_string(i), C, 1, s.rate_factors[i], aa);
    else e = make_block(m.s_enc[i], a, sp + ".encoder.ds_modules." + std::to_string(i), C, 0, 1, false);
    if (!e.empty()) return e;
    m.film.enc_off.push_back(frow);
    frow += 2 * C;
  }
  if (!(e = make_gru(m.s_gru, a, sp + ".encoder.gru", 0, OC, OC / 2)).empty()) return e;
  m.s_dec.resize(m.n_blocks);
  m.s_sig.resize(m.n_blocks);
  for (int j = 0; j < m.n_blocks; j++) {
    int C, dir, rate;
    if (s.extra_conv_block) {
      if (j == 0) { C = OC; dir = 0; rate = 1; }
      else { C = C0 << (n - j); dir = 2; rate = s.rate_factors[n - j]; }
    } else { C = C0 << (n - j - 1); dir = 2; rate = s.rate_factors[n - j - 1]; }
    if (!(e = make_block(m.s_dec[j], a, sp + ".decoder.up_modules." + std::to_string(j), C, dir, rate, aa)).empty()) return e;
    if (!(e = make_conv(m.s_sig[j], a, sp + ".decoder.signal_cond_proj." + std::to_string(j), CK_CONV, C, C, 1, 1, false, false)).empty()) return e;
    m.film.dec_off.push_back(frow);
    frow += 2 * C;
  }
  m.film.rows = frow;
  m.film.w_off = a.take((size_t)frow * D);
  m.film.b_off = a.take(frow);
  m.s_out = SmallConvL();
  m.s_out.name = sp + ".output_conv"; m.s_out.Cin = C0; m.s_out.Cout = 1; m.s_out.KW = s.fb_kernel_size;
  m.s_out.w_off = a.take((size_t)C0 * m.s_out.KW); m.s_out.b_off = a.take(1); m.s_out.a_off = a.take(2);

  // ---- conditioner --------------------------------------------------------------------------------
  m.mel.hop = tot;
  m.mel.n_fft = c.n_mel_oversample * tot;
  m.mel.n_freq = m.mel.n_fft / 2 + 1;
  m.mel.n_mels = c.n_mels;
  m.mel.pad_left = (m.mel.n_fft - tot) / 2;
  if (m.mel.n_fft > 2048 || c.n_mels > 256 || c.n_mels % 2) return "mel front-end size out of range";
  m.mel.win_off = a.take(m.mel.n_fft);
  m.mel.fb_off = a.take((size_t)m.mel.n_freq * c.n_mels);
  m.mel.tw_off = a.take((size_t)2 * m.mel.n_fft);
  if (!(e = make_conv(m.c_melconv, a, cp + ".input_mel.conv", CK_CONV, c.n_mels, OC, 3, 1, false, false)).empty()) return e;
  if (!(e = make_block(m.c_melblock, a, cp + ".input_mel.conv_block", OC, 0, 1, false)).empty()) return e;
  m.c_in = SmallConvL();
  m.c_in.name = cp + ".input_conv"; m.c_in.Cin = 1; m.c_in.Cout = C0; m.c_in.KW = c.fb_kernel_size;
  m.c_in.w_off = a.take((size_t)C0 * m.c_in.KW); m.c_in.b_off = a.take(C0);
  m.c_enc.resize(m.n_blocks);
  for (int i = 0; i < m.n_blocks; i++) {
    int C = C0 << (i < n ? i : n);
    if (i < n) e = make_block(m.c_enc[i], a, cp + ".encoder.ds_modules." + std::to_string(i), C, 1, c.rate_factors[i], false);
    else e = make_block(m.c_enc[i], a, cp + ".encoder.ds_modules." + std::to_string(i), C, 0, 1, false);
    if (!e.empty()) return e;
  }
  m.c_st.resize(n - 1);
  for (int i = 0; i < n - 1; i++) {
    int R = 1;
    for (int k = i; k < n; k++) R *= c.rate_factors[k];
    if (!(e = make_conv(m.c_st[i], a, cp + ".encoder.st_convs." + std::to_string(i), CK_ST, C0 << i, OC, R, R, false, true)).empty()) return e;
  }
  if (!(e = make_gru(m.c_gru0, a, cp + ".encoder.gru", 0, OC, OC / 2)).empty()) return e;
  if (!(e = make_gru(m.c_gru1, a, cp + ".encoder.gru", 1, OC, OC / 2)).empty()) return e;
  if (!(e = make_block(m.c_cb1, a, cp + ".encoder.conv_block1", OC, 0, 1, false)).empty()) return e;
  if (!(e = make_block(m.c_cb2, a, cp + ".encoder.conv_block2", OC, 0, 1, false)).empty()) return e;
  if (!(e = make_block(m.c_decin, a, cp + ".decoder.input_conv_block", OC, 0, 1, false)).empty()) return e;
  m.c_dec.resize(m.n_blocks);
  for (int j = 0; j < m.n_blocks; j++) {
    int C, dir, rate;
    if (c.extra_conv_block) {
      if (j == 0) { C = OC; dir = 0; rate = 1; }
      else { C = C0 << (n - j); dir = 2; rate = c.rate_factors[n - j]; }
    } else { C = C0 << (n - j - 1); dir = 2; rate = c.rate_factors[n - j - 1]; }
    if (!(e = make_block(m.c_dec[j], a, cp + ".decoder.up_modules." + std::to_string(j), C, dir, rate, c.use_antialiasing != 0)).empty()) return e;
  }
  // ---- signal decoupling layer (UNIVERSE++ aux_to_wav) ----------------------------------------------
  m.dec = DecouplingL();
  if (cfg.kind == OU_KIND_UNIVERSE_GAN && cfg.use_signal_decoupling) {
    m.dec.present = 1; m.dec.act = cfg.signal_decoupling_act; m.dec.C = C0;
    m.dec.alpha_off = a.take(C0); m.dec.up_off = a.take(30); m.dec.down_off = a.take(28); m.dec.prelu_off = a.take(1);
    m.dec.conv.name = "signal_decoupling_layer.conv"; m.dec.conv.Cin = C0; m.dec.conv.Cout = 1; m.dec.conv.KW = 3;
    m.dec.conv.w_off = a.take((size_t)C0 * 3); m.dec.conv.b_off = a.take(1);
  }
  m.total_floats = a.n;

  std::ostringstream os;
  os << "{\"total_floats\":" << m.total_floats << ",\"tot_ds\":" << m.tot_ds << ",\"film_rows\":" << m.film.rows
     << ",\"film_w_off\":" << m.film.w_off << ",\"film_b_off\":" << m.film.b_off << ",\n\"convs\":[\n";
  bool first = true;
  for (auto& b : m.s_enc) json_block(os, b, first);
  json_conv(os, m.s_gru.proj, first);
  for (auto& b : m.s_dec) json_block(os, b, first);
  for (auto& l : m.s_sig) json_conv(os, l, first);
  json_conv(os, m.c_melconv, first);
  json_block(os, m.c_melblock, first);
  for (auto& b : m.c_enc) json_block(os, b, first);
  for (auto& l : m.c_st) json_conv(os, l, first);
  json_conv(os, m.c_gru0.proj, first);
  json_conv(os, m.c_gru1.proj, first);
  json_block(os, m.c_cb1, first);
  json_block(os, m.c_cb2, first);
  json_block(os, m.c_decin, first);
  for (auto& b : m.c_dec) json_block(os, b, first);
  os << "\n]}";
  m.json = os.str();
  return "";
}

// ======================================================================================================
// packing
// ======================================================================================================
namespace {

struct Packer {
  const std::map<std::string, HostTensor>& sd;
  std::vector<float>& blob;
  std::string err;
  int code = OU_OK;

  const HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape) {
    auto it = sd.find(key);
    if (it == sd.end()) {
      if (err.empty()) { err = "missing tensor: " + key; code = OU_EMISSING; }
      return nullptr;
    }
    const HostTensor& t = it->second;
    bool ok = t.shape.size() == shape.size();
    size_t i = 0;
    for (auto d : shape) { if (ok && t.shape[i] != d) ok = false; i++; }
    if (!ok) {
      if (err.empty()) {
        std::string got = "(", want = "(";
        for (auto d : t.shape) got += std::to_string(d) + ",";
        for (auto d : shape) want += std::to_string(d) + ",";
        err = "shape mismatch for " + key + ": got " + got + ") expected " + want + ")";
        code = OU_ESHAPE;
      }
      return nullptr;
    }
    return &t;
  }
  bool has(const std::string& key) const { return sd.count(key) != 0; }

  // Effective (weight-norm folded) weight as doubles, same shape as weight_v / weight.
  // blocks.py:36-42: w = g * v / ||v||, norm over all dims but 0.
  bool eff_weight(const std::string& p, std::initializer_list<int64_t> shape, std::vector<double>& w) {
    size_t total = 1;
    for (auto d : shape) total *= (size_t)d;
    int64_t d0 = *shape.begin();
    w.resize(total);
    if (has(p + ".weight_g")) {
      const HostTensor* v = get(p + ".weight_v", shape);
      auto it = sd.find(p + ".weight_g");
      if (!v) return false;
      if ((int64_t)it->second.data.size() != d0) {
        if (err.empty()) { err = "shape mismatch for " + p + ".weight_g"; code = OU_ESHAPE; }
        return false;
      }
      size_t inner = total / (size_t)d0;
      for (int64_t r = 0; r < d0; r++) {
        double nrm = 0;
        for (size_t i = 0; i < inner; i++) { double x = v->data[r * inner + i]; nrm += x * x; }
        nrm = std::sqrt(nrm);
        double sc = (double)it->second.data[r] / nrm;
        for (size_t i = 0; i < inner; i++) w[r * inner + i] = (double)v->data[r * inner + i] * sc;
      }
      return true;
    }
    const HostTensor* t = get(p + ".weight", shape);
    if (!t) return false;
    for (size_t i = 0; i < total; i++) w[i] = t->data[i];
    return true;
  }

  void put(size_t off, const std::vector<double>& v) { for (size_t i = 0; i < v.size(); i++) blob[off + i] = (float)v[i]; }
  void putf(size_t off, const float* v, size_t n) { std::memcpy(&blob[off], v, n * sizeof(float)); }

  // W_eff[m][ci][k] (double, M x Cin x KW) -> blob in the kernel layout:
  //   row = (chunk*KW + tap)*CK + l  (ci = chunk*CK + l), columns m in [0, Mp)
  void store_conv(const ConvL& L, const std::vector<double>& W) {
    const int KW = L.KW, CK = L.CK, Mp = L.Mp;
    for (int ci = 0; ci < L.Cin; ci++) {
      int chunk = ci / CK, l = ci % CK;
      for (int k = 0; k < KW; k++) {
        size_t row = ((size_t)chunk * KW + k) * CK + l;
        float* dst = &blob[L.w_off + row * Mp];
        for (int mm = 0; mm < L.M; mm++) dst[mm] = (float)W[((size_t)mm * L.Cin + ci) * KW + k];
      }
    }
    if (L.KWP)  // (the blob starts zeroed: taps KW .. KWP - 1 and rows M .. Mp - 1 stay 0)
      for (int ci = 0; ci < L.Cin; ci++)
        for (int mm = 0; mm < L.M; mm++)
          for (int k = 0; k < KW; k++)
            blob[L.wd_off + ((size_t)ci * Mp + mm) * L.KWP + k] = (float)W[((size_t)mm * L.Cin + ci) * KW + k];
    if (L.ws_on) {  // three bf16 pieces of the fp32 weight the other kernels use (ou_split_pack.h)
      std::vector<float> Wf(W.size());
      for (size_t i = 0; i < W.size(); i++) Wf[i] = (float)W[i];
      pack_split(Wf.data(), L.M, L.Mp, L.Cin, L.KW, reinterpret_cast<uint16_t*>(&blob[L.ws_off]));
    }
    if (L.KWP) {
      // Winograd / Cook-Toom minimal filtering F(2, KW): two outputs from KW + 1 products instead of 2 KW.  U = G w in double,
      // rounded once.  F(2, 3): points 0, 1, -1, inf (Lavin & Gray's matrices).  F(2, 5): points 0, 1, -1, 1/2, -2, inf -- the
      // mixed pair 1/2, -2 loses 2 dB less than +-2 in fp32 (measured against a double evaluation: 123-125 dB per layer, plain
      // fp32 summation 131-133 dB); rows scaled so that the kernel's input transform B^T has small integer entries.
      static const double G3[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
      static const double G5[6][5] = {{1. / 2, 0, 0, 0, 0},
                                      {1. / 6, 1. / 6, 1. / 6, 1. / 6, 1. / 6},
                                      {1. / 6, -1. / 6, 1. / 6, -1. / 6, 1. / 6},
                                      {16. / 15, 8. / 15, 4. / 15, 2. / 15, 1. / 15},
                                      {1. / 30, -1. / 15, 2. / 15, -4. / 15, 8. / 15},
                                      {0, 0, 0, 0, 1. / 2}};
      const int NU = KW + 1;
      std::vector<float> Uf(L.wsw_on ? (size_t)L.M * L.Cin * NU : 0);
      for (int ci = 0; ci < L.Cin; ci++)
        for (int mm = 0; mm < L.M; mm++) {
          const double* w = &W[((size_t)mm * L.Cin + ci) * KW];
          for (int x = 0; x < NU; x++) {
            double u = 0;
            for (int k = 0; k < KW; k++) u += (KW == 3 ? G3[x][k] : G5[x][k]) * w[k];
            blob[L.wu_off + ((size_t)ci * Mp + mm) * L.KWP + x] = (float)u;
            if (L.wsw_on) Uf[((size_t)mm * L.Cin + ci) * NU + x] = (float)u;  // (the same fp32 U the fp32 kernels multiply with)
          }
        }
      // three bf16 pieces of every U value, as A fragments with the KW + 1 transformed taps in the place of the taps
      if (L.wsw_on) pack_split(Uf.data(), L.M, L.Mp, L.Cin, NU, reinterpret_cast<uint16_t*>(&blob[L.wsw_off]));
    }
  }

  void pack_conv(const ConvL& L) {
    std::vector<double> w, W;
    std::vector<double> bias(L.Cout, 0.0);
    const std::string& p = L.name;
    if (L.act) {
      const HostTensor* a = get(p + ".prelu.weight", {1});
      if (!a) return;
      blob[L.a_off] = a->data[0];
    }
    switch (L.kind) {
      case CK_CONV: {
        // PReLU_Conv wraps the conv as "<p>.conv"; bare convs (mel conv, signal_cond_proj) are "<p>" itself
        std::string cpfx = L.act ? p + ".conv" : p;
        if (!eff_weight(cpfx, {L.Cout, L.Cin, L.KW}, W)) return;
        const HostTensor* b = get(cpfx + ".bias", {L.Cout});
        if (!b) return;
        for (int i = 0; i < L.Cout; i++) bias[i] = b->data[i];
        break;
      }
      case CK_DOWN: {
        // blocks.py:213-217: y = conv_{k=s=r}(FIR(prelu(x))) + bias.  fir_mode 1: the FIR (+PReLU) runs as its own
        // bandwidth-bound pass, the conv keeps its native k = stride = r.  fir_mode 3: the binomial FIR (2r + 1 taps,
        // 'same' zero padding) is folded into the conv: W2[co][ci][m] = sum_{k + j = m} W[co][ci][k] f[j], m < 3r, the
        // conv then reads x[q r + m - r] -- identical arithmetic up to the order of the fp32 sums.
        const int r = L.rate;
        bool aa = has(p + ".low_pass_filter.weights");
        if ((L.fir_mode == 1 || L.fir_mode == 3) != aa) { if (err.empty()) { err = "anti-aliasing config/checkpoint mismatch at " + p; code = OU_ESHAPE; } return; }
        if (!eff_weight(p + ".conv", {L.Cout, L.Cin, r}, aa ? w : W)) return;
        const HostTensor* b = get(aa ? p + ".bias" : p + ".conv.bias", {L.Cout});
        if (!b) return;
        for (int i = 0; i < L.Cout; i++) bias[i] = b->data[i];
        if (aa) {
          const HostTensor* f = get(p + ".low_pass_filter.weights", {2 * r + 1});
          if (!f) return;
          if (L.fir_mode == 1) {
            putf(L.fir_off, f->data.data(), 2 * r + 1);
            W = w;
          } else {
            W.assign((size_t)L.Cout * L.Cin * 3 * r, 0.0);
            for (size_t oc = 0; oc < (size_t)L.Cout * L.Cin; oc++)
              for (int k = 0; k < r; k++)
                for (int j = 0; j <= 2 * r; j++) W[oc * 3 * r + k + j] += w[oc * r + k] * (double)f->data[j];
          }
        }
        break;
      }
      case CK_UP: {
        // blocks.py:217-225: y = FIR(convT_{k=s=r}(prelu(x))) + bias.  ConvTranspose1d weight is (in, out, k) with
        // weight-norm over dim 0 = in-channels; lowered to r phase-GEMMs: row m = co*r + ph, y[q*r + ph].
        // fir_mode 2: FIR + manual bias as a pass after the conv.  fir_mode 4, FIR folded in: output sample q r + ph sees u[q r + ph + j - r], j <= 2r, i.e. input
        // frames q - 1, q, q + 1:  W3[co r + ph][ci][dq + 1] = sum_{j : floor((ph + j - r) / r) = dq} f[j] W[ci][co][(ph + j - r) mod r]
        const int r = L.rate;
        if (!eff_weight(p + ".conv", {L.Cin, L.Cout, r}, w)) return;
        bool aa = has(p + ".low_pass_filter.weights");
        if ((L.fir_mode == 2 || L.fir_mode == 4) != aa) { if (err.empty()) { err = "anti-aliasing config/checkpoint mismatch at " + p; code = OU_ESHAPE; } return; }
        if (aa && L.fir_mode == 2) {
          const HostTensor* f = get(p + ".low_pass_filter.weights", {2 * r + 1});
          const HostTensor* b = get(p + ".bias", {L.Cout});
          if (!f || !b) return;
          W.assign((size_t)L.M * L.Cin, 0.0);
          for (int co = 0; co < L.Cout; co++)
            for (int ph = 0; ph < r; ph++)
              for (int ci = 0; ci < L.Cin; ci++) W[(size_t)(co * r + ph) * L.Cin + ci] = w[((size_t)ci * L.Cout + co) * r + ph];
          putf(L.fir_off, f->data.data(), 2 * r + 1);
          putf(L.fbias_off, b->data.data(), L.Cout);  // added after the FIR; the conv's own bias stays 0
        } else if (aa) {
          const HostTensor* f = get(p + ".low_pass_filter.weights", {2 * r + 1});
          const HostTensor* b = get(p + ".bias", {L.Cout});
          if (!f || !b) return;
          W.assign((size_t)L.M * L.Cin * 3, 0.0);
          for (int co = 0; co < L.Cout; co++)
            for (int ph = 0; ph < r; ph++)
              for (int j = 0; j <= 2 * r; j++) {
                const int sft = ph + j - r;                       // position relative to the start of frame q
                const int dq = sft < 0 ? -1 : (sft >= r ? 1 : 0);
                const int pp = sft - dq * r;
                for (int ci = 0; ci < L.Cin; ci++)
                  W[((size_t)(co * r + ph) * L.Cin + ci) * 3 + dq + 1] += (double)f->data[j] * w[((size_t)ci * L.Cout + co) * r + pp];
              }
          for (int i = 0; i < L.Cout; i++) bias[i] = b->data[i];  // the manual bias added after the FIR (the conv has none)
        } else {
          W.assign((size_t)L.M * L.Cin, 0.0);
          for (int co = 0; co < L.Cout; co++)
            for (int ph = 0; ph < r; ph++)
              for (int ci = 0; ci < L.Cin; ci++) W[(size_t)(co * r + ph) * L.Cin + ci] = w[((size_t)ci * L.Cout + co) * r + ph];
          const HostTensor* b = get(p + ".conv.bias", {L.Cout});
          if (!b) return;
          for (int i = 0; i < L.Cout; i++) bias[i] = b->data[i];
        }
        break;
      }
      case CK_ST: {
        // condition.py:53-59: PReLU -> Conv1d(C, OC, k=s=R); lowered to s2d + 1x1 with kk = ci*R + k
        const int R = L.rate, C = L.Cin / R;
        if (!eff_weight(p + ".conv", {L.Cout, C, R}, W)) return;  // (OC, C, R) == (OC, C*R, 1) flattened
        const HostTensor* b = get(p + ".conv.bias", {L.Cout});
        if (!b) return;
        for (int i = 0; i < L.Cout; i++) bias[i] = b->data[i];
        break;
      }
      case CK_GRU_PROJ: {
        // name = "<gru prefix>#l<layer>"
        size_t hash = p.find('#');
        std::string g = p.substr(0, hash), lay = p.substr(hash + 2);
        const int H = L.Cout / 6, I = L.Cin;
        W.assign((size_t)L.Cout * I, 0.0);
        for (int d = 0; d < 2; d++) {
          std::string sfx = "_l" + lay + (d ? "_reverse" : "");
          const HostTensor* wih = get(g + ".weight_ih" + sfx, {3 * H, I});
          const HostTensor* bih = get(g + ".bias_ih" + sfx, {3 * H});
          const HostTensor* bhh = get(g + ".bias_hh" + sfx, {3 * H});
          if (!wih || !bih || !bhh) return;
          for (int rr = 0; rr < 3 * H; rr++) {
            for (int i = 0; i < I; i++) W[((size_t)d * 3 * H + rr) * I + i] = wih->data[(size_t)rr * I + i];
            bias[d * 3 * H + rr] = (double)bih->data[rr] + (rr < 2 * H ? (double)bhh->data[rr] : 0.0);
          }
        }
        break;
      }
    }
    store_conv(L, W);
    put(L.b_off, bias);
  }

  void pack_block(const BlockL& B) {
    if (B.dir) pack_conv(B.rc);
    pack_conv(B.c1);
    pack_conv(B.c2);
    pack_conv(B.c3);
  }

  // Recurrent weights: canonical [dir][3H][H] row-major (each gru_cluster_kernel variant gathers its own register
  // image from it at kernel start), plus b_hn [dir][H].
  void pack_gru(const GruL& G) {
    pack_conv(G.proj);
    const int H = G.H;
    for (int d = 0; d < 2; d++) {
      std::string sfx = "_l" + std::to_string(G.layer) + (d ? "_reverse" : "");
      const HostTensor* whh = get(G.name + ".weight_hh" + sfx, {3 * H, H});
      const HostTensor* bhh = get(G.name + ".bias_hh" + sfx, {3 * H});
      if (!whh || !bhh) return;
      putf(G.whh_off + (size_t)d * 3 * H * H, whh->data.data(), (size_t)3 * H * H);
      for (int j = 0; j < H; j++) blob[G.bhn_off + (size_t)d * H + j] = bhh->data[2 * H + j];
    }
  }

  void pack_small_in(const SmallConvL& L, bool maybe_wn) {
    std::vector<double> w;
    (void)maybe_wn;
    if (!eff_weight(L.name, {L.Cout, 1, L.KW}, w)) return;
    const HostTensor* b = get(L.name + ".bias", {L.Cout});
    if (!b) return;
    put(L.w_off, w);
    putf(L.b_off, b->data.data(), L.Cout);
  }
};

}  // namespace

std::string pack_weights(const Model& m, const std::map<std::string, HostTensor>& sd, std::vector<float>& blob,
                         int& code) {
  blob.assign(m.total_floats, 0.0f);
  Packer P{sd, blob};
  const std::string& sp = m.score_prefix;
  // sigma block
  if (m.sigma.simple) {
    const HostTensor* w = P.get(sp + ".sigma_block.weight", {1, 1});
    const HostTensor* b = P.get(sp + ".sigma_block.bias", {1, 1});
    if (w && b) { blob[m.sigma.p_off] = w->data[0]; blob[m.sigma.p_off + 1] = b->data[0]; }
  } else {
    const int nr = m.sigma.n_rff;
    int dims[4] = {2 * nr, 4 * nr, 8 * nr, m.sigma.D};
    size_t off = m.sigma.p_off;
    const HostTensor* fq = P.get(sp + ".sigma_block.freq", {nr});
    if (fq) P.putf(off, fq->data.data(), nr);
    off += nr;
    for (int i = 0; i < 3; i++) {
      std::string q = sp + ".sigma_block.layer" + std::to_string(i + 1);
      const HostTensor* al = P.get(q + ".prelu.weight", {1});
      const HostTensor* w = P.get(q + ".lin.weight", {dims[i + 1], dims[i]});
      const HostTensor* b = P.get(q + ".lin.bias", {dims[i + 1]});
      if (al && w && b) {
        blob[off] = al->data[0];
        P.putf(off + 1, w->data.data(), w->data.size());
        P.putf(off + 1 + w->data.size(), b->data.data(), b->data.size());
      }
      off += 1 + (size_t)dims[i + 1] * dims[i] + dims[i + 1];
    }
  }
  P.pack_small_in(m.s_in, false);
  for (auto& b : m.s_enc) P.pack_block(b);
  P.pack_gru(m.s_gru);
  for (auto& b : m.s_dec) P.pack_block(b);
  for (auto& l : m.s_sig) P.pack_conv(l);
  // FiLM projections: encoder.cond_proj.i / decoder.noise_cond_proj.j (Linear D -> 2C, maybe weight-normed)
  {
    const int D = m.film.D;
    std::vector<double> w;
    for (int side = 0; side < 2; side++) {
      const auto& offs = side ? m.film.dec_off : m.film.enc_off;
      const auto& blocks = side ? m.s_dec : m.s_enc;
      for (size_t i = 0; i < offs.size(); i++) {
        std::string q = sp + (side ? ".decoder.noise_cond_proj." : ".encoder.cond_proj.") + std::to_string(i);
        int rows = 2 * blocks[i].C;
        if (!P.eff_weight(q, {rows, D}, w)) break;
        const HostTensor* b = P.get(q + ".bias", {rows});
        if (!b) break;
        P.put(m.film.w_off + (size_t)offs[i] * D, w);
        P.putf(m.film.b_off + offs[i], b->data.data(), rows);
      }
    }
  }
  // output conv: PReLU(score.prelu) -> PReLU(output_conv.prelu) -> Conv1d(C0 -> 1, k)   score.py:266-273,289
  {
    std::vector<double> w;
    if (P.eff_weight(m.s_out.name + ".conv", {1, m.s_out.Cin, m.s_out.KW}, w)) {
      const HostTensor* b = P.get(m.s_out.name + ".conv.bias", {1});
      const HostTensor* a1 = P.get(sp + ".prelu.weight", {1});
      const HostTensor* a2 = P.get(m.s_out.name + ".prelu.weight", {1});
      if (b && a1 && a2) {
        P.put(m.s_out.w_off, w);
        blob[m.s_out.b_off] = b->data[0];
        blob[m.s_out.a_off] = a1->data[0];
        blob[m.s_out.a_off + 1] = a2->data[0];
      }
    }
  }
  // conditioner
  {
    const std::string q = "condition_model.input_mel.mel_spec";
    const HostTensor* win = P.get(q + ".spectrogram.window", {m.mel.n_fft});
    const HostTensor* fb = P.get(q + ".mel_scale.fb", {m.mel.n_freq, m.mel.n_mels});
    if (win && fb) {
      P.putf(m.mel.win_off, win->data.data(), m.mel.n_fft);
      P.putf(m.mel.fb_off, fb->data.data(), fb->data.size());
    }
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < m.mel.n_fft; i++) {
      blob[m.mel.tw_off + i] = (float)std::cos(two_pi * i / m.mel.n_fft);
      blob[m.mel.tw_off + m.mel.n_fft + i] = (float)std::sin(two_pi * i / m.mel.n_fft);
    }
  }
  P.pack_conv(m.c_melconv);
  P.pack_block(m.c_melblock);
  P.pack_small_in(m.c_in, true);
  for (auto& b : m.c_enc) P.pack_block(b);
  for (auto& l : m.c_st) P.pack_conv(l);
  P.pack_gru(m.c_gru0);
  P.pack_gru(m.c_gru1);
  P.pack_block(m.c_cb1);
  P.pack_block(m.c_cb2);
  P.pack_block(m.c_decin);
  for (auto& b : m.c_dec) P.pack_block(b);
  if (m.dec.present) {
    const std::string p = "signal_decoupling_layer";
    if (m.dec.act == OU_ACT_SNAKE) {
      const HostTensor* al = P.get(p + ".prelu.act.act.alpha", {m.dec.C});
      const HostTensor* up = P.get(p + ".prelu.act.upsample.kernel", {2, 1, 15});
      const HostTensor* dn = P.get(p + ".prelu.act.downsample.kernel", {1, 1, 28});
      if (al && up && dn) {
        for (int i = 0; i < m.dec.C; i++) blob[m.dec.alpha_off + i] = std::exp(al->data[i]);  // log-scale alpha
        P.putf(m.dec.up_off, up->data.data(), 30);
        P.putf(m.dec.down_off, dn->data.data(), 28);
      }
    } else if (m.dec.act == OU_ACT_PRELU) {
      const HostTensor* a = P.get(p + ".prelu.weight", {1});
      if (a) blob[m.dec.prelu_off] = a->data[0];
    }
    const HostTensor* w = P.get(p + ".conv.weight", {1, m.dec.C, 3});
    const HostTensor* b = P.get(p + ".conv.bias", {1});
    if (w && b) { P.putf(m.dec.conv.w_off, w->data.data(), (size_t)m.dec.C * 3); blob[m.dec.conv.b_off] = b->data[0]; }
  }
  code = P.code;
  return P.err;
}

}  // namespace ou
