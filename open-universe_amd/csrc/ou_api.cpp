// C ABI of libouniverse.so (include/ouniverse.h): packer, handle, and the forward "runner" that walks
// the model and enqueues the gfx950 kernels on the caller's stream.  No allocation, no host sync in the
// forward calls.  There is no CPU path: without a HIP device ou_create fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ouniverse.h"
#include "../../include/ouniverse_tuning.h"
#include "ou_kernels.h"
#include "ou_model.h"

using namespace ou;

namespace {
thread_local std::string g_last_error;
constexpr int kMaxSteps = 256;
constexpr size_t kProfSlots = 32768;
constexpr float kInvSqrt2 = 0.70710678118654752440f;

struct TensorRef {
  size_t off;  // bytes into the workspace
  int C, T;
};
// Tuning / test switches of a handle (ou_set_option; the table of keys is kOptions below)
struct Options {
  int dbg = 0, xcd_map = -1, conv_direct = 5, fuse = -1, fuse_nc = 0, rate_small = 1, fuse_upfir = 1, block3 = 0, d4_fir = 1,
      d4_force = 0, d4_short = 1, d2_wk = 0, wino = 1, d2_map = -1, unfuse64 = 0, preact = 1;
  int gru_v = 2, gru_bmax = 0, gru_ts = 0, gru_upw = 0, gru_backoff = 0, gru_agent = -1, gru_dbg = 0;
  int split = -1;              // ConvArgs::split
  int dbg_dec0_under_gru = 0;  // measurement only, INVALID results (see run_score); experiments library only
  int tile_prefetch = 1;
  int trace = 0, no_overlap = 0, ts = 0, deep_factor = 8, mask_fused = 1, split_wino = 1, d2_tile_rule = 1;
  double tile_min = -1.0;      // < 0: the launcher's default
  std::string chain_ts;        // ou_set_stamp_layer (tuning)
};
}  // namespace

struct ou_packer {
  Model m;
  std::map<std::string, HostTensor> sd;
  std::vector<float> blob;
  std::string err;
};

struct ou_handle {
  Model m;
  const float* W = nullptr;  // packed blob (device)
  int device = 0;
  int num_cu = 256;
  std::string err;
  std::map<std::string, TensorRef> tensors;
  int n_launch = 0, n_conv = 0;
  // geometry of the last ou_condition (consumed by ou_score)
  int cond_B = 0, cond_T = 0;
  bool trace = false;
  Options opt;                    // ou_set_option
  std::string plan_with_options;  // ou_plan_json's answer (the plan + the current option values)
  int last_cfg = -1;
  long long* tstamps = nullptr;
  std::map<size_t, float> alphas;  // host copies of the PReLU slopes (blob offset -> value)
  // side streams for independent branches (mel front-end, st convs, first score-encoder pass), fork/joined to the
  // caller's stream with events: still no host synchronisation, still graph-capturable
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> events;
  size_t ev_used = 0;
  bool overlap = true;
  int force_cfg = -1, force_sc = 0;  // micro-benchmark overrides (ou_bench_conv)
  // per-launch HIP-event profiling of the generic conv kernel (bench.py roofline)
  bool profile = false;
  struct ProfRec { double flops, bytes; int cfg; };
  std::vector<ProfRec> prof;
  size_t prof_used = 0;
  unsigned long long* prof_dev = nullptr;  // [kProfSlots][32] device-side {16 x min start, 16 x ~max end} ticks
  int fuse_mode = -1;  // OU_FUSE: -1 auto (cost model), 0 never, 2 / 3 force that depth where the shape allows
  int fuse_nc = 0;     // OU_FUSE_NC: force 128 / 256 columns per tile
  // workspaces that ou_workspace_init has prepared (cleared status word, GRU tag epochs and exchange areas) and the
  // shape each was prepared for: the forward calls refuse anything else -- an uninitialised buffer would feed the
  // recurrence kernels a garbage epoch and garbage tags
  struct WsRec { const void* ws; size_t bytes; int B, T; };
  std::vector<WsRec> ws_ready;
  int gru_agent_stores = 0;  // ou_set_gru_publish_mode; also set by ou_check_device_status when the safety net had to act
  // enhance calls in flight side by side in this process, one handle + stream + workspace each (ou_set_lanes): the GRU
  // launches of all lanes have to be resident together
  int lanes = 1, lane = 0;
  int lane_max_b = 0;  // ou_set_lane_batch: largest batch size any lane of the pool runs (0: every call's own B)
  // A workspace prepared for (B, T0) serves every T of the same batch size that fits into it: everything ou_workspace_init
  // prepares (status words, tag epochs, GRU exchange areas) lies in a header whose layout depends on B alone, and the tag
  // epochs advance monotonically whatever the length of a pass -- a directory of files of different lengths runs on ONE
  // workspace sized for the longest.  (Too small a buffer is caught by the walk itself: OU_ENOMEM.)
  bool ws_ok(const void* ws, size_t bytes, int B, int T) const {
    (void)T;
    for (const WsRec& r : ws_ready)
      if (r.ws == ws) return r.B == B && r.bytes <= bytes;
    return false;
  }
};

namespace {

int fail(ou_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_last_error = msg;
  return code;
}

struct Tensor {
  float* p = nullptr;
  int C = 0, T = 0;
};

// Bump allocator over the caller's workspace.  In `dry` mode nothing is launched and `base` may be null:
// the same walk then only measures the footprint (ou_workspace_bytes) -> layout is a pure function of
// (config, B, T).
// Tuning / test switches.  Until ABI 4 these were ~35 OU_* environment variables read by the library; since ABI 5
// the library reads NO environment variable: every switch is a typed option of the handle (ou_set_option / ou_get_option,
// include/ouniverse.h), echoed by ou_plan_json, and a forward call works on a copy taken when it starts.
struct OptDesc {
  const char* key;
  int Options::*ip;
  double Options::*dp;
  bool experiments_only;  // switches that make a call return WRONG results by design: honoured by `make EXPERIMENTS=1` builds only
  const char* doc;
};
const OptDesc kOptions[] = {
    {"conv_direct", &Options::conv_direct, nullptr, false, "kernel generations the conv launcher may use (0 .. 5, ConvArgs::direct)"},
    {"split", &Options::split, nullptr, false, "-1 rule / 0 never / 1 wherever possible: conv_split_kernel (bf16-split operands)"},
    {"split_wino", &Options::split_wino, nullptr, false, "experiments build: 0 = the plain bf16-split kernel also where its minimal-filtering form would take the layer"},
    {"wino", &Options::wino, nullptr, false, "0: never the minimal-filtering (Winograd / Cook-Toom) kernel variants"},
    {"fuse", &Options::fuse, nullptr, false, "-1 cost model / 0 never / 2 / 3: depth of the fused ConvBlock body (conv_chain kernels)"},
    {"fuse_nc", &Options::fuse_nc, nullptr, false, "128 / 256: columns per tile of the fused ConvBlock body (0: launcher's choice)"},
    {"fuse_upfir", &Options::fuse_upfir, nullptr, false, "0: up-path anti-alias FIR always as its own pass"},
    {"rate_small", &Options::rate_small, nullptr, false, "0: outermost rate-change convs on the generic kernels (no rate_down / rate_up)"},
    {"preact", &Options::preact, nullptr, false, "0: every PReLU in its consumer's operand path (ConvArgs::out_act never set)"},
    {"unfuse64", &Options::unfuse64, nullptr, false, "1: 64-channel ConvBlock bodies as three split-K launches"},
    {"block3", &Options::block3, nullptr, false, "1: the three body convs of a deep-level ConvBlock in one launch (experiments build only)"},
    {"xcd_map", &Options::xcd_map, nullptr, false, "block -> tile mapping of the LDS-tiled conv kernel (-1: launcher's choice)"},
    {"d2_map", &Options::d2_map, nullptr, false, "block -> tile mapping of the wide-load split-K kernels (-1: launcher's choice)"},
    {"d2_tile_rule", &Options::d2_tile_rule, nullptr, false, "0: round 5's 64- vs 32-column rule of the wide-load split-K kernels (ignores that only 64-column tiles have minimal filtering)"},
    {"d2_wk", &Options::d2_wk, nullptr, false, "4 / 8: K slices of conv_direct2_kernel (0: launcher's rule)"},
    {"d4_fir", &Options::d4_fir, nullptr, false, "0: up convs with a fusable FIR stay on the first-generation fused kernel"},
    {"d4_short", &Options::d4_short, nullptr, false, "0: the 401-frame levels at batch 1 stay on the first-generation kernels"},
    {"d4_force", &Options::d4_force, nullptr, false, "10 TM + log2(WK): that conv_direct4 tile shape wherever a layer admits it"},
    {"tile_min", nullptr, &Options::tile_min, false, "wave tiles per SIMD from which the no-split-K kernels take a layer (< 0: 1.2)"},
    {"tile_prefetch", &Options::tile_prefetch, nullptr, false, "0: no LDS prefetch of the epilogue operand in conv_direct3_kernel"},
    {"deep_factor", &Options::deep_factor, nullptr, false, "blocks of 64 x 128 per CU up to which a layer counts as 'deep' (split-K kernels)"},
    {"mask_fused", &Options::mask_fused, nullptr, false, "ragged batches: 0 = a separate tail-mask launch after EVERY producer (reference form of the masks)"},
    {"gru_v", &Options::gru_v, nullptr, false, "2 ring kernel / 1 polling-wave kernel (experiments build)"},
    {"gru_bmax", &Options::gru_bmax, nullptr, false, "cap on the utterances per GRU launch (forces the chunked path)"},
    {"gru_upw", &Options::gru_upw, nullptr, false, "hidden units per workgroup of the ring kernel (0: from the batch size)"},
    {"gru_backoff", &Options::gru_backoff, nullptr, false, "poll back-off experiments of the ring kernel (0: none)"},
    {"gru_agent_stores", &Options::gru_agent, nullptr, false, "-1 handle's mode / 0 plain / 1 agent-scope publishes"},
    {"gru_dbg", &Options::gru_dbg, nullptr, false, "bit 0 no republish safety net, bit 1 system-scope publishes, bit 2 FAULT INJECTION (tests)"},
    {"gru_ts", &Options::gru_ts, nullptr, false, "1: per-wave cycle stamps of the GRU kernel into the end of the workspace (tuning)"},
    {"ts", &Options::ts, nullptr, false, "1: per-wave phase stamps in ou_bench_conv (tuning)"},
    {"trace", &Options::trace, nullptr, false, "1: one line per conv launch on stderr"},
    {"no_overlap", &Options::no_overlap, nullptr, false, "1: no side streams inside a call (every kernel alone on the device)"},
    {"dbg", &Options::dbg, nullptr, true, "phase ablation switches of the conv kernels: WRONG results by design"},
    {"dbg_dec0", &Options::dbg_dec0_under_gru, nullptr, true, "first decoder block under the GRU: upper-bound measurement, WRONG results"},
};
constexpr int kNumOptions = (int)(sizeof(kOptions) / sizeof(kOptions[0]));
const OptDesc* find_option(const char* key) {
  for (int i = 0; i < kNumOptions; i++)
    if (std::strcmp(kOptions[i].key, key) == 0) return &kOptions[i];
  return nullptr;
}
using EnvCfg = Options;

struct Runner {
  ou_handle* h;
  EnvCfg env;
  char* base;
  size_t cap;
  size_t off = 0;
  bool dry;
  hipStream_t st;
  int B;
  hipError_t herr = hipSuccess;
  bool oom = false;
  const char* where = "";

  Runner(ou_handle* h_, void* ws, size_t cap_, bool dry_, hipStream_t st_, int B_)
      : h(h_), env(h_->opt), base((char*)ws), cap(cap_), dry(dry_), st(st_), B(B_), main_st(st_) {
#ifndef OU_EXPERIMENTS
    env.dbg = 0; env.dbg_dec0_under_gru = 0;  // (switches with WRONG results by design: experiments library only)
#endif
  }

  float* alloc_raw(size_t floats) {
    size_t bytes = (floats * 4 + 255) & ~size_t(255);
    size_t o = off;
    off += bytes;
    if (!dry && off > cap) { oom = true; return (float*)base; }
    return dry ? nullptr : (float*)(base + o);
  }
  Tensor alloc(const std::string& name, int C, int T) {
    size_t o = off;
    Tensor t;
    t.p = alloc_raw((size_t)B * C * T);
    t.C = C;
    t.T = T;
    if (!name.empty()) h->tensors[name] = TensorRef{o, C, T};
    return t;
  }
  hipStream_t main_st = nullptr;
  hipEvent_t next_event() {
    if (h->ev_used == h->events.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      h->events.push_back(e);
    }
    return h->events[h->ev_used++];
  }
  // make side stream k wait for everything enqueued so far on `from`
  void fork(hipStream_t from, int k) {
    if (dry) return;
    hipEvent_t e = next_event();
    if (!e) { herr = hipErrorOutOfMemory; where = "event"; return; }
    chk(hipEventRecord(e, from), "fork record");
    chk(hipStreamWaitEvent(h->aux[k], e, 0), "fork wait");
  }
  // make `to` wait for everything enqueued so far on side stream k
  void join(int k, hipStream_t to) {
    if (dry) return;
    hipEvent_t e = next_event();
    if (!e) { herr = hipErrorOutOfMemory; where = "event"; return; }
    chk(hipEventRecord(e, h->aux[k]), "join record");
    chk(hipStreamWaitEvent(to, e, 0), "join wait");
  }
  // ---- ragged batch (ou_enhance_var): per-row lengths on every level, see ou_kernels.h
  bool ragged = false;
  const int* lens_dev = nullptr;  // [lv.n][B]
  const RowInfo* rows_dev = nullptr;
  LevelSpec lv;
  int level_T[kMaxLenLevels] = {0};
  // per-row lengths of a (B, C, Tl) tensor (null when all rows are whole)
  const int* lens_of(int Tl) {
    if (!ragged) return nullptr;
    for (int l = 0; l < lv.n; l++)
      if (level_T[l] == Tl) return lens_dev + (size_t)l * B;
    if (herr == hipSuccess) { herr = hipErrorInvalidValue; where = "ragged batch: tensor length on no level"; }
    return nullptr;
  }
  // keep the invariant "zero from the row's own length on" after a kernel that does not keep it itself
  void mask(float* p, int C, int T) {
    if (!ragged || dry || !ok()) return;
    const int* ln = lens_of(T);
    if (ln) chk(launch_mask_tail(p, ln, B, C, T, st), "mask tail");
  }
  void mask(const Tensor& t) { mask(t.p, t.C, t.T); }
  bool ok() const { return herr == hipSuccess && !oom; }
  void chk(hipError_t e, const char* w) {
    if (e != hipSuccess && herr == hipSuccess) { herr = e; where = w; }
    h->n_launch++;
  }
  const float* W(size_t off_floats) const { return h->W + off_floats; }

  struct Epi {
    const float* add = nullptr;
    float add_scale = 1.f;
    const float* film = nullptr;
    int film_bstride = 0;
    const float* res = nullptr;
    float res_scale = 1.f;
    const float* in_scale = nullptr;
    bool act = true;  // apply the layer's PReLU prologue (if it has one)
    // up-path anti-alias FIR fused into the conv epilogue: taps, 2r + 1, the manual bias added after the FIR.  conv()
    // sets `unsupported` instead of failing when no kernel with that epilogue fits the layer.
    const float* fir = nullptr;
    int fir_len = 0;
    const float* fir_bias = nullptr;
    // small-K rate-change conv of a wide level on rate_down_kernel (`fir` = the filter applied BEFORE the conv, or null)
    bool rate_down = false;
    bool rate_up = false;  // the last up conv on rate_up_kernel (`fir` = the filter AFTER the conv or null, fir_bias / res)
    // store prelu(y; out_alpha) instead of y (ConvArgs::out_act): for outputs whose only reader is the next PReLU_Conv.  conv()
    // leaves in `stored_act` whether the kernel that took the layer did so (else y is stored and the reader keeps its PReLU).
    bool out_act = false;
    float out_alpha = 1.f;
    bool no_mask = false;  // ragged batch: the caller fills the tail itself (GRU input projections)
  };
  bool stored_act = false;
  bool unsupported = false;
  // conv() in collect mode: the launch arguments are appended here instead of being launched (block(): the three body convs
  // of a deep-level ConvBlock in one launch, conv_block3_kernel)
  std::vector<ConvArgs>* collect = nullptr;
  unsigned* status_words = nullptr;        // workspace header (layout_persist)
  unsigned long long* block3_bar = nullptr;
  bool gru_shared = false;  // GRU launches enqueued now may run beside another GRU layer (overlapped conditioner / score pass)
  hipEvent_t pre_gru = nullptr;  // OU_DBG_DEC0: recorded right before the score net's GRU launch
  bool want_pre_gru = false;
  // GRU launches that can meet on one XCD: this call's own two layers when they overlap, times the lanes whose clusters are
  // dealt to the same XCDs (lane l deals its 2 B clusters from XCD 2 B l on)
  static int gru_share_of(int lanes, int B, bool overlap) {
    const int ncl = 2 * B < 8 ? 2 * B : 8;
    const int by_lanes = lanes > 1 ? (ncl * lanes + 7) / 8 : 1;
    return by_lanes * (overlap ? 2 : 1);
  }

  Tensor conv(const ConvL& L, const Tensor& in, const std::string& name, const Epi& e, const Tensor* dst = nullptr) {
    int Nq, Tout;
    if (L.stride > 1) { Nq = in.T / L.stride; Tout = Nq; }
    else { Nq = in.T; Tout = in.T * L.up; }
    Tensor out = dst ? *dst : alloc(name, L.Cout, Tout);
    if (dry || !ok()) return out;
    ConvArgs a;
    a.x = in.p; a.w = W(L.w_off); a.bias = W(L.b_off); a.y = out.p;
    if (L.KWP) { a.wd = W(L.wd_off); a.wu = W(L.wu_off); }
    if (L.ws_on) a.wsplit = W(L.ws_off);
    if (L.wsw_on) a.wsplitw = W(L.wsw_off);
    a.split = env.split; a.split_wino = env.split_wino;
    a.in_scale = e.in_scale;
    a.act = (L.act && e.act) ? 1 : 0;
    a.alpha_val = a.act ? h->alphas[L.a_off] : 0.f;
    a.add = e.add; a.add_scale = e.add_scale;
    a.film = e.film; a.film_bstride = e.film_bstride;
    a.res = e.res; a.res_scale = e.res_scale;
    if (e.out_act && !collect && !e.fir && !e.rate_down && !e.rate_up) { a.out_act = 1; a.out_alpha = e.out_alpha; }
    stored_act = false;
    if (e.fir && !e.rate_down) { a.fir = e.fir; a.fir_len = e.fir_len; a.bias = e.fir_bias; }  // (also rate_up)
    if (e.rate_down) { a.fir = e.fir; a.fir_len = e.fir ? e.fir_len : 0; }
    a.B = B; a.Cin = L.Cin; a.Tin = in.T; a.Cout = L.Cout; a.M = L.M; a.Mp = L.Mp; a.KW = L.KW;
    a.stride = L.stride; a.pad = L.pad; a.up = L.up; a.CK = L.CK; a.Nq = Nq; a.Tout = Tout;
    a.force_cfg = h->force_cfg; a.force_sc = h->force_sc;
    a.dbg = env.dbg; a.force_xcd_map = env.xcd_map; a.direct = env.conv_direct; a.d4_fir_unfused = env.d4_fir; a.d4_force = env.d4_force; a.d4_short = env.d4_short; a.deep_factor = env.deep_factor; a.d2_tile_rule = env.d2_tile_rule; a.d2_wk = env.d2_wk; a.wino = env.wino; a.d2_map = env.d2_map;
    a.tile_min = env.tile_min; a.tile_prefetch = env.tile_prefetch;
    a.tstamps = h->tstamps;
    // ragged batch: the kernel keeps "zero behind the row's own end" in its epilogue where its family can (conv_masks_rows)
    if (ragged && env.mask_fused && !e.no_mask) a.lens = lens_of(Tout);
    if (collect) { collect->push_back(a); return out; }
    int cfg = -1;
    if (h->profile && h->prof_dev && h->prof_used < kProfSlots) {
      ou_handle::ProfRec rec;
      // algorithmic (reference, un-folded) work of this layer: dense FLOPs, activations once, weights once
      const int kref = L.KW;  // the packed layers carry the reference's own kernel sizes
      rec.flops = 2.0 * L.M * (double)Nq * L.Cin * kref * B;
      rec.bytes = 4.0 * ((double)B * ((double)L.Cin * in.T + (double)L.Cout * Tout) + (double)L.M * L.Cin * kref);
      rec.cfg = -1;
      a.prof = h->prof_dev + 32 * h->prof_used;
      h->prof.push_back(rec);
      h->prof_used++;
    }
    unsupported = false;
    if (e.rate_down) {
      chk(launch_rate_down(a, st, &cfg), L.name.c_str());
    } else if (e.rate_up) {
      chk(launch_rate_up(a, st, &cfg), L.name.c_str());
    } else {
      hipError_t le = launch_conv(a, h->num_cu, st, &cfg);
      if (le == hipErrorNotSupported && a.out_act) {  // the kernel for this layer has no activating epilogue: store y
        a.out_act = 0;
        le = launch_conv(a, h->num_cu, st, &cfg);
      }
      stored_act = a.out_act != 0;
      if (le == hipErrorNotSupported && e.fir) {  // the caller falls back to conv + launch_fir
        if (a.prof) { h->prof.pop_back(); h->prof_used--; }
        unsupported = true;
        return out;
      }
      chk(le, L.name.c_str());
    }
    if (a.prof) h->prof.back().cfg = cfg;
    h->last_cfg = cfg;
    if (!e.no_mask && !(a.lens && conv_masks_rows(cfg))) mask(out);
    if (h->trace)
      std::fprintf(stderr, "OU_TRACE conv %-64s cfg=%d M=%d Nq=%d K=%d(Cin=%d KW=%d CK=%d) stride=%d up=%d B=%d MFLOP=%.1f\n",
                   name.c_str(), cfg, L.M, Nq, L.Cin * L.KW, L.Cin, L.KW, L.CK, L.stride, L.up, B,
                   2.0 * L.M * Nq * L.Cin * L.KW * B * 1e-6);
    h->n_conv++;
    return out;
  }

  struct BlockOut { Tensor h_next, v, c1; };
  // ConvBlock.forward  (blocks.py:327-412).  `res` for dir==0 blocks must already be folded into `hin`.
  // How the three body convs of a ConvBlock run: 0 = three generic launches, 3 = one fused launch, 2 = conv1 generic +
  // fused (conv2, conv3).  Pure function of (layer shapes, B, T, device, OU_FUSE*) -- estimated cycles, see
  // chain_cost(); the generic launches are priced at their measured ~45 TFLOP/s with a 12 us floor.
  int plan_chain(const BlockL& Bk, int T) {
    h->fuse_mode = env.fuse; h->fuse_nc = env.fuse_nc;
    if (h->fuse_mode == 0) return 0;
    // (ragged batch: conv_chainw_kernel zeroes its LDS tiles and its output behind every row's own end -- ChainArgs::lens)
    if (ragged && !env.mask_fused) return 0;  // (the separate-mask form has no place to mask inside a fused body)
    // Throughput regime: with >= ~2 wave tiles per SIMD the three convs run unfused on conv_direct3_kernel at 70-100 TFLOP/s
    // each, ahead of the fused body's ~75 (measured end to end: PP16 B = 4 19.2 -> 18.4 ms, OR16 B = 16 57.7 -> 55.5 ms, B = 8
    // even); below that the fused launch wins (B = 1: 24 us for all three convs).
    if (h->fuse_mode < 0 && Bk.C % 16 == 0 && Bk.c1.KWP && Bk.c2.KWP && Bk.c3.KWP) {
      if (env.conv_direct >= 3 && direct3_tiles_per_simd(Bk.C, T, B, h->num_cu) >= 1.9 && T >= 1024) return 0;
      // Round 5, measured and left off: with minimal filtering the three split-K launches of a 64-channel body (17.0 + 11.7 +
      // 11.7 us at B = 1, back to back) look level with conv1 + the fused pair (17.0 + 24.8) -- end to end the unfused form is
      // 0.04-0.07 ms per enhance SLOWER (6.88-6.91 vs 6.83-6.85 ms, three alternating runs).  OU_UNFUSE64=1 selects it.
      if (env.unfuse64 && env.conv_direct >= 5 && env.wino && Bk.C % 64 == 0 && T >= 1024) return 0;
    }
    auto shape = [&](int depth) {
      ChainArgs ca;
      ca.depth = depth; ca.B = B; ca.C = Bk.C; ca.T = T; ca.Mp = Bk.c1.Mp; ca.force_nc = h->fuse_nc;
      const ConvL* ls[3] = {&Bk.c1, &Bk.c2, &Bk.c3};
      for (int s2 = 0; s2 < depth; s2++) {
        const ConvL& L = *ls[3 - depth + s2];
        if (!L.act || L.stride != 1 || L.up != 1 || L.Cin != Bk.C || L.Cout != Bk.C || L.pad != (L.KW - 1) / 2 ||
            L.Mp != Bk.c1.Mp)
          ca.depth = 0;
        ca.cv[s2].KW = L.KW; ca.cv[s2].CK = L.CK;
        if (L.KWP) ca.cv[s2].wu = h->W;  // (a marker: the layer HAS the Winograd-domain copy -- what chain_cost's shape test asks)
      }
      ca.wino = env.wino && env.conv_direct >= 5;
      ca.lens = ragged ? lens_of(T) : nullptr;
      return ca;
    };
    auto generic = [&](const ConvL& L) {
      const double cyc = 2.0 * L.M * (double)T * L.Cin * L.KW * B / 45e12 * 2.3e9;
      return cyc > 28000.0 ? cyc : 28000.0;
    };
    const double c3 = chain_cost(shape(3), h->num_cu, nullptr);
    double c2 = chain_cost(shape(2), h->num_cu, nullptr);
    if (c2 >= 0) c2 += generic(Bk.c1);
    if (h->fuse_mode == 3) return c3 >= 0 ? 3 : 0;
    if (h->fuse_mode == 2) return c2 >= 0 ? 2 : 0;
    const double c0 = generic(Bk.c1) + generic(Bk.c2) + generic(Bk.c3);
    int best = 0;
    double bc = c0;
    if (c3 >= 0 && c3 < bc) { best = 3; bc = c3; }
    if (c2 >= 0 && c2 < bc) { best = 2; bc = c2; }
    return best;
  }

  // `c1_dst` / `v_dst`: caller-provided (persistent) tensors for the conv1 result / the block output, so that
  // conditioner outputs are produced in place instead of being copied out of the scratch area afterwards
  BlockOut block(const BlockL& Bk, const Tensor& hin, const std::string& nm, const float* film, int film_bs,
                 const float* input_cond, const float* res, bool need_c1 = false, const Tensor* c1_dst = nullptr,
                 const Tensor* v_dst = nullptr) {
    Tensor hu = hin;
    bool small_up = false;
    if (Bk.dir == 2 && (Bk.rc.fir_mode == 0 || Bk.rc.fir_mode == 2)) {  // pure function of the layer shape
      ConvArgs probe;
      probe.up = Bk.rc.up; probe.stride = Bk.rc.stride; probe.KW = Bk.rc.KW; probe.pad = Bk.rc.pad; probe.Tin = hin.T;
      probe.Nq = hin.T; probe.Tout = hin.T * Bk.rc.up; probe.M = Bk.rc.M; probe.Cout = Bk.rc.Cout; probe.Cin = Bk.rc.Cin;
      probe.fir = Bk.rc.fir_mode == 2 ? h->W : nullptr; probe.fir_len = Bk.rc.fir_len;
      small_up = env.rate_small != 0 && rate_up_supported(probe);
    }
    if (small_up) {
      Epi e;
      e.rate_up = true;
      e.res = res; e.res_scale = kInvSqrt2;
      if (Bk.rc.fir_mode == 2) { e.fir = W(Bk.rc.fir_off); e.fir_len = Bk.rc.fir_len; e.fir_bias = W(Bk.rc.fbias_off); }
      hu = conv(Bk.rc, hin, nm + ".up", e);
    } else if (Bk.dir == 2) {
      if (Bk.rc.fir_mode == 2) {
        // PReLU -> transposed conv (r phase GEMMs) -> FIR + bias + residual add: fused into the conv's epilogue where
        // the direct kernel takes the layer, else as one bandwidth pass after it
        Tensor u = alloc(nm + ".upc", Bk.rc.Cout, hin.T * Bk.rc.up);
        hu = alloc(nm + ".up", u.C, u.T);
        if (!dry && ok()) {
          const bool fuse = env.fuse_upfir != 0;
          bool done = false;
          if (fuse) {
            Epi e;
            e.res = res; e.res_scale = kInvSqrt2;
            e.fir = W(Bk.rc.fir_off); e.fir_len = Bk.rc.fir_len; e.fir_bias = W(Bk.rc.fbias_off);
            conv(Bk.rc, hin, nm + ".up", e, &hu);
            done = !unsupported;
          }
          if (!done) {
            conv(Bk.rc, hin, nm + ".upc", Epi(), &u);
            if (ok())
            {
              const int* ln = env.mask_fused ? lens_of(u.T) : nullptr;
              chk(launch_fir(u.p, W(Bk.rc.fir_off), Bk.rc.fir_len, 0.f, 0, W(Bk.rc.fbias_off), res, kInvSqrt2, hu.p, B,
                             u.C, u.T, st, ln), "fir(up)");
              if (!ln) mask(hu);
            }
          }
        }
      } else {
        // (fir_mode 4: FIR folded into 3-tap phase GEMMs by the packer, its manual bias = the conv bias)
        Epi e;
        e.res = res; e.res_scale = kInvSqrt2;  // blocks.py:374-376 fused into the up-conv epilogue
        hu = conv(Bk.rc, hin, nm + ".up", e);
      }
    }
    Epi e1;
    if (input_cond) { e1.add = input_cond; e1.add_scale = kInvSqrt2; }  // blocks.py:384-386
    e1.film = film; e1.film_bstride = film_bs;                           // blocks.py:393-394
    Tensor c1 = c1_dst ? *c1_dst : alloc(nm + ".c1", Bk.c1.Cout, hu.T);
    Tensor c2 = alloc(nm + ".c2", Bk.c2.Cout, hu.T);
    Tensor v = v_dst ? *v_dst : alloc(nm + ".v", Bk.c3.Cout, hu.T);
    auto chain_conv = [&](const ConvL& L) {
      ChainConv c;
      c.w = W(L.w_off); c.bias = W(L.b_off); c.alpha = h->alphas[L.a_off]; c.KW = L.KW; c.CK = L.CK;
      if (L.KWP) c.wu = W(L.wu_off);
      return c;
    };
    // wide, shallow levels: the body runs as one fused launch (conv_chain_kernel), or conv1 + a fused (conv2, conv3)
    const int depth = dry ? 0 : plan_chain(Bk, hu.T);

    if (depth == 3 || depth == 2) {
      if (depth == 2) conv(Bk.c1, hu, nm + ".c1", e1, &c1);
      if (ok()) {
        ChainArgs ca;
        ca.depth = depth; ca.B = B; ca.C = Bk.C; ca.T = hu.T; ca.Mp = Bk.c1.Mp;
        ca.x = depth == 3 ? hu.p : c1.p;
        ca.y = v.p; ca.res = hu.p; ca.res_scale = kInvSqrt2;
        if (depth == 3) {
          ca.add = e1.add; ca.add_scale = e1.add_scale; ca.film = e1.film; ca.film_bstride = e1.film_bstride;
          ca.c1_out = need_c1 ? c1.p : nullptr;
          ca.cv[0] = chain_conv(Bk.c1); ca.cv[1] = chain_conv(Bk.c2); ca.cv[2] = chain_conv(Bk.c3);
        } else {
          ca.cv[0] = chain_conv(Bk.c2); ca.cv[1] = chain_conv(Bk.c3);
        }
        ca.force_nc = h->fuse_nc;
        ca.wino = env.wino && env.conv_direct >= 5;
        ca.lens = lens_of(hu.T);
        if (!env.chain_ts.empty() && nm == env.chain_ts) ca.tstamps = (long long*)(base + cap - (16u << 20));
        int variant = -1;
        if (h->profile && h->prof_dev && h->prof_used < kProfSlots) {
          ou_handle::ProfRec rec;
          rec.flops = 0;
          double wbytes = 0;
          for (int s2 = 0; s2 < depth; s2++) {
            rec.flops += 2.0 * Bk.C * (double)hu.T * Bk.C * ca.cv[s2].KW * B;
            wbytes += 4.0 * Bk.C * Bk.C * ca.cv[s2].KW;
          }
          // activations: block input (also the residual) once, output once, the cond add when present
          rec.bytes = 4.0 * B * (double)Bk.C * hu.T * (2 + (depth == 2 ? 1 : 0) + (ca.add ? 1 : 0)) + wbytes;
          rec.cfg = -1;
          ca.prof = h->prof_dev + 32 * h->prof_used;
          h->prof.push_back(rec);
          h->prof_used++;
        }
        chk(launch_chain(ca, h->num_cu, st, &variant), nm.c_str());
        if (ca.prof) h->prof.back().cfg = variant;
        if (h->trace)
          std::fprintf(stderr, "OU_TRACE chain %-63s variant=%d depth=%d C=%d T=%d B=%d\n", nm.c_str(), variant, depth,
                       Bk.C, hu.T, B);
        h->n_conv++;
      }
    } else {
      Epi e3;
      e3.res = hu.p; e3.res_scale = kInvSqrt2;  // blocks.py:399
      if (dry) e3.res = nullptr;
      // Deep levels at batch 1: the three convs in ONE launch (conv_block3_kernel) where the shape fits -- on the caller's
      // stream only (its workgroups wait for each other: one such kernel at a time), not while profiling per layer.
      // OFF by default (OU_BLOCK3=1): 41.7 / 42.3 us per fused launch (C = 512 / 256) against 44.3 / 44.1 us for the three
      // launches with their gaps, and the enhance as a whole 0.1 ms SLOWER with it.
      bool fused = false;
      if (!dry && ok() && env.block3 != 0 && B == 1 && !ragged && st == main_st && block3_bar && !h->profile && !h->tstamps &&
          h->force_cfg < 0 && env.conv_direct >= 2) {
        std::vector<ConvArgs> cv;
        collect = &cv;
        conv(Bk.c1, hu, nm + ".c1", e1, &c1);
        conv(Bk.c2, c1, nm + ".c2", Epi(), &c2);
        conv(Bk.c3, c2, nm + ".v", e3, &v);
        collect = nullptr;
        int cfg = -1;
        const hipError_t le = cv.size() == 3 ? launch_conv_block3(cv.data(), block3_bar, status_words, h->num_cu, st, &cfg)
                                             : hipErrorInvalidConfiguration;
        if (le == hipSuccess) {
          fused = true;
          h->last_cfg = cfg;
          h->n_conv++;
          if (h->trace)
            std::fprintf(stderr, "OU_TRACE block3 %-62s cfg=%d C=%d T=%d\n", nm.c_str(), cfg, Bk.C, hu.T);
        } else if (le != hipErrorInvalidConfiguration) {
          chk(le, nm.c_str());
        }
      }
      if (!fused) {
        // c1 (unless it is exported as a condition) and c2 are read by the next conv only: stored ACTIVATED by the epilogue of the
        // conv that produces them, so that the reader's operand path has no PReLU (ConvArgs::out_act; bit-identical)
        const bool c1_private = !need_c1 && !c1_dst;
        if (env.preact && c1_private && Bk.c2.act) { e1.out_act = true; e1.out_alpha = h->alphas[Bk.c2.a_off]; }
        conv(Bk.c1, hu, nm + ".c1", e1, &c1);
        Epi e2;
        e2.act = !stored_act;
        if (env.preact && Bk.c3.act) { e2.out_act = true; e2.out_alpha = h->alphas[Bk.c3.a_off]; }
        conv(Bk.c2, c1, nm + ".c2", e2, &c2);
        e3.act = !stored_act;
        conv(Bk.c3, c2, nm + ".v", e3, &v);
      }
    }
    BlockOut o;
    o.v = v; o.c1 = c1; o.h_next = v;
    if (Bk.dir == 1) {  // blocks.py:401-410
      bool small = false;
      if (Bk.rc.fir_mode <= 1) {  // wide levels: FIR + strided conv in one launch (pure function of the layer shape)
          ConvArgs probe;
        probe.up = Bk.rc.up; probe.stride = Bk.rc.stride; probe.KW = Bk.rc.KW; probe.pad = Bk.rc.pad; probe.Tin = v.T;
        probe.Nq = v.T / Bk.rc.stride; probe.M = Bk.rc.M; probe.Cin = Bk.rc.Cin;
        probe.fir = Bk.rc.fir_mode == 1 ? h->W : nullptr; probe.fir_len = Bk.rc.fir_len;
        small = env.rate_small != 0 && rate_down_supported(probe);
      }
      if (small) {
        Epi e;
        e.rate_down = true;
        if (Bk.rc.fir_mode == 1) { e.fir = W(Bk.rc.fir_off); e.fir_len = Bk.rc.fir_len; }
        o.h_next = conv(Bk.rc, v, nm + ".h", e);
      } else if (Bk.rc.fir_mode == 1) {
        Tensor xf = alloc(nm + ".fir", v.C, v.T);
        if (!dry && ok())
          chk(launch_fir(v.p, W(Bk.rc.fir_off), Bk.rc.fir_len, h->alphas[Bk.rc.a_off], 1, nullptr, nullptr, 1.f, xf.p, B,
                         v.C, v.T, st), "fir(down)");
        // (ragged batch: no mask needed -- the k = s = r conv that reads xf has no halo, and its own output is masked)
        Epi e;
        e.act = false;  // PReLU applied by the FIR pass
        o.h_next = conv(Bk.rc, xf, nm + ".h", e);
      } else {
        o.h_next = conv(Bk.rc, v, nm + ".h", Epi());  // (fir_mode 3: FIR folded into the 3r-tap weights)
      }
    }
    return o;
  }

  // one bidirectional GRU layer: projection GEMM + cluster recurrence
  Tensor gru(const GruL& G, const Tensor& in, const std::string& nm, unsigned long long* xchg, unsigned* errw,
             unsigned* epoch, const float* res, float res_scale) {
    Epi e;
    e.act = false;
    e.no_mask = true;
    Tensor gx = conv(G.proj, in, nm + ".gx", e);
    Tensor out = alloc(nm, 2 * G.H, in.T);
    if (dry || !ok()) return out;
    if (ragged) {  // frames behind a row's own end hold the state (z = 1): see launch_gru_tail_fill
      const int* ln = lens_of(in.T);
      if (ln) chk(launch_gru_tail_fill(gx.p, ln, B, G.H, in.T, st), "gru tail fill");
    }
    GruArgs a;
    a.gx = gx.p; a.whh = W(G.whh_off); a.bhn = W(G.bhn_off); a.out = out.p; a.res = res; a.res_scale = res_scale;
    a.xchg = xchg; a.err = errw; a.epoch = epoch; a.B = B; a.T = in.T; a.H = G.H;
    // kernel generation: the ring kernel (every wave gathers h straight from L2, no polling wave, no workgroup barrier)
    // for every batch size; OU_GRU_V=1 selects the polling-wave kernel of round 1.  Its publishes: see below.
    a.version = env.gru_v;
    a.lanes = h->lanes;
    // (lanes whose calls differ in batch size must agree on the layout: shares and placement from the pool's largest batch)
    const int Bl = (h->lanes > 1 && h->lane_max_b > B) ? h->lane_max_b : B;
    a.share = gru_share_of(h->lanes, Bl, gru_shared);
    a.xcd_rot = h->lanes > 1 ? (2 * Bl * h->lane) % 8 : 0;
    a.force_bmax = env.gru_bmax;
    if (env.gru_ts) a.tstamps = (long long*)(base + cap - (1u << 20));
    a.force_upw = env.gru_upw;
    a.poll_backoff = env.gru_backoff;
    // publishes: PLAIN stores by default inside a cluster that shares one XCD (the L2 is that XCD's point of coherence; the
    // rendezvous proves the placement at every launch), agent-scope (sc1) stores otherwise, on request
    // (ou_set_gru_publish_mode) and -- for good -- from the moment ou_check_device_status sees that a publish really was
    // invisible to the gather's agent-scope loads on this handle (status word 33; word 20 also counts members that were
    // merely late).  A hipGraph captured before such a switch keeps the publish form it was captured with: re-capture.
    a.agent_stores = env.gru_agent >= 0 ? (env.gru_agent != 0) : h->gru_agent_stores;
    a.dbg = env.gru_dbg;
    if (h->profile && h->prof_dev && h->prof_used < kProfSlots) {
      // the recurrence proper (the input projection is a conv launch of its own): 2 directions x T steps x (3H x H) MACs
      ou_handle::ProfRec rec;
      rec.flops = 2.0 * 2.0 * 3.0 * G.H * G.H * (double)in.T * B;
      rec.bytes = 4.0 * ((double)B * (6.0 + 2.0 + (res ? 2.0 : 0.0)) * G.H * in.T + 2.0 * 3.0 * G.H * G.H);
      rec.cfg = 1000 + in.T;  // 1000 + steps per pass
      a.prof = h->prof_dev + 32 * h->prof_used;
      h->prof.push_back(rec);
      h->prof_used++;
    }
    if (want_pre_gru) {
      pre_gru = next_event();
      if (pre_gru) chk(hipEventRecord(pre_gru, st), "pre-gru record");
    }
    chk(launch_gru(a, h->num_cu, st), G.name.c_str());
    mask(out);
    return out;
  }
};

// Persistent part of the workspace (lives across ou_condition / ou_score / ou_enhance calls on it).
struct Persist {
  unsigned* status;
  StepCoef* coef;             // [kMaxSteps] or [B]
  float* stats;               // [B][4]
  RowInfo* rows;              // [B]   ragged batch: per-row geometry
  int* lens;                  // [kMaxLenLevels][B]   ... and lengths on every level
  unsigned long long* xchg;   // GRU granules (conditioner)
  unsigned long long* xchg2;  // GRU granules (score net; may run concurrently with the conditioner)
  float* mel_scale;           // [B]
  float* g;                   // [kMaxSteps][D]
  float* film;                // [kMaxSteps][rows]
  Tensor mixn, x, wav;        // (B,1,T)
  std::vector<Tensor> sc;     // signal_cond_proj(cond_j)   (B, C_j, T_j)
  std::vector<Tensor> cond;   // conditions
  Tensor aux, latent;
};

Persist layout_persist(Runner& r, int T) {
  const Model& m = r.h->m;
  Persist P;
  // 64 status / diagnostics words + the fused ConvBlock kernel's barrier area (8 groups x 40 x 8 bytes)
  P.status = (unsigned*)r.alloc_raw(64 + 8 * 40 * 2);
  r.status_words = P.status;
  r.block3_bar = (unsigned long long*)(P.status + 64);
  int ncoef = kMaxSteps > r.B ? kMaxSteps : r.B;
  P.coef = (StepCoef*)r.alloc_raw((size_t)ncoef * 8);
  P.stats = r.alloc_raw((size_t)r.B * 4);
  P.rows = (RowInfo*)r.alloc_raw((size_t)r.B * 4);
  P.lens = (int*)r.alloc_raw((size_t)r.B * kMaxLenLevels);
  P.xchg = (unsigned long long*)r.alloc_raw(gru_granules(r.B, m.OC / 2) * 2);
  P.xchg2 = (unsigned long long*)r.alloc_raw(gru_granules(r.B, m.OC / 2) * 2);
  P.mel_scale = r.alloc_raw(r.B);
  P.g = r.alloc_raw((size_t)ncoef * m.film.D);
  P.film = r.alloc_raw((size_t)ncoef * m.film.rows);
  P.mixn = r.alloc("mixn", 1, T);
  P.x = r.alloc("x", 1, T);
  P.wav = r.alloc("wav", 1, T);
  for (int j = 0; j < m.n_blocks; j++) {
    const BlockL& b = m.c_dec[j];
    int Tj = T / m.tot_ds;
    // length at the output of decoder block j
    int up = 1;
    for (int k = 0; k <= j; k++) if (m.c_dec[k].dir == 2) up *= m.c_dec[k].rate;
    Tj *= up;
    P.cond.push_back(r.alloc("cond.c" + std::to_string(j), b.C, Tj));
    P.sc.push_back(r.alloc("cond.sc" + std::to_string(j), b.C, Tj));
  }
  P.aux = r.alloc("cond.aux", m.C0, T);
  P.latent = r.alloc("cond.latent", m.OC, T / m.tot_ds);
  return P;
}

// ConditionerNetwork.forward(train=True)  condition.py:346-377
void run_condition(Runner& r, Persist& P, const float* mix_norm, int T) {
  const Model& m = r.h->m;
  const int L = T / m.tot_ds;
  const int n = m.n_levels - 1;
  hipStream_t main = r.st;
  const bool ov = r.h->overlap && !r.dry && r.h->lanes <= 1;
  // --- MelAdapter  condition.py:110-114 (independent of the encoder chain: side stream 0)
  if (ov) { r.fork(main, 0); r.st = r.h->aux[0]; }
  Tensor mel = r.alloc("cond.mel", m.mel.n_mels, L);
  float* esum = r.alloc_raw((size_t)r.B * L);
  if (!r.dry && r.ok()) {
    r.chk(launch_mel(mix_norm, r.W(m.mel.win_off), r.W(m.mel.tw_off), r.W(m.mel.fb_off), mel.p, esum, r.B, T,
                     m.mel.n_fft, m.mel.hop, m.mel.pad_left, m.mel.n_freq, m.mel.n_mels, L, r.st), "mel");
    r.chk(launch_mel_scale(esum, P.mel_scale, r.B, L, r.st, r.lens_of(L)), "mel_scale");
    r.mask(mel);  // (frames behind a row's end still see its last samples)
  }
  Runner::Epi em;
  em.in_scale = P.mel_scale;  // the global mel normalisation is linear: folded into the conv's input scale
  em.act = false;
  Tensor m0 = r.conv(m.c_melconv, mel, "cond.melconv", em);
  Tensor x_mel = r.block(m.c_melblock, m0, "cond.melblock", nullptr, 0, nullptr, nullptr).v;
  r.st = main;
  // --- input conv + encoder  condition.py:360, 189-206
  Tensor e0 = r.alloc("cond.in", m.C0, T);
  {
    const int* ln = r.env.mask_fused ? r.lens_of(T) : nullptr;
    if (!r.dry && r.ok())
      r.chk(launch_in_conv(mix_norm, r.W(m.c_in.w_off), r.W(m.c_in.b_off), nullptr, 0, e0.p, r.B, m.C0, T, m.c_in.KW, r.st, ln), "cond.in");
    if (!ln) r.mask(e0);
  }
  Tensor hcur = e0;
  std::vector<Tensor> outs;
  for (int i = 0; i < m.n_blocks; i++) {
    auto bo = r.block(m.c_enc[i], hcur, "cond.enc" + std::to_string(i), nullptr, 0, nullptr, nullptr);
    if (i < n - 1) {
      // strided "st" conv of this block's output: off the critical path (side stream 1)
      if (ov) { r.fork(main, 1); r.st = r.h->aux[1]; }
      const ConvL& S = m.c_st[i];
      const int R = S.rate, C = S.Cin / R;
      Tensor sd = r.alloc("cond.s2d" + std::to_string(i), S.Cin, bo.v.T / R);
      if (!r.dry && r.ok()) r.chk(launch_s2d(bo.v.p, r.W(S.a_off), sd.p, r.B, C, bo.v.T, R, r.st), "s2d");
      Runner::Epi es;
      es.act = false;  // PReLU already applied by the space-to-depth pass
      outs.push_back(r.conv(S, sd, "cond.st" + std::to_string(i), es));
      r.st = main;
    }
    hcur = bo.h_next;
  }
  outs.push_back(hcur);
  if (ov) { r.join(0, main); r.join(1, main); }
  Tensor sum = r.alloc("cond.enc_sum", m.OC, L);
  if (!r.dry && r.ok()) {
    const float* q[4] = {nullptr, nullptr, nullptr, nullptr};
    if (outs.size() > 4) { r.herr = hipErrorInvalidValue; r.where = "too many encoder outputs"; return; }
    for (size_t i = 0; i < outs.size(); i++) q[i] = outs[i].p;
    r.chk(launch_sum(x_mel.p, q[0], q[1], q[2], q[3], 1.0f / std::sqrt((float)(outs.size() + 1)), sum.p,
                     (size_t)r.B * m.OC * L, r.st), "enc_sum");
  }
  // --- conv_block1 -> 2-layer GRU (+residual) -> conv_block2   condition.py:208-216
  Tensor cb1 = r.block(m.c_cb1, sum, "cond.cb1", nullptr, 0, nullptr, nullptr).v;
  // status block: [0] error word, [2..3] / [4..5] = {tag epoch, finished-block count} of the two GRU exchange areas
  Tensor g0 = r.gru(m.c_gru0, cb1, "cond.gru0", P.xchg, P.status, P.status + 2, nullptr, 1.f);
  const bool gres = m.cfg.cond.encoder_gru_residual != 0;
  Tensor g1 = r.gru(m.c_gru1, g0, "cond.gru", P.xchg, P.status, P.status + 2, gres ? cb1.p : nullptr, kInvSqrt2);
  Tensor lat = r.block(m.c_cb2, g1, "cond.cb2", nullptr, 0, nullptr, nullptr, false, nullptr, &P.latent).v;
  // --- decoder  condition.py:264-270
  Tensor y = r.block(m.c_decin, lat, "cond.decin", nullptr, 0, nullptr, nullptr).v;
  for (int j = 0; j < m.n_blocks; j++) {
    // condition j = conv1 output of decoder block j (condition.py:264-270); the last block's output is the aux signal
    auto bo = r.block(m.c_dec[j], y, "cond.dec" + std::to_string(j), nullptr, 0, nullptr, nullptr, true, &P.cond[j],
                      j == m.n_blocks - 1 ? &P.aux : nullptr);
    y = bo.v;
    // score.py:208  sc = signal_cond_proj_j(cond_j): independent of x and sigma -> computed once here
    {
      Runner::Epi es;
      es.act = false;
      r.conv(m.s_sig[j], bo.c1, "cond.sc" + std::to_string(j), es, &P.sc[j]);
    }
  }
}

// ScoreNetwork.forward + EDM wrapper + sampler update for the coefficient rows at `coef`
//   film_row: pointer to this step's FiLM table row(s); film_bs / coef_bs: per-batch strides (0 = shared)
// Split in two halves: the encoder + GRU half does not depend on the conditioner (cond enters the decoder only),
// so the first step's encoder can run concurrently with it.
struct ScoreEnc {
  std::vector<Tensor> residuals;
  Tensor hg;
  bool fuse_res = false;
};
ScoreEnc run_score_enc(Runner& r, Persist& P, const float* x, const StepCoef* coef, int coef_bs,
                       const float* film_row, int film_bs, int T) {
  const Model& m = r.h->m;
  ScoreEnc E;
  Tensor e0 = r.alloc("score.in", m.C0, T);
  // the w_in scaling of the EDM wrapper (universe.py:199,202) rides on the input conv
  {
    const int* ln = r.env.mask_fused ? r.lens_of(T) : nullptr;
    if (!r.dry && r.ok())
      r.chk(launch_in_conv(x, r.W(m.s_in.w_off), r.W(m.s_in.b_off), coef, coef_bs, e0.p, r.B, m.C0, T, m.s_in.KW, r.st, ln),
            "score.in");
    if (!ln) r.mask(e0);
  }
  Tensor hcur = e0;
  for (int i = 0; i < m.n_blocks; i++) {
    const float* fr = film_row ? film_row + m.film.enc_off[i] : nullptr;
    auto bo = r.block(m.s_enc[i], hcur, "score.enc" + std::to_string(i), fr, film_bs, nullptr, nullptr);
    E.residuals.push_back(bo.v);
    hcur = bo.h_next;
  }
  // GRU bottleneck; when decoder block 0 has no rate change its residual add (blocks.py:374-376) is fused here
  E.fuse_res = m.s_dec[0].dir == 0;
  E.hg = r.gru(m.s_gru, hcur, "score.gru", P.xchg2, P.status, P.status + 4,
               E.fuse_res && !r.dry ? E.residuals[m.n_blocks - 1].p : nullptr, kInvSqrt2);
  return E;
}
void run_score_dec(Runner& r, Persist& P, const ScoreEnc& E, const float* x, const float* noise, float* out, int mode,
                   const StepCoef* coef, int coef_bs, const float* film_row, int film_bs, int T, const Tensor* dec0_done = nullptr) {
  const Model& m = r.h->m;
  Tensor y = E.hg;
  for (int j = 0; j < m.n_blocks; j++) {
    if (j == 0 && dec0_done) { y = *dec0_done; continue; }
    const float* fr = film_row ? film_row + m.film.dec_off[j] : nullptr;
    const Tensor& res = E.residuals[m.n_blocks - 1 - j];
    const float* resp = (j == 0 && E.fuse_res) ? nullptr : res.p;
    auto bo = r.block(m.s_dec[j], y, "score.dec" + std::to_string(j), fr, film_bs, P.sc[j].p, resp);
    y = bo.v;
  }
  const int* ln = r.env.mask_fused ? r.lens_of(T) : nullptr;
  if (!r.dry && r.ok())
    r.chk(launch_out_conv(y.p, r.W(m.s_out.w_off), r.W(m.s_out.b_off), r.W(m.s_out.a_off), x, noise, out, coef,
                          coef_bs, m.cfg.has_edm, mode, r.B, m.C0, T, m.s_out.KW, r.st, ln), "score.out");
  if (!ln) r.mask(out, 1, T);
}
static bool m_blocks_ok(const Runner& r) { return r.h->m.n_blocks >= 1 && r.h->m.s_dec[0].dir == 0; }
void run_score(Runner& r, Persist& P, const float* x, const float* noise, float* out, int mode,
               const StepCoef* coef, int coef_bs, const float* film_row, int film_bs, int T) {
  // OU_DBG_DEC0=1 -- MEASUREMENT ONLY, RESULTS INVALID: the first decoder block is launched on a side stream that waits for
  // what precedes the GRU launch instead of the GRU itself, i.e. its three convs run UNDER the recurrence (on whatever that has
  // written so far).  The time of a forward in this mode is a lower bound for any scheme that gates those convs on the
  // recurrence's progress: the gated version can only start later and wait more.
  const bool dec0_under = r.env.dbg_dec0_under_gru != 0 && !r.dry && r.h->overlap && r.h->lanes <= 1 && m_blocks_ok(r);
  r.want_pre_gru = dec0_under;
  ScoreEnc E = run_score_enc(r, P, x, coef, coef_bs, film_row, film_bs, T);
  r.want_pre_gru = false;
  if (dec0_under && r.pre_gru && r.ok()) {
    const Model& m = r.h->m;
    hipStream_t main = r.st;
    r.chk(hipStreamWaitEvent(r.h->aux[0], r.pre_gru, 0), "dec0 wait");
    r.st = r.h->aux[0];
    const float* fr = film_row ? film_row + m.film.dec_off[0] : nullptr;
    const Tensor& res = E.residuals[m.n_blocks - 1];
    auto bo = r.block(m.s_dec[0], E.hg, "score.dec0", fr, film_bs, P.sc[0].p, E.fuse_res ? nullptr : res.p);
    r.st = main;
    r.join(0, main);
    Tensor done = bo.v;
    run_score_dec(r, P, E, x, noise, out, mode, coef, coef_bs, film_row, film_bs, T, &done);
    return;
  }
  run_score_dec(r, P, E, x, noise, out, mode, coef, coef_bs, film_row, film_bs, T);
}

// universe.py:175-189, 197-209, 333-343 scalars for one sigma, computed in fp32 like the reference's tensors
StepCoef make_coef(const ou_config& cfg, float s, bool last, double eta, double beta, float s_next) {
  StepCoef c;
  const float s2 = s * s;
  if (cfg.has_edm) {
    // universe.py:176-178: sigma_data from edm.data_level_db, else from normalization_kwargs.level_db
    const double sd = std::pow(10.0, (double)(cfg.has_edm_data_level ? cfg.edm_data_level_db : cfg.level_db) / 20.0);
    const float sd2 = (float)(sd * sd);
    const float sn2 = s2 + sd2;
    const float sn = std::sqrt(sn2);
    c.w_skip = sd2 / sn2;
    c.w_in = 1.0f / sn;
    c.w_out = (s * (float)sd) / sn;
    c.sigma_net = cfg.edm_noise * s;
  } else {
    c.w_skip = 0.f; c.w_in = 1.f; c.w_out = 1.f; c.sigma_net = s;
  }
  c.sig2 = s2;
  c.c1 = last ? s2 : s2 * (float)eta;
  c.s_next = s_next;
  c.beta = (float)beta;
  return c;
}

void schedule(const ou_config& cfg, int n_steps, double epsilon, float* sigma, double* eta, double* beta) {
  const double ratio = (double)cfg.sigma_max / (double)cfg.sigma_min;
  const double delta_t = 1.0 / (n_steps - 1);
  const double gamma = std::pow(ratio, -delta_t);
  *eta = 1.0 - std::pow(gamma, epsilon);
  *beta = std::sqrt(1.0 - std::pow(gamma, 2.0 * (epsilon - 1.0)));
  // torch.linspace(0, 1, N) (fp32, symmetric evaluation) flipped, then s_min * (s_max/s_min) ** time
  const float step = 1.0f / (float)(n_steps - 1);
  for (int n = 0; n < n_steps; n++) {
    int i = n_steps - 1 - n;
    float t = (i < n_steps / 2) ? (float)i * step : 1.0f - (float)(n_steps - 1 - i) * step;
    sigma[n] = (float)cfg.sigma_min * std::pow((float)ratio, t);  // fp32 pow like torch.pow(Scalar, fp32 Tensor)
  }
}

void upload_coefs(Runner& r, StepCoef* dst, const std::vector<StepCoef>& rows) {
  for (size_t i = 0; i < rows.size(); i += 64) {
    CoefBlock blk;
    int n = (int)std::min<size_t>(64, rows.size() - i);
    for (int k = 0; k < n; k++) blk.c[k] = rows[i + k];
    r.chk(launch_upload_coef(dst + i, blk, n, r.st), "upload coef");
  }
}

int finish(ou_handle* h, Runner& r) {
  if (r.oom) return fail(h, OU_ENOMEM, "workspace too small: need " + std::to_string(r.off) + " bytes");
  if (r.herr != hipSuccess)
    return fail(h, OU_EHIP, std::string("HIP error at ") + r.where + ": " + hipGetErrorString(r.herr));
  return OU_OK;
}

}  // namespace

// ======================================================================================================
extern "C" {

const char* ou_version(void) {
#ifdef OU_EXPERIMENTS
  return "libouniverse 0.2 (gfx950, fp32 MFMA) +experiments";
#else
  return "libouniverse 0.2 (gfx950, fp32 MFMA)";
#endif
}
const char* ou_last_error(const ou_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }
const char* ou_packer_last_error(const ou_packer* p) { return p ? p->err.c_str() : g_last_error.c_str(); }

int ou_packer_create(const ou_config* cfg, ou_packer** out) {
  if (!cfg || !out) return fail(nullptr, OU_EINVAL, "null argument");
  auto* p = new ou_packer();
  std::string e = build_model(*cfg, p->m);
  if (!e.empty()) { delete p; return fail(nullptr, OU_ENOTIMPL, e); }
  *out = p;
  return OU_OK;
}

int ou_packer_set(ou_packer* p, const char* key, const float* data, const int64_t* shape, int32_t ndim) {
  if (!p || !key || !data || ndim < 0 || ndim > 8) return fail(nullptr, OU_EINVAL, "bad argument to ou_packer_set");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; i++) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  p->sd[key] = std::move(t);
  return OU_OK;
}

int ou_packer_finish(ou_packer* p, const float** blob_host, size_t* nbytes) {
  if (!p || !blob_host || !nbytes) return fail(nullptr, OU_EINVAL, "null argument");
  int code = OU_OK;
  std::string e = pack_weights(p->m, p->sd, p->blob, code);
  if (!e.empty()) { p->err = e; g_last_error = e; return code; }
  *blob_host = p->blob.data();
  *nbytes = p->blob.size() * sizeof(float);
  return OU_OK;
}

void ou_packer_destroy(ou_packer* p) { delete p; }

int ou_packed_bytes(const ou_config* cfg, size_t* nbytes) {
  if (!cfg || !nbytes) return fail(nullptr, OU_EINVAL, "null argument");
  Model m;
  std::string e = build_model(*cfg, m);
  if (!e.empty()) return fail(nullptr, OU_ENOTIMPL, e);
  *nbytes = m.total_floats * sizeof(float);
  return OU_OK;
}

const char* ou_packer_plan_json(const ou_packer* p) { return p ? p->m.json.c_str() : ""; }
const char* ou_plan_json(const ou_handle* hc) {
  if (!hc) return "";
  ou_handle* h = const_cast<ou_handle*>(hc);
  // the plan of the packed layers + the current values of the handle's options (ou_set_option)
  std::string o = "\"options\": {";
  for (int i = 0; i < kNumOptions; i++) {
    const OptDesc& d = kOptions[i];
    char buf[96];
    if (d.ip) std::snprintf(buf, sizeof buf, "%s\"%s\": %d", i ? ", " : "", d.key, h->opt.*(d.ip));
    else std::snprintf(buf, sizeof buf, "%s\"%s\": %.9g", i ? ", " : "", d.key, h->opt.*(d.dp));
    o += buf;
  }
  o += "}";
  const std::string& j = h->m.json;
  const size_t close = j.rfind('}');
  h->plan_with_options = close == std::string::npos ? "{" + o + "}" : j.substr(0, close) + ", " + o + j.substr(close);
  return h->plan_with_options.c_str();
}

int ou_set_option(ou_handle* h, const char* key, double value) {
  if (!h || !key) return fail(h, OU_EINVAL, "bad argument");
  const OptDesc* d = find_option(key);
  if (!d) return fail(h, OU_EMISSING, std::string("ou_set_option: no such option: ") + key);
#ifndef OU_EXPERIMENTS
  if (d->experiments_only && value != 0.0)
    return fail(h, OU_ENOTIMPL, std::string("ou_set_option: `") + key + "` makes calls return wrong results by design and exists in "
                "the experiments build (make EXPERIMENTS=1) only");
#endif
  if (d->ip) {
    if (value != std::floor(value) || std::fabs(value) > 2e9) return fail(h, OU_EINVAL, std::string("ou_set_option: `") + key + "` takes an integer");
    h->opt.*(d->ip) = (int)value;
  } else {
    h->opt.*(d->dp) = value;
  }
  h->trace = h->opt.trace != 0;
  h->overlap = h->opt.no_overlap == 0;
  return OU_OK;
}

int ou_get_option(const ou_handle* h, const char* key, double* value) {
  if (!h || !key || !value) return fail(const_cast<ou_handle*>(h), OU_EINVAL, "bad argument");
  const OptDesc* d = find_option(key);
  if (!d) return fail(const_cast<ou_handle*>(h), OU_EMISSING, std::string("ou_get_option: no such option: ") + key);
  *value = d->ip ? (double)(h->opt.*(d->ip)) : h->opt.*(d->dp);
  return OU_OK;
}

int ou_reset_options(ou_handle* h) {
  if (!h) return fail(h, OU_EINVAL, "bad argument");
  const std::string keep = h->opt.chain_ts;
  h->opt = Options();
  h->opt.chain_ts = keep;
  h->trace = false;
  h->overlap = true;
  return OU_OK;
}

int ou_option_count(void) { return kNumOptions; }
const char* ou_option_name(int32_t i) { return i >= 0 && i < kNumOptions ? kOptions[i].key : nullptr; }
const char* ou_option_doc(int32_t i) { return i >= 0 && i < kNumOptions ? kOptions[i].doc : nullptr; }
double ou_option_default(int32_t i) {
  if (i < 0 || i >= kNumOptions) return 0.0;
  const Options d;
  return kOptions[i].ip ? (double)(d.*(kOptions[i].ip)) : d.*(kOptions[i].dp);
}

int ou_set_stamp_layer(ou_handle* h, const char* block_name) {
  if (!h) return fail(h, OU_EINVAL, "bad argument");
  h->opt.chain_ts = block_name ? block_name : "";
  return OU_OK;
}

int ou_create(const ou_config* cfg, const void* weights_dev, size_t nbytes, int32_t device, ou_handle** out) {
  if (!cfg || !weights_dev || !out) return fail(nullptr, OU_EINVAL, "null argument");
  auto* h = new ou_handle();
  std::string e = build_model(*cfg, h->m);
  if (!e.empty()) { delete h; return fail(nullptr, OU_ENOTIMPL, e); }
  if (nbytes != h->m.total_floats * sizeof(float)) {
    size_t want = h->m.total_floats * sizeof(float);
    delete h;
    return fail(nullptr, OU_ESHAPE, "packed weight blob has " + std::to_string(nbytes) + " bytes, expected " + std::to_string(want));
  }
  int ndev = 0;
  hipError_t he = hipGetDeviceCount(&ndev);
  if (he != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    delete h;
    return fail(nullptr, OU_EHIP, "no HIP device available for ou_create (this library has no CPU path)");
  }
  hipDeviceProp_t prop;
  he = hipGetDeviceProperties(&prop, device);
  if (he != hipSuccess) { delete h; return fail(nullptr, OU_EHIP, hipGetErrorString(he)); }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::string arch = prop.gcnArchName;
    delete h;
    return fail(nullptr, OU_EHIP, "libouniverse is built for gfx950 (MI355X) only; device is " + arch);
  }
  h->num_cu = prop.multiProcessorCount;
  h->trace = false;
  h->device = device;
  h->W = (const float*)weights_dev;
  (void)hipSetDevice(device);
  {  // host copies of the conv PReLU slopes: passed to the kernels by value
    std::vector<const ConvL*> all;
    auto addb = [&](const BlockL& b) { if (b.dir) all.push_back(&b.rc); all.push_back(&b.c1); all.push_back(&b.c2); all.push_back(&b.c3); };
    const Model& m = h->m;
    for (auto& b : m.s_enc) addb(b);
    for (auto& b : m.s_dec) addb(b);
    addb(m.c_melblock);
    for (auto& b : m.c_enc) addb(b);
    for (auto& l : m.c_st) all.push_back(&l);
    addb(m.c_cb1); addb(m.c_cb2); addb(m.c_decin);
    for (auto& b : m.c_dec) addb(b);
    for (auto* l : all) {
      if (!l->act) continue;
      float v = 0.f;
      he = hipMemcpy(&v, h->W + l->a_off, sizeof(float), hipMemcpyDeviceToHost);
      if (he != hipSuccess) { delete h; return fail(nullptr, OU_EHIP, hipGetErrorString(he)); }
      h->alphas[l->a_off] = v;
    }
  }
  he = init_conv_kernels();
  if (he != hipSuccess) { delete h; return fail(nullptr, OU_EHIP, hipGetErrorString(he)); }
  for (int i = 0; i < 3; i++) {
    he = hipStreamCreateWithFlags(&h->aux[i], hipStreamNonBlocking);
    if (he != hipSuccess) { delete h; return fail(nullptr, OU_EHIP, hipGetErrorString(he)); }
  }
  h->overlap = true;
  *out = h;
  return OU_OK;
}

void ou_destroy(ou_handle* h) {
  if (!h) return;
  if (h->prof_dev) (void)hipFree(h->prof_dev);
  for (auto& e : h->events) (void)hipEventDestroy(e);
  for (int i = 0; i < 3; i++) if (h->aux[i]) (void)hipStreamDestroy(h->aux[i]);
  delete h;
}

int ou_workspace_bytes(const ou_handle* hc, int32_t B, int32_t T, size_t* nbytes) {
  ou_handle* h = const_cast<ou_handle*>(hc);
  if (!h || !nbytes || B < 1 || T < 1) return fail(h, OU_EINVAL, "bad argument");
  if (T % h->m.tot_ds) return fail(h, OU_EINVAL, "T must be a multiple of the total down-sampling factor");
  auto saved = h->tensors;
  Runner r(h, nullptr, 0, true, nullptr, B);
  Persist P = layout_persist(r, T);
  run_condition(r, P, nullptr, T);
  size_t mark = r.off;
  run_score(r, P, nullptr, nullptr, nullptr, OUT_UPDATE, nullptr, 0, nullptr, 0, T);
  (void)mark;
  // aux_to_wav scratch (2x up-sampled aux signal)
  r.alloc_raw((size_t)B * h->m.C0 * 2 * T);
  h->tensors = saved;
  *nbytes = r.off + 4096;
  return OU_OK;
}

int ou_schedule(const ou_config* cfg, int32_t n_steps, double epsilon, float* sigma_out, double* eta, double* beta) {
  if (!cfg || !sigma_out || !eta || !beta || n_steps < 2) return fail(nullptr, OU_EINVAL, "bad argument");
  schedule(*cfg, n_steps, epsilon, sigma_out, eta, beta);
  return OU_OK;
}

int ou_condition(ou_handle* h, const float* mix_norm, int32_t B, int32_t T, void* ws, size_t ws_bytes, ou_stream_t stream) {
  if (!h || !mix_norm || !ws || B < 1) return fail(h, OU_EINVAL, "bad argument");
  if (T % h->m.tot_ds || T <= 0) return fail(h, OU_EINVAL, "T must be a positive multiple of the total down-sampling factor");
  if (!h->ws_ok(ws, ws_bytes, B, T))
    return fail(h, OU_EINVAL, "workspace was not prepared by ou_workspace_init for this (B, T)");
  h->tensors.clear();
  h->n_launch = h->n_conv = 0;
  h->ev_used = 0;
  Runner r(h, ws, ws_bytes, false, (hipStream_t)stream, B);
  Persist P = layout_persist(r, T);
  if (r.oom) return finish(h, r);
  run_condition(r, P, mix_norm, T);
  h->cond_B = B;
  h->cond_T = T;
  return finish(h, r);
}

int ou_score(ou_handle* h, const float* x, const float* sigma_host, float* score_out, int32_t B, int32_t T, void* ws,
             size_t ws_bytes, ou_stream_t stream) {
  if (!h || !x || !sigma_host || !score_out || !ws) return fail(h, OU_EINVAL, "bad argument");
  if (B != h->cond_B || T != h->cond_T) return fail(h, OU_EINVAL, "ou_score: call ou_condition with the same (B, T) first");
  if (!h->ws_ok(ws, ws_bytes, B, T))
    return fail(h, OU_EINVAL, "workspace was not prepared by ou_workspace_init for this (B, T)");
  h->n_launch = h->n_conv = 0;
  Runner r(h, ws, ws_bytes, false, (hipStream_t)stream, B);
  auto keep = h->tensors;
  Persist P = layout_persist(r, T);
  {  // skip over the conditioner's region so that its intermediates stay inspectable
    Runner d(h, nullptr, 0, true, nullptr, B);
    Persist Pd = layout_persist(d, T);
    run_condition(d, Pd, nullptr, T);
    r.off = d.off;
  }
  for (auto& kv : keep) if (kv.first.rfind("cond.", 0) == 0) h->tensors[kv.first] = kv.second;
  std::vector<StepCoef> rows;
  for (int b = 0; b < B; b++) {
    if (!(sigma_host[b] > 0.f)) return fail(h, OU_EINVAL, "sigma must be positive");
    rows.push_back(make_coef(h->m.cfg, sigma_host[b], true, 0.0, 0.0, 0.f));
  }
  upload_coefs(r, P.coef, rows);
  const Model& m = h->m;
  r.chk(launch_sigma_embed(P.coef, B, r.W(m.sigma.p_off), m.sigma.simple, m.sigma.n_rff, m.film.D, P.g, r.st), "sigma");
  r.chk(launch_film(P.g, r.W(m.film.w_off), r.W(m.film.b_off), P.film, B, m.film.rows, m.film.D, r.st), "film");
  run_score(r, P, x, nullptr, score_out, OUT_SCORE, P.coef, 1, P.film, m.film.rows, T);
  return finish(h, r);
}

int ou_aux_to_wav(ou_handle* h, float* wav_out, int32_t B, int32_t T, void* ws, size_t ws_bytes, ou_stream_t stream) {
  if (!h || !wav_out || !ws) return fail(h, OU_EINVAL, "bad argument");
  if (B != h->cond_B || T != h->cond_T) return fail(h, OU_EINVAL, "ou_aux_to_wav: call ou_condition with the same (B, T) first");
  const Model& m = h->m;
  if (!m.dec.present) return fail(h, OU_ENOTIMPL, "model has no signal decoupling layer (aux signal is multi-channel)");
  if (m.dec.act != OU_ACT_SNAKE) return fail(h, OU_ENOTIMPL, "only the snake signal-decoupling activation is implemented");
  Runner r(h, ws, ws_bytes, false, (hipStream_t)stream, B);
  auto keep = h->tensors;
  Persist P = layout_persist(r, T);
  h->tensors = keep;
  // scratch at the very end of the workspace
  size_t need = (size_t)B * m.C0 * 2 * T * 4;
  if (ws_bytes < need + r.off) return fail(h, OU_ENOMEM, "workspace too small");
  float* tmp = (float*)((char*)ws + ((ws_bytes - need) & ~size_t(255)));
  r.chk(launch_decoupling(P.aux.p, r.W(m.dec.alpha_off), r.W(m.dec.up_off), r.W(m.dec.down_off), r.W(m.dec.conv.w_off),
                          r.W(m.dec.conv.b_off), tmp, wav_out, B, m.C0, T, r.st), "decoupling");
  return finish(h, r);
}

}  // extern "C"

namespace {
// ou_enhance / ou_enhance_var.  `t_raw`: host array of B row lengths (max = T_raw) or null (every row T_raw samples long).
int enhance_impl(ou_handle* h, const float* mix, float* out, const float* noise, int32_t B, int32_t T_raw,
                 const int32_t* t_raw, int32_t n_steps, double epsilon, const float* sigma_host, int32_t warm_start,
                 uint32_t flags, void* ws, size_t ws_bytes, ou_stream_t stream) {
  if (!h || !mix || !out || !ws || B < 1 || T_raw < 1) return fail(h, OU_EINVAL, "bad argument");
  if (t_raw) {
    int mx = 0;
    bool all_whole = true;
    for (int b = 0; b < B; b++) {
      if (t_raw[b] < 1 || t_raw[b] > T_raw) return fail(h, OU_EINVAL, "ou_enhance_var: 1 <= t_raw[b] <= T_raw_max");
      mx = t_raw[b] > mx ? t_raw[b] : mx;
      all_whole = all_whole && t_raw[b] == T_raw;
    }
    if (mx != T_raw) return fail(h, OU_EINVAL, "ou_enhance_var: T_raw_max must be the length of the longest row");
    if (all_whole) t_raw = nullptr;  // nothing ragged about this batch: the plain path (and its fused kernels)
  }
  const bool use_aux = (flags & OU_ENH_USE_AUX_SIGNAL) != 0;
  const bool saved_overlap = h->overlap;
  struct OverlapGuard { ou_handle* h; bool v; ~OverlapGuard() { h->overlap = v; } } overlap_guard{h, saved_overlap};
  // Side streams inside the call only when this is the one call on the device.  With several lanes (ou_set_lanes) every lane
  // is ONE chain on its caller's stream: HIP multiplexes its streams onto a handful of hardware queues, and four lanes with
  // four streams each alias there -- measured: 4 lanes 122 utt/s with side streams (87 % of the time ONE kernel on the
  // device), 209 utt/s as four chains (serial loop: 132).
  if ((flags & OU_ENH_SERIAL) || h->lanes > 1) h->overlap = false;
  if (!use_aux && !noise) return fail(h, OU_EINVAL, "noise must be given");
  if (n_steps < 2 || n_steps > kMaxSteps) return fail(h, OU_EINVAL, "n_steps must be in [2, 256]");
  if (warm_start >= n_steps) return fail(h, OU_EINVAL, "warm_start must be < n_steps");
  const Model& m = h->m;
  const int tot = m.tot_ds;
  const int pad = tot - T_raw % tot;  // universe.py:219-223 (a full block when already a multiple)
  const int pad_left = pad / 2;
  const int T = T_raw + pad;
  if (!h->ws_ok(ws, ws_bytes, B, T))
    return fail(h, OU_EINVAL, "workspace was not prepared by ou_workspace_init for this (B, T_raw + pad)");
  h->tensors.clear();
  h->n_launch = h->n_conv = 0;
  h->ev_used = 0;
  hipStream_t st = (hipStream_t)stream;
  Runner r(h, ws, ws_bytes, false, st, B);
  Persist P = layout_persist(r, T);
  if (r.oom) return finish(h, r);

  std::vector<float> sigma(n_steps);
  double eta, beta;
  schedule(m.cfg, n_steps, epsilon, sigma.data(), &eta, &beta);
  if (sigma_host) std::memcpy(sigma.data(), sigma_host, sizeof(float) * n_steps);
  std::vector<StepCoef> rows;
  for (int n = 0; n < n_steps; n++)
    rows.push_back(make_coef(m.cfg, sigma[n], n == n_steps - 1, eta, beta, n + 1 < n_steps ? sigma[n + 1] : 0.f));
  upload_coefs(r, P.coef, rows);

  if (t_raw) {
    // per-row geometry and the rows' lengths on every level of the network: T, 2 T (the decoupling layer's up-sampled grid)
    // and T / (r_0 .. r_i).  By value through kernel arguments: capturable, no host memory involved.
    LevelSpec& lv = r.lv;
    lv.n = 0;
    auto add_level = [&](int num, int den) {
      lv.num[lv.n] = num; lv.den[lv.n] = den; r.level_T[lv.n] = (int)((long long)T * num / den); lv.n++;
    };
    add_level(1, 1);
    add_level(2, 1);
    int cum = 1;
    for (int i = 0; i < m.cfg.score.n_rates && lv.n < kMaxLenLevels; i++) { cum *= m.cfg.score.rate_factors[i]; add_level(1, cum); }
    if (cum != tot) return fail(h, OU_EINVAL, "internal: rate factors do not multiply to the total down-sampling factor");
    for (int off = 0; off < B; off += 64) {
      RowBlock blk;
      const int n = B - off < 64 ? B - off : 64;
      for (int i = 0; i < 64; i++) blk.t_raw[i] = i < n ? t_raw[off + i] : 1;
      r.chk(launch_upload_rows(P.rows, P.lens, blk, n, off, B, tot, lv, st), "upload rows");
    }
    r.ragged = true;
    r.lens_dev = P.lens;
    r.rows_dev = P.rows;
  }
  const float level = (float)std::pow(10.0, (double)m.cfg.level_db / 20.0);
  // (launched behind the fork of the first score-encoder pass when there is one: the side stream's first kernel starts an event
  // latency -- ~10 us -- after the fork, and this launch is what the device has to do in the meantime)
  auto normalize = [&]() {
    if (r.ragged) r.chk(launch_pad_normalize_var(mix, P.mixn.p, P.stats, P.rows, B, T_raw, T, level, st), "normalize");
    else r.chk(launch_pad_normalize(mix, P.mixn.p, P.stats, B, T_raw, T, pad_left, level, st), "normalize");
  };
  bool normalized = false;
  const size_t nBT = (size_t)B * T;
  const int keep_rms = (flags & OU_ENH_KEEP_RMS) ? 1 : 0;
  const int peak = (flags & OU_ENH_NO_PEAK_GUARD) ? 0 : 1;
  const bool need_wav = use_aux || warm_start >= 0;
  if (need_wav && (!m.dec.present || m.dec.act != OU_ACT_SNAKE))
    return fail(h, OU_ENOTIMPL, "aux_to_wav needs the snake signal-decoupling layer (UNIVERSE++)");
  const int n_start = warm_start >= 0 ? warm_start : 0;

  // Where the conditioner's scratch ends (= where the per-step score scratch starts): layout is a pure function of
  // (config, B, T), so a dry walk gives it before anything is launched.
  size_t mark;
  {
    auto keep = h->tensors;
    Runner d(h, nullptr, 0, true, nullptr, B);
    Persist Pd = layout_persist(d, T);
    run_condition(d, Pd, nullptr, T);
    mark = d.off;
    h->tensors = keep;
  }

  // The first score-encoder pass (+ its GRU) does not depend on the conditioner: run it on side stream 2 while
  // the conditioner runs on the caller's stream (at batch 1 most CUs idle during the GRU passes of either).
  ScoreEnc E0;
  bool have_e0 = false;
  size_t off_after_enc = 0;
  if (!use_aux) {
    r.chk(launch_sigma_embed(P.coef, n_steps, r.W(m.sigma.p_off), m.sigma.simple, m.sigma.n_rff, m.film.D, P.g, st), "sigma");
    r.chk(launch_film(P.g, r.W(m.film.w_off), r.W(m.film.b_off), P.film, n_steps, m.film.rows, m.film.D, st), "film");
    // (only when two GRU layers fit on the machine side by side: their clusters spin on each other's publishes and must
    // all be resident)
    const int share2 = Runner::gru_share_of(h->lanes, B, true);
    const bool gru_fit = gru_ring_batch_cap(m.s_gru.H, h->num_cu, share2, 0, B, h->lanes) >= 1 &&
                         gru_ring_batch_cap(m.c_gru0.H, h->num_cu, share2, 0, B, h->lanes) >= 1;
    if (warm_start < 0 && h->overlap && gru_fit) {
      r.gru_shared = true;
      r.chk(launch_init_x(noise, nullptr, sigma[n_start], P.x.p, nBT, st), "init x");  // universe.py:325-327
      r.mask(P.x);
      const size_t save = r.off;
      r.fork(st, 2);
      normalize();
      normalized = true;
      r.st = h->aux[2];
      r.off = mark;
      E0 = run_score_enc(r, P, P.x.p, P.coef + n_start, 0, P.film + (size_t)n_start * m.film.rows, 0, T);
      off_after_enc = r.off;
      r.st = st;
      r.off = save;
      have_e0 = true;
    }
  }
  if (!normalized) normalize();
  run_condition(r, P, P.mixn.p, T);
  r.gru_shared = false;
  if (!r.dry && r.ok() && r.off != mark) return fail(h, OU_EINVAL, "internal: workspace layout mismatch");
  h->cond_B = r.ragged ? 0 : B;  // (the operator seams ou_score / ou_aux_to_wav take whole batches only)
  h->cond_T = T;

  if (need_wav) {
    float* tmp = r.alloc_raw((size_t)B * m.C0 * 2 * T);
    if (r.ok())
      r.chk(launch_decoupling(P.aux.p, r.W(m.dec.alpha_off), r.W(m.dec.up_off), r.W(m.dec.down_off),
                              r.W(m.dec.conv.w_off), r.W(m.dec.conv.b_off), tmp, P.wav.p, B, m.C0, T, st, r.lens_of(T),
                              r.lens_of(2 * T)), "decoupling");
  }
  auto post = [&](const float* x) {
    if (!r.ok()) return;
    if (r.ragged) r.chk(launch_post_var(x, P.stats, out, P.rows, B, T_raw, T, keep_rms, peak, st), "post");
    else r.chk(launch_post(x, P.stats, out, B, T_raw, T, pad_left, keep_rms, peak, st), "post");
  };
  if (use_aux) {
    post(P.wav.p);
    return finish(h, r);
  }
  // universe.py:325-331
  if (!have_e0) {
    r.chk(launch_init_x(noise, warm_start >= 0 ? P.wav.p : nullptr, sigma[n_start], P.x.p, nBT, st), "init x");
    r.mask(P.x);
  }
  const size_t step_mark = r.off;
  for (int n = n_start; n < n_steps; n++) {
    const bool last = n == n_steps - 1;
    const float* z = last ? nullptr : noise + (size_t)(n - n_start + 1) * nBT;
    const StepCoef* cf = P.coef + n;
    const float* fr = P.film + (size_t)n * m.film.rows;
    if (n == n_start && have_e0) {
      r.join(2, st);
      r.off = off_after_enc;
      run_score_dec(r, P, E0, P.x.p, z, P.x.p, OUT_UPDATE, cf, 0, fr, 0, T);
    } else {
      r.off = step_mark;  // every step re-uses the same scratch
      run_score(r, P, P.x.p, z, P.x.p, OUT_UPDATE, cf, 0, fr, 0, T);
    }
    if (!r.ok()) break;
  }
  post(P.x.p);
  return finish(h, r);
}
}  // namespace

extern "C" {

int ou_enhance(ou_handle* h, const float* mix, float* out, const float* noise, int32_t B, int32_t T_raw, int32_t n_steps,
               double epsilon, const float* sigma_host, int32_t warm_start, uint32_t flags, void* ws, size_t ws_bytes,
               ou_stream_t stream) {
  return enhance_impl(h, mix, out, noise, B, T_raw, nullptr, n_steps, epsilon, sigma_host, warm_start, flags, ws, ws_bytes,
                      stream);
}

int ou_enhance_var(ou_handle* h, const float* mix, float* out, const float* noise, int32_t B, int32_t T_raw_max,
                   const int32_t* t_raw, int32_t n_steps, double epsilon, const float* sigma_host, int32_t warm_start,
                   uint32_t flags, void* ws, size_t ws_bytes, ou_stream_t stream) {
  if (!t_raw) return fail(h, OU_EINVAL, "ou_enhance_var: t_raw must be given (ou_enhance takes batches of equal lengths)");
  return enhance_impl(h, mix, out, noise, B, T_raw_max, t_raw, n_steps, epsilon, sigma_host, warm_start, flags, ws, ws_bytes,
                      stream);
}

int ou_transform_frames(int32_t T, int32_t n_fft, int32_t hop) {
  if (T < 1 || n_fft < 2 || hop < 1) return OU_EINVAL;
  return 1 + (T + 2 * (n_fft / 2) - n_fft) / hop;
}

int ou_transform_forward(const float* x, int32_t B, int32_t T, const float* window, int32_t n_fft, int32_t hop,
                         int32_t transform_type, float abs_exponent, float factor, float* out, ou_stream_t stream) {
  if (!x || !window || !out || B < 1) return fail(nullptr, OU_EINVAL, "bad argument");
  if (transform_type < 0 || transform_type > 2) return fail(nullptr, OU_ENOTIMPL, "transform_type must be none | exponent | log");
  hipError_t e = launch_stft_forward(x, window, out, B, T, n_fft, hop, transform_type, abs_exponent, factor, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? OU_EINVAL : OU_EHIP, hipGetErrorString(e));
  return OU_OK;
}

int ou_transform_inverse(const float* spec, int32_t B, int32_t n_frames, const float* window, int32_t n_fft, int32_t hop,
                         int32_t transform_type, float abs_exponent, float factor, int32_t length, float* y,
                         float* scratch, ou_stream_t stream) {
  if (!spec || !window || !y || !scratch || B < 1) return fail(nullptr, OU_EINVAL, "bad argument");
  if (transform_type < 0 || transform_type > 2) return fail(nullptr, OU_ENOTIMPL, "transform_type must be none | exponent | log");
  hipError_t e = launch_stft_inverse(spec, window, scratch, y, B, n_frames, n_fft, hop, transform_type, abs_exponent, factor,
                                     length, (hipStream_t)stream);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? OU_EINVAL : OU_EHIP, hipGetErrorString(e));
  return OU_OK;
}

int ou_check_device_status(ou_handle* h, void* ws) {
  if (!h || !ws) return fail(h, OU_EINVAL, "bad argument");
  unsigned hdr[40];
  hipError_t e = hipMemcpy(hdr, ws, sizeof(hdr), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(h, OU_EHIP, hipGetErrorString(e));
  const unsigned v = hdr[0];
  // word 20: waits the GRU clusters' safety net cut short by repeating a publish (a member that was merely late counts
  // too); word 33: those among them where the awaited granule was visible to a system-scope load / an atomic but NOT to the
  // agent-scope load of the gather.  The first time word 33 moves, the cheap publish form has shown that it cannot be
  // relied upon on this device / in this process mix: agent-scope publishes from now on (+0.1 ms per GRU pass, no more
  // 0.1 ms recoveries).  Results are unaffected either way.
  if (hdr[33] != 0u && !h->gru_agent_stores) h->gru_agent_stores = 1;
  if (v) {
    (void)hipMemset(ws, 0, sizeof(v));  // sticky until read
    return fail(h, OU_ESYNC, "device-side timeout in the GRU cluster exchange (status word " + std::to_string(v) + ")");
  }
  return OU_OK;
}

int ou_workspace_init(ou_handle* h, int32_t B, int32_t T, void* ws, size_t ws_bytes, ou_stream_t stream) {
  if (!h || !ws || B < 1 || T < 1) return fail(h, OU_EINVAL, "bad argument");
  if (T % h->m.tot_ds) return fail(h, OU_EINVAL, "T must be a multiple of the total down-sampling factor");
  auto keep = h->tensors;
  Runner r(h, ws, ws_bytes, false, (hipStream_t)stream, B);
  Persist P = layout_persist(r, T);
  h->tensors = keep;
  if (r.oom) return finish(h, r);
  // header: status word, coefficient rows, statistics, both GRU exchange areas (everything in front of mel_scale)
  const size_t hdr = (size_t)((char*)P.mel_scale - (char*)ws);
  r.chk(hipMemsetAsync(ws, 0, hdr, r.st), "workspace init");
  const int rc = finish(h, r);
  if (rc == OU_OK) {
    // remember (pointer, size, shape); a buffer that comes back at the same address is re-registered by its own init
    auto& v = h->ws_ready;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].ws == ws) { v.erase(v.begin() + i); break; }
    if (v.size() >= 64) v.erase(v.begin());
    v.push_back(ou_handle::WsRec{ws, ws_bytes, B, T});
  }
  return rc;
}

int ou_set_lanes(ou_handle* h, int32_t lanes, int32_t lane) {
  if (!h || lanes < 1 || lanes > 8 || lane < 0 || lane >= lanes) return fail(h, OU_EINVAL, "ou_set_lanes: 1 <= lanes <= 8, 0 <= lane < lanes");
  h->lanes = lanes;
  h->lane = lane;
  return OU_OK;
}

int ou_lane_capacity(const ou_handle* h, int32_t max_batch) {
  if (!h || max_batch < 1) return 0;
  // the largest number of lanes whose GRU launches -- every lane at `max_batch` -- can all be resident at the same time
  const Model& m = h->m;
  for (int lanes = 8; lanes >= 1; lanes--) {
    const int share = Runner::gru_share_of(lanes, max_batch, false);
    bool ok = true;
    for (int H : {m.s_gru.H, m.c_gru0.H, m.c_gru1.H})
      if (H > 0 && gru_ring_batch_cap(H, h->num_cu, share, 0, max_batch, lanes) < 1) ok = false;
    if (ok) return lanes;
  }
  return 1;
}

int ou_set_lane_batch(ou_handle* h, int32_t max_batch) {
  if (!h || max_batch < 0) return fail(h, OU_EINVAL, "ou_set_lane_batch: max_batch >= 0");
  h->lane_max_b = max_batch;
  return OU_OK;
}

int ou_get_gru_publish_mode(const ou_handle* h) { return h ? h->gru_agent_stores : 0; }

int ou_set_gru_publish_mode(ou_handle* h, int32_t agent_scope) {
  if (!h) return fail(h, OU_EINVAL, "bad argument");
  h->gru_agent_stores = agent_scope ? 1 : 0;
  return OU_OK;
}

int ou_sampler_step(ou_handle* h, float* x, const float* score, const float* z, float c1, float c2, size_t n,
                    ou_stream_t stream) {
  if (!h || !x || !score) return fail(h, OU_EINVAL, "bad argument");
  hipError_t e = launch_sampler_step(x, score, z, c1, c2, n, (hipStream_t)stream);
  if (e != hipSuccess) return fail(h, OU_EHIP, hipGetErrorString(e));
  return OU_OK;
}

int ou_tensor(const ou_handle* h, const char* name, size_t* byte_offset, int32_t* C, int32_t* T) {
  if (!h || !name) return OU_EINVAL;
  auto it = h->tensors.find(name);
  if (it == h->tensors.end()) return OU_EMISSING;
  if (byte_offset) *byte_offset = it->second.off;
  if (C) *C = it->second.C;
  if (T) *T = it->second.T;
  return OU_OK;
}

int ou_launch_stats(const ou_handle* h, int32_t* n_launches, int32_t* n_conv_launches) {
  if (!h) return OU_EINVAL;
  if (n_launches) *n_launches = h->n_launch;
  if (n_conv_launches) *n_conv_launches = h->n_conv;
  return OU_OK;
}

// Micro-benchmark of ONE packed conv layer (measurement / tuning only): runs it `iters` times on random-ish data in
// the caller's workspace with HIP events around the batch; cfg/sc < 0: the launcher's own choice.
int ou_bench_conv(ou_handle* h, const char* layer, int32_t B, int32_t Tin, int32_t cfg, int32_t sc, int32_t with_res,
                  int32_t iters, void* ws, size_t ws_bytes, ou_stream_t stream, float* ms_per_iter, int32_t* cfg_used) {
  if (!h || !layer || !ws || !ms_per_iter) return fail(h, OU_EINVAL, "bad argument");
  const Model& m = h->m;
  std::vector<const ConvL*> all;
  auto addb = [&](const BlockL& b) { if (b.dir) all.push_back(&b.rc); all.push_back(&b.c1); all.push_back(&b.c2); all.push_back(&b.c3); };
  for (auto& b : m.s_enc) addb(b);
  for (auto& b : m.s_dec) addb(b);
  for (auto& l : m.s_sig) all.push_back(&l);
  all.push_back(&m.s_gru.proj); all.push_back(&m.c_melconv); addb(m.c_melblock);
  for (auto& b : m.c_enc) addb(b);
  for (auto& l : m.c_st) all.push_back(&l);
  all.push_back(&m.c_gru0.proj); all.push_back(&m.c_gru1.proj);
  addb(m.c_cb1); addb(m.c_cb2); addb(m.c_decin);
  for (auto& b : m.c_dec) addb(b);
  const ConvL* L = nullptr;
  for (auto* l : all) if (l->name == layer) L = l;
  if (!L) return fail(h, OU_EMISSING, std::string("no such conv layer: ") + layer);
  hipStream_t st = (hipStream_t)stream;
  Runner r(h, ws, ws_bytes, false, st, B);
  auto keep = h->tensors;
  // (kernels whose windows begin a few samples in front of a row read -- and mask -- the bytes in front of their input tensor: it
  // must not be the first bytes of the caller's allocation; in the model's own layout the status header comes first)
  r.alloc_raw(64);
  Tensor in = r.alloc("", L->Cin, Tin);
  if (r.oom) return finish(h, r);
  r.chk(hipMemsetAsync(in.p, 0x3c, (size_t)B * L->Cin * Tin * 4, st), "fill");
  h->force_cfg = cfg; h->force_sc = sc < 0 ? 0 : sc;
  if (h->opt.ts) h->tstamps = (long long*)((char*)ws + ws_bytes - (16u << 20));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  size_t mark = r.off;
  Runner::Epi e;
  Tensor out = r.conv(*L, in, "", e);  // warm-up + output allocation
  if (with_res) e.res = out.p;
  for (int w = 0; w < 2 && r.ok(); w++) { r.off = mark; r.conv(*L, in, "", e); }
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters && r.ok(); i++) { r.off = mark; r.conv(*L, in, "", e); }
  (void)hipEventRecord(e1, st);
  h->force_cfg = -1; h->force_sc = 0;
  h->tstamps = nullptr;
  h->tensors = keep;
  int rc = finish(h, r);
  if (rc == OU_OK) {
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_per_iter = ms / (iters > 0 ? iters : 1);
    if (cfg_used) *cfg_used = h->last_cfg;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

int ou_profile_enable(ou_handle* h, int32_t on) {
  if (!h) return OU_EINVAL;
  h->profile = on != 0;
  h->prof_used = 0;
  h->prof.clear();
  if (on) {
    // measurement buffer: owned by the library, allocated outside any forward call
    if (!h->prof_dev && hipMalloc((void**)&h->prof_dev, kProfSlots * 256) != hipSuccess)
      return fail(h, OU_EHIP, "hipMalloc(profile buffer) failed");
    if (hipMemset(h->prof_dev, 0xFF, kProfSlots * 256) != hipSuccess) return fail(h, OU_EHIP, "hipMemset failed");
  }
  return OU_OK;
}

// ... and the raw stamps of the same records: first block start / last block end of every launch on the device's 100 MHz
// constant clock (10 ns ticks; one clock for all handles of a process, so that the launches of several lanes can be laid on
// one timeline), plus the variant code
int ou_profile_read_ticks(ou_handle* h, int32_t max_records, uint64_t* t_start, uint64_t* t_end, int32_t* cfg,
                          int32_t* n_records) {
  if (!h || !n_records || !t_start || !t_end) return OU_EINVAL;
  int n = (int)h->prof_used;
  if (n > max_records) n = max_records;
  std::vector<unsigned long long> host((size_t)32 * (n > 0 ? n : 1));
  if (n > 0) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(host.data(), h->prof_dev, (size_t)n * 256, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(h, OU_EHIP, hipGetErrorString(e));
  }
  for (int i = 0; i < n; i++) {
    unsigned long long t0 = ~0ull, t1 = 0ull;
    for (int w = 0; w < 16; w++) {
      const unsigned long long a = host[32 * i + w], b = ~host[32 * i + 16 + w];
      if (a < t0) t0 = a;
      if (b > t1) t1 = b;
    }
    t_start[i] = t0;
    t_end[i] = t1 >= t0 ? t1 : t0;
    if (cfg) cfg[i] = h->prof[i].cfg;
  }
  *n_records = n;
  return OU_OK;
}

int ou_profile_read(ou_handle* h, int32_t max_records, float* ms, double* flops, double* bytes, int32_t* cfg,
                    int32_t* n_records) {
  if (!h || !n_records) return OU_EINVAL;
  int n = (int)h->prof_used;
  if (n > max_records) n = max_records;
  std::vector<unsigned long long> host((size_t)32 * (n > 0 ? n : 1));
  if (n > 0) {
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(host.data(), h->prof_dev, (size_t)n * 256, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(h, OU_EHIP, hipGetErrorString(e));
  }
  for (int i = 0; i < n; i++) {
    unsigned long long t0 = ~0ull, t1 = 0ull;
    for (int w = 0; w < 16; w++) {
      const unsigned long long a = host[32 * i + w], b = ~host[32 * i + 16 + w];
      if (a < t0) t0 = a;
      if (b > t1) t1 = b;
    }
    if (ms) ms[i] = (t1 >= t0) ? (float)((double)(t1 - t0) * 1e-5) : 0.f;  // 100 MHz constant clock: 10 ns ticks
    if (flops) flops[i] = h->prof[i].flops;
    if (bytes) bytes[i] = h->prof[i].bytes;
    if (cfg) cfg[i] = h->prof[i].cfg;
  }
  *n_records = n;
  return OU_OK;
}

}  // extern "C"
