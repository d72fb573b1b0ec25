// conv_split_kernel (open-universe_amd/csrc/ou_conv_split.hip) on its own: one stride-1 k3 / k5 conv layer with synthetic data,
// checked against a double evaluation (beside the error of a plain fp32 fmaf chain on the same data) and timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o split_conv.bin split_conv.hip
//   ./split_conv.bin M Cin KW T B [force_cfg] [iters]
#define OU_SPLIT_TUNING 1
#include "../../open-universe_amd/csrc/ou_conv_split.hip"
#include "../../open-universe_amd/csrc/ou_split_pack.h"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define CHK(x)                                                                          \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } \
  } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 128, Cin = argc > 2 ? atoi(argv[2]) : 128, KW = argc > 3 ? atoi(argv[3]) : 5;
  const int T = argc > 4 ? atoi(argv[4]) : 8020, B = argc > 5 ? atoi(argv[5]) : 8;
  const int force = argc > 6 ? atoi(argv[6]) : -1, iters = argc > 7 ? atoi(argv[7]) : 20;
  const int Mp = (M + 63) / 64 * 64;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> x((size_t)B * Cin * T), W((size_t)M * Cin * KW), bias(M), res((size_t)B * M * T);
  for (auto& v : x) v = nd(rng);
  const float ws = 1.0f / std::sqrt((float)Cin * KW);
  for (auto& v : W) v = ws * nd(rng);
  for (auto& v : bias) v = 0.1f * nd(rng);
  for (auto& v : res) v = nd(rng);
  std::vector<uint16_t> wsplit(ou::split_floats(Cin, KW, Mp) * 2);
  ou::pack_split(W.data(), M, Mp, Cin, KW, wsplit.data());
  float *dx, *dy, *db, *dr;
  void* dw;
  CHK(hipMalloc(&dx, x.size() * 4 + 256));
  dx += 32;  // (windows that start in front of a row are masked, never read)
  CHK(hipMalloc(&dy, res.size() * 4));
  CHK(hipMalloc(&dr, res.size() * 4));
  CHK(hipMalloc(&db, M * 4));
  CHK(hipMalloc(&dw, wsplit.size() * 2));
  CHK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dr, res.data(), res.size() * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(db, bias.data(), M * 4, hipMemcpyHostToDevice));
  CHK(hipMemcpy(dw, wsplit.data(), wsplit.size() * 2, hipMemcpyHostToDevice));
  CHK(hipMemset(dy, 0, res.size() * 4));
  CHK(ou::init_split_kernels());
  hipDeviceProp_t pr;
  CHK(hipGetDeviceProperties(&pr, 0));
  ou::ConvArgs a;
  a.x = dx; a.wsplit = dw; a.bias = db; a.y = dy; a.res = dr; a.res_scale = 0.70710678f;
  const int act = argc > 9 ? atoi(argv[9]) : 1;  // PReLU in the operand path (the variant without it: layers whose producer activated)
  a.act = act; a.alpha_val = 0.25f;
  a.B = B; a.Cin = Cin; a.Tin = T; a.Cout = M; a.M = M; a.Mp = Mp; a.KW = KW; a.pad = (KW - 1) / 2; a.Nq = T; a.Tout = T;
  a.force_cfg = force;
  a.dbg = argc > 8 ? atoi(argv[8]) : 0;  // tuning: 1 = no epilogue (wrong results), 16 / 32 + 256 x us = every second block starts late
  int cfg = 0;
  CHK(ou::launch_conv_split(a, pr.multiProcessorCount, 0, &cfg));
  CHK(hipDeviceSynchronize());
  std::vector<float> y(res.size());
  CHK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
  // sampled check: double evaluation, and a plain fp32 fmaf chain (channels ascending, taps ascending) for scale
  std::uniform_int_distribution<size_t> pick(0, y.size() - 1);
  double e_split = 0, e_f32 = 0, ref2 = 0, worst = 0;
  const int NS = 20000;
  for (int s = 0; s < NS + 4 * M; s++) {
    size_t idx = pick(rng);
    if (s >= NS) {  // edges: first / last two samples of row (s - NS) / 4 of the last batch element
      const int row = (s - NS) / 4, e = (s - NS) % 4;
      const int t = e < 2 ? e : T - 1 - (e - 2);
      idx = ((size_t)(B - 1) * M + row) * T + t;
    }
    const int t = idx % T, m = (idx / T) % M, b = idx / ((size_t)T * M);
    double acc = 0;
    float accf = 0.f;
    for (int ci = 0; ci < Cin; ci++)
      for (int k = 0; k < KW; k++) {
        const int tt = t + k - (KW - 1) / 2;
        if (tt < 0 || tt >= T) continue;
        float xv = x[((size_t)b * Cin + ci) * T + tt];
        if (act) xv = xv >= 0.f ? xv : 0.25f * xv;
        const float wv = W[((size_t)m * Cin + ci) * KW + k];
        acc += (double)wv * (double)xv;
        accf = fmaf(wv, xv, accf);
      }
    const double ref = (acc + bias[m] + res[idx]) * 0.70710678f;
    const float reff = (accf + bias[m] + res[idx]) * 0.70710678f;
    e_split += (y[idx] - ref) * (y[idx] - ref);
    e_f32 += (reff - ref) * (reff - ref);
    ref2 += ref * ref;
    worst = std::fmax(worst, std::fabs(y[idx] - ref));
  }
  printf("M=%d Cin=%d KW=%d T=%d B=%d cfg=%d act=%d: SNR vs double %.1f dB (plain fp32 fmaf chain on the same samples: %.1f dB), worst abs err %.3g\n",
         M, Cin, KW, T, B, cfg, act, 10 * std::log10(ref2 / (e_split + 1e-300)), 10 * std::log10(ref2 / (e_f32 + 1e-300)), worst);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int i = 0; i < 3; i++) CHK(ou::launch_conv_split(a, pr.multiProcessorCount, 0, nullptr));
  CHK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; i++) CHK(ou::launch_conv_split(a, pr.multiProcessorCount, 0, nullptr));
  CHK(hipEventRecord(e1, 0));
  CHK(hipEventSynchronize(e1));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters, gflop = 2.0 * M * Cin * KW * (double)T * B * 1e-9;
  printf("    %.1f us per launch: %.1f TFLOP/s algorithmic (fp32 MFMA peak 157.3), %.0f TFLOP/s on the bf16 pipe (6 products; peak 2500)\n", us,
         gflop / us * 1e3, gflop * 6 / us * 1e3);
  if (getenv("OU_TS")) {  // per-wave phase cycles of the main loop (s_memtime-class counter: shader clocks)
    const size_t nw = 4096 * 4;
    long long* dts;
    CHK(hipMalloc(&dts, nw * 8 * 8));
    CHK(hipMemset(dts, 0, nw * 8 * 8));
    a.tstamps = dts;
    CHK(ou::launch_conv_split(a, pr.multiProcessorCount, 0, nullptr));
    CHK(hipDeviceSynchronize());
    std::vector<long long> ts(nw * 8);
    CHK(hipMemcpy(ts.data(), dts, nw * 8 * 8, hipMemcpyDeviceToHost));
    double sum[6] = {0}, n = 0;
    for (size_t w = 0; w < nw; w++)
      if (ts[w * 8 + 5]) { n++; for (int i = 0; i < 6; i++) sum[i] += ts[w * 8 + i]; }
    printf("    main loop per wave (cycles, %d waves, %.0f steps): steps %.0f (%.1f per MFMA), barriers + first fragments %.0f, total %.0f\n",
           (int)n, sum[5] / n, sum[2] / n, sum[2] / n / (sum[5] / n * 12 * (cfg >= 900 ? 2 : 4)), sum[3] / n, sum[4] / n);
  }
  return 0;
}
