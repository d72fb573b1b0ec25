// gfx950 (CDNA4 / MI355X) kernels of the UNIVERSE(++) enhance path: bandwidth / VALU kernels (in / out conv, embedding, FiLM, pad / normalise / post, STFT, mel, s2d, FIR, sum, Snake)
// (one translation unit per kernel family; shared device helpers in ou_dev.h, cross-file launchers in ou_internal.h)
#include "ou_kernels.h"
#include "ou_internal.h"
#include "ou_dev.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace ou {

// =========================================================================================================
// small VALU kernels
// =========================================================================================================

// grid = (time tiles of 256, channel groups of IN_CONV_CG, batch): at batch 1 a (T/256)-block grid is one block per CU
// with 32 dependent stores per thread; splitting the channels gives the dispatcher 4x the blocks for the same traffic
constexpr int IN_CONV_CG = 8;
__global__ __launch_bounds__(256) void in_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const StepCoef* coef,
                                                      int coef_bstride, float* __restrict__ y, int C, int T, int KW,
                                                      const int* __restrict__ lens) {
  const int b = blockIdx.z;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const bool live = t < ragged_len(lens, b);  // ragged batch: 0 behind the row's own end (the bias would stand there)
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
    y[((size_t)b * C + c) * T + t] = live ? acc + bias[c] : 0.f;
  }
}

hipError_t launch_in_conv(const float* x, const float* w, const float* bias, const StepCoef* coef, int coef_bstride,
                          float* y, int B, int C, int T, int KW, hipStream_t s, const int* lens) {
  if (KW > 7) return hipErrorInvalidValue;
  hipLaunchKernelGGL(in_conv_kernel, dim3((T + 255) / 256, (C + IN_CONV_CG - 1) / IN_CONV_CG, B), dim3(256), 0, s, x, w,
                     bias, coef, coef_bstride, y, C, T, KW, lens);
  return hipGetLastError();
}

__global__ __launch_bounds__(512) void out_conv_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ alphas,
                                                       const float* x, const float* noise, float* out,
                                                       const StepCoef* coef, int coef_bstride, int edm, int mode, int C,
                                                       int T, int KW, const int* __restrict__ lens) {
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
  if (t >= ragged_len(lens, b)) { out[(size_t)b * T + t] = 0.f; return; }  // ragged batch: behind the row's own end
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
                           int B, int C, int T, int KW, hipStream_t st, const int* lens) {
  if (KW > 7) return hipErrorInvalidValue;
  hipLaunchKernelGGL(out_conv_kernel, dim3((T + 255) / 256, B), dim3(512), 0, st, s, w, bias, alphas, x, noise, out,
                     coef, coef_bstride, edm, mode, C, T, KW, lens);
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
__global__ __launch_bounds__(256) void mel_scale_kernel(const float* __restrict__ esum, float* scale, int L,
                                                        const int* __restrict__ lens) {
  __shared__ double shd[4];
  const int b = blockIdx.x;
  const int Lb = lens ? lens[b] : L;  // ragged batch: the row's own frames (same elements per thread, same order as alone)
  double s = 0;
  for (int f = threadIdx.x; f < Lb; f += 256) s += esum[(size_t)b * L + f];
  s = block_sum(s, shd);
  if (threadIdx.x == 0) scale[b] = 1.0f / fmaxf((float)sqrt(s / Lb), 1e-5f);
}
hipError_t launch_mel_scale(const float* esum, float* scale, int B, int L, hipStream_t st, const int* lens) {
  hipLaunchKernelGGL(mel_scale_kernel, dim3(B), dim3(256), 0, st, esum, scale, L, lens);
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
                                                  int T, const int* __restrict__ lens) {
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
    if (t >= ragged_len(lens, b)) acc = 0.f;
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
                                                   float res_scale, float* __restrict__ y, int C, int T,
                                                   const int* __restrict__ lens) {
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
  if (lens) o = ragged_mask4(o, t, lens[b]);
  if (full) *reinterpret_cast<f32x4u*>(y + row + t) = o;
  else {
#pragma unroll
    for (int e = 0; e < 4; e++) if (t + e < T) y[row + t + e] = o[e];
  }
}
hipError_t launch_fir(const float* x, const float* taps, int ntaps, float alpha, int act, const float* bias,
                      const float* res, float res_scale, float* y, int B, int C, int T, hipStream_t st, const int* lens) {
  if (ntaps > 39 || !(ntaps & 1)) return hipErrorInvalidValue;
  const dim3 grid((T + FIR_TILE - 1) / FIR_TILE, C, B);
  {  // (dwordx4 accesses at dword alignment: any T; the scalar kernel below takes the tap counts without an instantiation)
    void (*k)(const float*, const float*, float, int, const float*, const float*, float, float*, int, int, const int*) = nullptr;
    switch (ntaps) {
      case 5: k = fir4_kernel<5>; break;
      case 7: k = fir4_kernel<7>; break;
      case 9: k = fir4_kernel<9>; break;
      case 11: k = fir4_kernel<11>; break;
      case 17: k = fir4_kernel<17>; break;
      default: break;
    }
    if (k) {
      hipLaunchKernelGGL(k, grid, dim3(256), 0, st, x, taps, alpha, act, bias, res, res_scale, y, C, T, lens);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(fir_kernel, grid, dim3(256), 0, st, x, taps, ntaps, alpha, act, bias, res, res_scale, y, C, T, lens);
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
// signal decoupling layer (cold branch): 2x sinc upsample -> Snake -> 2x downsample -> Conv1d(C -> 1, k3)
//   torchaudio Resample(1->2): pad (7, 8), conv with the 2 phase kernels (15 taps), interleave, crop to 2T.
//   Resample(2->1): pad (13, 15), 28-tap kernel, stride 2.
// =========================================================================================================
__global__ __launch_bounds__(256) void snake_up_kernel(const float* __restrict__ aux, const float* __restrict__ alpha_exp,
                                                       const float* __restrict__ up_k, float* __restrict__ u, int C,
                                                       int T, const int* __restrict__ lens2) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * 256 + threadIdx.x;  // index in the 2T up-sampled signal
  if (i >= 2 * T) return;
  if (lens2 && i >= lens2[b]) { u[((size_t)b * C + c) * 2 * T + i] = 0.f; return; }  // ragged batch: past the row's end
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
                                                              float* __restrict__ out, int C, int Tfull,
                                                              const int* __restrict__ lens) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= Tfull) return;
  // ragged batch: the k3 conv's zero padding begins right behind the ROW's last sample
  const int T = lens ? lens[b] : Tfull;
  if (t >= T) { out[(size_t)b * Tfull + t] = 0.f; return; }
  float acc = 0.f;
  for (int c = 0; c < C; c++) {
    const float* ur = u + ((size_t)b * C + c) * 2 * Tfull;
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
  out[(size_t)b * Tfull + t] = acc + bias[0];
}
hipError_t launch_decoupling(const float* aux, const float* alpha_exp, const float* up_k, const float* down_k,
                             const float* w, const float* bias, float* tmp_up, float* out, int B, int C, int T,
                             hipStream_t st, const int* lens, const int* lens2) {
  hipLaunchKernelGGL(snake_up_kernel, dim3((2 * T + 255) / 256, C, B), dim3(256), 0, st, aux, alpha_exp, up_k, tmp_up, C, T,
                     lens2);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(snake_down_conv_kernel, dim3((T + 255) / 256, B), dim3(256), 0, st, tmp_up, down_k, w, bias, out, C, T,
                     lens);
  return hipGetLastError();
}

// =========================================================================================================
// ragged batches (ou_enhance_var): per-row geometry, tail masks, per-row pad / normalise / post
// =========================================================================================================
__global__ void upload_rows_kernel(RowInfo* rows, int* lens, RowBlock blk, int n, int off, int B, int tot_ds, LevelSpec lv) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const int t_raw = blk.t_raw[i];
  const int pad = tot_ds - t_raw % tot_ds;  // universe.py:219-223 (a full block when already a multiple)
  RowInfo r;
  r.t_raw = t_raw; r.pad_left = pad / 2; r.t_pad = t_raw + pad; r._r = 0;
  rows[off + i] = r;
  for (int l = 0; l < lv.n; l++) lens[l * B + off + i] = (int)((long long)r.t_pad * lv.num[l] / lv.den[l]);
}
hipError_t launch_upload_rows(RowInfo* rows, int* lens, const RowBlock& blk, int n, int off, int B, int tot_ds,
                              const LevelSpec& lv, hipStream_t st) {
  hipLaunchKernelGGL(upload_rows_kernel, dim3(1), dim3(64), 0, st, rows, lens, blk, n, off, B, tot_ds, lv);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void mask_tail_kernel(float* __restrict__ y, const int* __restrict__ lens, int C, int T) {
  const int c = blockIdx.x, b = blockIdx.y;
  float* yr = y + ((size_t)b * C + c) * T;
  for (int t = lens[b] + threadIdx.x; t < T; t += 256) yr[t] = 0.f;
}
hipError_t launch_mask_tail(float* y, const int* lens, int B, int C, int T, hipStream_t st) {
  hipLaunchKernelGGL(mask_tail_kernel, dim3(C, B), dim3(256), 0, st, y, lens, C, T);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void gru_tail_fill_kernel(float* __restrict__ gx, const int* __restrict__ lens, int H, int T) {
  const int row = blockIdx.x, b = blockIdx.y;  // row = dir * 3H + gate * H + unit
  const float v = (row / H) % 3 == 1 ? 1e4f : 0.f;
  float* yr = gx + ((size_t)b * 6 * H + row) * T;
  for (int t = lens[b] + threadIdx.x; t < T; t += 256) yr[t] = v;
}
hipError_t launch_gru_tail_fill(float* gx, const int* lens, int B, int H, int T, hipStream_t st) {
  hipLaunchKernelGGL(gru_tail_fill_kernel, dim3(6 * H, B), dim3(256), 0, st, gx, lens, H, T);
  return hipGetLastError();
}

// pad_normalize_kernel's general loops with the row's own (T_raw, pad_left, T_pad): same elements per thread and the same
// order of the double sums as the call on that row alone; the row is ZERO (not the padding value) from its own T_pad on
__global__ __launch_bounds__(1024) void pad_normalize_var_kernel(const float* __restrict__ mix, float* __restrict__ y,
                                                                 float* __restrict__ stats, const RowInfo* __restrict__ rows,
                                                                 int T_raw_max, int T_pad_max, float level) {
  __shared__ double shd[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const RowInfo ri = rows[b];
  const int T_raw = ri.t_raw, T_pad = ri.t_pad, pad_left = ri.pad_left;
  const float* xb = mix + (size_t)b * T_raw_max;
  float* yb = y + (size_t)b * T_pad_max;
  double s = 0, sq = 0;
  for (int t = tid; t < T_raw; t += 1024) { const double d = xb[t]; s += d; sq += d * d; }
  s = block_sum(s, shd);
  sq = block_sum(sq, shd);
  const float mean = (float)(s / T_pad);  // norm.py:62  (mean over the padded signal)
  double ss = 0;
  for (int t = tid; t < T_raw; t += 1024) { const double d = (double)(xb[t] - mean); ss += d * d; }
  ss = block_sum(ss, shd);
  ss += (double)(T_pad - T_raw) * (double)(0.f - mean) * (double)(0.f - mean);
  float sd = (float)sqrt(ss / (double)(T_pad - 1));  // unbiased std, norm.py:22-23
  sd = fmaxf(sd, 1e-5f);
  const float gain = level / sd;
  for (int t = tid; t < T_pad_max; t += 1024) {
    const int tr = t - pad_left;
    const float v = (tr >= 0 && tr < T_raw) ? xb[tr] : 0.f;
    yb[t] = t < T_pad ? (v - mean) * gain : 0.f;
  }
  if (tid == 0) {
    stats[b * 4 + 0] = mean;
    stats[b * 4 + 1] = gain;
    stats[b * 4 + 2] = (float)sqrt(sq / T_raw);  // mix_rms, universe.py:259
    stats[b * 4 + 3] = 0.f;
  }
}
hipError_t launch_pad_normalize_var(const float* mix, float* y, float* stats, const RowInfo* rows, int B, int T_raw_max,
                                    int T_pad_max, float level, hipStream_t st) {
  hipLaunchKernelGGL(pad_normalize_var_kernel, dim3(B), dim3(1024), 0, st, mix, y, stats, rows, T_raw_max, T_pad_max, level);
  return hipGetLastError();
}
// post_kernel with the row's own geometry; out (B, T_raw_max), 0 behind the row's own T_raw
__global__ __launch_bounds__(1024) void post_var_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                        float* __restrict__ out, const RowInfo* __restrict__ rows,
                                                        int T_raw_max, int T_pad_max, int keep_rms, int peak_guard) {
  __shared__ double shd[16];
  __shared__ float shf[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const RowInfo ri = rows[b];
  const int T_raw = ri.t_raw;
  const float* xb = x + (size_t)b * T_pad_max + ri.pad_left;
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
  for (int t = tid; t < T_raw; t += 1024) mx = fmaxf(mx, fabsf(xb[t] * g));
  mx = block_max(mx, shf);
  const bool div = peak_guard && mx > 1.0f;  // universe.py:356-357
  for (int t = tid; t < T_raw_max; t += 1024) {
    float v = t < T_raw ? xb[t] : 0.f;
    if (keep_rms) v = v * g;
    if (div) v = v / mx;
    out[(size_t)b * T_raw_max + t] = v;
  }
}
hipError_t launch_post_var(const float* x, const float* stats, float* out, const RowInfo* rows, int B, int T_raw_max,
                           int T_pad_max, int keep_rms, int peak_guard, hipStream_t st) {
  hipLaunchKernelGGL(post_var_kernel, dim3(B), dim3(1024), 0, st, x, stats, out, rows, T_raw_max, T_pad_max, keep_rms,
                     peak_guard);
  return hipGetLastError();
}


}  // namespace ou
