// Device-side helpers shared by the kernel files: vector typedefs, buffer descriptors, LDS-DMA, counted waits.
// No portability layer: wave = 64, fp32 MFMA, DPP row reductions, agent-scope granule hand-offs.
#pragma once
#include <hip/hip_runtime.h>

#include "ou_kernels.h"

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
  } else if (p.xcd_map == 3 || p.xcd_map == 4) {
    // 2-D ownership (round 5): the eight XCDs as a (2 x 4) or (4 x 2) grid of (row groups x column groups) -- an XCD's L2
    // then fetches 1/2 (1/4) of the weights and 1/4 (1/2) of the activations instead of all of one and an eighth of the
    // other: less memory-side traffic where the two operands are of similar size (the 256-channel level: 1.0 instead of
    // 1.3 MB per XCD for k3, 1.6 instead of 2.3 MB for k5)
    const int rg = p.xcd_map == 3 ? 2 : 4, cg = 8 / rg;
    const int x = L & 7, q = L >> 3, xr = x % rg, xc = x / rg;
    const int gmh = (gy + rg - 1) / rg, gnq = (gx + cg - 1) / cg;
    const int nn = q / gmh, mm = q - nn * gmh;
    tile_m = xr * gmh + mm;
    tile_n = xc * gnq + nn;
    if (nn >= gnq || tile_m >= gy || tile_n >= gx) return false;
  } else {
    tile_m = L / gx;
    tile_n = L - tile_m * gx;
  }
  return true;
}
// blocks a launch needs under mapping `xcd_map` (padding blocks of the XCD-aware mappings included)
inline int direct_grid_blocks(int xcd_map, int grid_m, int grid_n) {
  if (xcd_map == 2) return (grid_n + 7) / 8 * 8 * grid_m;
  if (xcd_map == 3 || xcd_map == 4) {
    const int rg = xcd_map == 3 ? 2 : 4, cg = 8 / rg;
    return 8 * ((grid_m + rg - 1) / rg) * ((grid_n + cg - 1) / cg);
  }
  return grid_m * grid_n;
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

__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.f ? v : a * v; }

// Ragged batches (ConvArgs::lens, ou_kernels.h): the epilogues that keep the invariant "zero from the row's own length on"
// themselves.  ragged_len: valid output samples of batch row b (everything when the batch is whole); ragged_mask4: the four
// consecutive samples t .. t + 3 of a row with the ones at or behind `len` zeroed.
__device__ __forceinline__ int ragged_len(const int* lens, int b) { return lens ? lens[b] : 0x7fffffff; }
__device__ __forceinline__ f32x4 ragged_mask4(f32x4 v, int t, int len) {
  const int n = len - t;
  v[0] = n > 0 ? v[0] : 0.f; v[1] = n > 1 ? v[1] : 0.f; v[2] = n > 2 ? v[2] : 0.f; v[3] = n > 3 ? v[3] : 0.f;
  return v;
}

// Input transform V = B^T d of the minimal-filtering kernels (conv_direct2w_kernel, conv_direct3w_kernel): d = KW + 1 consecutive
// samples of one channel, V = the KW + 1 values that meet U = G w (ou_model.cpp) in the element-wise products.
//   F(2, 3), points 0, 1, -1, inf:        rows [1 0 -1 0] [0 1 1 0] [0 -1 1 0] [0 1 0 -1]
//   F(2, 5), points 0, 1, -1, 1/2, -2, inf, rows scaled to small integers (the scale is in G):
//            [2 -3 -4 3 2 0] [0 -2 1 5 2 0] [0 -2 5 -1 -2 0] [0 2 1 -2 -1 0] [0 1 -2 -1 2 0] [0 2 -3 -4 3 2]
// A^T (applied to the accumulators): F(2, 3): y0 = M0 + M1 + M2, y1 = M1 - M2 - M3;
//   F(2, 5): y0 = M0 + M1 + M2 + M3 + M4, y1 = M1 - M2 + M3 / 2 - 2 M4 + M5.   tests/test_packing.py holds the three against
// the convolution they must reproduce.
template <int KW>
__device__ __forceinline__ void wino_bt(const float* X, float (&V)[KW + 1]) {
  if constexpr (KW == 3) {
    V[0] = X[0] - X[2];
    V[1] = X[1] + X[2];
    V[2] = X[2] - X[1];
    V[3] = X[1] - X[3];
  } else {
    const float p13 = X[1] - X[3], p24 = X[2] - X[4];
    V[3] = fmaf(2.f, p13, p24);
    V[4] = fmaf(-2.f, p24, p13);
    V[1] = fmaf(5.f, X[3], fmaf(2.f, X[4] - X[1], X[2]));
    V[2] = fmaf(5.f, X[2], fmaf(-2.f, X[1] + X[4], -X[3]));
    V[0] = fmaf(-4.f, X[2], fmaf(3.f, X[3] - X[1], 2.f * (X[0] + X[4])));
    V[5] = fmaf(-4.f, X[3], fmaf(3.f, X[4] - X[2], 2.f * (X[1] + X[5])));
  }
}

}  // namespace ou
