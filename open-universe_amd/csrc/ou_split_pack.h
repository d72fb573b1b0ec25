// Host side of conv_split_kernel (ou_conv_split.hip): an fp32 weight as three bf16 pieces, laid out as MFMA A fragments.
// Shared by the packer (ou_model.cpp) and the microbenchmark (tools/ubench/split_conv.hip).  Pure host C++.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace ou {

// round-to-nearest-even fp32 -> bf16 (bits); the pieces of finite weights never overflow
inline uint16_t bf16_rne(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// w = hi + mid + lo exactly (24 significand bits = 3 x 8) unless w is subnormal-adjacent; the subtractions are exact
inline void split3(float w, uint16_t (&pc)[3]) {
  pc[0] = bf16_rne(w);
  const float r1 = w - bf16_to_f32(pc[0]);
  pc[1] = bf16_rne(r1);
  const float r2 = r1 - bf16_to_f32(pc[1]);
  pc[2] = bf16_rne(r2);
}
// floats the split copy of a layer occupies in the blob (2 bf16 per float slot)
inline size_t split_floats(int Cin, int KW, int Mp) { return (size_t)Cin * KW * Mp * 3 / 2; }

// W[(m * Cin + ci) * KW + k] (M rows, any float-convertible type) -> dst[Cin / 16][KW][Mp / 32][3][64 lanes][8] bf16,
// lane l: row 32 mt + (l & 31), channels 16 cc + 8 (l >> 5) + j.  Rows M .. Mp - 1 are zero.
template <typename T>
inline void pack_split(const T* W, int M, int Mp, int Cin, int KW, uint16_t* dst) {
  const int MT = Mp / 32;
  for (int cc = 0; cc < Cin / 16; cc++)
    for (int k = 0; k < KW; k++)
      for (int mt = 0; mt < MT; mt++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 8; j++) {
            const int row = 32 * mt + (l & 31), ci = 16 * cc + 8 * (l >> 5) + j;
            uint16_t pc[3] = {0, 0, 0};
            if (row < M) split3((float)W[((size_t)row * Cin + ci) * KW + k], pc);
            const size_t frag = (((size_t)cc * KW + k) * MT + mt) * 3;
            for (int q = 0; q < 3; q++) dst[((frag + q) * 64 + l) * 8 + j] = pc[q];
          }
}

}  // namespace ou
