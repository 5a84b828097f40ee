// box_body.h -- pieces of the separable zero-padded box sums (box.hip) that a second translation unit uses too
// (expert_mix.hip: the gate mix's backward and the avg-pool experts' box means from one launch).
#pragma once
#include "common.h"

namespace {
constexpr int BOX_MAXI = 16;                     // items per thread (launcher guarantees items <= 256 * BOX_MAXI)

struct BoxItems {
  int n_items, c4n, W, H, D;
  uint32_t xyz[BOX_MAXI];                        // x | y << 10 | z << 20 of item tid + 256 j
  __device__ __forceinline__ void init(int V, int c4n_, int D_, int H_, int W_, int tid) {
    n_items = V * c4n_; c4n = c4n_; W = W_; H = H_; D = D_;
#pragma unroll
    for (int j = 0; j < BOX_MAXI; ++j) {
      const int i = tid + j * 256;
      const int v = i / c4n, x = v % W, t = v / W, y = t % H, z = t / H;
      xyz[j] = (uint32_t)x | ((uint32_t)y << 10) | ((uint32_t)z << 20);
    }
  }
};

// out[i] = sum over |d| <= R of in[i + d * step] where coordinate + d stays inside [0, extent)
template <int R>
__device__ __forceinline__ f32x4 box_line(const f32x4* in, int i, int step, int coord, int extent) {
  f32x4 t = in[i];
#pragma unroll
  for (int d = 1; d <= R; ++d) {
    if (coord - d >= 0) t += in[i - d * step];
    if (coord + d < extent) t += in[i + d * step];
  }
  return t;
}

}  // namespace
