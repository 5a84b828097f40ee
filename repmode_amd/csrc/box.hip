// box.hip -- zero-padded box means for the average-pool experts.
//
// The reference turns its two avg-pool experts into 5x5x5 filters: w1x1[o][i] * 1/k^3 broadcast over the
// centred k^3 support (fnet/nn_modules/RepMode.py:139-142, 161-163, 176-180).  By linearity
// conv(x, w1x1 (x) box_k) = w1x1 applied to box_k(x), so the per-expert ("unmerged") formulation used on
// the deep levels needs box_3(x) and box_5(x) (zero padding, divided by 27 / 125) as inputs of plain
// GEMMs, and the same operator (it is self-adjoint) in the backward pass.
//
//     out = box3(in3) + box5(in5)           (either input may be NULL)
//
// Channels-last float tensors [N][D][H][W][C].  The deep-level tensors this runs on are a few MB and
// live in L2, so the kernel is a direct gather: one thread per (voxel, 4 channels), 27 / 125 float4
// loads.  Memory-bound on L2, a few tens of microseconds; not on the MFMA path.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void box_sum_kernel(const float* __restrict__ in3, const float* __restrict__ in5,
                                                      float* __restrict__ out, int N, int D, int H, int W, int C) {
  const int c4n = (C + 3) / 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)N * D * H * W * c4n;
  if (idx >= total) return;
  const int c = (int)(idx % c4n) * 4;
  long v = idx / c4n;
  const int x = (int)(v % W); v /= W;
  const int y = (int)(v % H); v /= H;
  const int z = (int)(v % D);
  const int n = (int)(v / D);
  const bool vec = (C & 3) == 0;
  float a3[4] = {0.f, 0.f, 0.f, 0.f}, a5[4] = {0.f, 0.f, 0.f, 0.f};
  for (int dz = -2; dz <= 2; ++dz) {
    const int zi = z + dz;
    if ((unsigned)zi >= (unsigned)D) continue;
    for (int dy = -2; dy <= 2; ++dy) {
      const int yi = y + dy;
      if ((unsigned)yi >= (unsigned)H) continue;
      for (int dx = -2; dx <= 2; ++dx) {
        const int xi = x + dx;
        if ((unsigned)xi >= (unsigned)W) continue;
        const size_t off = ((((size_t)n * D + zi) * H + yi) * W + xi) * C + c;
        const bool inner = dz >= -1 && dz <= 1 && dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1;
        if (in5) {
          if (vec) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(in5 + off);
            a5[0] += t.x; a5[1] += t.y; a5[2] += t.z; a5[3] += t.w;
          } else {
            for (int k = 0; k < 4; ++k) if (c + k < C) a5[k] += in5[off + k];
          }
        }
        if (in3 && inner) {
          if (vec) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(in3 + off);
            a3[0] += t.x; a3[1] += t.y; a3[2] += t.z; a3[3] += t.w;
          } else {
            for (int k = 0; k < 4; ++k) if (c + k < C) a3[k] += in3[off + k];
          }
        }
      }
    }
  }
  const size_t o = ((((size_t)n * D + z) * H + y) * W + x) * C + c;
  float r[4];
  for (int k = 0; k < 4; ++k) r[k] = a3[k] * (1.0f / 27.0f) + a5[k] * (1.0f / 125.0f);
  if (vec) {
    *reinterpret_cast<f32x4*>(out + o) = f32x4{r[0], r[1], r[2], r[3]};
  } else {
    for (int k = 0; k < 4; ++k) if (c + k < C) out[o + k] = r[k];
  }
}

}  // namespace

extern "C" int repmode_box_sum(const float* in3, const float* in5, float* out, int n, int d, int h, int w, int c,
                               void* stream) {
  RM_REQUIRE(out && (in3 || in5), "box_sum: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && c > 0, "box_sum: bad shape");
  const long total = (long)n * d * h * w * ((c + 3) / 4);
  hipLaunchKernelGGL(box_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     in3, in5, out, n, d, h, w, c);
  RM_LAUNCH_CHECK("box_sum");
  return REPMODE_OK;
}
