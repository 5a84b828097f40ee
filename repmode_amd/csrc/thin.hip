// thin.hip -- the 1-channel ends of the network (first layer Cin = 1, last layer Cout = 1) on the general conv kernel.
//
// With one input (output) channel the implicit GEMM of conv5_igemm.hip pads the reduction (row) dimension from
// 1 to 16 (32): 94-97 % of its MFMA work multiplies zeros.  The five x taps are folded into that dimension instead:
//
//   Cin  == 1:  x5[v][dx] = x[v + (0,0,dx-2)]          -> a 5-channel input; the filter keeps its 25 (dz,dy) taps
//               and gets red = dx:  W'[(dz,dy)][co][dx] = W[(dz,dy,dx)][co][0]
//   Cout == 1:  W'[(dz,dy)][row = dx][ci] = W[(dz,dy,dx)][0][ci] -> a 5-row output y5[u][dx] (conv at x position u
//               with the centre x tap only), then  y[x] = sum_dx y5[x + dx - 2][dx]
//
// and the conv runs in its "dx centre" mode (repmode_conv5_ex flag bit 2): 25 taps instead of 125.  The three helper
// kernels here are the folding (shift5), the un-folding (unshift5) and the filter re-pack (thin_pack); each is one
// small memory-bound launch.
#include "common.h"

namespace {

template <typename TI>
__global__ void shift5_kernel(const TI* __restrict__ x, bf16_t* __restrict__ x5, long nrows, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * W) return;
  const int xx = (int)(i % W);
  const TI* row = x + (i - xx);
  float v[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const int xi = xx + d - 2;
    float t = 0.f;
    if ((unsigned)xi < (unsigned)W) {
      if constexpr (sizeof(TI) == 2) t = bf16_to_f32(row[xi]);
      else t = row[xi];
    }
    v[d] = t;
  }
  *reinterpret_cast<u32x4*>(x5 + i * 8) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], 0.f), 0u};
}

__global__ void unshift5_kernel(const float* __restrict__ y5, float* __restrict__ y, long nrows, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * W) return;
  const int xx = (int)(i % W);
  float t = 0.f;
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    const int u = xx + d - 2;
    if ((unsigned)u < (unsigned)W) t += y5[(i - xx + u) * 5 + d];
  }
  y[i] = t;
}

// w, out: [slot][125][ntiles][32 rows][16 red] bf16 (fragment-major; ntiles = row tiles x reduction chunks, of
// which the thin dimension has one).  Only the 25 taps (dz, dy, dx = 2) of out are written.
//   to_rows == 0: out(tap2, tile, r, red = dx) = w(tap(dx), tile, r, 0)        (thin reduction dim, tiles = row tiles)
//   to_rows != 0: out(tap2, tile, row = dx, k) = w(tap(dx), tile, 0, k)        (thin row dim, tiles = reduction chunks)
__global__ void thin_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int nslots, int ntiles,
                                 int to_rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (slot, zy, tile, r, k)
  if (i >= (long)nslots * 25 * ntiles * 512) return;
  const int k = (int)(i & 15), r = (int)((i >> 4) & 31);
  long t = i >> 9;
  const int tile = (int)(t % ntiles); t /= ntiles;
  const int zy = (int)(t % 25), slot = (int)(t / 25);
  const size_t tap_stride = (size_t)ntiles * 512;
  const size_t base = (size_t)slot * 125 * tap_stride + (size_t)tile * 512;
  const int dsel = to_rows ? r : k;                          // which dx this destination element takes
  bf16_t v = 0;
  if (dsel < 5) {
    const int tap = zy * 5 + dsel;
    v = to_rows ? w[base + (size_t)tap * tap_stride + k] : w[base + (size_t)tap * tap_stride + r * 16];
  }
  out[base + (size_t)(zy * 5 + 2) * tap_stride + r * 16 + k] = v;
}

}  // namespace

extern "C" int repmode_shift5(const void* x, int dtype, void* x5, long nrows, int w, void* stream) {
  RM_REQUIRE(x && x5 && nrows > 0 && w > 0, "shift5: bad argument");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "shift5: bad dtype %d", dtype);
  const long total = nrows * w;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == REPMODE_F32)
    hipLaunchKernelGGL(shift5_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)x, (bf16_t*)x5, nrows, w);
  else
    hipLaunchKernelGGL(shift5_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)x5, nrows, w);
  RM_LAUNCH_CHECK("shift5");
  return REPMODE_OK;
}

extern "C" int repmode_unshift5(const float* y5, float* y, long nrows, int w, void* stream) {
  RM_REQUIRE(y5 && y && nrows > 0 && w > 0, "unshift5: bad argument");
  const long total = nrows * w;
  hipLaunchKernelGGL(unshift5_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), y5, y, nrows, w);
  RM_LAUNCH_CHECK("unshift5");
  return REPMODE_OK;
}

extern "C" int repmode_thin_pack(const void* w, void* out, int nslots, int ntiles, int to_rows, void* stream) {
  RM_REQUIRE(w && out && nslots > 0 && ntiles > 0, "thin_pack: bad argument");
  const long total = (long)nslots * 25 * ntiles * 512;
  hipLaunchKernelGGL(thin_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const bf16_t*)w, (bf16_t*)out, nslots, ntiles, to_rows);
  RM_LAUNCH_CHECK("thin_pack");
  return REPMODE_OK;
}
