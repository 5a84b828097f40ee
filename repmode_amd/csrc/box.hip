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
#include "tail_jobs.h"

namespace {

// out = box3(in3) + box5(in5) [+ add0] [+ add1], stored as float or (OUT_BF16) bfloat16
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void box_sum_kernel(const float* __restrict__ in3, const float* __restrict__ in5,
                                                      const float* __restrict__ add0, const float* __restrict__ add1,
                                                      void* __restrict__ out_, int N, int D, int H, int W, int C) {
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
  for (int q = 0; q < 2; ++q) {
    const float* add = q ? add1 : add0;
    if (!add) continue;
    if (vec) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(add + o);
      r[0] += t.x; r[1] += t.y; r[2] += t.z; r[3] += t.w;
    } else {
      for (int k = 0; k < 4; ++k) if (c + k < C) r[k] += add[o + k];
    }
  }
  if constexpr (OUT_BF16) {
    bf16_t* out = static_cast<bf16_t*>(out_);
    if (vec) {
      *reinterpret_cast<u32x2*>(out + o) = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
    } else {
      for (int k = 0; k < 4; ++k) if (c + k < C) out[o + k] = f32_to_bf16(r[k]);
    }
  } else {
    float* out = static_cast<float*>(out_);
    if (vec) {
      *reinterpret_cast<f32x4*>(out + o) = f32x4{r[0], r[1], r[2], r[3]};
    } else {
      for (int k = 0; k < 4; ++k) if (c + k < C) out[o + k] = r[k];
    }
  }
}

}  // namespace

// Separable variant for volumes that fit in LDS (the deep levels, where this operator runs): a workgroup takes one
// sample and a group of CB channels, keeps the whole D x H x W volume in LDS and applies the three 1-D box sums
// (5 + 5 + 5 LDS reads per output instead of 125 gathered global loads -- the direct kernel above is bound by L1
// bandwidth: 152 float4 loads per output).
namespace {
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void box_sum_lds_kernel(const float* __restrict__ in3, const float* __restrict__ in5,
                                                          const float* __restrict__ add0, const float* __restrict__ add1,
                                                          void* __restrict__ out_, int D, int H, int W, int C, int CB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char box_smem[];
  const int V = D * H * W, c4n = CB / 4, items = V * c4n;
  f32x4* A = reinterpret_cast<f32x4*>(box_smem);
  f32x4* B = A + items;
  const int n = blockIdx.x, c0 = blockIdx.y * CB;
  const int tid = threadIdx.x;
  constexpr int MAXI = 16;                       // items per thread (launcher guarantees items <= 256 * MAXI)
  f32x4 res[MAXI];
#pragma unroll
  for (int j = 0; j < MAXI; ++j) res[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int pass = 0; pass < 2; ++pass) {
    const float* src = pass ? in5 : in3;
    if (!src) continue;                            // uniform
    const int r = pass ? 2 : 1;
    const float scale = pass ? 1.0f / 125.0f : 1.0f / 27.0f;
    __syncthreads();
    for (int i = tid; i < items; i += 256) {
      const int v = i / c4n, q = i % c4n;
      const int c = c0 + 4 * q;
      A[i] = c < C ? *reinterpret_cast<const f32x4*>(src + ((size_t)n * V + v) * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    for (int i = tid; i < items; i += 256) {       // along x: A -> B
      const int v = i / c4n, x = v % W;
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int d = -r; d <= r; ++d)
        if ((unsigned)(x + d) < (unsigned)W) t += A[i + d * c4n];
      B[i] = t;
    }
    __syncthreads();
    for (int i = tid; i < items; i += 256) {       // along y: B -> A
      const int v = i / c4n, y = (v / W) % H;
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int d = -r; d <= r; ++d)
        if ((unsigned)(y + d) < (unsigned)H) t += B[i + d * W * c4n];
      A[i] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {               // along z: A -> registers
      const int i = tid + j * 256;
      if (i < items) {
        const int v = i / c4n, z = v / (W * H);
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int d = -r; d <= r; ++d)
          if ((unsigned)(z + d) < (unsigned)D) t += A[i + d * H * W * c4n];
        res[j] += t * scale;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXI; ++j) {
    const int i = tid + j * 256;
    if (i >= items) continue;
    const int v = i / c4n, q = i % c4n;
    const int c = c0 + 4 * q;
    if (c >= C) continue;
    const size_t o = ((size_t)n * V + v) * C + c;
    f32x4 r = res[j];
    if (add0) r += *reinterpret_cast<const f32x4*>(add0 + o);
    if (add1) r += *reinterpret_cast<const f32x4*>(add1 + o);
    if constexpr (OUT_BF16) {
      *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(out_) + o) = u32x2{pack_bf16x2(r.x, r.y), pack_bf16x2(r.z, r.w)};
    } else {
      *reinterpret_cast<f32x4*>(static_cast<float*>(out_) + o) = r;
    }
  }
}
}  // namespace

// x (bf16 or float) -> out[0] = x, out[1] = box3(x), out[2] = box5(x), all float [n][v][c]: the inputs of the three 1x1
// experts' GEMM in ONE launch (a widening copy and two box launches before).  Same separable scheme as above.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void box_expand_lds_kernel(const T* __restrict__ x, float* __restrict__ out, size_t estride, int D,
                                                             int H, int W, int C, int CB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char box_smem[];
  const int V = D * H * W, c4n = CB / 4, items = V * c4n;
  f32x4* A = reinterpret_cast<f32x4*>(box_smem);
  f32x4* B = A + items;
  const int n = blockIdx.x, c0 = blockIdx.y * CB;
  const int tid = threadIdx.x;
  constexpr int MAXI = 16;
  for (int pass = 0; pass < 2; ++pass) {
    const int r = pass ? 2 : 1;
    const float scale = pass ? 1.0f / 125.0f : 1.0f / 27.0f;
    __syncthreads();
    for (int i = tid; i < items; i += 256) {
      const int v = i / c4n, q = i % c4n;
      const int c = c0 + 4 * q;
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < C) {
        const T* p = x + ((size_t)n * V + v) * C + c;
        if constexpr (sizeof(T) == 4) {
          t = *reinterpret_cast<const f32x4*>(p);
        } else {
          const u32x2 w = *reinterpret_cast<const u32x2*>(p);
          t = f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                    __uint_as_float(w.y & 0xffff0000u)};
        }
        if (pass == 0) *reinterpret_cast<f32x4*>(out + ((size_t)n * V + v) * C + c) = t;
      }
      A[i] = t;
    }
    __syncthreads();
    for (int i = tid; i < items; i += 256) {       // along x: A -> B
      const int v = i / c4n, xx = v % W;
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int d = -r; d <= r; ++d)
        if ((unsigned)(xx + d) < (unsigned)W) t += A[i + d * c4n];
      B[i] = t;
    }
    __syncthreads();
    for (int i = tid; i < items; i += 256) {       // along y: B -> A
      const int v = i / c4n, y = (v / W) % H;
      f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int d = -r; d <= r; ++d)
        if ((unsigned)(y + d) < (unsigned)H) t += B[i + d * W * c4n];
      A[i] = t;
    }
    __syncthreads();
    float* dst = out + (size_t)(pass + 1) * estride;
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {               // along z: A -> out
      const int i = tid + j * 256;
      if (i < items) {
        const int v = i / c4n, q = i % c4n, z = v / (W * H);
        const int c = c0 + 4 * q;
        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int d = -r; d <= r; ++d)
          if ((unsigned)(z + d) < (unsigned)D) t += A[i + d * H * W * c4n];
        if (c < C) *reinterpret_cast<f32x4*>(dst + ((size_t)n * V + v) * C + c) = t * scale;
      }
    }
  }
}
}  // namespace

// out: float [3][n][d][h][w][c].  Volumes that fit in LDS with c % 4 == 0 (the deep levels this runs on); returns
// REPMODE_EINVAL otherwise (the caller then uses a copy + repmode_box_sum_ex).
extern "C" int repmode_box_expand(const void* x, int dtype, float* out, int n, int d, int h, int w, int c, void* stream) {
  RM_REQUIRE(x && out, "box_expand: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && c > 0, "box_expand: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "box_expand: bad dtype %d", dtype);
  const long V = (long)d * h * w;
  int cb = 0;
  if ((c & 3) == 0)
    for (int t = 16; t >= 4; t >>= 1)
      if (V * t * 4 * 2 <= 60 * 1024 && V * (t / 4) <= 256 * 16) { cb = t; break; }
  RM_REQUIRE(cb > 0, "box_expand: the volume does not fit in LDS (or c %% 4 != 0)");
  const size_t lds = (size_t)V * cb * 4 * 2;
  const dim3 grid((unsigned)n, (unsigned)((c + cb - 1) / cb));
  const size_t estride = (size_t)n * V * c;
  if (dtype == REPMODE_BF16)
    hipLaunchKernelGGL(box_expand_lds_kernel<bf16_t>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), (const bf16_t*)x, out,
                       estride, d, h, w, c, cb);
  else
    hipLaunchKernelGGL(box_expand_lds_kernel<float>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), (const float*)x, out,
                       estride, d, h, w, c, cb);
  RM_LAUNCH_CHECK("box_expand");
  return REPMODE_OK;
}

extern "C" int repmode_box_sum_ex(const float* in3, const float* in5, const float* add0, const float* add1, void* out,
                                  int out_dtype, int n, int d, int h, int w, int c, void* stream) {
  RM_REQUIRE(out && (in3 || in5), "box_sum: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && c > 0, "box_sum: bad shape");
  RM_REQUIRE(out_dtype == REPMODE_F32 || out_dtype == REPMODE_BF16, "box_sum: bad dtype %d", out_dtype);
  // volumes that fit in LDS with at least 4 channels: separable kernel
  const long V = (long)d * h * w;
  int cb = 0;
  if ((c & 3) == 0)
    for (int t = 16; t >= 4; t >>= 1)
      if (V * t * 4 * 2 <= 60 * 1024 && V * (t / 4) <= 256 * 16) { cb = t; break; }
  if (cb) {
    const size_t lds = (size_t)V * cb * 4 * 2;
    const dim3 grid((unsigned)n, (unsigned)((c + cb - 1) / cb));
    if (out_dtype == REPMODE_BF16)
      hipLaunchKernelGGL(box_sum_lds_kernel<true>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), in3, in5, add0,
                         add1, out, d, h, w, c, cb);
    else
      hipLaunchKernelGGL(box_sum_lds_kernel<false>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), in3, in5, add0,
                         add1, out, d, h, w, c, cb);
    RM_LAUNCH_CHECK("box_sum(lds)");
    return REPMODE_OK;
  }
  const long total = (long)n * d * h * w * ((c + 3) / 4);
  if (out_dtype == REPMODE_BF16)
    hipLaunchKernelGGL(box_sum_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in3, in5, add0, add1, out, n, d, h, w, c);
  else
    hipLaunchKernelGGL(box_sum_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), in3, in5, add0, add1, out, n, d, h, w, c);
  RM_LAUNCH_CHECK("box_sum");
  return REPMODE_OK;
}

extern "C" int repmode_box_sum(const float* in3, const float* in5, float* out, int n, int d, int h, int w, int c,
                               void* stream) {
  return repmode_box_sum_ex(in3, in5, nullptr, nullptr, out, REPMODE_F32, n, d, h, w, c, stream);
}

// ------------------------------------------------------------------------------------------------
// Filter gradient, tap-major -> the experts' parameter layout: out[m][t] = in[tap(t)][m], m = (co, ci) pairs.
//   ntaps_out == 125: all taps;  ntaps_out == 27: the centred 3x3x3 taps of the 5x5x5 grid (in is still [125][M]);
//   ntaps_out == 8: the 2x2x2 stride-2 filters (in is [8][M]).
// A 64-column strip goes through LDS so that both the reads (64 floats of a tap row) and the writes (64 * ntaps
// contiguous floats) are whole lines; the generic strided copy it replaces ran at a quarter of that.
namespace {
__global__ __launch_bounds__(256) void tap_transpose_kernel(const float* __restrict__ in, float* __restrict__ out, long M,
                                                            int ntaps_out) {
  __shared__ float tile[125 * 65];
  // (the body is shared with the deferred form that rides in a conv5 launch: tail_jobs.h)
  tail_tap_transpose(in, out, M, ntaps_out, tile, blockIdx.x, threadIdx.x);
}
}  // namespace

extern "C" int repmode_tap_transpose_ex(const float* in, float* out, long m, int ntaps_out, int flags, void* stream) {
  RM_REQUIRE(in && out, "tap_transpose: null pointer");
  RM_REQUIRE(m > 0 && (ntaps_out == 125 || ntaps_out == 27 || ntaps_out == 8), "tap_transpose: bad shape");
  if (flags & REPMODE_DEFER) {
    TailJob j{};
    j.kind = 2;
    j.nblocks = (int)((m + 63) / 64);
    j.in0 = in; j.io1 = out; j.m = m; j.p0 = ntaps_out;
    return repmode_tail_push(j, static_cast<hipStream_t>(stream));
  }
  hipLaunchKernelGGL(tap_transpose_kernel, dim3((unsigned)((m + 63) / 64)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     in, out, m, ntaps_out);
  RM_LAUNCH_CHECK("tap_transpose");
  return REPMODE_OK;
}

extern "C" int repmode_tap_transpose(const float* in, float* out, long m, int ntaps_out, void* stream) {
  return repmode_tap_transpose_ex(in, out, m, ntaps_out, 0, stream);
}
