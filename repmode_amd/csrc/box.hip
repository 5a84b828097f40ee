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
#include "box_body.h"

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
// Round 4: BOTH box sums go through ONE sequence of phases (load, x, y, z: three barriers) with both inputs requested up
// front and every item's (x, y, z) decoded once -- these launches move a few MB on 128-256 workgroups, so their time is the
// length of the dependent chain: two passes of load / x / y / z with eight barriers and an index decode per item and phase
// were 13-18 us per launch.
namespace {
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void box_sum_lds_kernel(const float* __restrict__ in3, const float* __restrict__ in5,
                                                          const float* __restrict__ add0, const float* __restrict__ add1,
                                                          void* __restrict__ out_, float* __restrict__ out5_, int D, int H, int W, int C,
                                                          int CB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char box_smem[];
  const int V = D * H * W, c4n = CB / 4, items = V * c4n;
  f32x4* A3 = reinterpret_cast<f32x4*>(box_smem);
  f32x4* B3 = A3 + items;
  f32x4* A5 = B3 + items;
  f32x4* B5 = A5 + items;
  const int n = blockIdx.x, c0 = blockIdx.y * CB;
  const int tid = threadIdx.x;
  BoxItems it;
  it.init(V, c4n, D, H, W, tid);
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- both inputs, all of this thread's items, before anything waits
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {
    const int i = tid + j * 256;
    if (i < items) {
      const int v = i / c4n, q = i % c4n, c = c0 + 4 * q;
      const size_t o = ((size_t)n * V + v) * C + c;
      const f32x4 t3 = (in3 && c < C) ? *reinterpret_cast<const f32x4*>(in3 + o) : zero;
      const f32x4 t5 = (in5 && c < C) ? *reinterpret_cast<const f32x4*>(in5 + o) : zero;
      A3[i] = t3;
      A5[i] = t5;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {             // along x: A -> B
    const int i = tid + j * 256;
    if (i < items) {
      const int x = (int)(it.xyz[j] & 1023u);
      if (in3) B3[i] = box_line<1>(A3, i, c4n, x, W);
      if (in5) B5[i] = box_line<2>(A5, i, c4n, x, W);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {             // along y: B -> A
    const int i = tid + j * 256;
    if (i < items) {
      const int y = (int)((it.xyz[j] >> 10) & 1023u);
      if (in3) A3[i] = box_line<1>(B3, i, W * c4n, y, H);
      if (in5) A5[i] = box_line<2>(B5, i, W * c4n, y, H);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {             // along z: A -> out
    const int i = tid + j * 256;
    if (i >= items) continue;
    const int v = i / c4n, q = i % c4n, c = c0 + 4 * q;
    if (c >= C) continue;
    const int z = (int)(it.xyz[j] >> 20);
    const size_t o = ((size_t)n * V + v) * C + c;
    if (out5_) {      // the two box means apart (repmode_box_pair): float outputs, nothing added
      *reinterpret_cast<f32x4*>(static_cast<float*>(out_) + o) = box_line<1>(A3, i, H * W * c4n, z, D) * (1.0f / 27.0f);
      *reinterpret_cast<f32x4*>(out5_ + o) = box_line<2>(A5, i, H * W * c4n, z, D) * (1.0f / 125.0f);
      continue;
    }
    f32x4 r = zero;
    if (in3) r += box_line<1>(A3, i, H * W * c4n, z, D) * (1.0f / 27.0f);
    if (in5) r += box_line<2>(A5, i, H * W * c4n, z, D) * (1.0f / 125.0f);
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
// experts' GEMM in ONE launch (a widening copy and two box launches before).  Same separable scheme as above; x is staged
// once and feeds both box sums.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void box_expand_lds_kernel(const T* __restrict__ x, float* __restrict__ out, size_t estride, int D,
                                                             int H, int W, int C, int CB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char box_smem[];
  const int V = D * H * W, c4n = CB / 4, items = V * c4n;
  f32x4* A = reinterpret_cast<f32x4*>(box_smem);         // x, later the 5-sum along (x, y)
  f32x4* B3 = A + items;
  f32x4* B5 = B3 + items;
  f32x4* A3 = B5 + items;
  const int n = blockIdx.x, c0 = blockIdx.y * CB;
  const int tid = threadIdx.x;
  BoxItems it;
  it.init(V, c4n, D, H, W, tid);
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {
    const int i = tid + j * 256;
    if (i < items) {
      const int v = i / c4n, q = i % c4n, c = c0 + 4 * q;
      f32x4 t = zero;
      if (c < C) {
        const T* p = x + ((size_t)n * V + v) * C + c;
        if constexpr (sizeof(T) == 4) {
          t = *reinterpret_cast<const f32x4*>(p);
        } else {
          const u32x2 w = *reinterpret_cast<const u32x2*>(p);
          t = f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                    __uint_as_float(w.y & 0xffff0000u)};
        }
        *reinterpret_cast<f32x4*>(out + ((size_t)n * V + v) * C + c) = t;
      }
      A[i] = t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {             // along x: A -> B3, B5
    const int i = tid + j * 256;
    if (i < items) {
      const int xx = (int)(it.xyz[j] & 1023u);
      const f32x4 t3 = box_line<1>(A, i, c4n, xx, W);
      f32x4 t5 = t3;                                  // the 5-sum = the 3-sum + the two outer taps
      if (xx - 2 >= 0) t5 += A[i - 2 * c4n];
      if (xx + 2 < W) t5 += A[i + 2 * c4n];
      B3[i] = t3;
      B5[i] = t5;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {             // along y: B3 -> A3, B5 -> A
    const int i = tid + j * 256;
    if (i < items) {
      const int y = (int)((it.xyz[j] >> 10) & 1023u);
      A3[i] = box_line<1>(B3, i, W * c4n, y, H);
      A[i] = box_line<2>(B5, i, W * c4n, y, H);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {             // along z: -> out[1], out[2]
    const int i = tid + j * 256;
    if (i >= items) continue;
    const int v = i / c4n, q = i % c4n, c = c0 + 4 * q;
    if (c >= C) continue;
    const int z = (int)(it.xyz[j] >> 20);
    const size_t o = ((size_t)n * V + v) * C + c;
    *reinterpret_cast<f32x4*>(out + estride + o) = box_line<1>(A3, i, H * W * c4n, z, D) * (1.0f / 27.0f);
    *reinterpret_cast<f32x4*>(out + 2 * estride + o) = box_line<2>(A, i, H * W * c4n, z, D) * (1.0f / 125.0f);
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
      if (V * t * 4 * 4 <= 64 * 1024 && V * (t / 4) <= 256 * BOX_MAXI && d < 1024 && h < 1024 && w < 1024) { cb = t; break; }
  RM_REQUIRE(cb > 0, "box_expand: the volume does not fit in LDS (or c %% 4 != 0)");
  const size_t lds = (size_t)V * cb * 4 * 4;          // four staging buffers
  const dim3 grid((unsigned)n, (unsigned)((c + cb - 1) / cb));
  const size_t estride = (size_t)n * V * c;
  repmode_prof_begin(REPMODE_PROF_HELPER, (double)estride * (dtype == REPMODE_BF16 ? 14.0 : 16.0), static_cast<hipStream_t>(stream));
  if (dtype == REPMODE_BF16)
    hipLaunchKernelGGL(box_expand_lds_kernel<bf16_t>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), (const bf16_t*)x, out,
                       estride, d, h, w, c, cb);
  else
    hipLaunchKernelGGL(box_expand_lds_kernel<float>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), (const float*)x, out,
                       estride, d, h, w, c, cb);
  repmode_prof_end(static_cast<hipStream_t>(stream));
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
  repmode_prof_begin(REPMODE_PROF_HELPER, (double)n * V * c * 4.0 * ((in3 ? 1 : 0) + (in5 ? 1 : 0) + (add0 ? 1 : 0) + (add1 ? 1 : 0) + 1),
                     static_cast<hipStream_t>(stream));
  int cb = 0;
  if ((c & 3) == 0)
    for (int t = 16; t >= 4; t >>= 1)
      if (V * t * 4 * 4 <= 64 * 1024 && V * (t / 4) <= 256 * BOX_MAXI && d < 1024 && h < 1024 && w < 1024) { cb = t; break; }
  if (cb) {
    const size_t lds = (size_t)V * cb * 4 * 4;          // four staging buffers
    const dim3 grid((unsigned)n, (unsigned)((c + cb - 1) / cb));
    if (out_dtype == REPMODE_BF16)
      hipLaunchKernelGGL(box_sum_lds_kernel<true>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), in3, in5, add0,
                         add1, out, (float*)nullptr, d, h, w, c, cb);
    else
      hipLaunchKernelGGL(box_sum_lds_kernel<false>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), in3, in5, add0,
                         add1, out, (float*)nullptr, d, h, w, c, cb);
    repmode_prof_end(static_cast<hipStream_t>(stream));
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
  repmode_prof_end(static_cast<hipStream_t>(stream));
  RM_LAUNCH_CHECK("box_sum");
  return REPMODE_OK;
}

// out3 = box3(in3) / 27, out5 = box5(in5) / 125 (zero padding), float [n][d][h][w][c] each, in ONE launch: the avg-pool
// experts' operands of repmode_deep_mode_dgrad (the box mean commutes with the 1x1 channel mixing and with the gate scale).
// Volumes that fit in LDS with c % 4 == 0 (the deep levels); REPMODE_EINVAL otherwise.
extern "C" int repmode_box_pair(const float* in3, const float* in5, float* out3, float* out5, int n, int d, int h, int w, int c,
                                void* stream) {
  RM_REQUIRE(in3 && in5 && out3 && out5, "box_pair: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && w > 0 && c > 0, "box_pair: bad shape");
  const long V = (long)d * h * w;
  int cb = 0;
  if ((c & 3) == 0)
    for (int t = 16; t >= 4; t >>= 1)
      if (V * t * 4 * 4 <= 64 * 1024 && V * (t / 4) <= 256 * BOX_MAXI && d < 1024 && h < 1024 && w < 1024) { cb = t; break; }
  RM_REQUIRE(cb > 0, "box_pair: the volume does not fit in LDS (or c %% 4 != 0)");
  const size_t lds = (size_t)V * cb * 4 * 4;
  const dim3 grid((unsigned)n, (unsigned)((c + cb - 1) / cb));
  repmode_prof_begin(REPMODE_PROF_HELPER, (double)n * V * c * 16.0, static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(box_sum_lds_kernel<false>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), in3, in5,
                     (const float*)nullptr, (const float*)nullptr, (void*)out3, out5, d, h, w, c, cb);
  repmode_prof_end(static_cast<hipStream_t>(stream));
  RM_LAUNCH_CHECK("box_pair");
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

// ------------------------------------------------------------------------------------------------
// out[row] = a[row] | b[row] along the channel axis of two channels-last tensors (RepMode.py:106 torch.cat((skip, up), 1) for
// the block that takes the per-expert formulation: its kernels read ONE input tensor).  16-byte pieces.
namespace {
__global__ __launch_bounds__(256) void concat2_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ out,
                                                      uint32_t total, uint32_t ca, uint32_t cb) {
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= total) return;
  const uint32_t ct = ca + cb, row = idx / ct, c = idx % ct;
  out[idx] = c < ca ? a[(size_t)row * ca + c] : b[(size_t)row * cb + (c - ca)];
}
}  // namespace

// a: [rows][ca_bytes], b: [rows][cb_bytes] -> out: [rows][ca_bytes + cb_bytes]; byte counts multiples of 16.
extern "C" int repmode_concat_channels(const void* a, const void* b, void* out, long rows, int ca_bytes, int cb_bytes, void* stream) {
  RM_REQUIRE(a && b && out && rows > 0 && ca_bytes > 0 && cb_bytes > 0, "concat_channels: bad argument");
  RM_REQUIRE((ca_bytes & 15) == 0 && (cb_bytes & 15) == 0, "concat_channels: row pieces must be multiples of 16 bytes (%d, %d)", ca_bytes, cb_bytes);
  RM_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "concat_channels: pointers must be 16-byte aligned");
  const long total = rows * ((ca_bytes + cb_bytes) / 16);
  RM_REQUIRE(total < (1L << 32) - 256, "concat_channels: %ld pieces (32-bit index arithmetic)", total);
  hipStream_t s = static_cast<hipStream_t>(stream);
  repmode_prof_begin(REPMODE_PROF_HELPER, 2.0 * (double)rows * (ca_bytes + cb_bytes), s);
  hipLaunchKernelGGL(concat2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, static_cast<const u32x4*>(a),
                     static_cast<const u32x4*>(b), static_cast<u32x4*>(out), (uint32_t)total, (uint32_t)(ca_bytes / 16), (uint32_t)(cb_bytes / 16));
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("concat_channels");
  return REPMODE_OK;
}
