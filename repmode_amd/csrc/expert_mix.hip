// expert_mix.hip -- gate mixing of the per-expert ("unmerged") formulation used on the deep levels:
//
//     y[n][v][o] = sum_e g[n][e][o] * P[e][n][v][o]                        (forward,  RepMode.py:184-188 by linearity)
//     dg[n][e][o] = sum_v dy[n][v][o] * P[e][n][v][o]                      (backward: gate-probability gradients)
//     dye[e][n][v][o] = g[n][e][o] * dy[n][v][o]                           (backward: gate-scaled output gradients,
//                        experts 0-1 (the 5^3 / 3^3 convs) in the conv kernels' element type, 2-4 in float)
//
// P_e = conv(x, K_e) are the five expert outputs (float, [5][N][V][Co]).  One pass over P in each direction
// instead of a chain of broadcast-multiply / reduce / cast kernels with [5][N][V][Co] temporaries.
#include "common.h"
#include "box_body.h"

#ifndef RM_MIX_ITERS
#define RM_MIX_ITERS 4
#endif

namespace {

constexpr int E = REPMODE_NUM_EXPERTS;

__global__ __launch_bounds__(256) void expert_mix_fwd_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ y, int N, long V, int C) {
  // (32-bit index arithmetic: the launcher checks the element count; three 64-bit divisions were most of this kernel's
  // instructions)
  const uint32_t c4n = (uint32_t)(C + 3) / 4u;
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t total = (uint32_t)N * (uint32_t)V * c4n;
  if (idx >= total) return;
  const int c = (int)(idx % c4n) * 4;
  const uint32_t nv = idx / c4n;
  const int n = (int)(nv / (uint32_t)V);
  const size_t estride = (size_t)N * V * C;
  const size_t off = (size_t)nv * C + c;
  const bool vec = (C & 3) == 0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const float* gp = g + ((size_t)n * E + e) * C + c;
    const float* pp = p + e * estride + off;
    if (vec) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pp), gv = *reinterpret_cast<const f32x4*>(gp);
      acc[0] += gv.x * pv.x; acc[1] += gv.y * pv.y; acc[2] += gv.z * pv.z; acc[3] += gv.w * pv.w;
    } else {
      for (int k = 0; k < 4; ++k) if (c + k < C) acc[k] += gp[k] * pp[k];
    }
  }
  if (vec) *reinterpret_cast<f32x4*>(y + off) = f32x4{acc[0], acc[1], acc[2], acc[3]};
  else for (int k = 0; k < 4; ++k) if (c + k < C) y[off + k] = acc[k];
}

template <typename T>
__device__ __forceinline__ void store4(T* p, const float* v, int c, int C, bool vec);
template <>
__device__ __forceinline__ void store4<float>(float* p, const float* v, int c, int C, bool vec) {
  if (vec) *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  else for (int k = 0; k < 4; ++k) if (c + k < C) p[k] = v[k];
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* v, int c, int C, bool vec) {
  if (vec) *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  else for (int k = 0; k < 4; ++k) if (c + k < C) p[k] = f32_to_bf16(v[k]);
}

// grid = (voxel chunks, N).  A thread owns 4 channels and strides over the voxels of its chunk; the 5 x 4 partial
// gate gradients are reduced over the workgroup through LDS atomics, one global atomic per (e, channel) and workgroup.
template <typename T>
__device__ __forceinline__ void mix_bwd_body(const float* __restrict__ dy, const float* __restrict__ p, const float* __restrict__ g,
                                             float* __restrict__ dg, T* __restrict__ dye_lo, float* __restrict__ dye_hi, int N, long V,
                                             int C, long vchunk, long hi_stride, int det, int bx, int by) {
  __shared__ float red[E * 512];
  const int c4n = (C + 3) / 4;
  const int n = by;
  const int cg = threadIdx.x % c4n, r0 = threadIdx.x / c4n;
  const int rows = 256 / c4n;                       // voxel rows covered per iteration (C <= 512 -> c4n <= 128)
  const int c = cg * 4;
  const bool vec = (C & 3) == 0;
  const bool active = r0 < rows;
  const size_t estride = (size_t)N * V * C;
  float gv[E][4], part[E][4];
#pragma unroll
  for (int e = 0; e < E; ++e)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gv[e][k] = (c + k < C) ? g[((size_t)n * E + e) * C + c + k] : 0.f;
      part[e][k] = 0.f;
    }
  const long v_begin = (long)bx * vchunk, v_end = min(V, v_begin + vchunk);
  if (active) {
    for (long v = v_begin + r0; v < v_end; v += rows) {
      const size_t off = ((size_t)n * V + v) * C + c;
      float d[4];
      if (vec) { const f32x4 t = *reinterpret_cast<const f32x4*>(dy + off); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
      else for (int k = 0; k < 4; ++k) d[k] = (c + k < C) ? dy[off + k] : 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        float pv[4], o[4];
        if (vec) { const f32x4 t = *reinterpret_cast<const f32x4*>(p + e * estride + off); pv[0] = t.x; pv[1] = t.y; pv[2] = t.z; pv[3] = t.w; }
        else for (int k = 0; k < 4; ++k) pv[k] = (c + k < C) ? p[e * estride + off + k] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { part[e][k] += d[k] * pv[k]; o[k] = gv[e][k] * d[k]; }
        if (e < 2) store4<T>(dye_lo + e * estride + off, o, c, C, vec);
        else store4<float>(dye_hi + (e - 2) * (size_t)hi_stride + off, o, c, C, vec);
      }
    }
  }
  if (det) {
    // deterministic mode (at most two workgroups per sample): the row groups' partial sums are added in row order, and the
    // workgroup's total is ONE add onto the cleared dg
    __shared__ float all[256 * E * 4];                    // [row r0][e][channel group cg][4]: rows * 5 * C <= 5120 floats
    if (active) {
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int k = 0; k < 4; ++k) all[((size_t)(r0 * E + e) * c4n + cg) * 4 + k] = part[e][k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E * C; i += 256) {
      const int e = i / C, ch = i % C;
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += all[((size_t)(r * E + e) * c4n + ch / 4) * 4 + (ch & 3)];
      atomicAdd(dg + (size_t)n * E * C + i, t);
    }
    return;
  }
  for (int i = threadIdx.x; i < E * C; i += 256) red[i] = 0.f;
  __syncthreads();
  if (active) {
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
      for (int k = 0; k < 4; ++k) if (c + k < C) atomicAdd(&red[e * C + c + k], part[e][k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E * C; i += 256) atomicAdd(dg + (size_t)n * E * C + i, red[i]);
}

template <typename T>
__global__ __launch_bounds__(256) void expert_mix_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ p,
                                                             const float* __restrict__ g, float* __restrict__ dg,
                                                             T* __restrict__ dye_lo, float* __restrict__ dye_hi, int N,
                                                             long V, int C, long vchunk, long hi_stride, int det) {
  mix_bwd_body<T>(dy, p, g, dg, dye_lo, dye_hi, N, V, C, vchunk, hi_stride, det, blockIdx.x, blockIdx.y);
}

// The avg-pool experts' operands of the one-launch data gradient (deep_mode.hip), from dy directly:
//     hb3[n][v][c] = g[n][3][c] * box3(dy)[n][v][c] / 27,   hb5[n][v][c] = g[n][4][c] * box5(dy)[n][v][c] / 125
// (the gate probability is constant over a sample's voxels, so box(g * dy) = g * box(dy): no need to wait for the gate-scaled
// tensors the mix writes).  One workgroup = one sample x CB channels, the whole volume in LDS, the separable sums of
// box.hip's box_expand (x, then y, then z; the 5-sum along x = the 3-sum + the two outer taps).
__device__ __forceinline__ void box_gate_body(const float* __restrict__ dy, const float* __restrict__ g, float* __restrict__ hb3,
                                              float* __restrict__ hb5, int D, int H, int W, int C, int CB, int n, int cgroup,
                                              unsigned char* smem) {
  const int V = D * H * W, c4n = CB / 4, items = V * c4n;
  f32x4* A = reinterpret_cast<f32x4*>(smem);
  f32x4* B3 = A + items;
  f32x4* B5 = B3 + items;
  f32x4* A3 = B5 + items;
  const int c0 = cgroup * CB;
  const int tid = threadIdx.x;
  BoxItems it;
  it.init(V, c4n, D, H, W, tid);
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {
    const int i = tid + j * 256;
    if (i < items) {
      const int v = i / c4n, q = i % c4n, c = c0 + 4 * q;
      A[i] = c < C ? *reinterpret_cast<const f32x4*>(dy + ((size_t)n * V + v) * C + c) : zero;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BOX_MAXI; ++j) {             // along x: A -> B3, B5
    const int i = tid + j * 256;
    if (i < items) {
      const int xx = (int)(it.xyz[j] & 1023u);
      const f32x4 t3 = box_line<1>(A, i, c4n, xx, W);
      f32x4 t5 = t3;
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
  for (int j = 0; j < BOX_MAXI; ++j) {             // along z, the gate scale, out
    const int i = tid + j * 256;
    if (i >= items) continue;
    const int v = i / c4n, q = i % c4n, c = c0 + 4 * q;
    if (c >= C) continue;
    const int z = (int)(it.xyz[j] >> 20);
    const size_t o = ((size_t)n * V + v) * C + c;
    const f32x4 g3 = *reinterpret_cast<const f32x4*>(g + ((size_t)n * E + 3) * C + c);
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + ((size_t)n * E + 4) * C + c);
    *reinterpret_cast<f32x4*>(hb3 + o) = g3 * (box_line<1>(A3, i, H * W * c4n, z, D) * (1.0f / 27.0f));
    *reinterpret_cast<f32x4*>(hb5 + o) = g4 * (box_line<2>(A, i, H * W * c4n, z, D) * (1.0f / 125.0f));
  }
}

// the gate mix's backward (workgroups [0, nmix)) and the box means above (the rest) as ONE launch: they read the same dy and
// depend on nothing of each other -- one kernel boundary less per per-expert block and direction
template <typename T>
__global__ __launch_bounds__(256) void expert_mix_bwd_box_kernel(const float* __restrict__ dy, const float* __restrict__ p,
                                                                 const float* __restrict__ g, float* __restrict__ dg,
                                                                 T* __restrict__ dye_lo, float* __restrict__ dye_hi,
                                                                 float* __restrict__ hb3, float* __restrict__ hb5, int N, int D, int H,
                                                                 int W, int C, long vchunk, long hi_stride, int det, int mix_x,
                                                                 int nmix, int CB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mixbox_smem[];
  const int b = blockIdx.x;
  if (b < nmix) {
    mix_bwd_body<T>(dy, p, g, dg, dye_lo, dye_hi, N, (long)D * H * W, C, vchunk, hi_stride, det, b % mix_x, b / mix_x);
  } else {
    const int bb = b - nmix;
    box_gate_body(dy, g, hb3, hb5, D, H, W, C, CB, bb % N, bb / N, mixbox_smem);
  }
}

}  // namespace

extern "C" int repmode_expert_mix_fwd(const float* p, const float* g, float* y, int n, long v, int c, void* stream) {
  RM_REQUIRE(p && g && y && n > 0 && v > 0 && c > 0, "expert_mix_fwd: bad argument");
  const long total = (long)n * v * ((c + 3) / 4);
  RM_REQUIRE(total < (1L << 31), "expert_mix_fwd: %ld items (32-bit index arithmetic)", total);
  repmode_prof_begin(REPMODE_PROF_HELPER, (double)n * v * c * 4.0 * (E + 1), static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(expert_mix_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p, g, y, n, v, c);
  repmode_prof_end(static_cast<hipStream_t>(stream));
  RM_LAUNCH_CHECK("expert_mix_fwd");
  return REPMODE_OK;
}

// dg [N][5][C] (overwritten), dye_lo [2][N][V][C] in `dtype`, dye_hi [3][N][V][C] float.
extern "C" int repmode_expert_mix_bwd_ex(const float* dy, const float* p, const float* g, float* dg, void* dye_lo,
                                         float* dye_hi, long hi_stride, int n, long v, int c, int dtype, void* stream);

extern "C" int repmode_expert_mix_bwd(const float* dy, const float* p, const float* g, float* dg, void* dye_lo,
                                      float* dye_hi, int n, long v, int c, int dtype, void* stream) {
  return repmode_expert_mix_bwd_ex(dy, p, g, dg, dye_lo, dye_hi, (long)n * v * c, n, v, c, dtype, stream);
}

// hi_stride: elements between the three float outputs dye_hi[e] (>= n * v * c; lets the caller pad each of them)
extern "C" int repmode_expert_mix_bwd_ex(const float* dy, const float* p, const float* g, float* dg, void* dye_lo,
                                         float* dye_hi, long hi_stride, int n, long v, int c, int dtype, void* stream) {
  const int prezeroed = dtype & 16;     // bit 4 of dtype: dg has been cleared by the caller (pooled memset)
  dtype &= 15;
  RM_REQUIRE(hi_stride >= (long)n * v * c, "expert_mix_bwd: hi_stride too small");
  RM_REQUIRE(dy && p && g && dg && dye_lo && dye_hi && n > 0 && v > 0 && c > 0 && c <= 512, "expert_mix_bwd: bad argument");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "expert_mix_bwd: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!prezeroed) RM_HIP(hipMemsetAsync(dg, 0, (size_t)n * E * c * sizeof(float), s));
  const int rows = 256 / ((c + 3) / 4);
  // RM_MIX_ITERS row iterations per workgroup (level 3, in the step: 4 iterations on 128 workgroups 17.5 us; 1 iteration on
  // 512 workgroups 22.1 us -- every workgroup ends in 5 C float atomics on the sample's gate gradients)
  long chunks = (v + rows * RM_MIX_ITERS - 1) / (rows * RM_MIX_ITERS);
  if (chunks > 256) chunks = 256;
  const int det = repmode_deterministic() ? 1 : 0;
  if (chunks < 1) chunks = 1;
  if (det && chunks > repmode_det_cap(RM_DET_MIX)) chunks = repmode_det_cap(RM_DET_MIX);
  const long vchunk = (v + chunks - 1) / chunks;
  const dim3 grid((unsigned)((v + vchunk - 1) / vchunk), (unsigned)n);
  repmode_prof_begin(REPMODE_PROF_HELPER, (double)n * v * c * (4.0 * (E + 1) + 3 * 4.0 + 2 * (dtype == REPMODE_F32 ? 4.0 : 2.0)), s);
  if (dtype == REPMODE_F32)
    hipLaunchKernelGGL(expert_mix_bwd_kernel<float>, grid, dim3(256), 0, s, dy, p, g, dg, static_cast<float*>(dye_lo), dye_hi,
                       n, v, c, vchunk, hi_stride, det);
  else
    hipLaunchKernelGGL(expert_mix_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, dy, p, g, dg, static_cast<bf16_t*>(dye_lo),
                       dye_hi, n, v, c, vchunk, hi_stride, det);
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("expert_mix_bwd");
  return REPMODE_OK;
}

// expert_mix_bwd + the avg-pool experts' box means of the gate-scaled output gradient in ONE launch (see the kernel): dye_hi as
// repmode_expert_mix_bwd_ex writes it, and hb3 / hb5 [n][d][h][w][c] float = what repmode_box_pair(dye_hi[1], dye_hi[2]) gives.
// c % 4 == 0 and a volume that fits in LDS (the deep levels); REPMODE_EINVAL otherwise (the caller takes the two launches).
extern "C" int repmode_expert_mix_bwd_box(const float* dy, const float* p, const float* g, float* dg, void* dye_lo, float* dye_hi,
                                          long hi_stride, float* hb3, float* hb5, int n, int d, int h, int w, int c, int dtype,
                                          void* stream) {
  const int prezeroed = dtype & 16;
  dtype &= 15;
  const long v = (long)d * h * w;
  RM_REQUIRE(hi_stride >= (long)n * v * c, "expert_mix_bwd_box: hi_stride too small");
  RM_REQUIRE(dy && p && g && dg && dye_lo && dye_hi && hb3 && hb5 && n > 0 && d > 0 && h > 0 && w > 0 && c > 0 && c <= 512,
             "expert_mix_bwd_box: bad argument");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "expert_mix_bwd_box: bad dtype %d", dtype);
  int cb = 0;
  if ((c & 3) == 0)
    for (int t = 16; t >= 4; t >>= 1)
      if (v * t * 4 * 4 <= 64 * 1024 && v * (t / 4) <= 256 * BOX_MAXI && d < 1024 && h < 1024 && w < 1024) { cb = t; break; }
  RM_REQUIRE(cb > 0, "expert_mix_bwd_box: the volume does not fit in LDS (or c %% 4 != 0)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!prezeroed) RM_HIP(hipMemsetAsync(dg, 0, (size_t)n * E * c * sizeof(float), s));
  const int rows = 256 / ((c + 3) / 4);
  long chunks = (v + rows * RM_MIX_ITERS - 1) / (rows * RM_MIX_ITERS);
  if (chunks > 256) chunks = 256;
  const int det = repmode_deterministic() ? 1 : 0;
  if (chunks < 1) chunks = 1;
  if (det && chunks > repmode_det_cap(RM_DET_MIX)) chunks = repmode_det_cap(RM_DET_MIX);
  const long vchunk = (v + chunks - 1) / chunks;
  const int mix_x = (int)((v + vchunk - 1) / vchunk);
  const int nmix = mix_x * n, nbox = n * ((c + cb - 1) / cb);
  const size_t lds = (size_t)v * cb * 4 * 4;
  static bool attr_done[2][32] = {};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  repmode_prof_begin(REPMODE_PROF_HELPER, (double)n * v * c * (4.0 * (E + 1) + 5 * 4.0 + 2 * (dtype == REPMODE_F32 ? 4.0 : 2.0)), s);
  if (dtype == REPMODE_F32) {
    if (!attr_done[0][dev & 31]) {
      RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&expert_mix_bwd_box_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      attr_done[0][dev & 31] = true;
    }
    hipLaunchKernelGGL(expert_mix_bwd_box_kernel<float>, dim3((unsigned)(nmix + nbox)), dim3(256), lds, s, dy, p, g, dg,
                       static_cast<float*>(dye_lo), dye_hi, hb3, hb5, n, d, h, w, c, vchunk, hi_stride, det, mix_x, nmix, cb);
  } else {
    if (!attr_done[1][dev & 31]) {
      RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&expert_mix_bwd_box_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      attr_done[1][dev & 31] = true;
    }
    hipLaunchKernelGGL(expert_mix_bwd_box_kernel<bf16_t>, dim3((unsigned)(nmix + nbox)), dim3(256), lds, s, dy, p, g, dg,
                       static_cast<bf16_t*>(dye_lo), dye_hi, hb3, hb5, n, d, h, w, c, vchunk, hi_stride, det, mix_x, nmix, cb);
  }
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("expert_mix_bwd_box");
  return REPMODE_OK;
}
