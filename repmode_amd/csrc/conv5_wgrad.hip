// conv5_wgrad.hip -- filter gradient of the 5x5x5 per-slot convolution (autograd of
// fnet/nn_modules/RepMode.py:207, aten::convolution_backward weight grad), reduced over the
// samples of each slot:
//     dw[slot][tap][co][ci] = sum_{n in slot} sum_v dy[n][v][co] * x[n][v + tap][ci]
//
// GEMM view per tap: M = co, N = ci, K = voxels.  This first version computes in exact f32 on
// v_mfma_f32_16x16x4_f32 for both element types (bf16 inputs are widened while staging): with one
// f32 per lane per operand a tap shift is just an LDS address offset, so no operand transposes are
// needed although activations are channels-last (K-major for this GEMM).
//
// Work decomposition: a workgroup owns (sample, 32 co, 32 ci, one dz plane of taps, a chunk of
// output-voxel tiles); each of its 4 waves owns a 16x16 (co, ci) quadrant for the 25 (dy,dx) taps
// of that plane = 25 accumulator tiles (100 VGPRs).  Per 4x16-voxel tile it stages dy[64][32] and
// the (4+4)x(16+4) input halo plane [160][32] in LDS (rows padded to 48 floats: adjacent voxels fall
// in different bank halves) and issues 25 MFMAs per 4-voxel K step.  Partial sums are added to dw
// with f32 atomics (split-K over voxel chunks and over the samples of a slot).
#include "common.h"

namespace {

constexpr int TY = 4, TX = 16, TV = TY * TX;   // output voxels per tile
constexpr int HY = TY + 4, HX = TX + 4, HV = HY * HX;
constexpr int RS = 48;                          // LDS row stride in floats
constexpr int WG_LDS_FLOATS = (TV + HV) * RS;

struct WgradArgs {
  const void* x;
  const void* dy;
  const int32_t* sample_slot;
  float* dw;
  int N, D, H, W, Cin, Cout;
  int ncot, ncit, ntiles, tiles_per_block, nchunks, nty, ntx;
};

template <typename T>
__device__ __forceinline__ void load_row_f32(const T* p, int c, int cmax, bool vec_ok, float* out);

template <>
__device__ __forceinline__ void load_row_f32<float>(const float* p, int c, int cmax, bool vec_ok, float* out) {
  if (vec_ok) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (c + k < cmax) ? p[k] : 0.f;
  }
}

template <>
__device__ __forceinline__ void load_row_f32<bf16_t>(const bf16_t* p, int c, int cmax, bool vec_ok, float* out) {
  if (vec_ok) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
    out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (c + k < cmax) ? bf16_to_f32(p[k]) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv5_wgrad_f32c_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[WG_LDS_FLOATS];
  float* dys = smem;              // [TV][RS]
  float* xs = smem + TV * RS;     // [HV][RS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = wave & 1, ciq = wave >> 1;
  const int l15 = lane & 15, kq = lane >> 4;

  int bid = blockIdx.x;
  const int chunk = bid % a.nchunks; bid /= a.nchunks;
  const int dz = bid % 5;            bid /= 5;
  const int cit = bid % a.ncit;      bid /= a.ncit;
  const int cot = bid % a.ncot;
  const int n = bid / a.ncot;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const int slot = a.sample_slot[n];
  const T* __restrict__ xn = static_cast<const T*>(a.x) + (size_t)n * D * H * W * Cin;
  const T* __restrict__ dyn = static_cast<const T*>(a.dy) + (size_t)n * D * H * W * Cout;
  const bool vec_x = (Cin & 3) == 0, vec_dy = (Cout & 3) == 0;

  f32x4 acc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = chunk * a.tiles_per_block;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_block);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tx = tile % a.ntx, t2 = tile / a.ntx;
    const int ty = t2 % a.nty, z = t2 / a.nty;
    const int zin = z + dz - 2;
    if (zin < 0 || zin >= D) continue;   // whole input plane is padding: contributes nothing (uniform)
    const int y0 = ty * TY, x0 = tx * TX;
    __syncthreads();
    // stage dy tile: TV voxels x 32 co, items of 4 channels
    for (int it = tid; it < TV * 8; it += 256) {
      const int v = it >> 3, c4 = (it & 7) * 4;
      const int gy = y0 + v / TX, gx = x0 + v % TX, c = cot * 32 + c4;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (gy < H && gx < W && c < Cout)
        load_row_f32<T>(dyn + ((size_t)(z * H + gy) * W + gx) * Cout + c, c, Cout, vec_dy, e);
      *reinterpret_cast<f32x4*>(dys + v * RS + c4) = f32x4{e[0], e[1], e[2], e[3]};
    }
    // stage input halo plane: HV voxels x 32 ci
    for (int it = tid; it < HV * 8; it += 256) {
      const int vh = it >> 3, c4 = (it & 7) * 4;
      const int gy = y0 + vh / HX - 2, gx = x0 + vh % HX - 2, c = cit * 32 + c4;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin)
        load_row_f32<T>(xn + ((size_t)(zin * H + gy) * W + gx) * Cin + c, c, Cin, vec_x, e);
      *reinterpret_cast<f32x4*>(xs + vh * RS + c4) = f32x4{e[0], e[1], e[2], e[3]};
    }
    __syncthreads();
    // K loop: 4 consecutive x voxels per step (lane's k = lane >> 4)
#pragma unroll 2
    for (int ks = 0; ks < TV / 4; ++ks) {
      const int v = ks * 4 + kq;
      const int ly = v / TX, lx = v % TX;
      const float af = dys[v * RS + cq * 16 + l15];                 // A[i = co][k = voxel]
      const float* xb = xs + (ly * HX + lx) * RS + ciq * 16 + l15;  // B[k = voxel][j = ci], tap (0,0)
#pragma unroll
      for (int t = 0; t < 25; ++t) {
        const float bf = xb[((t / 5) * HX + (t % 5)) * RS];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[t], 0, 0, 0);
      }
    }
  }
  // 16x16 C/D layout: column (ci) = lane & 15, row (co) = (lane >> 4) * 4 + r
  const int ci = cit * 32 + ciq * 16 + l15;
  if (ci < Cin) {
#pragma unroll
    for (int t = 0; t < 25; ++t) {
      const int tap = dz * 25 + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * 32 + cq * 16 + kq * 4 + r;
        if (co < Cout) unsafeAtomicAdd(a.dw + (((size_t)slot * REPMODE_TAPS + tap) * Cout + co) * Cin + ci, acc[t][r]);
      }
    }
  }
}

}  // namespace

extern "C" int repmode_conv5_wgrad(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                                   float* dw, int n, int d, int h, int wdim, int cin, int cout, int dtype,
                                   void* stream) {
  RM_REQUIRE(x && dy && sample_slot && dw, "conv5_wgrad: null pointer");
  RM_REQUIRE(n > 0 && nslots > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0, "conv5_wgrad: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "conv5_wgrad: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  WgradArgs a{};
  a.x = x; a.dy = dy; a.sample_slot = sample_slot; a.dw = dw;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin; a.Cout = cout;
  a.ncot = ceil_div(cout, 32);
  a.ncit = ceil_div(cin, 32);
  a.nty = ceil_div(h, TY);
  a.ntx = ceil_div(wdim, TX);
  a.ntiles = d * a.nty * a.ntx;
  const long fixed = (long)n * a.ncot * a.ncit * 5;
  long want_chunks = (2048 + fixed - 1) / fixed;       // aim at >= 2048 workgroups
  if (want_chunks < 1) want_chunks = 1;
  if (want_chunks > a.ntiles) want_chunks = a.ntiles;
  a.tiles_per_block = ceil_div(a.ntiles, (int)want_chunks);
  a.nchunks = ceil_div(a.ntiles, a.tiles_per_block);
  const long grid = fixed * a.nchunks;
  RM_REQUIRE(grid < (1L << 31), "conv5_wgrad: grid too large");
  RM_HIP(hipMemsetAsync(dw, 0, (size_t)nslots * REPMODE_TAPS * cout * cin * sizeof(float), s));
  repmode_prof_begin(REPMODE_PROF_WGRAD, 2.0 * n * d * h * wdim * (double)cin * cout * REPMODE_TAPS, s);
  if (dtype == REPMODE_F32)
    hipLaunchKernelGGL(conv5_wgrad_f32c_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(conv5_wgrad_f32c_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, s, a);
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("conv5_wgrad");
  return REPMODE_OK;
}
