// k2s2.hip -- the stride-2 2x2x2 stages of the RepMode U-Net on channels-last activations.
//
//   Conv3d(C, C, kernel_size=2, stride=2, bias=False)            fnet/nn_modules/RepMode.py:81   (down)
//   ConvTranspose3d(Ci, Co, kernel_size=2, stride=2, bias=False) fnet/nn_modules/RepMode.py:98   (up)
//
// With kernel == stride the 8 taps of a coarse voxel m touch 8 disjoint fine voxels fine(m, p), so both
// stages (and each other's data gradients) are plain GEMMs with a gather / scatter of voxel rows:
//
//   GATHER : out[m][co]          = sum_p sum_ci in[fine(m,p)][ci] * W[p][co][ci]   (fine -> coarse)
//   SCATTER: out[fine(m,p)][co]  = sum_ci        in[m][ci]         * W[p][co][ci]   (coarse -> fine)
//
//   down forward = GATHER, down data-gradient = SCATTER, up forward = SCATTER, up data-gradient = GATHER.
//
// MFMA "A" operand = filter rows (32 output channels) from the fragment-major tensor
// W[p][row tile][red chunk][32][KC] (1 KiB contiguous per tile, as for the 5x5x5 conv), "B" operand =
// 32 voxels x KC input channels read straight from HBM/L2 (every input element is used exactly once per
// output-channel tile, so there is nothing to stage in LDS).  0.55 % of the network's FLOPs: these kernels
// are bandwidth/latency bound; what they buy is the removal of the permute copies around a library GEMM.
// k2s2_kernel: 128 coarse voxels per workgroup (the wide levels); k2s2_split_kernel: the under-filled deep levels (the waves
// of a workgroup split the taps of one 32-voxel tile).  Filter gradients: k2s2_wgrad_kernel (bf16), k2s2_wgrad_f32_kernel.
#include "common.h"

#include <cstdlib>

namespace {

template <typename T>
struct KElem;
template <>
struct KElem<float> {
  static constexpr int KV = 4;
  __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};
template <>
struct KElem<bf16_t> {
  static constexpr int KV = 8;
  __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

struct K2Args {
  const void* in;
  const void* w;
  void* out;
  long M;                 // coarse voxels (N * d * h * w)
  int d, h, wd;           // coarse spatial dims
  int Cin, Cout, CinP, CoutP;
};

template <typename T>
__device__ __forceinline__ u32x4 load_chunk(const T* row, int c, int C, bool vec) {
  constexpr int KV = KElem<T>::KV;
  if (c >= C) return u32x4{0u, 0u, 0u, 0u};
  if (vec) return *reinterpret_cast<const u32x4*>(row + c);
  T e[KV];
#pragma unroll
  for (int k = 0; k < KV; ++k) e[k] = (c + k < C) ? row[c + k] : (T)0;
  return *reinterpret_cast<const u32x4*>(e);
}

// 4 consecutive output channels (one accumulator quad) of one voxel row
template <typename T>
__device__ __forceinline__ void store_quad(T* row, int co, int Cout, float v0, float v1, float v2, float v3) {
  if (co >= Cout) return;
  if constexpr (sizeof(T) == 4) {
    if ((Cout & 3) == 0) { *reinterpret_cast<f32x4*>(row + co) = f32x4{v0, v1, v2, v3}; return; }
    row[co] = v0;
    if (co + 1 < Cout) row[co + 1] = v1;
    if (co + 2 < Cout) row[co + 2] = v2;
    if (co + 3 < Cout) row[co + 3] = v3;
  } else {
    if ((Cout & 3) == 0) { *reinterpret_cast<u32x2*>(row + co) = u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)}; return; }
    row[co] = f32_to_bf16(v0);
    if (co + 1 < Cout) row[co + 1] = f32_to_bf16(v1);
    if (co + 2 < Cout) row[co + 2] = f32_to_bf16(v2);
    if (co + 3 < Cout) row[co + 3] = f32_to_bf16(v3);
  }
}

// fine-grid row index of tap p = (pz, py, px) of coarse voxel m
// (32-bit arithmetic: every launcher checks that the fine grid has < 2^31 voxels.  A 64-bit division by a run-time value is
// a ~200-instruction routine, and these kernels' bodies are a few dozen instructions per voxel)
__device__ __forceinline__ size_t fine_row(long m_, int p, int d, int h, int w) {
  const uint32_t m = (uint32_t)m_, uw = (uint32_t)w, uh = (uint32_t)h, ud = (uint32_t)d;
  const uint32_t x = m % uw, t = m / uw, y = t % uh, t2 = t / uh, z = t2 % ud, n = t2 / ud;
  const uint32_t pz = (uint32_t)p >> 2, py = ((uint32_t)p >> 1) & 1u, px = (uint32_t)p & 1u;
  return (size_t)(((n * 2u * ud + 2u * z + pz) * (2u * uh) + 2u * y + py) * (2u * uw) + 2u * x + px);
}

// workgroup = 4 waves = 4 x 32 coarse voxels; blockIdx.y = output-channel tile (32)
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256, 2) void k2s2_kernel(K2Args a) {
  constexpr int KV = KElem<T>::KV, KC = 2 * KV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const long m = ((long)blockIdx.x * 4 + wave) * 32 + l31;
  const bool mvalid = m < a.M;
  const long mc = mvalid ? m : a.M - 1;
  const int rt = blockIdx.y;
  const int nkc = a.CinP / KC, nrt = a.CoutP / 32;
  const int Cin = a.Cin, Cout = a.Cout;
  const bool vec = (Cin % KV) == 0;
  const T* __restrict__ in = static_cast<const T*>(a.in);
  const T* __restrict__ wt = static_cast<const T*>(a.w) + ((size_t)rt * nkc) * (32 * KC) + l31 * KC + khalf * KV;
  const size_t tap_stride = (size_t)nrt * nkc * (32 * KC);
  T* __restrict__ out = static_cast<T*>(a.out);

  size_t frow[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) frow[p] = fine_row(mc, p, a.d, a.h, a.wd);

  if constexpr (!SCATTER) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
      u32x4 b[8], wa[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        b[p] = load_chunk<T>(in + frow[p] * Cin, kc * KC + khalf * KV, Cin, vec);
        wa[p] = *reinterpret_cast<const u32x4*>(wt + (size_t)p * tap_stride + (size_t)kc * (32 * KC));
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) KElem<T>::mma(wa[p], b[p], acc);
    }
    if (mvalid) {
      T* row = out + (size_t)m * Cout;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        store_quad<T>(row, rt * 32 + 8 * q + 4 * khalf, Cout, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
  } else {
    // blockIdx.z = pz: the four taps (pz, *, *) -- 64 accumulator registers instead of 128, so that four waves
    // per SIMD stay resident (with all eight taps the kernel needed > 256 registers: one wave per SIMD, and the
    // level-0 up-convolution took 54 us for 84 MB); the input row is read by both halves, it is the small side
    const int p0 = 4 * blockIdx.z;
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
      const u32x4 b = load_chunk<T>(in + (size_t)mc * Cin, kc * KC + khalf * KV, Cin, vec);
      u32x4 wa[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        wa[p] = *reinterpret_cast<const u32x4*>(wt + (size_t)(p0 + p) * tap_stride + (size_t)kc * (32 * KC));
#pragma unroll
      for (int p = 0; p < 4; ++p) KElem<T>::mma(wa[p], b, acc[p]);
    }
    if (mvalid) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        T* row = out + (blockIdx.z ? frow[4 + p] : frow[p]) * Cout;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          store_quad<T>(row, rt * 32 + 8 * q + 4 * khalf, Cout, acc[p][4 * q], acc[p][4 * q + 1], acc[p][4 * q + 2],
                        acc[p][4 * q + 3]);
      }
    }
  }
}

// Under-filled launches (the deep levels: a few hundred to a few thousand coarse voxels, hundreds of channels -- 16-64
// workgroups of the kernel above, each one dependent chain of `reduction chunks` round trips to L2): the four waves of a
// workgroup share ONE 32-voxel tile and split the taps (gather: two taps each, summed through LDS; scatter: one tap of
// the z half each, nothing to sum), and a wave requests U reduction chunks' operands before it multiplies the first.
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256, 2) void k2s2_split_kernel(K2Args a) {
  constexpr int KV = KElem<T>::KV, KC = 2 * KV;
  constexpr int U = 8;
  __shared__ float red[SCATTER ? 1 : 4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  const long m = (long)blockIdx.x * 32 + l31;
  const bool mvalid = m < a.M;
  const long mc = mvalid ? m : a.M - 1;
  const int rt = blockIdx.y;
  const int nkc = a.CinP / KC, nrt = a.CoutP / 32;
  const int Cin = a.Cin, Cout = a.Cout;
  const bool vec = (Cin % KV) == 0;
  const T* __restrict__ in = static_cast<const T*>(a.in);
  const T* __restrict__ wt = static_cast<const T*>(a.w) + ((size_t)rt * nkc) * (32 * KC) + l31 * KC + khalf * KV;
  const size_t tap_stride = (size_t)nrt * nkc * (32 * KC);
  T* __restrict__ out = static_cast<T*>(a.out);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (!SCATTER) {
    const int p0 = 2 * wave;
    const T* __restrict__ r0 = in + fine_row(mc, p0, a.d, a.h, a.wd) * Cin;
    const T* __restrict__ r1 = in + fine_row(mc, p0 + 1, a.d, a.h, a.wd) * Cin;
    const T* __restrict__ w0 = wt + (size_t)p0 * tap_stride;
    const T* __restrict__ w1 = w0 + tap_stride;
    for (int kc0 = 0; kc0 < nkc; kc0 += U) {
      u32x4 b0[U], b1[U], a0[U], a1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kc = min(kc0 + u, nkc - 1);
        b0[u] = load_chunk<T>(r0, kc * KC + khalf * KV, Cin, vec);
        b1[u] = load_chunk<T>(r1, kc * KC + khalf * KV, Cin, vec);
        a0[u] = *reinterpret_cast<const u32x4*>(w0 + (size_t)kc * (32 * KC));
        a1[u] = *reinterpret_cast<const u32x4*>(w1 + (size_t)kc * (32 * KC));
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (kc0 + u < nkc) { KElem<T>::mma(a0[u], b0[u], acc); KElem<T>::mma(a1[u], b1[u], acc); }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // wave q stores accumulator quad q (output channels rt 32 + 8 q + 4 khalf ...): the four partial sums in a fixed order
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = ((red[0][4 * wave + i][lane] + red[1][4 * wave + i][lane]) + red[2][4 * wave + i][lane]) + red[3][4 * wave + i][lane];
    if (mvalid) store_quad<T>(out + (size_t)m * Cout, rt * 32 + 8 * wave + 4 * khalf, Cout, v[0], v[1], v[2], v[3]);
  } else {
    const int p = 4 * blockIdx.z + wave;
    const T* __restrict__ r0 = in + (size_t)mc * Cin;
    const T* __restrict__ w0 = wt + (size_t)p * tap_stride;
    for (int kc0 = 0; kc0 < nkc; kc0 += U) {
      u32x4 b0[U], a0[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int kc = min(kc0 + u, nkc - 1);
        b0[u] = load_chunk<T>(r0, kc * KC + khalf * KV, Cin, vec);
        a0[u] = *reinterpret_cast<const u32x4*>(w0 + (size_t)kc * (32 * KC));
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (kc0 + u < nkc) KElem<T>::mma(a0[u], b0[u], acc);
    }
    if (mvalid) {
      T* row = out + fine_row(m, p, a.d, a.h, a.wd) * Cout;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        store_quad<T>(row, rt * 32 + 8 * q + 4 * khalf, Cout, acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
  }
}

}  // namespace

// in/out: channels-last, dtype.  scatter == 0: in is the FINE grid [N][2d][2h][2w][Cin], out the coarse grid
// [N][d][h][w][Cout];  scatter != 0: in coarse, out fine.  w: fragment-major [8][CoutP/32][CinP/KC][32][KC].
extern "C" int repmode_k2s2(const void* in, const void* w, void* out, int n, int d, int h, int wdim, int cin, int cout,
                            int dtype, int scatter, void* stream) {
  RM_REQUIRE(in && w && out, "k2s2: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0, "k2s2: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "k2s2: bad dtype %d", dtype);
  RM_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0,
             "k2s2: pointers must be 16-byte aligned");
  K2Args a{};
  a.in = in; a.w = w; a.out = out;
  a.M = (long)n * d * h * wdim; a.d = d; a.h = h; a.wd = wdim;
  RM_REQUIRE(a.M * 8 < (1L << 31), "k2s2: %ld fine voxels (the kernels decode positions in 32 bits)", a.M * 8);
  a.Cin = cin; a.Cout = cout;
  a.CinP = repmode_padded_channels(cin, dtype, 1);
  a.CoutP = repmode_padded_channels(cout, dtype, 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  static const int split_below = []() { const char* e = getenv("REPMODE_K2S2_SPLIT_BELOW"); return e ? atoi(e) : 256; }();
  if ((long)((a.M + 127) / 128) * (a.CoutP / 32) < split_below) {
    // under-filled: 32-voxel workgroups whose waves split the taps (k2s2_split_kernel)
    const dim3 grid((unsigned)((a.M + 31) / 32), (unsigned)(a.CoutP / 32), scatter ? 2u : 1u);
    if (dtype == REPMODE_F32) {
      if (scatter) hipLaunchKernelGGL((k2s2_split_kernel<float, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k2s2_split_kernel<float, false>), grid, dim3(256), 0, s, a);
    } else {
      if (scatter) hipLaunchKernelGGL((k2s2_split_kernel<bf16_t, true>), grid, dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k2s2_split_kernel<bf16_t, false>), grid, dim3(256), 0, s, a);
    }
    RM_LAUNCH_CHECK("k2s2");
    return REPMODE_OK;
  }
  const dim3 grid((unsigned)((a.M + 127) / 128), (unsigned)(a.CoutP / 32), scatter ? 2u : 1u);   // z: tap half (scatter)
  if (dtype == REPMODE_F32) {
    if (scatter) hipLaunchKernelGGL((k2s2_kernel<float, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k2s2_kernel<float, false>), grid, dim3(256), 0, s, a);
  } else {
    if (scatter) hipLaunchKernelGGL((k2s2_kernel<bf16_t, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k2s2_kernel<bf16_t, false>), grid, dim3(256), 0, s, a);
  }
  RM_LAUNCH_CHECK("k2s2");
  return REPMODE_OK;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of both stride-2 stages (bf16):
//     dw[p][a][b] = sum_m coarse[m][a] * fine[fine(m,p)][b]
// (down: coarse = dy, fine = x -> dW[p][co][ci];  up: coarse = x, fine = dy -> dWt[p][ci][co]).
// K = voxels and both operands are K-major in HBM (channels-last), so -- as in conv5_wgrad -- they are
// transposed once into LDS ([channel][voxel], two x-adjacent voxels per ds_write_b32) and consumed by
// v_mfma_f32_16x16x32_bf16.  A workgroup owns a 32 x 32 (a, b) tile for all 8 taps (8 accumulator tiles per
// wave quadrant), walks a chunk of 64-voxel tiles and adds its partial sums with f32 atomics.
namespace {

constexpr int KW_TM = 64;                       // coarse voxels per tile
constexpr int KW_RS = KW_TM * 2 + 16;           // bytes per channel row in LDS (odd multiple of 16)
constexpr int KW_NT = 512;                      // eight waves: 2 x 2 (a, b) quadrants x two halves of the taps; one workgroup per CU
                                                // is the planner's sweet spot and its tile step is instruction issue, shared by 2 waves per SIMD

// the launch planner's constants (repmode_k2s2_wgrad_ex): workgroups it wants in flight at once (one per CU), a
// workgroup's fixed cost and a tile step, in microseconds
#ifndef K2W_RESIDENT
#define K2W_RESIDENT 256
#endif
#ifndef K2W_FIXED_US
#define K2W_FIXED_US 4.0
#endif
#ifndef K2W_TILE_US
#define K2W_TILE_US 1.2
#endif
struct K2WArgs {
  const bf16_t* coarse;
  const bf16_t* fine;
  float* dw;              // [8][A][B], or the parameter's own [A][B][8] / [B][A][8] (param_layout 1 / 2)
  int param_layout;
  long M;
  int d, h, wd, A, B, ntiles, tiles_per_block;
  int direct;             // every (a, b) tile has ONE workgroup: plain stores, dw needs no clearing
};

__device__ __forceinline__ u32x4 kw_load8(const bf16_t* row, int c, int C, bool vec) {
  if (c >= C) return u32x4{0u, 0u, 0u, 0u};
  if (vec) return *reinterpret_cast<const u32x4*>(row + c);
  bf16_t e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = (c + k < C) ? row[c + k] : (bf16_t)0;
  return *reinterpret_cast<const u32x4*>(e);
}

__device__ __forceinline__ uint32_t kw_elem(const u32x4& v, int k) {
  const uint32_t w = v[k >> 1];
  return (k & 1) ? (w >> 16) : (w & 0xffffu);
}

// VEC: both channel counts are multiples of 8 and both tensors < 2 GiB -- every row is fetched with 16-byte buffer loads whose
// out-of-range cases (past the last voxel, past the last channel, threads without a coarse item) are an out-of-bounds OFFSET
// (returns 0), not a branch: the generic loader's per-load predicates made the tile step ~2 k instructions of exec-mask
// juggling per wave (1.9 us), which was the kernel's bound.
template <bool VEC>
__global__ __launch_bounds__(KW_NT) void k2s2_wgrad_kernel(K2WArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[(32 + 8 * 32) * KW_RS];
  unsigned char* cT = smem;                       // [32 a][64 voxels]
  unsigned char* fT = smem + 32 * KW_RS;          // [8 taps][32 b][64 voxels]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int aq = wave & 1, bq = (wave >> 1) & 1, th = wave >> 2, l15 = lane & 15, kg = lane >> 4;   // th: taps 4 th .. 4 th + 3
  const int at = blockIdx.y, bt = blockIdx.z;
  const bool vec_a = (a.A & 7) == 0, vec_b = (a.B & 7) == 0;
  f32x4 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int t_begin = blockIdx.x * a.tiles_per_block;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_block);
  // This thread's staging items, the same in every tile: the voxel pair q (coarse voxels m0 + 2q, + 1) and the channel group cg
  // of the coarse tile (threads 0..127) and of two taps p = ph + 4k of the fine tile.  The pair is decoded ONCE per tile --
  // the eight taps of a coarse voxel are fixed row offsets from its tap-0 fine row (the per-item fine_row() calls this
  // replaces were ~40 integer divisions per thread and tile: the loop was bound by address arithmetic, not by memory).
  const int q = tid & 31, cg = (tid >> 5) & 3, ph = tid >> 7;      // (512 threads: ph = 0..3)
  struct Rows { u32x4 c0, c1, f0[2], f1[2]; };       // a tile's fetched rows on their way to LDS
  Rows ra, rb;                                       // TWO tiles in flight: a tile step (~2.3 us with one) is fetch latency
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t toffk[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = ph + 4 * k;
    toffk[k] = (uint32_t)((p >> 2) * (2 * a.h) * (2 * a.wd) + ((p >> 1) & 1) * (2 * a.wd) + (p & 1));
  }
  auto fetch = [&](Rows& t, int tile) {
    // (VEC: NO early exit for a tile past the range -- its loads are issued with out-of-bounds offsets and cost nothing.  With
    // a conditional fetch the compiler cannot count the loads in flight and waits for ALL of them (vmcnt(0)) before each
    // transposition: the second register set then hides nothing)
    if (!VEC && tile >= t_end) return;
    const long m = (long)tile * KW_TM + 2 * q;
    // 32-bit decode (the launcher checks that the fine tensor has < 2^31 rows): 64-bit integer division is a ~200-instruction
    // routine, three of them per fetch
    const uint32_t mu = (uint32_t)m, uw = (uint32_t)a.wd, uh = (uint32_t)a.h, ud = (uint32_t)a.d;
    const uint32_t xq = mu % uw, t1 = mu / uw, yq = t1 % uh, t2 = t1 / uh, zq = t2 % ud, nq = t2 / ud;
    const uint32_t r0 = ((nq * 2 * ud + 2 * zq) * (2 * uh) + 2 * yq) * (2 * uw) + 2 * xq;
    uint32_t r1 = r0 + 2;
    if (xq + 1 >= uw) r1 = (uint32_t)fine_row(m + 1 < a.M ? m + 1 : m, 0, a.d, a.h, a.wd);     // (odd widths only)
    if constexpr (VEC) {
      const bool v0 = m < a.M && tile < t_end, v1 = m + 1 < a.M && tile < t_end;
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.coarse), 0, (int)(a.M * a.A * 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.fine), 0, (int)(a.M * 8 * a.B * 2), 0x00020000);
      const int ca = at * 32 + cg * 8, cb = bt * 32 + cg * 8;
      const bool ua = tid < 128 && ca < a.A, ub = cb < a.B;
      const uint32_t oc = (mu * (uint32_t)a.A + (uint32_t)ca) * 2u;
      t.c0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, (ua && v0) ? oc : OOB, 0, 0));
      t.c1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, (ua && v1) ? oc + (uint32_t)a.A * 2u : OOB, 0, 0));
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint32_t o0 = ((r0 + toffk[k]) * (uint32_t)a.B + (uint32_t)cb) * 2u, o1 = ((r1 + toffk[k]) * (uint32_t)a.B + (uint32_t)cb) * 2u;
        t.f0[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rf, (ub && v0) ? o0 : OOB, 0, 0));
        t.f1[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rf, (ub && v1) ? o1 : OOB, 0, 0));
      }
    } else {
      const u32x4 zero = u32x4{0u, 0u, 0u, 0u};
      t.c0 = zero; t.c1 = zero;
#pragma unroll
      for (int k = 0; k < 2; ++k) { t.f0[k] = zero; t.f1[k] = zero; }
      if (m >= a.M) return;
      const bool two = m + 1 < a.M;
      if (tid < 128) {
        const int c = at * 32 + cg * 8;
        t.c0 = kw_load8(a.coarse + (size_t)m * a.A, c, a.A, vec_a);
        if (two) t.c1 = kw_load8(a.coarse + (size_t)(m + 1) * a.A, c, a.A, vec_a);
      }
      const int c = bt * 32 + cg * 8;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        t.f0[k] = kw_load8(a.fine + (size_t)(r0 + toffk[k]) * a.B, c, a.B, vec_b);
        if (two) t.f1[k] = kw_load8(a.fine + (size_t)(r1 + toffk[k]) * a.B, c, a.B, vec_b);
      }
    }
  };
  auto stage = [&](const Rows& t) {
    if (tid < 128) {
      unsigned char* dst = cT + (cg * 8) * KW_RS + q * 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) *reinterpret_cast<uint32_t*>(dst + e * KW_RS) = kw_elem(t.c0, e) | (kw_elem(t.c1, e) << 16);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned char* dst = fT + ((size_t)(ph + 4 * k) * 32 + cg * 8) * KW_RS + q * 4;
#pragma unroll
      for (int e = 0; e < 8; ++e) *reinterpret_cast<uint32_t*>(dst + e * KW_RS) = kw_elem(t.f0[k], e) | (kw_elem(t.f1[k], e) << 16);
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int ks = 0; ks < KW_TM / 32; ++ks) {
      const u32x4 af = *reinterpret_cast<const u32x4*>(cT + (aq * 16 + l15) * KW_RS + (ks * 32 + kg * 8) * 2);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const u32x4 bf = *reinterpret_cast<const u32x4*>(fT + ((size_t)(th * 4 + p) * 32 + bq * 16 + l15) * KW_RS + (ks * 32 + kg * 8) * 2);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf),
                                                         acc[p], 0, 0, 0);
      }
    }
  };
  // tile t's rows are transposed into LDS two tile steps after they were requested
  fetch(ra, t_begin);
  fetch(rb, t_begin + 1);
  for (int tile = t_begin; tile < t_end; tile += 2) {
    __syncthreads();
    stage(ra);
    __syncthreads();
    fetch(ra, tile + 2);
    mma();
    if (VEC || tile + 1 < t_end) {                   // (VEC: a tile past the range is all zeros -- straight-line code, see fetch)
      __syncthreads();
      stage(rb);
      __syncthreads();
      fetch(rb, tile + 3);
      mma();
    }
  }
  // 16x16 C/D layout: column (b) = lane & 15, row (a) = (lane >> 4) * 4 + r
  const int bcol = bt * 32 + bq * 16 + l15;
  if (bcol < a.B) {
#pragma unroll
    for (int pj = 0; pj < 4; ++pj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int arow = at * 32 + aq * 16 + kg * 4 + r, p = th * 4 + pj;
        if (arow < a.A) {
          const size_t at = a.param_layout == 0 ? ((size_t)p * a.A + arow) * a.B + bcol
                          : a.param_layout == 1 ? ((size_t)arow * a.B + bcol) * 8 + p
                                                : ((size_t)bcol * a.A + arow) * 8 + p;
          if (a.direct) a.dw[at] = acc[pj][r];           // (one workgroup per (a, b) tile: nothing to add to)
          else unsafeAtomicAdd(a.dw + at, acc[pj][r]);
        }
      }
  }
}

// The same filter gradient for float32 tensors (the parity mode; round 2 used a library GEMM on gathered patches here, so
// the float32 goldens never ran this build's own stride-2 filter gradient).  Exact float32 FMAs on the vector units: a
// workgroup owns a 32 x 32 (a, b) tile of ONE tap and a chunk of the voxel range, stages 32 voxel rows of both tensors in
// LDS and every thread accumulates a 2 x 2 patch.  Not a fast path: float32 is 1/16 of the bf16 rate everywhere.
struct K2WArgsF {
  const float* coarse;
  const float* fine;
  float* dw;
  int param_layout;
  long M;
  int d, h, wd, A, B, rows_per_block;
};

__global__ __launch_bounds__(256) void k2s2_wgrad_f32_kernel(K2WArgsF a) {
  __shared__ float sc[32][33], sf[32][33];
  const int tid = threadIdx.x;
  const int nbt = (a.B + 31) / 32;
  const int at = blockIdx.y / nbt, bt = blockIdx.y % nbt, p = blockIdx.z;
  const int ta = tid >> 4, tb = tid & 15;                 // this thread's outputs: a = 2 ta + {0,1}, b = 2 tb + {0,1}
  const int lr = tid >> 3, lc = (tid & 7) * 4;            // staging: row lr, 4 consecutive channels from lc
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  const long m_begin = (long)blockIdx.x * a.rows_per_block, m_end = min(a.M, m_begin + a.rows_per_block);
  for (long m0 = m_begin; m0 < m_end; m0 += 32) {
    const long m = m0 + lr;
    float vc[4] = {0.f, 0.f, 0.f, 0.f}, vf[4] = {0.f, 0.f, 0.f, 0.f};
    if (m < m_end) {
      const float* cr = a.coarse + (size_t)m * a.A;
      const float* fr = a.fine + fine_row(m, p, a.d, a.h, a.wd) * a.B;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ca = at * 32 + lc + k, cb = bt * 32 + lc + k;
        if (ca < a.A) vc[k] = cr[ca];
        if (cb < a.B) vf[k] = fr[cb];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) { sc[lr][lc + k] = vc[k]; sf[lr][lc + k] = vf[k]; }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const float a0 = sc[k][2 * ta], a1 = sc[k][2 * ta + 1], b0 = sf[k][2 * tb], b1 = sf[k][2 * tb + 1];
      acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
      acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int arow = at * 32 + 2 * ta + i, bcol = bt * 32 + 2 * tb + j;
      if (arow < a.A && bcol < a.B) {
        const size_t o = a.param_layout == 0 ? ((size_t)p * a.A + arow) * a.B + bcol
                       : a.param_layout == 1 ? ((size_t)arow * a.B + bcol) * 8 + p
                                             : ((size_t)bcol * a.A + arow) * 8 + p;
        if (gridDim.x == 1) a.dw[o] = acc[i][j];
        else unsafeAtomicAdd(a.dw + o, acc[i][j]);
      }
    }
}

}  // namespace

// dw[8][A][B] (float, overwritten) = sum_m coarse[m][a] * fine[fine(m,p)][b].  coarse: [N][d][h][w][A] bf16,
// fine: [N][2d][2h][2w][B] bf16.
extern "C" int repmode_k2s2_wgrad_ex(const void* coarse, const void* fine, float* dw, int n, int d, int h, int wdim,
                                     int ca, int cb, int param_layout, void* stream);

extern "C" int repmode_k2s2_wgrad(const void* coarse, const void* fine, float* dw, int n, int d, int h, int wdim, int ca,
                                  int cb, void* stream) {
  return repmode_k2s2_wgrad_ex(coarse, fine, dw, n, d, h, wdim, ca, cb, 0, stream);
}

// param_layout: 0 = dw[8][A][B]; 1 = dw[A][B][2][2][2]; 2 = dw[B][A][2][2][2] (the Conv3d / ConvTranspose3d
// parameter layouts, so that the gradient needs no permute copy)
extern "C" int repmode_k2s2_wgrad_ex(const void* coarse, const void* fine, float* dw, int n, int d, int h, int wdim,
                                     int ca, int cb, int param_layout, void* stream) {
  RM_REQUIRE(coarse && fine && dw, "k2s2_wgrad: null pointer");
  const int prezeroed = param_layout & 4;        // bit 2: dw has been cleared by the caller
  const int is_f32 = param_layout & 8;           // bit 3: coarse and fine are float32 tensors (parity mode)
  param_layout &= 3;
  RM_REQUIRE(param_layout >= 0 && param_layout <= 2, "k2s2_wgrad: bad layout %d", param_layout);
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && ca > 0 && cb > 0, "k2s2_wgrad: bad shape");
  RM_REQUIRE(((uintptr_t)coarse & 15) == 0 && ((uintptr_t)fine & 15) == 0, "k2s2_wgrad: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  RM_REQUIRE((long)n * d * h * wdim * 8 < (1L << 31), "k2s2_wgrad: %ld fine voxels (the kernels decode positions in 32 bits)", (long)n * d * h * wdim * 8);
  if (is_f32) {
    K2WArgsF f{};
    f.coarse = static_cast<const float*>(coarse); f.fine = static_cast<const float*>(fine); f.dw = dw;
    f.param_layout = param_layout;
    f.M = (long)n * d * h * wdim; f.d = d; f.h = h; f.wd = wdim; f.A = ca; f.B = cb;
    const int tiles = ceil_div(ca, 32) * ceil_div(cb, 32);
    long chunks = (512 + 8L * tiles - 1) / (8L * tiles);      // enough workgroups to fill the chip
    const long max_chunks = (f.M + 31) / 32;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    if (repmode_deterministic() && chunks > repmode_det_cap(RM_DET_K2S2)) chunks = repmode_det_cap(RM_DET_K2S2);      // (deterministic: at most two addends per element of the cleared dw)
    f.rows_per_block = (int)(((f.M + chunks - 1) / chunks + 31) / 32 * 32);
    const int nchunks_f = (int)((f.M + f.rows_per_block - 1) / f.rows_per_block);
    if (!prezeroed && nchunks_f > 1) RM_HIP(hipMemsetAsync(dw, 0, (size_t)8 * ca * cb * sizeof(float), s));
    hipLaunchKernelGGL(k2s2_wgrad_f32_kernel, dim3(nchunks_f, tiles, 8), dim3(256), 0, s, f);
    RM_LAUNCH_CHECK("k2s2_wgrad_f32");
    return REPMODE_OK;
  }
  K2WArgs a{};
  a.coarse = static_cast<const bf16_t*>(coarse); a.fine = static_cast<const bf16_t*>(fine); a.dw = dw;
  a.param_layout = param_layout;
  a.M = (long)n * d * h * wdim; a.d = d; a.h = h; a.wd = wdim; a.A = ca; a.B = cb;
  a.ntiles = (int)((a.M + KW_TM - 1) / KW_TM);
  const int nat = ceil_div(ca, 32), nbt = ceil_div(cb, 32);
  // The voxel range is split over `want` workgroups per (a, b) tile.  Every workgroup but a lone one ends in 8 x 32 x 32 float
  // atomics (32 KB at the memory-side rate of ~1.2 TB/s; with one or two tiles all of them on the same few hundred lines), a
  // tile step is ~1.2 us, one workgroup per CU is the sweet spot (measured: 512 workgroups of the level-0 up stage 49 us, 256
  // 30 us): the split that minimises rounds x (fixed + steps + atomics) is taken.  Round 3's fixed 512 workgroups made every
  // launch's atomics 16.8 MB = 13.6 us whatever the level.  Same box, us per launch round 3 -> now, batch 8 (tools/
  // k2s2_microbench.py under rocprofv3): down stages 38.7 / 23.4 / 19.0 / 13.0 -> 22.9 / 15.5 / 11.1 / 8-12, up stages
  // 64.2 / 34.2 / 22.6 / 18.7 -> 30.1 / 19.1 / 14.0 / 9.8.
  static const int target_env = []() { const char* e = getenv("REPMODE_K2W_BLOCKS"); return e ? atoi(e) : 0; }();
  long want = 1;
  if (target_env > 0) {
    want = (target_env + (long)nat * nbt - 1) / ((long)nat * nbt);
  } else {
    double best = 1e30;
    for (long w = 1; w <= a.ntiles; w *= 2) {
      const long wgs = w * nat * nbt;
      const double rounds = (double)((wgs + K2W_RESIDENT - 1) / K2W_RESIDENT);
      const double steps = (double)ceil_div(a.ntiles, (int)w);
      const double per_wg = 0.012 + 0.02 * (nat * nbt >= 8 ? 8.0 / (nat * nbt) : 1.0);                 // us: the fewer tiles, the more adds meet on a line
      const double atom = w > 1 ? per_wg * (double)(wgs < K2W_RESIDENT ? wgs : K2W_RESIDENT) : 0.0;     // the round's atomics
      const double cost = rounds * (K2W_FIXED_US + steps * K2W_TILE_US + atom);
      if (cost < best * 0.98) { best = cost; want = w; }
      if (wgs >= 4 * K2W_RESIDENT) break;
    }
  }
  if (want > a.ntiles) want = a.ntiles;
  if (want < 1) want = 1;
  if (repmode_deterministic() && want > repmode_det_cap(RM_DET_K2S2)) want = repmode_det_cap(RM_DET_K2S2);             // (deterministic: at most two addends per element of the cleared dw)
  a.tiles_per_block = ceil_div(a.ntiles, (int)want);
  const int nchunks = ceil_div(a.ntiles, a.tiles_per_block);
  a.direct = nchunks == 1;
  if (!prezeroed && !a.direct) RM_HIP(hipMemsetAsync(dw, 0, (size_t)8 * ca * cb * sizeof(float), s));
  const bool vec = (ca & 7) == 0 && (cb & 7) == 0 && a.M * (long)ca * 2 < (1L << 31) && a.M * 8 * (long)cb * 2 < (1L << 31);
  if (vec) hipLaunchKernelGGL(k2s2_wgrad_kernel<true>, dim3(nchunks, nat, nbt), dim3(KW_NT), 0, s, a);
  else hipLaunchKernelGGL(k2s2_wgrad_kernel<false>, dim3(nchunks, nat, nbt), dim3(KW_NT), 0, s, a);
  RM_LAUNCH_CHECK("k2s2_wgrad");
  return REPMODE_OK;
}

// ------------------------------------------------------------------------------------------------
// 2x2x2 filter -> the fragment-major operand of k2s2_kernel, straight from the parameter tensor:
//   out[p][row tile][red chunk][32][KC] (dtype, zero padded) = w(row, red, p)
//   red_major == 0: w is [rows][red][8]   (Conv3d weight [Co][Ci][2][2][2] with rows = Co, ...)
//   red_major != 0: w is [red][rows][8]
// One launch instead of the pad / slice-copy / permute / cast chain.
namespace {
template <typename T>
__device__ __forceinline__ void k2_frags_one(const float* __restrict__ w, int rows, int red, int rowsP, int redP,
                                             int red_major, T* __restrict__ out, long first, long step) {
  constexpr int KC = sizeof(T) == 2 ? 16 : 8;
  const long total = 8L * rowsP * redP;
  for (long i = first; i < total; i += step) {
    const int kk = (int)(i % KC);
    long t = i / KC;
    const int r32 = (int)(t % 32); t /= 32;
    const int nkc = redP / KC;
    const int kc = (int)(t % nkc); t /= nkc;
    const int nrt = rowsP / 32;
    const int rt = (int)(t % nrt);
    const int p = (int)(t / nrt);
    const int row = rt * 32 + r32, k = kc * KC + kk;
    float v = 0.f;
    if (row < rows && k < red) v = red_major ? w[((size_t)k * rows + row) * 8 + p] : w[((size_t)row * red + k) * 8 + p];
    if constexpr (sizeof(T) == 2) out[i] = f32_to_bf16(v);
    else out[i] = v;
  }
}

// the operand of one direction, and (out2 != nullptr) of the opposite one (rows <-> red, the other memory order of
// w): forward and data-gradient filters of a stride-2 stage from one launch
template <typename T>
__global__ void k2_frags_kernel(const float* __restrict__ w, int rows, int red, int rowsP, int redP, int red_major,
                                T* __restrict__ out, int rowsP2, int redP2, T* __restrict__ out2) {
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
  k2_frags_one<T>(w, rows, red, rowsP, redP, red_major, out, first, step);
  if (out2) k2_frags_one<T>(w, red, rows, rowsP2, redP2, !red_major, out2, first, step);
}
}  // namespace

// ---- the same for SEVERAL filters in one launch (round 6): the eight stride-2 stages of the U-Net laid their operands out with
// one ~5 us launch each in front of their convolution -- eight launches of pure latency per forward pass.
namespace {
constexpr int K2M_MAX = REPMODE_K2_FRAGS_MULTI_MAX;
constexpr int K2M_CHUNK = 2048;                 // elements of one workgroup
struct K2MultiArgs {
  const float* w[K2M_MAX];
  void* out[K2M_MAX];
  void* out_t[K2M_MAX];
  int rows[K2M_MAX], red[K2M_MAX], red_major[K2M_MAX];
  int rowsP[K2M_MAX], redP[K2M_MAX], rowsP2[K2M_MAX], redP2[K2M_MAX];
  int first[K2M_MAX + 1];
  int n;
};
template <typename T>
__global__ __launch_bounds__(256) void k2_frags_multi_kernel(K2MultiArgs a) {
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.first[i + 1]) ++i;
  const long b = (long)blockIdx.x - a.first[i];
  const long first = b * K2M_CHUNK + threadIdx.x;
  // (k2_frags_one walks `first, first + step, ...` up to the tensor's end: here one chunk per workgroup)
  const long t1 = 8L * a.rowsP[i] * a.redP[i], t2 = 8L * a.rowsP2[i] * a.redP2[i];
  constexpr int KC = sizeof(T) == 2 ? 16 : 8;
  for (int role = 0; role < 2; ++role) {
    T* __restrict__ out = static_cast<T*>(role ? a.out_t[i] : a.out[i]);
    if (!out) continue;
    const int rows = role ? a.red[i] : a.rows[i], red = role ? a.rows[i] : a.red[i];
    const int rowsP = role ? a.rowsP2[i] : a.rowsP[i], redP = role ? a.redP2[i] : a.redP[i];
    const int red_major = role ? !a.red_major[i] : a.red_major[i];
    const long total = role ? t2 : t1;
    const float* __restrict__ w = a.w[i];
    for (long e = first; e < total && e < (b + 1) * K2M_CHUNK; e += 256) {
      const int kk = (int)(e % KC);
      long t = e / KC;
      const int r32 = (int)(t % 32); t /= 32;
      const int nkc = redP / KC;
      const int kc = (int)(t % nkc); t /= nkc;
      const int nrt = rowsP / 32;
      const int rt = (int)(t % nrt);
      const int p = (int)(t / nrt);
      const int row = rt * 32 + r32, k = kc * KC + kk;
      float v = 0.f;
      if (row < rows && k < red) v = red_major ? w[((size_t)k * rows + row) * 8 + p] : w[((size_t)row * red + k) * 8 + p];
      if constexpr (sizeof(T) == 2) out[e] = f32_to_bf16(v);
      else out[e] = v;
    }
  }
}
}  // namespace

// n <= REPMODE_K2_FRAGS_MULTI_MAX filters; out_t[i] may be NULL.  Same layouts as repmode_k2_frags2, element for element.
extern "C" int repmode_k2_frags_multi(int n, const float* const* w, const int* rows, const int* red, const int* red_major, int dtype,
                                      void* const* out, void* const* out_t, void* stream) {
  RM_REQUIRE(n > 0 && n <= K2M_MAX && w && rows && red && red_major && out && out_t, "k2_frags_multi: bad arguments");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "k2_frags_multi: bad dtype %d", dtype);
  K2MultiArgs a{};
  a.n = n;
  long blocks = 0;
  for (int i = 0; i < n; ++i) {
    RM_REQUIRE(w[i] && out[i] && rows[i] > 0 && red[i] > 0, "k2_frags_multi: bad filter %d", i);
    a.w[i] = w[i]; a.out[i] = out[i]; a.out_t[i] = out_t[i];
    a.rows[i] = rows[i]; a.red[i] = red[i]; a.red_major[i] = red_major[i] ? 1 : 0;
    a.rowsP[i] = repmode_padded_channels(rows[i], dtype, 0); a.redP[i] = repmode_padded_channels(red[i], dtype, 1);
    a.rowsP2[i] = repmode_padded_channels(red[i], dtype, 0); a.redP2[i] = repmode_padded_channels(rows[i], dtype, 1);
    const long t1 = 8L * a.rowsP[i] * a.redP[i], t2 = out_t[i] ? 8L * a.rowsP2[i] * a.redP2[i] : 0;
    a.first[i] = (int)blocks;
    blocks += ((t1 > t2 ? t1 : t2) + K2M_CHUNK - 1) / K2M_CHUNK;
  }
  a.first[n] = (int)blocks;
  RM_REQUIRE(blocks > 0 && blocks < (1L << 31), "k2_frags_multi: grid out of range");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == REPMODE_BF16) hipLaunchKernelGGL(k2_frags_multi_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k2_frags_multi_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  RM_LAUNCH_CHECK("k2_frags_multi");
  return REPMODE_OK;
}

extern "C" int repmode_k2_frags2(const float* w, int rows, int red, int red_major, int dtype, void* out, void* out_t,
                                 void* stream);

extern "C" int repmode_k2_frags(const float* w, int rows, int red, int red_major, int dtype, void* out, void* stream) {
  return repmode_k2_frags2(w, rows, red, red_major, dtype, out, nullptr, stream);
}

// out_t (may be NULL): the operand with the roles of rows and red exchanged (same w): if `out` is the forward filter
// of a stride-2 stage, out_t is the filter of its data gradient.  8 * padded(red) * padded(rows) elements.
extern "C" int repmode_k2_frags2(const float* w, int rows, int red, int red_major, int dtype, void* out, void* out_t,
                                 void* stream) {
  RM_REQUIRE(w && out, "k2_frags: null pointer");
  RM_REQUIRE(rows > 0 && red > 0, "k2_frags: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "k2_frags: bad dtype %d", dtype);
  const int rowsP = repmode_padded_channels(rows, dtype, 0), redP = repmode_padded_channels(red, dtype, 1);
  const int rowsP2 = repmode_padded_channels(red, dtype, 0), redP2 = repmode_padded_channels(rows, dtype, 1);
  const long total = 8L * rowsP * redP;
  const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == REPMODE_BF16)
    hipLaunchKernelGGL(k2_frags_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, rows, red, rowsP, redP, red_major, (bf16_t*)out,
                       rowsP2, redP2, (bf16_t*)out_t);
  else
    hipLaunchKernelGGL(k2_frags_kernel<float>, dim3(grid), dim3(256), 0, s, w, rows, red, rowsP, redP, red_major, (float*)out,
                       rowsP2, redP2, (float*)out_t);
  RM_LAUNCH_CHECK("k2_frags");
  return REPMODE_OK;
}
