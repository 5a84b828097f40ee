// bnrelu.hip -- BatchNorm3d + ReLU on channels-last activations, forward and backward.
//
// Replaces the `subsequent_layer` of the reference's MoDE block (fnet/nn_modules/RepMode.py:146-149, :212:
// BatchNorm3d(Co) + ReLU(inplace), training: batch mean / biased variance over N*D*H*W per channel,
// eps 1e-5, running statistics with momentum 0.1 and the UNBIASED variance; eval: running statistics)
// and the identical BN+ReLU pairs behind the stride-2 down/up convolutions (RepMode.py:80-84, 97-101).
//
// The tensor is [M][C] (M = voxels of the whole batch, C contiguous), so a thread owns a fixed group of
// consecutive channels (one 16-byte load per row) and strides over rows: per-channel sums stay in
// registers, are combined per workgroup through LDS float atomics and leave the workgroup as one global
// atomic per channel.  Four memory-bound passes per layer and step:
//   forward : stats (read x) ; normalise + ReLU + downcast (read x, write out)
//   backward: reduce (read x, dy) ; apply (read x, dy, write dx)
// against the six kernels plus separate ReLU / dtype-cast passes of the stock path.
#include "common.h"

#include <cstdlib>

namespace {

constexpr int BN_THREADS = 256;
constexpr int BN_MAXC = 512;
constexpr int BN_SLICES = 16;
static_assert((size_t)16 * 2 * 512 <= REPMODE_SCRATCH_BN_HALF, "the slices must fit one BatchNorm half of the scratch");   // partial-sum slices: workgroup b adds into slice b % 16 (16x less atomic contention)

template <typename T>
struct Vec;
template <>
struct Vec<float> {
  static constexpr int CV = 4;
  __device__ static __forceinline__ void load(const float* p, float* v) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static __forceinline__ void store(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  }
};
template <>
struct Vec<bf16_t> {
  static constexpr int CV = 8;
  __device__ static __forceinline__ void load(const bf16_t* p, float* v) {
    const u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = __uint_as_float(t[k] << 16);
      v[2 * k + 1] = __uint_as_float(t[k] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store(bf16_t* p, const float* v) {
    *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                         pack_bf16x2(v[6], v[7])};
  }
};

template <typename T>
__device__ __forceinline__ float load1(const T* p);
template <>
__device__ __forceinline__ float load1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T>
__device__ __forceinline__ void store1(T* p, float v);
template <>
__device__ __forceinline__ void store1<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store1<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// generic row loader: CV consecutive channels starting at c (vector when C % CV == 0)
template <typename T, int CV>
__device__ __forceinline__ void load_row(const T* row, int c, int C, bool vec, float* v) {
  if (vec) {
    if constexpr (CV == Vec<T>::CV) {
      Vec<T>::load(row + c, v);
    } else {   // CV == 8 over float: two 16-byte loads
      Vec<T>::load(row + c, v);
      Vec<T>::load(row + c + 4, v + 4);
    }
  } else {
#pragma unroll
    for (int k = 0; k < CV; ++k) v[k] = (c + k < C) ? load1<T>(row + c + k) : 0.f;
  }
}
template <typename T, int CV>
__device__ __forceinline__ void store_row(T* row, int c, int C, bool vec, const float* v) {
  if (vec) {
    if constexpr (CV == Vec<T>::CV) {
      Vec<T>::store(row + c, v);
    } else {
      Vec<T>::store(row + c, v);
      Vec<T>::store(row + c + 4, v + 4);
    }
  } else {
#pragma unroll
    for (int k = 0; k < CV; ++k) if (c + k < C) store1<T>(row + c + k, v[k]);
  }
}

// All kernels use CV = 8 channels per thread; ngroups = ceil(C / 8) threads cover a row.
constexpr int CV = 8;
// grid caps of the reduction / normalisation kernels (workgroups)
// Every workgroup starts by totalling the 16 partial-sum slices of ALL channels (the normalisation kernels) or ends in 2 C float
// atomics (the reductions): a fixed cost of a microsecond or two, which a workgroup with 8-16 KB to stream does not amortise.
// Tensors of >= BN_BIG_ELEMS elements (level 0 at batch 8) keep the wide grids, where bytes in flight matter more.  Same box,
// whole step, caps (reduce / apply) 1024 / 2048 everywhere -> 512 / 1024: 9.911 / 9.928 -> 9.861 / 9.865 ms; 256 / 512: 9.98;
// 2048 / 4096: 10.06; then with the wide grids kept above BN_BIG_ELEMS 10.00 -> 9.91, and 512 / 768: 9.765 / 9.716 ->
// 9.689 / 9.702 on another box (tools/sessions/r4_session25*.sh).
#ifndef BN_REDUCE_CAP
#define BN_REDUCE_CAP 512
#endif
#ifndef BN_APPLY_CAP
#define BN_APPLY_CAP 768
#endif
#ifndef BN_BIG_ELEMS
#define BN_BIG_ELEMS (16L << 20)
#endif

struct Geom {
  int ngroups, rows_per_iter;
};
__device__ __forceinline__ Geom geom(int C) {
  Geom g;
  g.ngroups = (C + CV - 1) / CV;
  g.rows_per_iter = BN_THREADS / g.ngroups;
  return g;
}

// sums[c] += sum_rows a(row, c) ; sums[C + c] += sum_rows b(row, c)
// nsl: partial-sum slices in use -- BN_SLICES, or (deterministic mode) half as many as the grid has workgroups: two writers each
__device__ __forceinline__ void block_commit(float* lds, const float* s, const float* ss, int c0, int C, bool active,
                                             float* __restrict__ sums, int nsl, bool det, bool returning = false) {
  if (det) {
    // fixed summation order: every thread's partial sums through LDS, channel i's total = its row groups in order; the
    // slice has at most two writers (grid <= 2 nsl): two addends on a zeroed entry commute
    __shared__ float all[BN_THREADS * 2 * CV];
    const int ngroups = (C + CV - 1) / CV, rows = BN_THREADS / ngroups;
#pragma unroll
    for (int k = 0; k < CV; ++k) { all[threadIdx.x * 2 * CV + k] = active ? s[k] : 0.f; all[threadIdx.x * 2 * CV + CV + k] = active ? ss[k] : 0.f; }
    __syncthreads();
    float* slice = sums + (size_t)(blockIdx.x % nsl) * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += BN_THREADS) {
      const int which = i / C, ch = i % C;
      float t = 0.f;
      for (int r = 0; r < rows; ++r) t += all[(r * ngroups + ch / CV) * 2 * CV + which * CV + ch % CV];
      atomicAdd(&slice[i], t);
    }
    return;
  }
  for (int i = threadIdx.x; i < 2 * C; i += BN_THREADS) lds[i] = 0.f;
  __syncthreads();
  // lanes l, l + ngroups, l + 2 ngroups ... of a wave hold the same channels: butterfly them together first, so
  // that one lane per (wave, channel group) goes to the LDS atomics instead of up to 64 on one address
  const int ngroups = (C + CV - 1) / CV;
  float sv[CV], ssv[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) { sv[k] = active ? s[k] : 0.f; ssv[k] = active ? ss[k] : 0.f; }
  bool commit = active;
  if ((ngroups & (ngroups - 1)) == 0 && ngroups < 64) {
    for (int off = ngroups; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < CV; ++k) { sv[k] += __shfl_xor(sv[k], off, 64); ssv[k] += __shfl_xor(ssv[k], off, 64); }
    }
    commit = (threadIdx.x & 63) < ngroups;
  }
  if (commit) {
#pragma unroll
    for (int k = 0; k < CV; ++k)
      if (c0 + k < C) {
        atomicAdd(&lds[c0 + k], sv[k]);
        atomicAdd(&lds[C + c0 + k], ssv[k]);
      }
  }
  __syncthreads();
  float* slice = sums + (size_t)(blockIdx.x % nsl) * 2 * C;
  if (returning) {
    // (the one-launch passes: an atomic WITH return has been performed when its value is back -- the arrival at the grid-wide
    // barrier that follows must not overtake these sums)
    for (int i = threadIdx.x; i < 2 * C; i += BN_THREADS) {
      const float old = atomicAdd(&slice[i], lds[i]);
      asm volatile("" ::"v"(old));
    }
    return;
  }
  for (int i = threadIdx.x; i < 2 * C; i += BN_THREADS) atomicAdd(&slice[i], lds[i]);
}

// Total of the BN_SLICES partial sums of entry i (i in [0, 2C)).  The slices live in one of the two BatchNorm
// halves of the library's zero scratch (common.h): the reduction kernel of call k accumulates into half h and
// clears the OTHER half (used by call k-1, whose readers are done by stream order); the apply kernel of call k
// reads half h -- every workgroup finishes the reduction for itself, workgroup 0 also writes the saved statistics.
// So a BatchNorm pass is two launches, with no finalize kernel and no memset.  (A "last workgroup finalizes"
// variant inside the reduction kernels was tried and lost: the device-scope fence it needs writes back and
// invalidates the XCD's L2 once per workgroup.)
__device__ __forceinline__ float slice_total(const float* __restrict__ sums, int i, int C, int nsl) {
  float t = 0.f;
  if (nsl == BN_SLICES) {
#pragma unroll
    for (int k = 0; k < BN_SLICES; ++k) t += sums[(size_t)k * 2 * C + i];
  } else {
    for (int k = 0; k < nsl; ++k) t += sums[(size_t)k * 2 * C + i];
  }
  return t;
}

__device__ __forceinline__ void clear_other_half(float* __restrict__ other) {
  for (size_t i = (size_t)blockIdx.x * BN_THREADS + threadIdx.x; i < REPMODE_SCRATCH_BN_HALF; i += (size_t)gridDim.x * BN_THREADS)
    other[i] = 0.f;
}

template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T* __restrict__ x, long M, int C,
                                                              float* __restrict__ sums, float* __restrict__ other, int nsl, int det) {
  __shared__ float lds[2 * BN_MAXC];
  clear_other_half(other);
  const Geom g = geom(C);
  const int cg = threadIdx.x % g.ngroups, r0 = threadIdx.x / g.ngroups;
  const bool active = r0 < g.rows_per_iter;
  const int c0 = cg * CV;
  const bool vec = (C % CV) == 0;
  float s[CV], ss[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) { s[k] = 0.f; ss[k] = 0.f; }
  if (active) {
    // Shifted sums: S1 = sum (x - x0), S2 = sum (x - x0)^2 with x0 = the channel's value in row 0 (every workgroup reads
    // the same row; bn_apply_relu_kernel adds it back).  var = S2/M - (S1/M)^2 then cancels digits of (mean - x0)^2, not of
    // mean^2: a channel whose |mean| is 1e3 standard deviations keeps its variance.
    float x0[CV];
    load_row<T, CV>(x, c0, C, vec, x0);
    // 4 rows per iteration, loads issued before use: the pass is bandwidth-bound only with enough bytes in flight
    const long stride = (long)gridDim.x * g.rows_per_iter;
    long r = (long)blockIdx.x * g.rows_per_iter + r0;
    for (; r + 3 * stride < M; r += 4 * stride) {
      float v[4][CV];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_row<T, CV>(x + (r + u * stride) * C, c0, C, vec, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < CV; ++k) { const float d = v[u][k] - x0[k]; s[k] += d; ss[k] += d * d; }
    }
    for (; r < M; r += stride) {
      float v[CV];
      load_row<T, CV>(x + r * C, c0, C, vec, v);
#pragma unroll
      for (int k = 0; k < CV; ++k) { const float d = v[k] - x0[k]; s[k] += d; ss[k] += d * d; }
    }
  }
  block_commit(lds, s, ss, c0, C, active, sums, nsl, det != 0);
}

// normalise + ReLU.  Prologue (every workgroup, through LDS): mean / invstd of all channels from the partial-sum
// slices (training) or from the running statistics (eval); workgroup 0 also writes save_mean / save_invstd for
// the backward pass and updates the running statistics (RepMode's BatchNorm3d defaults:
// rm = (1-m) rm + m mean ; rv = (1-m) rv + m var * M/(M-1)).
template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_relu_kernel(
    const TI* __restrict__ x, TO* __restrict__ out, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, float eps, float momentum, int training, long M, int C, int shifted, int nsl) {
  __shared__ float sc[BN_MAXC], sh[BN_MAXC];
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    float mean, invstd;
    if (training) {
      // shifted: the sums are of (x - x0[c]), x0 = row 0 (bn_stats_kernel); else of x itself (the conv epilogue's)
      const float m1 = slice_total(sums, c, C, nsl) / (float)M;
      const float var = fmaxf(slice_total(sums, C + c, C, nsl) / (float)M - m1 * m1, 0.f);
      mean = shifted ? m1 + to_f32<TI>(x[c]) : m1;
      invstd = rsqrtf(var + eps);
      if (blockIdx.x == 0) {
        const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
      }
    } else {
      mean = rmean[c];
      invstd = rsqrtf(rvar[c] + eps);
    }
    if (blockIdx.x == 0) { save_mean[c] = mean; save_invstd[c] = invstd; }
    const float a = gamma[c] * invstd;
    sc[c] = a;
    sh[c] = beta[c] - mean * a;
  }
  __syncthreads();
  const Geom g = geom(C);
  const int cg = threadIdx.x % g.ngroups, r0 = threadIdx.x / g.ngroups;
  const int c0 = cg * CV;
  const bool vec = (C % CV) == 0;
  float scale[CV], shift[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) {
    scale[k] = 0.f; shift[k] = 0.f;
    if (c0 + k < C) { scale[k] = sc[c0 + k]; shift[k] = sh[c0 + k]; }
  }
  if (r0 < g.rows_per_iter) {
    for (long r = (long)blockIdx.x * g.rows_per_iter + r0; r < M; r += (long)gridDim.x * g.rows_per_iter) {
      float v[CV];
      load_row<TI, CV>(x + r * C, c0, C, vec, v);
#pragma unroll
      for (int k = 0; k < CV; ++k) v[k] = fmaxf(v[k] * scale[k] + shift[k], 0.f);
      store_row<TO, CV>(out + r * C, c0, C, vec, v);
    }
  }
}

// backward reduce: dz = dy * [x*scale+shift > 0];  sums[c] += dz, sums[C+c] += dz * xhat
template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(
    const TI* __restrict__ x, const TO* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta, long M, int C,
    float* __restrict__ sums, float* __restrict__ other, int nsl, int det) {
  __shared__ float lds[2 * BN_MAXC];
  clear_other_half(other);
  const Geom g = geom(C);
  const int cg = threadIdx.x % g.ngroups, r0 = threadIdx.x / g.ngroups;
  const bool active = r0 < g.rows_per_iter;
  const int c0 = cg * CV;
  const bool vec = (C % CV) == 0;
  float mu[CV], is[CV], ga[CV], be[CV], s[CV], ss[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) {
    const bool ok = c0 + k < C;
    mu[k] = ok ? mean[c0 + k] : 0.f; is[k] = ok ? invstd[c0 + k] : 0.f;
    ga[k] = ok ? gamma[c0 + k] : 0.f; be[k] = ok ? beta[c0 + k] : 0.f;
    s[k] = 0.f; ss[k] = 0.f;
  }
  if (active) {
    const long stride = (long)gridDim.x * g.rows_per_iter;
    long r = (long)blockIdx.x * g.rows_per_iter + r0;
    for (; r + stride < M; r += 2 * stride) {          // 2 rows (4 loads) in flight per iteration
      float v[2][CV], d[2][CV];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        load_row<TI, CV>(x + (r + u * stride) * C, c0, C, vec, v[u]);
        load_row<TO, CV>(dy + (r + u * stride) * C, c0, C, vec, d[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < CV; ++k) {
          const float xh = (v[u][k] - mu[k]) * is[k];
          const float dz = (xh * ga[k] + be[k] > 0.f) ? d[u][k] : 0.f;
          s[k] += dz;
          ss[k] += dz * xh;
        }
    }
    for (; r < M; r += stride) {
      float v[CV], d[CV];
      load_row<TI, CV>(x + r * C, c0, C, vec, v);
      load_row<TO, CV>(dy + r * C, c0, C, vec, d);
#pragma unroll
      for (int k = 0; k < CV; ++k) {
        const float xh = (v[k] - mu[k]) * is[k];
        const float dz = (xh * ga[k] + be[k] > 0.f) ? d[k] : 0.f;
        s[k] += dz;
        ss[k] += dz * xh;
      }
    }
  }
  block_commit(lds, s, ss, c0, C, active, sums, nsl, det != 0);
}

// backward apply: dx = gamma * invstd * (dz - sum(dz)/M - xhat * sum(dz xhat)/M).  Prologue: every workgroup
// totals the slices of the reduction through LDS; workgroup 0 also writes them out: totals[c] = sum dz (= dbeta),
// totals[C + c] = sum dz * xhat (= dgamma).
template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(
    const TI* __restrict__ x, const TO* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ sums, float* __restrict__ totals, long M, int C, int training, TI* __restrict__ dx, int nsl) {
  __shared__ float tot[2 * BN_MAXC];
  for (int i = threadIdx.x; i < 2 * C; i += BN_THREADS) {
    const float t = slice_total(sums, i, C, nsl);
    tot[i] = t;
    if (blockIdx.x == 0) totals[i] = t;
  }
  __syncthreads();
  const Geom g = geom(C);
  const int cg = threadIdx.x % g.ngroups, r0 = threadIdx.x / g.ngroups;
  const int c0 = cg * CV;
  const bool vec = (C % CV) == 0;
  float mu[CV], is[CV], ga[CV], be[CV], m1[CV], m2[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) {
    const bool ok = c0 + k < C;
    mu[k] = ok ? mean[c0 + k] : 0.f; is[k] = ok ? invstd[c0 + k] : 0.f;
    ga[k] = ok ? gamma[c0 + k] : 0.f; be[k] = ok ? beta[c0 + k] : 0.f;
    // eval mode: the statistics are constants, only the scale survives
    m1[k] = (ok && training) ? tot[c0 + k] / (float)M : 0.f;
    m2[k] = (ok && training) ? tot[C + c0 + k] / (float)M : 0.f;
  }
  if (r0 < g.rows_per_iter) {
    for (long r = (long)blockIdx.x * g.rows_per_iter + r0; r < M; r += (long)gridDim.x * g.rows_per_iter) {
      float v[CV], d[CV];
      load_row<TI, CV>(x + r * C, c0, C, vec, v);
      load_row<TO, CV>(dy + r * C, c0, C, vec, d);
#pragma unroll
      for (int k = 0; k < CV; ++k) {
        const float xh = (v[k] - mu[k]) * is[k];
        const float dz = (xh * ga[k] + be[k] > 0.f) ? d[k] : 0.f;
        v[k] = ga[k] * is[k] * (dz - m1[k] - xh * m2[k]);
      }
      store_row<TI, CV>(dx + r * C, c0, C, vec, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Small tensors (the deep levels: a few thousand rows): ONE launch per pass.  A workgroup owns 8 channels for ALL rows, so
// the batch statistics never leave it -- no partial-sum slices, no second launch, no scratch.  The tensor (<= a few MB) is
// read a second time from L2 for the normalisation.  These layers were 2 launches of ~10 us each per pass, all latency.
constexpr int BN_SMALL_MAX_ROWS = 16384;

// wave butterfly + cross-wave LDS reduction of NV per-thread values; every thread gets the totals
template <int NV>
__device__ __forceinline__ void block_allreduce(float* v, float* lds /* [4][NV] */) {
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[wave * NV + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = lds[k] + lds[NV + k] + lds[2 * NV + k] + lds[3 * NV + k];
}

template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void bn_small_fwd_kernel(
    const TI* __restrict__ x, TO* __restrict__ out, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ save_mean, float* __restrict__ save_invstd, float eps,
    float momentum, long M, int C) {
  __shared__ float red[4 * 2 * CV];
  const int c0 = blockIdx.x * CV;
  const bool vec = (C % CV) == 0;
  float v[2 * CV];
#pragma unroll
  for (int k = 0; k < 2 * CV; ++k) v[k] = 0.f;
  for (long r = threadIdx.x; r < M; r += BN_THREADS) {
    float t[CV];
    load_row<TI, CV>(x + r * C, c0, C, vec, t);
#pragma unroll
    for (int k = 0; k < CV; ++k) { v[k] += t[k]; v[CV + k] += t[k] * t[k]; }
  }
  block_allreduce<2 * CV>(v, red);
  float scale[CV], shift[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) {
    const int c = c0 + k;
    scale[k] = 0.f; shift[k] = 0.f;
    if (c < C) {
      const float mean = v[k] / (float)M;
      const float var = fmaxf(v[CV + k] / (float)M - mean * mean, 0.f);
      const float invstd = rsqrtf(var + eps);
      if (threadIdx.x == 0) {
        const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
        save_mean[c] = mean;
        save_invstd[c] = invstd;
      }
      scale[k] = gamma[c] * invstd;
      shift[k] = beta[c] - mean * scale[k];
    }
  }
  for (long r = threadIdx.x; r < M; r += BN_THREADS) {
    float t[CV];
    load_row<TI, CV>(x + r * C, c0, C, vec, t);
#pragma unroll
    for (int k = 0; k < CV; ++k) t[k] = fmaxf(t[k] * scale[k] + shift[k], 0.f);
    store_row<TO, CV>(out + r * C, c0, C, vec, t);
  }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void bn_small_bwd_kernel(
    const TI* __restrict__ x, const TO* __restrict__ dy, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ totals, long M, int C, int training,
    TI* __restrict__ dx) {
  __shared__ float red[4 * 2 * CV];
  const int c0 = blockIdx.x * CV;
  const bool vec = (C % CV) == 0;
  float mu[CV], is[CV], ga[CV], be[CV], v[2 * CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) {
    const bool ok = c0 + k < C;
    mu[k] = ok ? mean[c0 + k] : 0.f; is[k] = ok ? invstd[c0 + k] : 0.f;
    ga[k] = ok ? gamma[c0 + k] : 0.f; be[k] = ok ? beta[c0 + k] : 0.f;
    v[k] = 0.f; v[CV + k] = 0.f;
  }
  for (long r = threadIdx.x; r < M; r += BN_THREADS) {
    float t[CV], d[CV];
    load_row<TI, CV>(x + r * C, c0, C, vec, t);
    load_row<TO, CV>(dy + r * C, c0, C, vec, d);
#pragma unroll
    for (int k = 0; k < CV; ++k) {
      const float xh = (t[k] - mu[k]) * is[k];
      const float dz = (xh * ga[k] + be[k] > 0.f) ? d[k] : 0.f;
      v[k] += dz;
      v[CV + k] += dz * xh;
    }
  }
  block_allreduce<2 * CV>(v, red);
  float m1[CV], m2[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) {
    const int c = c0 + k;
    if (c < C && threadIdx.x == 0) { totals[c] = v[k]; totals[C + c] = v[CV + k]; }
    m1[k] = training ? v[k] / (float)M : 0.f;
    m2[k] = training ? v[CV + k] / (float)M : 0.f;
  }
  for (long r = threadIdx.x; r < M; r += BN_THREADS) {
    float t[CV], d[CV];
    load_row<TI, CV>(x + r * C, c0, C, vec, t);
    load_row<TO, CV>(dy + r * C, c0, C, vec, d);
#pragma unroll
    for (int k = 0; k < CV; ++k) {
      const float xh = (t[k] - mu[k]) * is[k];
      const float dz = (xh * ga[k] + be[k] > 0.f) ? d[k] : 0.f;
      t[k] = ga[k] * is[k] * (dz - m1[k] - xh * m2[k]);
    }
    store_row<TI, CV>(dx + r * C, c0, C, vec, t);
  }
}

// ------------------------------------------------------------------------------------------------------------
// ONE launch per pass with a grid-wide barrier (round 6).  Every tensor of the step but level 0's
// is small enough for a grid of at most one workgroup per CU to hold ALL of it in registers -- a thread keeps up to
// 16 rows of its 8 channels (raw: 4 registers per bf16 row) -- so a pass is: read the tensor once, partial sums
// into the slices (as the two-launch pass), BARRIER, totals, normalise / gradient from the registers, write.  Against the
// two launches: one launch less (5-10 us each on the deep levels, where both are latency), and the second read of x
// (forward) or of x and dy (backward) is gone -- 50 instead of 84 MB for a level-1 backward pass.
// The barrier (grid_barrier below): arrivals counted with agent-scope atomics in the stream's scratch (common.h), two levels, a
// spin on the barrier's generation; the counters are back at zero before anybody leaves (graph replays find them zero).  All workgroups must
// be resident: the grid never exceeds the CU count, a workgroup is 256 threads with a few KB of LDS (three fit a CU beside
// each other), and it waits for nothing but its own grid -- two processes sharing a GPU (the tests' two ranks on one device)
// cannot hold each other's last workgroups out.  After the barrier the slices are read with agent-scope loads (the sums were
// added by atomics at the memory side; nothing else written before the barrier is read after it), so no cache is invalidated.
// float rows take 8 registers each: fewer of them per thread, so that every variant leaves room for a second workgroup on
// the CU (two processes' full grids fit the chip beside each other; float tensors are the parity mode's, and small)
template <typename TI, typename TO>
struct FuseRows { static constexpr int R = (sizeof(TI) == 2 && sizeof(TO) == 2) ? 16 : 4; };

template <typename T>
struct RowRaw;
template <>
struct RowRaw<bf16_t> {
  u32x4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void unpack(float* f) const {
#pragma unroll
    for (int k = 0; k < 4; ++k) { f[2 * k] = __uint_as_float(v[k] << 16); f[2 * k + 1] = __uint_as_float(v[k] & 0xffff0000u); }
  }
};
template <>
struct RowRaw<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
  __device__ __forceinline__ void unpack(float* f) const {
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
};

// bar: [0 .. 7] * 32 the arrival counters of the eight workgroup classes (blockIdx % 8: one XCD each, a 128-byte line each),
// [8 * 32] the counter of completed classes, [9 * 32] the generation.  Sense-reversing: a workgroup reads the generation, then
// arrives; the last arrival of a class puts its counter back to zero and arrives at the top; the last class puts the top
// counter back and advances the generation, on which everybody else spins.  Every counter is zero again before anybody
// leaves, so the next launch (and a graph replay) starts from zeros.  Two levels because arrivals on ONE address are
// serialised at the memory side (~40 ns each: a flat counter cost ~10 us for 256 workgroups -- measured, r06_bn_fused.txt).
constexpr int BAR_STRIDE = 32;
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned nwg) {
  __syncthreads();                 // (vmcnt(0) per wave: this workgroup's slice atomics have returned)
  if (threadIdx.x == 0) {
    unsigned* gen = bar + 9 * BAR_STRIDE;
    const unsigned g0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned cls = blockIdx.x & 7u, ncls = nwg < 8u ? nwg : 8u;
    const unsigned members = (nwg - cls + 7u) / 8u;
    unsigned* mine = bar + cls * BAR_STRIDE;
    if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
      __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned* top = bar + 8 * BAR_STRIDE;
      if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ncls - 1u) {
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                       // both zeros are on their way before the generation moves
        __hip_atomic_store(gen, g0 + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// total of the slices of entry i, read past every cache that another XCD's atomics do not reach
__device__ __forceinline__ float slice_total_coherent(const float* sums, int i, int C) {
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < BN_SLICES; ++k) t += __hip_atomic_load(sums + (size_t)k * 2 * C + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t;
}

// forward: statistics + normalise + ReLU.  C % 8 == 0; iters = rows per thread (<= BN_FUSE_ROWS); row j of a thread is
// (j * gridDim + blockIdx) * rows_per_iter + r0.
template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void bn_fused_fwd_kernel(
    const TI* __restrict__ x, TO* __restrict__ out, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ sums, float* __restrict__ other, unsigned* bar, float* __restrict__ rmean, float* __restrict__ rvar,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float eps, float momentum, long M, int C, int iters) {
  constexpr int BN_FUSE_ROWS = FuseRows<TI, TO>::R;
  __shared__ float lds[2 * BN_MAXC];
  clear_other_half(other);
  const Geom g = geom(C);
  const int cg = threadIdx.x % g.ngroups, r0 = threadIdx.x / g.ngroups;
  const bool active = r0 < g.rows_per_iter;
  const int c0 = cg * CV;
  RowRaw<TI> rows[BN_FUSE_ROWS];
  float s[CV], ss[CV], x0[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) { s[k] = 0.f; ss[k] = 0.f; x0[k] = 0.f; }
  const long stride = (long)gridDim.x * g.rows_per_iter;
  const long rbase = (long)blockIdx.x * g.rows_per_iter + r0;
  if (active) {
    load_row<TI, CV>(x, c0, C, true, x0);          // shifted sums, as bn_stats_kernel
#pragma unroll
    for (int j = 0; j < BN_FUSE_ROWS; ++j) {
      const long r = rbase + j * stride;
      if (j < iters && r < M) rows[j].load(x + r * C + c0);
    }
#pragma unroll
    for (int j = 0; j < BN_FUSE_ROWS; ++j) {
      const long r = rbase + j * stride;
      if (j < iters && r < M) {
        float v[CV];
        rows[j].unpack(v);
#pragma unroll
        for (int k = 0; k < CV; ++k) { const float d = v[k] - x0[k]; s[k] += d; ss[k] += d * d; }
      }
    }
  }
  block_commit(lds, s, ss, c0, C, active, sums, BN_SLICES, false, true);
  grid_barrier(bar, gridDim.x);
  float* sc = lds;
  float* sh = lds + BN_MAXC;
  for (int c = threadIdx.x; c < C; c += BN_THREADS) {
    const float m1 = slice_total_coherent(sums, c, C) / (float)M;
    const float var = fmaxf(slice_total_coherent(sums, C + c, C) / (float)M - m1 * m1, 0.f);
    const float mean = m1 + to_f32<TI>(x[c]);
    const float invstd = rsqrtf(var + eps);
    if (blockIdx.x == 0) {
      const float unbiased = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unbiased;
      save_mean[c] = mean; save_invstd[c] = invstd;
    }
    const float a = gamma[c] * invstd;
    sc[c] = a;
    sh[c] = beta[c] - mean * a;
  }
  __syncthreads();
  if (active) {
    float scale[CV], shift[CV];
#pragma unroll
    for (int k = 0; k < CV; ++k) { scale[k] = sc[c0 + k]; shift[k] = sh[c0 + k]; }
#pragma unroll
    for (int j = 0; j < BN_FUSE_ROWS; ++j) {
      const long r = rbase + j * stride;
      if (j < iters && r < M) {
        float v[CV];
        rows[j].unpack(v);
#pragma unroll
        for (int k = 0; k < CV; ++k) v[k] = fmaxf(v[k] * scale[k] + shift[k], 0.f);
        store_row<TO, CV>(out + r * C, c0, C, true, v);
      }
    }
  }
}

// backward (training): sums of dz and dz * xhat, then dx, from ONE read of x and dy
template <typename TI, typename TO>
__global__ __launch_bounds__(BN_THREADS) void bn_fused_bwd_kernel(
    const TI* __restrict__ x, const TO* __restrict__ dy, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ sums, float* __restrict__ other,
    unsigned* bar, float* __restrict__ totals, long M, int C, int iters, TI* __restrict__ dx) {
  constexpr int BN_FUSE_ROWS = FuseRows<TI, TO>::R;
  __shared__ float lds[2 * BN_MAXC];
  clear_other_half(other);
  const Geom g = geom(C);
  const int cg = threadIdx.x % g.ngroups, r0 = threadIdx.x / g.ngroups;
  const bool active = r0 < g.rows_per_iter;
  const int c0 = cg * CV;
  RowRaw<TI> xr[BN_FUSE_ROWS];
  RowRaw<TO> dr[BN_FUSE_ROWS];
  float mu[CV], is[CV], ga[CV], be[CV], s[CV], ss[CV];
#pragma unroll
  for (int k = 0; k < CV; ++k) {
    mu[k] = mean[c0 + k]; is[k] = invstd[c0 + k]; ga[k] = gamma[c0 + k]; be[k] = beta[c0 + k];
    s[k] = 0.f; ss[k] = 0.f;
  }
  const long stride = (long)gridDim.x * g.rows_per_iter;
  const long rbase = (long)blockIdx.x * g.rows_per_iter + r0;
  if (active) {
#pragma unroll
    for (int j = 0; j < BN_FUSE_ROWS; ++j) {
      const long r = rbase + j * stride;
      if (j < iters && r < M) { xr[j].load(x + r * C + c0); dr[j].load(dy + r * C + c0); }
    }
#pragma unroll
    for (int j = 0; j < BN_FUSE_ROWS; ++j) {
      const long r = rbase + j * stride;
      if (j < iters && r < M) {
        float v[CV], d[CV];
        xr[j].unpack(v); dr[j].unpack(d);
#pragma unroll
        for (int k = 0; k < CV; ++k) {
          const float xh = (v[k] - mu[k]) * is[k];
          const float dz = (xh * ga[k] + be[k] > 0.f) ? d[k] : 0.f;
          s[k] += dz;
          ss[k] += dz * xh;
        }
      }
    }
  }
  block_commit(lds, s, ss, c0, C, active, sums, BN_SLICES, false, true);
  grid_barrier(bar, gridDim.x);
  for (int i = threadIdx.x; i < 2 * C; i += BN_THREADS) {
    const float t = slice_total_coherent(sums, i, C);
    lds[i] = t;
    if (blockIdx.x == 0) totals[i] = t;
  }
  __syncthreads();
  if (active) {
    float m1[CV], m2[CV];
#pragma unroll
    for (int k = 0; k < CV; ++k) { m1[k] = lds[c0 + k] / (float)M; m2[k] = lds[C + c0 + k] / (float)M; }
#pragma unroll
    for (int j = 0; j < BN_FUSE_ROWS; ++j) {
      const long r = rbase + j * stride;
      if (j < iters && r < M) {
        float v[CV], d[CV];
        xr[j].unpack(v); dr[j].unpack(d);
#pragma unroll
        for (int k = 0; k < CV; ++k) {
          const float xh = (v[k] - mu[k]) * is[k];
          const float dz = (xh * ga[k] + be[k] > 0.f) ? d[k] : 0.f;
          v[k] = ga[k] * is[k] * (dz - m1[k] - xh * m2[k]);
        }
        store_row<TI, CV>(dx + r * C, c0, C, true, v);
      }
    }
  }
}

// REPMODE_BN_FUSED (default 1; 0: always the two-launch passes).  The plan of a one-launch pass: grid (<= CUs) and rows per
// thread; 0 = not eligible (level 0's tensors, channel counts that are no multiples of 8, the deterministic mode, whose
// slices have one writer each).
int g_bn_fused = []() { const char* e = getenv("REPMODE_BN_FUSED"); return e ? atoi(e) : 1; }();
int fused_plan(long M, int C, int in_dtype, int out_dtype, int* iters) {
  const int max_rows = (in_dtype == REPMODE_BF16 && out_dtype == REPMODE_BF16) ? FuseRows<bf16_t, bf16_t>::R : FuseRows<float, float>::R;
  if (!g_bn_fused || (C % CV) != 0 || repmode_deterministic()) return 0;
  static int cus_tab[32] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int& cus = cus_tab[dev & 31];
  if (!cus && (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)) cus = 256;
  const int gmax = cus - repmode_reserve_cus() > 8 ? cus - repmode_reserve_cus() : 8;
  const int rows_per_iter = BN_THREADS / ((C + CV - 1) / CV);
  const long need = (M + rows_per_iter - 1) / rows_per_iter;
  // the smallest grid that holds the tensor (fewer arrivals at the barrier; a workgroup streams up to 64 KB)
  const long grid = (need + max_rows - 1) / max_rows;
  if (grid > gmax) return 0;
  *iters = (int)((need + grid - 1) / grid);
  return (int)grid;
}

// OFF: measured slower than the two-launch passes (one MI355X, batch 8: the train step went from 13.6 to 15.1 ms;
// bn_small_fwd 50 us / bn_small_bwd 83 us per launch against 2 x 10-20 us) -- with one workgroup per 8 channels a lane
// reads 16 bytes of a row that is 256 B - 1 KB long, and C / 8 workgroups (16-64) cannot stream even these few MB.
// Kept for the record (-DBN_SMALL=1 builds it in); the cross-workgroup two-launch scheme above is the product path.
#ifndef BN_SMALL
#define BN_SMALL 0
#endif
inline bool bn_small(long m, int c) { return BN_SMALL && m <= BN_SMALL_MAX_ROWS && (long)m * c <= (4L << 20) && c >= 64; }

// the two reduction kernels end in 2C global atomics per workgroup: cap them at 4 workgroups per CU
int grid_for_reduce(long M, int C) {
  const int ngroups = (C + CV - 1) / CV;
  const int rows_per_iter = BN_THREADS / ngroups;
  long blocks = (M + rows_per_iter - 1) / rows_per_iter;
  const long cap = (long)M * C >= BN_BIG_ELEMS ? 2 * BN_REDUCE_CAP : BN_REDUCE_CAP;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// partial-sum slices of a call: BN_SLICES; deterministic mode: as many as fit the scratch half (one writer each, more
// workgroups for the large tensors, which have few channels)
int slices_for(int C) {
  if (!repmode_deterministic()) return BN_SLICES;
  long n = (long)REPMODE_SCRATCH_BN_HALF / (2L * C);
  return (int)(n > 256 ? 256 : (n < 1 ? 1 : n));
}
int grid_for_reduce_det(long M, int C, int nsl) {
  const int g = grid_for_reduce(M, C);
  const int cap = repmode_det_cap(RM_DET_BN) * nsl;
  return repmode_deterministic() ? (g < cap ? g : cap) : g;      // (two writers per slice: two addends on the cleared entry)
}

int grid_for(long M, int C) {
  const int ngroups = (C + CV - 1) / CV;
  const int rows_per_iter = BN_THREADS / ngroups;
  long blocks = (M + rows_per_iter - 1) / rows_per_iter;
  const long cap = (long)M * C >= BN_BIG_ELEMS ? 2 * BN_APPLY_CAP : BN_APPLY_CAP;   // grid-stride beyond 3 (6) workgroups per CU
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

extern "C" int repmode_set_bn_fused(int on) { g_bn_fused = on; return REPMODE_OK; }
extern "C" int repmode_get_bn_fused(void) { return g_bn_fused; }

// in_dtype: dtype of x (and dx); out_dtype: dtype of out (and dy).  REPMODE_F32 / REPMODE_BF16.
extern "C" int repmode_bn_relu_fwd_ex(const void* x, void* out, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float* save_mean, float* save_invstd, long m, int c, float eps,
                                      float momentum, int training, int in_dtype, int out_dtype, int stats_half, void* stream);

extern "C" int repmode_bn_relu_fwd(const void* x, void* out, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* save_mean, float* save_invstd, long m,
                                   int c, float eps, float momentum, int training, int in_dtype, int out_dtype,
                                   void* stream) {
  return repmode_bn_relu_fwd_ex(x, out, gamma, beta, running_mean, running_var, save_mean, save_invstd, m, c, eps, momentum, training,
                                in_dtype, out_dtype, -1, stream);
}

// stats_half >= 0 (training only): the batch statistics are already in that half of the library's BatchNorm scratch -- the
// producing convolution's epilogue put them there (repmode_conv5_epi) -- so only the normalise + ReLU launch runs.
extern "C" int repmode_bn_relu_fwd_ex(const void* x, void* out, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, float* save_mean, float* save_invstd, long m, int c, float eps,
                                      float momentum, int training, int in_dtype, int out_dtype, int stats_half, void* stream) {
  RM_REQUIRE(x && out && gamma && beta && running_mean && running_var && save_mean && save_invstd,
             "bn_relu_fwd: null pointer");
  RM_REQUIRE(m > 0 && c > 0 && c <= BN_MAXC, "bn_relu_fwd: bad shape (C <= %d)", BN_MAXC);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nsl = slices_for(c), det = repmode_deterministic() ? 1 : 0;
  (void)det;
  const int grid = grid_for(m, c);
  float* own = nullptr;
  RM_REQUIRE(stats_half < 0 || (training && stats_half <= 1), "bn_relu_fwd: bad statistics half %d", stats_half);
  if (training && stats_half < 0 && bn_small(m, c)) {
    const dim3 g((unsigned)((c + CV - 1) / CV));
#define RM_BN_SMALL(TI, TO)                                                                                        \
    hipLaunchKernelGGL((bn_small_fwd_kernel<TI, TO>), g, dim3(BN_THREADS), 0, s, (const TI*)x, (TO*)out, gamma, beta, \
                       running_mean, running_var, save_mean, save_invstd, eps, momentum, m, c)
    if (in_dtype == REPMODE_F32 && out_dtype == REPMODE_F32) RM_BN_SMALL(float, float);
    else if (in_dtype == REPMODE_F32) RM_BN_SMALL(float, bf16_t);
    else if (out_dtype == REPMODE_F32) RM_BN_SMALL(bf16_t, float);
    else RM_BN_SMALL(bf16_t, bf16_t);
#undef RM_BN_SMALL
    RM_LAUNCH_CHECK("bn_small_fwd");
    return REPMODE_OK;
  }
  int fiters = 0;
  const int fgrid = (training && stats_half < 0) ? fused_plan(m, c, in_dtype, out_dtype, &fiters) : 0;
  if (fgrid > 0) {
    float* scratch = repmode_zero_scratch(s);
    if (!scratch) return REPMODE_ELAUNCH;
    const int half = repmode_bn_scratch_half(s);
    own = scratch + (size_t)half * REPMODE_SCRATCH_BN_HALF;
    float* other = scratch + (size_t)(1 - half) * REPMODE_SCRATCH_BN_HALF;
    unsigned* bar = reinterpret_cast<unsigned*>(scratch + REPMODE_SCRATCH_BARRIER_OFF);
#define RM_BN_FUSED(TI, TO)                                                                                              \
    hipLaunchKernelGGL((bn_fused_fwd_kernel<TI, TO>), dim3(fgrid), dim3(BN_THREADS), 0, s, (const TI*)x, (TO*)out, gamma, beta, \
                       own, other, bar, running_mean, running_var, save_mean, save_invstd, eps, momentum, m, c, fiters)
    if (in_dtype == REPMODE_F32 && out_dtype == REPMODE_F32) RM_BN_FUSED(float, float);
    else if (in_dtype == REPMODE_F32) RM_BN_FUSED(float, bf16_t);
    else if (out_dtype == REPMODE_F32) RM_BN_FUSED(bf16_t, float);
    else RM_BN_FUSED(bf16_t, bf16_t);
#undef RM_BN_FUSED
    RM_LAUNCH_CHECK("bn_fused_fwd");
    return REPMODE_OK;
  }
  if (training && stats_half >= 0) {
    float* scratch = repmode_zero_scratch(s);
    if (!scratch) return REPMODE_ELAUNCH;
    own = scratch + (size_t)stats_half * REPMODE_SCRATCH_BN_HALF;
  } else if (training) {
    // partial sums: this call's half of the library's zero scratch (see slice_total)
    float* scratch = repmode_zero_scratch(s);
    if (!scratch) return REPMODE_ELAUNCH;
    const int half = repmode_bn_scratch_half(s);
    own = scratch + (size_t)half * REPMODE_SCRATCH_BN_HALF;
    float* other = scratch + (size_t)(1 - half) * REPMODE_SCRATCH_BN_HALF;
    if (in_dtype == REPMODE_F32)
      hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(grid_for_reduce_det(m, c, nsl)), dim3(BN_THREADS), 0, s, (const float*)x, m, c, own, other, nsl, det);
    else
      hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(grid_for_reduce_det(m, c, nsl)), dim3(BN_THREADS), 0, s, (const bf16_t*)x, m, c, own, other, nsl, det);
    RM_LAUNCH_CHECK("bn_stats");
  }
  const int shifted = (training && stats_half < 0) ? 1 : 0;     // bn_stats_kernel's sums are relative to row 0
#define RM_BN_APPLY(TI, TO)                                                                                      \
  hipLaunchKernelGGL((bn_apply_relu_kernel<TI, TO>), dim3(grid), dim3(BN_THREADS), 0, s, (const TI*)x, (TO*)out, \
                     gamma, beta, own, running_mean, running_var, save_mean, save_invstd, eps, momentum, training, m, c, shifted, \
                     stats_half >= 0 ? BN_SLICES : nsl)
  if (in_dtype == REPMODE_F32 && out_dtype == REPMODE_F32) RM_BN_APPLY(float, float);
  else if (in_dtype == REPMODE_F32) RM_BN_APPLY(float, bf16_t);
  else if (out_dtype == REPMODE_F32) RM_BN_APPLY(bf16_t, float);
  else RM_BN_APPLY(bf16_t, bf16_t);
#undef RM_BN_APPLY
  RM_LAUNCH_CHECK("bn_apply_relu");
  return REPMODE_OK;
}

// On return totals[0..c) = dbeta (sum dz), totals[c..2c) = dgamma (sum dz xhat).
extern "C" int repmode_bn_relu_bwd(const void* x, const void* dy, const float* gamma, const float* beta,
                                   const float* save_mean, const float* save_invstd, void* dx, float* totals, long m,
                                   int c, int training, int in_dtype, int out_dtype, void* stream) {
  RM_REQUIRE(x && dy && gamma && beta && save_mean && save_invstd && dx && totals, "bn_relu_bwd: null pointer");
  RM_REQUIRE(m > 0 && c > 0 && c <= BN_MAXC, "bn_relu_bwd: bad shape (C <= %d)", BN_MAXC);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nsl = slices_for(c), det = repmode_deterministic() ? 1 : 0;
  (void)det;
  const int grid = grid_for(m, c);
  if (bn_small(m, c)) {
    const dim3 g((unsigned)((c + CV - 1) / CV));
#define RM_BN_SMALL(TI, TO)                                                                                              \
    hipLaunchKernelGGL((bn_small_bwd_kernel<TI, TO>), g, dim3(BN_THREADS), 0, s, (const TI*)x, (const TO*)dy, save_mean,  \
                       save_invstd, gamma, beta, totals, m, c, training, (TI*)dx)
    if (in_dtype == REPMODE_F32 && out_dtype == REPMODE_F32) RM_BN_SMALL(float, float);
    else if (in_dtype == REPMODE_F32) RM_BN_SMALL(float, bf16_t);
    else if (out_dtype == REPMODE_F32) RM_BN_SMALL(bf16_t, float);
    else RM_BN_SMALL(bf16_t, bf16_t);
#undef RM_BN_SMALL
    RM_LAUNCH_CHECK("bn_small_bwd");
    return REPMODE_OK;
  }
  float* scratch = repmode_zero_scratch(s);
  if (!scratch) return REPMODE_ELAUNCH;
  const int half = repmode_bn_scratch_half(s);
  float* own = scratch + (size_t)half * REPMODE_SCRATCH_BN_HALF;
  float* other = scratch + (size_t)(1 - half) * REPMODE_SCRATCH_BN_HALF;
  int fiters = 0;
  const int fgrid = training ? fused_plan(m, c, in_dtype, out_dtype, &fiters) : 0;
  if (fgrid > 0) {
    unsigned* bar = reinterpret_cast<unsigned*>(scratch + REPMODE_SCRATCH_BARRIER_OFF);
#define RM_BN_FUSED(TI, TO)                                                                                              \
    hipLaunchKernelGGL((bn_fused_bwd_kernel<TI, TO>), dim3(fgrid), dim3(BN_THREADS), 0, s, (const TI*)x, (const TO*)dy,   \
                       save_mean, save_invstd, gamma, beta, own, other, bar, totals, m, c, fiters, (TI*)dx)
    if (in_dtype == REPMODE_F32 && out_dtype == REPMODE_F32) RM_BN_FUSED(float, float);
    else if (in_dtype == REPMODE_F32) RM_BN_FUSED(float, bf16_t);
    else if (out_dtype == REPMODE_F32) RM_BN_FUSED(bf16_t, float);
    else RM_BN_FUSED(bf16_t, bf16_t);
#undef RM_BN_FUSED
    RM_LAUNCH_CHECK("bn_fused_bwd");
    return REPMODE_OK;
  }
#define RM_BN_BWD(TI, TO)                                                                                              \
  do {                                                                                                                 \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<TI, TO>), dim3(grid_for_reduce_det(m, c, nsl)), dim3(BN_THREADS), 0, s, (const TI*)x, (const TO*)dy, \
                       save_mean, save_invstd, gamma, beta, m, c, own, other, nsl, det);                               \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<TI, TO>), dim3(grid), dim3(BN_THREADS), 0, s, (const TI*)x, (const TO*)dy,  \
                       save_mean, save_invstd, gamma, beta, own, totals, m, c, training, (TI*)dx, nsl);                     \
  } while (0)
  if (in_dtype == REPMODE_F32 && out_dtype == REPMODE_F32) RM_BN_BWD(float, float);
  else if (in_dtype == REPMODE_F32) RM_BN_BWD(float, bf16_t);
  else if (out_dtype == REPMODE_F32) RM_BN_BWD(bf16_t, float);
  else RM_BN_BWD(bf16_t, bf16_t);
#undef RM_BN_BWD
  RM_LAUNCH_CHECK("bn_relu_bwd");
  return REPMODE_OK;
}
