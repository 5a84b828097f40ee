// gatrep.hip -- gating re-parameterization (GatRep) forward and backward.
//
// Forward replaces fnet/nn_modules/RepMode.py:44-49 (one-hot), :198-200 (gate Linear + softmax
// over experts), :165-169 / :173-180 (expert padding, avg-pool-as-kernel) and :182-190 (the
// per-sample weighted sum).  The merged filter depends on the task only, so it is produced once
// per distinct task ("slot") of the batch, directly in the layouts the conv kernels read:
//   wf[slot][tap][CoP][CiP]        forward filter
//   wd[slot][124-tap][CiP][CoP]    data-gradient filter (taps flipped, channels transposed)
// Memory-bound: reads 155 expert floats, writes 125 (x2) merged elements per (co, ci) per slot.
//
// Backward is the autograd of the same lines: expert gradients, gate-probability gradients
// (reduced over ci and taps), then the softmax Jacobian and the gate Linear's weight/bias grads.
#include "common.h"

namespace {

constexpr int E = REPMODE_NUM_EXPERTS;
constexpr int TAPS = REPMODE_TAPS;

__global__ void gate_softmax_kernel(const float* __restrict__ gate_w, const float* __restrict__ gate_b,
                                    const int32_t* __restrict__ slot_task, int nslots, int num_tasks,
                                    int co, float* __restrict__ g) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nslots * co) return;
  const int s = idx / co, o = idx % co;
  const int task = slot_task[s];
  float logit[E], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    logit[e] = gate_w[(size_t)(e * co + o) * num_tasks + task] + gate_b[e * co + o];
    mx = fmaxf(mx, logit[e]);
  }
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) { logit[e] = expf(logit[e] - mx); sum += logit[e]; }
  const float inv = 1.f / sum;
#pragma unroll
  for (int e = 0; e < E; ++e) g[((size_t)s * E + e) * co + o] = logit[e] * inv;
}

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

__device__ __forceinline__ bool in_centre3(int tap, int& t3) {
  const int dz = tap / 25, dy = (tap / 5) % 5, dx = tap % 5;
  const bool in = dz >= 1 && dz <= 3 && dy >= 1 && dy <= 3 && dx >= 1 && dx <= 3;
  t3 = ((dz - 1) * 3 + (dy - 1)) * 3 + (dx - 1);
  return in;
}

// One thread per (fast index, slow index) pair of (co, ci); FAST_CI selects which of the two is
// the fast (lane) index so that the layout being written is coalesced along lanes.
// MAXS slots are processed per pass with their gate probabilities held in registers.
template <typename T, bool WRITE_WD>
__global__ __launch_bounds__(256) void gatrep_fwd_kernel(
    const float* __restrict__ k5, const float* __restrict__ k3, const float* __restrict__ k1,
    const float* __restrict__ a3, const float* __restrict__ a5, const float* __restrict__ g, int nslots,
    int co_n, int ci_n, int cop, int cip, T* __restrict__ wout) {
  // WRITE_WD == false: writes wf[s][tap][co][ci], lanes run along ci (padded range cip)
  // WRITE_WD == true : writes wd[s][124-tap][ci][co], lanes run along co (padded range cop)
  const int fast_n = WRITE_WD ? cop : cip;
  const int slow_n = WRITE_WD ? cip : cop;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)fast_n * slow_n) return;
  const int fast = (int)(idx % fast_n), slow = (int)(idx / fast_n);
  const int co = WRITE_WD ? fast : slow;
  const int ci = WRITE_WD ? slow : fast;
  const bool live = co < co_n && ci < ci_n;
  const size_t oi = live ? (size_t)co * ci_n + ci : 0;
  const float e2 = live ? k1[oi] : 0.f;
  const float e3 = live ? a3[oi] * (1.0f / 27.0f) : 0.f;
  const float e4 = live ? a5[oi] * (1.0f / 125.0f) : 0.f;
  const size_t slot_stride = (size_t)TAPS * cop * cip;
  for (int tap = 0; tap < TAPS; ++tap) {
    int t3;
    const bool c3 = in_centre3(tap, t3);
    const float v0 = live ? k5[oi * TAPS + tap] : 0.f;
    const float v1 = (live && c3) ? k3[oi * 27 + t3] : 0.f;
    const float v2 = (tap == 62) ? e2 : 0.f;
    const float v3 = c3 ? e3 : 0.f;
    const size_t off = WRITE_WD ? ((size_t)(TAPS - 1 - tap) * cip + ci) * cop + co
                                : ((size_t)tap * cop + co) * cip + ci;
    for (int s = 0; s < nslots; ++s) {
      float r = 0.f;
      if (live) {
        const float* gs = g + (size_t)s * E * co_n + co;
        // same association order as RepMode.py:184-188: ((((g0 k5 + g1 k3) + g2 k1) + g3 a3) + g4 a5)
        r = gs[0] * v0 + gs[co_n] * v1;
        r = r + gs[2 * co_n] * v2;
        r = r + gs[3 * co_n] * v3;
        r = r + gs[4 * co_n] * e4;
      }
      wout[s * slot_stride + off] = from_f32<T>(r);
    }
  }
}

// Expert gradients and gate-probability gradients from the per-slot filter gradient.
// One thread per (co, ci) with lanes along ci (dw is [s][tap][co][ci]); a block covers one co and
// up to 256 ci, reduces the 5 gate-probability partials per slot over its ci and adds them to dg.
__global__ __launch_bounds__(256) void gatrep_bwd_kernel(
    const float* __restrict__ dw, const float* __restrict__ k5, const float* __restrict__ k3,
    const float* __restrict__ k1, const float* __restrict__ a3, const float* __restrict__ a5,
    const float* __restrict__ g, int nslots, int co_n, int ci_n, float* __restrict__ dk5,
    float* __restrict__ dk3, float* __restrict__ dk1, float* __restrict__ da3, float* __restrict__ da5,
    float* __restrict__ dg) {
  __shared__ float red[4][E];
  const int co = blockIdx.y;
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = ci < ci_n;
  const size_t oi = (size_t)co * ci_n + (live ? ci : 0);
  const size_t tap_stride = (size_t)co_n * ci_n;
  const float w1 = live ? k1[oi] : 0.f;
  const float w3 = live ? a3[oi] * (1.0f / 27.0f) : 0.f;
  const float w5 = live ? a5[oi] * (1.0f / 125.0f) : 0.f;
  float acc1 = 0.f, acc3 = 0.f, acc5 = 0.f;
  // pass over slots outermost for dg (needs per-slot reductions), experts accumulate across slots
  for (int s = 0; s < nslots; ++s) {
    const float* gs = g + (size_t)s * E * co_n + co;
    const float g0 = gs[0], g1 = gs[co_n], g2 = gs[2 * co_n], g3 = gs[3 * co_n], g4 = gs[4 * co_n];
    const float* dws = dw + (size_t)s * TAPS * tap_stride + oi;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, s27 = 0.f, s125 = 0.f;
    for (int tap = 0; tap < TAPS; ++tap) {
      const float d = live ? dws[(size_t)tap * tap_stride] : 0.f;
      int t3;
      const bool c3 = in_centre3(tap, t3);
      s125 += d;
      if (live) {
        const float kv = k5[oi * TAPS + tap];
        p0 += kv * d;
        // dk5 is accumulated in place across slots: first slot overwrites
        float* o5 = dk5 + oi * TAPS + tap;
        *o5 = (s == 0 ? 0.f : *o5) + g0 * d;
        if (c3) {
          s27 += d;
          p1 += k3[oi * 27 + t3] * d;
          float* o3 = dk3 + oi * 27 + t3;
          *o3 = (s == 0 ? 0.f : *o3) + g1 * d;
          if (tap == 62) { p2 = w1 * d; acc1 += g2 * d; }
        }
      }
    }
    acc3 += g3 * s27;
    acc5 += g4 * s125;
    float part[E] = {p0, p1, p2, w3 * s27, w5 * s125};
    // block reduction over ci
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float v = part[e];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = v;
    }
    __syncthreads();
    if (threadIdx.x < E) {
      float v = 0.f;
      for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) v += red[wv][threadIdx.x];
      atomicAdd(dg + ((size_t)s * E + threadIdx.x) * co_n + co, v);
    }
    __syncthreads();
  }
  if (live) {
    dk1[oi] = acc1;
    da3[oi] = acc3 * (1.0f / 27.0f);
    da5[oi] = acc5 * (1.0f / 125.0f);
  }
}

// softmax Jacobian + gate Linear gradients.  One thread per (e, co); loops over slots.
__global__ void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ dg,
                                const int32_t* __restrict__ slot_task, int nslots, int num_tasks, int co_n,
                                float* __restrict__ dgate_w, float* __restrict__ dgate_b) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= E * co_n) return;
  const int e = idx / co_n, o = idx % co_n;
  float* wrow = dgate_w + (size_t)idx * num_tasks;
  for (int t = 0; t < num_tasks; ++t) wrow[t] = 0.f;
  float bsum = 0.f;
  for (int s = 0; s < nslots; ++s) {
    const float* gs = g + (size_t)s * E * co_n + o;
    const float* ds = dg + (size_t)s * E * co_n + o;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) dot += gs[k * co_n] * ds[k * co_n];
    const float dl = gs[e * co_n] * (ds[e * co_n] - dot);
    wrow[slot_task[s]] += dl;   // slots hold distinct tasks; += keeps duplicates correct too
    bsum += dl;
  }
  dgate_b[idx] = bsum;
}

}  // namespace

extern "C" int repmode_gate_softmax(const float* gate_w, const float* gate_b, const int32_t* slot_task,
                                    int nslots, int num_tasks, int co, float* g, void* stream) {
  RM_REQUIRE(gate_w && gate_b && slot_task && g, "gate_softmax: null pointer");
  RM_REQUIRE(nslots > 0 && num_tasks > 0 && co > 0, "gate_softmax: bad shape");
  const int total = nslots * co;
  hipLaunchKernelGGL(gate_softmax_kernel, dim3(ceil_div(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), gate_w, gate_b, slot_task, nslots, num_tasks, co, g);
  RM_LAUNCH_CHECK("gate_softmax");
  return REPMODE_OK;
}

template <typename T>
static int gatrep_fwd_t(const float* k5, const float* k3, const float* k1, const float* a3, const float* a5,
                        const float* g, int nslots, int co, int ci, int dtype, void* wf, void* wd, hipStream_t s) {
  // algorithmic bytes: 155 expert floats read once + 125 merged elements written per slot and layout
  const double bytes = (double)co * ci * (155.0 * 4 + 125.0 * nslots * sizeof(T) * ((wf ? 1 : 0) + (wd ? 1 : 0)));
  repmode_prof_begin(REPMODE_PROF_GATREP_FWD, bytes, s);
  if (wf) {
    // wf[tap][CoP rows (mult of 32)][CiP reduction]
    const int cop = repmode_padded_channels(co, dtype, 0), cip = repmode_padded_channels(ci, dtype, 1);
    const long total = (long)cop * cip;
    hipLaunchKernelGGL((gatrep_fwd_kernel<T, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k5, k3,
                       k1, a3, a5, g, nslots, co, ci, cop, cip, static_cast<T*>(wf));
    RM_LAUNCH_CHECK("gatrep_fwd(wf)");
  }
  if (wd) {
    // wd[124-tap][CiP' rows (mult of 32)][CoP' reduction]: the data-gradient conv swaps the roles of
    // the two channel counts, so each is padded for its role there
    const int cop = repmode_padded_channels(co, dtype, 1), cip = repmode_padded_channels(ci, dtype, 0);
    const long total = (long)cop * cip;
    hipLaunchKernelGGL((gatrep_fwd_kernel<T, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, k5, k3,
                       k1, a3, a5, g, nslots, co, ci, cop, cip, static_cast<T*>(wd));
    RM_LAUNCH_CHECK("gatrep_fwd(wd)");
  }
  repmode_prof_end(s);
  return REPMODE_OK;
}

extern "C" int repmode_gatrep_fwd(const float* k5, const float* k3, const float* k1, const float* a3,
                                  const float* a5, const float* g, int nslots, int co, int ci, int dtype,
                                  void* wf, void* wd, void* stream) {
  RM_REQUIRE(k5 && k3 && k1 && a3 && a5 && g, "gatrep_fwd: null pointer");
  RM_REQUIRE(wf || wd, "gatrep_fwd: at least one of wf / wd must be given");
  RM_REQUIRE(nslots > 0 && co > 0 && ci > 0, "gatrep_fwd: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "gatrep_fwd: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == REPMODE_F32) return gatrep_fwd_t<float>(k5, k3, k1, a3, a5, g, nslots, co, ci, dtype, wf, wd, s);
  return gatrep_fwd_t<bf16_t>(k5, k3, k1, a3, a5, g, nslots, co, ci, dtype, wf, wd, s);
}

extern "C" int repmode_gatrep_bwd(const float* dw, const float* k5, const float* k3, const float* k1,
                                  const float* a3, const float* a5, const float* g, const int32_t* slot_task,
                                  int nslots, int num_tasks, int co, int ci, float* dk5, float* dk3, float* dk1,
                                  float* da3, float* da5, float* dgate_w, float* dgate_b, float* dg_ws,
                                  void* stream) {
  RM_REQUIRE(dw && k5 && k3 && k1 && a3 && a5 && g && slot_task, "gatrep_bwd: null input");
  RM_REQUIRE(dk5 && dk3 && dk1 && da3 && da5 && dgate_w && dgate_b && dg_ws, "gatrep_bwd: null output");
  RM_REQUIRE(nslots > 0 && num_tasks > 0 && co > 0 && ci > 0, "gatrep_bwd: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  RM_HIP(hipMemsetAsync(dg_ws, 0, (size_t)nslots * E * co * sizeof(float), s));
  const int bt = ci >= 256 ? 256 : (ci > 128 ? 256 : (ci > 64 ? 128 : 64));
  repmode_prof_begin(REPMODE_PROF_GATREP_BWD, (double)co * ci * 4.0 * (125.0 * nslots + 2 * 155.0), s);
  hipLaunchKernelGGL(gatrep_bwd_kernel, dim3(ceil_div(ci, bt), co), dim3(bt), 0, s, dw, k5, k3, k1, a3, a5, g,
                     nslots, co, ci, dk5, dk3, dk1, da3, da5, dg_ws);
  RM_LAUNCH_CHECK("gatrep_bwd");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(ceil_div(E * co, 128)), dim3(128), 0, s, g, dg_ws, slot_task, nslots,
                     num_tasks, co, dgate_w, dgate_b);
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("gate_bwd");
  return REPMODE_OK;
}
