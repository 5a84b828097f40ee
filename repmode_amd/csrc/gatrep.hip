// gatrep.hip -- gating re-parameterization (GatRep) forward and backward.
//
// Forward replaces fnet/nn_modules/RepMode.py:44-49 (one-hot), :198-200 (gate Linear + softmax
// over experts), :165-169 / :173-180 (expert padding, avg-pool-as-kernel) and :182-190 (the
// per-sample weighted sum).  The merged filter depends on the task only, so it is produced once
// per distinct task ("slot") of the batch, directly in the layouts the conv kernels read:
//   wf[slot][tap][co tile][ci chunk][32][KC]        forward filter (fragment-major, see below)
//   wd[slot][124-tap][ci tile][co chunk][32][KC]    data-gradient filter (taps flipped, channels transposed)
// Memory-bound: reads 155 expert floats, writes 125 (x2) merged elements per (co, ci) per slot.
//
// Backward is the autograd of the same lines: expert gradients, gate-probability gradients
// (reduced over ci and taps), then the softmax Jacobian and the gate Linear's weight/bias grads.
#include "common.h"

#include <atomic>
#include <cstdlib>
#include "tail_jobs.h"

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

// several blocks' gates from one launch (the per-expert blocks of a forward pass: one "slot" per sample; round 3 launched one
// 6 us kernel per block)
constexpr int GSM_MAX = REPMODE_GATREP_MULTI_MAX;
struct GateMultiArgs {
  const float* gate_w[GSM_MAX]; const float* gate_b[GSM_MAX]; float* g[GSM_MAX];
  int co[GSM_MAX], first[GSM_MAX + 1];
  int nblocks, nslots, num_tasks;
};
__global__ __launch_bounds__(256) void gate_softmax_multi_kernel(GateMultiArgs a, const int32_t* __restrict__ slot_task) {
  int i = 0;
  while (i + 1 < a.nblocks && (int)blockIdx.x >= a.first[i + 1]) ++i;
  const int idx = ((int)blockIdx.x - a.first[i]) * 256 + threadIdx.x;
  const int co = a.co[i];
  if (idx >= a.nslots * co) return;
  const float* __restrict__ gate_w = a.gate_w[i];
  const float* __restrict__ gate_b = a.gate_b[i];
  const int s = idx / co, o = idx % co;
  const int task = slot_task[s];
  float logit[E], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    logit[e] = gate_w[(size_t)(e * co + o) * a.num_tasks + task] + gate_b[e * co + o];
    mx = fmaxf(mx, logit[e]);
  }
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < E; ++e) { logit[e] = expf(logit[e] - mx); sum += logit[e]; }
  const float inv = 1.f / sum;
#pragma unroll
  for (int e = 0; e < E; ++e) a.g[i][((size_t)s * E + e) * co + o] = logit[e] * inv;
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

// Forward merge into the conv kernels' FRAGMENT-MAJOR filter layout:
//
//     w[slot][tap][row tile (32 rows)][reduction chunk (KC)][row % 32][red % KC]
//
// i.e. every 32 x KC tile one MFMA "A" fragment load reads is 1 KiB contiguous (KC = 16 bf16 / 8 f32
// reduction channels; lane l of the conv kernel reads bytes [16 l, 16 l + 16) of it).  For the forward
// filter wf rows = co, reduction = ci; for the data-gradient filter wd rows = ci, reduction = co and
// the taps are flipped.  The expert tensors are [co][ci][taps] (taps contiguous), so a workgroup owns
// one tile, walks it in 8 groups of 4 rows, stages the 4 x KC x 125 expert values of a group in LDS
// (coalesced reads along taps) and writes the merged values of every slot with lanes running along
// the tile's memory order (4 * KC contiguous elements per tap).  The grid covers the padded tensor,
// so padding is written as zeros here (no memset).
template <typename T>
struct FragGeom {
  static constexpr int KC = 32 / sizeof(T);      // 16 (bf16) / 8 (f32): reduction channels per chunk
  static constexpr int PAIRS = 4 * KC;           // (row, red) pairs per group of 4 rows
  static constexpr int TQ = 256 / (PAIRS / 2);   // tap phases (a thread owns two adjacent red channels)
};

template <typename T>
__device__ __forceinline__ void store_pair(T* p, float a, float b);
template <>
__device__ __forceinline__ void store_pair<float>(float* p, float a, float b) {
  *reinterpret_cast<f32x2*>(p) = f32x2{a, b};
}
template <>
__device__ __forceinline__ void store_pair<bf16_t>(bf16_t* p, float a, float b) {
  *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(a, b);
}

// LDS of one workgroup of the forward merge (shared by both roles)
template <typename T>
struct GatrepFwdLds {
  float s5[FragGeom<T>::PAIRS * TAPS];
  float s3[FragGeom<T>::PAIRS * 27];
  float s1[FragGeom<T>::PAIRS], sa3[FragGeom<T>::PAIRS], sa5[FragGeom<T>::PAIRS];
  float sg[16][E][FragGeom<T>::KC];
};

// one workgroup: (reduction chunk kc, row tile rt, zidx = group of 4 rows x tap part); 256 threads =
// (PAIRS / 2 element pairs) x TQ tap phases
// Gate probabilities of slots [s0, s0 + ns) for the KC columns this workgroup needs, into L.sg.  Either read from g, or
// (gate_w != nullptr) computed here -- g[s][e][o] = softmax_e(gate_w[e*Co+o][task_s] + gate_b[e*Co+o]), RepMode.py:198-200
// -- which folds the gate softmax launch into this one; the forward-filter workgroups of reduction chunk 0 / tap part 0
// then also write their four channels to g_out (every channel exactly once), for the backward pass.
struct GateSrc {
  const float* g;          // precomputed probabilities [S][5][Co], or nullptr
  const float* gate_w;     // [5*Co][T]
  const float* gate_b;     // [5*Co]
  const int32_t* slot_task;
  int num_tasks;
  float* g_out;            // [S][5][Co] (may be nullptr)
};

template <typename T, bool WRITE_WD>
__device__ __forceinline__ void load_gate(GatrepFwdLds<T>& L, const GateSrc& gs, int s0, int ns, int co_n, int kc, int rt, int grp,
                                          bool writer) {
  constexpr int KC = FragGeom<T>::KC;
  const int tid = threadIdx.x;
  if (gs.gate_w == nullptr) {
    for (int i = tid; i < ns * E * KC; i += 256) {
      const int col = i % KC, e = (i / KC) % E, sl = i / (KC * E);
      // column -> co: wf: rows grp*4 + col (col < 4), wd: reduction channels kc*KC + col
      const int co = WRITE_WD ? kc * KC + col : rt * 32 + grp * 4 + (col & 3);
      L.sg[sl][e][col] = gs.g[((size_t)(s0 + sl) * E + e) * co_n + min(co, co_n - 1)];
    }
    return;
  }
  for (int i = tid; i < ns * KC; i += 256) {
    const int col = i % KC, sl = i / KC;
    const int cog = WRITE_WD ? kc * KC + col : rt * 32 + grp * 4 + (col & 3);
    const int co = min(cog, co_n - 1);
    const int task = gs.slot_task[s0 + sl];
    float logit[E], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      logit[e] = gs.gate_w[(size_t)(e * co_n + co) * gs.num_tasks + task] + gs.gate_b[e * co_n + co];
      mx = fmaxf(mx, logit[e]);
    }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { logit[e] = expf(logit[e] - mx); sum += logit[e]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const float p = logit[e] * inv;
      L.sg[sl][e][col] = p;
      if (writer && !WRITE_WD && col < 4 && cog < co_n && gs.g_out) gs.g_out[((size_t)(s0 + sl) * E + e) * co_n + cog] = p;
    }
  }
}

template <typename T, bool WRITE_WD>
__device__ __forceinline__ void gatrep_fwd_body(
    GatrepFwdLds<T>& L, const float* __restrict__ k5, const float* __restrict__ k3, const float* __restrict__ k1,
    const float* __restrict__ a3, const float* __restrict__ a5, const GateSrc& gs, int nslots,
    int co_n, int ci_n, int nrt, int nkc, int tsplit, T* __restrict__ wout, int kc, int rt, int zidx) {
  using G = FragGeom<T>;
  constexpr int KC = G::KC, PAIRS = G::PAIRS, TQ = G::TQ;
  float* s5 = L.s5; float* s3 = L.s3; float* s1 = L.s1; float* sa3 = L.sa3; float* sa5 = L.sa5;
  auto& sg = L.sg;
  const int tid = threadIdx.x;
  const int grp = zidx / tsplit, part = zidx % tsplit;
  const size_t tile_elems = 32 * KC;
  const size_t tap_stride = (size_t)nrt * nkc * tile_elems;
  const size_t slot_stride = (size_t)TAPS * tap_stride;
  // ---- stage the group's expert values: one wave copies one (row, red) pair's 125 + 27 contiguous
  // floats at a time (pairs taken in MEMORY order so that consecutive pairs are adjacent in HBM)
  {
    // all global loads of the workgroup are issued before the first LDS store (a load -> store loop
    // serialises on memory latency: ~50 us per launch on the small layers)
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NIT = PAIRS / 4;
    float va[NIT], vb[NIT], vc[NIT];
    // (round 3: buffer loads with 32-bit offsets -- a dead pair or a lane beyond the row reads through an out-of-range offset,
    // i.e. zero; the predicated 64-bit-address form was ~60 instructions per iteration and made the launch issue-bound)
    const __amdgpu_buffer_rsrc_t r5 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k5), 0, co_n * ci_n * TAPS * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(k3), 0, co_n * ci_n * 27 * 4, 0x00020000);
    constexpr int OOB = 0x7fffffff;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pm = wave + 4 * it;
      const int rr = WRITE_WD ? pm % 4 : pm / KC;
      const int kk = WRITE_WD ? pm / 4 : pm % KC;
      const int row = rt * 32 + grp * 4 + rr, red = kc * KC + kk;
      const int co = WRITE_WD ? red : row, ci = WRITE_WD ? row : red;
      const bool live = co < co_n && ci < ci_n;
      const int oi = co * ci_n + ci;
      va[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r5, live ? (oi * TAPS + lane) * 4 : OOB, 0, 0));
      vb[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r5, (live && lane + 64 < TAPS) ? (oi * TAPS + lane + 64) * 4 : OOB, 0, 0));
      vc[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r3, (live && lane < 27) ? (oi * 27 + lane) * 4 : OOB, 0, 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pm = wave + 4 * it;
      const int rr = WRITE_WD ? pm % 4 : pm / KC;
      const int kk = WRITE_WD ? pm / 4 : pm % KC;
      float* dst5 = s5 + (rr * KC + kk) * TAPS;
      dst5[lane] = va[it];
      if (lane + 64 < TAPS) dst5[lane + 64] = vb[it];
      if (lane < 27) s3[(rr * KC + kk) * 27 + lane] = vc[it];
    }
  }
  if (tid < PAIRS) {
    const int rr = tid / KC, kk = tid % KC;
    const int row = rt * 32 + grp * 4 + rr, red = kc * KC + kk;
    const int co = WRITE_WD ? red : row, ci = WRITE_WD ? row : red;
    const bool live = co < co_n && ci < ci_n;
    const size_t oi = live ? (size_t)co * ci_n + ci : 0;
    s1[tid] = live ? k1[oi] : 0.f;
    sa3[tid] = live ? a3[oi] * (1.0f / 27.0f) : 0.f;
    sa5[tid] = live ? a5[oi] * (1.0f / 125.0f) : 0.f;
  }
  // gate probabilities of the first 16 slots: loaded (or computed) here, with everything else the workgroup needs,
  // before the first barrier (one round trip to memory instead of two when the inputs are cold)
  const bool gate_writer = !WRITE_WD && kc == 0 && part == 0;
  load_gate<T, WRITE_WD>(L, gs, 0, min(16, nslots), co_n, kc, rt, grp, gate_writer);
  __syncthreads();
  // ---- merge: this thread owns elements pr and pr + 1 (adjacent reduction channels of one row)
  const int p2 = tid % (PAIRS / 2), tq = tid / (PAIRS / 2);
  const int pr = 2 * p2;
  const int r4 = pr / KC, k = pr % KC;
  const int row = rt * 32 + grp * 4 + r4, red = kc * KC + k;
  const size_t in_tile = (size_t)(grp * 4 + r4) * KC + k;
  // gate probabilities of the channels this workgroup touches go through LDS once (chunks of 16 slots):
  // the merge loop then has no dependent global load in it (it was latency-bound at ~50 us per launch)
  const size_t tile_off = ((size_t)rt * nkc + kc) * tile_elems + in_tile;
  // LDS column of this thread's two elements: wf -> the row's co (column r4), wd -> the reduction co (columns k, k+1)
  const int colA = WRITE_WD ? k : r4, colB = WRITE_WD ? k + 1 : r4;
  // slot-independent values of this thread's taps, hoisted out of the slot loop.  The workgroups of one
  // tile group may split the taps between them (blockIdx.z = grp * tsplit + part): small layers would
  // otherwise run as a handful of long, latency-bound single-wave loops.
  constexpr int MAXT = (TAPS + TQ - 1) / TQ;
  // Round 3: the slot loop below is branch-free and its stores are buffer stores -- ONE 32-bit byte offset per tap (invalid
  // taps: an out-of-range offset, the store is dropped by the range check) and the slot's offset in an SGPR.  Before, every
  // (slot, tap) body carried two branches and a 64-bit multiply-add for its address: 6.5 k instructions per wave, a third of
  // them address arithmetic and exec-mask juggling.  Same products in the same order (a masked term is added as an exact
  // + 0, or selected away), so the merged values are bit-identical.
  bool c3j[MAXT], ctrj[MAXT];
  int voff[MAXT];
  float v5a[MAXT], v5b[MAXT], v3a[MAXT], v3b[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int phase = tq + TQ * j;                 // tap phase index; phases are dealt round-robin to the parts
    const int tap = phase;
    const bool mine = tap < TAPS && (j % tsplit) == part;
    int t3 = 0;
    c3j[j] = mine && in_centre3(tap, t3);
    ctrj[j] = mine && tap == 62;
    v5a[j] = mine ? s5[pr * TAPS + tap] : 0.f;
    v5b[j] = mine ? s5[(pr + 1) * TAPS + tap] : 0.f;
    v3a[j] = c3j[j] ? s3[pr * 27 + t3] : 0.f;
    v3b[j] = c3j[j] ? s3[(pr + 1) * 27 + t3] : 0.f;
    const int tap_out = WRITE_WD ? TAPS - 1 - tap : tap;
    voff[j] = mine ? (int)(((size_t)tap_out * tap_stride + tile_off) * sizeof(T)) : 0x7fffffff;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      wout, 0, (int)((size_t)nslots * slot_stride * sizeof(T)), 0x00020000);       // (< 2 GiB: checked by the launchers)
  const float e2a = s1[pr], e2b = s1[pr + 1], e3a = sa3[pr], e3b = sa3[pr + 1], e4a = sa5[pr], e4b = sa5[pr + 1];
  for (int s0 = 0; s0 < nslots; s0 += 16) {
    const int ns = min(16, nslots - s0);
    if (s0 > 0) {
      __syncthreads();
      load_gate<T, WRITE_WD>(L, gs, s0, ns, co_n, kc, rt, grp, gate_writer);
      __syncthreads();
    }
    for (int sl = 0; sl < ns; ++sl) {
      const float ga0 = sg[sl][0][colA], ga1 = sg[sl][1][colA], ga2 = sg[sl][2][colA], ga3 = sg[sl][3][colA], ga4 = sg[sl][4][colA];
      const float gb0 = sg[sl][0][colB], gb1 = sg[sl][1][colB], gb2 = sg[sl][2][colB], gb3 = sg[sl][3][colB], gb4 = sg[sl][4][colB];
      const int soff = (int)((size_t)(s0 + sl) * slot_stride * sizeof(T));
      // same association order as RepMode.py:184-188: ((((g0 k5 + g1 k3) + g2 k1) + g3 a3) + g4 a5)
      const float ca3 = ga3 * e3a, cb3 = gb3 * e3b, ca4 = ga4 * e4a, cb4 = gb4 * e4b;
#pragma unroll
      for (int j = 0; j < MAXT; ++j) {
        float ra = ga0 * v5a[j], rb = gb0 * v5b[j];
        ra += ga1 * v3a[j];                          // (outside the centre cube v3 = 0: an exact + 0)
        rb += gb1 * v3b[j];
        const float ra2 = ra + ga2 * e2a, rb2 = rb + gb2 * e2b;
        ra = ctrj[j] ? ra2 : ra;
        rb = ctrj[j] ? rb2 : rb;
        const float ra3 = ra + ca3, rb3 = rb + cb3;
        ra = c3j[j] ? ra3 : ra;
        rb = c3j[j] ? rb3 : rb;
        ra += ca4;
        rb += cb4;
        if constexpr (sizeof(T) == 2) {
          __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(ra, rb), rs, voff[j], soff, 0);
        } else {
          const f32x2 v = {ra, rb};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff[j], soff, 0);
        }
      }
    }
  }
}

// Both filter roles in ONE launch: workgroups [0, nwf) build wf (rows = co), the rest wd (rows = ci, taps flipped).
// Either count may be zero.  On the small layers the two halves are latency-bound and simply overlap.
template <typename T>
__global__ __launch_bounds__(256) void gatrep_fwd_kernel(
    const float* __restrict__ k5, const float* __restrict__ k3, const float* __restrict__ k1,
    const float* __restrict__ a3, const float* __restrict__ a5, GateSrc gs, int nslots,
    int co_n, int ci_n, int nwf, int nrt_f, int nkc_f, int ts_f, T* __restrict__ wf, int nrt_d, int nkc_d, int ts_d,
    T* __restrict__ wd) {
  __shared__ GatrepFwdLds<T> L;
  int b = blockIdx.x;
  if (b < nwf) {
    const int kc = b % nkc_f; b /= nkc_f;
    const int rt = b % nrt_f;
    gatrep_fwd_body<T, false>(L, k5, k3, k1, a3, a5, gs, nslots, co_n, ci_n, nrt_f, nkc_f, ts_f, wf, kc, rt, b / nrt_f);
  } else {
    b -= nwf;
    const int kc = b % nkc_d; b /= nkc_d;
    const int rt = b % nrt_d;
    gatrep_fwd_body<T, true>(L, k5, k3, k1, a3, a5, gs, nslots, co_n, ci_n, nrt_d, nkc_d, ts_d, wd, kc, rt, b / nrt_d);
  }
}

constexpr int GF_CT = 32;      // channels per slab (backward kernel)
constexpr int GF_THREADS = 256;

// softmax Jacobian + gate Linear gradients.  One thread per (o, e), the five experts of a channel adjacent in one
// workgroup (160 threads = 32 channels); loops over slots.  `clear`: dg lives in the library's zero scratch --
// once every thread of the workgroup has read its channel's entries, they are zeroed again.
constexpr int GATE_BWD_THREADS = 32 * E;

__global__ __launch_bounds__(GATE_BWD_THREADS) void gate_bwd_kernel(
    const float* __restrict__ g, float* __restrict__ dg, const int32_t* __restrict__ slot_task, int nslots,
    int num_tasks, int co_n, float* __restrict__ dgate_w, float* __restrict__ dgate_b, int clear) {
  // (the body is shared with the deferred form that rides in a conv5 launch: tail_jobs.h)
  tail_gate_bwd(g, dg, slot_task, nslots, num_tasks, co_n, dgate_w, dgate_b, clear, blockIdx.x, threadIdx.x);
}


// Expert gradients and gate-probability gradients from the per-slot filter gradient dw[s][tap][co][ci].
// Block = (one co, 32 ci); thread = (ci = tid % 32, taps tid / 32 + (NTH / 32) k).  Reads of dw are 128-byte
// rows; expert gradients are accumulated over slots in registers and written back transposed through
// LDS ([ci][taps] contiguous, the experts' layout); the per-slot gate-probability partial sums are
// reduced over the block and added to dg[s][e][co] (one atomic per block, slot and expert).
// NTH threads: ci = tid % 32, tap group tq = tid / 32, taps tq + (NTH / 32) k.  Two shapes: 256 threads x 16 taps, 2 slots per
// round (large layers: two or three workgroups per CU) and 512 threads x 8 taps, 8 slots per round (small layers: the launch
// is one dependent chain of fetch -> reduce -> write, so a tile's work is spread over 8 waves).
template <int NTH, int SG>   // threads per workgroup; slots per reduction round
__global__ __launch_bounds__(NTH, NTH == 256 ? 2 : 1) void gatrep_bwd_kernel(
    const float* __restrict__ dw, const float* __restrict__ k5, const float* __restrict__ k3,
    const float* __restrict__ k1, const float* __restrict__ a3, const float* __restrict__ a5,
    const float* __restrict__ g, int nslots, int co_n, int ci_n, float* __restrict__ dk5,
    float* __restrict__ dk3, float* __restrict__ dk1, float* __restrict__ da3, float* __restrict__ da5,
    float* __restrict__ dg) {
  constexpr int NQ = NTH / GF_CT;                 // tap groups
  constexpr int NT = (TAPS + NQ - 1) / NQ;        // taps per thread
  constexpr int NW = NTH / 64;                    // waves
  constexpr int NV5 = (GF_CT * TAPS + NTH - 1) / NTH, NV3 = (GF_CT * 27 + NTH - 1) / NTH, NVG = (64 * E + NTH - 1) / NTH;
  __shared__ float s5[GF_CT * TAPS];        // k5 slab, later reused for the dk5 write-back
  __shared__ float s3[GF_CT * 27];
  __shared__ float part[SG * 5][NW][GF_CT]; // per-wave partial sums (the two tap groups of a wave already folded)
  __shared__ float cross[3][SG][GF_CT];     // per-slot contributions to dk1 / da3 / da5 of each ci
  __shared__ float sgb[64][E];
  const int tid = threadIdx.x;
  const int co = blockIdx.y;
  // A workgroup takes the 32-channel tiles bx, bx + gridDim.x, ... of its output channel.  Normally gridDim.x = the number of
  // tiles (one each); deterministic mode launches gridDim.x = 1, so that ONE thread adds a (slot, expert, co) entry of dg tile
  // after tile in program order (the tiles' atomics would otherwise meet in any order).
  const int ntile = (ci_n + GF_CT - 1) / GF_CT;
  for (int bx = blockIdx.x; bx < ntile; bx += gridDim.x) {
  if (bx != (int)blockIdx.x) __syncthreads();          // (the previous tile's LDS reads are done)
  const int c0 = bx * GF_CT;
  const int nlive = min(GF_CT, ci_n - c0);
  const size_t base = (size_t)co * ci_n + c0;
  const int c = tid & (GF_CT - 1), tq = tid / GF_CT;
  const bool live = c < nlive;
  const size_t oi = base + (live ? c : 0);
  const size_t tap_stride = (size_t)co_n * ci_n;            // (125 co ci < 2^30: checked by the launcher)
  // Every global load this workgroup needs before its first reduction is issued up front, BEFORE anything waits: the
  // first round's filter gradients (SG x NT per thread), the expert slabs, the 1x1 experts and the gate
  // probabilities.  In the train step these are cold (the filter gradient was just written by another kernel): the
  // prologue used to be three dependent round trips to HBM (50-70 us per launch cold vs 25 us warm).
  // Addresses: buffer loads -- a descriptor at the first slot of the round, the slot's byte offset in an SGPR and a 32-bit
  // per-lane offset per tap shared by all slots (per-load 64-bit addresses spilled).  Invalid taps and dead lanes load element
  // 0 of the tile (a valid address) unmasked: everything such a value is multiplied into is a zero constant (below) or is
  // never written back.
  const float* __restrict__ dwt = dw + base;
  const unsigned slot_bytes = (unsigned)(TAPS * tap_stride * sizeof(float));      // (8 slots < 4 GiB: checked by the launcher)
  bool tap_on[NT];
  int offk[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    tap_on[k] = tq + NQ * k < TAPS && live;
    offk[k] = tap_on[k] ? (int)(((unsigned)(tq + NQ * k) * (unsigned)tap_stride + (unsigned)c) * sizeof(float)) : 0;
  }
  float dpre[SG][NT];
  {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dwt), 0, -1, 0x00020000);
#pragma unroll
    for (int j = 0; j < SG; ++j)
#pragma unroll
      for (int k = 0; k < NT; ++k)
        dpre[j][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, offk[k], min(j, nslots - 1) * slot_bytes, 0));
  }
  const float w1 = live ? k1[oi] : 0.f;
  const float w3 = live ? a3[oi] * (1.0f / 27.0f) : 0.f;
  const float w5 = live ? a5[oi] * (1.0f / 125.0f) : 0.f;
  {
    float v5[NV5], v3[NV3], vg[NVG];
#pragma unroll
    for (int j = 0; j < NV5; ++j) { const int i = tid + j * NTH; v5[j] = i < nlive * TAPS ? k5[base * TAPS + i] : 0.f; }
#pragma unroll
    for (int j = 0; j < NV3; ++j) { const int i = tid + j * NTH; v3[j] = i < nlive * 27 ? k3[base * 27 + i] : 0.f; }
    // this output channel's gate probabilities for every slot (up to 64 slots through LDS: 320 values)
#pragma unroll
    for (int j = 0; j < NVG; ++j) {
      const int i = tid + j * NTH;
      vg[j] = i < min(nslots, 64) * E ? g[((size_t)(i / E) * E + i % E) * co_n + co] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NV5; ++j) { const int i = tid + j * NTH; if (i < GF_CT * TAPS) s5[i] = v5[j]; }
#pragma unroll
    for (int j = 0; j < NV3; ++j) { const int i = tid + j * NTH; if (i < GF_CT * 27) s3[i] = v3[j]; }
#pragma unroll
    for (int j = 0; j < NVG; ++j) { const int i = tid + j * NTH; if (i < 64 * E) sgb[i / E][i % E] = vg[j]; }
  }
  float acc5[NT], acc3[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) { acc5[k] = 0.f; acc3[k] = 0.f; }
  float acc1 = 0.f, acca3 = 0.f, acca5 = 0.f;     // meaningful in threads tid < 32 only
  __syncthreads();
  // Slots are taken SG at a time: their filter-gradient loads are independent (all in flight together) and the
  // per-slot partial sums stay in registers until ONE block reduction per round.  (One load -> reduce -> barrier
  // round trip per slot made every launch 50-60 us whatever the layer size.)
  const int rj = tid / GF_CT;               // reduction phase: this thread reduces slot s0 + rj for ci c
  // Per-tap constants of this thread, so that the slot loop below is branch-free (the unrolled slots x taps bodies with their
  // own predicates and exec-mask juggling were ~9 k instructions per wave: the small layers' launches were bound by VALU
  // issue, not by memory).  Invalid taps / dead lanes load d = 0 and carry zero constants.
  constexpr int K_C = 62 / NQ;              // the centre tap (62) is tap tq + NQ k of thread group tq = 62 % NQ at k = K_C
  float s5v[NT], s3v[NT], m1[NT], m3[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int tap = tq + NQ * k;
    int t3;
    const bool c3 = in_centre3(tap, t3) && tap_on[k];
    s5v[k] = tap_on[k] ? s5[c * TAPS + tap] : 0.f;     // (dead lanes must not touch the uninitialised slab tail)
    s3v[k] = c3 ? s3[c * 27 + t3] : 0.f;
    m1[k] = tap_on[k] ? 1.f : 0.f;
    m3[k] = c3 ? 1.f : 0.f;
  }
  const float mc = (tq == 62 % NQ && live) ? 1.f : 0.f;
  for (int s0 = 0; s0 < nslots; s0 += SG) {
    const int ns = min(SG, nslots - s0);
    float q[SG][5];
#pragma unroll
    for (int j = 0; j < SG; ++j) {
#pragma unroll
      for (int v = 0; v < 5; ++v) q[j][v] = 0.f;
      if (j < ns) {
        const int s = s0 + j;
        const bool in_lds = s < 64;
        const float* gs = g + (size_t)s * E * co_n + co;
        const float g0 = in_lds ? sgb[s][0] : gs[0], g1 = in_lds ? sgb[s][1] : gs[co_n];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
          const float d = dpre[j][k];
          q[j][4] = fmaf(m1[k], d, q[j][4]);
          q[j][0] = fmaf(s5v[k], d, q[j][0]);
          q[j][3] = fmaf(m3[k], d, q[j][3]);
          q[j][1] = fmaf(s3v[k], d, q[j][1]);
          if (k == K_C) q[j][2] = fmaf(mc, d, q[j][2]);
          acc5[k] = fmaf(g0, d, acc5[k]);          // (invalid taps, and for acc3 taps outside the centre cube: never written back)
          acc3[k] = fmaf(g1, d, acc3[k]);
        }
      }
    }
    // the next round's filter gradients are requested before this round's reduction
    if (s0 + SG < nslots) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dwt + (size_t)(s0 + SG) * TAPS * tap_stride), 0, -1, 0x00020000);
      const int nn = nslots - (s0 + SG);
#pragma unroll
      for (int j = 0; j < SG; ++j)
#pragma unroll
        for (int k = 0; k < NT; ++k)
          dpre[j][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, offk[k], min(j, nn - 1) * slot_bytes, 0));
    }
    // the two tap groups of a wave (lanes l and l + 32: same ci) fold in registers; lanes 0..31 store the wave's sums
#pragma unroll
    for (int j = 0; j < SG; ++j)
#pragma unroll
      for (int v = 0; v < 5; ++v) {
        const float t = q[j][v] + __shfl_xor(q[j][v], 32);
        if ((tid & 32) == 0) part[j * 5 + v][tid >> 6][c] = t;
      }
    __syncthreads();
    {
      // thread (c, rj): totals of slot s0 + rj for ci c over the waves
      float r[5];
#pragma unroll
      for (int v = 0; v < 5; ++v) {
        r[v] = 0.f;
        if (rj < SG) {
#pragma unroll
          for (int t = 0; t < NW; ++t) r[v] += part[rj * 5 + v][t][c];
        }
      }
      const int s = s0 + rj;
      const bool on = rj < ns;
      const bool in_lds = s < 64;
      const float* gs = g + (size_t)(on ? s : 0) * E * co_n + co;
      const float g2 = !on ? 0.f : in_lds ? sgb[s][2] : gs[2 * co_n];
      const float g3 = !on ? 0.f : in_lds ? sgb[s][3] : gs[3 * co_n];
      const float g4 = !on ? 0.f : in_lds ? sgb[s][4] : gs[4 * co_n];
      if (rj < SG) {
        cross[0][rj][c] = g2 * r[2];
        cross[1][rj][c] = g3 * r[3];
        cross[2][rj][c] = g4 * r[4];
      }
      // dg[s][e][co] += sum over this block's ci (one atomic per block, slot and expert)
      if (rj < SG) {                       // (wave-uniform: rj is constant over a half-wave, SG is even)
        float e[E] = {r[0], r[1], w1 * r[2], w3 * r[3], w5 * r[4]};
#pragma unroll
        for (int k = 0; k < E; ++k) {
          float v = e[k];
#pragma unroll
          for (int off = 16; off > 0; off >>= 1) v += __shfl_down(v, off, 32);
          if (c == 0 && on) atomicAdd(dg + ((size_t)s * E + k) * co_n + co, v);
        }
      }
    }
    __syncthreads();
    if (tid < GF_CT) {
#pragma unroll
      for (int j = 0; j < SG; ++j) { acc1 += cross[0][j][tid]; acca3 += cross[1][j][tid]; acca5 += cross[2][j][tid]; }
    }
  }
  __syncthreads();
  if (tid < GF_CT && live) {
    dk1[oi] = acc1;
    da3[oi] = acca3 * (1.0f / 27.0f);
    da5[oi] = acca5 * (1.0f / 125.0f);
  }
  // write dk5 / dk3 back in the experts' [ci][taps] layout through LDS (slabs are contiguous)
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int tap = tq + NQ * k;
    if (tap < TAPS) {
      s5[c * TAPS + tap] = acc5[k];
      int t3;
      if (in_centre3(tap, t3)) s3[c * 27 + t3] = acc3[k];
    }
  }
  __syncthreads();
  for (int i = tid; i < nlive * TAPS; i += NTH) dk5[base * TAPS + i] = s5[i];
  for (int i = tid; i < nlive * 27; i += NTH) dk3[base * 27 + i] = s3[i];
  }   // (tile loop)
}

}  // namespace

extern "C" int repmode_gate_softmax(const float* gate_w, const float* gate_b, const int32_t* slot_task,
                                    int nslots, int num_tasks, int co, float* g, void* stream) {
  RM_REQUIRE(gate_w && gate_b && slot_task && g, "gate_softmax: null pointer");
  RM_REQUIRE(nslots > 0 && num_tasks > 0 && co > 0, "gate_softmax: bad shape");
  const int total = nslots * co;
  // (recorded with the GatRep forward family: gate + GatRep + conv is the unit BASELINE's forward target names)
  repmode_prof_begin(REPMODE_PROF_GATREP_FWD, (double)total * 4.0 * (2 * E + E), static_cast<hipStream_t>(stream));
  hipLaunchKernelGGL(gate_softmax_kernel, dim3(ceil_div(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), gate_w, gate_b, slot_task, nslots, num_tasks, co, g);
  repmode_prof_end(static_cast<hipStream_t>(stream));
  RM_LAUNCH_CHECK("gate_softmax");
  return REPMODE_OK;
}

extern "C" int repmode_gate_softmax_multi(int nblocks, const float* const* gate_w, const float* const* gate_b, const int* co,
                                          const int32_t* slot_task, int nslots, int num_tasks, float* const* g, void* stream) {
  RM_REQUIRE(gate_w && gate_b && co && slot_task && g, "gate_softmax_multi: null pointer");
  RM_REQUIRE(nblocks > 0 && nblocks <= GSM_MAX, "gate_softmax_multi: 1..%d blocks per call, got %d", GSM_MAX, nblocks);
  RM_REQUIRE(nslots > 0 && num_tasks > 0, "gate_softmax_multi: bad shape");
  GateMultiArgs a{};
  a.nblocks = nblocks; a.nslots = nslots; a.num_tasks = num_tasks;
  long total = 0;
  double bytes = 0;
  for (int i = 0; i < nblocks; ++i) {
    RM_REQUIRE(gate_w[i] && gate_b[i] && g[i] && co[i] > 0, "gate_softmax_multi: bad block %d", i);
    a.gate_w[i] = gate_w[i]; a.gate_b[i] = gate_b[i]; a.g[i] = g[i]; a.co[i] = co[i];
    a.first[i] = (int)total;
    total += ceil_div(nslots * co[i], 256);
    bytes += (double)nslots * co[i] * 4.0 * (2 * E + E);
  }
  a.first[nblocks] = (int)total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  repmode_prof_begin(REPMODE_PROF_GATREP_FWD, bytes, s);
  hipLaunchKernelGGL(gate_softmax_multi_kernel, dim3((unsigned)total), dim3(256), 0, s, a, slot_task);
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("gate_softmax_multi");
  return REPMODE_OK;
}

template <typename T>
static int gatrep_fwd_t(const float* k5, const float* k3, const float* k1, const float* a3, const float* a5,
                        GateSrc gs, int nslots, int co, int ci, int dtype, void* wf, void* wd, hipStream_t s) {
  constexpr int KC = FragGeom<T>::KC;
  // algorithmic bytes: 155 expert floats read once + 125 merged elements written per slot and layout
  const double bytes = (double)co * ci * (155.0 * 4 + 125.0 * nslots * sizeof(T) * ((wf ? 1 : 0) + (wd ? 1 : 0)));
  repmode_prof_begin(REPMODE_PROF_GATREP_FWD, bytes, s);
  // wf: rows = co (padded to 32), reduction = ci (padded to KC); wd: rows = ci, reduction = co, taps flipped
  int nrt_f = 1, nkc_f = 1, ts_f = 1, nrt_d = 1, nkc_d = 1, ts_d = 1;
  long nwf = 0, nwd = 0;
  if (wf) {
    nrt_f = repmode_padded_channels(co, dtype, 0) / 32; nkc_f = repmode_padded_channels(ci, dtype, 1) / KC;
    ts_f = (long)nkc_f * nrt_f * 8 < 512 ? 4 : 1;   // small layers: split the taps over 4x more workgroups
    nwf = (long)nkc_f * nrt_f * 8 * ts_f;
  }
  if (wd) {
    nrt_d = repmode_padded_channels(ci, dtype, 0) / 32; nkc_d = repmode_padded_channels(co, dtype, 1) / KC;
    ts_d = (long)nkc_d * nrt_d * 8 < 512 ? 4 : 1;
    nwd = (long)nkc_d * nrt_d * 8 * ts_d;
  }
  RM_REQUIRE(nwf + nwd < (1L << 31), "gatrep_fwd: grid too large");
  RM_REQUIRE((double)nslots * TAPS * repmode_padded_channels(co, dtype, 0) * repmode_padded_channels(ci, dtype, 0) * sizeof(T) < 2.0e9,
             "gatrep_fwd: the merged filters of all slots must stay below 2 GB (32-bit store offsets)");
  hipLaunchKernelGGL((gatrep_fwd_kernel<T>), dim3((unsigned)(nwf + nwd)), dim3(256), 0, s, k5, k3, k1, a3, a5, gs, nslots, co,
                     ci, (int)nwf, nrt_f, nkc_f, ts_f, static_cast<T*>(wf), nrt_d, nkc_d, ts_d, static_cast<T*>(wd));
  RM_LAUNCH_CHECK("gatrep_fwd");
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
  const GateSrc gs{g, nullptr, nullptr, nullptr, 0, nullptr};
  if (dtype == REPMODE_F32) return gatrep_fwd_t<float>(k5, k3, k1, a3, a5, gs, nslots, co, ci, dtype, wf, wd, s);
  return gatrep_fwd_t<bf16_t>(k5, k3, k1, a3, a5, gs, nslots, co, ci, dtype, wf, wd, s);
}

// Gate softmax + GatRep forward in ONE launch (RepMode.py:198-200 + :165-192): the workgroups compute the probabilities
// of the channels they merge from gate_w / gate_b / slot_task themselves; g_out [nslots][5][co] receives them for the
// backward pass (needs wf: the forward-filter workgroups are the ones that write it).
extern "C" int repmode_gatrep_fwd_gate(const float* k5, const float* k3, const float* k1, const float* a3, const float* a5,
                                       const float* gate_w, const float* gate_b, const int32_t* slot_task, int nslots,
                                       int num_tasks, int co, int ci, int dtype, float* g_out, void* wf, void* wd, void* stream) {
  RM_REQUIRE(k5 && k3 && k1 && a3 && a5 && gate_w && gate_b && slot_task, "gatrep_fwd_gate: null pointer");
  RM_REQUIRE(wf && g_out, "gatrep_fwd_gate: the forward filter and g_out must be given");
  RM_REQUIRE(nslots > 0 && num_tasks > 0 && co > 0 && ci > 0, "gatrep_fwd_gate: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "gatrep_fwd_gate: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const GateSrc gs{nullptr, gate_w, gate_b, slot_task, num_tasks, g_out};
  if (dtype == REPMODE_F32) return gatrep_fwd_t<float>(k5, k3, k1, a3, a5, gs, nslots, co, ci, dtype, wf, wd, s);
  return gatrep_fwd_t<bf16_t>(k5, k3, k1, a3, a5, gs, nslots, co, ci, dtype, wf, wd, s);
}

// ------------------------------------------------------------------------------------------------
// Gate softmax + GatRep forward of SEVERAL MoDE blocks in one launch.  The forward filters depend on the parameters and
// the batch's tasks only, not on activations, so a train step can merge all its blocks before the first convolution; on
// the shallow levels these launches are 17-27 us of latency each for a few MB, which one grid amortises.
namespace {
constexpr int GM_MAX = REPMODE_GATREP_MULTI_MAX;
template <typename T>
struct GatrepMultiArgs {
  const float* k5[GM_MAX]; const float* k3[GM_MAX]; const float* k1[GM_MAX]; const float* a3[GM_MAX]; const float* a5[GM_MAX];
  const float* gate_w[GM_MAX]; const float* gate_b[GM_MAX];
  float* g_out[GM_MAX];
  T* wf[GM_MAX]; T* wd[GM_MAX];
  int co[GM_MAX], ci[GM_MAX], nwf[GM_MAX], nrt_f[GM_MAX], nkc_f[GM_MAX], ts_f[GM_MAX], nrt_d[GM_MAX], nkc_d[GM_MAX], ts_d[GM_MAX];
  int first[GM_MAX + 1];          // first workgroup of each block (prefix sums)
  int nblocks, nslots, num_tasks;
  const int32_t* slot_task;
  int order;          // 1: a tile's eight row groups on one XCD, dispatched together (see the kernel)
};

template <typename T>
__global__ __launch_bounds__(256) void gatrep_fwd_multi_kernel(GatrepMultiArgs<T> a) {
  __shared__ GatrepFwdLds<T> L;
  int i = 0;
  while (i + 1 < a.nblocks && (int)blockIdx.x >= a.first[i + 1]) ++i;       // (uniform; at most GM_MAX - 1 steps)
  int b = blockIdx.x - a.first[i];
  const GateSrc gs{nullptr, a.gate_w[i], a.gate_b[i], a.slot_task, a.num_tasks, a.g_out[i]};
  // Which (reduction chunk, row tile, tap part, group of 4 rows) a workgroup takes.  The eight groups of one 1 KiB fragment
  // tile each write a 128-byte piece of it per (slot, tap); with the group as the SLOWEST index (round 2) those pieces came
  // from workgroups thousands of launch slots apart, on different XCDs -- isolated 128-byte writes at a 1 KiB pitch.  Now
  // (REPMODE_GATREP_ORDER, default 1): workgroups b, b + 8, ... b + 56 of a block of 64 -- same XCD (b mod 8), dispatched
  // together -- are the eight groups of one tile, so a tile's pieces meet in one L2 within microseconds.
  auto decode = [&](int bb, int nkc, int nrt, int ts, int& kc, int& rt, int& zidx) {
    const int ntile = nkc * nrt * ts;
    if (a.order && (ntile & 7) == 0) {
      const int x = bb & 7, q = bb >> 3;
      const int grp = q & 7, t = (q >> 3) * 8 + x;
      kc = t % nkc;
      rt = (t / nkc) % nrt;
      zidx = grp * ts + t / (nkc * nrt);
    } else {
      kc = bb % nkc; bb /= nkc;
      rt = bb % nrt;
      zidx = bb / nrt;
    }
  };
  int kc, rt, zidx;
  if (b < a.nwf[i]) {
    decode(b, a.nkc_f[i], a.nrt_f[i], a.ts_f[i], kc, rt, zidx);
    gatrep_fwd_body<T, false>(L, a.k5[i], a.k3[i], a.k1[i], a.a3[i], a.a5[i], gs, a.nslots, a.co[i], a.ci[i], a.nrt_f[i], a.nkc_f[i],
                              a.ts_f[i], a.wf[i], kc, rt, zidx);
  } else {
    b -= a.nwf[i];
    decode(b, a.nkc_d[i], a.nrt_d[i], a.ts_d[i], kc, rt, zidx);
    gatrep_fwd_body<T, true>(L, a.k5[i], a.k3[i], a.k1[i], a.a3[i], a.a5[i], gs, a.nslots, a.co[i], a.ci[i], a.nrt_d[i], a.nkc_d[i],
                             a.ts_d[i], a.wd[i], kc, rt, zidx);
  }
}

template <typename T>
int gatrep_fwd_multi_t(int nblocks, const float* const* k5, const float* const* k3, const float* const* k1, const float* const* a3,
                       const float* const* a5, const float* const* gate_w, const float* const* gate_b, const int* co, const int* ci,
                       const int32_t* slot_task, int nslots, int num_tasks, int dtype, float* const* g_out, void* const* wf,
                       void* const* wd, hipStream_t s) {
  constexpr int KC = FragGeom<T>::KC;
  GatrepMultiArgs<T> a{};
  a.nblocks = nblocks; a.nslots = nslots; a.num_tasks = num_tasks; a.slot_task = slot_task;
  static const int order = []() { const char* e = getenv("REPMODE_GATREP_ORDER"); return e ? atoi(e) : 1; }();
  a.order = order;
  long total = 0;
  double bytes = 0;
  for (int i = 0; i < nblocks; ++i) {
    RM_REQUIRE(k5[i] && k3[i] && k1[i] && a3[i] && a5[i] && gate_w[i] && gate_b[i] && (wf[i] || wd[i]) && (g_out[i] || !wf[i]),
               "gatrep_fwd_multi: null pointer (block %d)", i);
    RM_REQUIRE(co[i] > 0 && ci[i] > 0, "gatrep_fwd_multi: bad shape (block %d)", i);
    RM_REQUIRE((double)nslots * TAPS * repmode_padded_channels(co[i], dtype, 0) * repmode_padded_channels(ci[i], dtype, 0) * sizeof(T) < 2.0e9,
               "gatrep_fwd_multi: the merged filters of all slots must stay below 2 GB (32-bit store offsets; block %d)", i);
    a.k5[i] = k5[i]; a.k3[i] = k3[i]; a.k1[i] = k1[i]; a.a3[i] = a3[i]; a.a5[i] = a5[i];
    a.gate_w[i] = gate_w[i]; a.gate_b[i] = gate_b[i]; a.g_out[i] = g_out[i];
    a.wf[i] = static_cast<T*>(wf[i]); a.wd[i] = static_cast<T*>(wd[i]);
    a.co[i] = co[i]; a.ci[i] = ci[i];
    a.nrt_f[i] = repmode_padded_channels(co[i], dtype, 0) / 32; a.nkc_f[i] = repmode_padded_channels(ci[i], dtype, 1) / KC;
    a.ts_f[i] = (long)a.nkc_f[i] * a.nrt_f[i] * 8 < 512 ? 4 : 1;
    a.nwf[i] = wf[i] ? a.nkc_f[i] * a.nrt_f[i] * 8 * a.ts_f[i] : 0;      // (no forward filter asked for: the data-gradient role only)
    long nwd = 0;
    a.nrt_d[i] = a.nkc_d[i] = a.ts_d[i] = 1;
    if (wd[i]) {
      a.nrt_d[i] = repmode_padded_channels(ci[i], dtype, 0) / 32; a.nkc_d[i] = repmode_padded_channels(co[i], dtype, 1) / KC;
      a.ts_d[i] = (long)a.nkc_d[i] * a.nrt_d[i] * 8 < 512 ? 4 : 1;
      nwd = (long)a.nkc_d[i] * a.nrt_d[i] * 8 * a.ts_d[i];
    }
    a.first[i] = (int)total;
    total += a.nwf[i] + nwd;
    bytes += (double)co[i] * ci[i] * (155.0 * 4 + 125.0 * nslots * sizeof(T) * ((wf[i] ? 1 : 0) + (wd[i] ? 1 : 0)));
  }
  a.first[nblocks] = (int)total;
  RM_REQUIRE(total > 0 && total < (1L << 31), "gatrep_fwd_multi: grid out of range");
  repmode_prof_begin(REPMODE_PROF_GATREP_FWD, bytes, s);
  hipLaunchKernelGGL((gatrep_fwd_multi_kernel<T>), dim3((unsigned)total), dim3(256), 0, s, a);
  RM_LAUNCH_CHECK("gatrep_fwd_multi");
  repmode_prof_end(s);
  return REPMODE_OK;
}
}  // namespace

extern "C" int repmode_gatrep_fwd_multi(int nblocks, const float* const* k5, const float* const* k3, const float* const* k1,
                                        const float* const* a3, const float* const* a5, const float* const* gate_w,
                                        const float* const* gate_b, const int* co, const int* ci, const int32_t* slot_task,
                                        int nslots, int num_tasks, int dtype, float* const* g_out, void* const* wf,
                                        void* const* wd, void* stream) {
  RM_REQUIRE(k5 && k3 && k1 && a3 && a5 && gate_w && gate_b && co && ci && slot_task && g_out && wf && wd, "gatrep_fwd_multi: null pointer");
  RM_REQUIRE(nblocks > 0 && nblocks <= GM_MAX, "gatrep_fwd_multi: 1..%d blocks per call, got %d", GM_MAX, nblocks);
  RM_REQUIRE(nslots > 0 && num_tasks > 0, "gatrep_fwd_multi: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "gatrep_fwd_multi: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == REPMODE_F32)
    return gatrep_fwd_multi_t<float>(nblocks, k5, k3, k1, a3, a5, gate_w, gate_b, co, ci, slot_task, nslots, num_tasks, dtype, g_out, wf, wd, s);
  return gatrep_fwd_multi_t<bf16_t>(nblocks, k5, k3, k1, a3, a5, gate_w, gate_b, co, ci, slot_task, nslots, num_tasks, dtype, g_out, wf, wd, s);
}

// ------------------------------------------------------------------------------------------------
// The raw 5x5x5 and 3x3x3 experts in the conv kernels' fragment-major bf16 layouts, for the per-expert
// formulation of the deep levels (ops._ModeConv3dUnmerged): two "slots" that are not merged with anything,
//   slot 0 = conv5x5 expert (125 taps),
//   slot 1 = conv3x3 expert on the centred support: ONLY the taps with dz, dy in [1,3] are written (dx = 0 and 4
//            as zeros) -- exactly what repmode_conv5_ex's centre3 mode reads; the other 80 taps stay untouched.
// A pure layout pass (f32 [co][ci][taps] -> bf16 tiles), so unlike gatrep_fwd_kernel there is no gate, no slot
// loop, and the experts are read once for both layouts' worth of 8-row tile quarters: a workgroup stages
// 8 rows x 16 reduction channels x (125 + 27) taps as packed bf16 pairs in LDS ([tap][pair couple], taps down the
// lanes while staging) and writes 256 contiguous bytes per tap.
namespace {
constexpr int XF_ROWS = 8, XF_KC = 16, XF_COUPLES = XF_ROWS * XF_KC / 2;   // 64 couples of adjacent reduction channels
constexpr int XF_S5 = XF_COUPLES + 1;                                       // padded LDS row (u32): conflict-free transposed writes

struct XfLds {
  uint32_t s5[TAPS * XF_S5];
  uint32_t s3[27 * XF_S5];
};

template <bool WRITE_WD>
__device__ __forceinline__ void expert_frags_body(XfLds& L, const float* __restrict__ k5, const float* __restrict__ k3, int co_n,
                                                  int ci_n, int nrt, int nkc, bf16_t* __restrict__ wout, int kc, int rt, int q) {
  uint32_t* s5 = L.s5;
  uint32_t* s3 = L.s3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;       // q: which 8 rows of the 32-row tile
  // ---- stage: wave w takes couples w, w+4, ...; a couple = elements (row, red) and (row, red + 1)
  constexpr int NB = 4;                                             // couples in flight per wave
  for (int c0 = wave; c0 < XF_COUPLES; c0 += 4 * NB) {
    float a0[NB], a1[NB], b0[NB], b1[NB], c3a[NB], c3b[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int cp = c0 + 4 * u;
      const int rr = cp / (XF_KC / 2), kk = (cp % (XF_KC / 2)) * 2;
      const int row = rt * 32 + q * XF_ROWS + rr, red = kc * XF_KC + kk;
      const int coA = WRITE_WD ? red : row, ciA = WRITE_WD ? row : red;
      const int coB = WRITE_WD ? red + 1 : row, ciB = WRITE_WD ? row : red + 1;
      const bool liveA = coA < co_n && ciA < ci_n, liveB = coB < co_n && ciB < ci_n;
      const size_t oa = liveA ? (size_t)coA * ci_n + ciA : 0, ob = liveB ? (size_t)coB * ci_n + ciB : 0;
      a0[u] = liveA ? k5[oa * TAPS + lane] : 0.f;
      a1[u] = (liveA && lane + 64 < TAPS) ? k5[oa * TAPS + lane + 64] : 0.f;
      b0[u] = liveB ? k5[ob * TAPS + lane] : 0.f;
      b1[u] = (liveB && lane + 64 < TAPS) ? k5[ob * TAPS + lane + 64] : 0.f;
      c3a[u] = (liveA && lane < 27) ? k3[oa * 27 + lane] : 0.f;
      c3b[u] = (liveB && lane < 27) ? k3[ob * 27 + lane] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int cp = c0 + 4 * u;
      s5[lane * XF_S5 + cp] = pack_bf16x2(a0[u], b0[u]);
      if (lane + 64 < TAPS) s5[(lane + 64) * XF_S5 + cp] = pack_bf16x2(a1[u], b1[u]);
      if (lane < 27) s3[lane * XF_S5 + cp] = pack_bf16x2(c3a[u], c3b[u]);
    }
  }
  __syncthreads();
  // ---- write: 64 couples (256 bytes) per tap, the tile quarter's memory order; 4 taps per pass of the workgroup
  const size_t tile_elems = 32 * XF_KC;
  const size_t tap_stride = (size_t)nrt * nkc * tile_elems;
  uint32_t* out = reinterpret_cast<uint32_t*>(wout + ((size_t)rt * nkc + kc) * tile_elems + (size_t)q * XF_ROWS * XF_KC);
  const int cp = tid & 63, tq = tid >> 6;
  for (int tap = tq; tap < TAPS; tap += 4) {
    const int tap_out = WRITE_WD ? TAPS - 1 - tap : tap;
    out[(size_t)tap_out * tap_stride / 2 + cp] = s5[tap * XF_S5 + cp];
    const int dz = tap / 25, dy = (tap / 5) % 5, dx = tap % 5;
    if (dz >= 1 && dz <= 3 && dy >= 1 && dy <= 3) {                  // slot 1: the rows centre3 convolutions read
      const bool in = dx >= 1 && dx <= 3;
      const int t3 = ((dz - 1) * 3 + (dy - 1)) * 3 + (dx - 1);
      out[((size_t)TAPS + tap_out) * tap_stride / 2 + cp] = in ? s3[t3 * XF_S5 + cp] : 0u;
    }
  }
}

template <bool WRITE_WD>
__global__ __launch_bounds__(256) void expert_frags_kernel(const float* __restrict__ k5, const float* __restrict__ k3,
                                                           int co_n, int ci_n, int nrt, int nkc,
                                                           bf16_t* __restrict__ wout) {
  __shared__ XfLds L;
  expert_frags_body<WRITE_WD>(L, k5, k3, co_n, ci_n, nrt, nkc, wout, blockIdx.x, blockIdx.y, blockIdx.z);
}

// several blocks' experts, both roles, in one launch (the per-expert levels of a train step: see repmode_gatrep_fwd_multi)
constexpr int XM_MAX = REPMODE_GATREP_MULTI_MAX;
struct XfMultiArgs {
  const float* k5[XM_MAX]; const float* k3[XM_MAX];
  bf16_t* wf[XM_MAX]; bf16_t* wd[XM_MAX];
  int co[XM_MAX], ci[XM_MAX], nrt_f[XM_MAX], nkc_f[XM_MAX], nrt_d[XM_MAX], nkc_d[XM_MAX], nwf[XM_MAX];
  int first[XM_MAX + 1];
  int nblocks;
  int order;
};
__global__ __launch_bounds__(256) void expert_frags_multi_kernel(XfMultiArgs a) {
  __shared__ XfLds L;
  int i = 0;
  while (i + 1 < a.nblocks && (int)blockIdx.x >= a.first[i + 1]) ++i;
  int b = blockIdx.x - a.first[i];
  // (as in gatrep_fwd_multi_kernel: the four row quarters of one 1 KiB tile -- 256-byte pieces per tap -- on one XCD, dispatched
  // together: workgroups b, b + 8, b + 16, b + 24 of a block of 32)
  auto decode = [&](int bb, int nkc, int nrt, int& kc, int& rt, int& q) {
    const int ntile = nkc * nrt;
    if (a.order && (ntile & 7) == 0) {
      const int x = bb & 7, w = bb >> 3;
      q = w & 3;
      const int t = (w >> 2) * 8 + x;
      kc = t % nkc;
      rt = t / nkc;
    } else {
      kc = bb % nkc; bb /= nkc;
      rt = bb % nrt;
      q = bb / nrt;
    }
  };
  int kc, rt, q;
  if (b < a.nwf[i]) {
    decode(b, a.nkc_f[i], a.nrt_f[i], kc, rt, q);
    expert_frags_body<false>(L, a.k5[i], a.k3[i], a.co[i], a.ci[i], a.nrt_f[i], a.nkc_f[i], a.wf[i], kc, rt, q);
  } else {
    b -= a.nwf[i];
    decode(b, a.nkc_d[i], a.nrt_d[i], kc, rt, q);
    expert_frags_body<true>(L, a.k5[i], a.k3[i], a.co[i], a.ci[i], a.nrt_d[i], a.nkc_d[i], a.wd[i], kc, rt, q);
  }
}
}  // namespace

// Do the stored operands still belong to the parameters?  XC_WGS workgroups per block each sample the same XC_SAMPLES
// elements of each expert tensor, round them as the layout does and compare with the forward-role operand -- so all of a
// block's workgroups reach the same verdict without talking to each other -- and, on any difference, lay the block out again
// between them (every XC_WGS-th tile quarter each, both roles): check and repair are ONE launch whose common case is a few
// thousand scattered loads.  What keeps operands across steps (the operator library's store) learns about parameter writes
// from autograd's version counters and an optimizer hook -- `p.data.copy_()`, a collective's broadcast into the parameters
// or a foreign kernel move neither; this looks at the BYTES, on the device, with no host synchronisation.
namespace {
constexpr int XC_SAMPLES = 1024;      // per tensor and block: a dense write is caught with certainty, a sparse one by chance
constexpr int XC_WGS = 64;            // workgroups per block (the repair's parallelism; the check is redundant across them)
struct XfCheckArgs {
  const float* k5[XM_MAX]; const float* k3[XM_MAX];
  bf16_t* wf[XM_MAX]; bf16_t* wd[XM_MAX];
  int co[XM_MAX], ci[XM_MAX], nrt_f[XM_MAX], nkc_f[XM_MAX], nrt_d[XM_MAX], nkc_d[XM_MAX];
  int* flags;
  int* sticky;        // per block: the epoch of the last launch in which some workgroup found a difference (stream scratch)
  int epoch;          // this launch's number (never 0)
};
__global__ __launch_bounds__(256) void expert_frags_verify_kernel(XfCheckArgs a) {
  __shared__ XfLds L;
  const int i = blockIdx.x / XC_WGS, j = blockIdx.x % XC_WGS, tid = threadIdx.x;
  const int co = a.co[i], ci = a.ci[i], nrt = a.nrt_f[i], nkc = a.nkc_f[i];
  const size_t tap_stride = (size_t)nrt * nkc * 512;
  const unsigned long long n5 = (unsigned long long)co * ci * TAPS, n3 = (unsigned long long)co * ci * 27;
  int diff = 0;
  for (int s = tid; s < XC_SAMPLES; s += 256) {
    const unsigned long long h = (unsigned long long)(s + 1) * 0x9E3779B97F4A7C15ull;
    {
      const unsigned long long idx = (h >> 11) % n5;
      const int tap = (int)(idx % TAPS);
      const unsigned long long m = idx / TAPS;
      const int c_i = (int)(m % ci), c_o = (int)(m / ci);
      const size_t fi = (size_t)tap * tap_stride + ((size_t)(c_o / 32) * nkc + c_i / 16) * 512 + (c_o % 32) * 16 + c_i % 16;
      diff |= (a.wf[i][fi] != (bf16_t)(pack_bf16x2(a.k5[i][idx], 0.f) & 0xffffu));
    }
    {
      const unsigned long long idx = (h >> 7) % n3;
      const int t3 = (int)(idx % 27);
      const unsigned long long m = idx / 27;
      const int c_i = (int)(m % ci), c_o = (int)(m / ci);
      const int tap = ((t3 / 9 + 1) * 5 + (t3 / 3) % 3 + 1) * 5 + t3 % 3 + 1;
      const size_t fi = ((size_t)TAPS + tap) * tap_stride + ((size_t)(c_o / 32) * nkc + c_i / 16) * 512 + (c_o % 32) * 16 + c_i % 16;
      diff |= (a.wf[i][fi] != (bf16_t)(pack_bf16x2(a.k3[i][idx], 0.f) & 0xffffu));
    }
  }
  // Check and repair share a launch with no barrier between workgroups: a workgroup that starts late would
  // sample operands the early ones have already laid out again, find nothing wrong and skip ITS share of the repair.  So a
  // workgroup that finds a difference says so BEFORE it repairs -- this launch's epoch into the block's sticky word -- and every
  // workgroup also takes that word as a verdict: whoever can see repaired bytes can see the word that was stored before them.
  const int seen = __hip_atomic_load(a.sticky + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == a.epoch;
  const int bad = __syncthreads_or(diff | seen);
  if (j == 0 && tid == 0 && a.flags) a.flags[i] = bad ? 1 : 0;
  if (!bad) return;
  if (tid == 0 && !seen) __hip_atomic_store(a.sticky + i, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();            // (the word is on its way before any of this workgroup's repair stores)
  // ---- repair (rare): this workgroup's share of the tile quarters of both roles
  const int ntf = nkc * nrt * 4;
  for (int t = j; t < ntf; t += XC_WGS) {
    expert_frags_body<false>(L, a.k5[i], a.k3[i], co, ci, nrt, nkc, a.wf[i], t % nkc, (t / nkc) % nrt, t / (nkc * nrt));
    __syncthreads();
  }
  if (a.wd[i]) {
    const int nrd = a.nrt_d[i], nkd = a.nkc_d[i], ntd = nkd * nrd * 4;
    for (int t = j; t < ntd; t += XC_WGS) {
      expert_frags_body<true>(L, a.k5[i], a.k3[i], co, ci, nrd, nkd, a.wd[i], t % nkd, (t / nkd) % nrd, t / (nkd * nrd));
      __syncthreads();
    }
  }
}
}  // namespace

static int expert_frags_multi_impl(int nblocks, const float* const* k5, const float* const* k3, const int* co, const int* ci,
                                   void* const* wf, void* const* wd, bool verify, int* flags, void* stream) {
  RM_REQUIRE(k5 && k3 && co && ci && wf && wd, "expert_frags_multi: null pointer");
  RM_REQUIRE(nblocks > 0 && nblocks <= XM_MAX, "expert_frags_multi: 1..%d blocks per call, got %d", XM_MAX, nblocks);
  XfMultiArgs a{};
  a.nblocks = nblocks;
  static const int order = []() { const char* e = getenv("REPMODE_GATREP_ORDER"); return e ? atoi(e) : 1; }();
  a.order = order;
  long total = 0;
  double bytes = 0;
  XfCheckArgs c{};
  c.flags = flags;
  for (int i = 0; i < nblocks; ++i) {
    RM_REQUIRE(k5[i] && k3[i] && (wf[i] || wd[i]) && co[i] > 0 && ci[i] > 0, "expert_frags_multi: bad block %d", i);
    RM_REQUIRE(!verify || wf[i], "expert_frags_refresh_multi: block %d has no forward-role operand to check", i);
    a.k5[i] = k5[i]; a.k3[i] = k3[i]; a.wf[i] = static_cast<bf16_t*>(wf[i]); a.wd[i] = static_cast<bf16_t*>(wd[i]);
    a.co[i] = co[i]; a.ci[i] = ci[i];
    a.nrt_f[i] = repmode_padded_channels(co[i], REPMODE_BF16, 0) / 32; a.nkc_f[i] = repmode_padded_channels(ci[i], REPMODE_BF16, 1) / 16;
    a.nrt_d[i] = repmode_padded_channels(ci[i], REPMODE_BF16, 0) / 32; a.nkc_d[i] = repmode_padded_channels(co[i], REPMODE_BF16, 1) / 16;
    a.nwf[i] = wf[i] ? a.nkc_f[i] * a.nrt_f[i] * 4 : 0;
    a.first[i] = (int)total;
    total += a.nwf[i] + (wd[i] ? (long)a.nkc_d[i] * a.nrt_d[i] * 4 : 0);
    bytes += (double)co[i] * ci[i] * (152.0 * 4 + (125.0 + 45.0) * 2 * ((wf[i] ? 1 : 0) + (wd[i] ? 1 : 0)));
    c.k5[i] = k5[i]; c.k3[i] = k3[i]; c.wf[i] = a.wf[i]; c.wd[i] = a.wd[i]; c.co[i] = co[i]; c.ci[i] = ci[i];
    c.nrt_f[i] = a.nrt_f[i]; c.nkc_f[i] = a.nkc_f[i]; c.nrt_d[i] = a.nrt_d[i]; c.nkc_d[i] = a.nkc_d[i];
  }
  a.first[nblocks] = (int)total;
  RM_REQUIRE(total < (1L << 31), "expert_frags_multi: grid too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (verify) {
    float* scratch = repmode_zero_scratch(s);
    if (!scratch) return REPMODE_ELAUNCH;
    static std::atomic<int> epoch{0};
    int e = ++epoch;
    if (e == 0) e = ++epoch;                      // (0 is what the scratch starts as)
    c.sticky = reinterpret_cast<int*>(scratch + REPMODE_SCRATCH_BARRIER_OFF) + REPMODE_SCRATCH_VERIFY_WORD;
    c.epoch = e;
    repmode_prof_begin(REPMODE_PROF_GATREP_FWD, (double)nblocks * XC_WGS * XC_SAMPLES * 2 * 6.0, s);
    hipLaunchKernelGGL(expert_frags_verify_kernel, dim3((unsigned)(nblocks * XC_WGS)), dim3(256), 0, s, c);
    RM_LAUNCH_CHECK("expert_frags_verify");
    repmode_prof_end(s);
    return REPMODE_OK;
  }
  repmode_prof_begin(REPMODE_PROF_GATREP_FWD, bytes, s);
  hipLaunchKernelGGL(expert_frags_multi_kernel, dim3((unsigned)total), dim3(256), 0, s, a);
  RM_LAUNCH_CHECK("expert_frags_multi");
  repmode_prof_end(s);
  return REPMODE_OK;
}

extern "C" int repmode_expert_frags_multi(int nblocks, const float* const* k5, const float* const* k3, const int* co, const int* ci,
                                          void* const* wf, void* const* wd, void* stream) {
  return expert_frags_multi_impl(nblocks, k5, k3, co, ci, wf, wd, false, nullptr, stream);
}

// Check-and-repair of operands kept across steps in ONE launch (expert_frags_verify_kernel): whether block i's stored
// forward-role operand still equals its parameters rounded to bf16 at XC_SAMPLES sampled positions per expert tensor; where it
// does not, the block is laid out again (both roles).  flags (device int[nblocks], may be NULL): the verdicts, for whoever wants
// to look.  No host synchronisation.
extern "C" int repmode_expert_frags_refresh_multi(int nblocks, const float* const* k5, const float* const* k3, const int* co,
                                                  const int* ci, void* const* wf, void* const* wd, int* flags, void* stream) {
  return expert_frags_multi_impl(nblocks, k5, k3, co, ci, wf, wd, true, flags, stream);
}

// wf: [2][125][CoP/32][CiP/16][32][16] bf16 (rows = co), wd: [2][125][CiP/32][CoP/16][32][16] (rows = ci, taps
// flipped); either may be NULL.  Slot 1 is valid on the centre3 rows only (see above).
extern "C" int repmode_expert_frags(const float* k5, const float* k3, int co, int ci, void* wf, void* wd, void* stream) {
  RM_REQUIRE(k5 && k3 && (wf || wd), "expert_frags: null pointer");
  RM_REQUIRE(co > 0 && ci > 0, "expert_frags: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double bytes = (double)co * ci * (152.0 * 4 + (125.0 + 45.0) * 2 * ((wf ? 1 : 0) + (wd ? 1 : 0)));
  repmode_prof_begin(REPMODE_PROF_GATREP_FWD, bytes, s);
  if (wf) {
    const int nrt = repmode_padded_channels(co, REPMODE_BF16, 0) / 32, nkc = repmode_padded_channels(ci, REPMODE_BF16, 1) / 16;
    hipLaunchKernelGGL(expert_frags_kernel<false>, dim3(nkc, nrt, 4), dim3(256), 0, s, k5, k3, co, ci, nrt, nkc,
                       static_cast<bf16_t*>(wf));
    RM_LAUNCH_CHECK("expert_frags(wf)");
  }
  if (wd) {
    const int nrt = repmode_padded_channels(ci, REPMODE_BF16, 0) / 32, nkc = repmode_padded_channels(co, REPMODE_BF16, 1) / 16;
    hipLaunchKernelGGL(expert_frags_kernel<true>, dim3(nkc, nrt, 4), dim3(256), 0, s, k5, k3, co, ci, nrt, nkc,
                       static_cast<bf16_t*>(wd));
    RM_LAUNCH_CHECK("expert_frags(wd)");
  }
  repmode_prof_end(s);
  return REPMODE_OK;
}

// Softmax-Jacobian + gate Linear gradients alone, from gate-probability gradients dg[s][5][Co] (used by the
// per-expert formulation, where "slots" are the samples themselves and dg[n][e][o] = <dy[n], P_e[n]>).
// One launch of the gate backward, or (defer) one queued job for the next conv5 launch on the stream
static int gate_bwd_launch(const float* g, float* dg, const int32_t* slot_task, int nslots, int num_tasks, int co, float* dgate_w,
                           float* dgate_b, int clear, bool defer, hipStream_t s) {
  if (defer) {
    TailJob j{};
    j.kind = 1;
    j.nblocks = ceil_div(co, 32);
    j.in0 = g; j.io1 = dg; j.ids = slot_task; j.out2 = dgate_w; j.out3 = dgate_b;
    j.p0 = nslots; j.p1 = num_tasks; j.p2 = co; j.p3 = clear;
    return repmode_tail_push(j, s);
  }
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(ceil_div(co, 32)), dim3(GATE_BWD_THREADS), 0, s, g, dg, slot_task, nslots, num_tasks, co,
                     dgate_w, dgate_b, clear);
  RM_LAUNCH_CHECK("gate_bwd");
  return REPMODE_OK;
}

extern "C" int repmode_gate_bwd_ex(const float* g, const float* dg, const int32_t* slot_task, int nslots, int num_tasks,
                                   int co, float* dgate_w, float* dgate_b, int flags, void* stream) {
  RM_REQUIRE(g && dg && slot_task && dgate_w && dgate_b, "gate_bwd: null pointer");
  RM_REQUIRE(nslots > 0 && num_tasks > 0 && co > 0, "gate_bwd: bad shape");
  return gate_bwd_launch(g, const_cast<float*>(dg), slot_task, nslots, num_tasks, co, dgate_w, dgate_b, 0, (flags & REPMODE_DEFER) != 0,
                         static_cast<hipStream_t>(stream));
}

extern "C" int repmode_gate_bwd(const float* g, const float* dg, const int32_t* slot_task, int nslots, int num_tasks,
                                int co, float* dgate_w, float* dgate_b, void* stream) {
  return repmode_gate_bwd_ex(g, dg, slot_task, nslots, num_tasks, co, dgate_w, dgate_b, 0, stream);
}

extern "C" int repmode_gatrep_bwd_ex(const float* dw, const float* k5, const float* k3, const float* k1,
                                     const float* a3, const float* a5, const float* g, const int32_t* slot_task,
                                     int nslots, int num_tasks, int co, int ci, float* dk5, float* dk3, float* dk1,
                                     float* da3, float* da5, float* dgate_w, float* dgate_b, float* dg_ws, int flags,
                                     void* stream) {
  RM_REQUIRE(dw && k5 && k3 && k1 && a3 && a5 && g && slot_task, "gatrep_bwd: null input");
  RM_REQUIRE(dk5 && dk3 && dk1 && da3 && da5 && dgate_w && dgate_b && dg_ws, "gatrep_bwd: null output");
  RM_REQUIRE(nslots > 0 && num_tasks > 0 && co > 0 && ci > 0, "gatrep_bwd: bad shape");
  RM_REQUIRE((long)TAPS * co * ci < (1L << 27), "gatrep_bwd: 8 slots of the filter gradient must stay below 4 GiB (32-bit offsets)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  // dg accumulates in the library's zero scratch when it fits (gate_bwd puts the zeros back: no memset launch);
  // otherwise in the caller's dg_ws
  const size_t ndg = (size_t)nslots * E * co;
  float* dg = dg_ws;
  int clear = 0;
  if (ndg <= REPMODE_ZERO_SCRATCH_FLOATS - REPMODE_SCRATCH_GATE_OFF) {
    dg = repmode_zero_scratch(s);
    if (!dg) return REPMODE_ELAUNCH;
    dg += REPMODE_SCRATCH_GATE_OFF;
    clear = 1;
  } else {
    RM_HIP(hipMemsetAsync(dg_ws, 0, ndg * sizeof(float), s));
  }
  repmode_prof_begin(REPMODE_PROF_GATREP_BWD, (double)co * ci * 4.0 * (125.0 * nslots + 2 * 155.0), s);
  // Small layers (at most one workgroup per CU) are one dependent chain per launch: 512 threads per tile, 8 slots' loads in
  // flight per thread.  Larger ones want the occupancy of the 256-thread, 2-slot shape.  Measured per launch, warm / cold
  // (tools/gatrep_microbench.py under rocprofv3, 8 slots): 32x32 13.3 / 15.4 us, 64x64 13.9 / 19.4, 128x64 15.6 / 25.5 (512
  // threads; 1024 x 4 slots: 16.7-22.0 / 20.0-30.3; 256 x 8: 15.8-18.3 / 18.2-28.3); 128x128 26.6 / 42.7, 128x256 49.9 / 79.6
  // (256 x 2; 1024 x 4: 41.2-78.3 / 57.1-104.4; 512 x 4: 32.8-62.9 / 47.2-86.7).
  const int gx = repmode_deterministic() ? 1 : ceil_div(ci, GF_CT);      // (deterministic: a workgroup walks all tiles of its co)
  if ((long)ceil_div(ci, GF_CT) * co <= 256)
    hipLaunchKernelGGL((gatrep_bwd_kernel<512, 8>), dim3(gx, co), dim3(512), 0, s, dw, k5, k3, k1, a3, a5, g, nslots, co, ci, dk5,
                       dk3, dk1, da3, da5, dg);
  else
    hipLaunchKernelGGL((gatrep_bwd_kernel<256, 2>), dim3(gx, co), dim3(256), 0, s, dw, k5, k3, k1, a3, a5, g, nslots, co, ci, dk5,
                       dk3, dk1, da3, da5, dg);
  RM_LAUNCH_CHECK("gatrep_bwd");
  // the gate backward: its own launch, or (REPMODE_DEFER) the first workgroups of the next conv5 launch on this stream
  const int rc = gate_bwd_launch(g, dg, slot_task, nslots, num_tasks, co, dgate_w, dgate_b, clear, (flags & REPMODE_DEFER) != 0, s);
  repmode_prof_end(s);
  return rc;
}

extern "C" int repmode_gatrep_bwd(const float* dw, const float* k5, const float* k3, const float* k1,
                                  const float* a3, const float* a5, const float* g, const int32_t* slot_task,
                                  int nslots, int num_tasks, int co, int ci, float* dk5, float* dk3, float* dk1,
                                  float* da3, float* da5, float* dgate_w, float* dgate_b, float* dg_ws,
                                  void* stream) {
  return repmode_gatrep_bwd_ex(dw, k5, k3, k1, a3, a5, g, slot_task, nslots, num_tasks, co, ci, dk5, dk3, dk1, da3, da5, dgate_w,
                               dgate_b, dg_ws, 0, stream);
}
