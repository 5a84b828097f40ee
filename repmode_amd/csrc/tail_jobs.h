// tail_jobs.h -- small independent jobs that ride in the grid of a conv5 launch instead of being launched on their own.
//
// A kernel boundary between dependent launches costs 4-5 us of GPU time, and a short kernel behind it fetches everything
// from HBM / Infinity Cache.  On the backward pass ~25 launches per step are a gate backward (softmax Jacobian + Linear
// gradients, RepMode.py:198-200 via autograd) or a gradient-layout transpose: a few microseconds of work whose inputs are
// complete and whose outputs nothing reads before the layer's backward returns -- while the layer's data-gradient
// convolution, which depends on neither, is still to be launched.  Such a job can be DEFERRED (`*_ex` entry points,
// flag REPMODE_DEFER): it is queued per stream and becomes the first workgroups of the next repmode_conv5* launch on that
// stream (they start at once, the conv's workgroups fill the chip behind them); repmode_tail_flush launches whatever is
// still queued on its own.  Same arithmetic, same kernels' bodies (below), one boundary less per job.
#pragma once
#include "common.h"

constexpr int TAIL_MAX_JOBS = 3;
constexpr int TAIL_THREADS = 256;
constexpr int TAIL_LDS_BYTES = 125 * 65 * 4;     // the transpose's tile; a host kernel provides at least this much LDS

struct TailJob {
  int kind;            // 1: gate backward, 2: tap transpose
  int nblocks;         // workgroups of TAIL_THREADS threads
  const float* in0;    // gate: g[S][5][Co]              transpose: in[taps][M]
  float* io1;          // gate: dg[S][5][Co] (read, cleared when p3)   transpose: out[M][taps_out]
  const int32_t* ids;  // gate: slot_task[S]
  float* out2;         // gate: dgate_w[5 Co][T]
  float* out3;         // gate: dgate_b[5 Co]
  long m;              // transpose: M = Co * Ci
  int p0, p1, p2, p3;  // gate: S, T, Co, clear            transpose: taps_out (125 / 27 / 8)
};

struct TailJobs {
  TailJob job[TAIL_MAX_JOBS];
  int njobs;
  int nblocks;         // all jobs' workgroups, rounded up to a multiple of 8 (keeps the host kernel's workgroup -> XCD map)
};

// host side (api.hip): the per-stream queue
int repmode_tail_push(const TailJob& job, hipStream_t s);        // REPMODE_OK, or an error code (queue full: flushes first)
void repmode_tail_take(hipStream_t s, TailJobs* out);            // moves the stream's queue into *out (njobs == 0: nothing queued)
int repmode_tail_launch(const TailJobs& t, hipStream_t s);       // the jobs as a kernel of their own

#ifdef __HIPCC__
constexpr int TAIL_E = 5;

// softmax Jacobian + gate Linear gradients.  One thread per (o, e), the five experts of a channel adjacent (160 threads =
// 32 channels per workgroup); loops over slots.  clear: dg lives in the library's zero scratch -- once every thread of
// the workgroup has read its channels' entries, they are zeroed again.
__device__ __forceinline__ void tail_gate_bwd(const float* __restrict__ g, float* __restrict__ dg, const int32_t* __restrict__ slot_task,
                                              int nslots, int num_tasks, int co_n, float* __restrict__ dgate_w,
                                              float* __restrict__ dgate_b, int clear, int block, int tid) {
  const int idx = block * (32 * TAIL_E) + tid;
  const int o = idx / TAIL_E, e = idx % TAIL_E;
  const bool on = tid < 32 * TAIL_E && o < co_n;
  if (on) {
    float* wrow = dgate_w + ((size_t)e * co_n + o) * num_tasks;
    for (int t = 0; t < num_tasks; ++t) wrow[t] = 0.f;
    float bsum = 0.f;
    for (int s = 0; s < nslots; ++s) {
      const float* gs = g + (size_t)s * TAIL_E * co_n + o;
      const float* ds = dg + (size_t)s * TAIL_E * co_n + o;
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < TAIL_E; ++k) dot += gs[k * co_n] * ds[k * co_n];
      const float dl = gs[e * co_n] * (ds[e * co_n] - dot);
      wrow[slot_task[s]] += dl;        // (this thread owns the row; two slots never share a task)
      bsum += dl;
    }
    dgate_b[e * co_n + o] = bsum;
  }
  if (clear) {
    __syncthreads();
    if (on)
      for (int s = 0; s < nslots; ++s) dg[((size_t)s * TAIL_E + e) * co_n + o] = 0.f;
  }
}

// [taps][M] -> [M][taps_out] (taps_out 27: the centred 3x3x3 sub-cube of 125 taps): a 64-column strip through LDS, so that
// both the reads (64 floats of a tap row) and the writes (64 * taps_out contiguous floats) are whole lines.
__device__ __forceinline__ void tail_tap_transpose(const float* __restrict__ in, float* __restrict__ out, long M, int ntaps_out,
                                                   float* tile, int block, int tid) {
  const long m0 = (long)block * 64;
  const int ncol = (int)min((long)64, M - m0);
  for (int i = tid; i < ntaps_out * 64; i += TAIL_THREADS) {
    const int t = i / 64, col = i % 64;
    int tap = t;
    if (ntaps_out == 27) tap = ((t / 9 + 1) * 5 + (t / 3) % 3 + 1) * 5 + t % 3 + 1;
    if (col < ncol) tile[t * 65 + col] = in[(size_t)tap * M + m0 + col];
  }
  __syncthreads();
  for (int i = tid; i < ncol * ntaps_out; i += TAIL_THREADS) {
    const int col = i / ntaps_out, t = i % ntaps_out;
    out[(size_t)m0 * ntaps_out + i] = tile[t * 65 + col];
  }
}

// workgroup `block` (< t.nblocks) of the queued jobs; lds: at least TAIL_LDS_BYTES
__device__ __forceinline__ void tail_run(const TailJobs& t, int block, int tid, float* lds) {
#pragma unroll 1
  for (int j = 0; j < t.njobs; ++j) {
    const TailJob& q = t.job[j];
    if (block < q.nblocks) {
      if (q.kind == 1) tail_gate_bwd(q.in0, q.io1, q.ids, q.p0, q.p1, q.p2, q.out2, q.out3, q.p3, block, tid);
      else tail_tap_transpose(q.in0, q.io1, q.m, q.p0, lds, block, tid);
      return;
    }
    block -= q.nblocks;
  }
}
#endif
