// adam.hip -- the optimizer pass of the train step as this build's own kernels (round 4).
//
// The reference's harness steps `torch.optim.Adam(net.parameters(), lr)` (fnet/fnet_model.py:55, :112): betas (0.9, 0.999),
// eps 1e-8, no weight decay, no amsgrad.  Per element, in float32, with the step's bias corrections computed on the host in
// double precision exactly as torch's single-tensor path does (torch/optim/adam.py `_single_tensor_adam`):
//     m <- m + (1 - beta1) (g - m)                       exp_avg.lerp_(grad, 1 - beta1)
//     v <- v beta2 + (1 - beta2) g g                     exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
//     p <- p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
//
// Two kernels:
//   adam_multi_kernel        any list of float tensors (<= ADAM_MAX per launch, the pointers travel in the launch arguments):
//                            a workgroup owns a 4096-element chunk of one tensor -- one read of p, g, m, v, one write of p, m, v.
//   adam_frags_kernel        the 5x5x5 and 3x3x3 experts of the blocks that run the per-expert formulation (the deep levels:
//                            84 % of the parameters).  The same update, and the UPDATED values leave the kernel a second time
//                            as the convolution kernels' fragment-major bf16 operands (both roles: rows = co for the forward
//                            conv, rows = ci with flipped taps for the data gradient) -- what repmode_expert_frags_multi
//                            re-derives from the parameters at the start of every forward pass (round 3: 238 us per step, a
//                            second read of 420 MB that this pass has in registers anyway).  A workgroup owns a 16 x 16
//                            (co, ci) tile with all 125 + 27 taps: the parameter layout [co][ci][taps] makes its 16 rows 8 KB
//                            contiguous runs; the updated values are staged as bf16 in LDS (parameter order, 78 KB; transposed on the way out) and written
//                            out as 512-byte pieces of the 1 KiB fragment tiles.
#include "common.h"

#include <cmath>

namespace {

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int rup(int a, int b) { return cdiv(a, b) * b; }

constexpr int ADAM_MAX = REPMODE_ADAM_MULTI_MAX;                   // tensors per adam_multi launch (launch arguments <= 4 KB)
constexpr int ADAM_CHUNK = 4096;               // elements per workgroup: 256 threads x 4 x float4

struct AdamHyper {
  float w1;          // 1 - beta1
  float beta2;
  float w2;          // 1 - beta2
  float bc2_sqrt;    // sqrt(1 - beta2^t)
  float eps;
  float neg_step;    // -(lr / (1 - beta1^t))
};

__device__ __forceinline__ void adam_elem(const AdamHyper& h, float& p, float g, float& m, float& v) {
  m = m + h.w1 * (g - m);
  v = v * h.beta2 + h.w2 * g * g;
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p = p + h.neg_step * (m / denom);
}

struct AdamMultiArgs {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  long numel[ADAM_MAX];
  int first[ADAM_MAX + 1];       // first workgroup of tensor i
  int nt;
  AdamHyper h;
  const AdamHyper* hd;           // non-null: the step's constants come from device memory (repmode_adam_hyper_dev: a captured step)
};

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMultiArgs a) {
  const AdamHyper h = a.hd ? *a.hd : a.h;
  int i = 0;
  while (i + 1 < a.nt && (int)blockIdx.x >= a.first[i + 1]) ++i;
  const long base = (long)((int)blockIdx.x - a.first[i]) * ADAM_CHUNK;
  const long n = a.numel[i];
  float* __restrict__ p = a.p[i];
  const float* __restrict__ g = a.g[i];
  float* __restrict__ m = a.m[i];
  float* __restrict__ v = a.v[i];
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec && base + ADAM_CHUNK <= n) {
    f32x4 P[4], G[4], M[4], V[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long o = base + (long)(u * 256 + threadIdx.x) * 4;
      P[u] = *reinterpret_cast<const f32x4*>(p + o);
      G[u] = *reinterpret_cast<const f32x4*>(g + o);
      M[u] = *reinterpret_cast<const f32x4*>(m + o);
      V[u] = *reinterpret_cast<const f32x4*>(v + o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long o = base + (long)(u * 256 + threadIdx.x) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pp = P[u][k], mm = M[u][k], vv = V[u][k];
        adam_elem(h, pp, G[u][k], mm, vv);
        P[u][k] = pp; M[u][k] = mm; V[u][k] = vv;
      }
      *reinterpret_cast<f32x4*>(p + o) = P[u];
      *reinterpret_cast<f32x4*>(m + o) = M[u];
      *reinterpret_cast<f32x4*>(v + o) = V[u];
    }
    return;
  }
  const long end = base + ADAM_CHUNK < n ? base + ADAM_CHUNK : n;
  for (long o = base + threadIdx.x; o < end; o += 256) {
    float pp = p[o], mm = m[o], vv = v[o];
    adam_elem(h, pp, g[o], mm, vv);
    p[o] = pp; m[o] = mm; v[o] = vv;
  }
}

// ---- the per-expert blocks' 5x5x5 / 3x3x3 experts: update + fragment-major bf16 operands
#ifndef RM_ADAM_TIMING
#define RM_ADAM_TIMING 0
#endif
#ifndef RM_ADAM_RPI
#define RM_ADAM_RPI 2
#endif
constexpr int AF_MAX = REPMODE_GATREP_MULTI_MAX;
constexpr int AF_T = 16;                                 // the workgroup's tile: AF_T output x AF_T input channels
constexpr int AF_TAPS5 = REPMODE_TAPS, AF_TAPS3 = 27;

struct AdamFragArgs {
  float* p5[AF_MAX]; const float* g5[AF_MAX]; float* m5[AF_MAX]; float* v5[AF_MAX];
  float* p3[AF_MAX]; const float* g3[AF_MAX]; float* m3[AF_MAX]; float* v3[AF_MAX];
  bf16_t* wf[AF_MAX]; bf16_t* wd[AF_MAX];
  int co[AF_MAX], ci[AF_MAX];
  int first[AF_MAX + 1];
  int nblocks;
  AdamHyper h;
  const AdamHyper* hd;           // non-null: the step's constants come from device memory (a captured step)
};

// One parameter tensor's part of the tile: rows co0 .. co0+15, input channels ci0 .. ci0+15, TAPS taps each.  A row is
// ncols * TAPS contiguous floats (16-byte aligned when ci is a multiple of 4: taps * 16 * 4 bytes per ci tile, and every row
// starts a multiple of ci * taps floats in).  Updated in place; the new value goes to LDS as bf16 IN THE PARAMETER'S OWN ORDER,
// lds[row][col * TAPS + tap] (row stride AfRow<TAPS>::RS): a thread's four consecutive floats are one 8-byte LDS write and a
// wave's writes are consecutive -- the [tap][row][col] staging this replaces put a wave's 2-byte writes 4 taps = 2 KB apart,
// all 64 lanes in one bank.  The transposition happens on the READ side (below), where the odd tap count spreads the
// lanes' (row, col) strides over all banks.
template <int TAPS>
struct AfRow {
  static constexpr int RS = AF_T * TAPS + 4;       // bf16 elements per staged row (+4: the co-pair reads of the data-gradient
};                                                 // role, 2 RS apart, then fall into 8 different banks instead of 4)

template <int TAPS, int NT>
__device__ __forceinline__ void adam_tile_rows(const AdamHyper& h, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int co_n, int ci_n, int co0, int ci0, bf16_t* lds) {
  constexpr int RS = AfRow<TAPS>::RS;
  const int nrows = max(0, min(AF_T, co_n - co0)), ncols = max(0, min(AF_T, ci_n - ci0));
  const int run = ncols * TAPS;                                  // floats of a row inside the tile
  const bool vec = (((long)ci_n * TAPS) & 3) == 0 && ((ci0 * TAPS) & 3) == 0 && (run & 3) == 0 &&
                   (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  // zero what the tile does not cover (ragged channel counts): the fragment tiles' padding must be zero
  if (nrows < AF_T || ncols < AF_T) {
    for (int i = threadIdx.x; i < AF_T * AF_T * TAPS; i += NT) {
      const int r = i / (AF_T * TAPS), e = i - r * (AF_T * TAPS);
      if (r >= nrows || e >= run) lds[r * RS + e] = 0;
    }
  }
  if (vec) {
    // items = (row, float4 of the row's run), all rows of the tile in ONE index space (the 27-tap expert's rows are 108
    // float4s: a loop per row kept 4/5 of the workgroup idle); RPI items in flight per thread and iteration
    const int nvec = run / 4;                                    // float4 items per row (full tile: TAPS = 125: 500, 27: 108)
    const int nitems = nrows * nvec;
    constexpr int RPI = RM_ADAM_RPI;
    for (int i0 = 0; i0 < nitems; i0 += RPI * NT) {
      f32x4 P[RPI], G[RPI], M[RPI], V[RPI];
      int row[RPI], it[RPI];
#pragma unroll
      for (int u = 0; u < RPI; ++u) {
        const int item = i0 + u * NT + threadIdx.x;
        row[u] = nvec == 4 * TAPS ? item / (4 * TAPS) : item / nvec;            // (full tiles: a constant divisor)
        it[u] = item - row[u] * nvec;
        if (item < nitems) {
          const long o = ((long)(co0 + row[u]) * ci_n + ci0) * TAPS + (long)it[u] * 4;
          P[u] = *reinterpret_cast<const f32x4*>(p + o);
          G[u] = *reinterpret_cast<const f32x4*>(g + o);
          M[u] = *reinterpret_cast<const f32x4*>(m + o);
          V[u] = *reinterpret_cast<const f32x4*>(v + o);
        }
      }
#pragma unroll
      for (int u = 0; u < RPI; ++u) {
        const int item = i0 + u * NT + threadIdx.x;
        if (item < nitems) {
          const long o = ((long)(co0 + row[u]) * ci_n + ci0) * TAPS + (long)it[u] * 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float pp = P[u][k], mm = M[u][k], vv = V[u][k];
            adam_elem(h, pp, G[u][k], mm, vv);
            P[u][k] = pp; M[u][k] = mm; V[u][k] = vv;
          }
          *reinterpret_cast<f32x4*>(p + o) = P[u];
          *reinterpret_cast<f32x4*>(m + o) = M[u];
          *reinterpret_cast<f32x4*>(v + o) = V[u];
#if RM_ADAM_TIMING < 2
          *reinterpret_cast<u32x2*>(lds + row[u] * RS + it[u] * 4) =
              u32x2{pack_bf16x2(P[u][0], P[u][1]), pack_bf16x2(P[u][2], P[u][3])};
#endif
        }
      }
    }
  } else {
    for (int r = 0; r < nrows; ++r) {
      for (int e = threadIdx.x; e < run; e += NT) {
        const long o = ((long)(co0 + r) * ci_n + ci0) * TAPS + e;
        float pp = p[o], mm = m[o], vv = v[o];
        adam_elem(h, pp, g[o], mm, vv);
        p[o] = pp; m[o] = mm; v[o] = vv;
        lds[r * RS + e] = f32_to_bf16(pp);
      }
    }
  }
}

// element (tap, row, col) and its right / lower neighbour of a staged tensor, packed as the operands' dword
template <int TAPS>
__device__ __forceinline__ uint32_t af_pair_cols(const bf16_t* lds, int tap, int row, int col) {     // (row, col), (row, col + 1)
  const bf16_t* q = lds + row * AfRow<TAPS>::RS + col * TAPS + tap;
  return (uint32_t)q[0] | ((uint32_t)q[TAPS] << 16);
}
template <int TAPS>
__device__ __forceinline__ uint32_t af_pair_rows(const bf16_t* lds, int tap, int row, int col) {     // (row, col), (row + 1, col)
  const bf16_t* q = lds + row * AfRow<TAPS>::RS + col * TAPS + tap;
  return (uint32_t)q[0] | ((uint32_t)q[AfRow<TAPS>::RS] << 16);
}

// NT threads per workgroup: two workgroups share a CU's LDS (78 KB each) and a workgroup alternates "load + update + stage" /
// barrier / "write the operands", so the thread count sets how much of one phase hides under the other (same box,
// tools/adam_microbench.py, the six per-expert blocks of the network, 104.6 M elements: 256 threads 701 us = 4.85 TB/s,
// 512 threads 660 us = 5.15 TB/s, 1024 threads (one workgroup per CU by registers) 745 us; rows in flight per iteration
// 1 / 2 / 4: no difference.  The plain kernel, short workgroups at full occupancy, streams at 6.5 TB/s.)
// Where the time goes (timing-only builds -DRM_ADAM_TIMING=1/2/3, one box, tools/sessions/r4_session14.sh): the whole pass
// 620-640 us; without the operand write 551 (its 0.42 GB at the pass's own rate); without the staging either 552-561; without
// the LDS reservation (8 workgroups per CU) 573: the update of 16 x 8 KB runs per tensor streams at 5.2 TB/s whatever the
// occupancy -- the tile shape the operands' layout asks for, not the staging, is what separates it from the plain kernel.
// Staging in parameter order instead of [tap][row][col] (conflict-free 8-byte LDS writes): 734 -> 717 us on one box.
template <int NT>
__global__ __launch_bounds__(NT) void adam_frags_kernel(AdamFragArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* s5 = reinterpret_cast<bf16_t*>(smem);                   // [16 co][16 ci x 125 taps (+4)]
  bf16_t* s3 = s5 + AF_T * AfRow<AF_TAPS5>::RS;                   // [16 co][16 ci x 27 taps (+4)]
  int i = 0;
  while (i + 1 < a.nblocks && (int)blockIdx.x >= a.first[i + 1]) ++i;
  const int b = blockIdx.x - a.first[i];
  const int co_n = a.co[i], ci_n = a.ci[i];
  // tiles over the PADDED extents (32-row tiles of either role): a tile beyond the data only writes the operands' zero padding
  const int nit = rup(ci_n, 32) / AF_T;
  // workgroup -> tile: the ci tile fastest (a row of the parameter tensor is walked by consecutive workgroups)
  const int it_ = b % nit, ct_ = b / nit;
  const int co0 = ct_ * AF_T, ci0 = it_ * AF_T;
  const AdamHyper h = a.hd ? *a.hd : a.h;
  adam_tile_rows<AF_TAPS5, NT>(h, a.p5[i], a.g5[i], a.m5[i], a.v5[i], co_n, ci_n, co0, ci0, s5);
  adam_tile_rows<AF_TAPS3, NT>(h, a.p3[i], a.g3[i], a.m3[i], a.v3[i], co_n, ci_n, co0, ci0, s3);
#if RM_ADAM_TIMING >= 1
  return;      // TIMING BUILDS ONLY (no operands written)
#endif
  __syncthreads();

  // ---- the fragment-major operands (layouts: include/repmode_hip.h, repmode_expert_frags).  Forward role: rows = co,
  // reduction = ci: tile (rt = co0 / 32, kc = ci0 / 16) of [slot][tap][CoP/32][CiP/16][32][16], this workgroup's 16 rows are
  // 512 contiguous bytes of it.  128 threads move one tap's piece as dwords; the two halves of the workgroup take two taps.
  constexpr int NG = NT / 128;                                   // taps in flight: 128 threads move one tap's piece
  const int half = threadIdx.x >> 7, t128 = threadIdx.x & 127;
  if (a.wf[i] && ci0 < rup(ci_n, 16)) {          // (the forward role pads its reduction, ci, to 16 only)
    const int coP = rup(co_n, 32), ciP = rup(ci_n, 16);
    const size_t tap_stride = (size_t)coP * ciP;                   // elements per tap of one slot
    const size_t tile = ((size_t)(co0 / 32) * (ciP / 16) + ci0 / 16) * (32 * 16) + (size_t)(co0 % 32) * 16;
    uint32_t* out = reinterpret_cast<uint32_t*>(a.wf[i] + tile);
    const int row = t128 >> 3, col = 2 * (t128 & 7);              // this thread's dword: row co = row, ci pair (col, col + 1)
    for (int tap = half; tap < AF_TAPS5; tap += NG) {
      out[(size_t)tap * tap_stride / 2 + t128] = af_pair_cols<AF_TAPS5>(s5, tap, row, col);
      const int dz = tap / 25, dy = (tap / 5) % 5, dx = tap % 5;
      if (dz >= 1 && dz <= 3 && dy >= 1 && dy <= 3) {               // slot 1: the rows a centred-3x3x3 convolution reads
        const bool in = dx >= 1 && dx <= 3;
        const int t3 = ((dz - 1) * 3 + (dy - 1)) * 3 + (dx - 1);
        out[((size_t)AF_TAPS5 + tap) * tap_stride / 2 + t128] = in ? af_pair_cols<AF_TAPS3>(s3, t3, row, col) : 0u;
      }
    }
  }
  // Data-gradient role: rows = ci, reduction = co, taps flipped: tile (rt = ci0 / 32, kc = co0 / 16) of
  // [slot][124 - tap][CiP/32][CoP/16][32][16]; element (row = ci, k = co) comes from lds[tap][co][ci]: a transposed read.
  if (a.wd[i] && co0 < rup(co_n, 16)) {
    const int ciP = rup(ci_n, 32), coP = rup(co_n, 16);
    const size_t tap_stride = (size_t)ciP * coP;
    const size_t tile = ((size_t)(ci0 / 32) * (coP / 16) + co0 / 16) * (32 * 16) + (size_t)(ci0 % 32) * 16;
    uint32_t* out = reinterpret_cast<uint32_t*>(a.wd[i] + tile);
    const int row = t128 >> 3, kp = t128 & 7;                       // this thread's dword: row ci = row, co pair (2 kp, 2 kp + 1)
    for (int tap = half; tap < AF_TAPS5; tap += NG) {
      const int tap_out = AF_TAPS5 - 1 - tap;
      out[(size_t)tap_out * tap_stride / 2 + t128] = af_pair_rows<AF_TAPS5>(s5, tap, 2 * kp, row);
      const int dz = tap / 25, dy = (tap / 5) % 5, dx = tap % 5;
      if (dz >= 1 && dz <= 3 && dy >= 1 && dy <= 3) {
        const bool in = dx >= 1 && dx <= 3;
        const int t3 = ((dz - 1) * 3 + (dy - 1)) * 3 + (dx - 1);
        out[((size_t)AF_TAPS5 + tap_out) * tap_stride / 2 + t128] = in ? af_pair_rows<AF_TAPS3>(s3, t3, 2 * kp, row) : 0u;
      }
    }
  }
}

#ifndef AF_THREADS
#define AF_THREADS 512
#endif
constexpr int AF_LDS_BYTES = AF_T * (AfRow<AF_TAPS5>::RS + AfRow<AF_TAPS3>::RS) * 2;

int fill_hyper(AdamHyper* h, double lr, double beta1, double beta2, double eps, long step) {
  RM_REQUIRE(step >= 1, "adam: step count starts at 1 (got %ld)", step);
  RM_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "adam: bad hyper-parameters");
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2 = 1.0 - std::pow(beta2, (double)step);
  h->w1 = (float)(1.0 - beta1);
  h->beta2 = (float)beta2;
  h->w2 = (float)(1.0 - beta2);
  h->bc2_sqrt = (float)std::sqrt(bc2);
  h->eps = (float)eps;
  h->neg_step = (float)(-(lr / bc1));
  return REPMODE_OK;
}

}  // namespace

// The step's constants on the DEVICE: *step_dev += 1, then the AdamHyper of that step count into hyper_dev -- what the host
// computes in fill_hyper, in the same double arithmetic -- so that a captured train step (a HIP graph replay runs no host code)
// advances its own step count.  One thread.
namespace {
__global__ void adam_hyper_kernel(long* __restrict__ step_dev, AdamHyper* __restrict__ out, double lr, double beta1, double beta2, double eps) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const long t = *step_dev + 1;
  *step_dev = t;
  const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
  out->w1 = (float)(1.0 - beta1);
  out->beta2 = (float)beta2;
  out->w2 = (float)(1.0 - beta2);
  out->bc2_sqrt = (float)sqrt(bc2);
  out->eps = (float)eps;
  out->neg_step = (float)(-(lr / bc1));
}
}  // namespace

extern "C" int repmode_adam_hyper_dev(long* step_dev, float* hyper_dev, double lr, double beta1, double beta2, double eps, void* stream) {
  RM_REQUIRE(step_dev && hyper_dev, "adam_hyper_dev: null pointer");
  RM_REQUIRE(((uintptr_t)step_dev & 7) == 0 && ((uintptr_t)hyper_dev & 15) == 0, "adam_hyper_dev: misaligned buffers");
  RM_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0, "adam: bad hyper-parameters");
  static_assert(sizeof(AdamHyper) <= REPMODE_ADAM_HYPER_FLOATS * sizeof(float), "hyper_dev holds an AdamHyper");
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), step_dev, reinterpret_cast<AdamHyper*>(hyper_dev),
                     lr, beta1, beta2, eps);
  RM_LAUNCH_CHECK("adam_hyper_dev");
  return REPMODE_OK;
}

static int adam_multi_impl(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v, const long* numel,
                           double lr, double beta1, double beta2, double eps, long step, const float* hyper_dev, void* stream) {
  RM_REQUIRE(p && g && m && v && numel, "adam_multi: null pointer");
  RM_REQUIRE(ntensors > 0 && ntensors <= ADAM_MAX, "adam_multi: 1..%d tensors per call, got %d", ADAM_MAX, ntensors);
  AdamMultiArgs a{};
  if (hyper_dev) a.hd = reinterpret_cast<const AdamHyper*>(hyper_dev);
  else if (int rc = fill_hyper(&a.h, lr, beta1, beta2, eps, step)) return rc;
  a.nt = ntensors;
  long total = 0;
  for (int i = 0; i < ntensors; ++i) {
    RM_REQUIRE(p[i] && g[i] && m[i] && v[i] && numel[i] > 0, "adam_multi: bad tensor %d", i);
    RM_REQUIRE((((uintptr_t)p[i] | (uintptr_t)g[i] | (uintptr_t)m[i] | (uintptr_t)v[i]) & 3) == 0, "adam_multi: tensor %d is not float-aligned", i);
    a.p[i] = p[i]; a.g[i] = g[i]; a.m[i] = m[i]; a.v[i] = v[i]; a.numel[i] = numel[i];
    a.first[i] = (int)total;
    total += (numel[i] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  }
  a.first[ntensors] = (int)total;
  RM_REQUIRE(total < (1L << 31), "adam_multi: grid too large");
  hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  RM_LAUNCH_CHECK("adam_multi");
  return REPMODE_OK;
}

extern "C" int repmode_adam_multi(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                  const long* numel, double lr, double beta1, double beta2, double eps, long step, void* stream) {
  return adam_multi_impl(ntensors, p, g, m, v, numel, lr, beta1, beta2, eps, step, nullptr, stream);
}

extern "C" int repmode_adam_multi_dev(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                                      const long* numel, const float* hyper_dev, void* stream) {
  RM_REQUIRE(hyper_dev, "adam_multi_dev: null pointer");
  return adam_multi_impl(ntensors, p, g, m, v, numel, 0, 0, 0, 0, 0, hyper_dev, stream);
}

static int adam_expert_frags_impl(int nblocks, float* const* p5, const float* const* g5, float* const* m5, float* const* v5,
                                  float* const* p3, const float* const* g3, float* const* m3, float* const* v3, const int* co,
                                  const int* ci, void* const* wf, void* const* wd, double lr, double beta1, double beta2,
                                  double eps, long step, const float* hyper_dev, void* stream) {
  RM_REQUIRE(p5 && g5 && m5 && v5 && p3 && g3 && m3 && v3 && co && ci && wf && wd, "adam_expert_frags: null pointer");
  RM_REQUIRE(nblocks > 0 && nblocks <= AF_MAX, "adam_expert_frags: 1..%d blocks per call, got %d", AF_MAX, nblocks);
  AdamFragArgs a{};
  if (hyper_dev) a.hd = reinterpret_cast<const AdamHyper*>(hyper_dev);
  else if (int rc = fill_hyper(&a.h, lr, beta1, beta2, eps, step)) return rc;
  a.nblocks = nblocks;
  long total = 0;
  for (int i = 0; i < nblocks; ++i) {
    RM_REQUIRE(p5[i] && g5[i] && m5[i] && v5[i] && p3[i] && g3[i] && m3[i] && v3[i] && co[i] > 0 && ci[i] > 0, "adam_expert_frags: bad block %d", i);
    RM_REQUIRE((((uintptr_t)wf[i] | (uintptr_t)wd[i]) & 15) == 0, "adam_expert_frags: operand buffers must be 16-byte aligned");
    a.p5[i] = p5[i]; a.g5[i] = g5[i]; a.m5[i] = m5[i]; a.v5[i] = v5[i];
    a.p3[i] = p3[i]; a.g3[i] = g3[i]; a.m3[i] = m3[i]; a.v3[i] = v3[i];
    a.wf[i] = static_cast<bf16_t*>(wf[i]); a.wd[i] = static_cast<bf16_t*>(wd[i]);
    a.co[i] = co[i]; a.ci[i] = ci[i];
    a.first[i] = (int)total;
    total += (long)(rup(co[i], 32) / AF_T) * (rup(ci[i], 32) / AF_T);
  }
  a.first[nblocks] = (int)total;
  RM_REQUIRE(total < (1L << 31), "adam_expert_frags: grid too large");
  static bool attr_done[32] = {};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  if (!attr_done[dev & 31]) {
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&adam_frags_kernel<AF_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_BYTES));
    attr_done[dev & 31] = true;
  }
  hipLaunchKernelGGL(adam_frags_kernel<AF_THREADS>, dim3((unsigned)total), dim3(AF_THREADS), RM_ADAM_TIMING >= 3 ? 1024 : AF_LDS_BYTES, static_cast<hipStream_t>(stream), a);
  RM_LAUNCH_CHECK("adam_expert_frags");
  return REPMODE_OK;
}

extern "C" int repmode_adam_expert_frags(int nblocks, float* const* p5, const float* const* g5, float* const* m5, float* const* v5,
                                         float* const* p3, const float* const* g3, float* const* m3, float* const* v3, const int* co,
                                         const int* ci, void* const* wf, void* const* wd, double lr, double beta1, double beta2,
                                         double eps, long step, void* stream) {
  return adam_expert_frags_impl(nblocks, p5, g5, m5, v5, p3, g3, m3, v3, co, ci, wf, wd, lr, beta1, beta2, eps, step, nullptr, stream);
}

extern "C" int repmode_adam_expert_frags_dev(int nblocks, float* const* p5, const float* const* g5, float* const* m5, float* const* v5,
                                             float* const* p3, const float* const* g3, float* const* m3, float* const* v3, const int* co,
                                             const int* ci, void* const* wf, void* const* wd, const float* hyper_dev, void* stream) {
  RM_REQUIRE(hyper_dev, "adam_expert_frags_dev: null pointer");
  return adam_expert_frags_impl(nblocks, p5, g5, m5, v5, p3, g3, m3, v3, co, ci, wf, wd, 0, 0, 0, 0, 0, hyper_dev, stream);
}
