// pipeline.hip -- the two ends of the train step that the reference runs on the host.
//
//   crop_flip   fnet/data/SSPdataset.py:137-155 (data_aug): random crop of a (signal, target) volume pair to the patch
//               size + flips along z / y / x, for a whole batch in one launch from DEVICE-RESIDENT volumes.  The random
//               draws stay on the host (same numpy call order as the reference, repmode_amd/data.py); the kernel does the
//               indexing: out[n][z][y][x] = vol_n[s0 + (fz ? pd-1-z : z)][s1 + (fy ? ..)][s2 + (fx ? ..)].
//   mse_loss    fnet/fnet_model.py:108-109, 115-122: MSELoss(reduction='none') -> mean, its gradient, the per-sample
//               means (`loss_diff`) and the per-task means of the logged dict, without a host synchronisation:
//               one pass over (output, target) writes d(loss)/d(output) and per-sample sums; a one-workgroup kernel
//               finishes loss, per-sample and per-task means.  Replaces ~10 stock elementwise / reduction launches.
//
// Both are HBM-bound single passes: crop_flip moves 2 * N * patch floats in and out, mse_loss reads 2 and writes 1
// float per voxel.
#include "common.h"

namespace {

constexpr int CF_MAX = REPMODE_CROP_MAX_SAMPLES;

struct CropArgs {
  const float* sig[CF_MAX];
  const float* tgt[CF_MAX];
  int dims[CF_MAX][3];     // source volume D, H, W
  int start[CF_MAX][3];
  int flip[CF_MAX];        // bit 0: z, bit 1: y, bit 2: x
  int pd, ph, pw;
};

// grid (ceil(pw/256 per row..), rows = pd*ph, 2n): one thread per output voxel; rows of the patch are contiguous in
// the source too (reversed when x is flipped), so reads and writes coalesce along x
__global__ __launch_bounds__(256) void crop_flip_kernel(CropArgs a, float* __restrict__ sig_out, float* __restrict__ tgt_out) {
  const int which = blockIdx.z & 1, n = blockIdx.z >> 1;
  const int row = blockIdx.y;                       // z * ph + y of the patch
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= a.pw) return;
  const int z = row / a.ph, y = row % a.ph;
  const int f = a.flip[n];
  const int sz = a.start[n][0] + ((f & 1) ? a.pd - 1 - z : z);
  const int sy = a.start[n][1] + ((f & 2) ? a.ph - 1 - y : y);
  const int sx = a.start[n][2] + ((f & 4) ? a.pw - 1 - x : x);
  const float* src = which ? a.tgt[n] : a.sig[n];
  float* dst = which ? tgt_out : sig_out;
  const size_t H = a.dims[n][1], W = a.dims[n][2];
  dst[((size_t)n * a.pd * a.ph + row) * a.pw + x] = src[((size_t)sz * H + sy) * W + sx];
}

constexpr int MSE_THREADS = 256;

// grid (blocks per sample, n): grid-stride over the sample's v voxels, 4 floats per thread per iteration
__global__ __launch_bounds__(MSE_THREADS) void mse_fwd_bwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                                 float* __restrict__ dout, float* __restrict__ sums, long v,
                                                                 float gscale) {
  const int n = blockIdx.y;
  const float* o = out + (size_t)n * v;
  const float* t = tgt + (size_t)n * v;
  float* d = dout ? dout + (size_t)n * v : nullptr;
  float acc = 0.f;
  const long v4 = v >> 2;
  for (long i = (long)blockIdx.x * MSE_THREADS + threadIdx.x; i < v4; i += (long)gridDim.x * MSE_THREADS) {
    const f32x4 a = reinterpret_cast<const f32x4*>(o)[i], b = reinterpret_cast<const f32x4*>(t)[i];
    const f32x4 e = a - b;
    acc += e.x * e.x + e.y * e.y + e.z * e.z + e.w * e.w;
    if (d) reinterpret_cast<f32x4*>(d)[i] = e * gscale;
  }
  if (blockIdx.x == 0) {
    for (long i = (v4 << 2) + threadIdx.x; i < v; i += MSE_THREADS) {
      const float e = o[i] - t[i];
      acc += e * e;
      if (d) d[i] = e * gscale;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float part[MSE_THREADS / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < MSE_THREADS / 64; ++w) s += part[w];
    atomicAdd(&sums[n], s);
  }
}

// one workgroup: per-sample means, the batch mean, per-task means (NaN-free: tasks absent from the batch report 0 with count 0)
__global__ void mse_finish_kernel(float* __restrict__ sums, const int32_t* __restrict__ sample_task, int n, long v, int num_tasks,
                                  float* __restrict__ loss, float* __restrict__ loss_sample, float* __restrict__ task_mean,
                                  float* __restrict__ task_count) {
  __shared__ float tsum[64], tcnt[64];
  const int tid = threadIdx.x;
  if (tid < 64) { tsum[tid] = 0.f; tcnt[tid] = 0.f; }
  __syncthreads();
  float total = 0.f;
  if (tid == 0) {
    for (int i = 0; i < n; ++i) {
      const float m = sums[i] / (float)v;
      loss_sample[i] = m;
      total += m;
      if (sample_task) {
        const int t = sample_task[i];
        if (t >= 0 && t < num_tasks && t < 64) { tsum[t] += m; tcnt[t] += 1.f; }
      }
      sums[i] = 0.f;                       // leave the accumulator clear for the next call
    }
    *loss = total / (float)n;
  }
  __syncthreads();
  if (task_mean && tid < num_tasks && tid < 64) {
    task_mean[tid] = tcnt[tid] > 0.f ? tsum[tid] / tcnt[tid] : 0.f;
    task_count[tid] = tcnt[tid];
  }
}

}  // namespace

extern "C" int repmode_crop_flip(const float* const* signal_vols, const float* const* target_vols, const int* dims,
                                 const int* starts, const int* flips, int n, int pd, int ph, int pw, float* signal_out,
                                 float* target_out, void* stream) {
  RM_REQUIRE(signal_vols && target_vols && dims && starts && flips && signal_out && target_out, "crop_flip: null pointer");
  RM_REQUIRE(n > 0 && n <= CF_MAX, "crop_flip: 1..%d samples per call, got %d", CF_MAX, n);
  RM_REQUIRE(pd > 0 && ph > 0 && pw > 0, "crop_flip: bad patch size");
  CropArgs a{};
  a.pd = pd; a.ph = ph; a.pw = pw;
  for (int i = 0; i < n; ++i) {
    RM_REQUIRE(signal_vols[i] && target_vols[i], "crop_flip: null volume %d", i);
    a.sig[i] = signal_vols[i];
    a.tgt[i] = target_vols[i];
    const int p[3] = {pd, ph, pw};
    for (int k = 0; k < 3; ++k) {
      a.dims[i][k] = dims[3 * i + k];
      a.start[i][k] = starts[3 * i + k];
      RM_REQUIRE(a.start[i][k] >= 0 && a.start[i][k] + p[k] <= a.dims[i][k], "crop_flip: crop of sample %d leaves its volume (axis %d)", i, k);
    }
    RM_REQUIRE(flips[i] >= 0 && flips[i] < 8, "crop_flip: bad flip mask %d", flips[i]);
    a.flip[i] = flips[i];
  }
  hipLaunchKernelGGL(crop_flip_kernel, dim3((pw + 255) / 256, pd * ph, 2 * n), dim3(256), 0, static_cast<hipStream_t>(stream), a,
                     signal_out, target_out);
  RM_LAUNCH_CHECK("crop_flip");
  return REPMODE_OK;
}

extern "C" int repmode_mse_loss(const float* out, const float* target, const int32_t* sample_task, int n, long v, int num_tasks,
                                float* dout, float* sums_ws, float* loss, float* loss_sample, float* task_mean, float* task_count,
                                void* stream) {
  RM_REQUIRE(out && target && sums_ws && loss && loss_sample, "mse_loss: null pointer");
  RM_REQUIRE(n > 0 && v > 0, "mse_loss: bad shape");
  RM_REQUIRE(((uintptr_t)out & 15) == 0 && ((uintptr_t)target & 15) == 0 && ((uintptr_t)dout & 15) == 0 && (v % 4 == 0 || n == 1),
             "mse_loss: 16-byte aligned tensors with a multiple of 4 voxels per sample");
  RM_REQUIRE(!task_mean || (sample_task && task_count && num_tasks > 0 && num_tasks <= 64), "mse_loss: per-task means need sample_task, task_count and at most 64 tasks");
  hipStream_t s = static_cast<hipStream_t>(stream);
  long blocks = (v / 4 + MSE_THREADS * 4 - 1) / (MSE_THREADS * 4);
  if (blocks < 1) blocks = 1;
  if (blocks * n > 2048) blocks = (2048 + n - 1) / n;
  if (repmode_deterministic() && blocks > repmode_det_cap(RM_DET_MSE)) blocks = repmode_det_cap(RM_DET_MSE);     // (two workgroups per sample: two addends on the cleared sum)
  hipLaunchKernelGGL(mse_fwd_bwd_kernel, dim3((unsigned)blocks, n), dim3(MSE_THREADS), 0, s, out, target, dout, sums_ws, v,
                     2.0f / ((float)n * (float)v));
  RM_LAUNCH_CHECK("mse_fwd_bwd");
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(64), 0, s, sums_ws, sample_task, n, v, num_tasks, loss, loss_sample, task_mean,
                     task_count);
  RM_LAUNCH_CHECK("mse_finish");
  return REPMODE_OK;
}

// ------------------------------------------------------------------------------------------------
// Sliding-window inference (fnet/fnet_model.py:149-223): the two ends of a batch of patches.
//   patch_gather  :196-205  the batch's crops of the (device-resident) volume, one launch (the reference slices and
//                 stacks them one by one)
//   patch_blend   :207-217  pred_sum[patch] += out * gaussian, weight_sum[patch] += gaussian for every patch of the
//                 batch.  Patches of one batch overlap, so a thread owns a VOXEL of the batch's bounding box and walks the
//                 patches in batch order: the float additions happen in the reference's order (one rounding per product
//                 and per sum, no fused multiply-add), without atomics.
// Patch origins travel in the kernel arguments (no device copy, no synchronisation).
namespace {

constexpr int PB_MAX = REPMODE_PATCH_MAX;

struct PatchArgs {
  int start[PB_MAX][3];
  int nb, pd, ph, pw, D, H, W;
  int lo[3], ext[3];        // bounding box of the batch (blend)
};

__global__ __launch_bounds__(256) void patch_gather_kernel(PatchArgs a, const float* __restrict__ vol, float* __restrict__ out) {
  // (rows folded into grid.x: grid.y / grid.z are limited to 65535, a patch has pd * ph rows -- advisor round 3)
  const int nxb = (a.pw + 255) / 256;
  const int n = blockIdx.y, row = blockIdx.x / nxb;
  const int x = (blockIdx.x % nxb) * 256 + threadIdx.x;
  if (x >= a.pw) return;
  const int z = row / a.ph, y = row % a.ph;
  const size_t src = ((size_t)(a.start[n][0] + z) * a.H + a.start[n][1] + y) * a.W + a.start[n][2] + x;
  out[((size_t)n * a.pd * a.ph + row) * a.pw + x] = vol[src];
}

template <typename T>
__global__ __launch_bounds__(256) void patch_blend_kernel(PatchArgs a, const T* __restrict__ out, const float* __restrict__ gauss,
                                                         float* __restrict__ pred_sum, float* __restrict__ weight_sum) {
  // (the bounding box's rows folded into grid.x: a LIFO batch that crosses a z-slab boundary has (pd + stride) * H rows, more
  // than grid.y's 65535 on volumes taller than ~1365 voxels -- advisor round 3)
  const int nxb = (a.ext[2] + 255) / 256;
  const int row = blockIdx.x / nxb;
  const int bx = (blockIdx.x % nxb) * 256 + threadIdx.x;
  if (bx >= a.ext[2]) return;
  const int z = a.lo[0] + row / a.ext[1], y = a.lo[1] + row % a.ext[1], x = a.lo[2] + bx;
  const size_t at = ((size_t)z * a.H + y) * a.W + x;
  float ps = 0.f, ws = 0.f;
  bool hit = false;
  for (int n = 0; n < a.nb; ++n) {
    const int lz = z - a.start[n][0], ly = y - a.start[n][1], lx = x - a.start[n][2];
    if ((unsigned)lz < (unsigned)a.pd && (unsigned)ly < (unsigned)a.ph && (unsigned)lx < (unsigned)a.pw) {
      if (!hit) { ps = pred_sum[at]; ws = weight_sum[at]; hit = true; }
      const size_t li = ((size_t)lz * a.ph + ly) * a.pw + lx;
      const float g = gauss[li];
      float o;
      if constexpr (sizeof(T) == 2) o = bf16_to_f32(out[(size_t)n * a.pd * a.ph * a.pw + li]);
      else o = out[(size_t)n * a.pd * a.ph * a.pw + li];
      ps = __fadd_rn(ps, __fmul_rn(o, g));
      ws = __fadd_rn(ws, g);
    }
  }
  if (hit) { pred_sum[at] = ps; weight_sum[at] = ws; }
}

int fill_patch_args(PatchArgs& a, const int* starts, int nb, int pd, int ph, int pw, int D, int H, int W, const char* who) {
  RM_REQUIRE(starts, "%s: null patch origins", who);
  RM_REQUIRE(nb > 0 && nb <= PB_MAX, "%s: 1..%d patches per call, got %d", who, PB_MAX, nb);
  RM_REQUIRE(pd > 0 && ph > 0 && pw > 0 && D > 0 && H > 0 && W > 0, "%s: bad shape", who);
  a.nb = nb; a.pd = pd; a.ph = ph; a.pw = pw; a.D = D; a.H = H; a.W = W;
  const int pdim[3] = {pd, ph, pw}, vdim[3] = {D, H, W};
  int hi[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k) a.lo[k] = vdim[k];
  for (int n = 0; n < nb; ++n)
    for (int k = 0; k < 3; ++k) {
      const int s = starts[3 * n + k];
      RM_REQUIRE(s >= 0 && s + pdim[k] <= vdim[k], "%s: patch %d leaves the volume", who, n);
      a.start[n][k] = s;
      if (s < a.lo[k]) a.lo[k] = s;
      if (s + pdim[k] > hi[k]) hi[k] = s + pdim[k];
    }
  for (int k = 0; k < 3; ++k) a.ext[k] = hi[k] - a.lo[k];
  return REPMODE_OK;
}

}  // namespace

// out[n][pd][ph][pw] = vol[start_n + (z, y, x)]; starts: HOST array [nb][3]
extern "C" int repmode_patch_gather(const float* vol, int D, int H, int W, const int* starts, int nb, int pd, int ph, int pw,
                                    float* out, void* stream) {
  RM_REQUIRE(vol && out, "patch_gather: null pointer");
  PatchArgs a{};
  const int rc = fill_patch_args(a, starts, nb, pd, ph, pw, D, H, W, "patch_gather");
  if (rc != REPMODE_OK) return rc;
  RM_REQUIRE((long)ceil_div(pw, 256) * pd * ph < (1L << 31), "patch_gather: patch too large");
  hipLaunchKernelGGL(patch_gather_kernel, dim3((unsigned)(ceil_div(pw, 256) * pd * ph), nb), dim3(256), 0, static_cast<hipStream_t>(stream), a, vol, out);
  RM_LAUNCH_CHECK("patch_gather");
  return REPMODE_OK;
}

// pred_sum[start_n + v] += out[n][v] * gauss[v], weight_sum[start_n + v] += gauss[v], n = 0 .. nb-1 in this order per voxel.
// out: float32 or bf16 (dtype); starts: HOST array [nb][3]
extern "C" int repmode_patch_blend(const void* out, int dtype, const float* gauss, const int* starts, int nb, int pd, int ph,
                                   int pw, float* pred_sum, float* weight_sum, int D, int H, int W, void* stream) {
  RM_REQUIRE(out && gauss && pred_sum && weight_sum, "patch_blend: null pointer");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "patch_blend: bad dtype %d", dtype);
  PatchArgs a{};
  const int rc = fill_patch_args(a, starts, nb, pd, ph, pw, D, H, W, "patch_blend");
  if (rc != REPMODE_OK) return rc;
  RM_REQUIRE((long)ceil_div(a.ext[2], 256) * a.ext[0] * a.ext[1] < (1L << 31), "patch_blend: bounding box too large");
  const dim3 grid((unsigned)(ceil_div(a.ext[2], 256) * a.ext[0] * a.ext[1]));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == REPMODE_F32)
    hipLaunchKernelGGL(patch_blend_kernel<float>, grid, dim3(256), 0, s, a, static_cast<const float*>(out), gauss, pred_sum, weight_sum);
  else
    hipLaunchKernelGGL(patch_blend_kernel<bf16_t>, grid, dim3(256), 0, s, a, static_cast<const bf16_t*>(out), gauss, pred_sum, weight_sum);
  RM_LAUNCH_CHECK("patch_blend");
  return REPMODE_OK;
}
