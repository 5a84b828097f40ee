// api.hip -- error reporting, ABI/device queries and the naive diagnostic kernels.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "tail_jobs.h"
#include <map>
#include <mutex>
#include <utility>

static thread_local char g_err[512] = "";

void repmode_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static int g_deterministic = []() { const char* e = getenv("REPMODE_DETERMINISTIC"); return e ? atoi(e) : 0; }();
bool repmode_deterministic() { return g_deterministic != 0; }
int repmode_det_cap(int site) {
  static const int single = []() { const char* e = getenv("REPMODE_DET_SINGLE"); return e ? atoi(e) : 0; }();
  return (single & site) ? 1 : 2;
}
extern "C" int repmode_set_deterministic(int on) { g_deterministic = on ? 1 : 0; return REPMODE_OK; }
extern "C" int repmode_get_deterministic(void) { return g_deterministic; }

// CUs the persistent grids (conv5_ws_kernel, the stream-K filter gradient: at most one workgroup per CU) leave free -- room for
// a communication kernel beside them (data-parallel training: RCCL's all-reduce kernels otherwise queue behind a launch whose
// workgroups hold every CU for its whole duration).  REPMODE_RESERVE_CUS / repmode_set_reserve_cus; 0 by default.
static int g_reserve_cus = []() { const char* e = std::getenv("REPMODE_RESERVE_CUS"); return e ? std::atoi(e) : 0; }();
int repmode_reserve_cus() { return g_reserve_cus; }
extern "C" int repmode_set_reserve_cus(int n) { g_reserve_cus = n > 0 ? n : 0; return REPMODE_OK; }
extern "C" int repmode_get_reserve_cus(void) { return g_reserve_cus; }

extern "C" int repmode_abi_version(void) { return REPMODE_ABI_VERSION; }
extern "C" const char* repmode_last_error(void) { return g_err; }

extern "C" int repmode_device_arch(int dev, char* buf, int buflen) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || dev < 0 || dev >= count) {
    (void)hipGetLastError();
    repmode_set_error("no HIP device %d (count %d)", dev, count);
    return REPMODE_ENODEV;
  }
  hipDeviceProp_t prop;
  RM_HIP(hipGetDeviceProperties(&prop, dev));
  if (buf && buflen > 0) {
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
  }
  return REPMODE_OK;
}

namespace {
struct ProfRec {
  hipEvent_t a, b;
  double work;
  int kind;
};
std::vector<ProfRec> g_prof;   // event pool, reused across enable() calls
size_t g_prof_n = 0;           // records in use
bool g_prof_on = false;
unsigned g_prof_mask = ~0u;   // bit k set: record kernel kind k
bool g_prof_open = false;
bool g_prof_paused = false;
std::mutex g_prof_mu;   // forward launches come from the caller's thread, backward launches from autograd's
}  // namespace

void repmode_prof_begin(int kind, double work, hipStream_t s) {
  if (!g_prof_on) return;                       // (cheap unlocked test: the flag only changes between steps)
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (!g_prof_on || g_prof_paused || !((g_prof_mask >> kind) & 1u)) return;
  if (g_prof_n == g_prof.size()) {
    ProfRec r{};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    g_prof.push_back(r);
  }
  ProfRec& r = g_prof[g_prof_n];
  r.kind = kind;
  r.work = work;
  (void)hipEventRecord(r.a, s);
  g_prof_open = true;
}

void repmode_prof_end(hipStream_t s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (!g_prof_on || !g_prof_open) return;
  (void)hipEventRecord(g_prof[g_prof_n].b, s);
  ++g_prof_n;
  g_prof_open = false;
}

namespace {
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, float*> g_scratch;
std::map<std::pair<int, hipStream_t>, int> g_bn_half;
}  // namespace

float* repmode_zero_scratch(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { repmode_set_error("zero_scratch: hipGetDevice failed"); return nullptr; }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  auto it = g_scratch.find({dev, s});
  if (it != g_scratch.end()) return it->second;
  float* p = nullptr;
  const size_t bytes = (REPMODE_ZERO_SCRATCH_FLOATS + REPMODE_SCRATCH_TAIL_WORDS) * sizeof(float);
  hipError_t e = hipMalloc(&p, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(p, 0, bytes, s);
  if (e != hipSuccess) { repmode_set_error("zero_scratch: %s", hipGetErrorString(e)); return nullptr; }
  g_scratch[{dev, s}] = p;
  return p;
}

int repmode_bn_scratch_half(hipStream_t s) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  int& h = g_bn_half[{dev, s}];
  const int mine = h;
  h ^= 1;
  return mine;
}

// ---- deferred small jobs (tail_jobs.h): per (device, stream) queue, taken by the next conv5 launch on the stream
namespace {
std::mutex g_tail_mu;
std::map<std::pair<int, hipStream_t>, TailJobs> g_tail;

__global__ __launch_bounds__(TAIL_THREADS) void tail_jobs_kernel(TailJobs t) {
  __shared__ float lds[TAIL_LDS_BYTES / 4];
  tail_run(t, blockIdx.x, threadIdx.x, lds);
}
}  // namespace

void repmode_tail_take(hipStream_t s, TailJobs* out) {
  out->njobs = 0;
  out->nblocks = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  std::lock_guard<std::mutex> lock(g_tail_mu);
  auto it = g_tail.find({dev, s});
  if (it == g_tail.end() || it->second.njobs == 0) return;
  *out = it->second;
  out->nblocks = (out->nblocks + 7) & ~7;
  it->second.njobs = 0;
  it->second.nblocks = 0;
}

int repmode_tail_launch(const TailJobs& t, hipStream_t s) {
  if (t.njobs == 0) return REPMODE_OK;
  hipLaunchKernelGGL(tail_jobs_kernel, dim3((unsigned)t.nblocks), dim3(TAIL_THREADS), 0, s, t);
  RM_LAUNCH_CHECK("tail_jobs");
  return REPMODE_OK;
}

int repmode_tail_push(const TailJob& job, hipStream_t s) {
  RM_REQUIRE(job.nblocks > 0 && (job.kind == 1 || job.kind == 2), "tail_push: bad job");
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  TailJobs full;
  full.njobs = 0;
  {
    std::lock_guard<std::mutex> lock(g_tail_mu);
    TailJobs& q = g_tail[{dev, s}];
    if (q.njobs == TAIL_MAX_JOBS) {       // no host launch came by: run what is queued, then queue this one
      full = q;
      full.nblocks = (full.nblocks + 7) & ~7;
      q.njobs = 0;
      q.nblocks = 0;
    }
    q.job[q.njobs++] = job;
    q.nblocks += job.nblocks;
  }
  return repmode_tail_launch(full, s);
}

// Drops what is queued without running it: after a failed call the queue may hold jobs whose tensors are gone.
extern "C" int repmode_tail_discard(void* stream) {
  TailJobs t;
  repmode_tail_take(static_cast<hipStream_t>(stream), &t);
  return REPMODE_OK;
}

extern "C" int repmode_tail_flush(void* stream) {
  TailJobs t;
  repmode_tail_take(static_cast<hipStream_t>(stream), &t);
  return repmode_tail_launch(t, static_cast<hipStream_t>(stream));
}

extern "C" int repmode_prof_enable(int on) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  if (on) g_prof_n = 0;
  g_prof_on = on != 0;
  // 2: the forward / data-gradient convolution kernels only (least perturbation)
  g_prof_mask = (on == 2) ? ((1u << REPMODE_PROF_CONV5) | (1u << REPMODE_PROF_CONV5_DEEP) | (1u << REPMODE_PROF_CONV5_THIN) | (1u << REPMODE_PROF_CONV5_WS) |
                               (1u << REPMODE_PROF_WGRAD) | (1u << REPMODE_PROF_WGRAD_THIN) | (1u << REPMODE_PROF_DEEP_MODE) | (1u << REPMODE_PROF_DEEP_MODE_DGRAD)) : ~0u;
  g_prof_open = false;
  g_prof_paused = false;
  return REPMODE_OK;
}

// pause / resume recording without discarding what has been recorded (bench.py samples every few steps)
extern "C" int repmode_prof_pause(int paused) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_prof_paused = paused != 0;
  return REPMODE_OK;
}

extern "C" int repmode_prof_summary(int kind, int* launches, double* total_ms, double* total_work) {
  RM_REQUIRE(launches && total_ms && total_work, "prof_summary: null pointer");
  RM_REQUIRE(kind >= 0 && kind < REPMODE_PROF_KINDS, "prof_summary: bad kind %d", kind);
  int n = 0;
  double ms = 0, work = 0;
  for (size_t i = 0; i < g_prof_n; ++i) {
    if (g_prof[i].kind != kind) continue;
    RM_HIP(hipEventSynchronize(g_prof[i].b));
    float t = 0.f;
    RM_HIP(hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b));
    ms += t;
    work += g_prof[i].work;
    ++n;
  }
  *launches = n;
  *total_ms = ms;
  *total_work = work;
  return REPMODE_OK;
}

extern "C" int repmode_prof_count(void) { return (int)g_prof_n; }

extern "C" int repmode_prof_record(int i, int* kind, double* ms, double* work) {
  RM_REQUIRE(kind && ms && work && i >= 0 && (size_t)i < g_prof_n, "prof_record: bad index %d", i);
  RM_HIP(hipEventSynchronize(g_prof[i].b));
  float t = 0.f;
  RM_HIP(hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b));
  *kind = g_prof[i].kind;
  *ms = t;
  *work = g_prof[i].work;
  return REPMODE_OK;
}

namespace {

// one thread per output element y[n][v][co]; f32 accumulate in tap-major, channel-minor order
template <typename T>
__global__ void conv5_naive_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                   const int32_t* __restrict__ sample_slot, float* __restrict__ y, int N, int D,
                                   int H, int W, int Cin, int Cout, int CinP, int CoutP) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)N * D * H * W * Cout;
  if (idx >= total) return;
  const int co = (int)(idx % Cout);
  long v = idx / Cout;
  const int gx = (int)(v % W); v /= W;
  const int gy = (int)(v % H); v /= H;
  const int gz = (int)(v % D);
  const int n = (int)(v / D);
  const T* ws = w + (size_t)sample_slot[n] * REPMODE_TAPS * CoutP * CinP;
  float acc = 0.f;
  for (int tap = 0; tap < REPMODE_TAPS; ++tap) {
    const int z = gz + tap / 25 - 2, yy = gy + (tap / 5) % 5 - 2, xx = gx + tap % 5 - 2;
    if ((unsigned)z >= (unsigned)D || (unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
    const T* xp = x + ((((size_t)n * D + z) * H + yy) * W + xx) * Cin;
    constexpr int KC = 32 / sizeof(T);     // fragment-major filter layout, see gatrep.hip
    const int nkc = CinP / KC;
    const T* wp = ws + (((size_t)tap * (CoutP / 32) + co / 32) * nkc) * (32 * KC) + (co % 32) * KC;
    for (int ci = 0; ci < Cin; ++ci) acc += to_f32<T>(xp[ci]) * to_f32<T>(wp[(size_t)(ci / KC) * (32 * KC) + ci % KC]);
  }
  y[idx] = acc;
}

// one thread per dw[slot][tap][co][ci]
template <typename T>
__global__ void wgrad_naive_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                   const int32_t* __restrict__ sample_slot, int nslots, float* __restrict__ dw,
                                   int N, int D, int H, int W, int Cin, int Cout) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)nslots * REPMODE_TAPS * Cout * Cin;
  if (idx >= total) return;
  const int ci = (int)(idx % Cin);
  long r = idx / Cin;
  const int co = (int)(r % Cout); r /= Cout;
  const int tap = (int)(r % REPMODE_TAPS);
  const int slot = (int)(r / REPMODE_TAPS);
  const int dz = tap / 25 - 2, dyy = (tap / 5) % 5 - 2, dx = tap % 5 - 2;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) {
    if (sample_slot[n] != slot) continue;
    for (int z = 0; z < D; ++z) {
      const int zi = z + dz;
      if ((unsigned)zi >= (unsigned)D) continue;
      for (int y = 0; y < H; ++y) {
        const int yi = y + dyy;
        if ((unsigned)yi >= (unsigned)H) continue;
        for (int xx = 0; xx < W; ++xx) {
          const int xi = xx + dx;
          if ((unsigned)xi >= (unsigned)W) continue;
          acc += to_f32<T>(dy[((((size_t)n * D + z) * H + y) * W + xx) * Cout + co]) *
                 to_f32<T>(x[((((size_t)n * D + zi) * H + yi) * W + xi) * Cin + ci]);
        }
      }
    }
  }
  dw[idx] = acc;
}

}  // namespace

extern "C" int repmode_debug_conv5_naive(const void* x, const void* w, const int32_t* sample_slot, float* y,
                                         int n, int d, int h, int wdim, int cin, int cout, int dtype,
                                         void* stream) {
  RM_REQUIRE(x && w && sample_slot && y, "conv5_naive: null pointer");
  const long total = (long)n * d * h * wdim * cout;
  const int cinp = repmode_padded_channels(cin, dtype, 1), coutp = repmode_padded_channels(cout, dtype, 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == REPMODE_F32)
    hipLaunchKernelGGL(conv5_naive_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (const float*)w,
                       sample_slot, y, n, d, h, wdim, cin, cout, cinp, coutp);
  else
    hipLaunchKernelGGL(conv5_naive_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)w, sample_slot, y, n, d, h, wdim, cin, cout, cinp, coutp);
  RM_LAUNCH_CHECK("conv5_naive");
  return REPMODE_OK;
}

extern "C" int repmode_debug_wgrad_naive(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                                         float* dw, int n, int d, int h, int wdim, int cin, int cout, int dtype,
                                         void* stream) {
  RM_REQUIRE(x && dy && sample_slot && dw, "wgrad_naive: null pointer");
  const long total = (long)nslots * REPMODE_TAPS * cout * cin;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == REPMODE_F32)
    hipLaunchKernelGGL(wgrad_naive_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, (const float*)dy,
                       sample_slot, nslots, dw, n, d, h, wdim, cin, cout);
  else
    hipLaunchKernelGGL(wgrad_naive_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x,
                       (const bf16_t*)dy, sample_slot, nslots, dw, n, d, h, wdim, cin, cout);
  RM_LAUNCH_CHECK("wgrad_naive");
  return REPMODE_OK;
}
