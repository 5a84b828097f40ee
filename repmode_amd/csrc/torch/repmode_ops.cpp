// repmode_ops.cpp -- the operator seam of the MI355X MoDE hot path: TORCH_LIBRARY ops in
// namespace `repmode`, C++ autograd, TORCH_CHECK errors, every launch on c10::hip::getCurrentHIPStream().
//
// One op call per MoDE block (gate softmax + GatRep + per-slot 5x5x5 convolution + BatchNorm3d + ReLU, forward; the
// whole backward chain lives in the C++ autograd nodes it records), one per stride-2 stage: a train step is ~30
// Python -> C++ transitions instead of ~410 ctypes calls, which is what kept the host as slow as the GPU.
//
// This file holds NO arithmetic: every FLOP is a kernel of librepmode_hip.so, reached through the C ABI of
// include/repmode_hip.h (plain pointers + sizes + stream).  What lives here is the choice of formulation per layer
// (per-task merged filter / per-expert convs on the deep levels / two-tensor skip connections / the 1-channel ends),
// output allocation through PyTorch's caching allocator, the pooled memset, the second stream of a layer, and the
// autograd bookkeeping -- the counterpart of what fnet/nn_modules/RepMode.py:171-214 leaves to stock PyTorch ops.
//
// Built by repmode_amd/csrc/build.sh into repmode_amd/librepmode_torch.so (in-tree, next to librepmode_hip.so).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include <chrono>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "repmode_hip.h"

namespace rm {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using OptTensor = c10::optional<Tensor>;

constexpr int64_t E = REPMODE_NUM_EXPERTS;
constexpr int64_t TAPS = REPMODE_TAPS;

#define RM_CALL(fn, ...)                                                                          \
  do {                                                                                            \
    const int rc__ = fn(__VA_ARGS__);                                                             \
    TORCH_CHECK(rc__ == 0, #fn " failed (code ", rc__, "): ", repmode_last_error());              \
  } while (0)

// Current-device guard on the HIP runtime itself (PyTorch-ROCm labels its device type "cuda"; the c10::hip guards want
// their own label -- the runtime call needs neither).  One process per GPU: normally a no-op.
struct DeviceGuard {
  int prev = -1, want = -1;
  explicit DeviceGuard(const at::Device& d) : want(d.index()) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (want >= 0 && prev != want) TORCH_CHECK(hipSetDevice(want) == hipSuccess, "repmode: cannot select HIP device ", want);
  }
  ~DeviceGuard() {
    if (prev >= 0 && want >= 0 && prev != want) (void)hipSetDevice(prev);
  }
};

inline void* stream_handle() { return static_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

inline int dtype_code(at::ScalarType t) {
  if (t == at::kFloat) return REPMODE_F32;
  if (t == at::kBFloat16) return REPMODE_BF16;
  TORCH_CHECK(false, "repmode computes in float32 or bfloat16, got ", t);
}
inline at::ScalarType code_dtype(int64_t code) {
  TORCH_CHECK(code == REPMODE_F32 || code == REPMODE_BF16, "repmode: bad dtype code ", code);
  return code == REPMODE_F32 ? at::kFloat : at::kBFloat16;
}

inline void require_hip(const Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), what, " is on ", t.device(), ": repmode runs on MI355X (HIP) tensors only and has no CPU fallback");
}

inline int64_t padded(int64_t c, int code, bool red) { return repmode_padded_channels((int)c, code, red ? 1 : 0); }

inline Tensor empty_like_opts(const Tensor& ref, at::IntArrayRef shape, at::ScalarType dt) {
  return at::empty(shape, ref.options().dtype(dt));
}

// ------------------------------------------------------------------------------------------------------------
// Which merged filter each sample uses (the Python side builds the three index vectors from the task ids on the
// host: repmode_amd.ops.TaskPlan; RepMode.py:44-49 one-hot embedding, :209-210 one slot in eval mode).
struct Plan {
  Tensor slot_task, sample_slot, sample_task;   // int32, on the device
  int64_t nslots = 0, n = 0, num_tasks = 0;
  bool training = true;
  int64_t task0 = 0;                            // task of slot 0 (key of the eval-mode filter cache)
};

// ------------------------------------------------------------------------------------------------------------
// One pre-zeroed buffer per train step for the float accumulation targets of the atomics-based kernels (split-K conv
// outputs, chunked filter gradients): ~60 memset launches per step become one.  The first step with a given key
// records the requested sizes in order; later steps allocate the total once, hand out independent tensors over that
// storage in the same order (not views: views would share ONE version counter) and tell the kernels to skip their own
// clearing.  Any divergence from the recorded sequence falls back to plain allocations for the rest of the step.
struct ZeroPool {
  static constexpr int64_t ALIGN = 64;   // floats (256 bytes)
  std::mutex mu;
  std::unordered_map<std::string, std::vector<int64_t>> plans;
  bool open = false;
  std::string key;
  std::vector<int64_t> req, plan;
  size_t pos = 0;
  int64_t off = 0;
  Tensor buf;

  void end_locked() {
    if (open && !req.empty()) plans[key] = req;
    open = false;
    req.clear();
    plan.clear();
    buf = Tensor();
  }
  void begin(const std::string& k, const Tensor& like) {
    std::lock_guard<std::mutex> lock(mu);
    end_locked();
    open = true;
    key = k;
    pos = 0;
    off = 0;
    auto it = plans.find(k);
    if (it != plans.end() && !it->second.empty()) {
      plan = it->second;
      int64_t total = 0;
      for (int64_t v : plan) total += (v + ALIGN - 1) / ALIGN * ALIGN;
      if (getenv("REPMODE_POOL_DEBUG")) {        // (developer aid: what the step's one memset covers)
        fprintf(stderr, "[repmode] zero pool '%s': %zu buffers, %lld floats:", k.c_str(), plan.size(), (long long)total);
        for (int64_t v : plan) fprintf(stderr, " %lld", (long long)v);
        fprintf(stderr, "\n");
      }
      buf = at::zeros({total}, like.options().dtype(at::kFloat));
    }
  }
  void end() {
    std::lock_guard<std::mutex> lock(mu);
    end_locked();
  }
  std::pair<Tensor, bool> take(at::IntArrayRef shape, const Tensor& like) {
    std::lock_guard<std::mutex> lock(mu);
    int64_t nel = 1;
    for (int64_t v : shape) nel *= v;
    if (open) {
      req.push_back(nel);
      if (buf.defined() && pos < plan.size() && plan[pos] == nel) {
        Tensor t = at::empty({0}, buf.options());
        std::vector<int64_t> strides(shape.size());
        int64_t s = 1;
        for (int64_t i = (int64_t)shape.size() - 1; i >= 0; --i) { strides[i] = s; s *= shape[i]; }
        t.set_(buf.storage(), off, shape, strides);
        off += (nel + ALIGN - 1) / ALIGN * ALIGN;
        ++pos;
        return {t, true};
      }
      buf = Tensor();   // sequence differs from the recorded one: plain allocations from here on
    }
    return {at::empty(shape, like.options().dtype(at::kFloat)), false};
  }
};
ZeroPool g_pool;

// ------------------------------------------------------------------------------------------------------------
// Where the MoDE gradient kernels put parameter gradients: a data-parallel reducer's communication buckets
// (repmode_amd.distributed.GradReducer) when registered, else fresh tensors.
struct SinkEntry {
  Tensor param, flat;
  int64_t offset;
};
std::mutex g_sink_mu;
std::unordered_map<void*, SinkEntry> g_sink;

// from_sink (optional out): the buffer is a slice of a reducer's bucket (NOT a fresh tensor)
Tensor grad_out(const Tensor& param, bool* from_sink = nullptr) {
  if (from_sink) *from_sink = false;
  {
    std::lock_guard<std::mutex> lock(g_sink_mu);
    if (!g_sink.empty()) {
      auto it = g_sink.find(param.data_ptr());
      if (it != g_sink.end() && !it->second.param.grad().defined() && it->second.param.sizes() == param.sizes()) {
        if (from_sink) *from_sink = true;
        return it->second.flat.as_strided(param.sizes(), param.strides(), it->second.offset);
      }
    }
  }
  return at::empty_like(param);
}
bool grad_sink_active() {
  std::lock_guard<std::mutex> lock(g_sink_mu);
  return !g_sink.empty();
}

// ------------------------------------------------------------------------------------------------------------
// A second HIP stream for the independent launches of one layer (deep levels: a launch has 128-256 workgroups for
// 512 slots; a layer's data gradient does not depend on its filter gradient + GatRep backward).  Everything a
// side-stream launch touches stays alive until the join, which comes before the op returns.
int64_t g_fork_max_w = []() {
  const char* e = std::getenv("REPMODE_FORK_MAX_W");
  return e ? (int64_t)std::atoi(e) : (int64_t)0;
}();

#define RM_HIP_CHECK(call)                                                                        \
  do {                                                                                            \
    const hipError_t e__ = (call);                                                                \
    TORCH_CHECK(e__ == hipSuccess, #call ": ", hipGetErrorString(e__));                          \
  } while (0)

// the library's own streams of a device: `side` for the second chain of one layer, `prep` for the step's filter
// preparation (separate, so that a layer's fork does not queue behind the remaining preparations)
struct SideStreams {
  c10::hip::HIPStream side, prep;
  hipEvent_t fork_ev, join_ev, prep_fork_ev;
};
SideStreams& side_streams(int dev) {
  static std::mutex mu;
  static std::unordered_map<int, SideStreams> per;
  std::lock_guard<std::mutex> lock(mu);
  auto it = per.find(dev);
  if (it == per.end()) {
    SideStreams d{c10::hip::getStreamFromPool(false, dev), c10::hip::getStreamFromPool(false, dev), nullptr, nullptr, nullptr};
    RM_HIP_CHECK(hipEventCreateWithFlags(&d.fork_ev, hipEventDisableTiming));
    RM_HIP_CHECK(hipEventCreateWithFlags(&d.join_ev, hipEventDisableTiming));
    RM_HIP_CHECK(hipEventCreateWithFlags(&d.prep_fork_ev, hipEventDisableTiming));
    it = per.emplace(dev, d).first;
  }
  return it->second;
}

struct Fork {
  using PerDevice = SideStreams;
  bool on = false;
  c10::hip::HIPStream main_s = c10::hip::getDefaultHIPStream();
  PerDevice* pd = nullptr;
  explicit Fork(bool enable, const Tensor& t) {
    if (!enable) return;
    on = true;
    const int dev = t.device().index();
    main_s = c10::hip::getCurrentHIPStream(dev);
    pd = &side_streams(dev);
    // the side stream is ordered after everything issued on the current stream so far
    RM_HIP_CHECK(hipEventRecord(pd->fork_ev, main_s.stream()));
    RM_HIP_CHECK(hipStreamWaitEvent(pd->side.stream(), pd->fork_ev, 0));
  }
  void to_side() const { if (on) c10::hip::setCurrentHIPStream(pd->side); }
  void to_main() const { if (on) c10::hip::setCurrentHIPStream(main_s); }
  void join() {
    if (!on) return;
    c10::hip::setCurrentHIPStream(main_s);
    RM_HIP_CHECK(hipEventRecord(pd->join_ev, pd->side.stream()));
    RM_HIP_CHECK(hipStreamWaitEvent(main_s.stream(), pd->join_ev, 0));
    on = false;
  }
  ~Fork() { if (on) c10::hip::setCurrentHIPStream(main_s); }   // (an exception between fork and join: restore the stream)
};
inline bool forks(const Tensor& x_cl) { return x_cl.size(3) > 0 && x_cl.size(3) <= g_fork_max_w; }
// Overlap of the HBM-bound GatRep kernels with the MFMA-bound convolutions (second stream): the forward's filter
// preparation of all blocks at the start of the step (op prepare_filters), and a layer's GatRep backward beside its
// data-gradient conv.  OFF by default (REPMODE_OVERLAP=1 / set_overlap): measured on one box, interleaved, 60 steps each,
// 13.71 / 13.72 ms per step without and 13.94 / 13.94 ms with -- a conv launch holds every CU's LDS and registers with
// two resident workgroups, so the GatRep workgroups only get in as the conv drains (gatrep_bwd: 30 -> 179 us per launch
// under rocprofv3) and the layer's join then waits for them.
// Volumes up to this x extent take the per-expert formulation when the batch has more than two distinct tasks
// (REPMODE_UNMERGED_MAX_W / set_unmerged_max_w; 8 = levels 3-4 of a 64-wide patch).  Measured on one box, interleaved,
// batch 8 with 8 tasks: 8 -> 13.73 ms per step, 4 (level 4 only) -> 14.37, 0 (merged everywhere) -> 16.55.
int64_t g_unmerged_max_w = []() {
  const char* e = std::getenv("REPMODE_UNMERGED_MAX_W");
  return e ? (int64_t)std::atoi(e) : (int64_t)8;
}();
// The 5x5x5 and the 3x3x3 expert's convolutions of the per-expert formulation as one launch (twice the workgroups on
// levels whose launches do not fill the chip, one launch gap less); REPMODE_DUAL_LAUNCH=0: two launches
bool g_dual_launch = []() {
  const char* e = std::getenv("REPMODE_DUAL_LAUNCH");
  return e ? std::atoi(e) != 0 : true;
}();
// BatchNorm work taken over by the forward conv's epilogue (Epi), a bit mask (REPMODE_BN_EPILOGUE / set_bn_epilogue):
//   1  eval mode without autograd: the whole BatchNorm + ReLU folded (scale into the filter, bias + ReLU in the epilogue)
//      on the layers whose conv writes the element-typed tensor -- default ON (sliding-window inference; time-neutral on
//      the 64x624x924 stack: 0.49 s with and without, the forward is conv-bound);
//   2  training: the batch statistics from the conv's epilogue instead of a statistics pass -- default OFF: measured on
//      one box, interleaved, 60 steps each: 13.90 / 13.89 ms per step with the separate pass, 14.10 / 14.12 ms with the
//      epilogue.  The pass it removes streams a tensor that is still in the 256 MB Infinity Cache at full rate (11-25 us
//      per layer); the epilogue's cross-lane reduction and atomics run inside the MFMA-bound, power-capped conv launch
//      and cost more than that.
// The per-expert levels' two convolutions through their own uniform-grid kernel (csrc/conv5_deep.hip) instead of the
// general kernel's dual-expert launch; REPMODE_DEEP=0 / set_deep_conv(False): the dual-expert launch
bool g_deep = []() {
  const char* e = std::getenv("REPMODE_DEEP");
  return e ? std::atoi(e) != 0 : true;
}();
// The FORWARD pair goes through conv5_deep only where it measured faster than the dual-expert launch (same box, us per
// launch deep / dual, batch 8: 128->256 50.0 / 41.8, 256->256 82.1 / 60.5, 512->256 121.1 / 109.7, level 4 32.3 / 28.1 and
// 48.6 / 47.6; batch 24: 114.9 / 96.6, 155.4 / 166.3, 272.2 / 322.2, level 4 62.0 / 46.9 and 117.7 / 87.1): two float outputs
// double its atomics, and a uniform grid leaves no short jobs whose atomics overlap the long jobs' MFMAs.  Rule: the
// level-3 tile with samples x input channels >= REPMODE_DEEP_FWD_MIN (default 6000; 0: always, also level 4).  The
// data-gradient form (one output) wins everywhere: 244 vs 288 us over the five layers at batch 8.
int64_t g_deep_fwd_min = []() {
  const char* e = std::getenv("REPMODE_DEEP_FWD_MIN");
  return e ? (int64_t)std::atoll(e) : (int64_t)6000;
}();
// A per-expert block as ONE launch per direction (csrc/deep_mode.hip: the five experts, the gate mix and the cross-wave
// reduction in one kernel; round 5) where repmode_deep_mode_plan takes the shape; REPMODE_DEEP_MODE=0 / set_deep_mode(0):
// round 4's five launches (conv5_deep / dual-expert launch + box + gemm3 + expert_mix).  Bit 0: forward, bit 1: data gradient,
// bit 2: the forward also leaves the BatchNorm statistics of its output (no statistics pass behind a per-expert block); bit 3:
// the data gradient's box-mean operands come out of the gate mix's backward launch (no box launch).
int64_t g_deep_mode = []() {
  const char* e = std::getenv("REPMODE_DEEP_MODE");
  return e ? (int64_t)std::atoi(e) : (int64_t)15;
}();
bool g_dual_wgrad = []() {          // (REPMODE_DUAL_WGRAD=0: the two filter gradients of a per-expert block as two launches)
  const char* e = std::getenv("REPMODE_DUAL_WGRAD");
  return e ? std::atoi(e) != 0 : true;
}();
int64_t g_bn_epilogue = []() {
  const char* e = std::getenv("REPMODE_BN_EPILOGUE");
  return e ? (int64_t)std::atoi(e) : (int64_t)1;
}();
// All merged blocks' forward filters (gate softmax + GatRep) from ONE launch at the start of a training forward pass
// (prepare_filters); REPMODE_PREPARE=0: every block merges its own filter when it runs
bool g_prepare = []() {
  const char* e = std::getenv("REPMODE_PREPARE");
  return e ? std::atoi(e) != 0 : true;
}();
bool g_overlap = []() {
  const char* e = std::getenv("REPMODE_OVERLAP");
  return e ? std::atoi(e) != 0 : false;
}();
// Deferred small jobs (csrc/tail_jobs.h): on the backward pass a block's gate backward and gradient-layout transposes are
// queued and run in the first workgroups of the block's data-gradient convolution instead of as launches of their own
// (a dependent kernel boundary costs 4-5 us; ~25 launches per step).  REPMODE_TAIL=0 / set_tail_jobs(False): launch each.
bool g_tail = []() {
  const char* e = std::getenv("REPMODE_TAIL");
  return e ? std::atoi(e) != 0 : true;
}();

// ------------------------------------------------------------------------------------------------------------
// thin wrappers of the C ABI (allocation + argument marshalling)

Tensor gate_softmax(const Tensor& gw, const Tensor& gb, const Tensor& ids, int64_t rows, int64_t num_tasks, int64_t co) {
  Tensor g = at::empty({rows, E, co}, gw.options());
  RM_CALL(repmode_gate_softmax, gw.data_ptr<float>(), gb.data_ptr<float>(), ids.data_ptr<int32_t>(), (int)rows, (int)num_tasks,
          (int)co, g.data_ptr<float>(), stream_handle());
  return g;
}

std::pair<Tensor, Tensor> gatrep_merge(const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3, const Tensor& a5,
                                       const Tensor& g, at::ScalarType dt, bool want_wf, bool want_wd) {
  const int64_t co = k5.size(0), ci = k5.size(1), s = g.size(0);
  const int code = dtype_code(dt);
  Tensor wf, wd;
  if (want_wf) wf = at::empty({s, TAPS, padded(co, code, false), padded(ci, code, true)}, k5.options().dtype(dt));
  if (want_wd) wd = at::empty({s, TAPS, padded(ci, code, false), padded(co, code, true)}, k5.options().dtype(dt));
  RM_CALL(repmode_gatrep_fwd, k5.data_ptr<float>(), k3.data_ptr<float>(), k1.data_ptr<float>(), a3.data_ptr<float>(),
          a5.data_ptr<float>(), g.data_ptr<float>(), (int)s, (int)co, (int)ci, code, want_wf ? wf.data_ptr() : nullptr,
          want_wd ? wd.data_ptr() : nullptr, stream_handle());
  return {wf, wd};
}

// gate softmax + GatRep in one launch (repmode_gatrep_fwd_gate): (g [S,5,Co], wf, wd or undefined)
std::tuple<Tensor, Tensor, Tensor> gatrep_merge_gate(const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3, const Tensor& a5,
                                                      const Tensor& gw, const Tensor& gb, const Tensor& slot_task, int64_t nslots,
                                                      int64_t num_tasks, at::ScalarType dt, bool want_wd) {
  const int64_t co = k5.size(0), ci = k5.size(1);
  const int code = dtype_code(dt);
  Tensor g = at::empty({nslots, E, co}, k5.options());
  Tensor wf = at::empty({nslots, TAPS, padded(co, code, false), padded(ci, code, true)}, k5.options().dtype(dt));
  Tensor wd;
  if (want_wd) wd = at::empty({nslots, TAPS, padded(ci, code, false), padded(co, code, true)}, k5.options().dtype(dt));
  RM_CALL(repmode_gatrep_fwd_gate, k5.data_ptr<float>(), k3.data_ptr<float>(), k1.data_ptr<float>(), a3.data_ptr<float>(),
          a5.data_ptr<float>(), gw.data_ptr<float>(), gb.data_ptr<float>(), slot_task.data_ptr<int32_t>(), (int)nslots, (int)num_tasks,
          (int)co, (int)ci, code, g.data_ptr<float>(), wf.data_ptr(), want_wd ? wd.data_ptr() : nullptr, stream_handle());
  return std::make_tuple(g, wf, wd);
}

Tensor expert_selector(int64_t co, const Tensor& like) {
  // g for two pseudo-slots that select the raw experts: slot 0 = conv5x5, slot 1 = zero-padded conv3x3
  Tensor g = at::zeros({2, E, co}, like.options().dtype(at::kFloat));
  g[0][0].fill_(1.0);
  g[1][1].fill_(1.0);
  return g;
}

// The raw 5^3 / 3^3 experts as two un-merged slots of the conv kernels' layouts (slot 1 = K3, valid for centre3
// convolutions only).  bf16: one layout kernel per role; float32 (parity mode): GatRep with one-hot gates.
std::pair<Tensor, Tensor> expert_frags(const Tensor& k5, const Tensor& k3, at::ScalarType dt, bool want_wd) {
  const int64_t co = k5.size(0), ci = k5.size(1);
  if (dt != at::kBFloat16) {
    Tensor z = at::zeros({co, ci}, k5.options());
    return gatrep_merge(k5, k3, z, z, z, expert_selector(co, k5), dt, true, want_wd);
  }
  const int code = REPMODE_BF16;
  Tensor wf = at::empty({2, TAPS, padded(co, code, false), padded(ci, code, true)}, k5.options().dtype(dt));
  Tensor wd;
  if (want_wd) wd = at::empty({2, TAPS, padded(ci, code, false), padded(co, code, true)}, k5.options().dtype(dt));
  RM_CALL(repmode_expert_frags, k5.data_ptr<float>(), k3.data_ptr<float>(), (int)co, (int)ci, wf.data_ptr(),
          want_wd ? wd.data_ptr() : nullptr, stream_handle());
  return {wf, wd};
}

// What the forward conv's epilogue takes over from the BatchNorm3d + ReLU behind it (RepMode.py:146-149, 212).
struct Epi {
  bool stats = false;   // training: accumulate the BatchNorm batch statistics of the stored outputs (no separate read pass)
  Tensor scale, bias;   // eval: the BatchNorm folded -- scale into the gate probabilities (hence the merged filter),
  bool relu = false;    //       bias + ReLU into the epilogue
  bool any() const { return stats || bias.defined() || relu; }
};
// the BatchNorm-scratch half the last statistics-producing conv of this thread used (-1: none pending); consumed by the
// BatchNorm forward that follows it
thread_local int tl_stats_half = -1;

// Does a convolution of this shape write its element-typed output (whole reduction in one workgroup), or the float one whose
// reduction is split over workgroups?  The kernel library decides (repmode_conv5_elem_out): per layer and direction.
inline bool elem_out(int64_t n, int64_t d, int64_t h, int64_t w, int64_t cin, int64_t cout, at::ScalarType dt) {
  return dt == at::kBFloat16 && repmode_conv5_elem_out((int)n, (int)d, (int)h, (int)w, (int)cin, (int)cout, REPMODE_BF16) != 0;
}

// y[n] = x[n] (*) w[sample_slot[n]], 5^3 'same' cross-correlation, NDHWC -- RepMode.py:204-210
// dual (per-expert formulation): 0 = off; else the 5x5x5 and the 3x3x3 expert's convolutions in ONE launch (see
// repmode_conv5_ex): DUAL_OUT2 = x holds n samples, the output 2 n (the two expert outputs); DUAL_IN2 = x holds 2 n samples
// (the two gate-scaled output gradients), both jobs add into the n-sample output.
constexpr int DUAL_OUT2 = 8 | 32, DUAL_IN2 = 8 | 16;
Tensor conv5(const Tensor& x_cl, const Tensor& w, const Tensor& sample_slot, int64_t cout, bool out_f32, OptTensor out = c10::nullopt,
             bool centre3 = false, bool accumulate = false, bool dxc = false, const Epi* epi = nullptr, int dual = 0) {
  const int64_t n = dual == DUAL_IN2 ? x_cl.size(0) / 2 : x_cl.size(0);
  const int64_t d = x_cl.size(1), h = x_cl.size(2), wd_ = x_cl.size(3), cin = x_cl.size(4);
  const int code = dtype_code(x_cl.scalar_type());
  const at::ScalarType odt = (out_f32 || x_cl.scalar_type() == at::kFloat) ? at::kFloat : x_cl.scalar_type();
  Tensor y;
  if (out.has_value()) {
    y = *out;
  } else if (odt == at::kFloat && x_cl.scalar_type() == at::kBFloat16) {
    // float output = the kernel may split the reduction and add with atomics: a pre-zeroed pool tensor saves its memset
    auto tk = g_pool.take({dual == DUAL_OUT2 ? 2 * n : n, d, h, wd_, cout}, x_cl);
    y = tk.first;
    accumulate = accumulate || tk.second;
  } else {
    y = at::empty({dual == DUAL_OUT2 ? 2 * n : n, d, h, wd_, cout}, x_cl.options().dtype(odt));
  }
  TORCH_CHECK(y.scalar_type() == odt && y.is_contiguous(), "conv5: bad output tensor");
  TORCH_CHECK(!accumulate || odt == at::kFloat, "conv5: accumulation needs a float output");
  const int flags = (centre3 ? 1 : 0) | (accumulate ? 2 : 0) | (dxc ? 4 : 0) | dual;
  if (epi && epi->any()) {
    const bool stats = epi->stats && odt == at::kBFloat16;
    // (a pool tensor is pre-zeroed for atomics; a bias / ReLU epilogue overwrites instead: drop the accumulate flag)
    int half = -1;
    RM_CALL(repmode_conv5_epi, x_cl.data_ptr(), nullptr, 0, w.data_ptr(), sample_slot.data_ptr<int32_t>(), y.data_ptr(), (int)n, (int)d,
            (int)h, (int)wd_, (int)cin, (int)cout, code, odt == at::kFloat ? 1 : 0, (epi->bias.defined() || epi->relu) ? (flags & ~2) : flags,
            epi->bias.defined() ? epi->bias.data_ptr<float>() : nullptr, epi->relu ? 1 : 0, stats ? 1 : 0, &half, stream_handle());
    tl_stats_half = stats ? half : -1;
    return y;
  }
  RM_CALL(repmode_conv5_ex, x_cl.data_ptr(), w.data_ptr(), sample_slot.data_ptr<int32_t>(), y.data_ptr(), (int)n, (int)d, (int)h,
          (int)wd_, (int)cin, (int)cout, code, odt == at::kFloat ? 1 : 0, flags, stream_handle());
  return y;
}

Tensor thin_pack(const Tensor& w, bool to_rows) {
  const int64_t s_ = w.size(0), rp = w.size(2), kp = w.size(3);
  TORCH_CHECK((kp == 16 && !to_rows) || (rp == 32 && to_rows), "thin_pack: not a thin layer's filter");
  Tensor out = at::empty_like(w);     // only the 25 (dz, dy, dx=2) taps are written -- and read
  RM_CALL(repmode_thin_pack, w.data_ptr(), out.data_ptr(), (int)s_, (int)((rp / 32) * (kp / 16)), to_rows ? 1 : 0, stream_handle());
  return out;
}

Tensor shift5(const Tensor& t_cl) {
  const int64_t n = t_cl.size(0), d = t_cl.size(1), h = t_cl.size(2), w_ = t_cl.size(3);
  Tensor out = at::empty({n, d, h, w_, 8}, t_cl.options().dtype(at::kBFloat16));
  RM_CALL(repmode_shift5, t_cl.data_ptr(), dtype_code(t_cl.scalar_type()), out.data_ptr(), (long)(n * d * h), (int)w_, stream_handle());
  return out;
}

// The one-channel layers through their own kernels (csrc/thin_conv.hip); REPMODE_THIN=0 / set_thin_kernels(False): round 2's
// form (x taps folded into channels / rows around the general kernel's dx-centre mode, csrc/thin.hip)
bool g_thin = []() {
  const char* e = std::getenv("REPMODE_THIN");
  return e ? std::atoi(e) != 0 : true;
}();

// conv5 for a ONE-channel input (the first layer's forward, the last layer's data gradient)
Tensor thin_conv_in1(const Tensor& x_cl, const Tensor& w, const Tensor& sample_slot, int64_t cout, bool out_f32, const Epi* epi = nullptr) {
  if (g_thin && !(epi && epi->stats)) {
    Tensor x1 = x_cl.contiguous();
    const int64_t n = x1.size(0), d = x1.size(1), h = x1.size(2), w_ = x1.size(3);
    Tensor y = at::empty({n, d, h, w_, cout}, x1.options().dtype(out_f32 ? at::kFloat : at::kBFloat16));
    const bool bias = epi && epi->bias.defined();
    RM_CALL(repmode_conv5_thin_in1, x1.data_ptr(), w.data_ptr(), sample_slot.data_ptr<int32_t>(), y.data_ptr(), (int)n, (int)d, (int)h, (int)w_,
            (int)cout, out_f32 ? 1 : 0, bias ? epi->bias.data_ptr<float>() : nullptr, (epi && epi->relu) ? 1 : 0, stream_handle());
    tl_stats_half = -1;
    return y;
  }
  // the five x taps become channels, 25 instead of 125 taps (csrc/thin.hip)
  return conv5(shift5(x_cl.contiguous()), thin_pack(w, false), sample_slot, cout, out_f32, c10::nullopt, false, false, true, epi);
}

// conv5 for ONE output channel.  float [N,D,H,W,1]
Tensor thin_conv_out1(const Tensor& x_cl, const Tensor& w, const Tensor& sample_slot) {
  const int64_t n = x_cl.size(0), d = x_cl.size(1), h = x_cl.size(2), w_ = x_cl.size(3);
  if (g_thin) {
    Tensor y = at::empty({n, d, h, w_, 1}, x_cl.options().dtype(at::kFloat));
    RM_CALL(repmode_conv5_thin_out1, x_cl.data_ptr(), w.data_ptr(), sample_slot.data_ptr<int32_t>(), y.data_ptr<float>(), (int)n, (int)d, (int)h,
            (int)w_, (int)x_cl.size(4), stream_handle());
    return y;
  }
  // the five x taps become output rows, then a 5-tap diagonal sum (csrc/thin.hip)
  Tensor y5 = conv5(x_cl, thin_pack(w, true), sample_slot, 5, true, c10::nullopt, false, false, true);
  Tensor y = at::empty({n, d, h, w_, 1}, x_cl.options().dtype(at::kFloat));
  RM_CALL(repmode_unshift5, y5.data_ptr<float>(), y.data_ptr<float>(), (long)(n * d * h), (int)w_, stream_handle());
  return y;
}

// dw[s, tap, o, i] (float32) summed over the samples of each slot
Tensor conv5_wgrad(const Tensor& x_cl, const Tensor& dy_cl, const Tensor& sample_slot, int64_t nslots, int64_t cout, bool centre3 = false) {
  const int64_t n = x_cl.size(0), d = x_cl.size(1), h = x_cl.size(2), wd_ = x_cl.size(3), cin = x_cl.size(4);
  const bool thin = x_cl.scalar_type() == at::kBFloat16 && ((cin == 1) != (cout == 1)) && !centre3;
  // a launch that writes every element with plain stores needs no cleared buffer: it stays out of the step's pooled memset
  int direct = 0;
  // (not the centre3 job: it writes the dz in [1, 3] planes only -- 50 of the 125 taps per (co, ci) would be uninitialised memory
  // for any consumer that reads more than the centred taps; that job keeps a cleared buffer)
  if (!thin && !centre3 && x_cl.scalar_type() == at::kBFloat16)
    RM_CALL(repmode_conv5_wgrad_plan, (int)nslots, (int)n, (int)d, (int)h, (int)wd_, (int)cin, (int)cout, REPMODE_BF16, centre3 ? 1 : 0, &direct);
  if (direct) {
    Tensor dw = at::empty({nslots, TAPS, cout, cin}, x_cl.options().dtype(at::kFloat));
    RM_CALL(repmode_conv5_wgrad_ex, x_cl.data_ptr(), dy_cl.data_ptr(), sample_slot.data_ptr<int32_t>(), (int)nslots, dw.data_ptr<float>(),
            (int)n, (int)d, (int)h, (int)wd_, (int)cin, (int)cout, REPMODE_BF16, (centre3 ? 1 : 0) | 8, stream_handle());
    return dw;
  }
  auto tk = g_pool.take({nslots, TAPS, cout, cin}, x_cl);
  Tensor dw = tk.first;
  const bool pre = tk.second;
  if (thin) {
    // thin layer: taps stand in for the missing channel dimension (conv5_wgrad_thin)
    const bool first = cin == 1;
    const Tensor& a_t = first ? dy_cl : x_cl;
    const Tensor& b_t = first ? x_cl : dy_cl;
    RM_CALL(repmode_conv5_wgrad_thin, a_t.data_ptr(), b_t.data_ptr(), sample_slot.data_ptr<int32_t>(), (int)nslots,
            dw.data_ptr<float>(), (int)n, (int)d, (int)h, (int)wd_, (int)(first ? cout : cin), (first ? 0 : 1) | (pre ? 2 : 0),
            stream_handle());
    return dw;
  }
  RM_CALL(repmode_conv5_wgrad_ex, x_cl.data_ptr(), dy_cl.data_ptr(), sample_slot.data_ptr<int32_t>(), (int)nslots, dw.data_ptr<float>(),
          (int)n, (int)d, (int)h, (int)wd_, (int)cin, (int)cout, dtype_code(x_cl.scalar_type()), (centre3 ? 1 : 0) | (pre ? 8 : 0),
          stream_handle());
  return dw;
}

// single slot, gradient written directly in the experts' parameter layout: [Co, Ci, 5,5,5] or [Co, Ci, 3,3,3]
Tensor conv5_wgrad_expert_layout(const Tensor& x_cl, const Tensor& dy_cl, const Tensor& sample_slot, int64_t cout, int64_t k, Tensor out) {
  const int64_t n = x_cl.size(0), d = x_cl.size(1), h = x_cl.size(2), wd_ = x_cl.size(3), cin = x_cl.size(4);
  TORCH_CHECK(out.scalar_type() == at::kFloat && out.is_contiguous() && out.numel() == cout * cin * k * k * k, "wgrad: bad output");
  RM_CALL(repmode_conv5_wgrad_ex, x_cl.data_ptr(), dy_cl.data_ptr(), sample_slot.data_ptr<int32_t>(), 1, out.data_ptr<float>(), (int)n,
          (int)d, (int)h, (int)wd_, (int)cin, (int)cout, dtype_code(x_cl.scalar_type()), k == 5 ? 2 : 3, stream_handle());
  return out;
}

Tensor tap_transpose(const Tensor& dw_taps, at::IntArrayRef shape, Tensor out, bool defer = false) {
  const int64_t co = shape[0], ci = shape[1], k = shape[2];
  TORCH_CHECK(out.sizes() == shape && out.scalar_type() == at::kFloat && out.is_contiguous(), "tap_transpose: bad output");
  RM_CALL(repmode_tap_transpose_ex, dw_taps.data_ptr<float>(), out.data_ptr<float>(), (long)(co * ci), (int)(k * k * k),
          defer ? REPMODE_DEFER : 0, stream_handle());
  return out;
}

Tensor box_sum(const Tensor* in3, const Tensor* in5, const Tensor* add0, const Tensor* add1, OptTensor out, at::ScalarType out_dtype) {
  const Tensor& ref = in3 ? *in3 : *in5;
  Tensor o = out.has_value() ? *out : at::empty(ref.sizes(), ref.options().dtype(out_dtype));
  RM_CALL(repmode_box_sum_ex, in3 ? in3->data_ptr<float>() : nullptr, in5 ? in5->data_ptr<float>() : nullptr,
          add0 ? add0->data_ptr<float>() : nullptr, add1 ? add1->data_ptr<float>() : nullptr, o.data_ptr(), dtype_code(o.scalar_type()),
          (int)ref.size(0), (int)ref.size(1), (int)ref.size(2), (int)ref.size(3), (int)ref.size(4), stream_handle());
  return o;
}

// the three 1x1 experts' GEMMs in one launch (csrc/gemm3.hip): C_i[m][n] = sum_k A_i(m, k) B_i(n, k)
void gemm3(const Tensor a[3], int64_t a_ms, int64_t a_ks, const Tensor b[3], int64_t b_ns, int64_t b_ks, Tensor c[3], int64_t ldc,
           int64_t m, int64_t n, int64_t k, bool c_is_zero, bool bf16_mfma) {
  const float* ap[3] = {a[0].data_ptr<float>(), a[1].data_ptr<float>(), a[2].data_ptr<float>()};
  const float* bp[3] = {b[0].data_ptr<float>(), b[1].data_ptr<float>(), b[2].data_ptr<float>()};
  float* cp[3] = {c[0].data_ptr<float>(), c[1].data_ptr<float>(), c[2].data_ptr<float>()};
  RM_CALL(repmode_gemm3, ap, (long)a_ms, (long)a_ks, bp, (long)b_ns, (long)b_ks, cp, (int)ldc, (int)m, (int)n, (int)k, c_is_zero ? 1 : 0,
          bf16_mfma ? 1 : 0, stream_handle());
}

// [x | box3(x) | box5(x)] as float [3][n][d][h][w][c]: one launch when the volume fits in LDS (repmode_box_expand's rule)
Tensor box_expand(const Tensor& x_cl) {
  const int64_t n = x_cl.size(0), d = x_cl.size(1), h = x_cl.size(2), w = x_cl.size(3), c = x_cl.size(4);
  Tensor xb = at::empty({3, n, d, h, w, c}, x_cl.options().dtype(at::kFloat));
  const int64_t v = d * h * w;
  bool fits = false;
  if ((c & 3) == 0)
    for (int64_t t = 16; t >= 4; t >>= 1)
      if (v * t * 8 <= 60 * 1024 && v * (t / 4) <= 256 * 16) { fits = true; break; }
  if (fits) {
    RM_CALL(repmode_box_expand, x_cl.data_ptr(), dtype_code(x_cl.scalar_type()), xb.data_ptr<float>(), (int)n, (int)d, (int)h, (int)w, (int)c,
            stream_handle());
  } else {
    Tensor xb0 = xb[0], xb1 = xb[1], xb2 = xb[2];
    xb0.copy_(x_cl);
    box_sum(&xb0, nullptr, nullptr, nullptr, xb1, at::kFloat);
    box_sum(nullptr, &xb0, nullptr, nullptr, xb2, at::kFloat);
  }
  return xb;
}

// ------------------------------------------------------------------------------------------------------------
// eval-mode filter cache: inside [begin, end) the merged filter of an eval-mode block (one slot, RepMode.py:209-210)
// is computed once per (block, task, dtype): sliding-window inference re-uses it for every batch of patches.
struct EvalKey {
  void* k5;
  int64_t task;
  int dt;      // element type, + 64 when an eval-mode BatchNorm scale is folded into the filter
  bool operator==(const EvalKey& o) const { return k5 == o.k5 && task == o.task && dt == o.dt; }
};
struct EvalKeyHash {
  size_t operator()(const EvalKey& k) const { return std::hash<void*>()(k.k5) ^ (size_t)(k.task * 1315423911u) ^ (size_t)k.dt; }
};
std::mutex g_eval_mu;
int g_eval_depth = 0;
std::unordered_map<EvalKey, std::pair<Tensor, Tensor>, EvalKeyHash> g_eval;

struct Merged {
  Tensor g, wf, wd;
};

// Filters prepared ahead of the forward pass on the `prep` stream (op prepare_filters): they depend on the parameters and
// the batch's tasks only, not on activations, so all 19 blocks' gate softmax + GatRep (or expert layout) launches are
// issued when the step starts and run beside the first convolutions.  A block takes its entry (ordering its own stream
// behind the entry's event) or, when there is none that fits, computes the filters itself.
struct PrepEntry {
  Tensor g, wf, wd;
  hipEvent_t ev = nullptr;
  int64_t rows = 0;          // slots (merged) or samples (per-expert)
  at::ScalarType dt = at::kFloat;
  bool unmerged = false;
};
std::mutex g_prep_mu;
std::unordered_map<void*, PrepEntry> g_prep;
std::vector<hipEvent_t> g_prep_events;

bool take_prepared(const Tensor& k5, int64_t rows, at::ScalarType dt, bool unmerged, bool want_wd, Merged* out) {
  PrepEntry e;
  {
    std::lock_guard<std::mutex> lock(g_prep_mu);
    if (g_prep.empty()) return false;
    auto it = g_prep.find(k5.data_ptr());
    if (it == g_prep.end()) return false;
    e = it->second;
    g_prep.erase(it);
  }
  if (e.rows != rows || e.dt != dt || e.unmerged != unmerged || (want_wd && !e.wd.defined())) return false;
  if (e.ev) RM_HIP_CHECK(hipStreamWaitEvent(c10::hip::getCurrentHIPStream(k5.device().index()).stream(), e.ev, 0));
  out->g = e.g;
  out->wf = e.wf;
  out->wd = want_wd ? e.wd : Tensor();
  return true;
}
// fold_scale (eval only): per-output-channel factor gamma / sqrt(running_var + eps) of the BatchNorm behind the block; it
// multiplies the gate probabilities, which multiply the experts per output channel -- so the merged filter comes out
// scaled with no change to the GatRep kernel.
Merged merged_filters(const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3, const Tensor& a5, const Tensor& gw,
                      const Tensor& gb, const Plan& plan, at::ScalarType dt, bool want_wd, bool grad_enabled,
                      const Tensor* fold_scale = nullptr) {
  const bool cacheable = !(plan.training || want_wd || grad_enabled);
  EvalKey key{k5.data_ptr(), plan.task0, (int)dt + (fold_scale ? 64 : 0)};
  if (cacheable) {
    std::lock_guard<std::mutex> lock(g_eval_mu);
    if (g_eval_depth > 0) {
      auto it = g_eval.find(key);
      if (it != g_eval.end()) return {it->second.first, it->second.second, Tensor()};
    }
  }
  Merged m;
  // (prepared filters belong to the training forward pass that prepared them: an eval-mode block never takes one -- after a
  // training forward that threw half-way, a stale entry merged for the failed batch's task could otherwise end up in the
  // eval filter cache of a whole volume)
  if (plan.training && !fold_scale && take_prepared(k5, plan.nslots, dt, false, want_wd, &m)) return m;
  if (fold_scale) {
    m.g = gate_softmax(gw, gb, plan.slot_task, plan.nslots, plan.num_tasks, k5.size(0));
    m.g = m.g * fold_scale->view({1, 1, -1});
    auto w = gatrep_merge(k5, k3, k1, a3, a5, m.g, dt, true, want_wd);
    m.wf = w.first;
    m.wd = w.second;
  } else {
    std::tie(m.g, m.wf, m.wd) = gatrep_merge_gate(k5, k3, k1, a3, a5, gw, gb, plan.slot_task, plan.nslots, plan.num_tasks, dt, want_wd);
  }
  if (cacheable) {
    std::lock_guard<std::mutex> lock(g_eval_mu);
    if (g_eval_depth > 0) g_eval[key] = {m.g, m.wf};
  }
  return m;
}

// GatRep backward: per-slot filter gradient dw [S, 125, Co, Ci] -> (dk5, dk3, dk1, da3, da5, dgate_w, dgate_b)
std::vector<Tensor> filter_and_expert_grads(const Tensor& dw, const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3,
                                            const Tensor& a5, const Tensor& g, const Plan& plan, bool defer_gate = false) {
  const int64_t co = k5.size(0), ci = k5.size(1);
  Tensor dk5 = grad_out(k5), dk3 = grad_out(k3), dk1 = grad_out(k1), da3 = grad_out(a3), da5 = grad_out(a5);
  Tensor dgw = at::empty({E * co, plan.num_tasks}, k5.options());
  Tensor dgb = at::empty({E * co}, k5.options());
  Tensor dg_ws = at::empty_like(g);
  // defer_gate: the gate part rides in the caller's data-gradient conv launch (the library defers only when its
  // accumulator is its own scratch, so dg_ws may go out of scope here)
  RM_CALL(repmode_gatrep_bwd_ex, dw.data_ptr<float>(), k5.data_ptr<float>(), k3.data_ptr<float>(), k1.data_ptr<float>(),
          a3.data_ptr<float>(), a5.data_ptr<float>(), g.data_ptr<float>(), plan.slot_task.data_ptr<int32_t>(), (int)plan.nslots,
          (int)plan.num_tasks, (int)co, (int)ci, dk5.data_ptr<float>(), dk3.data_ptr<float>(), dk1.data_ptr<float>(),
          da3.data_ptr<float>(), da5.data_ptr<float>(), dgw.data_ptr<float>(), dgb.data_ptr<float>(), dg_ws.data_ptr<float>(),
          defer_gate ? REPMODE_DEFER : 0, stream_handle());
  return {dk5, dk3, dk1, da3, da5, dgw, dgb};
}

// ------------------------------------------------------------------------------------------------------------
// autograd nodes.  Input order of every MoDE function: activations first, then k5 k3 k1 a3 a5 gate_w gate_b.

// Fused gate-softmax + GatRep + per-slot 5^3 convolution (the "merged" formulation), forward and backward.
struct ModeConvMerged : public torch::autograd::Function<ModeConvMerged> {
  static Tensor forward(AutogradContext* ctx, Tensor x_cl, Tensor k5, Tensor k3, Tensor k1, Tensor a3, Tensor a5, Tensor gw, Tensor gb,
                        Plan plan, bool out_f32, bool grad_enabled, Epi epi) {
    const int64_t co = k5.size(0), ci = k5.size(1);
    const bool need_dx = grad_enabled && x_cl.requires_grad();
    // the data-gradient filter comes out of the same pass over the experts (one launch, one read of the weights)
    Merged m = merged_filters(k5, k3, k1, a3, a5, gw, gb, plan, x_cl.scalar_type(), need_dx, grad_enabled,
                              epi.scale.defined() ? &epi.scale : nullptr);
    const bool thin = x_cl.scalar_type() == at::kBFloat16 && ((ci == 1) != (co == 1));
    Tensor y;
    tl_stats_half = -1;
    if (thin && ci == 1) {                                 // first layer: x taps folded into input channels
      y = thin_conv_in1(x_cl, m.wf, plan.sample_slot, co, out_f32, &epi);
    } else if (thin) {                                     // last layer: x taps folded into output rows
      TORCH_CHECK(!epi.any(), "the one-output-channel layer has no BatchNorm to take over");
      y = thin_conv_out1(x_cl, m.wf, plan.sample_slot);
      if (!out_f32) y = y.to(x_cl.scalar_type());
    } else {
      y = conv5(x_cl, m.wf, plan.sample_slot, co, out_f32, c10::nullopt, false, false, false, &epi);
    }
    ctx->save_for_backward({x_cl, k5, k3, k1, a3, a5, m.g, m.wd.defined() ? m.wd : Tensor()});
    ctx->saved_data["slot_task"] = plan.slot_task;
    ctx->saved_data["sample_slot"] = plan.sample_slot;
    ctx->saved_data["nslots"] = plan.nslots;
    ctx->saved_data["num_tasks"] = plan.num_tasks;
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    RM_CALL(repmode_tail_discard, stream_handle());      // (jobs a failed call left queued must not ride in this node's convs)
    auto sv = ctx->get_saved_variables();
    const Tensor &x_cl = sv[0], &k5 = sv[1], &k3 = sv[2], &k1 = sv[3], &a3 = sv[4], &a5 = sv[5], &g = sv[6];
    Tensor wd = sv[7];
    Plan plan;
    plan.slot_task = ctx->saved_data["slot_task"].toTensor();
    plan.sample_slot = ctx->saved_data["sample_slot"].toTensor();
    plan.nslots = ctx->saved_data["nslots"].toInt();
    plan.num_tasks = ctx->saved_data["num_tasks"].toInt();
    const int64_t co = k5.size(0), ci = k5.size(1);
    const at::ScalarType dt = x_cl.scalar_type();
    Tensor dy = grads[0].to(dt).contiguous();
    const bool need_dx = ctx->needs_input_grad(0) && wd.defined();
    // Neither the filter gradient nor the GatRep backward depends on the data gradient.  Deep levels (a launch does not
    // fill the chip): both on the second stream.  Elsewhere: the filter gradient stays in line with the data gradient
    // (both MFMA-bound), only the HBM-bound GatRep backward goes beside the data-gradient conv.
    const bool whole = need_dx && forks(x_cl);
    Tensor dw;
    if (!whole) dw = conv5_wgrad(x_cl, dy, plan.sample_slot, plan.nslots, co);
    Fork fork(need_dx && (whole || g_overlap), x_cl);
    fork.to_side();
    if (whole) dw = conv5_wgrad(x_cl, dy, plan.sample_slot, plan.nslots, co);
    // (one stream: the data-gradient conv below hosts the job -- unless it is one of the one-channel layers' own kernels,
    // which host nothing: the job is then launched on its own at once instead of queued and flushed)
    const bool thin_dx = g_thin && dt == at::kBFloat16 && ((ci == 1) != (co == 1));
    const bool defer = g_tail && need_dx && !(whole || g_overlap) && !thin_dx;
    std::vector<Tensor> pg = filter_and_expert_grads(dw, k5, k3, k1, a3, a5, g, plan, defer);
    fork.to_main();
    Tensor dx;
    if (need_dx) {
      // deep levels (small volumes) split the channel reduction over workgroups -> float output
      const bool f32 = !elem_out(x_cl.size(0), x_cl.size(1), x_cl.size(2), x_cl.size(3), co, ci, dt);
      if (dt == at::kBFloat16 && co == 1 && ci != 1) dx = thin_conv_in1(dy, wd, plan.sample_slot, ci, f32);   // last layer: dy has one channel
      else if (dt == at::kBFloat16 && ci == 1 && co != 1) dx = thin_conv_out1(dy, wd, plan.sample_slot);
      else dx = conv5(dy, wd, plan.sample_slot, ci, f32);
      if (defer) RM_CALL(repmode_tail_flush, stream_handle());      // (nothing left unless the conv above took a path that hosts no jobs)
      if (dx.scalar_type() != dt) dx = dx.to(dt);
    }
    fork.join();
    return {dx, pg[0], pg[1], pg[2], pg[3], pg[4], pg[5], pg[6], Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// The same for a skip connection: the block's input is the channel concatenation of two tensors (RepMode.py:106
// torch.cat((x_skip, up), 1)), which is never materialised (repmode_conv5_pair, repmode_conv5_wgrad_part).
struct ModeConvPair : public torch::autograd::Function<ModeConvPair> {
  static Tensor forward(AutogradContext* ctx, Tensor xa, Tensor xb, Tensor k5, Tensor k3, Tensor k1, Tensor a3, Tensor a5, Tensor gw,
                        Tensor gb, Plan plan, bool out_f32, bool grad_enabled, Epi epi) {
    const int64_t co = k5.size(0), ci = k5.size(1), ca = xa.size(4);
    const int64_t n = xa.size(0), d = xa.size(1), h = xa.size(2), w_ = xa.size(3);
    const bool need_dx = grad_enabled && (xa.requires_grad() || xb.requires_grad());
    Merged m = merged_filters(k5, k3, k1, a3, a5, gw, gb, plan, xa.scalar_type(), need_dx, grad_enabled,
                              epi.scale.defined() ? &epi.scale : nullptr);
    tl_stats_half = -1;
    const int code = dtype_code(xa.scalar_type());
    const at::ScalarType odt = (out_f32 || xa.scalar_type() == at::kFloat) ? at::kFloat : xa.scalar_type();
    int flags = 0;
    Tensor y;
    if (odt == at::kFloat && xa.scalar_type() == at::kBFloat16) {
      auto tk = g_pool.take({n, d, h, w_, co}, xa);
      y = tk.first;
      flags = tk.second ? 2 : 0;
    } else {
      y = at::empty({n, d, h, w_, co}, xa.options().dtype(odt));
    }
    if (epi.any()) {
      const bool stats = epi.stats && odt == at::kBFloat16;
      int half = -1;
      RM_CALL(repmode_conv5_epi, xa.data_ptr(), xb.data_ptr(), (int)ca, m.wf.data_ptr(), plan.sample_slot.data_ptr<int32_t>(), y.data_ptr(),
              (int)n, (int)d, (int)h, (int)w_, (int)ci, (int)co, code, odt == at::kFloat ? 1 : 0, (epi.bias.defined() || epi.relu) ? 0 : flags,
              epi.bias.defined() ? epi.bias.data_ptr<float>() : nullptr, epi.relu ? 1 : 0, stats ? 1 : 0, &half, stream_handle());
      tl_stats_half = stats ? half : -1;
    } else {
      RM_CALL(repmode_conv5_pair, xa.data_ptr(), xb.data_ptr(), (int)ca, m.wf.data_ptr(), plan.sample_slot.data_ptr<int32_t>(), y.data_ptr(),
              nullptr, 0, (int)n, (int)d, (int)h, (int)w_, (int)ci, (int)co, code, odt == at::kFloat ? 1 : 0, flags, stream_handle());
    }
    ctx->save_for_backward({xa, xb, k5, k3, k1, a3, a5, m.g, m.wd.defined() ? m.wd : Tensor()});
    ctx->saved_data["slot_task"] = plan.slot_task;
    ctx->saved_data["sample_slot"] = plan.sample_slot;
    ctx->saved_data["nslots"] = plan.nslots;
    ctx->saved_data["num_tasks"] = plan.num_tasks;
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    RM_CALL(repmode_tail_discard, stream_handle());      // (jobs a failed call left queued must not ride in this node's convs)
    auto sv = ctx->get_saved_variables();
    const Tensor &xa = sv[0], &xb = sv[1], &k5 = sv[2], &k3 = sv[3], &k1 = sv[4], &a3 = sv[5], &a5 = sv[6], &g = sv[7];
    Tensor wd = sv[8];
    Plan plan;
    plan.slot_task = ctx->saved_data["slot_task"].toTensor();
    plan.sample_slot = ctx->saved_data["sample_slot"].toTensor();
    plan.nslots = ctx->saved_data["nslots"].toInt();
    plan.num_tasks = ctx->saved_data["num_tasks"].toInt();
    const int64_t co = k5.size(0), ci = k5.size(1), ca = xa.size(4), cb = xb.size(4);
    const int64_t n = xa.size(0), d = xa.size(1), h = xa.size(2), w_ = xa.size(3);
    const at::ScalarType dt = xa.scalar_type();
    const int code = dtype_code(dt);
    Tensor dy = grads[0].to(dt).contiguous();
    const bool whole = wd.defined() && forks(xa);
    Tensor dw;
    auto filter_grad = [&]() {
      // the two channel ranges of one buffer: cleared (pooled memset), unless both launches write their range with plain stores
      int da = 0, db = 0;
      if (dt == at::kBFloat16) {
        RM_CALL(repmode_conv5_wgrad_plan, (int)plan.nslots, (int)n, (int)d, (int)h, (int)w_, (int)ca, (int)co, REPMODE_BF16, 0, &da);
        RM_CALL(repmode_conv5_wgrad_plan, (int)plan.nslots, (int)n, (int)d, (int)h, (int)w_, (int)cb, (int)co, REPMODE_BF16, 0, &db);
      }
      if (da && db) {
        dw = at::empty({plan.nslots, TAPS, co, ci}, xa.options().dtype(at::kFloat));
      } else {
        auto tk = g_pool.take({plan.nslots, TAPS, co, ci}, xa);
        dw = tk.first;
        if (!tk.second) dw.zero_();
      }
      const Tensor* parts[2] = {&xa, &xb};
      const int64_t offs[2] = {0, ca};
      for (int i = 0; i < 2; ++i)
        RM_CALL(repmode_conv5_wgrad_part, parts[i]->data_ptr(), dy.data_ptr(), plan.sample_slot.data_ptr<int32_t>(), (int)plan.nslots,
                dw.data_ptr<float>(), (int)n, (int)d, (int)h, (int)w_, (int)parts[i]->size(4), (int)ci, (int)offs[i], (int)co, code, 8,
                stream_handle());
    };
    if (!whole) filter_grad();
    Fork fork(wd.defined() && (whole || g_overlap), xa);      // (as ModeConvMerged::backward)
    fork.to_side();
    if (whole) filter_grad();
    const bool defer = g_tail && wd.defined() && !(whole || g_overlap);
    std::vector<Tensor> pg = filter_and_expert_grads(dw, k5, k3, k1, a3, a5, g, plan, defer);
    fork.to_main();
    Tensor dxa, dxb;
    if (wd.defined()) {
      const bool f32 = dt == at::kFloat || !elem_out(n, d, h, w_, co, ci, dt);   // deep levels: split reduction -> float output
      int flags = 0;
      if (f32 && dt == at::kBFloat16) {
        auto ta = g_pool.take({n, d, h, w_, ca}, xa);
        auto tb = g_pool.take({n, d, h, w_, cb}, xa);
        dxa = ta.first;
        dxb = tb.first;
        if (ta.second && tb.second) {
          flags = 2;
        } else if (ta.second || tb.second) {      // (cannot happen with a consistent pool; stay correct anyway)
          dxa.zero_();
          dxb.zero_();
          flags = 2;
        }
      } else {
        const at::ScalarType odt = f32 ? at::kFloat : dt;
        dxa = at::empty({n, d, h, w_, ca}, xa.options().dtype(odt));
        dxb = at::empty({n, d, h, w_, cb}, xa.options().dtype(odt));
      }
      RM_CALL(repmode_conv5_pair, dy.data_ptr(), nullptr, 0, wd.data_ptr(), plan.sample_slot.data_ptr<int32_t>(), dxa.data_ptr(),
              dxb.data_ptr(), (int)ca, (int)n, (int)d, (int)h, (int)w_, (int)co, (int)ci, code, f32 ? 1 : 0, flags, stream_handle());
      if (defer) RM_CALL(repmode_tail_flush, stream_handle());
      if (dxa.scalar_type() != dt) {
        dxa = dxa.to(dt);
        dxb = dxb.to(dt);
      }
    }
    fork.join();
    return {dxa, dxb, pg[0], pg[1], pg[2], pg[3], pg[4], pg[5], pg[6], Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// constant index vectors of the slot-less per-expert convolutions (slot 0 = K5, slot 1 = K3), cached per (n, device)
Tensor single_slot(int64_t n, int64_t slot, const Tensor& like) {
  static std::mutex mu;
  static std::unordered_map<int64_t, Tensor> cache;
  const int64_t key = (n * 4 + slot) * 64 + like.device().index();
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it == cache.end()) it = cache.emplace(key, at::full({n}, slot, like.options().dtype(at::kInt))).first;
  return it->second;
}

// The same MoDE block by linearity of the convolution (property 3):
//     y[n] = sum_e g[n, e, :] * conv(x[n], K_e)
// The experts are shared by all samples, so nothing is merged per task: the 5^3 and 3^3 experts are laid out once as
// two pseudo-slots and the HIP conv kernels run them for the whole batch; the three 1x1 experts (conv1x1, avg3, avg5)
// are GEMMs on x and its box means.  Used on the deep levels, where the weights (84 % of the parameters) dwarf the
// activations and per-task merged filters / filter gradients are pure HBM traffic.
struct ModeConvUnmerged : public torch::autograd::Function<ModeConvUnmerged> {
  static Tensor forward(AutogradContext* ctx, Tensor x_cl, Tensor k5, Tensor k3, Tensor k1, Tensor a3, Tensor a5, Tensor gw, Tensor gb,
                        Plan plan, bool grad_enabled, bool want_stats) {
    const int64_t co = k5.size(0), ci = k5.size(1);
    const int64_t n = x_cl.size(0), d = x_cl.size(1), h = x_cl.size(2), w = x_cl.size(3);
    const bool need_dx = grad_enabled && x_cl.requires_grad();
    tl_stats_half = -1;
    Tensor gn;                                                                            // g per SAMPLE [N, 5, Co]
    std::pair<Tensor, Tensor> fr;
    Merged pm;
    if (take_prepared(k5, plan.n, x_cl.scalar_type(), true, need_dx, &pm)) {
      gn = pm.g.defined() ? pm.g : gate_softmax(gw, gb, plan.sample_task, plan.n, plan.num_tasks, co);
      fr = {pm.wf, pm.wd};
    } else {
      gn = gate_softmax(gw, gb, plan.sample_task, plan.n, plan.num_tasks, co);
      fr = expert_frags(k5, k3, x_cl.scalar_type(), need_dx);
    }
    const int dm = ((g_deep_mode & 1) && x_cl.scalar_type() == at::kBFloat16)
                       ? repmode_deep_mode_plan(0, (int)n, (int)d, (int)h, (int)w, (int)ci, (int)co, REPMODE_BF16) : 0;
    if (dm) {
      // ONE launch: the five experts' convolutions, the gate mix and the stores of P_e (csrc/deep_mode.hip); the avg experts'
      // operands come from the box kernel.  dm > 1: that many workgroups add into each output element (level 4): zeroed
      // float outputs out of the step's pooled memset
      Tensor xb = box_expand(x_cl);
      Tensor p, y;
      if (dm == 1) {
        p = at::empty({E, n, d, h, w, co}, x_cl.options().dtype(at::kFloat));
        y = at::empty({n, d, h, w, co}, x_cl.options().dtype(at::kFloat));
      } else {
        auto tp = g_pool.take({E, n, d, h, w, co}, x_cl);
        auto ty = g_pool.take({n, d, h, w, co}, x_cl);
        p = tp.first;
        y = ty.first;
        if (!tp.second) p.zero_();
        if (!ty.second) y.zero_();
      }
      int half = -1;
      RM_CALL(repmode_deep_mode_fwd_ex, x_cl.data_ptr(), fr.first.data_ptr(), xb.data_ptr<float>(), k1.data_ptr<float>(), a3.data_ptr<float>(),
              a5.data_ptr<float>(), gn.data_ptr<float>(), p.data_ptr<float>(), y.data_ptr<float>(), (int)n, (int)d, (int)h, (int)w, (int)ci,
              (int)co, want_stats ? 1 : 0, &half, stream_handle());
      tl_stats_half = half;          // (the BatchNorm behind the block skips its statistics pass: op_mode_block)
      ctx->save_for_backward({x_cl, k5, k3, k1, a3, a5, gn, xb, p, fr.second.defined() ? fr.second : Tensor()});
      ctx->saved_data["sample_task"] = plan.sample_task;
      ctx->saved_data["num_tasks"] = plan.num_tasks;
      return y;
    }
    Tensor s0 = single_slot(n, 0, x_cl), s1 = single_slot(n, 1, x_cl);
    auto tk = g_pool.take({E, n, d, h, w, co}, x_cl);                                     // expert outputs P_e
    Tensor p = tk.first;
    const bool pre = tk.second;
    Tensor xb;
    // the 5^3 (+ 3^3) expert's conv on this stream, the small experts beside it on the second one
    Fork fork(forks(x_cl), x_cl);
    fork.to_side();
    {
      // the 3^3 expert (unless it shares the 5^3 expert's launch), and the three 1x1 experts as ONE batched GEMM launch:
      // [x | box3(x) | box5(x)] @ [K1 | A3 | A5]^T  -> P_2..P_4
      if (!g_dual_launch) conv5(x_cl, fr.first, s1, co, true, p[1], true, pre);
      xb = box_expand(x_cl);
      const Tensor am[3] = {xb[0], xb[1], xb[2]}, bm[3] = {k1, a3, a5};
      Tensor cm[3] = {p[2], p[3], p[4]};
      gemm3(am, ci, 1, bm, ci, 1, cm, co, n * d * h * w, co, ci, pre, x_cl.scalar_type() == at::kBFloat16);      // (P lives in the step's pre-zeroed pool tensor)
    }
    fork.to_main();
    const bool deep = g_deep && g_dual_launch && repmode_conv5_deep_supported((int)w, (int)ci, dtype_code(x_cl.scalar_type())) != 0 &&
                      (g_deep_fwd_min == 0 || (w > 4 && n * ci >= g_deep_fwd_min));
    if (deep) {
      Tensor p2 = p.narrow(0, 0, 2);
      RM_CALL(repmode_conv5_deep, x_cl.data_ptr(), fr.first.data_ptr(), p2.data_ptr<float>(), (int)n, (int)d, (int)h, (int)w, (int)ci, (int)co,
              pre ? 2 : 0, stream_handle());
    } else if (g_dual_launch) {
      conv5(x_cl, fr.first, s0, co, true, p.narrow(0, 0, 2).view({2 * n, d, h, w, co}), false, pre, false, nullptr, DUAL_OUT2);
    } else {
      conv5(x_cl, fr.first, s0, co, true, p[0], false, pre);
    }
    fork.join();
    Tensor y = at::empty({n, d, h, w, co}, x_cl.options().dtype(at::kFloat));
    RM_CALL(repmode_expert_mix_fwd, p.data_ptr<float>(), gn.data_ptr<float>(), y.data_ptr<float>(), (int)n, (long)(d * h * w), (int)co,
            stream_handle());
    ctx->save_for_backward({x_cl, k5, k3, k1, a3, a5, gn, xb, p, fr.second.defined() ? fr.second : Tensor()});
    ctx->saved_data["sample_task"] = plan.sample_task;
    ctx->saved_data["num_tasks"] = plan.num_tasks;
    return y;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    RM_CALL(repmode_tail_discard, stream_handle());      // (jobs a failed call left queued must not ride in this node's convs)
    auto sv = ctx->get_saved_variables();
    const Tensor &x_cl = sv[0], &k5 = sv[1], &k3 = sv[2], &k1 = sv[3], &a3 = sv[4], &a5 = sv[5], &gn = sv[6], &xb = sv[7], &p = sv[8];
    Tensor wd2 = sv[9];
    const Tensor sample_task = ctx->saved_data["sample_task"].toTensor();
    const int64_t num_tasks = ctx->saved_data["num_tasks"].toInt();
    const int64_t co = k5.size(0), ci = k5.size(1);
    const int64_t n = x_cl.size(0), d = x_cl.size(1), h = x_cl.size(2), w = x_cl.size(3);
    const at::ScalarType dt = x_cl.scalar_type();
    Tensor dy = grads[0].to(at::kFloat).contiguous();
    // ---- gate: dg[n,e,o] = <dy, P_e>, softmax Jacobian, Linear grads (RepMode.py:198-200); and the gate-scaled dy per
    // expert: the two conv experts' in the element type (operands of the conv kernels), the three 1x1 experts' in float
    const int64_t m = n * d * h * w;
    const int64_t pad = 0;
    auto tdg = g_pool.take({n, E, co}, x_cl);
    Tensor dg = tdg.first;
    Tensor lo = at::empty({2, n, d, h, w, co}, x_cl.options().dtype(dt));
    Tensor hi = at::empty({3, m, co}, x_cl.options().dtype(at::kFloat));
    // one-launch data gradient ahead (deep_mode.hip): the avg-pool experts' box-mean operands come out of the SAME launch as the
    // gate mix's backward (both read dy and nothing of each other; bit 3 of REPMODE_DEEP_MODE off: box_pair as its own launch)
    const bool need_dx_early = ctx->needs_input_grad(0) && wd2.defined();
    const int dmb = (need_dx_early && (g_deep_mode & 2) && dt == at::kBFloat16)
                        ? repmode_deep_mode_plan(1, (int)n, (int)d, (int)h, (int)w, (int)ci, (int)co, REPMODE_BF16) : 0;
    Tensor hb;
    bool hb_done = false;
    if (dmb) {
      hb = at::empty({2, m, co}, x_cl.options().dtype(at::kFloat));
      if ((g_deep_mode & 8) && (co & 3) == 0 && d * h * w * 16 * 16 <= 64 * 1024) {
        RM_CALL(repmode_expert_mix_bwd_box, dy.data_ptr<float>(), p.data_ptr<float>(), gn.data_ptr<float>(), dg.data_ptr<float>(), lo.data_ptr(),
                hi.data_ptr<float>(), (long)((m + pad) * co), hb[0].data_ptr<float>(), hb[1].data_ptr<float>(), (int)n, (int)d, (int)h, (int)w,
                (int)co, dtype_code(dt) | (tdg.second ? 16 : 0), stream_handle());
        hb_done = true;
      }
    }
    if (!hb_done)
      RM_CALL(repmode_expert_mix_bwd_ex, dy.data_ptr<float>(), p.data_ptr<float>(), gn.data_ptr<float>(), dg.data_ptr<float>(), lo.data_ptr(),
              hi.data_ptr<float>(), (long)((m + pad) * co), (int)n, (long)(d * h * w), (int)co, dtype_code(dt) | (tdg.second ? 16 : 0),
              stream_handle());
    Tensor dgw = at::empty({E * co, num_tasks}, k5.options());
    Tensor dgb = at::empty({E * co}, k5.options());
    Tensor s0 = single_slot(n, 0, x_cl), s1 = single_slot(n, 1, x_cl);
    const bool need_dx = ctx->needs_input_grad(0) && wd2.defined();
    // the gate backward and the two layout transposes below ride in the data-gradient conv's launch (one stream only)
    const bool defer = g_tail && need_dx && !forks(x_cl) && g_dual_launch;
    RM_CALL(repmode_gate_bwd_ex, gn.data_ptr<float>(), dg.data_ptr<float>(), sample_task.data_ptr<int32_t>(), (int)n, (int)num_tasks, (int)co,
            dgw.data_ptr<float>(), dgb.data_ptr<float>(), defer ? REPMODE_DEFER : 0, stream_handle());
    // the expert gradients do not depend on the data gradient: second stream
    Fork fork(need_dx && forks(x_cl), x_cl);
    fork.to_side();
    Tensor dk5, dk3, dk1, da3, da5, keep5, keep3;
    {
      // filter gradients of the gate-scaled dy, all samples in one slot.  Large layers (every workgroup owns its outputs: no
      // atomics) write the parameters' [Co][Ci][taps] layout directly; the others accumulate tap-major + one transpose launch.
      const int64_t tiles = ((co + 31) / 32) * ((ci + 31) / 32);
      Tensor lo0 = lo[0], lo1 = lo[1];
      bool direct5 = dt == at::kBFloat16 && tiles * 5 >= 512, direct3 = dt == at::kBFloat16 && tiles * 3 >= 512;
      // In the dual launch the two jobs fill the grid TOGETHER (tiles x (5 + 3) units): from REPMODE_WGRAD_DIRECT_UNITS units
      // (default 256 = one per CU) both write the experts' layout -- no tap-major accumulators in the step's pooled memset
      // (level 3 at batch 8: 57 M floats of its 131 M), no transposition launches, and the wave-specialised form applies.
      static const int64_t direct_units = []() { const char* e = getenv("REPMODE_WGRAD_DIRECT_UNITS"); return e ? (int64_t)atol(e) : (int64_t)256; }();
      if (dt == at::kBFloat16 && g_dual_launch && g_dual_wgrad && tiles * 8 >= direct_units) direct5 = direct3 = true;
      if (dt == at::kBFloat16 && g_dual_launch && g_dual_wgrad) {
        // both experts' filter gradients from ONE launch (the 3x3x3 job alone is 27 taps on three planes: latency)
        dk5 = grad_out(k5);
        dk3 = grad_out(k3);
        Tensor t5, t3;                       // tap-major accumulators of the jobs that cannot store directly
        bool pre = true;
        if (!direct5) { auto tk = g_pool.take({1, TAPS, co, ci}, x_cl); t5 = tk.first; pre = pre && tk.second; }
        if (!direct3) { auto tk = g_pool.take({1, TAPS, co, ci}, x_cl); t3 = tk.first; pre = pre && tk.second; }
        RM_CALL(repmode_conv5_wgrad_dual, x_cl.data_ptr(), lo0.data_ptr(), lo1.data_ptr(), direct5 ? dk5.data_ptr<float>() : t5.data_ptr<float>(),
                direct3 ? dk3.data_ptr<float>() : t3.data_ptr<float>(), (int)n, (int)d, (int)h, (int)w, (int)ci, (int)co,
                (direct5 ? 2 : 0) | (pre ? 8 : 0), (direct3 ? 3 : 1) | (pre ? 8 : 0), stream_handle());
        if (!direct5) tap_transpose(t5[0], k5.sizes(), dk5, defer);
        if (!direct3) tap_transpose(t3[0], k3.sizes(), dk3, defer);
        keep5 = t5;             // (a deferred job reads them from inside the data-gradient conv's launch)
        keep3 = t3;
      } else {
        if (direct5) dk5 = conv5_wgrad_expert_layout(x_cl, lo0, s0, co, 5, grad_out(k5));
        else dk5 = tap_transpose(conv5_wgrad(x_cl, lo0, s0, 1, co)[0], k5.sizes(), grad_out(k5));
        if (direct3) dk3 = conv5_wgrad_expert_layout(x_cl, lo1, s0, co, 3, grad_out(k3));
        else dk3 = tap_transpose(conv5_wgrad(x_cl, lo1, s0, 1, co, true)[0], k3.sizes(), grad_out(k3));
      }
      // the 1x1 experts' filter gradients dW_e[co][ci] = sum_m G_e[m][co] X_e[m][ci], straight into the parameters' gradients
      // (K = all voxel rows: split over workgroups when the three outputs can come pre-zeroed out of the step's pool --
      // not when they are slices of a data-parallel reducer's bucket)
      bool zeroed = true;
      Tensor* outs[3] = {&dk1, &da3, &da5};
      const Tensor* prm[3] = {&k1, &a3, &a5};
      for (int e = 0; e < 3; ++e) {
        bool from_sink = false;
        if (grad_sink_active()) *outs[e] = grad_out(*prm[e], &from_sink);
        if (from_sink) {
          zeroed = false;
        } else {
          auto tk = g_pool.take(prm[e]->sizes(), x_cl);
          *outs[e] = tk.first;
          zeroed = zeroed && tk.second;
        }
      }
      const Tensor am[3] = {hi[0], hi[1], hi[2]}, bm[3] = {xb[0], xb[1], xb[2]};
      Tensor cm[3] = {dk1, da3, da5};
      gemm3(am, 1, co, bm, 1, ci, cm, ci, co, ci, m, zeroed, dt == at::kBFloat16);
    }
    fork.to_main();
    Tensor dx;
    if (dmb) {
      // ONE launch: all five experts' data gradients into one accumulator (csrc/deep_mode.hip).  The avg experts' parts go
      // through the box means of their gate-scaled output gradients (the box mean commutes with the 1x1 channel mixing).
      if (!hb_done)
        RM_CALL(repmode_box_pair, hi[1].data_ptr<float>(), hi[2].data_ptr<float>(), hb[0].data_ptr<float>(), hb[1].data_ptr<float>(), (int)n,
                (int)d, (int)h, (int)w, (int)co, stream_handle());
      Tensor dxo;
      if (dmb == 1) {
        dxo = at::empty({n, d, h, w, ci}, x_cl.options().dtype(dt));
      } else {
        auto tk = g_pool.take({n, d, h, w, ci}, x_cl);
        dxo = tk.first;
        if (!tk.second) dxo.zero_();
      }
      RM_CALL(repmode_deep_mode_dgrad, lo.data_ptr(), wd2.data_ptr(), hi[0].data_ptr<float>(), hb[0].data_ptr<float>(), hb[1].data_ptr<float>(),
              k1.data_ptr<float>(), a3.data_ptr<float>(), a5.data_ptr<float>(), dxo.data_ptr(), dtype_code(dxo.scalar_type()), (int)n, (int)d,
              (int)h, (int)w, (int)ci, (int)co, stream_handle());
      if (defer) RM_CALL(repmode_tail_flush, stream_handle());
      dx = dxo.scalar_type() == dt ? dxo : dxo.to(dt);
    } else if (need_dx) {
      Tensor lo0 = lo[0], lo1 = lo[1];
      Tensor dxf;
      if (g_deep && g_dual_launch && repmode_conv5_deep_supported((int)w, (int)co, dtype_code(dt)) != 0) {
        // (float accumulation target out of the step's pre-zeroed pool, as conv5 takes its own)
        auto tk = g_pool.take({n, d, h, w, ci}, x_cl);
        dxf = tk.first;
        RM_CALL(repmode_conv5_deep, lo.data_ptr(), wd2.data_ptr(), dxf.data_ptr<float>(), (int)n, (int)d, (int)h, (int)w, (int)co, (int)ci,
                1 | (tk.second ? 2 : 0), stream_handle());
        if (defer) RM_CALL(repmode_tail_flush, stream_handle());
      } else if (g_dual_launch) {
        dxf = conv5(lo.view({2 * n, d, h, w, co}), wd2, s0, ci, true, c10::nullopt, false, false, false, nullptr, DUAL_IN2);
        if (defer) RM_CALL(repmode_tail_flush, stream_handle());
      } else {
        dxf = conv5(lo0, wd2, s0, ci, true);
        conv5(lo1, wd2, s1, ci, true, dxf, true, true);
      }
      // 1x1 experts: one batched GEMM launch gives the three partial data gradients T_e = G_e W_e; the zero-padded box mean is
      // self-adjoint, so the avg experts' parts go back through box3 / box5 -- summed with the two conv parts and cast in
      // the same kernel
      auto tt = g_pool.take({3, m, ci}, x_cl);
      Tensor t = tt.first;
      {
        const Tensor am[3] = {hi[0], hi[1], hi[2]}, bm[3] = {k1, a3, a5};
        Tensor cm[3] = {t[0], t[1], t[2]};
        gemm3(am, co, 1, bm, 1, ci, cm, ci, m, ci, co, tt.second, dt == at::kBFloat16);
      }
      Tensor t0 = t[0].view(dxf.sizes()), t1 = t[1].view(dxf.sizes()), t2 = t[2].view(dxf.sizes());
      dx = box_sum(&t1, &t2, &dxf, &t0, c10::nullopt, dt);
    }
    fork.join();
    return {dx, dk5, dk3, dk1, da3, da5, dgw, dgb, Tensor(), Tensor(), Tensor()};
  }
};

// BatchNorm3d + ReLU on a channels-last tensor [..., C] (RepMode.py:146-149, 212; :80-84; :97-101)
struct BnRelu : public torch::autograd::Function<BnRelu> {
  // stats_half >= 0: the producing conv's epilogue left the batch statistics in that half of the library's scratch
  static Tensor forward(AutogradContext* ctx, Tensor x_cl, Tensor weight, Tensor bias, Tensor rm_, Tensor rv_, bool batch_stats,
                        double momentum, double eps, at::ScalarType out_dtype, int64_t stats_half) {
    const int64_t c = x_cl.size(-1), m = x_cl.numel() / c;
    Tensor out = at::empty(x_cl.sizes(), x_cl.options().dtype(out_dtype));
    Tensor save = at::empty({2, c}, x_cl.options().dtype(at::kFloat));
    RM_CALL(repmode_bn_relu_fwd_ex, x_cl.data_ptr(), out.data_ptr(), weight.data_ptr<float>(), bias.data_ptr<float>(), rm_.data_ptr<float>(),
            rv_.data_ptr<float>(), save[0].data_ptr<float>(), save[1].data_ptr<float>(), (long)m, (int)c, (float)eps, (float)momentum,
            batch_stats ? 1 : 0, dtype_code(x_cl.scalar_type()), dtype_code(out_dtype), batch_stats ? (int)stats_half : -1, stream_handle());
    ctx->save_for_backward({x_cl, weight, bias, save});
    ctx->saved_data["batch_stats"] = batch_stats;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto sv = ctx->get_saved_variables();
    const Tensor &x_cl = sv[0], &weight = sv[1], &bias = sv[2], &save = sv[3];
    const int64_t c = x_cl.size(-1), m = x_cl.numel() / c;
    Tensor dy = grads[0].contiguous();
    Tensor dx = at::empty_like(x_cl);
    Tensor tot = at::empty({2 * c}, x_cl.options().dtype(at::kFloat));    // [0:c) = dbeta, [c:2c) = dgamma
    RM_CALL(repmode_bn_relu_bwd, x_cl.data_ptr(), dy.data_ptr(), weight.data_ptr<float>(), bias.data_ptr<float>(), save[0].data_ptr<float>(),
            save[1].data_ptr<float>(), dx.data_ptr(), tot.data_ptr<float>(), (long)m, (int)c, ctx->saved_data["batch_stats"].toBool() ? 1 : 0,
            dtype_code(x_cl.scalar_type()), dtype_code(dy.scalar_type()), stream_handle());
    return {dx, tot.narrow(0, c, c), tot.narrow(0, 0, c), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ---- the stride-2 2x2x2 stages
// Operands laid out ahead of a forward pass for ALL stages in one launch (op_prepare_stage_filters, round 6: eight ~5 us
// layout launches per pass before); a stage takes its entry, op_finish_prepared forgets what nobody took.
struct K2Prep { Tensor out, out_t; at::ScalarType dt; bool red_major; uint32_t version; };
std::mutex g_k2_prep_mu;
std::unordered_map<void*, K2Prep> g_k2_prep;

void op_prepare_stage_filters(const std::vector<Tensor>& weights, const std::vector<int64_t>& up, int64_t dtype, bool both) {
  const size_t n = weights.size();
  TORCH_CHECK(up.size() == n, "prepare_stage_filters: one `up` flag per weight");
  if (n == 0) return;
  const at::ScalarType dt = dtype == REPMODE_BF16 ? at::kBFloat16 : at::kFloat;
  const int code = (int)dtype;
  std::vector<K2Prep> ents(n);
  for (size_t b0 = 0; b0 < n; b0 += REPMODE_K2_FRAGS_MULTI_MAX) {
    const size_t cnt = std::min<size_t>(REPMODE_K2_FRAGS_MULTI_MAX, n - b0);
    std::vector<const float*> w(cnt);
    std::vector<int> rows(cnt), red(cnt), rm(cnt);
    std::vector<void*> out(cnt), out_t(cnt);
    for (size_t i = 0; i < cnt; ++i) {
      const Tensor& W = weights[b0 + i];
      TORCH_CHECK(W.is_cuda() && W.scalar_type() == at::kFloat && W.is_contiguous() && W.dim() == 5, "prepare_stage_filters: bad weight");
      // Conv3d weight [Co][Ci][2][2][2]: rows = Co, red = Ci; ConvTranspose3d weight [Ci][Co][2][2][2]: rows = Co, red = Ci, red-major
      const bool u = up[b0 + i] != 0;
      rows[i] = (int)(u ? W.size(1) : W.size(0));
      red[i] = (int)(u ? W.size(0) : W.size(1));
      rm[i] = u ? 1 : 0;
      K2Prep& e = ents[b0 + i];
      e.dt = dt; e.red_major = u; e.version = W._version();
      e.out = at::empty({8, padded(rows[i], code, false), padded(red[i], code, true)}, W.options().dtype(dt));
      if (both) e.out_t = at::empty({8, padded(red[i], code, false), padded(rows[i], code, true)}, W.options().dtype(dt));
      w[i] = W.data_ptr<float>();
      out[i] = e.out.data_ptr();
      out_t[i] = both ? e.out_t.data_ptr() : nullptr;
    }
    RM_CALL(repmode_k2_frags_multi, (int)cnt, w.data(), rows.data(), red.data(), rm.data(), code, out.data(), out_t.data(), stream_handle());
  }
  std::lock_guard<std::mutex> lock(g_k2_prep_mu);
  g_k2_prep.clear();
  for (size_t i = 0; i < n; ++i) g_k2_prep[weights[i].data_ptr()] = ents[i];
}

std::pair<Tensor, Tensor> k2_weight_frags(const Tensor& weight, int64_t rows, int64_t red, bool red_major, at::ScalarType dt, bool both) {
  {
    std::lock_guard<std::mutex> lock(g_k2_prep_mu);
    auto it = g_k2_prep.find(weight.data_ptr());
    if (it != g_k2_prep.end()) {
      K2Prep e = it->second;
      g_k2_prep.erase(it);
      if (e.dt == dt && e.red_major == red_major && e.version == weight._version() && (!both || e.out_t.defined()) &&
          e.out.size(1) == padded(rows, dtype_code(dt), false) && e.out.size(2) == padded(red, dtype_code(dt), true))
        return {e.out, both ? e.out_t : Tensor()};
    }
  }
  const int code = dtype_code(dt);
  Tensor out = at::empty({8, padded(rows, code, false), padded(red, code, true)}, weight.options().dtype(dt));
  if (!both) {
    RM_CALL(repmode_k2_frags, weight.data_ptr<float>(), (int)rows, (int)red, red_major ? 1 : 0, code, out.data_ptr(), stream_handle());
    return {out, Tensor()};
  }
  Tensor out_t = at::empty({8, padded(red, code, false), padded(rows, code, true)}, weight.options().dtype(dt));
  RM_CALL(repmode_k2_frags2, weight.data_ptr<float>(), (int)rows, (int)red, red_major ? 1 : 0, code, out.data_ptr(), out_t.data_ptr(),
          stream_handle());
  return {out, out_t};
}

Tensor k2s2(const Tensor& in_cl, const Tensor& w_frag, int64_t cout, bool scatter) {
  const int64_t n = in_cl.size(0), a = in_cl.size(1), b = in_cl.size(2), c = in_cl.size(3), cin = in_cl.size(4);
  const int64_t d = scatter ? a : a / 2, h = scatter ? b : b / 2, w = scatter ? c : c / 2;
  Tensor out = scatter ? at::empty({n, 2 * d, 2 * h, 2 * w, cout}, in_cl.options()) : at::empty({n, d, h, w, cout}, in_cl.options());
  RM_CALL(repmode_k2s2, in_cl.data_ptr(), w_frag.data_ptr(), out.data_ptr(), (int)n, (int)d, (int)h, (int)w, (int)cin, (int)cout,
          dtype_code(in_cl.scalar_type()), scatter ? 1 : 0, stream_handle());
  return out;
}

// dw in a parameter's layout [A, B, 2, 2, 2] (A = coarse channels, B = fine channels)
Tensor k2s2_wgrad_param(const Tensor& coarse_cl, const Tensor& fine_cl) {
  const int64_t n = coarse_cl.size(0), d = coarse_cl.size(1), h = coarse_cl.size(2), w = coarse_cl.size(3), ca = coarse_cl.size(4);
  const int64_t cb = fine_cl.size(4);
  // the kernel accumulates tap-major (atomics into the parameter layout, 32-byte stride, measured 5x slower); one small
  // transpose launch behind it.  bf16: MFMA on LDS-transposed operands; float32 (parity mode): exact-f32 FMAs -- both this
  // build's own kernels (csrc/k2s2.hip), so the float32 goldens verify the same code path shape.
  const bool f32 = coarse_cl.scalar_type() == at::kFloat;
  auto tk = g_pool.take({8, ca, cb}, coarse_cl);
  Tensor acc8 = tk.first;
  const bool zero = tk.second;
  RM_CALL(repmode_k2s2_wgrad_ex, coarse_cl.data_ptr(), fine_cl.data_ptr(), acc8.data_ptr<float>(), (int)n, (int)d, (int)h, (int)w, (int)ca,
          (int)cb, (zero ? 4 : 0) | (f32 ? 8 : 0), stream_handle());
  Tensor dw = at::empty({ca, cb, 2, 2, 2}, coarse_cl.options().dtype(at::kFloat));
  RM_CALL(repmode_tap_transpose, acc8.data_ptr<float>(), dw.data_ptr<float>(), (long)(ca * cb), 8, stream_handle());
  return dw;
}

// Conv3d(C, C, kernel_size=2, stride=2, bias=False) on channels-last data (RepMode.py:81)
struct Down2 : public torch::autograd::Function<Down2> {
  static Tensor forward(AutogradContext* ctx, Tensor x_cl, Tensor weight, bool grad_enabled) {
    const int64_t co = weight.size(0), ci = weight.size(1);
    auto fr = k2_weight_frags(weight, co, ci, false, x_cl.scalar_type(), grad_enabled && x_cl.requires_grad());
    ctx->save_for_backward({x_cl, weight, fr.second.defined() ? fr.second : Tensor()});
    return k2s2(x_cl, fr.first, co, false);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto sv = ctx->get_saved_variables();
    const Tensor &x_cl = sv[0], &weight = sv[1];
    Tensor wb = sv[2];
    Tensor dy = grads[0].to(x_cl.scalar_type()).contiguous();
    Tensor dx;
    if (ctx->needs_input_grad(0) && wb.defined()) dx = k2s2(dy, wb, weight.size(1), true);
    Tensor dw = k2s2_wgrad_param(dy, x_cl);                              // [Co, Ci, 2, 2, 2]
    return {dx, dw, Tensor()};
  }
};

// ConvTranspose3d(Ci, Co, kernel_size=2, stride=2, bias=False) on channels-last data (RepMode.py:98)
struct Up2 : public torch::autograd::Function<Up2> {
  static Tensor forward(AutogradContext* ctx, Tensor x_cl, Tensor weight, bool grad_enabled) {
    const int64_t ci = weight.size(0), co = weight.size(1);
    auto fr = k2_weight_frags(weight, co, ci, true, x_cl.scalar_type(), grad_enabled && x_cl.requires_grad());
    ctx->save_for_backward({x_cl, weight, fr.second.defined() ? fr.second : Tensor()});
    return k2s2(x_cl, fr.first, co, true);
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    auto sv = ctx->get_saved_variables();
    const Tensor &x_cl = sv[0], &weight = sv[1];
    Tensor wb = sv[2];
    Tensor dy = grads[0].to(x_cl.scalar_type()).contiguous();
    Tensor dx;
    if (ctx->needs_input_grad(0) && wb.defined()) dx = k2s2(dy, wb, weight.size(0), false);
    Tensor dw = k2s2_wgrad_param(x_cl, dy);                              // [Ci, Co, 2, 2, 2]
    return {dx, dw, Tensor()};
  }
};

// MSELoss(reduction='none') -> mean with the per-sample and per-task means the reference logs (fnet_model.py:108-109,
// 115-122), one pass + a one-workgroup finish; d loss / d output comes out of the same pass.
// accumulator of the fused loss: one per (device, stream) -- two models or streams computing losses concurrently must not
// share the atomics' target (the finish kernel leaves it zero again)
Tensor mse_sums_ws(const Tensor& like) {
  static std::mutex mu;
  static std::unordered_map<std::string, std::pair<Tensor, bool>> ws;       // accumulator, pinned (a captured graph holds its address)
  const std::string key = std::to_string(like.device().index()) + ":" + std::to_string((uintptr_t)stream_handle());
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(static_cast<hipStream_t>(stream_handle()), &cap);
  std::lock_guard<std::mutex> lock(mu);
  auto it = ws.find(key);
  if (it == ws.end()) {
    // One 4 KB accumulator per stream a loss ever ran on would otherwise live for the process: the map is bounded -- a stream's
    // accumulator is all zero between calls, so forgetting it costs one allocation on its next use.  NOT the entries a capture
    // has seen (advisor round 4): a replayed graph keeps adding into the address it captured; those stay for the process.
    if (ws.size() >= 16)
      for (auto e = ws.begin(); e != ws.end();) e = e->second.second ? std::next(e) : ws.erase(e);
    it = ws.emplace(key, std::make_pair(at::zeros({1024}, like.options().dtype(at::kFloat)), false)).first;
  }
  if (cap != hipStreamCaptureStatusNone) it->second.second = true;
  return it->second.first;
}

struct MseLoss : public torch::autograd::Function<MseLoss> {
  static variable_list forward(AutogradContext* ctx, Tensor out, Tensor target, Tensor sample_task, int64_t num_tasks, bool want_grad) {
    const int64_t n = out.size(0), v = out.numel() / n;
    Tensor loss = at::empty({}, out.options()), loss_sample = at::empty({n}, out.options());
    Tensor task_mean = at::empty({num_tasks}, out.options()), task_count = at::empty({num_tasks}, out.options());
    Tensor dout = want_grad ? at::empty_like(out) : Tensor();
    Tensor ws = mse_sums_ws(out);
    try {
      RM_CALL(repmode_mse_loss, out.data_ptr<float>(), target.data_ptr<float>(), sample_task.data_ptr<int32_t>(), (int)n, (long)v,
              (int)num_tasks, want_grad ? dout.data_ptr<float>() : nullptr, ws.data_ptr<float>(), loss.data_ptr<float>(),
              loss_sample.data_ptr<float>(), task_mean.data_ptr<float>(), task_count.data_ptr<float>(), stream_handle());
    } catch (...) {
      ws.zero_();        // (a launch that failed between "accumulate" and "finish" must not poison the stream's later losses)
      throw;
    }
    ctx->save_for_backward({dout});
    ctx->mark_non_differentiable({loss_sample, task_mean, task_count});
    return {loss, loss_sample, task_mean, task_count};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    Tensor dout = ctx->get_saved_variables()[0];
    TORCH_CHECK(dout.defined(), "mse_loss: the forward pass ran without autograd");
    return {dout * grads[0], Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

std::tuple<Tensor, Tensor, Tensor, Tensor> op_mse_loss(const Tensor& out, const Tensor& target, const Tensor& sample_task, int64_t num_tasks) {
  require_hip(out, "output");
  TORCH_CHECK(out.scalar_type() == at::kFloat && target.scalar_type() == at::kFloat && out.sizes() == target.sizes() && out.dim() >= 2 &&
                  target.device() == out.device(), "mse_loss: float32 output / target of one shape on one device");
  TORCH_CHECK(sample_task.scalar_type() == at::kInt && sample_task.numel() == out.size(0) && sample_task.device() == out.device(),
              "mse_loss: sample_task must be int32 [N] on the output's device");
  TORCH_CHECK(num_tasks >= 1 && num_tasks <= 64 && out.size(0) <= 1024, "mse_loss: at most 64 tasks and 1024 samples");
  DeviceGuard guard(out.device());
  auto r = MseLoss::apply(out.contiguous(), target.contiguous(), sample_task, num_tasks, at::GradMode::is_enabled() && out.requires_grad());
  return std::make_tuple(r[0], r[1], r[2], r[3]);
}

// ------------------------------------------------------------------------------------------------------------
// op entry points

inline Tensor to_cl(const Tensor& x, at::ScalarType dt) {       // logical NCDHW (any strides) -> contiguous NDHWC in dt
  return x.to(dt).permute({0, 2, 3, 4, 1}).contiguous();
}
inline Tensor from_cl(const Tensor& x_cl) { return x_cl.permute({0, 4, 1, 2, 3}); }

void check_params(const Tensor& x_cl, const Tensor* x2_cl, const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3,
                  const Tensor& a5, const Tensor& gw, const Tensor& gb, const Plan& plan) {
  require_hip(x_cl, "input");
  TORCH_CHECK(x_cl.dim() == 5, "mode_conv3d: input must be [N, D, H, W, C], got ", x_cl.sizes());
  TORCH_CHECK(x_cl.scalar_type() == at::kFloat || x_cl.scalar_type() == at::kBFloat16, "mode_conv3d: float32 or bfloat16 input, got ",
              x_cl.scalar_type());
  const int64_t cin = x_cl.size(4) + (x2_cl ? x2_cl->size(4) : 0);
  TORCH_CHECK(k5.dim() == 5 && k5.size(2) == 5 && k5.size(3) == 5 && k5.size(4) == 5, "mode_conv3d: expert_conv5x5 must be [Co, Ci, 5, 5, 5]");
  const int64_t co = k5.size(0), ci = k5.size(1);
  TORCH_CHECK(ci == cin, "mode_conv3d: input has ", cin, " channels, the experts take ", ci);
  TORCH_CHECK(k3.sizes() == at::IntArrayRef({co, ci, 3, 3, 3}), "mode_conv3d: expert_conv3x3 must be [Co, Ci, 3, 3, 3]");
  for (const Tensor* t : {&k1, &a3, &a5}) TORCH_CHECK(t->numel() == co * ci, "mode_conv3d: 1x1 experts must be [Co, Ci, 1, 1, 1]");
  TORCH_CHECK(gw.dim() == 2 && gw.size(0) == E * co && gw.size(1) == plan.num_tasks, "mode_conv3d: gate.weight must be [5*Co, T], got ", gw.sizes());
  TORCH_CHECK(gb.numel() == E * co, "mode_conv3d: gate.bias must be [5*Co]");
  for (const Tensor* t : {&k5, &k3, &k1, &a3, &a5, &gw, &gb}) {
    TORCH_CHECK(t->scalar_type() == at::kFloat, "MoDE parameters must be float32");
    TORCH_CHECK(t->device() == x_cl.device(), "mode_conv3d: parameter on ", t->device(), ", input on ", x_cl.device());
  }
  TORCH_CHECK(plan.n == x_cl.size(0), "task plan is for ", plan.n, " samples, input has ", x_cl.size(0));
  TORCH_CHECK(plan.nslots >= 1 && plan.nslots <= plan.n && plan.sample_slot.numel() == plan.n && plan.slot_task.numel() == plan.nslots &&
                  plan.sample_task.numel() == plan.n, "mode_conv3d: inconsistent task plan");
  for (const Tensor* t : {&plan.slot_task, &plan.sample_slot, &plan.sample_task})
    TORCH_CHECK(t->scalar_type() == at::kInt && t->device() == x_cl.device(), "mode_conv3d: plan vectors must be int32 on the input's device");
  if (x2_cl) {
    TORCH_CHECK(x2_cl->dim() == 5 && x2_cl->sizes().slice(0, 4) == x_cl.sizes().slice(0, 4) && x2_cl->scalar_type() == x_cl.scalar_type() &&
                    x2_cl->device() == x_cl.device(), "mode_conv3d: the two inputs of a skip connection must agree in shape, dtype and device");
  }
}

// torch.cat((skip, up), 1) (RepMode.py:106) on channels-last tensors as a kernel of the library (one launch, visible to the
// library's per-launch timing); the gradient is the two channel ranges of the incoming one.
struct Cat2 : public torch::autograd::Function<Cat2> {
  static Tensor forward(AutogradContext* ctx, Tensor a, Tensor b) {
    const int64_t ca = a.size(-1), cb = b.size(-1), rows = a.numel() / ca;
    const int64_t es = (int64_t)a.element_size();
    std::vector<int64_t> shape = a.sizes().vec();
    shape.back() = ca + cb;
    Tensor out = at::empty(shape, a.options());
    RM_CALL(repmode_concat_channels, a.data_ptr(), b.data_ptr(), out.data_ptr(), (long)rows, (int)(ca * es), (int)(cb * es), stream_handle());
    ctx->saved_data["ca"] = ca;
    ctx->saved_data["cb"] = cb;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const int64_t ca = ctx->saved_data["ca"].toInt(), cb = ctx->saved_data["cb"].toInt();
    return {grads[0].narrow(-1, 0, ca), grads[0].narrow(-1, ca, cb)};
  }
};

inline Tensor cat_channels(const Tensor& a, const Tensor& b) {
  const int64_t es = (int64_t)a.element_size();
  if (a.is_contiguous() && b.is_contiguous() && (a.size(-1) * es) % 16 == 0 && (b.size(-1) * es) % 16 == 0 && a.numel() + b.numel() < (1LL << 31))
    return Cat2::apply(a, b);
  return at::cat({a, b}, -1);
}

// Heuristic: small volumes (levels 3-4) with several distinct tasks in the batch take the per-expert formulation.
inline bool use_unmerged(const Tensor& x_cl, const Plan& plan) { return plan.training && plan.nslots > 2 && x_cl.size(3) <= g_unmerged_max_w; }

inline bool pair_shapes_ok(const Tensor& xa, const Tensor& xb) {
  const int64_t kc = xa.scalar_type() == at::kBFloat16 ? 16 : 8;
  return xa.size(4) % 32 == 0 && xb.size(4) % kc == 0;
}

// The MoDE block up to (not including) BN/ReLU on channels-last tensors.  mode: 0 auto, 1 merged, 2 unmerged (per-expert), 3 merged two-tensor form (needs x2).
Tensor mode_conv3d_cl(const Tensor& x_cl_in, const OptTensor& x2_cl_in, const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3,
                      const Tensor& a5, const Tensor& gw, const Tensor& gb, const Plan& plan, bool out_f32, int64_t mode,
                      const Epi& epi = Epi()) {
  require_hip(x_cl_in, "input");
  DeviceGuard guard(x_cl_in.device());
  Tensor x_cl = x_cl_in.contiguous();
  Tensor x2_cl = x2_cl_in.has_value() ? x2_cl_in->contiguous() : Tensor();
  Tensor ps[7] = {k5.contiguous(), k3.contiguous(), k1.contiguous(), a3.contiguous(), a5.contiguous(), gw.contiguous(), gb.contiguous()};
  check_params(x_cl, x2_cl.defined() ? &x2_cl : nullptr, ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6], plan);
  const bool grad_enabled = at::GradMode::is_enabled();
  if (x2_cl.defined()) {
    // two-tensor kernels: the merged formulation with channel counts on tile boundaries (mode 3 insists on them)
    const bool shapes_ok = pair_shapes_ok(x_cl, x2_cl);
    TORCH_CHECK(mode != 3 || shapes_ok, "mode_conv3d: the two-tensor form needs ", x_cl.size(4), " % 32 == 0 and ", x2_cl.size(4), " % 16 (8) == 0");
    if (mode == 3 || (mode == 1 && shapes_ok) || (mode == 0 && shapes_ok && !use_unmerged(x_cl, plan)))
      return ModeConvPair::apply(x_cl, x2_cl, ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6], plan, out_f32, grad_enabled, epi);
    x_cl = cat_channels(x_cl, x2_cl);       // per-expert formulation / odd channel counts: the concatenated tensor
  }
  if (mode == 0) mode = use_unmerged(x_cl, plan) ? 2 : 1;
  if (mode == 2) {
    TORCH_CHECK(!epi.bias.defined() && !epi.relu, "the per-expert formulation is a training-mode path: no folded BatchNorm");
    return ModeConvUnmerged::apply(x_cl, ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6], plan, grad_enabled, epi.stats);
  }
  return ModeConvMerged::apply(x_cl, ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], ps[6], plan, out_f32, grad_enabled, epi);
}

Tensor bn_relu_cl(const Tensor& x_cl_in, const Tensor& weight, const Tensor& bias, const Tensor& rm_, const Tensor& rv_, bool batch_stats,
                  double momentum, double eps, at::ScalarType out_dtype, int64_t stats_half = -1) {
  require_hip(x_cl_in, "input");
  DeviceGuard guard(x_cl_in.device());
  Tensor x_cl = x_cl_in.contiguous();
  const int64_t c = x_cl.size(-1);
  TORCH_CHECK(c <= 512, "bn_relu: at most 512 channels, got ", c);
  for (const Tensor* t : {&weight, &bias, &rm_, &rv_})
    TORCH_CHECK(t->scalar_type() == at::kFloat && t->numel() == c && t->is_contiguous() && t->device() == x_cl.device(),
                "bn_relu: BatchNorm parameters / running statistics must be contiguous float32 [C] on the input's device");
  return BnRelu::apply(x_cl, weight, bias, rm_, rv_, batch_stats, momentum, eps, out_dtype, stats_half);
}

Plan make_plan(const Tensor& slot_task, const Tensor& sample_slot, const Tensor& sample_task, int64_t nslots, int64_t num_tasks, bool training,
               int64_t task0) {
  Plan p;
  p.slot_task = slot_task;
  p.sample_slot = sample_slot;
  p.sample_task = sample_task;
  p.nslots = nslots;
  p.n = sample_slot.numel();
  p.num_tasks = num_tasks;
  p.training = training;
  p.task0 = task0;
  return p;
}

// ---- exported ops (schemas at the bottom)

Tensor op_mode_conv3d(const Tensor& x_cl, const OptTensor& x2_cl, const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3,
                      const Tensor& a5, const Tensor& gw, const Tensor& gb, const Tensor& slot_task, const Tensor& sample_slot,
                      const Tensor& sample_task, int64_t nslots, int64_t num_tasks, bool training, int64_t task0, bool out_f32, int64_t mode) {
  return mode_conv3d_cl(x_cl, x2_cl, k5, k3, k1, a3, a5, gw, gb, make_plan(slot_task, sample_slot, sample_task, nslots, num_tasks, training, task0),
                        out_f32, mode);
}

// One MoDE block (RepMode.py:194-214) on logical NCDHW tensors: cast + channels-last, fused op, BatchNorm3d + ReLU when the
// block has them ('normal'), NCDHW view of the channels-last result.
Tensor op_mode_block(const Tensor& x, const OptTensor& x2, const Tensor& k5, const Tensor& k3, const Tensor& k1, const Tensor& a3,
                     const Tensor& a5, const Tensor& gw, const Tensor& gb, const OptTensor& bn_w, const OptTensor& bn_b, const OptTensor& bn_rm,
                     const OptTensor& bn_rv, bool bn_batch_stats, double bn_momentum, double bn_eps, const Tensor& slot_task,
                     const Tensor& sample_slot, const Tensor& sample_task, int64_t nslots, int64_t num_tasks, bool training, int64_t task0,
                     int64_t dtype, bool out_f32) {
  require_hip(x, "input");
  TORCH_CHECK(x.dim() == 5, "MoDE block: input must be [N, C, D, H, W], got ", x.sizes());
  const at::ScalarType dt = code_dtype(dtype);
  {
    // float output also where the convolution splits its channel reduction over workgroups (small volumes: the deep levels)
    DeviceGuard guard(x.device());
    const int64_t cin = x.size(1) + (x2.has_value() ? x2->size(1) : 0);
    if (dt == at::kBFloat16 && !elem_out(x.size(0), x.size(2), x.size(3), x.size(4), cin, k5.size(0), dt)) out_f32 = true;
  }
  Plan plan = make_plan(slot_task, sample_slot, sample_task, nslots, num_tasks, training, task0);
  OptTensor x2_cl;
  if (x2.has_value()) x2_cl = to_cl(*x2, dt);
  Tensor x_cl = to_cl(x, dt);
  const bool has_bn = bn_w.has_value();
  if (has_bn) TORCH_CHECK(bn_b.has_value() && bn_rm.has_value() && bn_rv.has_value(), "MoDE block: incomplete BatchNorm state");
  Epi epi;
  bool folded = false;
  const bool unmerged = use_unmerged(x_cl, plan);
  if (has_bn && bn_batch_stats && unmerged) {
    // a per-expert block (the deep levels): the one-launch forward (csrc/deep_mode.hip) leaves the statistics of its output in
    // the BatchNorm scratch wherever it writes y with plain stores -- there the statistics pass is a launch of pure latency
    epi.stats = (g_deep_mode & 4) && (g_deep_mode & 1) && dt == at::kBFloat16 && k5.size(0) <= 512;
  } else if (has_bn && g_bn_epilogue) {
    if (bn_batch_stats && (g_bn_epilogue & 2)) {
      // training: the batch statistics come out of the conv's epilogue where the conv writes the element-typed tensor the
      // BatchNorm normalises (bf16, levels 0-1: 94 % of the normalised bytes); elsewhere the separate statistics pass
      epi.stats = dt == at::kBFloat16 && !out_f32 && k5.size(0) <= 512;
    } else if (!bn_batch_stats && (g_bn_epilogue & 1) && !plan.training && !at::GradMode::is_enabled() &&
               ((dt == at::kBFloat16 && !out_f32) || dt == at::kFloat)) {
      // eval, no autograd: y = relu(gamma (conv - mean) / sqrt(var + eps) + beta) = relu(conv with scaled filter + bias).
      // Not where the bf16 path writes a float tensor (small volumes): a bias / ReLU epilogue cannot split the reduction
      // over workgroups, which is what fills the chip there (measured on the 64x624x924 stack: 0.49 -> 0.57 s when folded
      // everywhere); those levels keep the separate normalise + ReLU launch.
      epi.scale = *bn_w * at::rsqrt(*bn_rv + bn_eps);
      epi.bias = (*bn_b - *bn_rm * epi.scale).contiguous();
      epi.relu = true;
      folded = true;
    }
  }
  Tensor y = mode_conv3d_cl(x_cl, x2_cl, k5, k3, k1, a3, a5, gw, gb, plan, out_f32, 0, epi);
  if (has_bn) {
    if (folded) {
      if (y.scalar_type() != dt) y = y.to(dt);
    } else {
      const int half = epi.stats ? tl_stats_half : -1;
      tl_stats_half = -1;
      y = bn_relu_cl(y, *bn_w, *bn_b, *bn_rm, *bn_rv, bn_batch_stats, bn_momentum, bn_eps, dt, half);
    }
  }
  return from_cl(y);
}

Tensor op_bn_relu(const Tensor& x_cl, const Tensor& weight, const Tensor& bias, const Tensor& rm_, const Tensor& rv_, bool batch_stats,
                  double momentum, double eps, int64_t out_dtype) {
  return bn_relu_cl(x_cl, weight, bias, rm_, rv_, batch_stats, momentum, eps, code_dtype(out_dtype));
}

void check_k2(const Tensor& x_cl, const Tensor& weight, const char* what) {
  require_hip(x_cl, "input");
  TORCH_CHECK(x_cl.dim() == 5 && weight.dim() == 5 && weight.size(2) == 2 && weight.size(3) == 2 && weight.size(4) == 2 &&
                  weight.scalar_type() == at::kFloat && weight.device() == x_cl.device(), what, ": bad input / weight");
}

Tensor op_down2(const Tensor& x_cl, const Tensor& weight) {
  check_k2(x_cl, weight, "down2");
  TORCH_CHECK(x_cl.size(1) % 2 == 0 && x_cl.size(2) % 2 == 0 && x_cl.size(3) % 2 == 0 && x_cl.size(4) == weight.size(1), "down2: bad shape ", x_cl.sizes());
  DeviceGuard guard(x_cl.device());
  return Down2::apply(x_cl.contiguous(), weight.contiguous(), at::GradMode::is_enabled());
}

Tensor op_up2(const Tensor& x_cl, const Tensor& weight) {
  check_k2(x_cl, weight, "up2");
  TORCH_CHECK(x_cl.size(4) == weight.size(0), "up2: bad shape ", x_cl.sizes());
  DeviceGuard guard(x_cl.device());
  return Up2::apply(x_cl.contiguous(), weight.contiguous(), at::GradMode::is_enabled());
}

// Conv3d(k2, s2) / ConvTranspose3d(k2, s2) + BatchNorm3d + ReLU of the encoder / decoder (RepMode.py:80-84, 97-101), NCDHW in and out
Tensor op_stage2_bn_relu(const Tensor& x, const Tensor& weight, const Tensor& bn_w, const Tensor& bn_b, const Tensor& bn_rm, const Tensor& bn_rv,
                         bool bn_batch_stats, double bn_momentum, double bn_eps, bool up, int64_t out_dtype) {
  TORCH_CHECK(x.dim() == 5, "stride-2 stage: input must be [N, C, D, H, W]");
  Tensor x_cl = x.permute({0, 2, 3, 4, 1});
  Tensor y = up ? op_up2(x_cl, weight) : op_down2(x_cl, weight);
  return from_cl(bn_relu_cl(y, bn_w, bn_b, bn_rm, bn_rv, bn_batch_stats, bn_momentum, bn_eps, code_dtype(out_dtype)));
}

// ------------------------------------------------------------------------------------------------------------
// The per-expert blocks' experts in the conv kernels' layout, kept ACROSS steps (round 4).  They depend on the parameters
// only, so the optimizer pass that has just written the parameters emits them as well (op adam_step ->
// repmode_adam_expert_frags) and the next forward pass finds them ready: no layout launch, no second read of the experts.
// An entry is valid for one version of its two parameters (autograd's version counters: any in-place write from outside --
// load_state_dict, another optimizer -- makes it stale and prepare_filters lays the experts out again into the same buffers);
// weak references tell a dead parameter from a new one at the same address.
struct FragEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> w5, w3;
  Tensor wf, wd;
  int64_t co = 0, ci = 0;
  uint32_t ver5 = 0, ver3 = 0;
  bool wf_valid = false, wd_valid = false;
  bool used = false;             // a forward pass since the last optimizer step took it
  bool pinned = false;           // a stream capture took wf / wd: replays hold their raw addresses (see op_pinned_operands)
  FragEntry(const Tensor& k5, const Tensor& k3)
      : w5(c10::intrusive_ptr<c10::TensorImpl>::reclaim_copy(k5.unsafeGetTensorImpl())),
        w3(c10::intrusive_ptr<c10::TensorImpl>::reclaim_copy(k3.unsafeGetTensorImpl())) {}
  bool same_params(const Tensor& k5, const Tensor& k3) const {
    return !w5.expired() && !w3.expired() && w5._unsafe_get_target() == k5.unsafeGetTensorImpl() &&
           w3._unsafe_get_target() == k3.unsafeGetTensorImpl();
  }
  bool current(const Tensor& k5, const Tensor& k3) const { return same_params(k5, k3) && ver5 == k5._version() && ver3 == k3._version(); }
};
std::mutex g_frag_mu;
std::unordered_map<void*, FragEntry> g_frag_store;
// Operand buffers that a captured graph reads and writes by raw address: when their entry leaves the store --
// another optimizer stepped, another Model was built, a launch failed -- the buffers must outlive the graph, or its replays
// would read filters from freed memory and write the bf16 operands into whatever the allocator put there.  So the holder of
// the graph holds the tensors too: Model asks for them right after a capture (pinned_operands) and keeps them beside the
// graph.  A replay after the entry left the store still finds CORRECT operands: the captured pass contains the on-device check
// (expert_frags_refresh_multi), which lays a block out again when the stored operands no longer match the parameters' bytes.
std::vector<Tensor> op_pinned_operands() {
  std::lock_guard<std::mutex> lock(g_frag_mu);
  std::vector<Tensor> out;
  for (auto& kv : g_frag_store)
    if (kv.second.pinned) {
      if (kv.second.wf.defined()) out.push_back(kv.second.wf);
      if (kv.second.wd.defined()) out.push_back(kv.second.wd);
    }
  return out;
}
bool g_frag_keep = []() {          // (REPMODE_FRAG_STORE=0: lay the experts out at every forward pass, as round 3 did)
  const char* e = std::getenv("REPMODE_FRAG_STORE");
  return e ? std::atoi(e) != 0 : true;
}();

// REPMODE_FRAG_VERIFY=0 / set_frag_verify(False): trust version counters + the optimizer hook alone (round 4's behaviour)
bool g_frag_verify = []() {
  const char* e = std::getenv("REPMODE_FRAG_VERIFY");
  return e ? std::atoi(e) != 0 : true;
}();
Tensor g_frag_flags[32];        // per device: int32[REPMODE_GATREP_MULTI_MAX], the check's verdicts (device-side only)
void op_set_frag_verify(bool on) { g_frag_verify = on; }

void op_clear_frag_store() {
  std::lock_guard<std::mutex> lock(g_frag_mu);
  g_frag_store.clear();
}
// (a model whose train step is replayed as a HIP graph switches the store off: a replay updates the parameters with no
// version counter and no optimizer hook to say so)
void op_set_frag_store(bool on) {
  std::lock_guard<std::mutex> lock(g_frag_mu);
  g_frag_keep = on;
  if (!on) g_frag_store.clear();
}
int64_t op_frag_store_size() {
  std::lock_guard<std::mutex> lock(g_frag_mu);
  return (int64_t)g_frag_store.size();
}

// torch.optim.Adam's step (fnet_model.py:55, 112) over the whole parameter list through the build's own kernels
// (csrc/adam.hip): the 5x5x5 / 3x3x3 experts of the blocks whose last forward pass ran the per-expert formulation go
// through repmode_adam_expert_frags (update + the conv operands of the next forward pass), everything else through
// repmode_adam_multi.  `step`: the 1-based count of this update (all tensors of a call share it).
// hyper_dev != nullptr: a capturable step -- the step count lives on the device (step_dev, int64[1]); repmode_adam_hyper_dev
// advances it and leaves the step's constants in hyper_dev, which both passes read (no host scalar in the launch arguments,
// so a HIP-graph replay of the step takes the next step count).
void adam_step_impl(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& exp_avgs,
                    const std::vector<Tensor>& exp_avg_sqs, double lr, double beta1, double beta2, double eps, int64_t step,
                    const Tensor* step_dev, const Tensor* hyper_dev) {
  const size_t n = params.size();
  TORCH_CHECK(grads.size() == n && exp_avgs.size() == n && exp_avg_sqs.size() == n, "adam_step: list lengths differ");
  if (n == 0) return;
  require_hip(params[0], "parameters");
  DeviceGuard guard(params[0].device());
  std::unordered_map<const void*, size_t> index;
  for (size_t i = 0; i < n; ++i) {
    for (const Tensor* t : {&params[i], &grads[i], &exp_avgs[i], &exp_avg_sqs[i]})
      TORCH_CHECK(t->scalar_type() == at::kFloat && t->is_contiguous() && t->device() == params[0].device() && t->numel() == params[i].numel(),
                  "adam_step: parameter ", i, ": float32 contiguous tensors of one shape on one device (parameter, gradient, exp_avg, exp_avg_sq)");
    index[params[i].data_ptr()] = i;
  }
  const float* hd = nullptr;
  if (hyper_dev) {
    TORCH_CHECK(step_dev && step_dev->is_cuda() && step_dev->scalar_type() == at::kLong && step_dev->numel() == 1 && hyper_dev->is_cuda() &&
                    hyper_dev->scalar_type() == at::kFloat && hyper_dev->numel() >= REPMODE_ADAM_HYPER_FLOATS && hyper_dev->is_contiguous(),
                "adam_step_dev: step_dev int64[1] and hyper_dev float[", REPMODE_ADAM_HYPER_FLOATS, "] on the device");
    RM_CALL(repmode_adam_hyper_dev, reinterpret_cast<long*>(step_dev->data_ptr<int64_t>()), hyper_dev->data_ptr<float>(), lr, beta1, beta2, eps,
            stream_handle());
    hd = hyper_dev->data_ptr<float>();
  }
  std::vector<char> done(n, 0);
  // ---- the per-expert blocks
  struct Blk { size_t i5, i3; FragEntry* fe; };
  std::vector<Blk> blks;
  {
    std::lock_guard<std::mutex> lock(g_frag_mu);
    try {
    // entries whose parameters are gone (a network that was rebuilt or freed) give their operands back (advisor round 4)
    for (auto it = g_frag_store.begin(); it != g_frag_store.end();) {
      if (it->second.w5.expired() || it->second.w3.expired()) it = g_frag_store.erase(it);
      else ++it;
    }
    for (auto& kv : g_frag_store) {
      FragEntry& fe = kv.second;
      if (!fe.used) continue;
      auto it5 = index.find(kv.first);
      if (it5 == index.end()) continue;
      const Tensor& k5 = params[it5->second];
      if (fe.w3.expired()) continue;
      auto it3 = index.find(fe.w3._unsafe_get_target()->data());
      if (it3 == index.end() || !fe.same_params(k5, params[it3->second])) continue;
      blks.push_back({it5->second, it3->second, &fe});
    }
    for (size_t b0 = 0; b0 < blks.size(); b0 += REPMODE_GATREP_MULTI_MAX) {
      const int cnt = (int)std::min<size_t>(REPMODE_GATREP_MULTI_MAX, blks.size() - b0);
      std::vector<float*> p5, m5, v5, p3, m3, v3;
      std::vector<const float*> g5, g3;
      std::vector<int> co, ci;
      std::vector<void*> wf, wd;
      for (int j = 0; j < cnt; ++j) {
        const Blk& b = blks[b0 + j];
        p5.push_back(params[b.i5].data_ptr<float>()); g5.push_back(grads[b.i5].data_ptr<float>());
        m5.push_back(exp_avgs[b.i5].data_ptr<float>()); v5.push_back(exp_avg_sqs[b.i5].data_ptr<float>());
        p3.push_back(params[b.i3].data_ptr<float>()); g3.push_back(grads[b.i3].data_ptr<float>());
        m3.push_back(exp_avgs[b.i3].data_ptr<float>()); v3.push_back(exp_avg_sqs[b.i3].data_ptr<float>());
        co.push_back((int)b.fe->co); ci.push_back((int)b.fe->ci);
        wf.push_back(b.fe->wf.defined() ? b.fe->wf.data_ptr() : nullptr);
        wd.push_back(b.fe->wd.defined() ? b.fe->wd.data_ptr() : nullptr);
        done[b.i5] = done[b.i3] = 1;
      }
      if (hd)
        RM_CALL(repmode_adam_expert_frags_dev, cnt, p5.data(), g5.data(), m5.data(), v5.data(), p3.data(), g3.data(), m3.data(), v3.data(),
                co.data(), ci.data(), wf.data(), wd.data(), hd, stream_handle());
      else
        RM_CALL(repmode_adam_expert_frags, cnt, p5.data(), g5.data(), m5.data(), v5.data(), p3.data(), g3.data(), m3.data(), v3.data(),
                co.data(), ci.data(), wf.data(), wd.data(), lr, beta1, beta2, eps, (long)step, stream_handle());
    }
    // ---- every other tensor
    std::vector<float*> p, m, v;
    std::vector<const float*> g;
    std::vector<long> numel;
    auto flush = [&]() {
      if (p.empty()) return;
      if (hd)
        RM_CALL(repmode_adam_multi_dev, (int)p.size(), p.data(), g.data(), m.data(), v.data(), numel.data(), hd, stream_handle());
      else
        RM_CALL(repmode_adam_multi, (int)p.size(), p.data(), g.data(), m.data(), v.data(), numel.data(), lr, beta1, beta2, eps, (long)step,
                stream_handle());
      p.clear(); g.clear(); m.clear(); v.clear(); numel.clear();
    };
    for (size_t i = 0; i < n; ++i) {
      if (done[i] || params[i].numel() == 0) continue;
      p.push_back(params[i].data_ptr<float>()); g.push_back(grads[i].data_ptr<float>());
      m.push_back(exp_avgs[i].data_ptr<float>()); v.push_back(exp_avg_sqs[i].data_ptr<float>());
      numel.push_back((long)params[i].numel());
      if (p.size() == REPMODE_ADAM_MULTI_MAX) flush();
    }
    flush();
    // the parameters changed in place: autograd's version counters say so, and the operands emitted above belong to the new version
    for (size_t i = 0; i < n; ++i) params[i].unsafeGetTensorImpl()->bump_version();
    for (const Blk& b : blks) {
      b.fe->ver5 = params[b.i5]._version();
      b.fe->ver3 = params[b.i3]._version();
      b.fe->wf_valid = b.fe->wf.defined();
      b.fe->wd_valid = b.fe->wd.defined();
      b.fe->used = false;
    }
    } catch (...) {
      // a launch failed half way: some parameters are updated, some are not -- say so to autograd, and no stored operand
      // may pass for current
      for (size_t i = 0; i < n; ++i) params[i].unsafeGetTensorImpl()->bump_version();
      g_frag_store.clear();
      throw;
    }
  }
}

void op_adam_step(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& exp_avgs,
                  const std::vector<Tensor>& exp_avg_sqs, double lr, double beta1, double beta2, double eps, int64_t step) {
  adam_step_impl(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step, nullptr, nullptr);
}
void op_adam_step_dev(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& exp_avgs,
                      const std::vector<Tensor>& exp_avg_sqs, double lr, double beta1, double beta2, double eps, const Tensor& step_dev,
                      const Tensor& hyper_dev) {
  adam_step_impl(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, 0, &step_dev, &hyper_dev);
}

// All blocks' forward filters on the `prep` stream, ahead of the forward pass (see PrepEntry).  w_in[i]: the x extent of
// block i's input (selects the formulation exactly as mode_conv3d does), need_dx[i]: the block's input needs a gradient.
void op_prepare_filters(const std::vector<Tensor>& k5, const std::vector<Tensor>& k3, const std::vector<Tensor>& k1,
                        const std::vector<Tensor>& a3, const std::vector<Tensor>& a5, const std::vector<Tensor>& gw,
                        const std::vector<Tensor>& gb, const std::vector<int64_t>& w_in, const std::vector<int64_t>& need_dx,
                        const Tensor& slot_task, const Tensor& sample_slot, const Tensor& sample_task, int64_t nslots, int64_t num_tasks,
                        bool training, int64_t dtype) {
  const size_t nb = k5.size();
  TORCH_CHECK(k3.size() == nb && k1.size() == nb && a3.size() == nb && a5.size() == nb && gw.size() == nb && gb.size() == nb &&
                  w_in.size() == nb && need_dx.size() == nb, "prepare_filters: list lengths differ");
  {
    std::lock_guard<std::mutex> lock(g_prep_mu);
    g_prep.clear();                        // (entries a failed forward left behind)
  }
  if (nb == 0 || !g_prepare) return;
  require_hip(k5[0], "parameters");
  const at::ScalarType dt = code_dtype(dtype);
  DeviceGuard guard(k5[0].device());
  if (!g_overlap) {
    // in line, on the caller's stream: the gate softmax + GatRep of every block that takes the merged formulation as ONE
    // launch (repmode_gatrep_fwd_multi); the per-expert blocks lay their experts out themselves
    Plan plan = make_plan(slot_task, sample_slot, sample_task, nslots, num_tasks, training, 0);
    const int code = dtype_code(dt);
    std::vector<const float*> p5, p3, p1, pa3, pa5, pgw, pgb;
    std::vector<float*> pg;
    std::vector<void*> pwf, pwd;
    std::vector<int> cos, cis;
    std::vector<PrepEntry> ents;
    std::vector<void*> keys;
    std::vector<Tensor> keep;          // contiguous copies (if any) must outlive the launch call
    // ... and the raw 5x5x5 / 3x3x3 experts of every per-expert block in the conv kernels' layout as another one
    // (repmode_expert_frags_multi; bf16 only -- float32 lays them out through GatRep with one-hot gates, block by block)
    {
      std::vector<const float*> x5, x3;
      std::vector<void*> xwf, xwd;
      std::vector<int> xco, xci;
      std::vector<PrepEntry> xe;
      std::vector<void*> xk;
      // (under stream capture the operands are the graph's own tensors: what a replay writes cannot be tracked by versions)
      hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(c10::hip::getCurrentHIPStream(k5[0].device().index()).stream(), &cap_status);
      const bool capturing = cap_status != hipStreamCaptureStatusNone;
      std::vector<const float*> v5, v3;          // entries of the store taken as current: verified (and repaired) on the device
      std::vector<void*> vwf, vwd;
      std::vector<int> vco, vci;
      std::vector<const float*> ggw, ggb;       // the per-expert blocks' gates (per SAMPLE), all from one launch
      std::vector<float*> ggo;
      std::vector<int> gco;
      try {      // (entries of the store are marked current below, ahead of the launches that make them so: any failure empties it)
      for (size_t i = 0; i < nb && dt == at::kBFloat16; ++i) {
        if (!(plan.training && plan.nslots > 2 && w_in[i] <= g_unmerged_max_w)) continue;
        Tensor K5 = k5[i].contiguous(), K3 = k3[i].contiguous();
        keep.push_back(K5); keep.push_back(K3);
        const int64_t co = K5.size(0), ci = K5.size(1);
        PrepEntry e;
        e.dt = dt;
        e.unmerged = true;
        e.rows = plan.n;
        {
          Tensor GW = gw[i].contiguous(), GB = gb[i].contiguous();
          keep.push_back(GW); keep.push_back(GB);
          e.g = at::empty({plan.n, E, co}, K5.options());
          ggw.push_back(GW.data_ptr<float>()); ggb.push_back(GB.data_ptr<float>());
          ggo.push_back(e.g.data_ptr<float>()); gco.push_back((int)co);
        }
        bool ready = false;
        bool use_store = g_frag_keep && K5.is_same(k5[i]) && K3.is_same(k3[i]);
        if (use_store && capturing) {
          // Under stream capture the store is used only where an entry is READY (laid out and kept current by the eager warm-up
          // steps of the build's own optimizer): the captured optimizer pass then keeps it current in every replay
          // (repmode_adam_expert_frags_dev writes these very buffers), and nothing is allocated or laid out for the store
          // inside a graph.  Otherwise the graph lays the operands out into its own tensors at every replay.
          std::lock_guard<std::mutex> lock(g_frag_mu);
          auto it = g_frag_store.find(K5.data_ptr());
          use_store = it != g_frag_store.end() && it->second.current(K5, K3) && it->second.co == co && it->second.ci == ci &&
                      it->second.wf_valid && it->second.wf.defined() && (!need_dx[i] || (it->second.wd_valid && it->second.wd.defined()));
        }
        if (use_store) {
          // the layouts live across steps (FragEntry): the optimizer pass leaves them current; otherwise lay out into them
          std::lock_guard<std::mutex> lock(g_frag_mu);
          auto it = g_frag_store.find(K5.data_ptr());
          if (it != g_frag_store.end() && !(it->second.same_params(K5, K3) && it->second.co == co && it->second.ci == ci)) {
            g_frag_store.erase(it);
            it = g_frag_store.end();
          }
          if (it == g_frag_store.end()) it = g_frag_store.emplace(K5.data_ptr(), FragEntry(K5, K3)).first;
          FragEntry& fe = it->second;
          fe.co = co; fe.ci = ci;
          const bool cur = fe.current(K5, K3);
          if (!fe.wf.defined()) fe.wf = at::empty({2, TAPS, padded(co, code, false), padded(ci, code, true)}, K5.options().dtype(dt));
          if (need_dx[i] && !fe.wd.defined()) fe.wd = at::empty({2, TAPS, padded(ci, code, false), padded(co, code, true)}, K5.options().dtype(dt));
          ready = cur && fe.wf_valid && (!need_dx[i] || fe.wd_valid);
          e.wf = fe.wf;
          if (need_dx[i]) e.wd = fe.wd;
          if (!ready) {
            // (laid out below, on this stream, for the parameters' current version: both layouts the entry holds)
            fe.ver5 = K5._version(); fe.ver3 = K3._version();
            fe.wf_valid = true;
            fe.wd_valid = fe.wd.defined();
          }
          fe.used = true;
          if (capturing) fe.pinned = true;       // (the graph holds wf / wd by address from here on)
          if (!ready) {
            x5.push_back(K5.data_ptr<float>()); x3.push_back(K3.data_ptr<float>());
            xwf.push_back(fe.wf.data_ptr()); xwd.push_back(fe.wd.defined() ? fe.wd.data_ptr() : nullptr);
            xco.push_back((int)co); xci.push_back((int)ci);
          } else {
            // kept from the last step: checked against the parameters' BYTES on the device below (a write that moved no
            // version counter -- p.data.copy_(), a broadcast, a foreign kernel -- must not leave stale filters in use)
            v5.push_back(K5.data_ptr<float>()); v3.push_back(K3.data_ptr<float>());
            vwf.push_back(fe.wf.data_ptr()); vwd.push_back(fe.wd.defined() ? fe.wd.data_ptr() : nullptr);
            vco.push_back((int)co); vci.push_back((int)ci);
          }
        } else {
          e.wf = at::empty({2, TAPS, padded(co, code, false), padded(ci, code, true)}, K5.options().dtype(dt));
          if (need_dx[i]) e.wd = at::empty({2, TAPS, padded(ci, code, false), padded(co, code, true)}, K5.options().dtype(dt));
          x5.push_back(K5.data_ptr<float>()); x3.push_back(K3.data_ptr<float>());
          xwf.push_back(e.wf.data_ptr()); xwd.push_back(e.wd.defined() ? e.wd.data_ptr() : nullptr);
          xco.push_back((int)co); xci.push_back((int)ci);
        }
        xe.push_back(e); xk.push_back(K5.data_ptr());
      }
      for (size_t b0 = 0; b0 < ggw.size(); b0 += REPMODE_GATREP_MULTI_MAX) {
        const int cnt = (int)std::min<size_t>(REPMODE_GATREP_MULTI_MAX, ggw.size() - b0);
        RM_CALL(repmode_gate_softmax_multi, cnt, ggw.data() + b0, ggb.data() + b0, gco.data() + b0, plan.sample_task.data_ptr<int32_t>(),
                (int)plan.n, (int)plan.num_tasks, ggo.data() + b0, stream_handle());
      }
      for (size_t b0 = 0; b0 < x5.size(); b0 += REPMODE_GATREP_MULTI_MAX) {
        const int cnt = (int)std::min<size_t>(REPMODE_GATREP_MULTI_MAX, x5.size() - b0);
        RM_CALL(repmode_expert_frags_multi, cnt, x5.data() + b0, x3.data() + b0, xco.data() + b0, xci.data() + b0, xwf.data() + b0,
                xwd.data() + b0, stream_handle());
      }
      if (g_frag_verify && !v5.empty()) {
        Tensor& flags = g_frag_flags[k5[0].device().index() & 31];
        if (!flags.defined()) {
          TORCH_CHECK(!capturing, "prepare_filters: the operand check's flag buffer must exist before a capture (run a step launch by launch first)");
          flags = at::zeros({REPMODE_GATREP_MULTI_MAX}, k5[0].options().dtype(at::kInt));
        }
        for (size_t b0 = 0; b0 < v5.size(); b0 += REPMODE_GATREP_MULTI_MAX) {
          const int cnt = (int)std::min<size_t>(REPMODE_GATREP_MULTI_MAX, v5.size() - b0);
          RM_CALL(repmode_expert_frags_refresh_multi, cnt, v5.data() + b0, v3.data() + b0, vco.data() + b0, vci.data() + b0, vwf.data() + b0,
                  vwd.data() + b0, flags.data_ptr<int>(), stream_handle());
        }
      }
      } catch (...) {
        op_clear_frag_store();          // (an allocation or a launch failed: no entry may claim operands that were never written)
        throw;
      }
      std::lock_guard<std::mutex> lock(g_prep_mu);
      for (size_t i = 0; i < xe.size(); ++i) g_prep[xk[i]] = xe[i];
    }
    for (size_t i = 0; i < nb; ++i) {
      if (plan.training && plan.nslots > 2 && w_in[i] <= g_unmerged_max_w) continue;
      Tensor t[7] = {k5[i].contiguous(), k3[i].contiguous(), k1[i].contiguous(), a3[i].contiguous(), a5[i].contiguous(),
                     gw[i].contiguous(), gb[i].contiguous()};
      for (auto& x : t) keep.push_back(x);
      const int64_t co = t[0].size(0), ci = t[0].size(1);
      PrepEntry e;
      e.dt = dt;
      e.unmerged = false;
      e.rows = plan.nslots;
      e.g = at::empty({plan.nslots, E, co}, t[0].options());
      e.wf = at::empty({plan.nslots, TAPS, padded(co, code, false), padded(ci, code, true)}, t[0].options().dtype(dt));
      if (need_dx[i]) e.wd = at::empty({plan.nslots, TAPS, padded(ci, code, false), padded(co, code, true)}, t[0].options().dtype(dt));
      p5.push_back(t[0].data_ptr<float>()); p3.push_back(t[1].data_ptr<float>()); p1.push_back(t[2].data_ptr<float>());
      pa3.push_back(t[3].data_ptr<float>()); pa5.push_back(t[4].data_ptr<float>());
      pgw.push_back(t[5].data_ptr<float>()); pgb.push_back(t[6].data_ptr<float>());
      pg.push_back(e.g.data_ptr<float>()); pwf.push_back(e.wf.data_ptr()); pwd.push_back(e.wd.defined() ? e.wd.data_ptr() : nullptr);
      cos.push_back((int)co); cis.push_back((int)ci);
      ents.push_back(e);
      keys.push_back(t[0].data_ptr());
    }
    for (size_t b0 = 0; b0 < ents.size(); b0 += REPMODE_GATREP_MULTI_MAX) {
      const int cnt = (int)std::min<size_t>(REPMODE_GATREP_MULTI_MAX, ents.size() - b0);
      RM_CALL(repmode_gatrep_fwd_multi, cnt, p5.data() + b0, p3.data() + b0, p1.data() + b0, pa3.data() + b0, pa5.data() + b0,
              pgw.data() + b0, pgb.data() + b0, cos.data() + b0, cis.data() + b0, plan.slot_task.data_ptr<int32_t>(), (int)plan.nslots,
              (int)plan.num_tasks, code, pg.data() + b0, pwf.data() + b0, pwd.data() + b0, stream_handle());
    }
    std::lock_guard<std::mutex> lock(g_prep_mu);
    for (size_t i = 0; i < ents.size(); ++i) g_prep[keys[i]] = ents[i];
    return;
  }
  const int dev = k5[0].device().index();
  SideStreams& ss = side_streams(dev);
  const c10::hip::HIPStream main_s = c10::hip::getCurrentHIPStream(dev);
  {
    std::lock_guard<std::mutex> lock(g_prep_mu);
    while (g_prep_events.size() < nb) {
      hipEvent_t ev;
      RM_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      g_prep_events.push_back(ev);
    }
  }
  Plan plan = make_plan(slot_task, sample_slot, sample_task, nslots, num_tasks, training, 0);
  RM_HIP_CHECK(hipEventRecord(ss.prep_fork_ev, main_s.stream()));
  RM_HIP_CHECK(hipStreamWaitEvent(ss.prep.stream(), ss.prep_fork_ev, 0));
  c10::hip::setCurrentHIPStream(ss.prep);
  try {
    for (size_t i = 0; i < nb; ++i) {
      Tensor K5 = k5[i].contiguous(), K3 = k3[i].contiguous();
      const int64_t co = K5.size(0);
      PrepEntry e;
      e.dt = dt;
      e.unmerged = plan.training && plan.nslots > 2 && w_in[i] <= g_unmerged_max_w;
      if (e.unmerged) {
        e.rows = plan.n;
        e.g = gate_softmax(gw[i].contiguous(), gb[i].contiguous(), plan.sample_task, plan.n, plan.num_tasks, co);
        auto fr = expert_frags(K5, K3, dt, need_dx[i] != 0);
        e.wf = fr.first;
        e.wd = fr.second;
      } else {
        e.rows = plan.nslots;
        std::tie(e.g, e.wf, e.wd) = gatrep_merge_gate(K5, K3, k1[i].contiguous(), a3[i].contiguous(), a5[i].contiguous(), gw[i].contiguous(),
                                                      gb[i].contiguous(), plan.slot_task, plan.nslots, plan.num_tasks, dt, need_dx[i] != 0);
      }
      e.ev = g_prep_events[i];
      RM_HIP_CHECK(hipEventRecord(e.ev, ss.prep.stream()));
      std::lock_guard<std::mutex> lock(g_prep_mu);
      g_prep[K5.data_ptr()] = e;
    }
  } catch (...) {
    c10::hip::setCurrentHIPStream(main_s);
    throw;
  }
  c10::hip::setCurrentHIPStream(main_s);
}
// Forget prepared filters nobody took (the end of a forward pass).  Orders the current stream behind the preparation
// stream first, so that no launch of it is left running unobserved (and a capturing stream is re-joined).
void op_finish_prepared(const Tensor& like) {
  bool any;
  {
    std::lock_guard<std::mutex> lock(g_prep_mu);
    any = !g_prep.empty();
    g_prep.clear();
  }
  {
    std::lock_guard<std::mutex> lock(g_k2_prep_mu);
    g_k2_prep.clear();
  }
  if (!any || !like.is_cuda() || !g_overlap) return;
  const int dev = like.device().index();
  SideStreams& ss = side_streams(dev);
  RM_HIP_CHECK(hipEventRecord(ss.prep_fork_ev, ss.prep.stream()));
  RM_HIP_CHECK(hipStreamWaitEvent(c10::hip::getCurrentHIPStream(dev).stream(), ss.prep_fork_ev, 0));
}
void op_set_bn_epilogue(int64_t mask) { g_bn_epilogue = mask; }
void op_set_unmerged_max_w(int64_t w) { g_unmerged_max_w = w; }
void op_set_dual_launch(bool on) { g_dual_launch = on; }
void op_set_deep_conv(bool on) { g_deep = on; }
void op_set_deep_mode(int64_t mask) { g_deep_mode = mask; }
int64_t op_get_deep_mode() { return g_deep_mode; }
void op_set_deep_fwd_min(int64_t v) { g_deep_fwd_min = v; }
void op_set_thin_kernels(bool on) { g_thin = on; }
bool op_get_deep_conv() { return g_deep; }
int64_t op_get_unmerged_max_w() { return g_unmerged_max_w; }
void op_set_overlap(bool on) { g_overlap = on; }
void op_set_tail_jobs(bool on) { g_tail = on; }
bool op_get_tail_jobs() { return g_tail; }
void op_set_prepare(bool on) { g_prepare = on; }
bool op_get_overlap() { return g_overlap; }

void op_zero_pool_begin(const std::string& key, const Tensor& like) {
  if (like.is_cuda()) {       // start of a train step: nothing a failed step deferred may ride in this step's convs
    DeviceGuard guard(like.device());
    RM_CALL(repmode_tail_discard, stream_handle());
  }
  g_pool.begin(key, like);
}
void op_zero_pool_end() { g_pool.end(); }
bool op_zero_pool_has_plan(const std::string& key) {
  std::lock_guard<std::mutex> lock(g_pool.mu);
  auto it = g_pool.plans.find(key);
  return it != g_pool.plans.end() && !it->second.empty();
}
std::tuple<Tensor, bool> op_zero_pool_take(std::vector<int64_t> shape, const Tensor& like) {
  auto tk = g_pool.take(shape, like);
  return std::make_tuple(tk.first, tk.second);
}
Tensor op_grad_out(const Tensor& param) { return grad_out(param, nullptr); }
int64_t op_eval_cache_size() {
  std::lock_guard<std::mutex> lock(g_eval_mu);
  return (int64_t)g_eval.size();
}
// diagnostics: host microseconds per launch of a tiny library kernel (n back-to-back launches on the current stream), and
// per allocation of a small tensor -- what a launch / an output buffer costs the host on this box
double op_debug_launch_cost(const Tensor& t, int64_t n) {
  require_hip(t, "input");
  Tensor y5 = at::zeros({4, 8, 5}, t.options().dtype(at::kFloat)), y = at::empty({4, 8}, t.options().dtype(at::kFloat));
  void* s = stream_handle();
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < n; ++i) RM_CALL(repmode_unshift5, y5.data_ptr<float>(), y.data_ptr<float>(), 4L, 8, s);
  const auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / (double)n;
}
double op_debug_alloc_cost(const Tensor& t, int64_t n) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < n; ++i) { Tensor a = at::empty({1024}, t.options()); (void)a; }
  const auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / (double)n;
}
void op_set_fork_max_w(int64_t w) { g_fork_max_w = w; }
int64_t op_get_fork_max_w() { return g_fork_max_w; }
void op_eval_cache_begin() {
  std::lock_guard<std::mutex> lock(g_eval_mu);
  ++g_eval_depth;
}
void op_eval_cache_end() {
  std::lock_guard<std::mutex> lock(g_eval_mu);
  if (g_eval_depth > 0 && --g_eval_depth == 0) g_eval.clear();
}
void op_grad_sink_set(const std::vector<Tensor>& params, const std::vector<Tensor>& flats, const std::vector<int64_t>& offsets) {
  TORCH_CHECK(params.size() == flats.size() && params.size() == offsets.size(), "grad_sink_set: list lengths differ");
  std::lock_guard<std::mutex> lock(g_sink_mu);
  g_sink.clear();
  for (size_t i = 0; i < params.size(); ++i) g_sink[params[i].data_ptr()] = SinkEntry{params[i], flats[i], offsets[i]};
}
void op_grad_sink_clear() {
  std::lock_guard<std::mutex> lock(g_sink_mu);
  g_sink.clear();
}

}  // namespace rm

TORCH_LIBRARY(repmode, m) {
  m.def("mode_conv3d(Tensor x_cl, Tensor? x2_cl, Tensor k5, Tensor k3, Tensor k1, Tensor a3, Tensor a5, Tensor gate_w, Tensor gate_b, "
        "Tensor slot_task, Tensor sample_slot, Tensor sample_task, int nslots, int num_tasks, bool training, int task0, bool out_f32, "
        "int mode) -> Tensor", &rm::op_mode_conv3d);
  m.def("mode_block(Tensor x, Tensor? x2, Tensor k5, Tensor k3, Tensor k1, Tensor a3, Tensor a5, Tensor gate_w, Tensor gate_b, "
        "Tensor? bn_w, Tensor? bn_b, Tensor? bn_rm, Tensor? bn_rv, bool bn_batch_stats, float bn_momentum, float bn_eps, "
        "Tensor slot_task, Tensor sample_slot, Tensor sample_task, int nslots, int num_tasks, bool training, int task0, int dtype, "
        "bool out_f32) -> Tensor", &rm::op_mode_block);
  m.def("bn_relu(Tensor x_cl, Tensor weight, Tensor bias, Tensor running_mean, Tensor running_var, bool batch_stats, float momentum, "
        "float eps, int out_dtype) -> Tensor", &rm::op_bn_relu);
  m.def("down2(Tensor x_cl, Tensor weight) -> Tensor", &rm::op_down2);
  m.def("up2(Tensor x_cl, Tensor weight) -> Tensor", &rm::op_up2);
  m.def("stage2_bn_relu(Tensor x, Tensor weight, Tensor bn_w, Tensor bn_b, Tensor bn_rm, Tensor bn_rv, bool bn_batch_stats, "
        "float bn_momentum, float bn_eps, bool up, int out_dtype) -> Tensor", &rm::op_stage2_bn_relu);
  m.def("mse_loss(Tensor out, Tensor target, Tensor sample_task, int num_tasks) -> (Tensor, Tensor, Tensor, Tensor)", &rm::op_mse_loss);
  m.def("prepare_filters(Tensor[] k5, Tensor[] k3, Tensor[] k1, Tensor[] a3, Tensor[] a5, Tensor[] gate_w, Tensor[] gate_b, int[] w_in, "
        "int[] need_dx, Tensor slot_task, Tensor sample_slot, Tensor sample_task, int nslots, int num_tasks, bool training, "
        "int dtype) -> ()", &rm::op_prepare_filters);
  m.def("finish_prepared(Tensor like) -> ()", &rm::op_finish_prepared);
  m.def("prepare_stage_filters(Tensor[] weights, int[] up, int dtype, bool both) -> ()", &rm::op_prepare_stage_filters);
  m.def("adam_step(Tensor[] params, Tensor[] grads, Tensor[] exp_avgs, Tensor[] exp_avg_sqs, float lr, float beta1, float beta2, "
        "float eps, int step) -> ()", &rm::op_adam_step);
  m.def("adam_step_dev(Tensor[] params, Tensor[] grads, Tensor[] exp_avgs, Tensor[] exp_avg_sqs, float lr, float beta1, float beta2, "
        "float eps, Tensor(a!) step_dev, Tensor(b!) hyper_dev) -> ()", &rm::op_adam_step_dev);
  m.def("clear_frag_store() -> ()", &rm::op_clear_frag_store);
  m.def("pinned_operands() -> Tensor[]", &rm::op_pinned_operands);
  m.def("set_frag_store(bool on) -> ()", &rm::op_set_frag_store);
  m.def("frag_store_size() -> int", &rm::op_frag_store_size);
  m.def("set_frag_verify(bool on) -> ()", &rm::op_set_frag_verify);
  m.def("set_bn_epilogue(int mask) -> ()", &rm::op_set_bn_epilogue);
  m.def("set_unmerged_max_w(int w) -> ()", &rm::op_set_unmerged_max_w);
  m.def("set_dual_launch(bool on) -> ()", &rm::op_set_dual_launch);
  m.def("set_deep_conv(bool on) -> ()", &rm::op_set_deep_conv);
  m.def("set_deep_mode(int mask) -> ()", &rm::op_set_deep_mode);
  m.def("get_deep_mode() -> int", &rm::op_get_deep_mode);
  m.def("set_deep_fwd_min(int v) -> ()", &rm::op_set_deep_fwd_min);
  m.def("set_thin_kernels(bool on) -> ()", &rm::op_set_thin_kernels);
  m.def("get_deep_conv() -> bool", &rm::op_get_deep_conv);
  m.def("get_unmerged_max_w() -> int", &rm::op_get_unmerged_max_w);
  m.def("set_overlap(bool on) -> ()", &rm::op_set_overlap);
  m.def("set_tail_jobs(bool on) -> ()", &rm::op_set_tail_jobs);
  m.def("get_tail_jobs() -> bool", &rm::op_get_tail_jobs);
  m.def("set_prepare(bool on) -> ()", &rm::op_set_prepare);
  m.def("get_overlap() -> bool", &rm::op_get_overlap);
  m.def("zero_pool_begin(str key, Tensor like) -> ()", &rm::op_zero_pool_begin);
  m.def("zero_pool_end() -> ()", &rm::op_zero_pool_end);
  m.def("zero_pool_has_plan(str key) -> bool", &rm::op_zero_pool_has_plan);
  m.def("zero_pool_take(int[] shape, Tensor like) -> (Tensor, bool)", &rm::op_zero_pool_take);
  m.def("grad_out(Tensor param) -> Tensor", &rm::op_grad_out);
  m.def("eval_cache_size() -> int", &rm::op_eval_cache_size);
  m.def("debug_launch_cost(Tensor t, int n) -> float", &rm::op_debug_launch_cost);
  m.def("debug_alloc_cost(Tensor t, int n) -> float", &rm::op_debug_alloc_cost);
  m.def("set_fork_max_w(int w) -> ()", &rm::op_set_fork_max_w);
  m.def("get_fork_max_w() -> int", &rm::op_get_fork_max_w);
  m.def("eval_cache_begin() -> ()", &rm::op_eval_cache_begin);
  m.def("eval_cache_end() -> ()", &rm::op_eval_cache_end);
  m.def("grad_sink_set(Tensor[] params, Tensor[] flats, int[] offsets) -> ()", &rm::op_grad_sink_set);
  m.def("grad_sink_clear() -> ()", &rm::op_grad_sink_clear);
}
