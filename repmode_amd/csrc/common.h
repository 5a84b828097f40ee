// Shared device/host helpers for librepmode_hip.so (gfx950 only; no other targets).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "repmode_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;  // storage type of a bfloat16 element

void repmode_set_error(const char* fmt, ...);

#define RM_REQUIRE(cond, ...)           \
  do {                                  \
    if (!(cond)) {                      \
      repmode_set_error(__VA_ARGS__);   \
      return REPMODE_EINVAL;            \
    }                                   \
  } while (0)

#define RM_LAUNCH_CHECK(what)                                                \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      repmode_set_error("%s: %s", what, hipGetErrorString(e__));             \
      return REPMODE_ELAUNCH;                                                \
    }                                                                        \
  } while (0)

#define RM_HIP(call)                                                         \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess) {                                                 \
      repmode_set_error("%s: %s", #call, hipGetErrorString(e__));            \
      return REPMODE_ELAUNCH;                                                \
    }                                                                        \
  } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN kept quiet
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> packed bf16 pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 (the software form above is ~10
// VALU operations per element)
typedef __bf16 rm_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rm_bf16x2));
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }

// f(std::integral_constant<int, I>{}) for I = lo .. n - 1: loops whose index must be a compile-time constant in the body
// (`if constexpr` plans of which instruction goes where; register arrays that must never be indexed at run time)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// The dispatcher places workgroup b on XCD b % 8 (observed, speed only).  Remap so that each
// XCD receives a contiguous range of logical ids: neighbouring bricks then share one L2.
// Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// Optional per-launch timing (HIP events on the launch stream), switched on by repmode_prof_enable().
// kind: REPMODE_PROF_* ; work: algorithmic FLOPs (conv kernels) or bytes (GatRep kernels) of the launch.
void repmode_prof_begin(int kind, double work, hipStream_t s);
// Library-owned, per (device, stream) scratch of REPMODE_ZERO_SCRATCH_FLOATS floats that is ALL ZERO whenever no
// library call is executing on that stream: kernels that accumulate into it with atomics put the zeros back
// themselves ("last workgroup cleans up"), so no call pays a memset launch.  nullptr on failure (error string set).
constexpr size_t REPMODE_ZERO_SCRATCH_FLOATS = 64 * 1024;
float* repmode_zero_scratch(hipStream_t s);
// Regions of the scratch: two halves that the BatchNorm reductions use alternately (the reduction kernel of one
// call clears the half the previous call used, so no kernel is needed to put zeros back), and the gate-gradient
// accumulator.  repmode_bn_scratch_half returns this call's half index (0/1) and flips the per-stream state.
constexpr size_t REPMODE_SCRATCH_BN_HALF = 16 * 1024;                       // floats per BatchNorm half
constexpr size_t REPMODE_SCRATCH_GATE_OFF = 2 * REPMODE_SCRATCH_BN_HALF;    // gate accumulator: the upper 32 K floats
// behind the floats: the two counters of the grid-wide barrier of the one-launch BatchNorm passes (bnrelu.hip: arrivals,
// departures; the last workgroup to leave puts the zeros back)
constexpr size_t REPMODE_SCRATCH_BARRIER_OFF = REPMODE_ZERO_SCRATCH_FLOATS;
constexpr size_t REPMODE_SCRATCH_TAIL_WORDS = 512;
// (words [0, 320) of the tail: the barrier; [384, 384 + REPMODE_GATREP_MULTI_MAX): the operand check's sticky verdicts, gatrep.hip)
constexpr size_t REPMODE_SCRATCH_VERIFY_WORD = 384;
int repmode_bn_scratch_half(hipStream_t s);
void repmode_prof_end(hipStream_t s);

// REPMODE_DETERMINISTIC=1 / repmode_set_deterministic(1): run-to-run bitwise reproducible results -- every float sum gets a
// fixed order (no input-channel split of the convolutions, one workgroup per filter-gradient tile, ordered in-workgroup
// reductions, one writer per BatchNorm partial-sum slice ...) at the price of the parallelism those splits buy.  A float
// atomicAdd stays where at most two addends meet on a zeroed location (a + b is commutative; three are not associative).
bool repmode_deterministic();
// how many workgroups may add to one element of a cleared output in deterministic mode: 2 (a + b commutes), or 1 for the
// sites named in REPMODE_DET_SINGLE (a bit mask, developer switch): 1 conv5 split, (2 gemm3: always 1,) 4 k2s2, 8 loss,
// 16 expert_mix, 32 BatchNorm, 64 filter gradient
enum { RM_DET_CONV = 1, RM_DET_GEMM3 = 2, RM_DET_K2S2 = 4, RM_DET_MSE = 8, RM_DET_MIX = 16, RM_DET_BN = 32, RM_DET_WGRAD = 64 };
int repmode_det_cap(int site);
// CUs the persistent grids leave free (api.hip: REPMODE_RESERVE_CUS / repmode_set_reserve_cus)
int repmode_reserve_cus();

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
