// Issue rate of v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 in the filter gradient's shape: ONE MFMA wave per
// SIMD (a 256-thread workgroup per CU), 25 independent 16x16 accumulators fed from one A operand and five B operands (the
// five x shifts of a window) -- cycles per MFMA by s_memtime, with 0 .. 3 v_perm_b32 between the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/build/mfma_rate && tools/build/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned rnd(unsigned& h) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x80ff80ffu) | 0x3f003f00u; }

template <int FILL, bool BIG, int KIND = 0>
__global__ __launch_bounds__(256, 1) void rate_kernel(float* out, long long* cycles, int iters, unsigned seed) {
  unsigned h = (threadIdx.x + 1u) * 2654435761u ^ seed * 40503u;
  u32x4 a = {rnd(h), rnd(h), rnd(h), rnd(h)};
  u32x4 b[5];
  for (int j = 0; j < 5; ++j) b[j] = u32x4{rnd(h), rnd(h), rnd(h), rnd(h)};
  unsigned f0 = rnd(h), f1 = rnd(h), sfill = seed;
  f32x4 acc[25];
  f32x16 big[6];
  for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) big[t][r] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (BIG) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[t]) : "v"(a), "v"(b[t % 5]));
#pragma unroll
        for (int k = 0; k < FILL; ++k) { f0 = __builtin_amdgcn_perm(f0, f1, 0x05040100u + k); __builtin_amdgcn_sched_barrier(0); }
      }
    } else {
#pragma unroll
      for (int t = 0; t < 25; ++t) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b[t % 5]));   // (the builtin's loop gets accumulator copies and s_nops from the compiler)
#pragma unroll
        for (int k = 0; k < FILL; ++k) {
          if constexpr (KIND == 0) f0 = __builtin_amdgcn_perm(f0, f1, 0x05040100u + k);
          else if constexpr (KIND == 1) asm volatile("s_nop 0");
          else if constexpr (KIND == 2) asm volatile("v_nop");
          else if constexpr (KIND == 3) asm volatile("s_nop 3");
          else asm volatile("s_add_u32 %0, %0, 1" : "+s"(sfill));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  f0 += sfill;
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = (float)f0;
  for (int t = 0; t < 25; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int t = 0; t < 6; ++t) for (int r = 0; r < 16; ++r) s += big[t][r];
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int FILL, bool BIG, int KIND = 0>
void run(float* out, long long* cyc, const char* what) {
  const int iters = 400;
  for (int l = 0; l < 3; ++l) rate_kernel<FILL, BIG, KIND><<<256, 256>>>(out, cyc, iters, 1u + l);
  (void)hipDeviceSynchronize();
  long long c = 0;
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int per = BIG ? 6 : 25;
  printf("%-18s %d v_perm per MFMA: %.2f cycles per MFMA (%.0f flop per cycle and SIMD)\n", what, FILL, (double)c / (iters * per),
         (BIG ? 32768.0 : 16384.0) * iters * per / (double)c);
}

int main() {
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, 4);
  (void)hipMalloc(&cyc, 8);
  run<0, false>(out, cyc, "16x16x32 bf16"); run<1, false>(out, cyc, "16x16x32 bf16"); run<2, false>(out, cyc, "16x16x32 bf16");
  run<3, false>(out, cyc, "16x16x32 bf16"); run<4, false>(out, cyc, "16x16x32 bf16");
  run<0, true>(out, cyc, "32x32x16 bf16"); run<2, true>(out, cyc, "32x32x16 bf16"); run<4, true>(out, cyc, "32x32x16 bf16");
  run<6, true>(out, cyc, "32x32x16 bf16");
  run<1, false, 1>(out, cyc, "16x16 + s_nop 0"); run<2, false, 1>(out, cyc, "16x16 + s_nop 0");
  run<1, false, 2>(out, cyc, "16x16 + v_nop"); run<2, false, 2>(out, cyc, "16x16 + v_nop");
  run<1, false, 3>(out, cyc, "16x16 + s_nop 3");
  run<1, false, 4>(out, cyc, "16x16 + s_add"); run<2, false, 4>(out, cyc, "16x16 + s_add");
  return 0;
}
