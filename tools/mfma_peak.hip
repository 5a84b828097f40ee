// Sustained dense-MFMA ceiling of the device under its power cap: a register-only loop of
// v_mfma_f32_32x32x16_bf16 (4 independent accumulators per wave, 2 waves per SIMD -- the shape of the conv
// kernel's inner loop without any data movement).  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool RANDOM>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters, unsigned seed) {
  // RANDOM: operands with random mantissa/sign bits and exponents near 1.0 (bf16 values in +-[0.5, 2)), the
  // toggle rate of real activations; otherwise near-constant words (an optimistic, low-power case)
  u32x4 a, b;
  if (RANDOM) {
    unsigned h = (threadIdx.x + 1u) * 2654435761u ^ seed * 40503u;
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (h & 0x80ff80ffu) | 0x3f003f00u; };
    a = u32x4{rnd(), rnd(), rnd(), rnd()};
    b = u32x4{rnd(), rnd(), rnd(), rnd()};
  } else {
    a = u32x4{seed + threadIdx.x, seed * 3u, 0x3f803f80u, 0x3f803f80u};
    b = u32x4{0x3f803f80u, seed + 7u * threadIdx.x, 0x3f803f80u, seed};
  }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, a), c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, b), c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4);
  const int iters = 4000, blocks = 512 * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 8; ++rep) {
    const int launches = 400;
    const bool random_ops = rep >= 4;
    (void)hipEventRecord(e0);
    for (int l = 0; l < launches; ++l) {
      if (random_ops) mfma_loop<true><<<blocks, 256>>>(out, iters, 1u + l);
      else mfma_loop<false><<<blocks, 256>>>(out, iters, 1u + l);
    }
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 16 * 4.0 * iters * 4 /*waves*/ * blocks * launches;
    printf("rep %d (%s operands): %.1f ms, %.1f TFLOP/s\n", rep, random_ops ? "random" : "near-constant", ms, fl / ms / 1e9);
  }
  return 0;
}
