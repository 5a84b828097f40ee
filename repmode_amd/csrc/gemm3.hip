// gemm3.hip -- the three 1x1x1 experts of the per-expert formulation as small batched GEMMs on the matrix cores.
//
// By linearity (property 3) the conv1x1 / avg3x3 / avg5x5 experts of a MoDE block
// (fnet/nn_modules/RepMode.py:135-142, 175-180) applied to a sample are three 1x1 convolutions of x, box3(x) and
// box5(x): plain GEMMs over the voxel rows.  The deep levels need them in three shapes (e = expert 2, 3, 4):
//
//   forward        P_e [M][Co]  = X_e [M][Ci]   * W_e [Co][Ci]^T          K = Ci   (both operands K-contiguous)
//   filter grad    dW_e[Co][Ci] = G_e [M][Co]^T * X_e [M][Ci]             K = M    (both operands K-strided)
//   data grad      T_e [M][Ci]  = G_e [M][Co]   * W_e [Co][Ci]            K = Co   (A K-contiguous, B K-strided)
//
// One kernel covers them through element strides: C_b[m][n] = sum_k A_b(m, k) * B_b(n, k) for b = 0..2, with
// A_b(m, k) at a[b] + m * a_ms + k * a_ks and B_b(n, k) at b[b] + n * b_ns + k * b_ks.  float32 in and out; the products on
// v_mfma_f32_16x16x4_f32 (exact f32: parity mode) or, operands rounded to bf16 while staging, v_mfma_f32_16x16x32_bf16
// (throughput mode, like every other kernel of it).  These matrices are a few MB and live in L2 / Infinity Cache; the
// launches are latency-bound.
//
// Workgroup = 4 waves = a 64 x 64 tile of C; K in steps of 16 through LDS (a thread's global loads run along whichever
// of the operand's two axes is contiguous).  Grid: (N tiles, M tiles, 3).
#include "common.h"

namespace {

#ifndef GEMM3_GK
#define GEMM3_GK 32
#endif
constexpr int GT = 64, GK = GEMM3_GK, GPAD = 1;
constexpr int GLD = GT * GK / 256;   // elements per operand, K step and thread
static_assert(GK % 32 == 0, "the bf16 MFMA takes 32 K values");

struct Gemm3Args {
  const float* a[3];
  const float* b[3];
  float* c[3];
  long a_ms, a_ks, b_ns, b_ks;
  int M, N, K, ldc;
  int ksplit;      // > 1: blockIdx.z = matrix * ksplit + K slice, results ADDED to C (which the caller cleared)
};

// BF16: operands rounded to bfloat16 while staging, v_mfma_f32_16x16x32_bf16 (16x the f32 rate) -- the throughput mode,
// whose other kernels multiply bf16 operands with f32 accumulation as well; !BF16: exact float32 (parity mode).
// VEC (BF16, operand A K-contiguous: the forward and data-gradient shapes; the host checks every extent and non-unit stride
// a multiple of 4, 16-byte aligned matrices below 2 GiB): an operand's 64 x 32 step is TWO 16-byte buffer loads per thread
// along its contiguous axis instead of eight predicated 4-byte loads; out-of-range pieces (ragged last tile, K beyond this
// slice) are an out-of-bounds OFFSET, which returns 0 without a branch; and the next step is fetched unconditionally (past the
// end: all offsets out of bounds, no traffic), so the compiler can count the loads in flight instead of waiting for all of
// them.  These launches are chains of a few K steps: their time is the instructions between two barriers.  Same box, GPU time
// per launch (tools/sessions/r4_session19.sh): forward shapes 18.6 -> 14.7, 20.8 -> 13.5, 10.0 -> 8.5 us, data gradient 17.8 ->
// 15.4, 14.8 -> 13.6; the filter-gradient shape (both operands K-strided: four rows per lane land on 4 LDS banks) loses
// (17.0 -> 18.7) and keeps the scalar loader.
template <bool BF16, bool VEC = false>
__global__ __launch_bounds__(256) void gemm3_kernel(Gemm3Args g) {
  static_assert(!VEC || BF16, "the vector path stages bf16");
  constexpr int KP = GK + 8;                                                   // bf16 row: K contiguous, 16 bytes of padding
  __shared__ __attribute__((aligned(16))) float As[BF16 ? GT * KP / 2 : GK * (GT + GPAD)];      // f32: [k][m]; bf16: [m][k]
  __shared__ __attribute__((aligned(16))) float Bs[BF16 ? GT * KP / 2 : GK * (GT + GPAD)];
  bf16_t* Ab = reinterpret_cast<bf16_t*>(As);
  bf16_t* Bb = reinterpret_cast<bf16_t*>(Bs);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bz = blockIdx.z / g.ksplit, kz = blockIdx.z % g.ksplit;
  const float* __restrict__ A = g.a[bz];
  const float* __restrict__ B = g.b[bz];
  float* __restrict__ C = g.c[bz];
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;      // this wave's 32 x 32 quadrant
  const int l15 = lane & 15, kq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool a_kfast = g.a_ks == 1, b_kfast = g.b_ks == 1;
  // 64 x 32 elements per operand and K step, 8 per thread; consecutive threads run along the operand's contiguous axis.
  // The loads of the next step are issued before the MFMAs of this one (register prefetch).  These launches are
  // latency-bound (a few MB from L2 / Infinity Cache, tens of workgroups): the K range is split over workgroups until the
  // grid fills the chip, each slice adding its part with float atomics.
  const int ksteps = (g.K + GK - 1) / GK;
  const int k_begin = (int)((long)kz * ksteps / g.ksplit) * GK, k_end = min(g.K, (int)((long)(kz + 1) * ksteps / g.ksplit) * GK);
  auto mma_bf16 = [&]() {
#pragma unroll
    for (int ks = 0; ks < GK; ks += 32) {
      bf16x8 af[2], bf[2];                  // lane: row / column lane & 15, k = 8 (lane >> 4) .. + 7
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + (wm + i * 16 + l15) * KP + ks + 8 * kq);
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(Bb + (wn + j * 16 + l15) * KP + ks + 8 * kq);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  if constexpr (VEC) {
    constexpr uint32_t OOB = 0x80000000u;
    constexpr int NV = GT * GK / 4 / 256;            // float4 pieces per operand, K step and thread (2)
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, 0x7fffffff, 0x00020000);
    // this thread's pieces: K-contiguous operand -> (row f / 8, k 4 (f % 8)); row-contiguous -> (rows 4 (f % 16), k f / 16)
    int prow[2][NV], pk[2][NV];
    uint32_t poff[2][NV];                            // byte offset at k0 = 0 (OOB: the rows are outside the matrix)
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int f = u * 256 + tid;
      prow[0][u] = a_kfast ? f / (GK / 4) : (f % (GT / 4)) * 4;  pk[0][u] = a_kfast ? (f % (GK / 4)) * 4 : f / (GT / 4);
      prow[1][u] = b_kfast ? f / (GK / 4) : (f % (GT / 4)) * 4;  pk[1][u] = b_kfast ? (f % (GK / 4)) * 4 : f / (GT / 4);
      poff[0][u] = m0 + prow[0][u] < g.M ? (uint32_t)(((long)(m0 + prow[0][u]) * g.a_ms + (long)pk[0][u] * g.a_ks) * 4) : OOB;
      poff[1][u] = n0 + prow[1][u] < g.N ? (uint32_t)(((long)(n0 + prow[1][u]) * g.b_ns + (long)pk[1][u] * g.b_ks) * 4) : OOB;
    }
    const uint32_t kstep_a = (uint32_t)(g.a_ks * 4), kstep_b = (uint32_t)(g.b_ks * 4);      // bytes per unit of k
    f32x4 va[NV], vb[NV];
    auto fetchv = [&](int k0) {
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        va[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                  rsa, (poff[0][u] != OOB && k0 + pk[0][u] < k_end) ? poff[0][u] + (uint32_t)k0 * kstep_a : OOB, 0, 0));
        vb[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                  rsb, (poff[1][u] != OOB && k0 + pk[1][u] < k_end) ? poff[1][u] + (uint32_t)k0 * kstep_b : OOB, 0, 0));
      }
    };
    auto put = [&](bf16_t* dst, bool kfast, int row, int k, const f32x4& v) {
      if (kfast) {        // four consecutive k of one row: one 8-byte write
        *reinterpret_cast<u32x2*>(dst + row * KP + k) = u32x2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
      } else {            // four consecutive rows at one k
        const uint32_t lo = pack_bf16x2(v.x, v.y), hi = pack_bf16x2(v.z, v.w);
        dst[(row + 0) * KP + k] = (bf16_t)(lo & 0xffffu); dst[(row + 1) * KP + k] = (bf16_t)(lo >> 16);
        dst[(row + 2) * KP + k] = (bf16_t)(hi & 0xffffu); dst[(row + 3) * KP + k] = (bf16_t)(hi >> 16);
      }
    };
    fetchv(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += GK) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        put(Ab, a_kfast, prow[0][u], pk[0][u], va[u]);
        put(Bb, b_kfast, prow[1][u], pk[1][u], vb[u]);
      }
      __syncthreads();
      fetchv(k0 + GK);
      mma_bf16();
    }
  } else {
  float ra[GLD], rb[GLD];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < GLD; ++u) {
      const int e = u * 256 + tid;
      const int m = a_kfast ? e / GK : e % GT, ka = a_kfast ? e % GK : e / GT;
      const int gm = m0 + m, gka = k0 + ka;
      ra[u] = (gm < g.M && gka < k_end) ? A[(size_t)gm * g.a_ms + (size_t)gka * g.a_ks] : 0.f;
      const int n = b_kfast ? e / GK : e % GT, kb = b_kfast ? e % GK : e / GT;
      const int gn = n0 + n, gkb = k0 + kb;
      rb[u] = (gn < g.N && gkb < k_end) ? B[(size_t)gn * g.b_ns + (size_t)gkb * g.b_ks] : 0.f;
    }
  };
  fetch(k_begin);
  for (int k0 = k_begin; k0 < k_end; k0 += GK) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GLD; ++u) {
      const int e = u * 256 + tid;
      const int ka = a_kfast ? e % GK : e / GT, ma = a_kfast ? e / GK : e % GT;
      const int kb = b_kfast ? e % GK : e / GT, nb = b_kfast ? e / GK : e % GT;
      if constexpr (BF16) {
        Ab[ma * KP + ka] = f32_to_bf16(ra[u]);
        Bb[nb * KP + kb] = f32_to_bf16(rb[u]);
      } else {
        As[ka * (GT + GPAD) + ma] = ra[u];
        Bs[kb * (GT + GPAD) + nb] = rb[u];
      }
    }
    __syncthreads();
    if (k0 + GK < k_end) fetch(k0 + GK);
    if constexpr (BF16) {
      mma_bf16();
    } else {
#pragma unroll
      for (int ks = 0; ks < GK; ks += 4) {
        float af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = As[(ks + kq) * (GT + GPAD) + wm + i * 16 + l15];      // A[i = row][k]
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = Bs[(ks + kq) * (GT + GPAD) + wn + j * 16 + l15];      // B[k][j = column]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  }
  // 16x16 C/D layout: column = lane & 15, row = (lane >> 4) * 4 + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 16 + l15;
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + kq * 4 + r;
        if (m < g.M) {
          if (g.ksplit > 1) unsafeAtomicAdd(C + (size_t)m * g.ldc + n, acc[i][j][r]);
          else C[(size_t)m * g.ldc + n] = acc[i][j][r];
        }
      }
    }
}

}  // namespace

extern "C" int repmode_gemm3(const float* const* a, long a_ms, long a_ks, const float* const* b, long b_ns, long b_ks,
                             float* const* c, int ldc, int m, int n, int k, int c_is_zero, int bf16_mfma, void* stream) {
  RM_REQUIRE(a && b && c, "gemm3: null pointer");
  RM_REQUIRE(m > 0 && n > 0 && k > 0 && ldc >= n, "gemm3: bad shape");
  Gemm3Args g{};
  for (int i = 0; i < 3; ++i) {
    RM_REQUIRE(a[i] && b[i] && c[i], "gemm3: null matrix %d", i);
    g.a[i] = a[i]; g.b[i] = b[i]; g.c[i] = c[i];
  }
  g.a_ms = a_ms; g.a_ks = a_ks; g.b_ns = b_ns; g.b_ks = b_ks;
  g.M = m; g.N = n; g.K = k; g.ldc = ldc;
  // split K until ~512 workgroups, at least two K steps each; only when the caller vouches for a cleared C
  const long tiles = (long)((n + GT - 1) / GT) * ((m + GT - 1) / GT) * 3;
  int ks = 1;
  // (deterministic mode: no split.  Two slices -- two addends on a cleared C, which commute -- did NOT reproduce bitwise
  // here (tools/r3_session12.sh: the only site of the seven that failed with two), so this one stays whole)
  const int ks_max = repmode_deterministic() ? 1 : 32;
  static const long split_target = []() { const char* e = getenv("REPMODE_GEMM3_TARGET"); return e ? atol(e) : 512L; }();
  if (c_is_zero) while (tiles * ks < split_target && (k + GK - 1) / GK >= 4 * ks && ks < ks_max) ks *= 2;
  g.ksplit = ks;
  const dim3 grid((n + GT - 1) / GT, (m + GT - 1) / GT, 3 * ks);
  // the vector path's conditions (see the kernel)
  auto mult4 = [](long v) { return (v & 3) == 0; };
  bool vec = bf16_mfma && a_ks == 1 && mult4(m) && mult4(n) && mult4(k) && mult4(a_ms) &&
             (b_ks == 1 ? mult4(b_ns) : (b_ns == 1 && mult4(b_ks))) &&
             ((long)m * a_ms + (long)k * a_ks) * 4 < (1L << 31) && ((long)n * b_ns + (long)k * b_ks) * 4 < (1L << 31);
  for (int i = 0; i < 3; ++i) vec = vec && (((uintptr_t)a[i] | (uintptr_t)b[i]) & 15) == 0;
  repmode_prof_begin(REPMODE_PROF_HELPER, 12.0 * ((double)m * k + (double)n * k + (double)m * n), static_cast<hipStream_t>(stream));
  if (vec) hipLaunchKernelGGL((gemm3_kernel<true, true>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), g);
  else if (bf16_mfma) hipLaunchKernelGGL(gemm3_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), g);
  else hipLaunchKernelGGL(gemm3_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), g);
  repmode_prof_end(static_cast<hipStream_t>(stream));
  RM_LAUNCH_CHECK("gemm3");
  return REPMODE_OK;
}
