// conv5_deep.hip -- the per-expert formulation's two convolutions on the deep U-Net levels (x extent <= 8) as ONE
// uniform grid: every workgroup runs the 5x5x5 expert's 125 taps AND the 3x3x3 expert's 27 taps over the same staged
// halo image, for several samples at a time, with the experts' filters shared by the whole batch.
//
// Replaces, for the blocks that take the per-expert form (SURVEY.md section 4 property 3, RepMode.py:171-192 by
// linearity:  y[n] = sum_e g[n,e,:] * conv(x[n], K_e)), the two F.conv3d-shaped terms of RepMode.py:204-208
//     forward        P5[n] = conv(x[n], K5),  P3[n] = conv(x[n], pad(K3))              (one input, two outputs)
//     data gradient  dx[n] = conv(G5[n], flip(K5)^T) + conv(G3[n], flip(pad(K3))^T)     (two inputs, one output)
// which repmode_conv5_ex's "dual-expert launch" ran as two jobs of very different length (125 x 5 against 9 x 5 taps) in
// the two halves of one grid: with the contiguous workgroup -> XCD map the short jobs landed on XCDs 4-7 and idled while
// XCDs 0-3 held 82 % of the work, every XCD read every filter byte, and the level-4 tile fed ONE MFMA per 1 KiB filter
// fragment.  Here
//   * a workgroup = (filter class c = (output-channel tile, input-channel slice), group g of SU spatial units); all
//     workgroups do the same work, so there is nothing to balance;
//   * class c runs on XCD c % 8 (when the class count allows), its groups consecutively: a filter byte crosses the
//     fabric into ONE L2 and is shared by the samples there (8 x less HBM / Infinity-Cache -> L2 filter traffic);
//   * level 4 (2 x 4 x 4 bricks, 32 voxels a sample) puts SU = 4 (forward) / 2 (data gradient) samples into one
//     workgroup's GEMM M dimension: a filter fragment feeds 4 / 2 MFMAs instead of 1;
//   * the 3x3x3 pass takes dx in 1..3 only (27 taps, not the 45 of the dz/dy-restricted general kernel).
// Same MFMA / LDS scheme as conv5_igemm.hip (voxel-linear halo planes, tap shift = constant offset, filter fragments
// straight from L2 a row of taps ahead); bf16 only (the float32 parity mode keeps the general kernel); float output
// accumulated with atomics over the input-channel slices (y must be zero on entry).
#include "common.h"
#include "tail_jobs.h"

#include <atomic>
#include <cstdlib>

namespace {

struct DeepArgs {
  const bf16_t* x;   // forward: [N][D][H][W][Cin]; data gradient: [2 N][D][H][W][Cin] (job 0 reads n, job 1 reads N + n)
  const bf16_t* w;   // [2][125][CoutP/32][CinP/16][32][16]: slot 0 = 5x5x5 expert, slot 1 = padded 3x3x3 expert (centre rows)
  float* y;          // forward: [2 N][D][H][W][Cout] (P5 then P3); data gradient: [N][D][H][W][Cout]
  int N, D, H, W, Cin, Cout, CinP, CoutP;
  int nbz, nby, nbx, nbricks;   // bricks of one sample
  int nunits, G;                // spatial units (sample, brick) and groups of SU of them
  int ncot, ksplit;             // output-channel tiles, input-channel slices
  int xcd_classes;              // class count is a multiple of 8: class c -> XCD c % 8
  TailJobs tail;                // deferred small jobs riding in this launch (tail_jobs.h)
};

// BZ x BY x BX bricks; SU spatial units per workgroup; WV x WC waves (voxel x channel); VW 32-voxel sub-tiles per wave
template <int BZ_, int BY_, int BX_, int SU_, int WV_, int WC_, int VW_>
struct DCfg {
  static constexpr int BZ = BZ_, BY = BY_, BX = BX_, SU = SU_, WV = WV_, WC = WC_, VW = VW_;
  static constexpr int NV = BZ * BY * BX;
  static constexpr int TU = NV / 32;            // 32-voxel sub-tiles of one unit
  static constexpr int COT = 32 * WC;
  static constexpr int NT = 64 * WV * WC;
  static constexpr int BZH = BZ + 4, BYH = BY + 4, BXH = BX + 4;
  static constexpr int VH = BZH * BYH * BXH;
  static constexpr int PLS = ((VH + 7) / 8) * 8 + 4;   // plane stride in 16-byte slots (== 4 mod 8: see conv5_igemm.hip)
  static constexpr int IMG = 2 * PLS;                  // one unit's halo image: two planes (channel groups of 8)
  static_assert(NV % 32 == 0 && SU * TU == WV * VW, "sub-tiles of the group = sub-tiles of the waves");
};

__device__ __forceinline__ void mma_bf16(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One expert's taps over the staged images: P = 0 the 5x5x5 expert (dx 0..4), P = 1 the 3x3x3 expert (dx 1..3).
// vb[vs]: LDS slot of sub-tile vs' voxel at tap (0,0,0); wrow: this lane's filter fragment at tap 0 of the chunk.
template <typename C, int P>
__device__ __forceinline__ void tap_pass(const u32x4* __restrict__ lds, const int (&vb)[C::VW], const bf16_t* __restrict__ wrow,
                                         size_t tap_stride, int dz_lo, int dz_hi, int dy_lo, int dy_hi, f32x16 (&acc)[C::VW]) {
  constexpr int VW = C::VW, BYH = C::BYH, BXH = C::BXH;
  constexpr int DX0 = P ? 1 : 0, NDX = P ? 3 : 5;
  if (dz_lo > dz_hi || dy_lo > dy_hi) return;
  auto wfrag = [&](int tap) -> u32x4 { return *reinterpret_cast<const u32x4*>(wrow + (size_t)tap * tap_stride); };
  const int nrows = (dz_hi - dz_lo + 1) * (dy_hi - dy_lo + 1);
  int dz = dz_lo, dy = dy_lo;
  u32x4 a_cur[NDX], a_nxt[NDX], b_cur[VW], b_nxt[VW];
#pragma unroll
  for (int i = 0; i < NDX; ++i) a_cur[i] = wfrag((dz * 5 + dy) * 5 + DX0 + i);
  {
    const int off0 = (dz * BYH + dy) * BXH + DX0;
#pragma unroll
    for (int vs = 0; vs < VW; ++vs) b_cur[vs] = lds[vb[vs] + off0];
  }
  for (int row = 0; row < nrows; ++row) {
    int dzn = dz, dyn = dy + 1;
    if (dyn > dy_hi) { dyn = dy_lo; dzn = dz + 1; }
    const bool more = row + 1 < nrows;
    if (more) {
#pragma unroll
      for (int i = 0; i < NDX; ++i) a_nxt[i] = wfrag((dzn * 5 + dyn) * 5 + DX0 + i);
    }
    const int rowoff = (dz * BYH + dy) * BXH + DX0;
    const int rowoff_n = more ? (dzn * BYH + dyn) * BXH + DX0 : rowoff;
#pragma unroll
    for (int i = 0; i < NDX; ++i) {
      const int offn = (i < NDX - 1) ? rowoff + i + 1 : rowoff_n;
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) b_nxt[vs] = lds[vb[vs] + offn];
      // (fences: the next tap's LDS reads stay AHEAD of this tap's MFMAs -- left alone, the scheduler sinks each read to
      // just before the MFMA that consumes it, and with one or two waves per SIMD nothing covers the LDS latency)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) mma_bf16(b_cur[vs], a_cur[i], acc[vs]);      // A = voxels, B = filter rows
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) b_cur[vs] = b_nxt[vs];
    }
#pragma unroll
    for (int i = 0; i < NDX; ++i) a_cur[i] = a_nxt[i];
    dz = dzn;
    dy = dyn;
  }
}

template <typename C, bool TWO_IN>
__global__ __launch_bounds__(C::NT, 2) void conv5_deep_kernel(DeepArgs a) {
  constexpr int KV = 8, KC = 16;
  constexpr int BZ = C::BZ, BY = C::BY, BX = C::BX, SU = C::SU, VW = C::VW, TU = C::TU;
  constexpr int BYH = C::BYH, BXH = C::BXH, VH = C::VH, PLS = C::PLS, IMG = C::IMG, NT = C::NT;
  constexpr int NIN = TWO_IN ? 2 : 1, NOUT = TWO_IN ? 1 : 2;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wv = wave % C::WV, wc = wave / C::WV;
  const int khalf = lane >> 5, l31 = lane & 31;

  if (a.tail.nblocks) {
    if ((int)blockIdx.x < a.tail.nblocks) {
      tail_run(a.tail, blockIdx.x, tid, reinterpret_cast<float*>(smem));
      return;
    }
  }
  const int conv_block = blockIdx.x - a.tail.nblocks, conv_blocks = gridDim.x - a.tail.nblocks;
  int c, g;
  if (a.xcd_classes) {
    // workgroup b runs on XCD b % 8 (observed; speed only): class c = 8 k + xcd, its groups consecutive in time
    const int xcd = conv_block & 7, j = conv_block >> 3;
    c = (j / a.G) * 8 + xcd;
    g = j % a.G;
  } else {
    const int id = xcd_remap(conv_block, conv_blocks);
    c = id / a.G;
    g = id % a.G;
  }
  const int kz = c % a.ksplit, cot = c / a.ksplit;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, CinP = a.CinP, CoutP = a.CoutP;
  const int N = a.N;
  const size_t vol = (size_t)D * H * W;

  // ---- geometry of this lane's voxel in each of its sub-tiles
  int vb[VW];
#pragma unroll
  for (int vs = 0; vs < VW; ++vs) {
    const int t = wv * VW + vs, ul = t / TU, piece = t % TU;
    const int m = piece * 32 + l31;
    const int lx = m % BX, ly = (m / BX) % BY, lz = m / (BX * BY);
    vb[vs] = ul * IMG + khalf * PLS + (lz * BYH + ly) * BXH + lx;
  }
  const int nkc = CinP / KC, nrt = CoutP / 32;
  const size_t tap_stride = (size_t)CoutP * CinP;
  const int rt = min(cot * (C::COT / 32) + wc, nrt - 1);      // (tile wider than the filter: clamped, never stored)
  const bf16_t* __restrict__ wrow0 = a.w + (size_t)rt * nkc * (32 * KC) + l31 * KC + khalf * KV;

  // taps whose input plane / row is padding for the whole brick are skipped -- when every unit of the group has the
  // same brick origin (one brick per sample: the deep levels of the network); otherwise the zero-filled halo does it
  int dz_lo = 0, dz_hi = 4, dy_lo = 0, dy_hi = 4;
  if (a.nbricks == 1) {
    dz_lo = max(0, 2 - (BZ - 1)); dz_hi = min(4, D + 1);
    dy_lo = max(0, 2 - (BY - 1)); dy_hi = min(4, H + 1);
  }

  f32x16 acc[NOUT][VW];
#pragma unroll
  for (int p = 0; p < NOUT; ++p)
#pragma unroll
    for (int vs = 0; vs < VW; ++vs)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][vs][r] = 0.f;

  const int nchunks = CinP / KC;
  const int c_begin = (int)((long)kz * nchunks / a.ksplit);
  const int c_end = (int)((long)(kz + 1) * nchunks / a.ksplit);

  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int ci0 = chunk * KC;
    __syncthreads();      // all waves finished reading the previous chunk's images
    // ---- stage the halo images: item = (input, unit, halo voxel, plane)
    constexpr int NITEMS = NIN * SU * 2 * VH;
    constexpr int UNR = (NITEMS + NT - 1) / NT >= 9 ? 9 : 4;
    for (int it0 = 0; it0 < NITEMS; it0 += NT * UNR) {
      u32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int it = it0 + u * NT + tid;
        v[u] = u32x4{0u, 0u, 0u, 0u};
        if (it < NITEMS) {
          const int img = it / (2 * VH), r = it % (2 * VH);
          const int jin = img / SU, ul = img % SU;
          const int unit = g * SU + ul;
          const int pl = r & 1, vh = r >> 1;
          const int xx = vh % BXH, t2 = vh / BXH;
          const int yy = t2 % BYH, zz = t2 / BYH;
          if (unit < a.nunits) {
            const int n = unit / a.nbricks, br = unit % a.nbricks;
            const int bx = br % a.nbx, by = (br / a.nbx) % a.nby, bz = br / (a.nbx * a.nby);
            const int gz = bz * BZ + zz - 2, gy = by * BY + yy - 2, gx = bx * BX + xx - 2;
            const int cch = ci0 + pl * KV;
            if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && cch < Cin)
              v[u] = *reinterpret_cast<const u32x4*>(a.x + ((size_t)(jin * N + n) * vol + (size_t)(gz * H + gy) * W + gx) * Cin + cch);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int it = it0 + u * NT + tid;
        if (it < NITEMS) {
          const int img = it / (2 * VH), r = it % (2 * VH);
          lds[img * IMG + (r & 1) * PLS + (r >> 1)] = v[u];
        }
      }
    }
    __syncthreads();
    // ---- the two experts' taps.  Forward: both read image set 0 and keep their own accumulators; data gradient: expert p
    // reads image set p (its gate-scaled output gradient) and both add into one accumulator.
    const bf16_t* wchunk = wrow0 + (size_t)chunk * (32 * KC);
    {
      int vbp[VW];
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) vbp[vs] = vb[vs];
      tap_pass<C, 0>(lds, vbp, wchunk, tap_stride, dz_lo, dz_hi, dy_lo, dy_hi, acc[0]);
    }
    {
      int vbp[VW];
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) vbp[vs] = vb[vs] + (TWO_IN ? SU * IMG : 0);
      tap_pass<C, 1>(lds, vbp, wchunk + (size_t)REPMODE_TAPS * tap_stride, tap_stride, max(dz_lo, 1), min(dz_hi, 3), max(dy_lo, 1),
                     min(dy_hi, 3), acc[TWO_IN ? 0 : 1]);
    }
  }

  // ---- epilogue: 32x32 C/D layout, rows = voxels ((r & 3) + 8 (r >> 2) + 4 khalf), column = this lane's output channel:
  // the 32 lanes of a half-wave add 128 contiguous bytes
  const int co = cot * C::COT + wc * 32 + l31;
  if (co >= Cout) return;
#pragma unroll
  for (int p = 0; p < NOUT; ++p) {
#pragma unroll
    for (int vs = 0; vs < VW; ++vs) {
      const int t = wv * VW + vs, ul = t / TU, piece = t % TU;
      const int unit = g * SU + ul;
      if (unit >= a.nunits) continue;
      const int n = unit / a.nbricks, br = unit % a.nbricks;
      const int bx = br % a.nbx, by = (br / a.nbx) % a.nby, bz = br / (a.nbx * a.nby);
      float* yn = a.y + (size_t)(p * N + n) * vol * Cout + co;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = piece * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        const int lx = m % BX, ly = (m / BX) % BY, lz = m / (BX * BY);
        const int gz = bz * BZ + lz, gy = by * BY + ly, gx = bx * BX + lx;
        if (gz >= D || gy >= H || gx >= W) continue;
#ifdef RM_CONV_NOEPI
        if (acc[p][vs][r] == 12345.678f)      // TIMING BUILD ONLY: the sums are computed and (practically) never written
#endif
        unsafeAtomicAdd(yn + ((size_t)(gz * H + gy) * W + gx) * Cout, acc[p][vs][r]);
      }
    }
  }
}

// smallest input-channel split that gives at least `target` workgroups (REPMODE_DEEP_TARGET, default 512 = two per CU)
static const int g_deep_target = []() { const char* e = getenv("REPMODE_DEEP_TARGET"); return e ? atoi(e) : 512; }();
static const bool g_deep_target_set = getenv("REPMODE_DEEP_TARGET") != nullptr;      // (a sweep: no second-guessing of the split)

template <typename C, bool TWO_IN>
int launch_deep(DeepArgs a, bool y_is_zero, hipStream_t stream) {
  a.nbz = ceil_div(a.D, C::BZ);
  a.nby = ceil_div(a.H, C::BY);
  a.nbx = ceil_div(a.W, C::BX);
  a.nbricks = a.nbz * a.nby * a.nbx;
  a.nunits = a.N * a.nbricks;
  a.G = ceil_div(a.nunits, C::SU);
  a.ncot = ceil_div(a.CoutP, C::COT);
  const int nchunks = a.CinP / 16;
  int ks = 1;
  while (ks < nchunks && (long)a.G * a.ncot * ks < g_deep_target) ks *= 2;
  if (ks > nchunks) ks = nchunks;
  // One brick per workgroup (level 3) and a single 16-channel chunk per slice: half the slices, two chunks each, on one
  // workgroup per CU -- the same MFMA work per CU, half the float atomics of the epilogue, which are 35-50 % of such a launch
  // (profiles/r04_split_atomics.txt).  Same box, 256 -> 256 at 4 x 8 x 8, batch 8: data gradient 63.2 -> 51.8 us, forward
  // 90.6 -> 64.3.  With more chunks per workgroup the staging latency of a lone workgroup costs more than the atomics save
  // (512 -> 256: 87 -> 107 us), and the level-4 bricks lose as well (33.8 -> 40.2): both keep two workgroups per CU.
  if (C::SU == 1 && !g_deep_target_set && ks >= 2 && nchunks / ks == 1 && (long)a.G * a.ncot * ks >= 512) ks /= 2;
  a.ksplit = ks;
  const long nclass = (long)a.ncot * ks;
  a.xcd_classes = (nclass % 8 == 0) ? 1 : 0;
  repmode_tail_take(stream, &a.tail);
  const long grid = nclass * a.G + a.tail.nblocks;
  RM_REQUIRE(grid > 0 && grid < (1L << 31), "conv5_deep: grid %ld out of range", grid);
  constexpr int IMG_BYTES = (TWO_IN ? 2 : 1) * C::SU * C::IMG * 16;
  constexpr int LDS_MAX = IMG_BYTES > TAIL_LDS_BYTES ? IMG_BYTES : TAIL_LDS_BYTES;
  const int lds_bytes = a.tail.nblocks ? LDS_MAX : IMG_BYTES;
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  if (!((attr_set.load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_deep_kernel<C, TWO_IN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               LDS_MAX));
    attr_set.fetch_or(1u << (dev & 31), std::memory_order_release);
  }
  if (!y_is_zero)
    RM_HIP(hipMemsetAsync(a.y, 0, (size_t)(TWO_IN ? 1 : 2) * a.N * a.D * a.H * a.W * a.Cout * sizeof(float), stream));
  // algorithmic FLOPs = the layer's merged 125-tap convolution, once per layer and direction; the 3x3x3
  // expert's 27 taps are executed work, not algorithmic
  const double alg = 2.0 * a.N * a.D * a.H * a.W * (double)a.Cin * a.Cout * REPMODE_TAPS;
  repmode_prof_begin(REPMODE_PROF_CONV5_DEEP, alg, stream);
  hipLaunchKernelGGL((conv5_deep_kernel<C, TWO_IN>), dim3((unsigned)grid), dim3(C::NT), lds_bytes, stream, a);
  repmode_prof_end(stream);
  RM_LAUNCH_CHECK("conv5_deep");
  return REPMODE_OK;
}

// Tile menu.            BZ BY BX  SU WV WC VW
using DCfgX8 = DCfg<4, 8, 8, 1, 2, 2, 4>;      // level 3: one 256-voxel brick x 64 channels
using DCfgX4F = DCfg<2, 4, 4, 4, 1, 4, 4>;     // level 4 forward: four 32-voxel bricks x 128 channels
using DCfgX4D = DCfg<2, 4, 4, 2, 1, 4, 2>;     // level 4 data gradient (two image sets): two bricks x 128 channels

}  // namespace

extern "C" int repmode_conv5_deep_supported(int wdim, int cin, int dtype) {
  if (repmode_deterministic()) return 0;      // (its input-channel slices add with atomics: the general kernel without a split)
  return (dtype == REPMODE_BF16 && wdim > 0 && wdim <= 8 && cin > 0 && cin % 8 == 0) ? 1 : 0;
}

// flags bit 0: data-gradient form (x holds 2 n samples, y n); else forward form (x n samples, y 2 n).  Bit 1: y is zero
// already (else cleared here).
extern "C" int repmode_conv5_deep(const void* x, const void* w, float* y, int n, int d, int h, int wdim, int cin, int cout, int flags,
                                  void* stream) {
  RM_REQUIRE(x && w && y, "conv5_deep: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0, "conv5_deep: bad shape");
  RM_REQUIRE(repmode_conv5_deep_supported(wdim, cin, REPMODE_BF16), "conv5_deep: x extent %d / %d input channels not supported (x extent <= 8, channels %% 8 == 0)",
             wdim, cin);
  RM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0, "conv5_deep: pointers must be 16-byte aligned");
  DeepArgs a{};
  a.x = static_cast<const bf16_t*>(x);
  a.w = static_cast<const bf16_t*>(w);
  a.y = y;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin; a.Cout = cout;
  a.CinP = repmode_padded_channels(cin, REPMODE_BF16, 1);
  a.CoutP = repmode_padded_channels(cout, REPMODE_BF16, 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool two_in = (flags & 1) != 0, zero = (flags & 2) != 0;
  if (wdim <= 4 && h <= 4 && d <= 2) {
    if (two_in) return launch_deep<DCfgX4D, true>(a, zero, s);
    return launch_deep<DCfgX4F, false>(a, zero, s);
  }
  if (two_in) return launch_deep<DCfgX8, true>(a, zero, s);
  return launch_deep<DCfgX8, false>(a, zero, s);
}
