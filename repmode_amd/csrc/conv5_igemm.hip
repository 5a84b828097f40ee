// conv5_igemm.hip -- 5x5x5 "same" 3-D cross-correlation with a per-sample (per-slot) filter as an
// implicit GEMM on the gfx950 matrix cores.
//
// Replaces the per-sample F.conv3d loop of the reference (fnet/nn_modules/RepMode.py:204-208, and
// the single-filter eval form :209-210) and, fed with the flipped/transposed filter, the input
// gradient of its autograd.
//
// GEMM view:  Y[co][voxel] = sum_{tap, ci} Wm[tap][co][ci] * X[voxel + tap][ci]
//   MFMA "A" operand = filter rows (32 output channels), "B" operand = 32 voxels, K = input
//   channels of one tap.  bf16: one v_mfma_f32_32x32x16_bf16 per 16 input channels; f32: four
//   v_mfma_f32_32x32x2_f32 per 8 input channels (exact f32).  Either way a lane's fragment of
//   one operand is 16 bytes (8 bf16 / 4 f32 of consecutive input channels), so both element
//   types share every byte offset below.
//
// Data movement (all kernels of this file):
//   * activations are NDHWC; a workgroup owns a BZ x BY x BX brick of output voxels of one sample
//     and, per chunk of KC input channels, stages the (BZ+4)(BY+4)(BX+4) halo brick once into LDS
//     (zero filled outside the volume) and reuses it for all 125 taps.
//   * LDS image: two planes per chunk, plane p = 16-byte channel group p of every halo voxel,
//     voxel-linear.  Lane l of a wave reads voxel (l & 31) of plane (l >> 5): 32 consecutive
//     16-byte slots -> conflict-free ds_read_b128, and a tap shift is a constant byte offset.
//   * filter fragments come straight from global/L2 (fragment-major layout: each 32 x KC tile is
//     1 KiB contiguous = one fully coalesced wave load).
//
// Three kernels:
//   conv5_igemm_kernel  the general one: every dtype, tile, epilogue (float output with split-K atomics, bias + ReLU,
//                       BatchNorm statistics, pair / dual-expert launches); two workgroups per CU that alternate
//                       "stage an image" / "125 taps" / epilogue.  Since round 3 it runs levels 2-4 and the eval /
//                       float32 paths.  (Its ROWSTAT / MERGE instantiations and the RM_CONV_* macros are measured
//                       experiments)
//   conv5_ws_kernel     round 3, the wide levels (x extent >= 32, bf16, element-typed output: 3/4 of the network's conv
//                       FLOPs): one workgroup per CU as a software pipeline of four MFMA waves and four loader waves over
//                       a double-buffered image -- see the comment above it.
//   conv5_pipe_kernel   the same pipeline without loader waves (the step before; kept for A/B and its timing builds).
#include "common.h"
#include "tail_jobs.h"

#include <atomic>
#include <cstdlib>

namespace {

template <typename T>
struct Elem;

template <>
struct Elem<float> {
  static constexpr int KV = 4;  // elements per 16-byte fragment
  __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

template <>
struct Elem<bf16_t> {
  static constexpr int KV = 8;
  __device__ static __forceinline__ void mma(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

#ifdef RM_CONV_TIMING
// developer build only (REPMODE_EXTRA_FLAGS=-DRM_CONV_TIMING): shader-clock stamps of the first workgroups'
// phases, read back with repmode_debug_conv_timing (tools/conv_phase_timing.py)
__device__ unsigned long long g_conv_timing[64 * 64];
#define RM_STAMP(slot)                                                                        \
  do {                                                                                        \
    if (tid == 0 && blockIdx.x < 64 && (slot) < 64)                                           \
      g_conv_timing[blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memtime();                 \
  } while (0)
#else
#define RM_STAMP(slot) do {} while (0)
#endif

// Experiment (REPMODE_EXTRA_FLAGS=-DRM_CONV_SCHED, tools/ab_variant.sh): without the fence the machine scheduler
// sinks half of the "one tap ahead" LDS reads to just before the MFMA that consumes them (taps 1 and 3 of every row:
// ds_read; s_waitcnt lgkmcnt(1); v_mfma), so those MFMAs wait out the LDS latency unless the SIMD's other wave covers.
// Measured on level 0 (same box, sustained): 226.6 vs 227.1 us per launch -- the other wave does cover; off by default.
#ifdef RM_CONV_SCHED
#define RM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define RM_SCHED_FENCE() do {} while (0)
#endif

// Experiment (REPMODE_EXTRA_FLAGS=-DRM_CONV_PRIO): s_setprio 1 for the MFMA (tap) phase of a chunk, 0 for its staging
// phase -- with two workgroups per CU one is usually staging while the other multiplies (cdna_hip_programming.md T5).
#ifdef RM_CONV_PRIO
#define RM_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define RM_PRIO(p) do {} while (0)
#endif

// conv5_ws_kernel on 16-voxel bricks, one channel sub-tile per wave: filter rows / voxel-fragment taps in flight ahead
#ifndef RM_WS16_RA
#define RM_WS16_RA 6
#endif
#ifndef RM_WS16_TA
#define RM_WS16_TA 3
#endif
#ifndef CONV_SPLIT_TARGET
#define CONV_SPLIT_TARGET 512
#endif
struct ConvArgs {
  const void* x;
  const void* w;
  const int32_t* sample_slot;
  void* y;
  int N, D, H, W, Cin, Cout, CinP, CoutP;
  int nbz, nby, nbx, ncot, ksplit;
  int ksplit2;         // dual-expert launch (round 4): the 3x3x3 job's own split factor (0: one factor for both jobs, round 3's grid)
  int out_f32;  // 0: store as T; 1: float output (SWAP kernels; atomicAdd when ksplit > 1)
  int tap_lo, tap_hi;  // dz and dy are restricted to [tap_lo, tap_hi] (0..4: full filter; 1..3: a 3x3 support)
  int accum;           // float output only: add to y (atomics, y is not cleared) instead of overwriting it
  int dxc;             // only the centre x tap (dx = 2) of every (dz, dy) row: see repmode_conv5_ex
  // skip connections without a concatenated tensor (repmode_conv5_pair): channels [0, Cin1) of the input come from
  // x, [Cin1, Cin) from x2 (Cin1 == 0: one input); output channels [0, Cout1) go to y, the rest to y2 (Cout1 == 0: one)
  const void* x2;
  void* y2;
  int Cin1, Cout1;
  // epilogue extras (repmode_conv5_epi).  bias / relu: y = max(acc + bias[co], 0) -- an eval-mode BatchNorm folded into
  // the merged filter (scale, in the gate probabilities) and this bias (RepMode.py:209-212).  stats: per-channel sum and
  // sum of squares of the STORED (rounded) outputs, added to slice blockIdx % 16 of a [16][2 Cout] float array -- the
  // batch statistics of the training-mode BatchNorm that follows (RepMode.py:146-149, 212) without a read pass;
  // stats_clear: 16 K floats this launch puts back to zero (the BatchNorm scratch half the previous call used).
  const float* bias;
  int relu;
  float* stats;
  float* stats_clear;
  // dual-expert launch (per-expert formulation of the deep levels): 2 N "virtual samples" in one grid -- virtual sample
  // v < N is sample v with slot 0 and all 125 taps (the 5x5x5 expert), v >= N is sample v - N with slot 1 and the
  // centred 3x3x3 support (the padded 3x3x3 expert); sample_slot is not read.  dual & 1: on; dual & 2: the input holds
  // 2 N samples (the two gate-scaled output gradients), else both jobs read sample v % N; dual & 4: the output holds
  // 2 N samples (the two expert outputs), else both jobs ADD into sample v % N.
  int dual;
  int wide;            // element-typed bf16 output: 16-byte stores through v_permlane32_swap (see the epilogue)
  int zfast;           // conv5_ws_kernel: a workgroup's items run along z first (channel tile, z, x, y, sample) instead of x first
  // MERGE kernels only: w = the experts' two un-merged slots (repmode_expert_frags), the 1x1 experts' parameters [Cout][Cin]
  // and the gate probabilities g[slot][5][Cout]
  const float* mk1;
  const float* ma3;
  const float* ma5;
  const float* gates;
  // deferred small jobs (tail_jobs.h) that ride in this launch: workgroups [0, tail.nblocks) run them, the convolution's
  // workgroups follow (tail.nblocks is a multiple of 8, so their workgroup -> XCD map is unchanged)
  TailJobs tail;
};

// Tile configuration.  BZ*BY*BX output voxels = 32 * WV * VW; 32 * WC * CW output channels.
template <int BZ_, int BY_, int BX_, int WV_, int WC_, int VW_, int CW_>
struct Cfg {
  static constexpr int BZ = BZ_, BY = BY_, BX = BX_, WV = WV_, WC = WC_, VW = VW_, CW = CW_;
  static constexpr int NV = BZ * BY * BX;
  static constexpr int COT = 32 * WC * CW;
  static constexpr int NT = 64 * WV * WC;
  static constexpr int BZH = BZ + 4, BYH = BY + 4, BXH = BX + 4;
  static constexpr int VH = BZH * BYH * BXH;
  // plane stride in 16-byte units; == 4 (mod 8) so that the two planes of one voxel land in
  // different halves of the 128-byte ds_write_b128 bank window
  static constexpr int PLS = ((VH + 7) / 8) * 8 + 4;
  static constexpr int LDS_BYTES = 2 * PLS * 16;
  static_assert(NV == 32 * WV * VW, "voxel tile must be WV*VW MFMA columns blocks");
};

// SWAP selects the MFMA operand roles.  false: A = filter rows, B = voxels -> a lane ends up with 4
// consecutive output channels of one voxel per register quad (8/16-byte channel-contiguous stores; used
// for the element-typed output).  true: A = voxels, B = filter rows -> a lane holds ONE output channel
// for 16 voxels, so the 32 lanes of a half-wave cover 32 consecutive channels of a voxel: 128-byte
// contiguous float stores / atomics (used for the float output, in particular the split-K atomics,
// which otherwise touch one cache line per lane).
// PAIR: the two-tensor form (repmode_conv5_pair); a separate instantiation keeps the one-tensor kernels free of its selects
// DXC: the dx-centre mode (one tap per (dz, dy) row: the thin first / last layers) as its own instantiation, so that
// neither path carries the other's registers and branches
// ROWSTAT: the row-stationary tap loop (4 x 4 x 32 bricks only, see the loop's comment)
// MERGE: the filter fragment is MERGED IN THE KERNEL from the experts (the A/B experiment of repmode_conv5_merged below)
#ifndef RM_CONV_X16_WAVES
#define RM_CONV_X16_WAVES 2     // (experiment: -DRM_CONV_X16_WAVES=3 asks for three waves per SIMD on the 4 x 4 x 16 tile)
#endif
template <typename T, typename C, bool SWAP, bool PAIR, bool DXC = false, bool ROWSTAT = false, bool MERGE = false>
__global__ __launch_bounds__(C::NT, (C::BX == 16 && !MERGE) ? RM_CONV_X16_WAVES : 2) void conv5_igemm_kernel(ConvArgs a) {
  constexpr int KV = Elem<T>::KV;
  constexpr int KC = 2 * KV;
  constexpr int BZ = C::BZ, BY = C::BY, BX = C::BX, VW = C::VW, CW = C::CW;
  constexpr int BYH = C::BYH, BXH = C::BXH, VH = C::VH, PLS = C::PLS, NT = C::NT;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wv = wave % C::WV;
  const int wc = wave / C::WV;
  const int khalf = lane >> 5;
  const int l31 = lane & 31;

  if (a.tail.nblocks) {
    if ((int)blockIdx.x < a.tail.nblocks) {
      tail_run(a.tail, blockIdx.x, tid, reinterpret_cast<float*>(smem));
      return;
    }
  }
  const int conv_block = blockIdx.x - a.tail.nblocks, conv_blocks = gridDim.x - a.tail.nblocks;
  int bid = xcd_remap(conv_block, conv_blocks);
  // Dual-expert launch with per-job split factors (round 4): the 5x5x5 job is 125 taps, the centred 3x3x3 job 45 -- with one
  // factor for both, the 3x3x3 job's workgroups finish in a third of the launch and their CUs idle.  Here a (brick, channel
  // tile) owns ksplit + ksplit2 consecutive workgroups: the first ksplit split the 5x5x5 job's channel chunks, the others
  // the 3x3x3 job's (neighbours: they stage the same halo bricks, one XCD's L2 serves both).
  const bool two_ks = a.dual && a.ksplit2 > 0;
  const int per_unit = two_ks ? a.ksplit + a.ksplit2 : a.ksplit;
  const int j = bid % per_unit;  bid /= per_unit;
  const bool second_ks = two_ks && j >= a.ksplit;
  const int ksplit = second_ks ? a.ksplit2 : a.ksplit;
  const int kz = second_ks ? j - a.ksplit : j;
  const int cot = bid % a.ncot;   bid /= a.ncot;
  const int bx = bid % a.nbx;     bid /= a.nbx;
  const int by = bid % a.nby;     bid /= a.nby;
  const int bz = bid % a.nbz;
  const int nv = two_ks ? bid / a.nbz + (second_ks ? a.N : 0) : bid / a.nbz;     // (virtual) sample of this workgroup
  const bool second = a.dual && nv >= a.N;               // dual launch: the 3x3x3 expert's job
  const int n = (a.dual && !(a.dual & 2)) ? (second ? nv - a.N : nv) : nv;      // input sample
  const int n_out = (a.dual && !(a.dual & 4)) ? (second ? nv - a.N : nv) : nv;  // output sample
  const int z0 = bz * BZ, y0 = by * BY, x0 = bx * BX;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, CinP = a.CinP, CoutP = a.CoutP;

  const int Cin1 = PAIR ? a.Cin1 : 0, Cout1 = PAIR ? a.Cout1 : 0;
  const int slot = a.dual ? (second ? 1 : 0) : a.sample_slot[n];
  const int tap_lo = second ? 1 : a.tap_lo, tap_hi = second ? 3 : a.tap_hi;
  const T* __restrict__ xn = static_cast<const T*>(a.x) + (size_t)n * D * H * W * (Cin1 > 0 ? Cin1 : Cin);
  const T* __restrict__ xn2 = Cin1 > 0 ? static_cast<const T*>(a.x2) + (size_t)n * D * H * W * (Cin - Cin1) : nullptr;
  // filter layout (fragment-major): [slot][tap][co tile (32)][ci chunk (KC)][32][KC]; a lane's 16 bytes
  // of an A fragment are bytes [16 lane, 16 lane + 16) of the 1 KiB tile
  const T* __restrict__ wsl = static_cast<const T*>(a.w) + (size_t)(MERGE ? 0 : slot) * REPMODE_TAPS * CoutP * CinP;
  const size_t tap_stride = (size_t)CoutP * CinP;
  const int nkc = CinP / KC, nrt = CoutP / 32;

  // halo index of this lane's voxel in each of its voxel sub-tiles (tap (0,0,0))
  int vbase[VW];
#pragma unroll
  for (int vs = 0; vs < VW; ++vs) {
    const int m = (wv * VW + vs) * 32 + l31;
    const int lx = m % BX, ly = (m / BX) % BY, lz = m / (BX * BY);
    vbase[vs] = khalf * PLS + (lz * BYH + ly) * BXH + lx;
  }
  // filter row of this lane in each of its channel sub-tiles
  const T* wrow[CW];
#pragma unroll
  for (int cs = 0; cs < CW; ++cs) {
    // row tiles beyond CoutP (block tile wider than the filter) are clamped; their results are never stored
    const int rt = min(cot * (C::COT / 32) + wc * CW + cs, nrt - 1);
    wrow[cs] = wsl + (size_t)rt * nkc * (32 * KC) + l31 * KC + khalf * KV;
  }

  // taps whose input plane/row lies outside the volume for every voxel of the brick are skipped
  const int dz_lo = max(tap_lo, 2 - z0 - (BZ - 1)), dz_hi = min(tap_hi, D + 1 - z0);
  const int dy_lo = max(tap_lo, 2 - y0 - (BY - 1)), dy_hi = min(tap_hi, H + 1 - y0);
  const int ndy = dy_hi - dy_lo + 1;
  const int nrows = (dz_hi - dz_lo + 1) * ndy;

  f32x16 acc[CW][VW];
#pragma unroll
  for (int cs = 0; cs < CW; ++cs)
#pragma unroll
    for (int vs = 0; vs < VW; ++vs)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cs][vs][r] = 0.f;

  const int nchunks = CinP / KC;
  const int c_begin = (int)((long)kz * nchunks / ksplit);
  const int c_end = (int)((long)(kz + 1) * nchunks / ksplit);
  const bool vec_ok = (Cin % KV) == 0;

#ifdef RM_CONV_DMA
  // Experiment (REPMODE_EXTRA_FLAGS=-DRM_CONV_DMA), measured and OFF: same box, interleaved, us per launch register path /
  // LDS-DMA: 32->32 (level 0) 224.9 / 232.1, 64->32 457.8 / 488.8, 64->64 (level 1) 112.9 / 116.0, 128->64 216.9 / 223.7,
  // level 2 float output 81.4 / 83.5 -- correct (zeros outside the volume included: the whole parity suite passes on it),
  // but a 16-byte-per-lane gather through the DMA path moves fewer bytes per clock than the two register batches.
  // LDS-DMA staging (buffer_load_dwordx4 ... lds): a wave instruction moves 64 halo voxels' 16-byte channel groups from
  // global memory straight into 64 consecutive slots of one plane of the image -- no VGPR round trip, no ds_write pass, all
  // of a wave's instructions of a chunk in flight together (the register path took two dependent batches).  Wave w fills
  // plane w & 1, segments (w >> 1) + (NT / 128) k.  Per lane only the voxel index of each segment is kept (chunk invariant);
  // positions outside the volume get an out-of-range offset: the buffer load returns zeros for them.
  static_assert(VH % 64 == 0 && (NT / 64) % 2 == 0 && (VH / 64) % (NT / 128) == 0, "LDS-DMA staging: whole wave instructions per plane");
  constexpr int DMA_K = (VH / 64) / (NT / 128);          // instructions per wave per chunk
  int dma_vox[DMA_K];
#pragma unroll
  for (int k = 0; k < DMA_K; ++k) {
    const int vh = ((wave >> 1) + (NT / 128) * k) * 64 + lane;
    const int xx = vh % BXH, t2 = vh / BXH;
    const int yy = t2 % BYH, zz = t2 / BYH;
    const int gz = z0 + zz - 2, gy = y0 + yy - 2, gx = x0 + xx - 2;
    dma_vox[k] = ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (gz * H + gy) * W + gx : -1;
  }
#endif
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const int ci0 = chunk * KC;
    // The first filter row of the chunk is requested before the halo staging, not behind its barrier: the two
    // fetches are independent, and after a kernel boundary both come from HBM / MALL, not L2.
    u32x4 a_first[5];
    if constexpr (CW == 1 && !DXC) {
      {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          // (row-stationary loop: the five dy taps of (dz_lo, dx = 0); else the five dx taps of the first row)
          const int tap = ROWSTAT ? (dz_lo * 5 + i) * 5 : (dz_lo * 5 + dy_lo) * 5 + i;
          a_first[i] = *reinterpret_cast<const u32x4*>(wrow[0] + (size_t)tap * tap_stride + (size_t)chunk * (32 * KC));
        }
      }
    }
    RM_STAMP((chunk - c_begin) * 4 + 0);
    __syncthreads();  // all waves finished reading the previous chunk's halo image
    RM_STAMP((chunk - c_begin) * 4 + 1);
    // ---- stage the halo brick: item = (halo voxel, plane), two 16-byte items per voxel
    const bool from2 = Cin1 > 0 && ci0 >= Cin1;
    const T* __restrict__ xsrc = from2 ? xn2 : xn;
    const int csrc = Cin1 > 0 ? (from2 ? Cin - Cin1 : Cin1) : Cin;     // channel stride of the source tensor
    const int cbase = from2 ? Cin1 : 0;                                     // first channel the source holds
    constexpr int NITEMS = 2 * VH;
#ifdef RM_CONV_DMA
    if (vec_ok) {
      const int pl = wave & 1;
      const int c = ci0 + pl * KV;
      const size_t sample_bytes = (size_t)D * H * W * csrc * sizeof(T);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xsrc), 0, (int)sample_bytes, 0x00020000);
      const int chan_off = (c - cbase) * (int)sizeof(T), vox_bytes = csrc * (int)sizeof(T);
      const bool plane_live = c < Cin;
#pragma unroll
      for (int k = 0; k < DMA_K; ++k) {
        const int seg = (wave >> 1) + (NT / 128) * k;
        const int voff = (plane_live && dma_vox[k] >= 0) ? dma_vox[k] * vox_bytes + chan_off : 0x7fffffff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((size_t)pl * PLS + seg * 64) * 16), 16,
                                                 voff, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
#endif
    {
#ifdef RM_CONV_DMA
      constexpr int UNR = 1;     // (only channel counts that are no multiple of 8 (4) come here: one item at a time, few registers)
#else
      constexpr int UNR = (NITEMS + NT - 1) / NT >= 9 ? 9 : 4;   // loads in flight per thread per batch (latency-bound phase)
#endif
      for (int it0 = 0; it0 < NITEMS; it0 += NT * UNR) {
        u32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int it = it0 + u * NT + tid;
          v[u] = u32x4{0u, 0u, 0u, 0u};
          if (it < NITEMS) {
            const int pl = it & 1, vh = it >> 1;
            const int xx = vh % BXH, t2 = vh / BXH;
            const int yy = t2 % BYH, zz = t2 / BYH;
            const int gz = z0 + zz - 2, gy = y0 + yy - 2, gx = x0 + xx - 2;
            const int c = ci0 + pl * KV;
            if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin) {
              // (two-input mode: this chunk's channels lie entirely in one of the two tensors, Cin1 % KC == 0)
              const T* p = xsrc + ((size_t)(gz * H + gy) * W + gx) * csrc + (c - cbase);
              if (vec_ok) {
                v[u] = *reinterpret_cast<const u32x4*>(p);
              } else {
                T e[KV];
#pragma unroll
                for (int k = 0; k < KV; ++k) e[k] = (c + k < Cin) ? p[k] : (T)0;
                v[u] = *reinterpret_cast<const u32x4*>(e);
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int it = it0 + u * NT + tid;
          if (it < NITEMS) lds[(it & 1) * PLS + (it >> 1)] = v[u];
        }
      }
    }
    __syncthreads();
    RM_STAMP((chunk - c_begin) * 4 + 2);
    RM_PRIO(1);     // (RM_CONV_PRIO builds) the tap phase outranks the co-resident workgroup's staging phase

    // ---- 125 taps from the staged image.  Filter fragments are prefetched from L2 one (dz,dy)
    // row (5 taps) ahead when one channel sub-tile is held (CW == 1), one tap ahead otherwise
    // (register budget: two waves per SIMD must fit, i.e. <= 256 VGPR+AGPR).
    auto wfrag = [&](int cs, int tap) -> u32x4 {
      return *reinterpret_cast<const u32x4*>(wrow[cs] + (size_t)tap * tap_stride + (size_t)chunk * (32 * KC));
    };
    int dz = dz_lo, dy = dy_lo;
    if constexpr (DXC && CW == 1) {
      // one tap per (dz, dy) row (the thin first / last layers, whose x taps were folded into channels): a whole dz plane's
      // filter fragments (its <= 5 dy rows) are in flight while the previous plane is multiplied -- one fragment per row,
      // fetched a row ahead, left this path waiting on L2 latency for every tap (first layer: 140 us for 8.4 GFLOP)
      u32x4 a_cur[5], a_nxt[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int dyy = min(dy_lo + i, dy_hi);
        a_cur[i] = wfrag(0, (dz_lo * 5 + dyy) * 5 + 2);
      }
      u32x4 b_cur[VW], b_nxt[VW];
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) b_cur[vs] = lds[vbase[vs] + (dz_lo * BYH + dy_lo) * BXH + 2];
      for (int dzc = dz_lo; dzc <= dz_hi; ++dzc) {
        const bool more_z = dzc < dz_hi;
        if (more_z) {
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const int dyy = min(dy_lo + i, dy_hi);
            a_nxt[i] = wfrag(0, ((dzc + 1) * 5 + dyy) * 5 + 2);
          }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          if (i < ndy) {
            // the voxel fragment of the next row: next dy of this plane, or the first row of the next plane
            const bool last_row = i + 1 >= ndy;
            const int dzn = last_row ? (more_z ? dzc + 1 : dzc) : dzc;
            const int dyn = last_row ? (more_z ? dy_lo : dy_lo + i) : dy_lo + i + 1;
#pragma unroll
            for (int vs = 0; vs < VW; ++vs) b_nxt[vs] = lds[vbase[vs] + (dzn * BYH + dyn) * BXH + 2];
#pragma unroll
            for (int vs = 0; vs < VW; ++vs) {
              if constexpr (SWAP) Elem<T>::mma(b_cur[vs], a_cur[i], acc[0][vs]);
              else Elem<T>::mma(a_cur[i], b_cur[vs], acc[0][vs]);
            }
#pragma unroll
            for (int vs = 0; vs < VW; ++vs) b_cur[vs] = b_nxt[vs];
          }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) a_cur[i] = a_nxt[i];
      }
    } else if constexpr (DXC) {
      // one tap per (dz, dy) row, several channel sub-tiles per wave: one row ahead
      u32x4 a_c[CW], a_n[CW], b_c[VW], b_n[VW];
#pragma unroll
      for (int cs = 0; cs < CW; ++cs) a_c[cs] = wfrag(cs, (dz * 5 + dy) * 5 + 2);
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) b_c[vs] = lds[vbase[vs] + (dz * BYH + dy) * BXH + 2];
      for (int row = 0; row < nrows; ++row) {
        int dzn = dz, dyn = dy + 1;
        if (dyn > dy_hi) { dyn = dy_lo; dzn = dz + 1; }
        const bool more = row + 1 < nrows;
        if (!more) { dzn = dz; dyn = dy; }               // harmless reload on the last row
#pragma unroll
        for (int cs = 0; cs < CW; ++cs) a_n[cs] = wfrag(cs, (dzn * 5 + dyn) * 5 + 2);
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) b_n[vs] = lds[vbase[vs] + (dzn * BYH + dyn) * BXH + 2];
#pragma unroll
        for (int cs = 0; cs < CW; ++cs)
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) {
            if constexpr (SWAP) Elem<T>::mma(b_c[vs], a_c[cs], acc[cs][vs]);
            else Elem<T>::mma(a_c[cs], b_c[vs], acc[cs][vs]);
          }
#pragma unroll
        for (int cs = 0; cs < CW; ++cs) a_c[cs] = a_n[cs];
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) b_c[vs] = b_n[vs];
        dz = dzn;
        dy = dyn;
      }
    } else if constexpr (MERGE) {
      // GatRep INSIDE the conv (RepMode.py:171-192 fused with :204-208, what BASELINE's north star words as "fuse"): the
      // filter fragment of a tap is built in registers from the raw 5x5x5 / 3x3x3 expert fragments (bf16, shared by every
      // sample: slots 0 / 1 of repmode_expert_frags), the three 1x1 experts' values of this lane's (co, 8 ci) and the
      // sample's five gate probabilities:   W = g0 K5 + [centre 27] (g1 K3 + g3 A3 / 27) + [centre] g2 K1 + g4 A5 / 125.
      // No merged filter is written to or read from HBM; the price is ~20 (36 on the 27 centre taps) VALU operations per
      // fragment next to the 4 MFMAs it feeds.
      static_assert(sizeof(T) == 2 && CW == 1, "in-kernel merge: bf16, one channel sub-tile per wave");
      const int co_c = min(cot * C::COT + wc * 32 + l31, Cout - 1);
      const float* gp = a.gates + (size_t)slot * 5 * Cout + co_c;
      const float g0 = gp[0], g1 = gp[Cout], g2 = gp[2 * Cout], g3 = gp[3 * Cout], g4 = gp[4 * Cout];
      float cA[8], cB[8], cC[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ci = ci0 + khalf * 8 + i;
        const bool ok = ci < Cin;
        const size_t o = (size_t)co_c * Cin + (ok ? ci : 0);
        cA[i] = ok ? g4 * (a.ma5[o] * (1.0f / 125.0f)) : 0.f;
        cB[i] = ok ? cA[i] + g3 * (a.ma3[o] * (1.0f / 27.0f)) : 0.f;
        cC[i] = ok ? cB[i] + g2 * a.mk1[o] : 0.f;
      }
      const size_t slot1 = (size_t)REPMODE_TAPS * tap_stride;          // the 3x3x3 expert's slot
      auto merge = [&](const u32x4& r5, const u32x4& r3, int cls) -> u32x4 {      // cls 0: outside the centre 27, 1: inside, 2: the centre tap
        const uint32_t w5[4] = {r5.x, r5.y, r5.z, r5.w}, w3[4] = {r3.x, r3.y, r3.z, r3.w};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float lo = g0 * __uint_as_float(w5[q] << 16), hi = g0 * __uint_as_float(w5[q] & 0xffff0000u);
          if (cls == 0) {
            lo += cA[2 * q]; hi += cA[2 * q + 1];
          } else {
            lo = fmaf(g1, __uint_as_float(w3[q] << 16), lo); hi = fmaf(g1, __uint_as_float(w3[q] & 0xffff0000u), hi);
            lo += (cls == 2 ? cC[2 * q] : cB[2 * q]); hi += (cls == 2 ? cC[2 * q + 1] : cB[2 * q + 1]);
          }
          o[q] = pack_bf16x2(lo, hi);
        }
        return u32x4{o[0], o[1], o[2], o[3]};
      };
      u32x4 a_cur[5], a_nxt[5], c_cur[3], c_nxt[3];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) a_cur[dx] = a_first[dx];
      {
        const bool rowc = dz >= 1 && dz <= 3 && dy >= 1 && dy <= 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) c_cur[j] = u32x4{0u, 0u, 0u, 0u};
        if (rowc) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            c_cur[j] = *reinterpret_cast<const u32x4*>(wrow[0] + slot1 + (size_t)((dz * 5 + dy) * 5 + 1 + j) * tap_stride + (size_t)chunk * (32 * KC));
        }
      }
      u32x4 b_cur[VW], b_nxt[VW];
      {
        const int rowoff0 = (dz * BYH + dy) * BXH;
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) b_cur[vs] = lds[vbase[vs] + rowoff0];
      }
      for (int row = 0; row < nrows; ++row) {
        int dzn = dz, dyn = dy + 1;
        if (dyn > dy_hi) { dyn = dy_lo; dzn = dz + 1; }
        const bool more = row + 1 < nrows;
        const bool rowc = dz >= 1 && dz <= 3 && dy >= 1 && dy <= 3;
        const bool rowc_n = more && dzn >= 1 && dzn <= 3 && dyn >= 1 && dyn <= 3;
        if (more) {
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) a_nxt[dx] = wfrag(0, (dzn * 5 + dyn) * 5 + dx);
          if (rowc_n) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
              c_nxt[j] = *reinterpret_cast<const u32x4*>(wrow[0] + slot1 + (size_t)((dzn * 5 + dyn) * 5 + 1 + j) * tap_stride + (size_t)chunk * (32 * KC));
          }
        }
        const int rowoff = (dz * BYH + dy) * BXH;
        const int rowoff_n = more ? (dzn * BYH + dyn) * BXH : rowoff;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const int offn = (dx < 4) ? rowoff + dx + 1 : rowoff_n;
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_nxt[vs] = lds[vbase[vs] + offn];
          u32x4 wm;
          if (rowc && dx >= 1 && dx <= 3) wm = merge(a_cur[dx], c_cur[dx - 1], (dz == 2 && dy == 2 && dx == 2) ? 2 : 1);
          else wm = merge(a_cur[dx], a_cur[dx], 0);
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) {
            if constexpr (SWAP) Elem<T>::mma(b_cur[vs], wm, acc[0][vs]);
            else Elem<T>::mma(wm, b_cur[vs], acc[0][vs]);
          }
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_cur[vs] = b_nxt[vs];
        }
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) a_cur[dx] = a_nxt[dx];
#pragma unroll
        for (int j = 0; j < 3; ++j) c_cur[j] = c_nxt[j];
        dz = dzn;
        dy = dyn;
      }
    } else if constexpr (ROWSTAT) {
      // Row-stationary order (4 x 4 x 32 bricks: a wave owns the four y rows of one z plane, sub-tile vs = row).  For a
      // fixed (dz, dx) the voxel fragment of halo row y' = vs + dy serves every (vs, dy) pair that lands on it: EIGHT
      // LDS reads (y' = 0..7) and five filter fragments (dy = 0..4) feed TWENTY MFMAs -- 0.4 ds_read_b128 per MFMA
      // instead of 1.0 in the tap-major order below, same filter traffic.  The next group's 8 + 5 fragments are
      // requested before this group's MFMAs (double buffers: 104 registers + 64 accumulators).  Halo rows outside the
      // volume are zero in LDS, so all five dy are always taken (no skipped-tap bookkeeping); dz keeps its range.
      static_assert(CW == 1 && VW == 4 && BY == 4 && BX == 32 && C::WV == 4, "row-stationary loop: 4 x 4 x 32 bricks");
      const int vb0 = vbase[0];
      u32x4 a_cur[5], a_nxt[5], b_cur[8], b_nxt[8];
#pragma unroll
      for (int i = 0; i < 5; ++i) a_cur[i] = a_first[i];
#pragma unroll
      for (int yy = 0; yy < 8; ++yy) b_cur[yy] = lds[vb0 + (dz_lo * BYH + yy) * BXH];
      for (int dzc = dz_lo; dzc <= dz_hi; ++dzc) {
        const bool more_z = dzc < dz_hi;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          // next group: (dzc, dx + 1), or (dzc + 1, 0); on the very last group a harmless reload of this one
          const int dzn = (dx < 4) ? dzc : (more_z ? dzc + 1 : dzc);
          const int dxn = (dx < 4) ? dx + 1 : (more_z ? 0 : 4);
#pragma unroll
          for (int i = 0; i < 5; ++i) a_nxt[i] = wfrag(0, (dzn * 5 + i) * 5 + dxn);
#pragma unroll
          for (int yy = 0; yy < 8; ++yy) b_nxt[yy] = lds[vb0 + (dzn * BYH + yy) * BXH + dxn];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int dyy = 0; dyy < 5; ++dyy)
#pragma unroll
            for (int vs = 0; vs < 4; ++vs) {
              if constexpr (SWAP) Elem<T>::mma(b_cur[vs + dyy], a_cur[dyy], acc[0][vs]);
              else Elem<T>::mma(a_cur[dyy], b_cur[vs + dyy], acc[0][vs]);
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 5; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
          for (int yy = 0; yy < 8; ++yy) b_cur[yy] = b_nxt[yy];
        }
      }
#ifdef RM_CONV_PRE2
    } else if constexpr (CW == 1) {
      // Experiment (REPMODE_EXTRA_FLAGS=-DRM_CONV_PRE2 [-DRM_CONV_BPRE2], tools/ab_variant.sh): the PMC pass of the level-0
      // launch (tools/pmc_conv.sh) shows 31 % of the wave cycles parked in s_waitcnt / barriers and only 4 % stalled on LDS
      // issue -- the filter fragments, requested ONE row (5 taps) ahead from L2, are the suspect.  Here they are requested
      // TWO rows ahead (+20 registers); RM_CONV_BPRE2: the voxel fragments two taps ahead as well (+16).
      u32x4 a_cur[5], a_n1[5], a_n2[5];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) a_cur[dx] = a_first[dx];
      auto next_row = [&](int& z, int& y) { if (++y > dy_hi) { y = dy_lo; ++z; } };
      {
        int z1 = dz, y1 = dy;
        next_row(z1, y1);
        if (nrows > 1) {
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) a_n1[dx] = wfrag(0, (z1 * 5 + y1) * 5 + dx);
        }
      }
#ifdef RM_CONV_BPRE2
      u32x4 b0[VW], b1[VW], b2[VW];
      {
        const int rowoff0 = (dz * BYH + dy) * BXH;
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) { b0[vs] = lds[vbase[vs] + rowoff0]; b1[vs] = lds[vbase[vs] + rowoff0 + 1]; }
      }
#else
      u32x4 b_cur[VW], b_nxt[VW];
      {
        const int rowoff0 = (dz * BYH + dy) * BXH;
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) b_cur[vs] = lds[vbase[vs] + rowoff0];
      }
#endif
      for (int row = 0; row < nrows; ++row) {
        int dzn = dz, dyn = dy;
        next_row(dzn, dyn);
        int dzn2 = dzn, dyn2 = dyn;
        next_row(dzn2, dyn2);
        const bool more = row + 1 < nrows, more2 = row + 2 < nrows;
        if (more2) {
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) a_n2[dx] = wfrag(0, (dzn2 * 5 + dyn2) * 5 + dx);
        }
        const int rowoff = (dz * BYH + dy) * BXH;
        const int rowoff_n = more ? (dzn * BYH + dyn) * BXH : rowoff;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
#ifdef RM_CONV_BPRE2
          const int off2 = (dx < 3) ? rowoff + dx + 2 : rowoff_n + (dx - 3);       // the tap after next (clamped on the last row)
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b2[vs] = lds[vbase[vs] + off2];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) {
            if constexpr (SWAP) Elem<T>::mma(b0[vs], a_cur[dx], acc[0][vs]);
            else Elem<T>::mma(a_cur[dx], b0[vs], acc[0][vs]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) { b0[vs] = b1[vs]; b1[vs] = b2[vs]; }
#else
          const int offn = (dx < 4) ? rowoff + dx + 1 : rowoff_n;
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_nxt[vs] = lds[vbase[vs] + offn];
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) {
            if constexpr (SWAP) Elem<T>::mma(b_cur[vs], a_cur[dx], acc[0][vs]);
            else Elem<T>::mma(a_cur[dx], b_cur[vs], acc[0][vs]);
          }
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_cur[vs] = b_nxt[vs];
#endif
        }
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) { a_cur[dx] = a_n1[dx]; a_n1[dx] = a_n2[dx]; }
        dz = dzn;
        dy = dyn;
      }
#else
    } else if constexpr (CW == 1) {
      u32x4 a_cur[5], a_nxt[5];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) a_cur[dx] = a_first[dx];
      // voxel fragments are double-buffered one tap ahead so that the LDS latency of tap t+1 hides
      // under the MFMAs of tap t
      u32x4 b_cur[VW], b_nxt[VW];
      {
        const int rowoff0 = (dz * BYH + dy) * BXH;
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) b_cur[vs] = lds[vbase[vs] + rowoff0];
      }
      for (int row = 0; row < nrows; ++row) {
        int dzn = dz, dyn = dy + 1;
        if (dyn > dy_hi) { dyn = dy_lo; dzn = dz + 1; }
        const bool more = row + 1 < nrows;
        if (more) {
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) a_nxt[dx] = wfrag(0, (dzn * 5 + dyn) * 5 + dx);
        }
        const int rowoff = (dz * BYH + dy) * BXH;
        const int rowoff_n = more ? (dzn * BYH + dyn) * BXH : rowoff;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const int offn = (dx < 4) ? rowoff + dx + 1 : rowoff_n;
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_nxt[vs] = lds[vbase[vs] + offn];
          RM_SCHED_FENCE();   // (RM_CONV_SCHED builds: keep the next tap's LDS reads ahead of this tap's MFMAs)
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) {
            if constexpr (SWAP) Elem<T>::mma(b_cur[vs], a_cur[dx], acc[0][vs]);
            else Elem<T>::mma(a_cur[dx], b_cur[vs], acc[0][vs]);
          }
          RM_SCHED_FENCE();
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_cur[vs] = b_nxt[vs];
        }
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) a_cur[dx] = a_nxt[dx];
        dz = dzn;
        dy = dyn;
      }
#endif
    } else {
      u32x4 a_cur[CW], a_nxt[CW];
#pragma unroll
      for (int cs = 0; cs < CW; ++cs) a_cur[cs] = wfrag(cs, (dz * 5 + dy) * 5);
      u32x4 b_cur[VW], b_nxt[VW];
      {
        const int rowoff0 = (dz * BYH + dy) * BXH;
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) b_cur[vs] = lds[vbase[vs] + rowoff0];
      }
      for (int row = 0; row < nrows; ++row) {
        int dzn = dz, dyn = dy + 1;
        if (dyn > dy_hi) { dyn = dy_lo; dzn = dz + 1; }
        const bool more = row + 1 < nrows;
        const int rowoff = (dz * BYH + dy) * BXH;
        const int rowoff_n = more ? (dzn * BYH + dyn) * BXH : rowoff;
        const int tap0 = (dz * 5 + dy) * 5;
        // the tap after dx = 4 is the first tap of the next row (clamped on the last row: harmless reload)
        const int tap_next_row = more ? (dzn * 5 + dyn) * 5 : tap0;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const int tapn = (dx < 4) ? tap0 + dx + 1 : tap_next_row;
          const int offn = (dx < 4) ? rowoff + dx + 1 : rowoff_n;
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) a_nxt[cs] = wfrag(cs, tapn);
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_nxt[vs] = lds[vbase[vs] + offn];
#pragma unroll
          for (int cs = 0; cs < CW; ++cs)
#pragma unroll
            for (int vs = 0; vs < VW; ++vs) {
              if constexpr (SWAP) Elem<T>::mma(b_cur[vs], a_cur[cs], acc[cs][vs]);
              else Elem<T>::mma(a_cur[cs], b_cur[vs], acc[cs][vs]);
            }
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) a_cur[cs] = a_nxt[cs];
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) b_cur[vs] = b_nxt[vs];
        }
        dz = dzn;
        dy = dyn;
      }
    }
    RM_PRIO(0);
    RM_STAMP((chunk - c_begin) * 4 + 3);
  }

  // ---- epilogue.  32x32 C/D layout: column j = lane & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  if constexpr (SWAP) {
    // rows = voxels, column = this lane's output channel; float output (store, or atomicAdd for split-K)
#pragma unroll
    for (int cs = 0; cs < CW; ++cs) {
      const int co = cot * C::COT + (wc * CW + cs) * 32 + l31;
      if (co >= Cout) continue;
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wv * VW + vs) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          const int lx = m % BX, ly = (m / BX) % BY, lz = m / (BX * BY);
          const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
          if (gz >= D || gy >= H || gx >= W) continue;
          const bool out2 = Cout1 > 0 && co >= Cout1;                   // (two-output mode: the data gradient of a pair)
          const int cw = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
          float* yp = static_cast<float*>(out2 ? a.y2 : a.y) + (((size_t)(n_out * D + gz) * H + gy) * W + gx) * cw +
                      (out2 ? co - Cout1 : co);
          if (ksplit > 1 || a.accum || (a.dual && !(a.dual & 4))) {
#ifdef RM_CONV_NOEPI
            if (acc[cs][vs][r] == 12345.678f)      // TIMING BUILD ONLY
#endif
            unsafeAtomicAdd(yp, acc[cs][vs][r]);
          } else {
            float v = acc[cs][vs][r];
            if (a.bias) v += a.bias[co];
            if (a.relu) v = fmaxf(v, 0.f);
            *yp = v;
          }
        }
      }
    }
  } else {
    // rows = output channels (4 consecutive per register quad), column = this lane's voxel; T output
    const bool want_stats = a.stats != nullptr;
    bool stored = false;
    if constexpr (sizeof(T) == 2) {
      if (a.wide && !want_stats) {
        // 16-byte stores (cdna_hip_programming.md T21): lane i holds channels 8q .. 8q+3 of its voxel, lane i + 32 channels
        // 8q+4 .. 8q+7.  One v_permlane32_swap per packed dword of the quad pair (q, q+1) leaves 8 consecutive channels of
        // quad q in the lower and of quad q + 1 in the upper half-wave: half as many store instructions for the same bytes
        // (the host sets `wide` only when every 16-channel group lies inside one output tensor).
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) {
          const int m = (wv * VW + vs) * 32 + l31;
          const int lx = m % BX, ly = (m / BX) % BY, lz = m / (BX * BY);
          const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
          const bool inside = gz < D && gy < H && gx < W;
          const size_t vox = ((size_t)(n_out * D + gz) * H + gy) * W + gx;
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) {
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
              const int co16 = cot * C::COT + (wc * CW + cs) * 32 + 16 * qp;     // first channel of this 16-channel group
              if (co16 >= Cout) continue;                                          // (wave-uniform)
              uint32_t pk[2][2];
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int q = 2 * qp + g;
                float v0 = acc[cs][vs][4 * q + 0], v1 = acc[cs][vs][4 * q + 1];
                float v2 = acc[cs][vs][4 * q + 2], v3 = acc[cs][vs][4 * q + 3];
                if (a.bias) {
                  const float* bp = a.bias + co16 + 8 * g + 4 * khalf;
                  v0 += bp[0]; v1 += bp[1]; v2 += bp[2]; v3 += bp[3];
                }
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                pk[g][0] = pack_bf16x2(v0, v1);
                pk[g][1] = pack_bf16x2(v2, v3);
              }
              // vdst = quad q's dword, src = quad q + 1's: the upper half of vdst swaps with the lower half of src
              const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
              if (!inside) continue;
              const bool out2 = Cout1 > 0 && co16 >= Cout1;
              const int Cout_ = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
              const int co = (out2 ? co16 - Cout1 : co16) + 8 * khalf;
              bf16_t* yp = static_cast<bf16_t*>(out2 ? a.y2 : a.y) + vox * Cout_ + co;
              *reinterpret_cast<u32x4*>(yp) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
          }
        }
        stored = true;
      }
    }
    if (!stored) {
    float ssum[CW][16], ssq[CW][16];
    if (want_stats) {
#pragma unroll
      for (int cs = 0; cs < CW; ++cs)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ssum[cs][r] = 0.f; ssq[cs][r] = 0.f; }
      // put the zeros back into the BatchNorm scratch half the previous call used (what bn_stats_kernel does)
      for (int i = conv_block * NT + tid; i < (int)REPMODE_SCRATCH_BN_HALF; i += conv_blocks * NT) a.stats_clear[i] = 0.f;
    }
#pragma unroll
    for (int vs = 0; vs < VW; ++vs) {
      const int m = (wv * VW + vs) * 32 + l31;
      const int lx = m % BX, ly = (m / BX) % BY, lz = m / (BX * BY);
      const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
      const bool inside = gz < D && gy < H && gx < W;
      if (!inside && !want_stats) continue;
      const size_t vox = ((size_t)(n_out * D + gz) * H + gy) * W + gx;
#pragma unroll
      for (int cs = 0; cs < CW; ++cs) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co_all = cot * C::COT + (wc * CW + cs) * 32 + 8 * q + 4 * khalf;
          if (co_all >= Cout) continue;
          const bool out2 = Cout1 > 0 && co_all >= Cout1;              // (Cout1 is a multiple of 4: a quad never straddles)
          const int Cout_ = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
          const int co = out2 ? co_all - Cout1 : co_all;
          void* ybase = out2 ? a.y2 : a.y;
          float v0 = acc[cs][vs][4 * q + 0], v1 = acc[cs][vs][4 * q + 1];
          float v2 = acc[cs][vs][4 * q + 2], v3 = acc[cs][vs][4 * q + 3];
          if (a.bias) {
            v0 += a.bias[co_all];
            if (co_all + 1 < Cout) v1 += a.bias[co_all + 1];
            if (co_all + 2 < Cout) v2 += a.bias[co_all + 2];
            if (co_all + 3 < Cout) v3 += a.bias[co_all + 3];
          }
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          if constexpr (sizeof(T) == 4) {
            if (want_stats && inside) {
              ssum[cs][4 * q + 0] += v0; ssq[cs][4 * q + 0] += v0 * v0;
              ssum[cs][4 * q + 1] += v1; ssq[cs][4 * q + 1] += v1 * v1;
              ssum[cs][4 * q + 2] += v2; ssq[cs][4 * q + 2] += v2 * v2;
              ssum[cs][4 * q + 3] += v3; ssq[cs][4 * q + 3] += v3 * v3;
            }
            if (!inside) continue;
            float* yp = static_cast<float*>(ybase) + vox * Cout_ + co;
            if ((Cout_ & 3) == 0) {
              *reinterpret_cast<f32x4*>(yp) = f32x4{v0, v1, v2, v3};
            } else {
              yp[0] = v0;
              if (co + 1 < Cout_) yp[1] = v1;
              if (co + 2 < Cout_) yp[2] = v2;
              if (co + 3 < Cout_) yp[3] = v3;
            }
          } else {
            const uint32_t p01 = pack_bf16x2(v0, v1), p23 = pack_bf16x2(v2, v3);
            if (want_stats && inside) {
              // the statistics of what is stored: the bf16-rounded values the normalisation will read back
              const float r0 = __uint_as_float(p01 << 16), r1 = __uint_as_float(p01 & 0xffff0000u);
              const float r2 = __uint_as_float(p23 << 16), r3 = __uint_as_float(p23 & 0xffff0000u);
              ssum[cs][4 * q + 0] += r0; ssq[cs][4 * q + 0] += r0 * r0;
              ssum[cs][4 * q + 1] += r1; ssq[cs][4 * q + 1] += r1 * r1;
              ssum[cs][4 * q + 2] += r2; ssq[cs][4 * q + 2] += r2 * r2;
              ssum[cs][4 * q + 3] += r3; ssq[cs][4 * q + 3] += r3 * r3;
            }
            if (!inside) continue;
            bf16_t* yp = static_cast<bf16_t*>(ybase) + vox * Cout_ + co;
            if ((Cout_ & 3) == 0) {
              *reinterpret_cast<u32x2*>(yp) = u32x2{p01, p23};
            } else {
              yp[0] = (bf16_t)(p01 & 0xffffu);
              if (co + 1 < Cout_) yp[1] = (bf16_t)(p01 >> 16);
              if (co + 2 < Cout_) yp[2] = (bf16_t)(p23 & 0xffffu);
              if (co + 3 < Cout_) yp[3] = (bf16_t)(p23 >> 16);
            }
          }
        }
      }
    }
    if (want_stats) {
      // lanes with equal khalf hold the same 16 channels of different voxels: butterfly over the 32 lanes, then one lane
      // per half-wave adds the wave's totals to this workgroup's slice (BatchNorm's slice layout, bnrelu.hip)
      float* slice = a.stats + (size_t)(conv_block & 15) * 2 * Cout;
#pragma unroll
      for (int cs = 0; cs < CW; ++cs) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float sv = ssum[cs][r], qv = ssq[cs][r];
#pragma unroll
          for (int off = 1; off < 32; off <<= 1) { sv += __shfl_xor(sv, off, 64); qv += __shfl_xor(qv, off, 64); }
          const int co_all = cot * C::COT + (wc * CW + cs) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          if (l31 == 0 && co_all < Cout) {
            unsafeAtomicAdd(slice + co_all, sv);
            unsafeAtomicAdd(slice + Cout + co_all, qv);
          }
        }
      }
    }
    }   // (!stored)
  }
  RM_STAMP(60);
}

// =====================================================================================================================
// conv5_pipe_kernel -- the wide levels' (x extent >= 32) element-typed bf16 convolution as ONE software pipeline per CU.
//
// The kernel above gives a CU two workgroups that each alternate "stage a 16-channel halo image" / "125 taps" / epilogue and
// relies on their drifting out of phase; the PMC picture of its level-0 launches (profiles/r03_pmc_conv_level0.txt) is 53 %
// MFMA-busy with a third of the wave cycles parked in s_waitcnt / barriers.  Here a CU runs ONE workgroup of four waves (one
// per SIMD, the whole register file each) that never leaves the tap loop:
//   * the halo image is double-buffered in LDS (2 x 73.7 KB); while the 125 taps of image `cur` run, the 18 sixteen-byte
//     items per thread of the NEXT image (next channel chunk, or the first chunk of the workgroup's next brick) are requested
//     two per tap row during rows 0..8 (buffer loads: positions outside the volume read as zero through the range check) and
//     stored into the other buffer two per row during rows 14..22 -- one barrier per image, nothing waits for memory;
//   * the 25 (dz, dy) rows are unrolled, so every LDS read is an immediate offset from one address register, every filter
//     fragment one buffer load with a scalar offset; filter fragments run 1280 MFMA cycles ahead, across image boundaries;
//   * a workgroup owns a contiguous range of (brick, channel tile) items (persistent grid: at most one workgroup per CU), so
//     the epilogue's stores drain under the next brick's taps;
//   * CW = 2 (layers with more than 32 output channels): a voxel fragment feeds two MFMAs -- 128 accumulator registers, which
//     the two-workgroup form could not hold.
// Out-of-volume taps are not skipped (the image holds zeros there); the launcher sends only volumes of several bricks here.
template <int CW>
__global__ __launch_bounds__(256, 1) void conv5_pipe_kernel(ConvArgs a, int nitems) {
  using C = Cfg<4, 4, 32, 4, 1, 4, CW>;
  constexpr int KV = 8, KC = 16, VW = 4;
  constexpr int BZ = C::BZ, BY = C::BY, BX = C::BX, BYH = C::BYH, BXH = C::BXH, VH = C::VH, PLS = C::PLS;
  constexpr int BUF = 2 * PLS;                  // 16-byte slots of one halo image
  constexpr int NIT = (2 * VH) / 256;           // halo items per thread
  static_assert((2 * VH) % 256 == 0 && NIT == 18, "halo items: two per tap row over nine rows");
  constexpr uint32_t OOB = 0x7fffffffu;         // beyond every descriptor's range: the load returns zeros

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int khalf = lane >> 5, l31 = lane & 31;
  if (a.tail.nblocks) {
    if ((int)blockIdx.x < a.tail.nblocks) {
      tail_run(a.tail, blockIdx.x, tid, reinterpret_cast<float*>(smem));
      return;
    }
  }
  const int conv_block = blockIdx.x - a.tail.nblocks, conv_blocks = gridDim.x - a.tail.nblocks;
  const int L = xcd_remap(conv_block, conv_blocks);
  const int per = nitems / conv_blocks, rem = nitems % conv_blocks;
  int item = L * per + min(L, rem);
  const int item_end = item + per + (L < rem ? 1 : 0);
  if (item >= item_end) return;

  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, CinP = a.CinP, CoutP = a.CoutP;
  const int Cin1 = a.Cin1, Cout1 = a.Cout1;
  const int nkc = CinP / KC, nrt = CoutP / 32;
  const uint32_t ts_bytes = (uint32_t)CoutP * (uint32_t)CinP * 2u;               // one tap of one slot
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 0x7ffffffe, 0x00020000);
  const int c1 = Cin1 > 0 ? Cin1 : Cin;                                           // channels of the first input tensor
  const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.x), 0, (int)((size_t)a.N * D * H * W * c1 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(Cin1 > 0 ? a.x2 : a.x), 0, (int)((size_t)a.N * D * H * W * (Cin1 > 0 ? Cin - Cin1 : c1) * 2), 0x00020000);

  // ---- per-thread constants of the halo items: item u is halo voxel vh = 128 u + tid / 2, plane tid & 1
  const int pl = tid & 1, vh0 = tid >> 1;
  int hvox[NIT], hpc[NIT];
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    const int vh = u * 128 + vh0;
    const int xx = vh % BXH, t2 = vh / BXH;
    const int yy = t2 % BYH, zz = t2 / BYH;
    hvox[u] = (zz * H + yy) * W + xx;
    hpc[u] = zz | (yy << 8) | (xx << 16);
  }
  const int lds_item = pl * PLS + vh0;                            // + 128 u
  const int lane_w = (l31 * KC + khalf * KV) * 2;                 // this lane's 16 bytes of a filter fragment
  const u32x4* lb0 = lds + khalf * PLS + (wave * BYH) * BXH + l31;   // this lane's voxel of sub-tile vs: + vs * BXH

  // ---- the pipeline's unit: image (item, chunk)
  struct Img { int n, z0, y0, x0, cot, chunk; uint32_t wbase[CW]; };
  auto decode = [&](int it, int chunk) -> Img {
    Img g;
    int b = it;
    g.cot = b % a.ncot; b /= a.ncot;
    const int bx = b % a.nbx; b /= a.nbx;
    const int by = b % a.nby; b /= a.nby;
    const int bz = b % a.nbz;
    g.n = b / a.nbz;
    g.z0 = bz * BZ; g.y0 = by * BY; g.x0 = bx * BX;
    g.chunk = chunk;
    const int slot = a.sample_slot[g.n];
#pragma unroll
    for (int cs = 0; cs < CW; ++cs) {
      const int rt = min(g.cot * CW + cs, nrt - 1);         // (a tile beyond the padded filter is clamped; never stored)
      g.wbase[cs] = (uint32_t)slot * REPMODE_TAPS * ts_bytes + (uint32_t)((rt * nkc + chunk) * (32 * KC) * 2);
    }
    return g;
  };
  u32x4 hv[NIT];
  // the scalars of an image's halo requests, computed once per image (a descriptor select inside the tap loop is a branch)
  struct Halo { __amdgpu_buffer_rsrc_t rs; int csrc, coff, base_vox, z0, y0, x0; bool on; };
  auto halo_of = [&](const Img& g, bool on) -> Halo {
    Halo q;
    const int ci0 = g.chunk * KC;
    const bool from2 = Cin1 > 0 && ci0 >= Cin1;
    q.rs = from2 ? rx2 : rx1;
    q.csrc = Cin1 > 0 ? (from2 ? Cin - Cin1 : Cin1) : Cin;
    const int c = ci0 + pl * KV;                                   // (per-thread: the plane's first channel)
    q.coff = c - (from2 ? Cin1 : 0);
#ifdef RM_PIPE_NOHALO
    on = false;        // (timing experiment, -DRM_PIPE_NOHALO: no halo fetches after the first image -- WRONG RESULTS)
#endif
    q.on = on && c < Cin;
#ifdef RM_PIPE_SAMEHALO
    // (timing experiment: every image is fetched from ONE interior brick of sample 0 -- real data, always L2-resident: what
    // the kernel would run at if no halo fetch ever missed.  WRONG RESULTS)
    q.base_vox = ((0 * D + 4 - 2) * H + 4 - 2) * W + 32 - 2;
    q.z0 = 2; q.y0 = 2; q.x0 = 30;
#else
    q.base_vox = ((g.n * D + g.z0 - 2) * H + g.y0 - 2) * W + g.x0 - 2;
    q.z0 = g.z0 - 2; q.y0 = g.y0 - 2; q.x0 = g.x0 - 2;
#endif
    return q;
  };
  auto halo_load = [&](const Halo& q, int u) {
    const int zz = hpc[u] & 0xff, yy = (hpc[u] >> 8) & 0xff, xx = hpc[u] >> 16;
    const bool ok = q.on && (unsigned)(q.z0 + zz) < (unsigned)D && (unsigned)(q.y0 + yy) < (unsigned)H && (unsigned)(q.x0 + xx) < (unsigned)W;
    const uint32_t off = (uint32_t)((q.base_vox + hvox[u]) * q.csrc + q.coff) * 2u;
    hv[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(q.rs, ok ? off : OOB, 0, 0));
  };
  auto wfrag = [&](const Img& g, int cs, int tap, int voff) -> u32x4 {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff, g.wbase[cs] + (uint32_t)tap * ts_bytes, 0));
  };

  f32x16 acc[CW][VW];
#pragma unroll
  for (int cs = 0; cs < CW; ++cs)
#pragma unroll
    for (int vs = 0; vs < VW; ++vs)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cs][vs][r] = 0.f;

  // ---- prologue: the first image into buffer 0, the first two filter rows into registers
  Img cur_g = decode(item, 0);
  // filter rows in flight ahead of the one being multiplied: two (10 taps = 1280 MFMA cycles) with one channel sub-tile,
  // one (5 taps of two sub-tiles: the same 1280 cycles) with two -- whose 128 accumulators leave no room for a third set
  constexpr int RA = CW == 1 ? 2 : 1;
  u32x4 a_cur[CW][5], a_n1[CW][5], a_n2[CW][5];
  {
    const Halo q0 = halo_of(cur_g, true);
#pragma unroll
    for (int u = 0; u < NIT; ++u) halo_load(q0, u);
  }
#pragma unroll
  for (int cs = 0; cs < CW; ++cs)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      a_cur[cs][dx] = wfrag(cur_g, cs, dx, lane_w);
      if constexpr (RA == 2) a_n1[cs][dx] = wfrag(cur_g, cs, 5 + dx, lane_w);
    }
#pragma unroll
  for (int u = 0; u < NIT; ++u) lds[lds_item + 128 * u] = hv[u];
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int cur = 0;
#ifdef RM_CONV_TIMING
  int st_ = 0;          // image counter of the shader-clock stamps (tools/conv_phase_timing.py): 4 per image, 16 images
#endif

  for (;;) {
    RM_STAMP(st_ * 4 + 0);
    // the image after this one: next channel chunk of the brick, or the first chunk of the next item
    const bool last_chunk = cur_g.chunk + 1 >= nkc;
    const bool have_next = !last_chunk || item + 1 < item_end;
    Img nxt_g = cur_g;
    if (have_next) nxt_g = last_chunk ? decode(item + 1, 0) : decode(item, cur_g.chunk + 1);
#ifdef RM_PIPE_NOFILT
    const int lane_w_nxt = (int)OOB;     // (timing experiment, -DRM_PIPE_NOFILT: filter fragments read as zeros -- WRONG RESULTS)
#else
    const int lane_w_nxt = have_next ? lane_w : (int)OOB;
#endif
    const Halo hq = halo_of(nxt_g, have_next);
    const u32x4* lb = lb0 + cur * BUF;
    u32x4* lw = lds + (cur ^ 1) * BUF + lds_item;
    // Voxel fragments run TWO taps ahead in three rotating register sets (tap t uses set t % 3; everything below is
    // unrolled, so the rotation is register naming, not moves), and each LDS read is issued in the shadow of one MFMA:
    // M d M d M d M d -- a lone wave has ~5 issue slots per 32-cycle MFMA, and a read requested only one tap (128 cycles)
    // ahead was not back in time.
    auto tap_off = [&](int t) -> int { return (((t / 5) / 5) * BYH + ((t / 5) % 5)) * BXH + t % 5; };
    u32x4 bq[3][VW];
#pragma unroll
    for (int vs = 0; vs < VW; ++vs) { bq[0][vs] = lb[vs * BXH + tap_off(0)]; bq[1][vs] = lb[vs * BXH + tap_off(1)]; }
#pragma unroll
    for (int r = 0; r < 25; ++r) {
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int t = r * 5 + dx;
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) {
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) Elem<bf16_t>::mma(a_cur[cs][dx], bq[t % 3][vs], acc[cs][vs]);
          if (t + 2 < 125) bq[(t + 2) % 3][vs] = lb[vs * BXH + tap_off(t + 2)];
          __builtin_amdgcn_sched_barrier(0);
        }
        // this tap's filter fragment of the row RA ahead (the last RA rows: the next image's first rows)
        // (No branch in here: `have_next` false turns the next image's requests into out-of-range offsets -- zeros, no memory
        // traffic -- and its stores into writes of a buffer nobody reads.  A uniform branch per row split the unrolled loop into
        // blocks with full s_waitcnt at their seams: 660 instead of 217 us on level 0.)
        if (r + RA < 25) {
#pragma unroll
#ifdef RM_PIPE_NOFILT
          for (int cs = 0; cs < CW; ++cs) a_n2[cs][dx] = wfrag(cur_g, cs, (r + RA) * 5 + dx, (int)OOB);
#else
          for (int cs = 0; cs < CW; ++cs) a_n2[cs][dx] = wfrag(cur_g, cs, (r + RA) * 5 + dx, lane_w);
#endif
        } else {
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) a_n2[cs][dx] = wfrag(nxt_g, cs, (r + RA - 25) * 5 + dx, lane_w_nxt);
        }
        // the next image: two items requested per row in rows 0..8, stored per row in rows 14..22
        if (dx == 1 || dx == 3) {
          const int k = dx >> 1;
          if (r < 9) halo_load(hq, 2 * r + k);
          if (r >= 14 && r < 23) lw[128 * (2 * (r - 14) + k)] = hv[2 * (r - 14) + k];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (r == 12) RM_STAMP(st_ * 4 + 1);
#pragma unroll
      for (int cs = 0; cs < CW; ++cs)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          if constexpr (RA == 2) { a_cur[cs][dx] = a_n1[cs][dx]; a_n1[cs][dx] = a_n2[cs][dx]; }
          else a_cur[cs][dx] = a_n2[cs][dx];
        }
    }
    RM_STAMP(st_ * 4 + 2);
    // every wave is done with image `cur` and has stored its part of the next one
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    RM_STAMP(st_ * 4 + 3);

    if (last_chunk) {
      // ---- epilogue of the brick (the stores drain under the next brick's taps).  32x32 C/D layout: column = lane & 31
      // (voxel), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (output channels)
      const int z0 = cur_g.z0, y0 = cur_g.y0, x0 = cur_g.x0, n_out = cur_g.n, cot = cur_g.cot;
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) {
        const int gz = z0 + wave, gy = y0 + vs, gx = x0 + l31;       // voxel m = (wave 4 + vs) 32 + l31 of the brick
        const bool inside = gz < D && gy < H && gx < W;
        const size_t vox = ((size_t)(n_out * D + gz) * H + gy) * W + gx;
#pragma unroll
        for (int cs = 0; cs < CW; ++cs) {
          if (a.wide) {
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
              const int co16 = cot * C::COT + cs * 32 + 16 * qp;
              if (co16 >= Cout) continue;
              uint32_t pk[2][2];
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int q = 2 * qp + g;
                float v0 = acc[cs][vs][4 * q + 0], v1 = acc[cs][vs][4 * q + 1];
                float v2 = acc[cs][vs][4 * q + 2], v3 = acc[cs][vs][4 * q + 3];
                if (a.bias) {
                  const float* bp = a.bias + co16 + 8 * g + 4 * khalf;
                  v0 += bp[0]; v1 += bp[1]; v2 += bp[2]; v3 += bp[3];
                }
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                pk[g][0] = pack_bf16x2(v0, v1);
                pk[g][1] = pack_bf16x2(v2, v3);
              }
              const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
              if (!inside) continue;
              const bool out2 = Cout1 > 0 && co16 >= Cout1;
              const int Cout_ = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
              const int co = (out2 ? co16 - Cout1 : co16) + 8 * khalf;
              bf16_t* yp = static_cast<bf16_t*>(out2 ? a.y2 : a.y) + vox * Cout_ + co;
              *reinterpret_cast<u32x4*>(yp) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int co_all = cot * C::COT + cs * 32 + 8 * q + 4 * khalf;
              if (co_all >= Cout || !inside) continue;
              const bool out2 = Cout1 > 0 && co_all >= Cout1;
              const int Cout_ = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
              const int co = out2 ? co_all - Cout1 : co_all;
              float v0 = acc[cs][vs][4 * q + 0], v1 = acc[cs][vs][4 * q + 1];
              float v2 = acc[cs][vs][4 * q + 2], v3 = acc[cs][vs][4 * q + 3];
              if (a.bias) {
                v0 += a.bias[co_all];
                if (co_all + 1 < Cout) v1 += a.bias[co_all + 1];
                if (co_all + 2 < Cout) v2 += a.bias[co_all + 2];
                if (co_all + 3 < Cout) v3 += a.bias[co_all + 3];
              }
              if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
              const uint32_t p01 = pack_bf16x2(v0, v1), p23 = pack_bf16x2(v2, v3);
              bf16_t* yp = static_cast<bf16_t*>(out2 ? a.y2 : a.y) + vox * Cout_ + co;
              if ((Cout_ & 3) == 0) {
                *reinterpret_cast<u32x2*>(yp) = u32x2{p01, p23};
              } else {
                yp[0] = (bf16_t)(p01 & 0xffffu);
                if (co + 1 < Cout_) yp[1] = (bf16_t)(p01 >> 16);
                if (co + 2 < Cout_) yp[2] = (bf16_t)(p23 & 0xffffu);
                if (co + 3 < Cout_) yp[3] = (bf16_t)(p23 >> 16);
              }
            }
          }
        }
      }
#pragma unroll
      for (int cs = 0; cs < CW; ++cs)
#pragma unroll
        for (int vs = 0; vs < VW; ++vs)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[cs][vs][r] = 0.f;
      ++item;
    }
    if (!have_next) break;
    cur_g = nxt_g;
    cur ^= 1;
#ifdef RM_CONV_TIMING
    ++st_;
#endif
  }
}

// =====================================================================================================================
// conv5_ws_kernel -- conv5_pipe_kernel with the waves SPECIALISED: eight waves per workgroup, one MFMA wave and one loader
// wave per SIMD.  The stamps of the pipelined kernel (tools/conv_phase_timing.py, timing build) show its tap rows 13..24 at
// 94 % of the MFMA rate but rows 0..12 -- where each wave also requests the next halo image -- at 71 %, and level 1 losing
// 23 % to halo fetches that miss L2 (a timing build that fetches every image from one resident brick: 96 -> 74 us): on gfx950
// a wave's vector-memory operations retire IN ORDER, so every filter fragment requested behind a slow halo fetch waits for
// it.  Here the MFMA waves' memory queue holds nothing but filter fragments (L2 hits) and the brick's output stores; the
// loader waves (wave 4 + s on SIMD s, ~150 instructions per image) fetch the next image into the other LDS buffer and meet
// the MFMA waves at the one barrier per image.  Two waves per SIMD: 256 registers each, which the MFMA waves can afford once
// the halo staging registers (108) are gone -- filter fragments run 4 rows ahead (one channel sub-tile) / 2 rows (two).
// PLAIN: the training path's epilogue (16-byte stores, no bias, no ReLU) as straight-line code -- the general one is a
// chain of uniform branches per (sub-tile, quad pair) that cost 4.3 k cycles per brick (stamps), 11 % of a level-0 brick.
// RS (CW = 1 only): row-stationary tap order -- for a fixed (dz, dx) the voxel fragment of halo row y' = vs + dy serves every
// (sub-tile vs, dy) pair that lands on it: 8 LDS reads + 5 filter fragments feed 20 MFMAs (0.4 reads per MFMA instead of 1).
// The summation order of the 125 taps differs from the tap-major kernels (float rounding: not bit-identical to them).
// BXT = 16 (round 4): the same pipeline on a 4 x 4 x 16 brick for volumes 16 voxels wide (level 2 of the network at the
// 32 x 64 x 64 patch) -- a wave's sub-tile is two x rows of 16 voxels (lane l: row l >> 4, x = l & 15), two sub-tiles per wave,
// 41 KB per halo image.  At batch 8 level 2 has 2048 output tiles of 32 x 32 -- two per MFMA wave of the chip -- so an item is
// (brick, 32 channels) with the whole channel reduction inside: 256 items, no split reduction, bf16 output (round 3 ran these
// layers through the two-workgroup kernel with the reduction split four ways over float atomics: 2.56 x the algorithmic traffic).
template <int CW, bool PLAIN, bool RS = false, int BXT = 32>
__global__ __launch_bounds__(512, 2) void conv5_ws_kernel(ConvArgs a, int nitems) {
  static_assert(!RS || (CW == 1 && BXT == 32), "row-stationary order: one channel sub-tile per wave, 32-voxel rows");
  static_assert(BXT == 32 || BXT == 16, "brick width");
  constexpr int VW = BXT == 32 ? 4 : 2;            // voxel sub-tiles per MFMA wave
  constexpr int SUBROWS = 32 / BXT;                // x rows per sub-tile
  using C = Cfg<4, 4, BXT, 4, 1, VW, CW>;
  constexpr int KV = 8, KC = 16;
  constexpr int BZ = C::BZ, BY = C::BY, BX = C::BX, BYH = C::BYH, BXH = C::BXH, VH = C::VH, PLS = C::PLS;
  constexpr int BUF = 2 * PLS;                  // 16-byte slots of one halo image
  constexpr int NIT = (2 * VH) / 256;           // halo items per loader thread
  static_assert((2 * VH) % 256 == 0, "halo items: whole rounds of the 256 loader threads");
  constexpr uint32_t OOB = 0x7fffffffu;         // beyond every descriptor's range: the load returns zeros

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int khalf = lane >> 5, l31 = lane & 31;
  if (a.tail.nblocks) {
    if ((int)blockIdx.x < a.tail.nblocks) {
      if (tid < 256) tail_run(a.tail, blockIdx.x, tid, reinterpret_cast<float*>(smem));   // (the jobs are written for 256 threads)
      return;
    }
  }
  const int conv_block = blockIdx.x - a.tail.nblocks, conv_blocks = gridDim.x - a.tail.nblocks;
  const int L = xcd_remap(conv_block, conv_blocks);
  const int per = nitems / conv_blocks, rem = nitems % conv_blocks;
  int item = L * per + min(L, rem);
  const int item_end = item + per + (L < rem ? 1 : 0);
  if (item >= item_end) return;

  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, CinP = a.CinP, CoutP = a.CoutP;
  const int Cin1 = a.Cin1, Cout1 = a.Cout1;
  const int nkc = CinP / KC, nrt = CoutP / 32;

  // brick of an item (both roles walk the same sequence of images: items item .. item_end-1, chunks 0 .. nkc-1 of each)
  struct Brick { int n, z0, y0, x0, cot; };
  auto brick_of = [&](int it) -> Brick {
    Brick g;
    int b = it;
    g.cot = b % a.ncot; b /= a.ncot;
    if (a.zfast) {
      // z first: consecutive items of a workgroup share half of their halo images (a brick is 4 planes thick, its image 8),
      // and all workgroups, in step, sit in one z slab of the batch -- the y neighbours' shared rows are in L2
      const int bz = b % a.nbz; b /= a.nbz;
      const int bx = b % a.nbx; b /= a.nbx;
      const int by = b % a.nby;
      g.n = b / a.nby;
      g.z0 = bz * BZ; g.y0 = by * BY; g.x0 = bx * BX;
      return g;
    }
    const int bx = b % a.nbx; b /= a.nbx;
    const int by = b % a.nby; b /= a.nby;
    const int bz = b % a.nbz;
    g.n = b / a.nbz;
    g.z0 = bz * BZ; g.y0 = by * BY; g.x0 = bx * BX;
    return g;
  };
  // the item after g (item + 1) without the five divisions: carry through (cot, bx, by, bz, n) / (cot, bz, bx, by, n)
  auto brick_next = [&](Brick g) -> Brick {
    if (++g.cot < a.ncot) return g;
    g.cot = 0;
    if (a.zfast) {
      if ((g.z0 += BZ) < a.nbz * BZ) return g;
      g.z0 = 0;
      if ((g.x0 += BX) < a.nbx * BX) return g;
      g.x0 = 0;
      if ((g.y0 += BY) < a.nby * BY) return g;
      g.y0 = 0;
      ++g.n;
      return g;
    }
    if ((g.x0 += BX) < a.nbx * BX) return g;
    g.x0 = 0;
    if ((g.y0 += BY) < a.nby * BY) return g;
    g.y0 = 0;
    if ((g.z0 += BZ) < a.nbz * BZ) return g;
    g.z0 = 0;
    ++g.n;
    return g;
  };

  if (wave >= 4) {
    // =============================== loader waves: the halo images =========================================
    const int ht = tid - 256;
    const int pl = ht & 1, vh0 = ht >> 1;
    const int c1 = Cin1 > 0 ? Cin1 : Cin;                                           // channels of the first input tensor
    const __amdgpu_buffer_rsrc_t rx1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(a.x), 0, (int)((size_t)a.N * D * H * W * c1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(Cin1 > 0 ? a.x2 : a.x), 0, (int)((size_t)a.N * D * H * W * (Cin1 > 0 ? Cin - Cin1 : c1) * 2), 0x00020000);
    int hvox[NIT], hpc[NIT];                    // item u: halo voxel vh = 128 u + ht / 2, plane ht & 1
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int vh = u * 128 + vh0;
      const int xx = vh % BXH, t2 = vh / BXH;
      const int yy = t2 % BYH, zz = t2 / BYH;
      hvox[u] = (zz * H + yy) * W + xx;
      hpc[u] = zz | (yy << 8) | (xx << 16);
    }
    u32x4* lw0 = lds + pl * PLS + vh0;          // + 128 u, + BUF for the second buffer
    auto fetch = [&](const Brick& g, int chunk, int buf) {
      const int ci0 = chunk * KC;
      const bool from2 = Cin1 > 0 && ci0 >= Cin1;
      const __amdgpu_buffer_rsrc_t rs = from2 ? rx2 : rx1;
      const int csrc = Cin1 > 0 ? (from2 ? Cin - Cin1 : Cin1) : Cin;
      const int c = ci0 + pl * KV;
      const int coff = c - (from2 ? Cin1 : 0);
      const bool on = c < Cin;
      const int base_vox = ((g.n * D + g.z0 - 2) * H + g.y0 - 2) * W + g.x0 - 2;
      u32x4 hv[NIT];
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const int zz = hpc[u] & 0xff, yy = (hpc[u] >> 8) & 0xff, xx = hpc[u] >> 16;
        const bool ok = on && (unsigned)(g.z0 - 2 + zz) < (unsigned)D && (unsigned)(g.y0 - 2 + yy) < (unsigned)H &&
                        (unsigned)(g.x0 - 2 + xx) < (unsigned)W;
        const uint32_t off = (uint32_t)((base_vox + hvox[u]) * csrc + coff) * 2u;
        hv[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : OOB, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < NIT; ++u) lw0[buf * BUF + 128 * u] = hv[u];
    };
    Brick g = brick_of(item);
    int chunk = 0, cur = 0;
    fetch(g, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (;;) {
      const bool last_chunk = chunk + 1 >= nkc;
      const bool have_next = !last_chunk || item + 1 < item_end;
      if (have_next) {
        if (last_chunk) { ++item; chunk = 0; g = brick_next(g); } else { ++chunk; }
        fetch(g, chunk, cur ^ 1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (!have_next) break;
      cur ^= 1;
    }
    return;
  }

  // ================================= MFMA waves: the taps =================================================
  const uint32_t ts_bytes = (uint32_t)CoutP * (uint32_t)CinP * 2u;               // one tap of one slot
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 0x7ffffffe, 0x00020000);
  const int lane_w = (l31 * KC + khalf * KV) * 2;                 // this lane's 16 bytes of a filter fragment
  // this lane's voxel of sub-tile vs: + vs * SUBROWS * BXH
  const u32x4* lb0 = lds + khalf * PLS + (wave * BYH + (BXT == 32 ? 0 : (l31 >> 4))) * BXH + (BXT == 32 ? l31 : (l31 & 15));
  struct Img { Brick b; int chunk; uint32_t wbase[CW]; };
  // the samples' slots: one vector load for the first 64 samples, then a lane read per image (a load per image sat at the
  // head of the wave's in-order memory queue, in front of the image's filter fragments)
  const int slot64 = a.sample_slot[min(lane, a.N - 1)];
  auto image_of = [&](const Brick& b, int chunk) -> Img {
    Img g;
    g.b = b; g.chunk = chunk;
    const int n_u = __builtin_amdgcn_readfirstlane(b.n);
    const int slot = n_u < 64 ? __builtin_amdgcn_readlane(slot64, n_u) : a.sample_slot[n_u];
#pragma unroll
    for (int cs = 0; cs < CW; ++cs) {
      const int rt = min(b.cot * CW + cs, nrt - 1);         // (a tile beyond the padded filter is clamped; never stored)
      g.wbase[cs] = (uint32_t)slot * REPMODE_TAPS * ts_bytes + (uint32_t)((rt * nkc + chunk) * (32 * KC) * 2);
    }
    return g;
  };
  auto wfrag = [&](const Img& g, int cs, int tap, int voff) -> u32x4 {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, voff, g.wbase[cs] + (uint32_t)tap * ts_bytes, 0));
  };

  f32x16 acc[CW][VW];
#pragma unroll
  for (int cs = 0; cs < CW; ++cs)
#pragma unroll
    for (int vs = 0; vs < VW; ++vs)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cs][vs][r] = 0.f;

  // filter rows in flight ahead of the one being multiplied, voxel fragments (taps) in flight ahead
  // (16-voxel bricks: a tap is VW * CW = 2 or 4 MFMAs, half of the wide brick's -- the same distance in cycles is twice the rows)
  constexpr int RA = BXT == 16 ? (CW == 1 ? RM_WS16_RA : 2)
                               : RS ? 3 : CW == 1 ? 4 : PLAIN ? 2 : 1;     // filter rows (RS: (dz, dx) groups) ahead; the eval epilogue needs the registers
  constexpr int TA = BXT == 16 ? (CW == 1 ? RM_WS16_TA : 2) : CW == 1 ? 2 : 1;
  Img cur_g = image_of(brick_of(item), 0);
  u32x4 aq[RA + 1][CW][5];
#pragma unroll
  for (int k = 0; k < RA; ++k)
#pragma unroll
    for (int cs = 0; cs < CW; ++cs)
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) aq[k][cs][dx] = wfrag(cur_g, cs, RS ? ((k / 5) * 5 + dx) * 5 + k % 5 : k * 5 + dx, lane_w);
  asm volatile("s_barrier" ::: "memory");          // the loaders have stored the first image
  int cur = 0;
#ifdef RM_CONV_TIMING
  int st_ = 0;
#endif

  for (;;) {
    RM_STAMP(st_ * 4 + 0);
    const bool last_chunk = cur_g.chunk + 1 >= nkc;
    const bool have_next = !last_chunk || item + 1 < item_end;
    Img nxt_g = cur_g;
    if (have_next) nxt_g = last_chunk ? image_of(brick_next(cur_g.b), 0) : image_of(cur_g.b, cur_g.chunk + 1);
    const int lane_w_nxt = have_next ? lane_w : (int)OOB;
    const u32x4* lb = lb0 + cur * BUF;
    if constexpr (RS) {
      // group q = (dz, dx): halo rows y' = 0..7 of plane wave + dz at x shift dx; filter fragments of taps (dz, dy = 0..4, dx)
      auto grp_off = [&](int q, int yy) -> int { return ((q / 5) * BYH + yy) * BXH + q % 5; };
      auto grp_tap = [&](int q, int dy) -> int { return ((q / 5) * 5 + dy) * 5 + q % 5; };
      u32x4 bg[2][8];
#pragma unroll
      for (int yy = 0; yy < 8; ++yy) bg[0][yy] = lb[grp_off(0, yy)];
#pragma unroll
      for (int q = 0; q < 25; ++q) {
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
#pragma unroll
          for (int vs = 0; vs < VW; ++vs) {
            Elem<bf16_t>::mma(aq[0][0][dy], bg[q & 1][vs + dy], acc[0][vs]);
            // the next group's eight voxel fragments in the shadows of this group's first eight MFMAs
            if (q + 1 < 25 && dy * VW + vs < 8) bg[(q + 1) & 1][dy * VW + vs] = lb[grp_off(q + 1, dy * VW + vs)];
            __builtin_amdgcn_sched_barrier(0);
          }
          if (q + RA < 25) aq[RA][0][dy] = wfrag(cur_g, 0, grp_tap(q + RA, dy), lane_w);
          else aq[RA][0][dy] = wfrag(nxt_g, 0, grp_tap(q + RA - 25, dy), lane_w_nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (q == 12) RM_STAMP(st_ * 4 + 1);
#pragma unroll
        for (int k = 0; k < RA; ++k)
#pragma unroll
          for (int dy = 0; dy < 5; ++dy) aq[k][0][dy] = aq[k + 1][0][dy];
      }
    } else {
    auto tap_off = [&](int t) -> int { return (((t / 5) / 5) * BYH + ((t / 5) % 5)) * BXH + t % 5; };
    u32x4 bq[TA + 1][VW];
#pragma unroll
    for (int k = 0; k < TA; ++k)
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) bq[k][vs] = lb[vs * SUBROWS * BXH + tap_off(k)];
#pragma unroll
    for (int r = 0; r < 25; ++r) {
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int t = r * 5 + dx;
#pragma unroll
        for (int vs = 0; vs < VW; ++vs) {
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) Elem<bf16_t>::mma(aq[0][cs][dx], bq[t % (TA + 1)][vs], acc[cs][vs]);
          if (t + TA < 125) bq[(t + TA) % (TA + 1)][vs] = lb[vs * SUBROWS * BXH + tap_off(t + TA)];
          __builtin_amdgcn_sched_barrier(0);
        }
        // this tap's filter fragment of the row RA ahead (the last RA rows: the next image's first rows; no next image:
        // out-of-range offsets, zeros without memory traffic -- no branch in the unrolled loop)
        if (r + RA < 25) {
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) aq[RA][cs][dx] = wfrag(cur_g, cs, (r + RA) * 5 + dx, lane_w);
        } else {
#pragma unroll
          for (int cs = 0; cs < CW; ++cs) aq[RA][cs][dx] = wfrag(nxt_g, cs, (r + RA - 25) * 5 + dx, lane_w_nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (r == 12) RM_STAMP(st_ * 4 + 1);
#pragma unroll
      for (int k = 0; k < RA; ++k)
#pragma unroll
        for (int cs = 0; cs < CW; ++cs)
#pragma unroll
          for (int dx = 0; dx < 5; ++dx) aq[k][cs][dx] = aq[k + 1][cs][dx];
    }
    }
    RM_STAMP(st_ * 4 + 2);
    // every MFMA wave is done with image `cur`, every loader wave has stored the next one
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    RM_STAMP(st_ * 4 + 3);

    if (last_chunk) {
      // ---- epilogue of the brick (the stores drain under the next brick's taps).  32x32 C/D layout: column = lane & 31
      // (voxel), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (output channels)
      const int z0 = cur_g.b.z0, y0 = cur_g.b.y0, x0 = cur_g.b.x0, n_out = cur_g.b.n, cot = cur_g.b.cot;
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) {
        // voxel m = (wave VW + vs) 32 + l31 of the brick
        const int gz = z0 + wave, gy = y0 + vs * SUBROWS + (BXT == 32 ? 0 : (l31 >> 4)), gx = x0 + (BXT == 32 ? l31 : (l31 & 15));
        const bool inside = gz < D && gy < H && gx < W;
        const size_t vox = ((size_t)(n_out * D + gz) * H + gy) * W + gx;
#pragma unroll
        for (int cs = 0; cs < CW; ++cs) {
          if constexpr (PLAIN) {
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
              const int co16 = cot * C::COT + cs * 32 + 16 * qp;
              const uint32_t p00 = pack_bf16x2(acc[cs][vs][8 * qp + 0], acc[cs][vs][8 * qp + 1]);
              const uint32_t p01 = pack_bf16x2(acc[cs][vs][8 * qp + 2], acc[cs][vs][8 * qp + 3]);
              const uint32_t p10 = pack_bf16x2(acc[cs][vs][8 * qp + 4], acc[cs][vs][8 * qp + 5]);
              const uint32_t p11 = pack_bf16x2(acc[cs][vs][8 * qp + 6], acc[cs][vs][8 * qp + 7]);
              const auto r0 = __builtin_amdgcn_permlane32_swap(p00, p10, false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(p01, p11, false, false);
              const bool out2 = Cout1 > 0 && co16 >= Cout1;
              const int Cout_ = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
              const int co = (out2 ? co16 - Cout1 : co16) + 8 * khalf;
              bf16_t* yp = static_cast<bf16_t*>(out2 ? a.y2 : a.y) + vox * Cout_ + co;
              if (inside && co16 < Cout) *reinterpret_cast<u32x4*>(yp) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
          } else if (a.wide) {
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
              const int co16 = cot * C::COT + cs * 32 + 16 * qp;
              if (co16 >= Cout) continue;
              uint32_t pk[2][2];
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int q = 2 * qp + g;
                float v0 = acc[cs][vs][4 * q + 0], v1 = acc[cs][vs][4 * q + 1];
                float v2 = acc[cs][vs][4 * q + 2], v3 = acc[cs][vs][4 * q + 3];
                if (a.bias) {
                  const float* bp = a.bias + co16 + 8 * g + 4 * khalf;
                  v0 += bp[0]; v1 += bp[1]; v2 += bp[2]; v3 += bp[3];
                }
                if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                pk[g][0] = pack_bf16x2(v0, v1);
                pk[g][1] = pack_bf16x2(v2, v3);
              }
              const auto r0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
              const auto r1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
              if (!inside) continue;
              const bool out2 = Cout1 > 0 && co16 >= Cout1;
              const int Cout_ = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
              const int co = (out2 ? co16 - Cout1 : co16) + 8 * khalf;
              bf16_t* yp = static_cast<bf16_t*>(out2 ? a.y2 : a.y) + vox * Cout_ + co;
              *reinterpret_cast<u32x4*>(yp) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int co_all = cot * C::COT + cs * 32 + 8 * q + 4 * khalf;
              if (co_all >= Cout || !inside) continue;
              const bool out2 = Cout1 > 0 && co_all >= Cout1;
              const int Cout_ = Cout1 > 0 ? (out2 ? Cout - Cout1 : Cout1) : Cout;
              const int co = out2 ? co_all - Cout1 : co_all;
              float v0 = acc[cs][vs][4 * q + 0], v1 = acc[cs][vs][4 * q + 1];
              float v2 = acc[cs][vs][4 * q + 2], v3 = acc[cs][vs][4 * q + 3];
              if (a.bias) {
                v0 += a.bias[co_all];
                if (co_all + 1 < Cout) v1 += a.bias[co_all + 1];
                if (co_all + 2 < Cout) v2 += a.bias[co_all + 2];
                if (co_all + 3 < Cout) v3 += a.bias[co_all + 3];
              }
              if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
              const uint32_t p01 = pack_bf16x2(v0, v1), p23 = pack_bf16x2(v2, v3);
              bf16_t* yp = static_cast<bf16_t*>(out2 ? a.y2 : a.y) + vox * Cout_ + co;
              if ((Cout_ & 3) == 0) {
                *reinterpret_cast<u32x2*>(yp) = u32x2{p01, p23};
              } else {
                yp[0] = (bf16_t)(p01 & 0xffffu);
                if (co + 1 < Cout_) yp[1] = (bf16_t)(p01 >> 16);
                if (co + 2 < Cout_) yp[2] = (bf16_t)(p23 & 0xffffu);
                if (co + 3 < Cout_) yp[3] = (bf16_t)(p23 >> 16);
              }
            }
          }
        }
      }
#pragma unroll
      for (int cs = 0; cs < CW; ++cs)
#pragma unroll
        for (int vs = 0; vs < VW; ++vs)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[cs][vs][r] = 0.f;
      ++item;
    }
    if (!have_next) break;
    cur_g = nxt_g;
    cur ^= 1;
#ifdef RM_CONV_TIMING
    ++st_;
#endif
  }
}

// eligibility + launch of the pipelined form (see the kernel's comment); returns -1 when the launch is not its kind.
// REPMODE_CONV_PIPE / repmode_set_conv_pipe: bit 0 = on, bit 1 = one channel sub-tile per wave everywhere, bit 2 = also on grids
// smaller than the chip (the parity tests' volumes), bit 3 = the wave-specialised kernel (conv5_ws_kernel), bit 4 = its items
// along z first (HBM reads per launch at batch 8: 87.4 -> 61.0 MB, 64->32 401.5 -> 396.6 us), bit 5 = row-stationary tap
// order on the 32-channel layers (8 LDS reads per 20 MFMAs: level 0 200.5 -> 189.4 us, 64->32 403.1 -> 381.1; the kernel is
// power-limited and an LDS read costs energy); default 57.  Same box, interleaved, us per launch two-workgroup form /
// pipelined: 32->32 (level 0) 227.6 / 216.7, 64->32 460.0 / 433.5, 64->64 (level 1) 116.4 / 98.1, 128->64 224.1 / 189.2
// (one sub-tile per wave: 106.2, 206.8); conv5 launches of the train step 3945 -> 3702 us.  The step itself moves less
// (11.82 -> 11.75 ms): with the convolutions drawing more power every other kernel of the step runs 2-6 % slower
// (profiles/r03_pipe_ab.txt) -- the chip is power-limited over the step, not per kernel.
static int g_pipe = []() { const char* e = getenv("REPMODE_CONV_PIPE"); return e ? atoi(e) : 121; }();

int device_cus() {
  // per device (advisor round 3: a function-static count belonged to whichever device called first)
  static std::atomic<int> cus_of[32];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int c = cus_of[dev & 31].load(std::memory_order_relaxed);
  if (c <= 0) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    cus_of[dev & 31].store(c, std::memory_order_relaxed);
  }
  const int r = repmode_reserve_cus();           // (CUs left free for a communication kernel beside the persistent grid)
  return c - r > 8 ? c - r : (c > 8 ? 8 : c);
}

// The shape side of the pipelined form's eligibility (what does not depend on the call's epilogue / tap flags): the brick
// width (32, or 16 on volumes 16..31 voxels wide: bit 6), channel sub-tiles per wave, item count.  false: not its kind.
struct PipePlan { int bx, cw, nbz, nby, nbx, ncot; long nitems; };
bool pipe_plan(int N, int D, int H, int W, int Cin, int CoutP, PipePlan* p) {
  if (!g_pipe || W < 16 || D < 4 || H < 4 || (Cin & 7) != 0) return false;
  const bool x16 = W < 32;
  if (x16 && (g_pipe & (8 | 64)) != (8 | 64)) return false;            // (16-voxel bricks: the wave-specialised kernel only)
  const int cus = device_cus();
  p->bx = x16 ? 16 : 32;
  p->nbz = ceil_div(D, 4);
  p->nby = ceil_div(H, 4);
  p->nbx = ceil_div(W, p->bx);
  const long bricks = (long)N * p->nbz * p->nby * p->nbx;
  int cw = (g_pipe & 2) ? 1 : (CoutP >= 64 ? 2 : 1);
  if (x16 && cw == 2) {
    // 16-voxel bricks: an item of two channel sub-tiles is twice as long; take it only where the shorter grid does not cost
    // rounds (batch 8 of level 2: 128 two-tile items would leave half of the chip idle, 256 one-tile items fill it).
    // Priced in one-tile item units: ceil(items / CUs) per workgroup, a two-tile item at 1.7 (it reads LDS half as often).
    const long i1 = bricks * ceil_div(CoutP, 32), i2 = bricks * ceil_div(CoutP, 64);
    const double t1 = (double)((i1 + cus - 1) / cus), t2 = 1.7 * (double)((i2 + cus - 1) / cus);
    if (t1 <= t2) cw = 1;
  }
  p->cw = cw;
  p->ncot = ceil_div(CoutP, 32 * cw);
  p->nitems = bricks * p->ncot;
  // under-filled launches keep the two-workgroup form (bit 2: the parity tests' small volumes); a 16-voxel launch from half a chip
  const long need = (g_pipe & 4) ? 1 : (x16 ? cus / 2 : cus);
  return p->nitems >= need && p->nitems < (1L << 30);
}

template <int BXT>
int launch_pipe_bx(ConvArgs a, const PipePlan& pl, hipStream_t stream) {
  using C1 = Cfg<4, 4, BXT, 4, 1, BXT == 32 ? 4 : 2, 1>;
  const int cus = device_cus();
  repmode_tail_take(stream, &a.tail);
  const int nwg = (int)(pl.nitems < cus ? pl.nitems : cus);
  const long grid = nwg + a.tail.nblocks;
  constexpr int LDS_BYTES = 2 * C1::LDS_BYTES;
  static_assert(LDS_BYTES >= TAIL_LDS_BYTES && LDS_BYTES <= 160 * 1024, "two halo images per workgroup");
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  if (!((attr_set.load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
    if constexpr (BXT == 32) {
      RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_pipe_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
      RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_pipe_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
      RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_ws_kernel<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
      RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_ws_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    }
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_ws_kernel<1, true, false, BXT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_ws_kernel<1, false, false, BXT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_ws_kernel<2, true, false, BXT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_ws_kernel<2, false, false, BXT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set.fetch_or(1u << (dev & 31), std::memory_order_release);
  }
  const int cw = pl.cw;
  const int nitems = (int)pl.nitems;
  const double alg = 2.0 * a.N * a.D * a.H * a.W * (double)a.Cin * a.Cout * REPMODE_TAPS;
  repmode_prof_begin(REPMODE_PROF_CONV5_WS, alg, stream);
  const dim3 g((unsigned)grid);
  if ((g_pipe & 8) || BXT == 16) {       // the wave-specialised form
    const bool plain = a.wide && !a.bias && !a.relu;
    const bool rs = BXT == 32 && (g_pipe & 32);
    if (cw == 2 && plain) hipLaunchKernelGGL((conv5_ws_kernel<2, true, false, BXT>), g, dim3(512), LDS_BYTES, stream, a, nitems);
    else if (cw == 2) hipLaunchKernelGGL((conv5_ws_kernel<2, false, false, BXT>), g, dim3(512), LDS_BYTES, stream, a, nitems);
    else if (plain && rs) { if constexpr (BXT == 32) hipLaunchKernelGGL((conv5_ws_kernel<1, true, true>), g, dim3(512), LDS_BYTES, stream, a, nitems); }
    else if (plain) hipLaunchKernelGGL((conv5_ws_kernel<1, true, false, BXT>), g, dim3(512), LDS_BYTES, stream, a, nitems);
    else if (rs) { if constexpr (BXT == 32) hipLaunchKernelGGL((conv5_ws_kernel<1, false, true>), g, dim3(512), LDS_BYTES, stream, a, nitems); }
    else hipLaunchKernelGGL((conv5_ws_kernel<1, false, false, BXT>), g, dim3(512), LDS_BYTES, stream, a, nitems);
  } else if constexpr (BXT == 32) {
    if (cw == 2) hipLaunchKernelGGL(conv5_pipe_kernel<2>, g, dim3(256), LDS_BYTES, stream, a, nitems);
    else hipLaunchKernelGGL(conv5_pipe_kernel<1>, g, dim3(256), LDS_BYTES, stream, a, nitems);
  }
  repmode_prof_end(stream);
  RM_LAUNCH_CHECK("conv5_pipe");
  return REPMODE_OK;
}

int launch_pipe(ConvArgs a, hipStream_t stream) {
  if (a.dual || a.dxc || a.tap_lo != 0 || a.tap_hi != 4 || a.stats || a.out_f32) return -1;
  PipePlan pl;
  if (!pipe_plan(a.N, a.D, a.H, a.W, a.Cin, a.CoutP, &pl)) return -1;
  // one descriptor spans each tensor: 32-bit byte offsets
  const size_t vox = (size_t)a.N * a.D * a.H * a.W;
  if (vox * (size_t)(a.Cin1 > 0 ? (a.Cin1 > a.Cin - a.Cin1 ? a.Cin1 : a.Cin - a.Cin1) : a.Cin) * 2 >= ((size_t)1 << 31)) return -1;
  // filter offsets are 32-bit too (the number of slots is sample data: at most one per sample)
  if ((size_t)a.N * REPMODE_TAPS * a.CoutP * a.CinP * 2 >= ((size_t)1 << 31)) return -1;
  a.nbz = pl.nbz; a.nby = pl.nby; a.nbx = pl.nbx; a.ncot = pl.ncot;
  a.ksplit = 1;
  a.zfast = (g_pipe & 16) ? 1 : 0;
  return pl.bx == 32 ? launch_pipe_bx<32>(a, pl, stream) : launch_pipe_bx<16>(a, pl, stream);
}

template <typename T, typename C, bool SWAP, bool PAIR, bool DXC = false, bool ROWSTAT = false, bool MERGE = false>
int launch_cfg(ConvArgs a, hipStream_t stream) {
  a.nbz = ceil_div(a.D, C::BZ);
  a.nby = ceil_div(a.H, C::BY);
  a.nbx = ceil_div(a.W, C::BX);
  a.ncot = ceil_div(a.CoutP, C::COT);
  // split the input-channel reduction (float output, f32 atomics) until the grid can fill the chip: two
  // workgroups per CU, and at least two channel chunks per workgroup -- every slice pays one un-overlapped halo
  // staging and a full tile of atomics (same-box sweep: 1024 -> 512 workgroups is +16..24 % on levels 2-3)
  const int nchunks = a.CinP / (2 * Elem<T>::KV);
  long base = (long)(a.dual ? 2 * a.N : a.N) * a.nbz * a.nby * a.nbx * a.ncot;
  int ks = 1;
  static const int min_chunks = []() { const char* e = getenv("REPMODE_CONV_MIN_CHUNKS"); return e ? atoi(e) : 4; }();
  static const int split_target = []() { const char* e = getenv("REPMODE_CONV_SPLIT_TARGET"); return e ? atoi(e) : CONV_SPLIT_TARGET; }();
  // (deterministic mode: at most TWO addends per element of the cleared output -- a + b is commutative, three slices'
  // atomics would meet in any order; a dual launch whose two jobs share the output already has its two)
  const int ks_max = !repmode_deterministic() ? 64 : (a.dual && !(a.dual & 4)) ? 1 : repmode_det_cap(RM_DET_CONV);
  if (SWAP && !a.bias && !a.relu) {   // (a bias / ReLU epilogue needs the whole sum in one workgroup)
    while (ks * min_chunks <= nchunks && base * ks < split_target && ks < ks_max) ks *= 2;
  }
  RM_REQUIRE(!a.stats || !SWAP, "conv5: output statistics need the element-typed output path (bf16 input, not out_f32)");
  a.ksplit = ks;
  a.ksplit2 = 0;
  long nblocks = base * ks;
  // dual-expert launch: per-job split factors (see the kernel) -- the 5x5x5 job twice as finely split as the 3x3x3 job where
  // it still leaves a workgroup a whole chunk and the grid stays within two workgroups per CU.  Built, measured, OFF
  // (REPMODE_DUAL_KS=1 to try; same box, interleaved, us per forward launch one factor / per-job factors,
  // profiles/r04_dual_ks.txt): level 3 128->256 42.2 / 43.9, 256->256 60.8 / 59.9, 512->256 111.3 / 104.7; level 4 256->512
  // 28.0 / 39.6, 512->512 47.5 / 82.5; train step 10.24 / 10.28 ms.  The 5x5x5 job's extra slices pay a halo staging and a
  // tile of float atomics each; on level 4 (one 32-voxel brick per sample) that is most of a workgroup's time.
  static const int dual_ks = []() { const char* e = getenv("REPMODE_DUAL_KS"); return e ? atoi(e) : 0; }();
  if (a.dual && dual_ks && SWAP && !repmode_deterministic()) {
    const long per_job = base / 2;                        // (brick, channel tile) units of one job
    int k3 = ks, k5 = ks;
    if (2 * ks <= nchunks && per_job * 3 * ks <= 2 * device_cus()) k5 = 2 * ks;                 // finer 5x5x5 job, same 3x3x3 job
    else if (ks >= 2) k3 = ks / 2;                                                              // coarser 3x3x3 job
    if (k5 != k3) {
      a.ksplit = k5; a.ksplit2 = k3;
      nblocks = per_job * (k5 + k3);
    }
  }
  // small jobs deferred to this launch (tail_jobs.h) become its first workgroups
  repmode_tail_take(stream, &a.tail);
  const long grid = nblocks + a.tail.nblocks;
  RM_REQUIRE(grid > 0 && grid < (1L << 31), "conv5: grid %ld out of range", grid);
  constexpr int LDS_BYTES = C::LDS_BYTES > TAIL_LDS_BYTES ? C::LDS_BYTES : TAIL_LDS_BYTES;   // (level 4's tile is smaller than a job's)
  const int lds_bytes = a.tail.nblocks ? LDS_BYTES : C::LDS_BYTES;
  // per instantiation and per device (one bit each; a racing second thread at worst repeats the idempotent call)
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  if (!((attr_set.load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_igemm_kernel<T, C, SWAP, PAIR, DXC, ROWSTAT, MERGE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set.fetch_or(1u << (dev & 31), std::memory_order_release);
  }
  if ((ks > 1 || a.ksplit > 1 || (a.dual && !(a.dual & 4))) && !a.accum) {
    const size_t vox = (size_t)((a.dual & 4) ? 2 * a.N : a.N) * a.D * a.H * a.W;
    RM_HIP(hipMemsetAsync(a.y, 0, vox * (a.Cout1 > 0 ? a.Cout1 : a.Cout) * sizeof(float), stream));
    if (a.Cout1 > 0) RM_HIP(hipMemsetAsync(a.y2, 0, vox * (a.Cout - a.Cout1) * sizeof(float), stream));
  }
  // algorithmic FLOPs = the layer's merged 125-tap convolution, counted ONCE per layer and direction: a
  // launch restricted to the 3x3x3 support is the per-expert formulation's second conv of a layer whose 125 taps
  // the 5x5x5 expert's launch has already been credited with -- executed work, not algorithmic: 0
  // (dx-centre mode: 25 taps x the 5 folded x taps of the thin dimension = the original layer's 125 taps x 1 channel)
  const double alg = a.dxc ? 2.0 * a.N * a.D * a.H * a.W * 25.0 * (a.Cin == 8 ? 5.0 * a.Cout : (double)a.Cin * a.Cout)
                           : a.tap_lo ? 0.0 : 2.0 * a.N * a.D * a.H * a.W * (double)a.Cin * a.Cout * REPMODE_TAPS;
  repmode_prof_begin(REPMODE_PROF_CONV5, alg, stream);
  hipLaunchKernelGGL((conv5_igemm_kernel<T, C, SWAP, PAIR, DXC, ROWSTAT, MERGE>), dim3((unsigned)grid), dim3(C::NT), lds_bytes, stream, a);
  repmode_prof_end(stream);
  RM_LAUNCH_CHECK("conv5_igemm");
  return REPMODE_OK;
}

// Tile menu.           BZ BY BX  WV WC VW CW
using CfgX32 = Cfg<4, 4, 32, 4, 1, 4, 1>;      // 512 voxels x 32 channels   (levels 0-1)
using CfgX16 = Cfg<4, 4, 16, 2, 2, 4, 1>;      // 256 voxels x 64 channels   (level 2)
using CfgX8 = Cfg<4, 8, 8, 2, 2, 4, 1>;        // 256 voxels x 64 channels   (level 3)
using CfgX4 = Cfg<2, 4, 4, 1, 4, 1, 1>;        // 32 voxels x 128 channels   (level 4)

// REPMODE_CONV_ROWSTAT=1: the row-stationary tap loop on the 4 x 4 x 32 tile (full 5x5x5 support only).  Built, measured,
// OFF: same box, interleaved, us per launch tap-major / row-stationary: 32->32 237.2 / 256.1, 64->32 471.1 / 498.7,
// 64->64 (level 1) 119.8 / 126.1, 128->64 228.7 / 234.2; train step 12.49 / 12.68 ms.  2.5 x fewer LDS reads buy nothing
// (the LDS is not what the tap loop waits for) and the 104 registers of double buffers cost a spill in the staging phase.
static const int g_rowstat = []() { const char* e = getenv("REPMODE_CONV_ROWSTAT"); return e ? atoi(e) : 0; }();
// REPMODE_CONV_WIDE (default 1): 16-byte bf16 stores through v_permlane32_swap (same box, interleaved: 237.2 -> 235.2 us on
// 32->32 at level 0, 471.1 -> 467.0 on 64->32; train step 12.49 -> 12.44 ms)
static const int g_wide = []() { const char* e = getenv("REPMODE_CONV_WIDE"); return e ? atoi(e) : 1; }();

// Experiment (REPMODE_CONV_X16_AT=<cout>): layers at least that wide in output channels take the 4 x 4 x 16 tile (256 voxels x 64
// channels: two channel sub-tiles share a staged image) also when the volume is 32 or more voxels wide
static const int g_x16_at = []() { const char* e = getenv("REPMODE_CONV_X16_AT"); return e ? atoi(e) : 0; }();
// (Measured and removed: the 4 x 4 x 32 brick on 512 threads, four waves per SIMD at 128 registers -- 260 vs 222 us on level 0,
// 130 vs 113 on level 1; profiles/r03_pipe_ab.txt.  More waves is not what the tap loop lacks.)

template <typename T, bool SWAP, bool PAIR>
int dispatch_tile(ConvArgs a, hipStream_t stream) {
  if constexpr (sizeof(T) == 2 && !SWAP) {
    const int rc = launch_pipe(a, stream);
    if (rc != -1) return rc;
  }
  if (a.W >= 32 && g_x16_at > 0 && a.Cout >= g_x16_at) return launch_cfg<T, CfgX16, SWAP, PAIR>(a, stream);
  if (a.W >= 32) {
    if (g_rowstat && a.tap_lo == 0 && a.tap_hi == 4 && !a.dual) return launch_cfg<T, CfgX32, SWAP, PAIR, false, true>(a, stream);
    return launch_cfg<T, CfgX32, SWAP, PAIR>(a, stream);
  }
  if (a.W >= 16) return launch_cfg<T, CfgX16, SWAP, PAIR>(a, stream);
  if (a.W >= 8) return launch_cfg<T, CfgX8, SWAP, PAIR>(a, stream);
  return launch_cfg<T, CfgX4, SWAP, PAIR>(a, stream);
}

template <typename T, bool SWAP>
int dispatch(ConvArgs a, hipStream_t stream) {
  if (a.dxc) {
    // the thin layers (x taps folded into channels): bf16, one tensor in and out, on the two widest tiles
    if constexpr (sizeof(T) == 2) {
      RM_REQUIRE(a.Cin1 == 0 && a.Cout1 == 0, "conv5: the dx-centre mode takes one input and one output tensor");
      if (a.W >= 32) return launch_cfg<T, CfgX32, SWAP, false, true>(a, stream);
      return launch_cfg<T, CfgX16, SWAP, false, true>(a, stream);
    } else {
      RM_REQUIRE(false, "conv5: the dx-centre mode is a bf16 path");
    }
  }
  if (a.Cin1 > 0 || a.Cout1 > 0) return dispatch_tile<T, SWAP, true>(a, stream);
  return dispatch_tile<T, SWAP, false>(a, stream);
}

}  // namespace

extern "C" int repmode_set_conv_pipe(int mode) { g_pipe = mode; return REPMODE_OK; }
extern "C" int repmode_get_conv_pipe(void) { return g_pipe; }

// 1: a 5x5x5 convolution of this shape writes its element-typed (bf16) output with the whole channel reduction inside one
// workgroup at a grid that fills the chip -- volumes 32 or more voxels wide, and (round 4) 16-wide volumes with enough bricks
// for the wave-specialised kernel's 16-voxel form; 0: the caller should ask for the float output, whose reduction is split over
// workgroups (float atomics) until the grid fills the chip.  The operator library asks per layer and direction.
extern "C" int repmode_conv5_elem_out(int n, int d, int h, int wdim, int cin, int cout, int dtype) {
  if (dtype != REPMODE_BF16 || n <= 0 || d <= 0 || h <= 0 || wdim <= 0 || cin <= 0 || cout <= 0) return 0;
  if (wdim >= 32) return 1;
  PipePlan pl;
  return pipe_plan(n, d, h, wdim, cin, round_up(cout, 32), &pl) ? 1 : 0;
}

extern "C" int repmode_padded_channels(int channels, int dtype, int is_reduction_dim) {
  if (channels <= 0) return 0;
  if (!is_reduction_dim) return round_up(channels, 32);
  return round_up(channels, dtype == REPMODE_BF16 ? 16 : 8);
}

extern "C" int repmode_conv5_ex(const void* x, const void* w, const int32_t* sample_slot, void* y, int n,
                                int d, int h, int wdim, int cin, int cout, int dtype, int out_f32,
                                int centre3, void* stream);
static int conv5_common(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot, void* y, void* y2,
                        int cout1, int n, int d, int h, int wdim, int cin, int cout, int dtype, int out_f32, int flags,
                        const float* bias, int relu, int want_stats, int* stats_half, void* stream);

extern "C" int repmode_conv5(const void* x, const void* w, const int32_t* sample_slot, void* y, int n,
                             int d, int h, int wdim, int cin, int cout, int dtype, int out_f32,
                             void* stream) {
  return repmode_conv5_ex(x, w, sample_slot, y, n, d, h, wdim, cin, cout, dtype, out_f32, 0, stream);
}

extern "C" int repmode_conv5_pair(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot,
                                  void* y, void* y2, int cout1, int n, int d, int h, int wdim, int cin, int cout,
                                  int dtype, int out_f32, int flags, void* stream);
extern "C" int repmode_conv5_epi(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot, void* y,
                                 int n, int d, int h, int wdim, int cin, int cout, int dtype, int out_f32, int flags,
                                 const float* bias, int relu, int want_stats, int* stats_half, void* stream);

extern "C" int repmode_conv5_ex(const void* x, const void* w, const int32_t* sample_slot, void* y, int n,
                                int d, int h, int wdim, int cin, int cout, int dtype, int out_f32,
                                int centre3, void* stream) {
  return repmode_conv5_pair(x, nullptr, 0, w, sample_slot, y, nullptr, 0, n, d, h, wdim, cin, cout, dtype, out_f32,
                            centre3, stream);
}

// The same convolution with the input and / or the output channels split over two tensors -- a U-Net skip
// connection without the concatenated copy (RepMode.py:106 `torch.cat((x_skip, up), 1)`): input channels
// [0, cin1) are read from x ([N][D][H][W][cin1]), [cin1, cin) from x2; output channels [0, cout1) are written to y
// ([...][cout1]), the rest to y2 (the data gradient of such a layer).  cin1 == 0 / cout1 == 0: one tensor.
// cin1 must be a multiple of 16 (bf16) / 8 (f32) and cin - cin1 of 8 / 4; cout1 of 32.
extern "C" int repmode_conv5_pair(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot,
                                  void* y, void* y2, int cout1, int n, int d, int h, int wdim, int cin, int cout,
                                  int dtype, int out_f32, int flags, void* stream) {
  return conv5_common(x, x2, cin1, w, sample_slot, y, y2, cout1, n, d, h, wdim, cin, cout, dtype, out_f32, flags, nullptr, 0, 0,
                      nullptr, stream);
}

// The forward conv of a MoDE block with an epilogue that takes over part of the BatchNorm3d + ReLU behind it
// (RepMode.py:146-149, 212).  One or two input tensors (cin1 as repmode_conv5_pair), one output.
//   bias != NULL / relu: y = max(conv + bias[co], 0): an eval-mode BatchNorm whose scale was folded into the merged filter.
//   want_stats (bf16 input, element-typed output only): per-channel sum / sum of squares of the stored outputs go to the
//     library's BatchNorm scratch; *stats_half receives the half to hand to repmode_bn_relu_fwd_ex, which then skips its
//     own statistics pass.  The two calls must follow each other on the same stream.
extern "C" int repmode_conv5_epi(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot, void* y,
                                 int n, int d, int h, int wdim, int cin, int cout, int dtype, int out_f32, int flags,
                                 const float* bias, int relu, int want_stats, int* stats_half, void* stream) {
  RM_REQUIRE(!want_stats || stats_half, "conv5_epi: stats_half must be given with want_stats");
  RM_REQUIRE(!want_stats || (dtype == REPMODE_BF16 && !out_f32), "conv5_epi: statistics need bf16 in and out");
  RM_REQUIRE(!want_stats || cout <= 512, "conv5_epi: statistics for at most 512 channels");
  RM_REQUIRE(!(flags & 2) || !(bias || relu), "conv5_epi: bias / ReLU cannot be applied to an accumulating output");
  return conv5_common(x, x2, cin1, w, sample_slot, y, nullptr, 0, n, d, h, wdim, cin, cout, dtype, out_f32, flags, bias, relu,
                      want_stats, stats_half, stream);
}

static int conv5_common(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot, void* y, void* y2,
                        int cout1, int n, int d, int h, int wdim, int cin, int cout, int dtype, int out_f32, int flags,
                        const float* bias, int relu, int want_stats, int* stats_half, void* stream) {
  RM_REQUIRE(x && w && sample_slot && y, "conv5: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0, "conv5: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "conv5: bad dtype %d", dtype);
  RM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
             ((uintptr_t)x2 & 15) == 0 && ((uintptr_t)y2 & 15) == 0, "conv5: pointers must be 16-byte aligned");
  const int kv = dtype == REPMODE_F32 ? 4 : 8;
  RM_REQUIRE(cin1 == 0 || (x2 && cin1 > 0 && cin1 < cin && cin1 % (2 * kv) == 0 && (cin - cin1) % kv == 0),
             "conv5_pair: bad input split %d of %d", cin1, cin);
  RM_REQUIRE(cout1 == 0 || (y2 && cout1 > 0 && cout1 < cout && cout1 % 32 == 0 && (cout - cout1) % 4 == 0),
             "conv5_pair: bad output split %d of %d", cout1, cout);
  RM_REQUIRE((size_t)d * h * wdim * cin * (dtype == REPMODE_F32 ? 4 : 2) < ((size_t)1 << 31),
             "conv5: one sample of the input must be smaller than 2 GiB (32-bit buffer offsets)");
  ConvArgs a{};
  a.x = x; a.w = w; a.sample_slot = sample_slot; a.y = y;
  a.x2 = x2; a.y2 = y2; a.Cin1 = cin1; a.Cout1 = cout1;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin; a.Cout = cout;
  a.CinP = repmode_padded_channels(cin, dtype, 1);
  a.CoutP = repmode_padded_channels(cout, dtype, 0);
  a.out_f32 = (out_f32 != 0) || dtype == REPMODE_F32;
  // `flags`: bit 0 = 3x3x3 support, bit 1 = accumulate into a float y, bit 2 = centre x tap only
  a.tap_lo = (flags & 1) ? 1 : 0;
  a.tap_hi = (flags & 1) ? 3 : 4;
  a.accum = (flags & 2) ? 1 : 0;
  a.dxc = (flags & 4) ? 1 : 0;
  // bits 3-5: dual-expert launch (see ConvArgs::dual): 8 = on, 16 = the input holds 2 n samples, 32 = the output does
  a.dual = (flags & 8) ? (1 | ((flags & 16) ? 2 : 0) | ((flags & 32) ? 4 : 0)) : 0;
  RM_REQUIRE(!a.dual || (a.out_f32 && !(flags & 1) && !a.dxc && cin1 == 0 && cout1 == 0 && !bias && !relu && !want_stats),
             "conv5: a dual-expert launch is a float-output, full-support, one-tensor convolution");
  RM_REQUIRE(!a.accum || a.out_f32, "conv5: accumulation needs a float output");
  hipStream_t s = static_cast<hipStream_t>(stream);
  a.bias = bias;
  a.relu = relu ? 1 : 0;
  if (want_stats) {
    float* scratch = repmode_zero_scratch(s);
    if (!scratch) return REPMODE_ELAUNCH;
    const int half = repmode_bn_scratch_half(s);
    a.stats = scratch + (size_t)half * REPMODE_SCRATCH_BN_HALF;
    a.stats_clear = scratch + (size_t)(1 - half) * REPMODE_SCRATCH_BN_HALF;
    *stats_half = half;
  }
  if (dtype == REPMODE_F32) return dispatch<float, true>(a, s);
  if (a.out_f32) return dispatch<bf16_t, true>(a, s);
  RM_REQUIRE(!want_stats || !repmode_deterministic(), "conv5_epi: the statistics epilogue adds with atomics; not in deterministic mode");
  a.wide = g_wide && !want_stats && (cout1 > 0 ? (cout1 % 16 == 0 && (cout - cout1) % 16 == 0) : cout % 16 == 0);
  return dispatch<bf16_t, false>(a, s);
}

#ifdef RM_CONV_TIMING
extern "C" int repmode_debug_conv_timing(unsigned long long* out) {
  RM_HIP(hipDeviceSynchronize());
  RM_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_timing), sizeof(unsigned long long) * 64 * 64));
  return 0;
}
#endif

// EXPERIMENT: the forward convolution of a merged-formulation MoDE block with
// GatRep inside the kernel -- no merged filter in HBM.  w2: repmode_expert_frags' two slots (shared by all samples), k1 / a3 /
// a5: the 1x1 experts' parameters [cout][cin] float, gates: [nslots][5][cout] float (repmode_gate_softmax), y: float
// [n][d][h][w][cout] (flags bit 1: add to y, which is zero already; else cleared / overwritten).  bf16, the 4 x 4 x 16 tile
// (level 2 of the network).  Measured against repmode_gatrep_fwd + repmode_conv5 by tools/merge_ab.py; not on the product path.
extern "C" int repmode_conv5_merged(const void* x, const void* w2, const float* k1, const float* a3, const float* a5, const float* gates,
                                    const int32_t* sample_slot, float* y, int n, int d, int h, int wdim, int cin, int cout, int flags,
                                    void* stream) {
  RM_REQUIRE(x && w2 && k1 && a3 && a5 && gates && sample_slot && y, "conv5_merged: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0 && cin % 8 == 0, "conv5_merged: bad shape (cin %% 8 == 0)");
  ConvArgs a{};
  a.x = x; a.w = w2; a.sample_slot = sample_slot; a.y = y;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin; a.Cout = cout;
  a.CinP = repmode_padded_channels(cin, REPMODE_BF16, 1);
  a.CoutP = repmode_padded_channels(cout, REPMODE_BF16, 0);
  a.out_f32 = 1;
  a.tap_lo = 0; a.tap_hi = 4;
  a.accum = (flags & 2) ? 1 : 0;
  a.mk1 = k1; a.ma3 = a3; a.ma5 = a5; a.gates = gates;
  return launch_cfg<bf16_t, CfgX16, true, false, false, false, true>(a, static_cast<hipStream_t>(stream));
}
