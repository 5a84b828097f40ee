// conv5_wgrad.hip -- filter gradient of the 5x5x5 per-slot convolution (autograd of
// fnet/nn_modules/RepMode.py:207, aten::convolution_backward weight grad), reduced over the
// samples of each slot:
//     dw[slot][tap][co][ci] = sum_{n in slot} sum_v dy[n][v][co] * x[n][v + tap][ci]
//
// GEMM view per tap: M = co, N = ci, K = voxels.  This first version computes in exact f32 on
// v_mfma_f32_16x16x4_f32 for both element types (bf16 inputs are widened while staging): with one
// f32 per lane per operand a tap shift is just an LDS address offset, so no operand transposes are
// needed although activations are channels-last (K-major for this GEMM).
//
// Work decomposition: a workgroup owns (sample, 32 co, 32 ci, one dz plane of taps, a chunk of
// output-voxel tiles); each of its 4 waves owns a 16x16 (co, ci) quadrant for the 25 (dy,dx) taps
// of that plane = 25 accumulator tiles (100 VGPRs).  Per 4x16-voxel tile it stages dy[64][32] and
// the (4+4)x(16+4) input halo plane [160][32] in LDS (rows padded to 48 floats: adjacent voxels fall
// in different bank halves) and issues 25 MFMAs per 4-voxel K step.  Partial sums are added to dw
// with f32 atomics (split-K over voxel chunks and over the samples of a slot).
#include "common.h"
#include "wgrad_col.h"

#include <cmath>
#include <cstdlib>

namespace {
// REPMODE_WGRAD_WS / repmode_set_wgrad_ws: the wave-specialised form of the bf16 filter gradient -- 0 never, 1 where a workgroup
// has a long tile loop, 2 wherever the tile allows, 3 (default, round 4) as 1 + the stream-K form where it is eligible
int g_wgrad_ws = []() { const char* e = getenv("REPMODE_WGRAD_WS"); return e ? atoi(e) : 3; }();

constexpr int TY = 4, TX = 16, TV = TY * TX;   // output voxels per tile
constexpr int HY = TY + 4, HX = TX + 4, HV = HY * HX;
constexpr int RS = 48;                          // LDS row stride in floats
constexpr int WG_LDS_FLOATS = (TV + HV) * RS;

struct WgradArgs {
  const void* x;
  const void* dy;
  const int32_t* sample_slot;
  float* dw;
  int N, D, H, W, Cin, Cout;
  int ncot, ncit, ntiles, tiles_per_block, nchunks, nty, ntx;
  int dz_lo, ndz;       // dz planes computed: dz_lo .. dz_lo + ndz - 1 (all five, or 1..3 for a 3x3x3 support)
  int layout;           // 0: dw[slot][tap][co][ci]; 1: dw[co][ci][125] (expert layout, nslots == 1);
                        // 2: dw[co][ci][27] (expert layout of a centred 3x3x3 filter, nslots == 1)
  int nslots, direct;   // direct: every workgroup owns its output completely -> plain stores, no memset
  int prezeroed;   // dw is known to be all zero already: no memset before the atomics
  int CinTot, ci_off;   // layout 0: dw rows have CinTot input channels and this call fills [ci_off, ci_off + Cin)
                        // (the two halves of a skip connection's filter gradient); CinTot == Cin, ci_off == 0 otherwise
  // Second job of a dual launch (repmode_conv5_wgrad_dual, bf16 kernels): workgroups [grid0, gridDim) compute another
  // output gradient's filter gradient over the same input -- the 3x3x3 expert's beside the 5x5x5 expert's in the
  // per-expert formulation.  grid0 == 0: single job.
  // Stream-K form of the wave-specialised kernel (round 4): > 0 = the launch's total number of tile steps; the grid is a set of
  // persistent workgroups that each take an equal range of the global step sequence (see the kernel).  0: one (unit, chunk) per
  // workgroup as planned by nchunks / tiles_per_block.
  long sk_total;
  int* plan_out;        // host only: not NULL = plan the launch, report whether every element of dw will be written by plain
                        // stores (1) or the output must be all zero on entry (0), and do not launch (repmode_conv5_wgrad_plan)
  int grid0;
  const void* dy2;
  float* dw2;
  int dz_lo2, ndz2, layout2, direct2, nchunks2, tiles_per_block2;
};

template <typename T>
__device__ __forceinline__ void load_row_f32(const T* p, int c, int cmax, bool vec_ok, float* out);

template <>
__device__ __forceinline__ void load_row_f32<float>(const float* p, int c, int cmax, bool vec_ok, float* out) {
  if (vec_ok) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (c + k < cmax) ? p[k] : 0.f;
  }
}

template <>
__device__ __forceinline__ void load_row_f32<bf16_t>(const bf16_t* p, int c, int cmax, bool vec_ok, float* out) {
  if (vec_ok) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    out[0] = __uint_as_float(v.x << 16); out[1] = __uint_as_float(v.x & 0xffff0000u);
    out[2] = __uint_as_float(v.y << 16); out[3] = __uint_as_float(v.y & 0xffff0000u);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = (c + k < cmax) ? bf16_to_f32(p[k]) : 0.f;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv5_wgrad_f32c_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[WG_LDS_FLOATS];
  float* dys = smem;              // [TV][RS]
  float* xs = smem + TV * RS;     // [HV][RS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cq = wave & 1, ciq = wave >> 1;
  const int l15 = lane & 15, kq = lane >> 4;

  // the five dz workgroups of one voxel chunk read the same dy / x tiles: adjacent logical ids, and
  // xcd_remap keeps adjacent ids on one XCD, so they share those tiles through one L2
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int dz = a.dz_lo + bid % a.ndz; bid /= a.ndz;
  const int chunk = bid % a.nchunks; bid /= a.nchunks;
  const int cit = bid % a.ncit;      bid /= a.ncit;
  const int cot = bid % a.ncot;
  const int slot = bid / a.ncot;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;
  const bool vec_x = (Cin & 3) == 0, vec_dy = (Cout & 3) == 0;

  f32x4 acc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = chunk * a.tiles_per_block;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_block);
  for (int n = 0; n < a.N; ++n) {
  if (a.sample_slot[n] != slot) continue;             // the workgroup sums over the samples of its slot
  const T* __restrict__ xn = static_cast<const T*>(a.x) + (size_t)n * D * H * W * Cin;
  const T* __restrict__ dyn = static_cast<const T*>(a.dy) + (size_t)n * D * H * W * Cout;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tx = tile % a.ntx, t2 = tile / a.ntx;
    const int ty = t2 % a.nty, z = t2 / a.nty;
    const int zin = z + dz - 2;
    if (zin < 0 || zin >= D) continue;   // whole input plane is padding: contributes nothing (uniform)
    const int y0 = ty * TY, x0 = tx * TX;
    __syncthreads();
    // stage dy tile: TV voxels x 32 co, items of 4 channels
    for (int it = tid; it < TV * 8; it += 256) {
      const int v = it >> 3, c4 = (it & 7) * 4;
      const int gy = y0 + v / TX, gx = x0 + v % TX, c = cot * 32 + c4;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (gy < H && gx < W && c < Cout)
        load_row_f32<T>(dyn + ((size_t)(z * H + gy) * W + gx) * Cout + c, c, Cout, vec_dy, e);
      *reinterpret_cast<f32x4*>(dys + v * RS + c4) = f32x4{e[0], e[1], e[2], e[3]};
    }
    // stage input halo plane: HV voxels x 32 ci
    for (int it = tid; it < HV * 8; it += 256) {
      const int vh = it >> 3, c4 = (it & 7) * 4;
      const int gy = y0 + vh / HX - 2, gx = x0 + vh % HX - 2, c = cit * 32 + c4;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin)
        load_row_f32<T>(xn + ((size_t)(zin * H + gy) * W + gx) * Cin + c, c, Cin, vec_x, e);
      *reinterpret_cast<f32x4*>(xs + vh * RS + c4) = f32x4{e[0], e[1], e[2], e[3]};
    }
    __syncthreads();
    // K loop: 4 consecutive x voxels per step (lane's k = lane >> 4)
#pragma unroll 2
    for (int ks = 0; ks < TV / 4; ++ks) {
      const int v = ks * 4 + kq;
      const int ly = v / TX, lx = v % TX;
      const float af = dys[v * RS + cq * 16 + l15];                 // A[i = co][k = voxel]
      const float* xb = xs + (ly * HX + lx) * RS + ciq * 16 + l15;  // B[k = voxel][j = ci], tap (0,0)
#pragma unroll
      for (int t = 0; t < 25; ++t) {
        const float bf = xb[((t / 5) * HX + (t % 5)) * RS];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[t], 0, 0, 0);
      }
    }
  }
  }
  // 16x16 C/D layout: column (ci) = lane & 15, row (co) = (lane >> 4) * 4 + r
  const int ci = cit * 32 + ciq * 16 + l15;
  if (ci < Cin) {
#pragma unroll
    for (int t = 0; t < 25; ++t) {
      const int tap = dz * 25 + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * 32 + cq * 16 + kq * 4 + r;
        if (co < Cout) {
          float* p;
          if (a.layout == 0) {
            p = a.dw + (((size_t)slot * REPMODE_TAPS + tap) * Cout + co) * a.CinTot + a.ci_off + ci;
          } else if (a.layout == 1) {
            p = a.dw + ((size_t)co * Cin + ci) * REPMODE_TAPS + tap;
          } else {
            const int ty = t / 5, tx = t % 5;            // (dy, dx) of this tap; dz in [1,3] by construction
            if (ty < 1 || ty > 3 || tx < 1 || tx > 3) continue;
            p = a.dw + ((size_t)co * Cin + ci) * 27 + ((dz - 1) * 3 + (ty - 1)) * 3 + (tx - 1);
          }
          if (a.direct) *p = acc[t][r];
          else unsafeAtomicAdd(p, acc[t][r]);
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// bf16 version on v_mfma_f32_16x16x32_bf16 (16x the f32 rate).
//
// For this GEMM both operands are K-major in HBM (K = voxels, activations are channels-last) while
// the MFMA wants 8 consecutive K values per lane.  Each staged element is reused by 25 taps x 16
// channels of the other operand, so the operands are transposed once while staging:
//   dyT[co][voxel]                       (bf16, channel rows 32 bytes more than a multiple of 256 apart)
//   xT [ci][z][halo row][RG * 8]         (bf16; a halo row starts at x0-2 and takes RG = 8 / 4 / 2 sixteen-byte
//                                         slots for TX = 32 / 16 / 8: a power of two, so a (dy) step is an
//                                         immediate offset; channel rows again 32 bytes off a multiple of 256)
// A lane's B fragment for tap shift s = dx-2 in [-2,2] is the 8-voxel window starting at element 2+s of the 12 elements
// x0+8g-2 .. x0+8g+9 = slot g of the halo row and the first half of slot g+1: one ds_read_b128 + one ds_read_b64 per
// (dy) row give all five windows -- shift -2 is the first load itself, the other even shifts are register moves,
// the odd shifts five v_alignbit_b32 (shared between s = -1 and s = +1).
// LDS banks (MI355X_MICROARCH.md, LDS): ds_read_b128 serves lanes {0-3,12-15,20-27}, {4-11,16-19,28-31} (and the same
// of the upper half) in one cycle each -- 8 channel rows of one voxel group and 8 of the next.  With channel rows an
// odd multiple of 32 bytes apart (32 mod 256 here), row r sits 2 r slots further, so the first 8 take the even and the other 8 the odd slots
// of the 16: conflict-free, for both operands (TX = 8, whose two groups of a pair are two halo rows: the two slots of
// a halo row swapped on odd channel rows in addition).  Round 2, before: rows an odd multiple of 16 bytes apart with
// the halo words read by ds_read_b32 -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.64 (level 0) .. 0.70 (level 2),
// LDS busy 74 % of the level-2 launch, two thirds of all wave stalls waiting on it.
// Staging packs two x-adjacent voxels per ds_write_b32 (8 channels each from one 16-byte load).
template <int TZ, int TY, int TX>
struct WgTile {
  static constexpr int TV = TZ * TY * TX;          // voxels per tile (multiple of 32)
  static constexpr int HY = TY + 4;
  static constexpr int NGX = TX / 8;               // 8-voxel groups per row
  static constexpr int RG = TX >= 32 ? 8 : TX >= 16 ? 4 : 2;   // 16-byte slots per stored halo row (TX + 4 elements fit)
  static constexpr bool SWZ = TX < 16;             // TX = 8: the two slots of a halo row are swapped on odd channel rows
  static constexpr int KSTEPS = TV / 32;
  static constexpr int ROW_C = TZ * HY * RG * 16 + 32;  // bytes per input channel
  static constexpr int DYS = TV * 2 + 32;               // bytes per output channel
  static constexpr int LDS_OPERANDS = 32 * ROW_C + 32 * DYS;
  static constexpr int LDS = LDS_OPERANDS > 4 * 64 * 25 * 4 ? LDS_OPERANDS : 4 * 64 * 25 * 4;   // (the expert-layout epilogue's buffer)
  // The K loop's plan (see the kernel: a K step = 32 voxels = GPR tile rows, a window = one halo row of a plane):
  static constexpr int GPR = 4 / NGX;                // tile rows per K step
  static constexpr int NWROW = TY - GPR + 5;         // distinct halo rows (relative to the lane's) the taps of a plane touch
  static constexpr int NW = TZ * NWROW;              // windows of a tile
  // tap row dy of K step ks in window w, or -1
  static constexpr int dyi_of(int w, int ks) {
    const int zz = w / NWROW, rr = w % NWROW, grb = ks * GPR, d = rr - grb % TY;
    return (grb / TY == zz && d >= 0 && d < 5) ? d : -1;
  }
  static constexpr int mfmas_of(int w) { int c = 0; for (int k = 0; k < KSTEPS; ++k) if (dyi_of(w, k) >= 0) c += 5; return c; }
  static constexpr int steps_before(int w, int ks) { int c = 0; for (int k = 0; k < ks; ++k) if (dyi_of(w, k) >= 0) ++c; return c; }
  static constexpr int last_window_of(int ks) { int l = -1; for (int w = 0; w < NW; ++w) if (dyi_of(w, ks) >= 0) l = w; return l; }
  static_assert(TV % 32 == 0 && TX % 8 == 0 && TX <= 32 && (TX + 4) * 2 <= RG * 16, "tile shape");
  static_assert((ROW_C / 16) % 4 == 2 && (DYS / 16) % 4 == 2,
                "channel rows an odd multiple of 32 bytes apart: 8 rows take the even (odd) 16-byte slots -> conflict-free ds_read_b128");
};

__device__ __forceinline__ u32x4 load8_bf16(const bf16_t* p, int c, int cmax, bool vec_ok) {
  if (vec_ok) return *reinterpret_cast<const u32x4*>(p);
  bf16_t e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = (c + k < cmax) ? p[k] : (bf16_t)0;
  return *reinterpret_cast<const u32x4*>(e);
}

// word k (two bf16) of a 16-byte register -> element index 0..7
__device__ __forceinline__ uint32_t bf16_elem(const u32x4& v, int k) {
  const uint32_t w = v[k >> 1];
  return (k & 1) ? (w >> 16) : (w & 0xffffu);
}

// Experiment (REPMODE_EXTRA_FLAGS=-DRM_WGRAD_PRIO): s_setprio 1 around a tile's MFMAs (the co-resident workgroup is
// transposing its next tile into LDS meanwhile)
#ifdef RM_WGRAD_PRIO
#define RM_WPRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define RM_WPRIO(p) do {} while (0)
#endif

// The next B-operand window is requested behind a scheduling fence: without it the scheduler sinks a window's two LDS
// loads to just before the s_waitcnt of the code that consumes them (-DRM_WGRAD_NOSCHED).
#ifdef RM_WGRAD_NOSCHED
#define RM_WSCHED_FENCE() do {} while (0)
#else
#define RM_WSCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#ifdef RM_CONV_TIMING
// developer build only (REPMODE_EXTRA_FLAGS=-DRM_CONV_TIMING): shader-clock stamps of the first workgroups'
// phases, read back with repmode_debug_wgrad_timing (tools/wgrad_phase_timing.py)
__device__ unsigned long long g_wgrad_timing[64 * 64];
#define RM_WSTAMP(slot)                                                                       \
  do {                                                                                        \
    if (tid == 0 && blockIdx.x < 64 && (slot) < 64)                                           \
      g_wgrad_timing[blockIdx.x * 64 + (slot)] = __builtin_amdgcn_s_memtime();                \
  } while (0)
#else
#define RM_WSTAMP(slot) do {} while (0)
#endif

// VEC (both channel counts multiples of 8, every layer of the network but the thin first/last ones, which
// have their own kernel): the tile sequence of the workgroup is software-pipelined -- the 16-byte loads of the
// NEXT tile are issued (buffer loads, 32-bit offsets, out-of-volume positions return 0 through the range check)
// before the MFMAs of the current one and only transposed into LDS after them.  Without it the phase timing
// (tools/wgrad_phase_timing.py) showed ~7k cycles of serialized load->LDS staging next to ~5k cycles of MFMAs
// per tile.  !VEC keeps the simple stage-then-compute loop with per-element loads.
// WS (with VEC): the waves are SPECIALISED as in conv5_ws_kernel -- eight waves, one MFMA wave and one loader wave per SIMD,
// one workgroup per CU, the operand tiles double-buffered in LDS.  The stamps of the two-workgroup form
// (profiles/r03_pmc_wgrad.txt) show its two workgroups in step: ~2.2 k cycles of fetch-wait + transposition, then ~6.5 k for
// the 200 MFMAs of BOTH workgroups' waves on a SIMD -- 43 % MFMA-busy.  Here the loader waves fetch and transpose tile k + 1
// into the other buffer while the MFMA waves multiply tile k; one barrier per tile.
template <int TZ, int TY, int TX, bool VEC, bool DENSE = false, bool WS = false>
__global__ __launch_bounds__(WS ? 512 : 256, DENSE ? 3 : 2) void conv5_wgrad_bf16_kernel(WgradArgs a) {
  using G = WgTile<TZ, TY, TX>;
  constexpr int TV = G::TV, HY = G::HY, RG = G::RG, NGX = G::NGX, ROW_C = G::ROW_C, DYS = G::DYS;
  static_assert(!WS || (VEC && !DENSE), "wave specialisation: the software-pipelined form only");
  constexpr int LDS_SET = G::LDS_OPERANDS;                   // one buffer: x tile + dy tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[WS ? (2 * LDS_SET > G::LDS ? 2 * LDS_SET : G::LDS) : G::LDS];
  unsigned char* xT = smem;
  unsigned char* dyT = smem + 32 * ROW_C;

  const int tid = WS ? (int)(threadIdx.x & 255) : (int)threadIdx.x;       // index within the role (staging items, stamps)
  const int lane = tid & 63, wave = tid >> 6;
  const bool loader = WS && threadIdx.x >= 256;
#ifdef RM_WS_PRIO
  if (WS && !loader) __builtin_amdgcn_s_setprio(RM_WS_PRIO);      // experiment: the MFMA waves ahead of the loader wave of their SIMD
#endif
  const int cq = wave & 1, ciq = wave >> 1;
  const int l15 = lane & 15, kg = lane >> 4;
  RM_WSTAMP(59);

  // dual launch: the second job's workgroups take its parameters (uniform per workgroup: scalar selects)
  int bid_raw = blockIdx.x, grid_n = gridDim.x;
  if (a.grid0 > 0) {
    if (bid_raw >= a.grid0) {
      bid_raw -= a.grid0; grid_n -= a.grid0;
      a.dy = a.dy2; a.dw = a.dw2; a.dz_lo = a.dz_lo2; a.ndz = a.ndz2; a.layout = a.layout2; a.direct = a.direct2;
      a.nchunks = a.nchunks2; a.tiles_per_block = a.tiles_per_block2;
    } else {
      grid_n = a.grid0;
    }
  }
  // the five dz workgroups of one voxel chunk read the same dy / x tiles: adjacent logical ids, and
  // xcd_remap keeps adjacent ids on one XCD, so they share those tiles through one L2
  // (not const: the stream-K form walks several units per workgroup)
  int bid = xcd_remap(bid_raw, grid_n);
  int dz = a.dz_lo + bid % a.ndz; bid /= a.ndz;
  const int chunk = bid % max(a.nchunks, 1); bid /= max(a.nchunks, 1);
  int cit = bid % a.ncit;      bid /= a.ncit;
  int cot = bid % a.ncot;
  int slot = bid / a.ncot;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;

  f32x4 acc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // which samples belong to this workgroup's slot: one vector load + ballot for the first 64 samples instead of a
  // chain of dependent scalar loads (cold after the kernel boundary: ~2 us each before the first tile is fetched)
  const unsigned long long mine64 =
      (a.nslots == 1 || (WS && a.sk_total > 0)) ? ~0ull : __ballot(lane < a.N && a.sample_slot[min(lane, a.N - 1)] == slot);   // one slot: every sample
  auto in_slot = [&](int n) -> bool {
    return n < 64 ? ((mine64 >> n) & 1ull) != 0 : (a.nslots == 1 || a.sample_slot[n] == slot);
  };

  const int t_begin = chunk * a.tiles_per_block;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_block);
  constexpr int NPAIR = TX / 2 + 2;                // x pairs covering x0-2 .. x0+TX+1
  constexpr int NIT_X = TZ * HY * NPAIR * 4;       // x items: (plane, halo row, x pair, channel group of 8)
  constexpr int NIT_DY = (TV / 2) * 4;             // dy items: (voxel pair, channel group of 8)

  // ---- K loop over the staged tile: 32 voxels per step = 4 groups of 8 consecutive x; this lane's group = 4*ks + kg
  // A K step takes 32 voxels = 4 groups of 8 consecutive x, one per 16-lane quarter kg: group g = 4 ks + kg is x-group
  // g % NGX of row g / NGX of the tile.  With GPR = 4 / NGX rows per step, this lane's row is ks * GPR + kg / NGX:
  // a step-dependent part (compile time) plus a lane-dependent one that goes into the lane's base address.
  constexpr int GPR = 4 / NGX;                       // tile rows per K step
  constexpr int NWROW = TY - GPR + 5;                // distinct halo rows (relative to the lane's) the taps of a plane touch
  const unsigned char* xlane = xT + (ciq * 16 + l15) * ROW_C + ((kg / NGX) * RG + kg % NGX) * 16;
  const unsigned char* alane = dyT + (cq * 16 + l15) * DYS + kg * 16;
  // ---- Round 6: the tile's MFMAs as a pipeline of windows.  v_mfma_f32_16x16x32_bf16 issues back to back every ~17 cycles on
  // a SIMD (tools/mfma_rate.hip: 16.9; 17.2 / 18.0 with one / two vector operations behind every MFMA, 22.3 with three), a block
  // of other instructions between two MFMAs is idle matrix pipe.  Round 3's form prepared a window's five operands as a block
  // of 13 vector operations (5 v_alignbit + 8 moves: the shift-0 / shift+1 operands start at odd registers of the window's six
  // words, and a 128-bit MFMA operand must start at an even one) between the MFMAs of two windows (stamps: 3980 cycles per 200
  // MFMAs).  Here the 16 operations of window w + 1 (each shifted operand into registers of its own: 2 x 4 v_alignbit, 2 x 4
  // moves) are placed one by one behind the MFMAs of window w, and windows are requested two ahead.
  // PIPE (the stream-K loop, NW a multiple of 6 so that the register rings line up): the pipeline runs ACROSS tiles.  A tile's
  // last LDS read is the request of its last window, two windows before its end: there the MFMA waves wait for their LDS data,
  // pass the barrier (the next tile is staged; this tile's buffer is free) and fetch the next tile's A fragments and first
  // windows behind the remaining MFMAs -- the ~350 cycles a tile spent on its first LDS round trip and the first window's
  // preparation (9 % of a 3980-cycle tile) overlap with MFMAs.
  struct Win { u32x4 lo; u32x2 hi; };               // words 0..3 / 4, 5 of a window's six
  struct Ops { u32x4 b1, b2, b3, b4; };             // shifts -1, 0, +1, +2 (shift -2 is `lo` itself)
  constexpr int NW = G::NW;
  static_assert(G::GPR == GPR && G::NWROW == NWROW, "WgTile's plan");
  Win p_raw[3];
  Ops p_ops[2];
  bf16x8 p_afr[G::KSTEPS];
  auto w_request = [&](int off, int w, Win& o) __attribute__((always_inline)) {
    const int zz = w / NWROW, rr = w % NWROW;
    const unsigned char* xb = xlane + off + ((zz * HY + rr) * RG) * 16;
    o.lo = *reinterpret_cast<const u32x4*>(xb);
    o.hi = *reinterpret_cast<const u32x2*>(xb + 16);
  };
  auto w_prep = [&](const Win& r, int i, Ops& o) __attribute__((always_inline)) {      // operation i of 16
    const uint32_t w[6] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y};
    const int e = i & 3;
    if (i < 4) o.b1[e] = __builtin_amdgcn_alignbit(w[e + 1], w[e], 16);            // elements 1..8
    else if (i < 8) o.b3[e] = __builtin_amdgcn_alignbit(w[e + 2], w[e + 1], 16);   // elements 3..10
    else {
      // elements 2..9 / 4..11: moves, as instructions of their own (a plain element copy has no place in the instruction
      // stream: the compiler emitted them in blocks of 3-6 in front of the MFMA that reads the operand)
      uint32_t t;
      asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(i < 12 ? w[e + 1] : w[e + 2]));
      if (i < 12) o.b2[e] = t; else o.b4[e] = t;
    }
  };
  auto a_load = [&](int off, int ks) __attribute__((always_inline)) { p_afr[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(alane + off + ks * 64)); };
  auto pipe_prologue = [&](int off) __attribute__((always_inline)) {
    static_for<0, G::KSTEPS>([&](auto KS) { a_load(off, KS.value); });
    w_request(off, 0, p_raw[0]);
    if constexpr (NW > 1) w_request(off, 1, p_raw[1]);
    static_for<0, 16>([&](auto I) { w_prep(p_raw[0], I.value, p_ops[0]); });
  };
  auto pipe_body = [&](int off, int noff, auto PIPE_) __attribute__((always_inline)) {
    constexpr bool PIPE = decltype(PIPE_)::value;
    static_for<0, NW>([&](auto WI) {
      constexpr int w = WI.value;
      Win& cur = p_raw[w % 3];
      Ops& op = p_ops[w & 1];
      if constexpr (PIPE && w == 0)                    // the A fragments still in use when the previous tile fetched the others
        static_for<0, G::KSTEPS>([&](auto KS) { if constexpr (G::last_window_of(KS.value) >= NW - 2) a_load(off, KS.value); });
      if constexpr (w + 2 < NW) w_request(off, w + 2, p_raw[(w + 2) % 3]);
      if constexpr (PIPE && w == NW - 2) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // this tile is in registers; the next one is staged
        static_for<0, G::KSTEPS>([&](auto KS) { if constexpr (G::last_window_of(KS.value) < NW - 2) a_load(noff, KS.value); });
        w_request(noff, 0, p_raw[NW % 3]);
      }
      if constexpr (PIPE && w == NW - 1) w_request(noff, 1, p_raw[(NW + 1) % 3]);
      RM_WSCHED_FENCE();
      constexpr int nm = G::mfmas_of(w);
      constexpr int per = nm > 0 ? (16 + nm - 1) / nm : 16;
      constexpr bool has_next = w + 1 < NW || PIPE;      // (PIPE: the next tile's window 0 takes ring slots NW % 3 = 0, NW & 1 = 0)
      if constexpr (nm == 0 && has_next)                 // (no tile shape has such a window; kept correct)
        static_for<0, 16>([&](auto I) { w_prep(p_raw[(w + 1) % 3], I.value, p_ops[(w + 1) & 1]); });
      const bf16x8 o0 = __builtin_bit_cast(bf16x8, cur.lo), o1 = __builtin_bit_cast(bf16x8, op.b1),
                   o2 = __builtin_bit_cast(bf16x8, op.b2), o3 = __builtin_bit_cast(bf16x8, op.b3),
                   o4 = __builtin_bit_cast(bf16x8, op.b4);
      static_for<0, G::KSTEPS>([&](auto KS) {
        constexpr int ks = KS.value, dyi = G::dyi_of(w, ks);
        if constexpr (dyi >= 0) {
          constexpr int before = G::steps_before(w, ks);
          static_for<0, 5>([&](auto J) {
            constexpr int j = J.value, m = before * 5 + j;
            const bf16x8 bj = j == 0 ? o0 : j == 1 ? o1 : j == 2 ? o2 : j == 3 ? o3 : o4;
            acc[dyi * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p_afr[ks], bj, acc[dyi * 5 + j], 0, 0, 0);
            if constexpr (has_next && m * per < 16) {
              static_for<m * per, (m + 1) * per < 16 ? (m + 1) * per : 16>([&](auto I) { w_prep(p_raw[(w + 1) % 3], I.value, p_ops[(w + 1) & 1]); });
              RM_WSCHED_FENCE();
            }
          });
        }
      });
    });
  };
  constexpr bool SK_PIPE = WS && TZ == 1 && !G::SWZ && !DENSE && NW % 6 == 0;      // (the stream-K loop's cross-tile form)
  auto mma_tile = [&](int boff) __attribute__((always_inline)) {      // (called from two loops: without the attribute one instantiation kept it as a FUNCTION)
    if constexpr (!G::SWZ && !DENSE) {        // (the TX = 8 tiles: 5.5 MFMAs per window, 16 operations per window measured slower than 13: 2700 -> 3050 cycles per tile)
      pipe_prologue(boff);
      pipe_body(boff, 0, std::false_type{});
      return;
    }
    // The B operands of step (ks, dyi) are the window of halo row yyb(ks) + dyi of plane zz(ks): steps with equal sums share
    // it (TX = 32: 40 steps, 12 windows).  So the loop runs over WINDOWS -- one ds_read_b128 + one ds_read_b64 and five
    // v_perm / alignbit each, requested one window ahead -- and every window feeds the MFMAs of all steps that use it.
    bf16x8 afr[G::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks)
      afr[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(alane + boff + ks * 64));
    auto window = [&](int w, u32x4& lo, u32x2& hi) {
      const int zz = w / NWROW, rr = w % NWROW;
      const unsigned char* xb = xlane + boff + ((zz * HY + rr) * RG) * 16;
      lo = *reinterpret_cast<const u32x4*>(G::SWZ ? xb + (l15 & 1) * 16 : xb);                 // words 0..3: elements 0..7
      hi = *reinterpret_cast<const u32x2*>(G::SWZ ? xb + 16 - (l15 & 1) * 16 : xb + 16);      // words 4, 5: elements 8..11
    };
    u32x4 lo_n;
    u32x2 hi_n;
    if (!DENSE) window(0, lo_n, hi_n);
#pragma unroll
    for (int w = 0; w < TZ * NWROW; ++w) {
      const int zz = w / NWROW, rr = w % NWROW;
      u32x4 lo = lo_n;
      u32x2 hi = hi_n;
      if (DENSE) window(w, lo, hi);              // (three workgroups per CU: no register room to run a window ahead)
      // From here on the window is a register value (opaque to the optimiser, which otherwise re-reads parts of it from
      // LDS to assemble the shifted operands)
      asm volatile("" : "+v"(lo), "+v"(hi));
      if (!DENSE && w + 1 < TZ * NWROW) window(w + 1, lo_n, hi_n);
      RM_WSCHED_FENCE();      // keep the request ahead of this window's MFMAs (the scheduler sinks it to its use otherwise)
      const uint32_t a10 = __builtin_amdgcn_alignbit(lo.y, lo.x, 16);
      const uint32_t a21 = __builtin_amdgcn_alignbit(lo.z, lo.y, 16);
      const uint32_t a32 = __builtin_amdgcn_alignbit(lo.w, lo.z, 16);
      const uint32_t a43 = __builtin_amdgcn_alignbit(hi.x, lo.w, 16);
      const uint32_t a54 = __builtin_amdgcn_alignbit(hi.y, hi.x, 16);
      const bf16x8 b0 = __builtin_bit_cast(bf16x8, lo);                                  // shift -2: elements 0..7
      const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{a10, a21, a32, a43}));          // shift -1: elements 1..8
      const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{lo.y, lo.z, lo.w, hi.x}));      // shift  0: elements 2..9
      const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{a21, a32, a43, a54}));          // shift +1
      const bf16x8 b4 = __builtin_bit_cast(bf16x8, (u32x4{lo.z, lo.w, hi.x, hi.y}));      // shift +2: elements 4..11
#pragma unroll
      for (int ks = 0; ks < G::KSTEPS; ++ks) {
        const int grb = ks * GPR;                  // first tile row of the step
        const int dyi = rr - grb % TY;
        if (grb / TY == zz && dyi >= 0 && dyi < 5) {
          acc[dyi * 5 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], b0, acc[dyi * 5 + 0], 0, 0, 0);
          acc[dyi * 5 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], b1, acc[dyi * 5 + 1], 0, 0, 0);
          acc[dyi * 5 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], b2, acc[dyi * 5 + 2], 0, 0, 0);
          acc[dyi * 5 + 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], b3, acc[dyi * 5 + 3], 0, 0, 0);
          acc[dyi * 5 + 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[ks], b4, acc[dyi * 5 + 4], 0, 0, 0);
        }
      }
    }
  };
  // byte offset, inside a channel row of xT, of x pair p (x0 - 2 + 2p, + 1) of halo row yr; `odd`: odd channel row
  auto x_pair_off = [&](int yr, int p, int odd) -> int {
    const int slot = G::SWZ ? ((p >> 2) ^ odd) : (p >> 2);
    return (yr * RG + slot) * 16 + (p & 3) * 4;
  };
  // two x-adjacent voxels (8 channels each) -> eight 4-byte stores into the transposed tile
  auto put_pair = [&](unsigned char* dst, int stride, const u32x4& v0, const u32x4& v1) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *reinterpret_cast<uint32_t*>(dst + k * stride) = bf16_elem(v0, k) | (bf16_elem(v1, k) << 16);
  };
  // the same into xT: channel rows cg*8 + k, halo row yr, x pair p
  auto put_x_pair = [&](int cg, int yr, int p, const u32x4& v0, const u32x4& v1, int boff) {
    unsigned char* base = xT + boff + (cg * 8) * ROW_C;
    const int off0 = x_pair_off(yr, p, 0), off1 = x_pair_off(yr, p, 1);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *reinterpret_cast<uint32_t*>(base + k * ROW_C + ((k & 1) ? off1 : off0)) = bf16_elem(v0, k) | (bf16_elem(v1, k) << 16);
  };

  if constexpr (VEC) {
    constexpr int NX = (NIT_X + 255) / 256, NDY = (NIT_DY + 255) / 256;
    constexpr uint32_t OOB = 0x80000000u;
    // a tile's fetched operands on their way to LDS (the stream-K loader waves keep TWO tiles in flight: sets a and b)
    struct TileRegs { u32x4 x0[NX], x1[NX], d0[NDY], d1[NDY]; };
    TileRegs ra, rb;

    // work sequence: (sample of this slot, tile of this chunk), skipping tiles whose input planes for this dz are
    // all padding.  Everything here is wave-uniform (scalar loads of sample_slot).
    // The tile's coordinates (tile plane / row / column: atz, aty, atx) move along with `tile` -- the divisions once per
    // workgroup (round 6: with them in every call, the MFMA waves of the per-expert levels spent 800 cycles between two tiles
    // of 2700).
    int n = -1, tile = t_end;
    int atz = 0, aty = 0, atx = 0;
    const int bz0 = t_begin / max(a.ntx * a.nty, 1), by0 = (t_begin - bz0 * a.ntx * a.nty) / max(a.ntx, 1),
              bx0 = t_begin - (bz0 * a.nty + by0) * a.ntx;
    auto advance = [&]() -> bool {
      for (;;) {
        if (++tile >= t_end) {
          tile = t_begin; atz = bz0; aty = by0; atx = bx0;
          do { ++n; } while (n < a.N && !in_slot(n));
          if (n >= a.N) return false;
        } else if (++atx == a.ntx) {
          atx = 0;
          if (++aty == a.nty) { aty = 0; ++atz; }
        }
        const int z0 = atz * TZ;
        if (!(z0 + TZ - 1 + dz - 2 < 0 || z0 + dz - 2 >= D)) return true;
      }
    };
    // `real` false: the same twelve loads with every offset out of range (zeros, no memory traffic) -- the stream-K loader
    // issues a fetch on EVERY path, because the compiler's s_waitcnt counts must hold on all of them: with the fetch of the
    // tile after next under an `if`, it waited for vmcnt(0) before a transposition, i.e. for the loads issued a moment earlier
    // (round 6: one tile's load latency exposed per step, 213 -> 17x us on level 0)
    auto fetch_to = [&](TileRegs& tr, bool real = true) {
#ifdef RM_WG_NOLOAD
      {                                     // TIMING BUILD ONLY: the tile loop without its global loads (register contents as they are)
#pragma unroll
        for (int u = 0; u < NX; ++u) asm volatile("" : "=v"(tr.x0[u]), "=v"(tr.x1[u]));
#pragma unroll
        for (int u = 0; u < NDY; ++u) asm volatile("" : "=v"(tr.d0[u]), "=v"(tr.d1[u]));
        return;
      }
#endif
      const int z0 = atz * TZ, y0 = aty * TY, x0 = atx * TX;            // (the tile `advance` stands on)
      const uint32_t xbytes = (uint32_t)((size_t)D * H * W * Cin * 2), dybytes = (uint32_t)((size_t)D * H * W * Cout * 2);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(static_cast<const bf16_t*>(a.x)) + (size_t)n * D * H * W * Cin, 0, (int)xbytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<bf16_t*>(static_cast<const bf16_t*>(a.dy)) + (size_t)n * D * H * W * Cout, 0, (int)dybytes, 0x00020000);
      int tid_ = tid;                       // keep the item arithmetic from being hoisted out of the loops
      asm volatile("" : "+v"(tid_));        // (it would pin registers next to the 100 accumulators)
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int it = u * 256 + tid_;
        const int p = it % NPAIR; int r = it / NPAIR;
        const int cg = r & 3; r >>= 2;
        const int hy = r % HY, zz = r / HY;
        const int zin = z0 + zz + dz - 2, gy = y0 + hy - 2, gx = x0 - 2 + 2 * p;
        const int c = cit * 32 + cg * 8;
        const bool row_ok = real && it < NIT_X && (unsigned)zin < (unsigned)D && (unsigned)gy < (unsigned)H && c < Cin;
        const uint32_t off = (uint32_t)((((zin * H + gy) * W + gx) * Cin + c) * 2);
        tr.x0[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                     rx, (row_ok && (unsigned)gx < (unsigned)W) ? off : OOB, 0, 0));
        tr.x1[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                     rx, (row_ok && (unsigned)(gx + 1) < (unsigned)W) ? off + (uint32_t)Cin * 2 : OOB, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < NDY; ++u) {
        const int it = u * 256 + tid_;
        const int q = it % (TV / 2), cg = it / (TV / 2);
        const int m = 2 * q;
        const int xx = m % TX, yy = (m / TX) % TY, zz = m / (TX * TY);
        const int gz = z0 + zz, gy = y0 + yy, gx = x0 + xx;
        const int c = cot * 32 + cg * 8;
        const bool row_ok = real && it < NIT_DY && gz < D && gy < H && c < Cout;
        const uint32_t off = (uint32_t)((((gz * H + gy) * W + gx) * Cout + c) * 2);
        tr.d0[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, (row_ok && gx < W) ? off : OOB, 0, 0));
        tr.d1[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                     rdy, (row_ok && gx + 1 < W) ? off + (uint32_t)Cout * 2 : OOB, 0, 0));
      }
    };
    auto stage_from = [&](TileRegs& tr, int boff) {
#ifdef RM_WG_NOSTAGE
      {                                     // TIMING BUILD ONLY: the loads are waited for, nothing is transposed into LDS
#pragma unroll
        for (int u = 0; u < NX; ++u) asm volatile("" :: "v"(tr.x0[u]), "v"(tr.x1[u]));
#pragma unroll
        for (int u = 0; u < NDY; ++u) asm volatile("" :: "v"(tr.d0[u]), "v"(tr.d1[u]));
        return;
      }
#endif
      int tid_ = tid;
      asm volatile("" : "+v"(tid_));
#pragma unroll
      for (int u = 0; u < NX; ++u) {
        const int it = u * 256 + tid_;
        const int p = it % NPAIR; int r = it / NPAIR;
        const int cg = r & 3; r >>= 2;
        const int hy = r % HY, zz = r / HY;
        if (it < NIT_X) put_x_pair(cg, zz * HY + hy, p, tr.x0[u], tr.x1[u], boff);
      }
#pragma unroll
      for (int u = 0; u < NDY; ++u) {
        const int it = u * 256 + tid_;
        const int q = it % (TV / 2), cg = it / (TV / 2);
        if (it < NIT_DY) put_pair(dyT + boff + (cg * 8) * DYS + q * 4, DYS, tr.d0[u], tr.d1[u]);
      }
    };
    auto fetch = [&]() { fetch_to(ra); };
    auto stage = [&](int boff) { stage_from(ra, boff); };
    bool have = (WS && a.sk_total > 0) ? false : advance();      // (stream-K walks its own sequence)
#ifdef RM_CONV_TIMING
    int tl_ = 0;
#endif
    if constexpr (WS) {
      // ---- The loader waves' tile fetch + transposition (round 6).  A wave issues one instruction per four cycles at best,
      // and there is ONE loader wave per SIMD: round 5's loader (fetch_to / stage_from above) ran ~950 instructions per tile
      // (430 vector: the decomposition of every item index into pair / channel group / halo row, range checks, 3 operations
      // per transposed dword; 430 scalar: tile index divisions, spilled-SGPR traffic) = 3.8 k cycles of issue alone beside the
      // 3.2 k (stamped: 4.0 k) cycles of a tile's MFMAs, and the launches ran at the loaders' pace (level 0 stream-K: 213 us
      // against 162 with the loads removed, 172 with the transposition removed, 174 with both -- profiles/r06_wgrad_loader.txt).
      // Here everything that depends on the thread only is computed ONCE (the loader waves have the MFMA waves' register
      // budget and no accumulators): per item its offset relative to the tile's origin, its halo row / x pair / plane and its
      // LDS address; a tile costs an add per coordinate, range checks as sign bits, one v_perm_b32 per transposed dword.
      constexpr int BIG = 0x40000000;
      uint32_t xrel[NX], drel[NDY];
      int xhy[NX], xx2[NX], xzz[NX], xlds0[NX], xlds1[NX], dyy[NDY], dxx[NDY], dzz[NDY], dlds[NDY];
      int cit_cur = -1, cot_cur = -1;
      auto lean_init = [&]() {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
          const int it = u * 256 + tid;
          const int pp = it % NPAIR; int r = it / NPAIR;
          const int cg = r & 3; r >>= 2;
          const int hy = r % HY, zz = r / HY;
          xrel[u] = (uint32_t)((((zz * H + hy) * W + 2 * pp) * Cin + cg * 8) * 2);
          xx2[u] = 2 * pp; xzz[u] = zz;
          xlds0[u] = (cg * 8) * ROW_C + x_pair_off(zz * HY + hy, pp, 0);
          xlds1[u] = (cg * 8) * ROW_C + x_pair_off(zz * HY + hy, pp, 1);
        }
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
          const int it = u * 256 + tid;
          const int qq = it % (TV / 2), cg = it / (TV / 2), m = 2 * qq;
          const int xx = m % TX, yy = (m / TX) % TY, zz = m / (TX * TY);
          drel[u] = (uint32_t)((((zz * H + yy) * W + xx) * Cout + cg * 8) * 2);
          dxx[u] = xx; dzz[u] = zz;
          dlds[u] = 32 * ROW_C + (cg * 8) * DYS + qq * 4;
        }
      };
      // per unit: items whose channel group lies behind the tensors' channels (and the items behind the tile's last) get a
      // halo row that fails every tile's range check
      auto lean_refresh = [&]() {
        cit_cur = cit; cot_cur = cot;
        const int crx = Cin - cit * 32, crd = Cout - cot * 32;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
          int r = (u * 256 + tid) / NPAIR;
          const int cg = r & 3; r >>= 2;
          xhy[u] = (r / HY < TZ && cg * 8 < crx) ? r % HY : BIG;
        }
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
          const int it = u * 256 + tid;
          const int cg = it / (TV / 2), m = 2 * (it % (TV / 2));
          dyy[u] = (cg < 4 && cg * 8 < crd) ? (m / TX) % TY : BIG;
        }
      };
      // the tile at output voxel (z0, y0, x0) of sample n_, for this workgroup's dz / cit / cot
      auto fetch_lean = [&](TileRegs& tr, bool real, int n_, int z0, int y0, int x0) __attribute__((always_inline)) {
#ifdef RM_WG_NOLOAD
        {                                 // TIMING BUILD ONLY: the tile loop without its global loads
#pragma unroll
          for (int u = 0; u < NX; ++u) asm volatile("" : "=v"(tr.x0[u]), "=v"(tr.x1[u]));
#pragma unroll
          for (int u = 0; u < NDY; ++u) asm volatile("" : "=v"(tr.d0[u]), "=v"(tr.d1[u]));
          return;
        }
#endif
        if (cit != cit_cur || cot != cot_cur) lean_refresh();
        const int zin = z0 + dz - 2;
        const uint32_t xbytes = (uint32_t)((size_t)D * H * W * Cin * 2), dybytes = (uint32_t)((size_t)D * H * W * Cout * 2);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(static_cast<const bf16_t*>(a.x)) + (size_t)n_ * D * H * W * Cin, 0, (int)xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_t*>(static_cast<const bf16_t*>(a.dy)) + (size_t)n_ * D * H * W * Cout, 0, (int)dybytes, 0x00020000);
        const uint32_t xb = (uint32_t)((((zin * H + y0 - 2) * W + x0 - 2) * Cin + cit * 32) * 2);
        const uint32_t db = (uint32_t)((((z0 * H + y0) * W + x0) * Cout + cot * 32) * 2);
        // Range checks as sign bits (a coordinate t is inside [0, n) iff neither t nor n - 1 - t is negative), OR-ed into bit 31
        // of the offset: out of the buffer's range, the load returns zeros.  (Written with compare + select, the compiler
        // turned the shared row check into branches around the loads -- and waited for vmcnt(0) inside them.)
        const int Hm1 = real ? H - 1 : -1;                  // (a fetch behind the last step: every row fails)
        const uint32_t cin2 = (uint32_t)Cin * 2, cout2 = (uint32_t)Cout * 2;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
          const int gy = xhy[u] + (y0 - 2), gx = xx2[u] + (x0 - 2), wx = (W - 1) - gx;
          int my = gy | (Hm1 - gy);
          if constexpr (TZ > 1) { const int gz = xzz[u] + zin; my |= gz | ((D - 1) - gz); }       // (TZ = 1: the step exists only where the plane does)
          const uint32_t off = xrel[u] + xb;
          const uint32_t o0 = ((uint32_t)(my | gx | wx) & OOB) | off;
          const uint32_t o1 = ((uint32_t)(my | (gx + 1) | (wx - 1)) & OOB) | (off + cin2);
          tr.x0[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o0, 0, 0));
          tr.x1[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, o1, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
          const int gy = dyy[u] + y0, wx = (W - 1) - (dxx[u] + x0);      // (gy, gx, gz never negative)
          int my = Hm1 - gy;
          if constexpr (TZ > 1) my |= (D - 1) - (dzz[u] + z0);
          const uint32_t off = drel[u] + db;
          const uint32_t o0 = ((uint32_t)(my | wx) & OOB) | off;
          const uint32_t o1 = ((uint32_t)(my | (wx - 1)) & OOB) | (off + cout2);
          tr.d0[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, o0, 0, 0));
          tr.d1[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, o1, 0, 0));
        }
      };
      // two x-adjacent voxels' eight channels -> eight dwords (channel k: voxel 0 in the low half), one v_perm_b32 each;
      // even channel rows at dst0, odd ones at dst1 (the swizzled TX = 8 tile; the same address otherwise)
      auto put8 = [&](unsigned char* dst0, unsigned char* dst1, int stride, const u32x4& v0, const u32x4& v1) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
          *reinterpret_cast<uint32_t*>(((kk & 1) ? dst1 : dst0) + kk * stride) =
              __builtin_amdgcn_perm(v1[kk >> 1], v0[kk >> 1], (kk & 1) ? 0x07060302u : 0x05040100u);
      };
      auto stage_lean = [&](TileRegs& tr, int buf) __attribute__((always_inline)) {
#ifdef RM_WG_NOSTAGE
        {                                 // TIMING BUILD ONLY: the loads are waited for, nothing is transposed into LDS
#pragma unroll
          for (int u = 0; u < NX; ++u) asm volatile("" :: "v"(tr.x0[u]), "v"(tr.x1[u]));
#pragma unroll
          for (int u = 0; u < NDY; ++u) asm volatile("" :: "v"(tr.d0[u]), "v"(tr.d1[u]));
          return;
        }
#endif
#pragma unroll
        for (int u = 0; u < NX; ++u)
          if ((u + 1) * 256 <= NIT_X || u * 256 + tid < NIT_X)
            put8(smem + buf + xlds0[u], smem + buf + (G::SWZ ? xlds1[u] : xlds0[u]), ROW_C, tr.x0[u], tr.x1[u]);
#pragma unroll
        for (int u = 0; u < NDY; ++u)
          if ((u + 1) * 256 <= NIT_DY || u * 256 + tid < NIT_DY) put8(smem + buf + dlds[u], smem + buf + dlds[u], DYS, tr.d0[u], tr.d1[u]);
      };
      if (loader) lean_init();
      if constexpr (TZ == 1) {
      if (a.sk_total > 0) {
        // ---- stream-K (round 4): a launch is ONE sequence of tile steps -- units (slot, co tile, ci tile, dz plane) in the
        // order of the regular grid, inside a unit the samples of its slot, inside a sample the tiles whose input plane
        // exists for this dz -- and every workgroup (at most one per CU, all resident) takes an equal range of it.  A range
        // crosses unit boundaries: the MFMA waves flush the 25 accumulator tiles at the end of a unit (plain stores when the
        // whole unit ran here, float atomics onto the cleared dw when it is shared with a neighbour) and go on with the next
        // tile, which the loader waves have staged meanwhile -- the 6 k cycles of prologue and 25-30 k cycles of atomics
        // epilogue that every (unit, chunk) workgroup of the regular grid pays (a third of a level-1 workgroup's
        // time) are paid once per workgroup / overlapped, and no round of the grid is partly empty.
        const int tpp = a.nty * a.ntx;                       // tiles of one z plane
        const int G2 = a.ncot * a.ncit;
        auto zlo = [&](int dzz) -> int { return max(0, 2 - dzz); };            // planes z with 0 <= z + dz - 2 < D
        auto zhi = [&](int dzz) -> int { return max(zlo(dzz), min(D, D + 2 - dzz)); };
        int VT = 0;
        for (int i = 0; i < a.ndz; ++i) VT += (zhi(a.dz_lo + i) - zlo(a.dz_lo + i)) * tpp;
        const int slot64 = a.sample_slot[min(lane, a.N - 1)];                 // (the host sends at most 64 samples here)
        auto mask_of = [&](int sl) -> unsigned long long { return __ballot(lane < a.N && slot64 == sl); };
        auto kth = [&](unsigned long long m, int kk) -> int {
          for (int i = 0; i < kk; ++i) m &= m - 1;
          return __ffsll((long long)m) - 1;
        };
        const long wg = xcd_remap(blockIdx.x, gridDim.x);    // neighbouring ranges (same operands' tiles) on one XCD
        const long g0 = a.sk_total * wg / gridDim.x, g1 = a.sk_total * (wg + 1) / gridDim.x;
        const int steps = (int)(g1 - g0);
        if (steps <= 0) return;
        // decode the first step
        unsigned long long mask;
        int cnt, dzi = 0, t_lo = 0, t_hi = 0, k = 0;
        long rem = g0;
        slot = 0;
        for (;;) {
          mask = mask_of(slot); cnt = __popcll(mask);
          const long span = (long)cnt * VT * G2;
          if (rem < span) break;
          rem -= span; ++slot;
        }
        {
          const long per = (long)cnt * VT;
          const int grp = (int)(rem / per);
          rem -= (long)grp * per;
          cot = grp / a.ncit; cit = grp % a.ncit;
        }
        for (;;) {
          dz = a.dz_lo + dzi; t_lo = zlo(dz) * tpp; t_hi = zhi(dz) * tpp;
          const long span = (long)cnt * (t_hi - t_lo);
          if (rem < span) break;
          rem -= span; ++dzi;
        }
        k = (int)(rem / (t_hi - t_lo));
        tile = t_lo + (int)(rem % (t_hi - t_lo));
        n = kth(mask, k);
        // the tile's coordinates (output plane, tile row, tile column) move along with `tile`: divisions only where it jumps
        int zt = 0, tyi = 0, txi = 0;
        auto decomp = [&]() {
          zt = tile / tpp;
          const int r = tile - zt * tpp;
          tyi = r / a.ntx; txi = r - tyi * a.ntx;
        };
        auto step_coords = [&]() {
          if (++txi == a.ntx) { txi = 0; if (++tyi == a.nty) { tyi = 0; ++zt; } }
        };
        decomp();
        bool head = k == 0 && tile == t_lo;                 // the current unit started in this workgroup, at its first step
        auto last_of_unit = [&]() -> bool { return tile + 1 >= t_hi && k + 1 >= cnt; };
        auto next = [&]() {                                 // the following step of the sequence (never called behind the last)
          if (++tile < t_hi) { step_coords(); return; }
          txi = 0; tyi = 0;
          if (++k < cnt) { n = kth(mask, k); tile = t_lo; zt = zlo(dz); return; }
          k = 0;
          do {
            if (++dzi == a.ndz) {
              dzi = 0;
              if (++cit == a.ncit) {
                cit = 0;
                if (++cot == a.ncot) { cot = 0; ++slot; mask = mask_of(slot); cnt = __popcll(mask); }
              }
            }
            dz = a.dz_lo + dzi; t_lo = zlo(dz) * tpp; t_hi = zhi(dz) * tpp;
          } while (t_hi <= t_lo || cnt == 0);            // (a dz without input planes, a slot without samples: no steps)
          tile = t_lo; zt = zlo(dz);
          n = kth(mask, 0);
        };
        int boff = 0;
        if (loader) {
          // two tiles in flight (register sets a / b): a tile step is 1.6 k (16-voxel tile) .. 3.2 k MFMA cycles, a fetch from
          // HBM / Infinity Cache 2-3 k -- with one tile ahead the level-2 launches ran at the loaders' pace (round 4: 90 us
          // stream-K against 83 us regular on 128 -> 128)
          static_assert(!G::SWZ, "stream-K: tiles at least 16 voxels wide");
          fetch_lean(ra, true, n, zt, tyi * TY, txi * TX);
          { const bool more = steps > 1; if (more) next(); fetch_lean(rb, more, n, zt, tyi * TY, txi * TX); }
          // (timing build: the loader's stamps go to slots 30 .. 57 -- per tile: before the transposition, behind it, behind the
          // next fetch's issue, behind the barrier)
#ifdef RM_CONV_TIMING
#define RM_LSTAMP(k) do { if (i < 7) RM_WSTAMP(30 + i * 4 + (k)); } while (0)
#else
#define RM_LSTAMP(k) do {} while (0)
#endif
          // Pairs of steps -- register set a always lands in buffer 0, b in buffer 1 -- with every load on every path (a fetch
          // behind the last step asks for out-of-range offsets): the compiler's s_waitcnt counts must hold on all paths, and
          // with a fetch under an `if` it waited for vmcnt(0) before each transposition.  An odd last step behind the loop.
          int i = 0;
          for (; i + 1 < steps; i += 2) {
            RM_LSTAMP(0);
            stage_lean(ra, 0);
            RM_LSTAMP(1);
            { const bool more = i + 2 < steps; if (more) next(); fetch_lean(ra, more, n, zt, tyi * TY, txi * TX); }
            RM_LSTAMP(2);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            RM_LSTAMP(3);
            stage_lean(rb, LDS_SET);
            { const bool more = i + 3 < steps; if (more) next(); fetch_lean(rb, more, n, zt, tyi * TY, txi * TX); }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          }
          if (i < steps) {
            stage_lean(ra, 0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          }
          if constexpr (SK_PIPE) asm volatile("s_barrier" ::: "memory");      // (the last tile's barrier "for the next tile")
#undef RM_LSTAMP
          return;
        }
        if constexpr (SK_PIPE) {
          asm volatile("s_barrier" ::: "memory");
          pipe_prologue(boff);
        }
        for (int i = 0; i < steps; ++i) {
          RM_WSTAMP(i * 3 + 0);
          if constexpr (!SK_PIPE) asm volatile("s_barrier" ::: "memory");
          RM_WSTAMP(i * 3 + 1);
          if constexpr (SK_PIPE) pipe_body(boff, boff ^ LDS_SET, std::true_type{});     // (with the barrier for the next tile inside)
          else mma_tile(boff);
          RM_WSTAMP(i * 3 + 2);
          boff ^= LDS_SET;
          const bool last = last_of_unit();
          if (last || i + 1 == steps) {
            // 16x16 C/D layout: column (ci) = lane & 15, row (co) = (lane >> 4) * 4 + r
            const bool atomic = !(head && last);
            const int ci = cit * 32 + ciq * 16 + l15;
            if (ci < Cin) {
#pragma unroll
              for (int t = 0; t < 25; ++t) {
                const int tap = dz * 25 + t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int co = cot * 32 + cq * 16 + kg * 4 + r;
                  if (co < Cout) {
                    float* p = a.dw + (((size_t)slot * REPMODE_TAPS + tap) * Cout + co) * a.CinTot + a.ci_off + ci;
#ifdef RM_CONV_NOEPI
                    if (acc[t][r] != 12345.678f) continue;      // TIMING BUILD ONLY: sums computed, (practically) never written
#endif
                    if (atomic) unsafeAtomicAdd(p, acc[t][r]);
                    else *p = acc[t][r];
                  }
                }
              }
            }
#pragma unroll
            for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            head = true;
          }
          if (i + 1 < steps) next();
        }
        return;
      }
      }
      // both roles walk the same tile sequence; barrier k separates "tile k staged in buffer k & 1" from its MFMAs, and a
      // buffer is staged again only after the barrier behind the MFMAs that read it
      int boff = 0;
      if (loader) {
        // two tiles in flight (register sets a / b; a lands in buffer 0, b in buffer 1), every load on every path: a fetch
        // behind the last tile asks for out-of-range offsets
        auto tile_fetch = [&](TileRegs& tr, bool real) { fetch_lean(tr, real, n, atz * TZ, aty * TY, atx * TX); };
        if (!have) return;
        tile_fetch(ra, true);
        bool have_b = advance();
        tile_fetch(rb, have_b);
        for (;;) {
          stage_lean(ra, 0);
          if (!have_b) break;
          const bool have_a = advance();
          tile_fetch(ra, have_a);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          stage_lean(rb, LDS_SET);
          if (!have_a) break;
          have_b = advance();
          tile_fetch(rb, have_b);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
      }
      while (have) {
        RM_WSTAMP(tl_ * 3 + 0);
        asm volatile("s_barrier" ::: "memory");
        RM_WSTAMP(tl_ * 3 + 1);
        mma_tile(boff);
        RM_WSTAMP(tl_ * 3 + 2);
        have = advance();
        boff ^= LDS_SET;
#ifdef RM_CONV_TIMING
        ++tl_;
#endif
      }
    } else {
    if (have) fetch();
    while (have) {
      RM_WSTAMP(tl_ * 3 + 0);
      __syncthreads();     // every wave is done with the previous tile in LDS
      stage(0);
      __syncthreads();
      RM_WSTAMP(tl_ * 3 + 1);
      have = advance();
      if (have) fetch();   // in flight during the MFMAs below
      RM_WPRIO(1);
      mma_tile(0);
      RM_WPRIO(0);
      RM_WSTAMP(tl_ * 3 + 2);
#ifdef RM_CONV_TIMING
      ++tl_;
#endif
    }
    }
  } else {
  const bool vec_x = (Cin & 7) == 0, vec_dy = (Cout & 7) == 0;
  for (int n = 0; n < a.N; ++n) {
  if (!in_slot(n)) continue;                          // the workgroup sums over the samples of its slot
  const bf16_t* __restrict__ xn = static_cast<const bf16_t*>(a.x) + (size_t)n * D * H * W * Cin;
  const bf16_t* __restrict__ dyn = static_cast<const bf16_t*>(a.dy) + (size_t)n * D * H * W * Cout;
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int txi = tile % a.ntx, t2 = tile / a.ntx;
    const int tyi = t2 % a.nty, tzi = t2 / a.nty;
    const int z0 = tzi * TZ, y0 = tyi * TY, x0 = txi * TX;
    // every input plane this tile needs for this dz is padding -> nothing to add (uniform branch)
    if (z0 + TZ - 1 + dz - 2 < 0 || z0 + dz - 2 >= D) continue;
    __syncthreads();
    // ---- stage x (transposed)
    for (int it = tid; it < NIT_X; it += 256) {
      const int p = it % NPAIR; int r = it / NPAIR;
      const int cg = r & 3; r >>= 2;
      const int hy = r % HY, zz = r / HY;
      const int zin = z0 + zz + dz - 2, gy = y0 + hy - 2, gx = x0 - 2 + 2 * p;
      const int c = cit * 32 + cg * 8;
      u32x4 v0 = u32x4{0u, 0u, 0u, 0u}, v1 = v0;
      if ((unsigned)zin < (unsigned)D && (unsigned)gy < (unsigned)H && c < Cin) {
        const bf16_t* rowp = xn + ((size_t)(zin * H + gy) * W) * Cin + c;
        if ((unsigned)gx < (unsigned)W) v0 = load8_bf16(rowp + (size_t)gx * Cin, c, Cin, vec_x);
        if ((unsigned)(gx + 1) < (unsigned)W) v1 = load8_bf16(rowp + (size_t)(gx + 1) * Cin, c, Cin, vec_x);
      }
      put_x_pair(cg, zz * HY + hy, p, v0, v1, 0);
    }
    // ---- stage dy (transposed)
    for (int it = tid; it < NIT_DY; it += 256) {
      const int q = it % (TV / 2), cg = it / (TV / 2);
      const int m = 2 * q;
      const int xx = m % TX, yy = (m / TX) % TY, zz = m / (TX * TY);
      const int gz = z0 + zz, gy = y0 + yy, gx = x0 + xx;
      const int c = cot * 32 + cg * 8;
      u32x4 v0 = u32x4{0u, 0u, 0u, 0u}, v1 = v0;
      if (gz < D && gy < H && c < Cout) {
        const bf16_t* rowp = dyn + ((size_t)(gz * H + gy) * W) * Cout + c;
        if (gx < W) v0 = load8_bf16(rowp + (size_t)gx * Cout, c, Cout, vec_dy);
        if (gx + 1 < W) v1 = load8_bf16(rowp + (size_t)(gx + 1) * Cout, c, Cout, vec_dy);
      }
      put_pair(dyT + (cg * 8) * DYS + m * 2, DYS, v0, v1);
    }
    __syncthreads();
    mma_tile(0);
  }
  }
  }
  RM_WSTAMP(60);
  if (a.layout != 0 && a.direct) {
    // The experts' own layout ([co][ci][125] or, K3, [co][ci][27]): a (co, ci) pair's 25 taps of this dz plane are
    // 100 contiguous bytes.  Straight from the accumulators that is 4-byte stores 500 B apart (measured 5x slower
    // than tap-major stores); so each wave transposes its tile through the idle staging LDS, one accumulator
    // register (4 co x 16 ci = 64 pairs) at a time, and stores whole runs.
    __syncthreads();
    float* wl = reinterpret_cast<float*>(smem) + wave * (64 * 25);
    const int nt = a.layout == 1 ? 25 : 9;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pair = kg * 16 + l15;
      if (a.layout == 1) {
#pragma unroll
        for (int t = 0; t < 25; ++t) wl[pair * 25 + t] = acc[t][r];
      } else {
#pragma unroll
        for (int t = 0; t < 25; ++t) {
          const int ty = t / 5, tx = t % 5;
          if (ty >= 1 && ty <= 3 && tx >= 1 && tx <= 3) wl[pair * 9 + (ty - 1) * 3 + (tx - 1)] = acc[t][r];
        }
      }
      for (int i = lane; i < 64 * nt; i += 64) {
        const int pr = i / nt, t = i % nt;
        const int co = cot * 32 + cq * 16 + (pr >> 4) * 4 + r;
        const int cic = cit * 32 + ciq * 16 + (pr & 15);
        if (co < Cout && cic < Cin) {
          float* p = a.layout == 1 ? a.dw + ((size_t)co * Cin + cic) * REPMODE_TAPS + dz * 25 + t
                                   : a.dw + ((size_t)co * Cin + cic) * 27 + (dz - 1) * 9 + t;
#ifdef RM_CONV_NOEPI
          if (wl[i] == 12345.678f)      // TIMING BUILD ONLY: sums computed, (practically) never written
#endif
          *p = wl[i];
        }
      }
    }
    return;
  }
  // 16x16 C/D layout: column (ci) = lane & 15, row (co) = (lane >> 4) * 4 + r
  const int ci = cit * 32 + ciq * 16 + l15;
  if (ci < Cin) {
#pragma unroll
    for (int t = 0; t < 25; ++t) {
      const int tap = dz * 25 + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * 32 + cq * 16 + kg * 4 + r;
        if (co < Cout) {
          float* p;
          if (a.layout == 0) {
            p = a.dw + (((size_t)slot * REPMODE_TAPS + tap) * Cout + co) * a.CinTot + a.ci_off + ci;
          } else if (a.layout == 1) {
            p = a.dw + ((size_t)co * Cin + ci) * REPMODE_TAPS + tap;
          } else {
            const int ty = t / 5, tx = t % 5;            // (dy, dx) of this tap; dz in [1,3] by construction
            if (ty < 1 || ty > 3 || tx < 1 || tx > 3) continue;
            p = a.dw + ((size_t)co * Cin + ci) * 27 + ((dz - 1) * 3 + (ty - 1)) * 3 + (tx - 1);
          }
#ifdef RM_CONV_NOEPI
          if (acc[t][r] != 12345.678f) continue;
#endif
          if (a.direct) *p = acc[t][r];
          else unsafeAtomicAdd(p, acc[t][r]);
        }
      }
    }
  }
#ifdef RM_CONV_TIMING
  __builtin_amdgcn_s_waitcnt(0);      // (stamp 61 after the stores have left)
#endif
  RM_WSTAMP(61);
}

#ifdef RM_CONV_TIMING
}  // namespace
extern "C" int repmode_debug_wgrad_timing(unsigned long long* out) {
  RM_HIP(hipDeviceSynchronize());
  RM_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgrad_timing), sizeof(unsigned long long) * 64 * 64));
  return 0;
}
namespace {
#endif

#ifndef WGRAD_PIPE_MINW
#define WGRAD_PIPE_MINW 16
#endif
#ifndef WGRAD_DIRECT_MIN
#define WGRAD_DIRECT_MIN 512
#endif
#ifndef WGRAD_ROUNDS
#define WGRAD_ROUNDS 2
#endif
template <int TZ, int TY, int TX>
int launch_wgrad_bf16(WgradArgs a, int n, hipStream_t s) {
  a.nty = ceil_div(a.H, TY);
  a.ntx = ceil_div(a.W, TX);
  a.ntiles = ceil_div(a.D, TZ) * a.nty * a.ntx;
  // The voxel range of a (slot, co tile, ci tile, dz plane) is split over `nchunks` workgroups (float atomics from 2 on).
  // The grid runs as rounds of the workgroups resident at once (occupancy x CUs), and a workgroup's fixed costs (first
  // fetch, ~24 k cycles of epilogue = 3.5 tiles of 8x32 voxels, 6 of the smaller ones; more with atomics) are comparable to
  // a short tile loop: candidates are priced as rounds x (overhead + tiles per workgroup), in tile units.  Same box, old
  // rule (two rounds' worth of workgroups) -> priced, batch 8: level 0 254 -> 239 us, level 1 (32 -> 64) 112 -> 80,
  // (64 -> 64) 148 -> 131, level 2 (64 -> 128) 96 -> 83.
  const bool vec = (a.Cin & 7) == 0 && (a.Cout & 7) == 0 && a.W >= WGRAD_PIPE_MINW &&
                   (size_t)a.D * a.H * a.W * (a.Cin > a.Cout ? a.Cin : a.Cout) * 2 < ((size_t)1 << 31);
  // The wave-specialised form (see the kernel's comment) where a workgroup gets a long tile loop: one workgroup per CU halves
  // the workgroups in flight, and its un-overlapped prologue + epilogue (~40 k cycles = 11 of its tile steps) must be a small
  // part of the loop.  Same box, interleaved, us per launch two-workgroup / specialised: level 0 32->32 240.4 / 221.2,
  // 64->32 464.1 / 427.2 (86 and 171 tiles per workgroup); level 1 64->64 131.5 / 135.9, 128->64 252.3 / 270.8 (22 tiles).
  // REPMODE_WGRAD_WS: 0 never, 1 (default) by the tile count, 2 always.
  const int ws_mode = g_wgrad_ws;
  bool ws = false;
  if (vec && TX >= 32 && ws_mode != 0 && !a.dy2) {
    const long fixed1 = (long)a.nslots * a.ncot * a.ncit * a.ndz;
    const long est_nc = fixed1 >= 256 ? 1 : 256 / fixed1;
    const double per_wg = (double)a.ntiles * (n > a.nslots ? (double)n / a.nslots : 1.0) / (double)est_nc;
    ws = ws_mode == 2 || per_wg >= 48.0;
  }
  // workgroups resident at once, per instantiation AND per device (advisor round 3: function statics belonged to whichever
  // device called first): the regular kernel / the three-per-CU form (8x16 tile only) / the wave-specialised form
  static long res_tab[32][3] = {};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  long* res = res_tab[dev & 31];
  if (!res[0]) {
    int per_cu = 0, cus = 0;
    RM_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
    if constexpr (TX >= 16 && TZ == 1) {          // (the wave-specialised instantiation: one workgroup per CU)
      RM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv5_wgrad_bf16_kernel<TZ, TY, TX, true, false, true>, 512, 0));
      res[2] = (long)(per_cu > 0 ? per_cu : 1) * cus;
    }
    if (TX == 16) {
      RM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv5_wgrad_bf16_kernel<TZ, TY, TX, true, TX == 16>, 256, 0));
      res[1] = (long)(per_cu > 0 ? per_cu : 1) * cus;
    }
    RM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv5_wgrad_bf16_kernel<TZ, TY, TX, true>, 256, 0));
    res[0] = (long)(per_cu > 0 ? per_cu : 1) * cus;
  }
  const long resident2 = res[0], resident3 = res[1], resident_ws = res[2];
  // Stream-K form (see the kernel): merged-formulation launches (slot layout, one job) on volumes >= 32 voxels wide, at most 64
  // samples (one ballot tells a wave its slot's samples).  REPMODE_WGRAD_WS / repmode_set_wgrad_ws = 3 (default): where
  // eligible; 0-2 keep the regular grid (1: wave-specialised by tile count, 2: always).
  // The 16-voxel tile (level 2): a tile step is 100 MFMAs = 1.6 k cycles, about what four loader waves need to fetch AND
  // transpose a tile (~150 VALU operations + 23 ds_write_b32 per thread), so stream-K runs at the loaders' pace there:
  // same box, us per launch regular / stream-K (profiles/r04_wgrad_sk.txt): 64 -> 128 (320 units) 79.7 / 64.9, but 128 -> 128
  // (640 units: the regular grid is one full round of the three-per-CU form) 79.6 / 85-91.  REPMODE_WGRAD_SK16: 0 never,
  // 1 (default) below 512 units, 2 always.
  static const int sk16 = []() { const char* e = getenv("REPMODE_WGRAD_SK16"); return e ? atoi(e) : 1; }();
  const long sk_units = (long)a.nslots * a.ncot * a.ncit * a.ndz;
  if (TX >= 16 && (TX >= 32 || sk16 == 2 || (sk16 == 1 && sk_units < 512)) && TZ == 1 && ws_mode == 3 && vec && !a.dy2 && a.layout == 0 && n <= 64 && a.sample_slot &&
      !repmode_deterministic() && resident_ws > 0) {
    long vt = 0;
    for (int i = 0; i < a.ndz; ++i) {
      const int dzz = a.dz_lo + i, lo = 2 - dzz > 0 ? 2 - dzz : 0, hi = a.D + 2 - dzz < a.D ? a.D + 2 - dzz : a.D;
      if (hi > lo) vt += (long)(hi - lo) * a.nty * a.ntx;
    }
    const long total = (long)n * vt * a.ncot * a.ncit;
    // Workgroups: at most one per CU (all resident: nobody waits for a slot), none shorter than ~8 tile steps, and no more
    // than the float atomics can follow: a workgroup shares at most the first and the last unit of its range with a
    // neighbour -- two flushes of 25 x 32 x 32 floats through memory-side atomics that run at ~1.2 TB/s
    // (profiles/r03_atomics.txt): 0.17 us of the chip's atomic throughput per workgroup, against total / g tile steps of
    // KSTEPS x 25 MFMAs (16 cycles each, ~1.6 GHz under load) each.
    const double t_step = (double)(TZ * TY * TX / 32) * 25.0 * 16.0 / 1600.0, t_atomic = 2.0 * 25.0 * 32.0 * 32.0 * 4.0 / 1.23e6;
    long g = (long)std::sqrt((double)total * t_step / t_atomic);
    const long room = resident_ws - repmode_reserve_cus() > 8 ? resident_ws - repmode_reserve_cus() : 8;   // (CUs left to a communication kernel)
    if (g > room) g = room;
    if (g > total / 8) g = total / 8;
    if (g < 1) g = 1;
    if (total > 0 && total < (1L << 40)) {
      if (a.plan_out) { *a.plan_out = 0; return REPMODE_OK; }     // (shared units are added with float atomics)
      a.sk_total = total;
      a.nchunks = 1; a.tiles_per_block = a.ntiles; a.direct = 0;
      if (!a.prezeroed) RM_HIP(hipMemsetAsync(a.dw, 0, (size_t)a.nslots * REPMODE_TAPS * a.Cout * a.CinTot * sizeof(float), s));
      repmode_prof_begin(REPMODE_PROF_WGRAD, 2.0 * n * a.D * a.H * a.W * (double)a.Cin * a.Cout * REPMODE_TAPS, s);
      if constexpr (TX >= 16 && TZ == 1)
        hipLaunchKernelGGL((conv5_wgrad_bf16_kernel<TZ, TY, TX, true, false, true>), dim3((unsigned)g), dim3(512), 0, s, a);
      return REPMODE_OK;
    }
  }
  const double ov = TX >= 32 ? 3.5 : 6.0;
  const double tiles = (double)a.ntiles * (n > a.nslots ? (double)n / a.nslots : 1.0);
  // best chunk count of a job of `ndz` planes on `resident` slots; returns its price
  auto price = [&](int ndz, long resident, long* chunks) -> double {
    const long fixed = (long)a.nslots * a.ncot * a.ncit * ndz;
    double best = 1e30;
    *chunks = 1;
    const long nc_max = repmode_deterministic() ? repmode_det_cap(RM_DET_WGRAD) : 64;      // (deterministic: at most two addends per element of the cleared dw)
    for (long nc = 1; nc <= nc_max && nc <= a.ntiles; ++nc) {
      const double rounds = (double)((fixed * nc + resident - 1) / resident);
      const double cost = rounds * ((nc > 1 ? ov + 2.0 : ov) + (double)((long)((tiles + nc - 1) / nc)));
      if (cost < best * 0.97) { best = cost; *chunks = nc; }
    }
    return best;
  };
  // The 8x16 tile (level 2) has a second form with three workgroups per CU (168 registers: no window ahead, a few
  // spilled dwords): a slower round of 1.5x the workgroups.  It pays when it saves a round -- 640 workgroups (8 slots,
  // 128 -> 128) are one round of 768 instead of two of 512: 88 -> 77 us; 960 (12 slots) are two rounds either way:
  // 189 -> 235 us -- so both forms are priced, the dense one at 1.25 per round.
  bool dense = false;
  if (TX == 16 && vec && !ws && resident3 > resident2) {
    long c2, c3;
    double p2 = price(a.ndz, resident2, &c2), p3 = 1.25 * price(a.ndz, resident3, &c3);
    if (a.dy2) { p2 += price(a.ndz2, resident2, &c2); p3 += 1.25 * price(a.ndz2, resident3, &c3); }
    dense = p3 < p2;
  }
  const long resident = ws ? resident_ws : dense ? resident3 : resident2;
  auto plan = [&](int ndz, int layout, int* tiles_per_block, int* nchunks, int* direct) -> long {
    const long fixed = (long)a.nslots * a.ncot * a.ncit * ndz;
    long want_chunks = 1;
    // (the experts' own layout is written by whole-range workgroups only: atomics 500 bytes apart are 5x slower than
    // the tile loop they would shorten)
    if (layout == 0) price(ndz, resident, &want_chunks);
    *tiles_per_block = ceil_div(a.ntiles, (int)want_chunks);
    *nchunks = ceil_div(a.ntiles, *tiles_per_block);
    *direct = *nchunks == 1;
    return fixed * *nchunks;
  };
  long grid = plan(a.ndz, a.layout, &a.tiles_per_block, &a.nchunks, &a.direct);
  if (a.plan_out) { *a.plan_out = a.direct && !a.dy2; return REPMODE_OK; }
  if (!a.direct && !a.prezeroed) RM_HIP(hipMemsetAsync(a.dw, 0, (size_t)a.nslots * (a.layout == 2 ? 27 : REPMODE_TAPS) * a.Cout * a.CinTot * sizeof(float), s));
  if (a.dy2) {                       // dual launch: the second job is planned the same way, its workgroups follow the first's
    a.grid0 = (int)grid;
    grid += plan(a.ndz2, a.layout2, &a.tiles_per_block2, &a.nchunks2, &a.direct2);
    RM_REQUIRE(a.layout2 == 0 || a.direct2, "conv5_wgrad_dual: job 2 cannot write the expert layout with atomics");
    if (!a.direct2 && !a.prezeroed) RM_HIP(hipMemsetAsync(a.dw2, 0, (size_t)REPMODE_TAPS * a.Cout * a.CinTot * sizeof(float), s));
  }
  RM_REQUIRE(grid < (1L << 31), "conv5_wgrad: grid too large");
  repmode_prof_begin(REPMODE_PROF_WGRAD, 2.0 * n * a.D * a.H * a.W * (double)a.Cin * a.Cout * REPMODE_TAPS, s);
  // (levels with W < 16 hand a workgroup only a tile or two per sample: nothing to overlap, and the plain loop is ~15 % faster there)
  // The per-expert levels' tiles (two z planes, x extent 8): the wave-specialised form with whole-range workgroups writing
  // the experts' layout -- a tile step there is 100 MFMAs per wave (1.6 k cycles) against a 3-4 k-cycle fetch, a second
  // register set does not fit beside the 100 accumulators (tried: 90 spilled dwords), so the fetch + transposition go to
  // loader waves with registers of their own.  One workgroup per CU, so a launch is rounds of 256 workgroups, each with its
  // own un-overlapped first fetch and epilogue: same box, both experts' gradients of a level-3 layer at batch 8, two-workgroup
  // form -> this one: 128 -> 256 (256 units) 58.8 -> 45.9 us, 256 -> 256 (512) 74.0 -> 69.3, 512 -> 256 (1024) 130 -> 131;
  // level 4 (2048 units of <= 8 short steps) 54.4 -> 64.4, 95.8 -> 120: taken for level 3 up to two rounds.
  // Round 6, with the loader waves' lean fetch (two tiles in flight, ~1/3 of the instructions): the MFMA waves no longer wait
  // at the barrier (stamps: 900 -> 128 cycles per tile), 256 -> 256 71 -> 62 us, and the form also wins beyond two rounds
  // (512 -> 256, 1024 units: 127.8 -> 109.7 us); level 4 unchanged (54.1 / 53.8, 95.7 / 100.3): level 3 always, level 4 never.
  // REPMODE_WGRAD_WS8: 0 never, 1 (default) level 3, 2 wherever the layouts allow.
  static const int ws8_env = []() { const char* e = getenv("REPMODE_WGRAD_WS8"); return e ? atoi(e) : 1; }();
  bool ws8 = false;
  if constexpr (TZ == 2)      // (`vec` is off below WGRAD_PIPE_MINW: the two-workgroup form's own pipelining loses there)
    ws8 = ws8_env != 0 && (a.Cin & 7) == 0 && (a.Cout & 7) == 0 &&
          (size_t)a.D * a.H * a.W * (a.Cin > a.Cout ? a.Cin : a.Cout) * 2 < ((size_t)1 << 31) && a.layout != 0 &&
          (!a.dy2 || a.layout2 != 0) && a.nchunks == 1 && (TY == 8 || ws8_env >= 2);
  if (ws8) {
    if constexpr (TZ == 2)
      hipLaunchKernelGGL((conv5_wgrad_bf16_kernel<TZ, TY, TX, true, false, true>), dim3((unsigned)grid), dim3(512), 0, s, a);
  } else if (ws)
    hipLaunchKernelGGL((conv5_wgrad_bf16_kernel<TZ, TY, TX, true, false, TX >= 32>), dim3((unsigned)grid), dim3(TX >= 32 ? 512 : 256), 0, s, a);      // (ws is only ever set for TX >= 32)
  else if (vec && dense)
    hipLaunchKernelGGL((conv5_wgrad_bf16_kernel<TZ, TY, TX, true, TX == 16>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else if (vec)
    hipLaunchKernelGGL((conv5_wgrad_bf16_kernel<TZ, TY, TX, true>), dim3((unsigned)grid), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv5_wgrad_bf16_kernel<TZ, TY, TX, false>), dim3((unsigned)grid), dim3(256), 0, s, a);
  return REPMODE_OK;
}

}  // namespace

extern "C" int repmode_set_wgrad_ws(int mode) { g_wgrad_ws = mode; return REPMODE_OK; }
extern "C" int repmode_get_wgrad_ws(void) { return g_wgrad_ws; }

extern "C" int repmode_conv5_wgrad_ex(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                                      float* dw, int n, int d, int h, int wdim, int cin, int cout, int dtype,
                                      int centre3, void* stream);

extern "C" int repmode_conv5_wgrad(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                                   float* dw, int n, int d, int h, int wdim, int cin, int cout, int dtype,
                                   void* stream) {
  return repmode_conv5_wgrad_ex(x, dy, sample_slot, nslots, dw, n, d, h, wdim, cin, cout, dtype, 0, stream);
}

extern "C" int repmode_conv5_wgrad_part(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                                        float* dw, int n, int d, int h, int wdim, int cin, int cin_total, int ci_off,
                                        int cout, int dtype, int centre3, void* stream);

extern "C" int repmode_conv5_wgrad_ex(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                                      float* dw, int n, int d, int h, int wdim, int cin, int cout, int dtype,
                                      int centre3, void* stream) {
  return repmode_conv5_wgrad_part(x, dy, sample_slot, nslots, dw, n, d, h, wdim, cin, cin, 0, cout, dtype, centre3, stream);
}

// Filter gradient of the input channels [ci_off, ci_off + cin) of a layer with cin_total input channels: x holds
// only those cin channels ([N][D][H][W][cin]) and dw is the whole layer's [nslots][125][cout][cin_total] (slot
// layout only).  The two tensors of a skip connection (repmode_conv5_pair) are handled by two such calls on one dw,
// which the caller must have cleared (mode bit 3).
extern "C" int repmode_conv5_wgrad_part(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                                        float* dw, int n, int d, int h, int wdim, int cin, int cin_total, int ci_off,
                                        int cout, int dtype, int centre3, void* stream) {
  RM_REQUIRE(x && dy && sample_slot && dw, "conv5_wgrad: null pointer");
  RM_REQUIRE(cin_total >= cin && ci_off >= 0 && ci_off + cin <= cin_total, "conv5_wgrad: bad channel range");
  RM_REQUIRE(cin_total == cin || ((centre3 & 8) && (centre3 & 7) < 2),
             "conv5_wgrad: a partial channel range needs the slot layout and a cleared dw (mode bit 3)");
  RM_REQUIRE(n > 0 && nslots > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0, "conv5_wgrad: bad shape");
  RM_REQUIRE(dtype == REPMODE_F32 || dtype == REPMODE_BF16, "conv5_wgrad: bad dtype %d", dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  WgradArgs a{};
  a.x = x; a.dy = dy; a.sample_slot = sample_slot; a.dw = dw;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin; a.Cout = cout;
  a.CinTot = cin_total; a.ci_off = ci_off;
  a.ncot = ceil_div(cout, 32);
  a.ncit = ceil_div(cin, 32);
  a.nslots = nslots;
  // centre3: 0 = all taps, slot layout; 1 = planes dz in [1,3] only, slot layout; 2 = all taps written in the
  // experts' own [co][ci][125] layout; 3 = centred 3x3x3 taps written as [co][ci][27] (2, 3: nslots == 1)
  // (bit 3 of the mode word: dw has been cleared by the caller -- one pooled memset per step instead of one per call)
  a.prezeroed = (centre3 & 8) ? 1 : 0;
  centre3 &= 7;
  RM_REQUIRE(centre3 >= 0 && centre3 <= 3 && (centre3 < 2 || nslots == 1), "conv5_wgrad: bad mode %d", centre3);
  a.dz_lo = (centre3 == 1 || centre3 == 3) ? 1 : 0;
  a.ndz = (centre3 == 1 || centre3 == 3) ? 3 : 5;
  a.layout = centre3 == 2 ? 1 : (centre3 == 3 ? 2 : 0);
  if (dtype == REPMODE_BF16) {
    RM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "conv5_wgrad: pointers must be 16-byte aligned");
    if (centre3 == 0) {
      // the column walk (csrc/conv5_wgrad_col.hip): a workgroup owns all 125 taps of a (slot, 16 co, 16 ci) tile
      const WgColCall cc{x, dy, sample_slot, dw, n, d, h, wdim, cin, cout, cin_total, ci_off, nslots, a.prezeroed, nullptr};
      if (repmode_wgrad_col_eligible(cc)) {
        const int rc = repmode_wgrad_col_launch(cc, s);
        if (rc != REPMODE_OK) return rc;
        repmode_prof_end(s);
        RM_LAUNCH_CHECK("conv5_wgrad_col");
        return REPMODE_OK;
      }
    }
    int rc;
    if (wdim >= 32 && h >= 8) rc = launch_wgrad_bf16<1, 8, 32>(a, n, s);
    else if (wdim >= 32) rc = launch_wgrad_bf16<1, 4, 32>(a, n, s);
    else if (wdim >= 16) rc = launch_wgrad_bf16<1, 8, 16>(a, n, s);
    else if (wdim >= 8) rc = launch_wgrad_bf16<2, 8, 8>(a, n, s);
    else rc = launch_wgrad_bf16<2, 4, 8>(a, n, s);
    if (rc != REPMODE_OK) return rc;
  } else {
    a.nty = ceil_div(h, TY);
    a.ntx = ceil_div(wdim, TX);
    a.ntiles = d * a.nty * a.ntx;
    const long fixed = (long)nslots * a.ncot * a.ncit * a.ndz;
    long want_chunks = (2048 + fixed - 1) / fixed;       // aim at >= 2048 workgroups
    if (want_chunks < 1) want_chunks = 1;
    if (repmode_deterministic() && want_chunks > repmode_det_cap(RM_DET_WGRAD)) want_chunks = repmode_det_cap(RM_DET_WGRAD);
    if (want_chunks > a.ntiles) want_chunks = a.ntiles;
    a.tiles_per_block = ceil_div(a.ntiles, (int)want_chunks);
    a.nchunks = ceil_div(a.ntiles, a.tiles_per_block);
    a.direct = a.nchunks == 1;
    const long grid = fixed * a.nchunks;
    RM_REQUIRE(grid < (1L << 31), "conv5_wgrad: grid too large");
    if (!a.direct && !a.prezeroed) RM_HIP(hipMemsetAsync(dw, 0, (size_t)nslots * (a.layout == 2 ? 27 : REPMODE_TAPS) * cout * a.CinTot * sizeof(float), s));
    repmode_prof_begin(REPMODE_PROF_WGRAD, 2.0 * n * d * h * wdim * (double)cin * cout * REPMODE_TAPS, s);
    hipLaunchKernelGGL(conv5_wgrad_f32c_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, a);
  }
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("conv5_wgrad");
  return REPMODE_OK;
}

// Does repmode_conv5_wgrad[_ex / _part] with these arguments write EVERY element of its dw range by plain stores (*direct = 1:
// dw needs no clearing -- the caller may hand over uninitialised memory, and pass mode bit 3 to say "no memset"), or does it
// add partial sums with float atomics onto a dw that must be all zero (*direct = 0)?  The same planning code as the launch
// (same switches, same device), without launching.  The operator library asks before it takes the buffer out of the step's
// pooled memset: at batch 8 that keeps 260 MB of level-2 filter gradients out of it.
extern "C" int repmode_conv5_wgrad_plan(int nslots, int n, int d, int h, int wdim, int cin, int cout, int dtype, int centre3, int* direct) {
  RM_REQUIRE(direct, "conv5_wgrad_plan: null pointer");
  RM_REQUIRE(n > 0 && nslots > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0, "conv5_wgrad_plan: bad shape");
  *direct = 0;
  centre3 &= 7;
  static const int enabled = []() { const char* e = getenv("REPMODE_WGRAD_PLAN"); return e ? atoi(e) : 1; }();   // (0: always a cleared buffer, for A/B)
  if (!enabled || dtype != REPMODE_BF16 || centre3 > 1) return REPMODE_OK;          // (float32 parity path / expert layouts: keep the cleared buffer)
  static const int32_t dummy_slot = 0;
  if (centre3 == 0) {
    WgColCall cc{nullptr, nullptr, &dummy_slot, nullptr, n, d, h, wdim, cin, cout, cin, 0, nslots, 0, direct};
    if (repmode_wgrad_col_eligible(cc)) return repmode_wgrad_col_launch(cc, nullptr);
  }
  WgradArgs a{};
  a.sample_slot = &dummy_slot;            // (only tested for NULL by the planner)
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin; a.Cout = cout;
  a.CinTot = cin; a.ci_off = 0;
  a.ncot = ceil_div(cout, 32);
  a.ncit = ceil_div(cin, 32);
  a.nslots = nslots;
  a.dz_lo = centre3 == 1 ? 1 : 0;
  a.ndz = centre3 == 1 ? 3 : 5;
  a.layout = 0;
  a.plan_out = direct;
  if (wdim >= 32 && h >= 8) return launch_wgrad_bf16<1, 8, 32>(a, n, nullptr);
  if (wdim >= 32) return launch_wgrad_bf16<1, 4, 32>(a, n, nullptr);
  if (wdim >= 16) return launch_wgrad_bf16<1, 8, 16>(a, n, nullptr);
  if (wdim >= 8) return launch_wgrad_bf16<2, 8, 8>(a, n, nullptr);
  return launch_wgrad_bf16<2, 4, 8>(a, n, nullptr);
}

// Two filter gradients over the SAME input in one launch (bf16, every sample in one slot): the per-expert formulation's
// 5x5x5 expert (dy_a: its gate-scaled output gradient) and 3x3x3 expert (dy_b).  mode_a / mode_b as repmode_conv5_wgrad_ex's
// mode word (0 / 1 tap-major, 2 / 3 the experts' own layouts, bit 3: dw cleared by the caller).  The 3x3x3 job alone is a
// latency-bound launch (27 taps, three planes); behind the 5x5x5 job's workgroups in one grid it fills their tail.
extern "C" int repmode_conv5_wgrad_dual(const void* x, const void* dy_a, const void* dy_b, float* dw_a, float* dw_b, int n, int d,
                                        int h, int wdim, int cin, int cout, int mode_a, int mode_b, void* stream) {
  RM_REQUIRE(x && dy_a && dy_b && dw_a && dw_b, "conv5_wgrad_dual: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0 && cout > 0, "conv5_wgrad_dual: bad shape");
  RM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy_a & 15) == 0 && ((uintptr_t)dy_b & 15) == 0, "conv5_wgrad_dual: pointers must be 16-byte aligned");
  RM_REQUIRE(((mode_a ^ mode_b) & 8) == 0, "conv5_wgrad_dual: both outputs cleared by the caller, or neither");
  hipStream_t s = static_cast<hipStream_t>(stream);
  WgradArgs a{};
  a.x = x; a.dy = dy_a; a.dw = dw_a; a.dy2 = dy_b; a.dw2 = dw_b; a.sample_slot = nullptr;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin; a.Cout = cout;
  a.CinTot = cin; a.ci_off = 0;
  a.ncot = ceil_div(cout, 32);
  a.ncit = ceil_div(cin, 32);
  a.nslots = 1;
  a.prezeroed = (mode_a & 8) ? 1 : 0;
  const int ma = mode_a & 7, mb = mode_b & 7;
  RM_REQUIRE(ma >= 0 && ma <= 3 && mb >= 0 && mb <= 3, "conv5_wgrad_dual: bad mode");
  if (ma == 2 && mb == 3) {
    // the column walk's dual form (csrc/conv5_wgrad_col.hip): both experts' gradients from one staging of x, several samples
    // per step, the 3x3x3 job as 27 taps
    WgColCall cc{x, dy_a, nullptr, dw_a, n, d, h, wdim, cin, cout, cin, 0, 1, 0, nullptr};
    cc.dy2 = dy_b; cc.dw2 = dw_b;
    if (repmode_wgrad_col_eligible(cc)) {
      const int rc = repmode_wgrad_col_launch(cc, s);
      if (rc != REPMODE_OK) return rc;
      repmode_prof_end(s);
      RM_LAUNCH_CHECK("conv5_wgrad_col (dual)");
      return REPMODE_OK;
    }
  }
  a.dz_lo = (ma == 1 || ma == 3) ? 1 : 0;   a.ndz = (ma == 1 || ma == 3) ? 3 : 5;   a.layout = ma == 2 ? 1 : (ma == 3 ? 2 : 0);
  a.dz_lo2 = (mb == 1 || mb == 3) ? 1 : 0;  a.ndz2 = (mb == 1 || mb == 3) ? 3 : 5;  a.layout2 = mb == 2 ? 1 : (mb == 3 ? 2 : 0);
  int rc;
  if (wdim >= 32 && h >= 8) rc = launch_wgrad_bf16<1, 8, 32>(a, n, s);
  else if (wdim >= 32) rc = launch_wgrad_bf16<1, 4, 32>(a, n, s);
  else if (wdim >= 16) rc = launch_wgrad_bf16<1, 8, 16>(a, n, s);
  else if (wdim >= 8) rc = launch_wgrad_bf16<2, 8, 8>(a, n, s);
  else rc = launch_wgrad_bf16<2, 4, 8>(a, n, s);
  if (rc != REPMODE_OK) return rc;
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("conv5_wgrad_dual");
  return REPMODE_OK;
}

// ------------------------------------------------------------------------------------------------
// Thin filter gradient: one of the two channel counts is 1 (first layer: Cin = 1; last layer: Cout = 1).
//
//   Cin  == 1:  dw[tap][co] = sum_v dy[v][co] * x[v + tap]          (A = dy, b = x, shift +tap)
//   Cout == 1:  dw[tap][ci] = sum_v dy[v] * x[v + tap][ci]
//                           = sum_u x[u][ci] * dy[u - tap]          (A = x,  b = dy, shift -tap)
//
// The general kernel would spend a 32x32 channel tile on a 32x1 problem.  Here the 125 taps take the
// place of the missing channel dimension: GEMM rows = taps (8 tiles of 16), columns = the 32 channels of
// the multi-channel tensor A, K = voxels.  The single-channel tensor b becomes a Toeplitz operand: the
// fragment of tap row t and voxel group v0..v0+7 is the 8-element window of b starting at v0 + shift(t),
// read from a halo tile of b in LDS with five ds_read_b32 and four v_alignbit_b32 (2-byte alignment
// differs per lane).  A is transposed to [c][voxel] while staging, exactly as in the general kernel.
namespace {

constexpr int TH_TY = 8, TH_TX = 32;                 // voxel tile: one z plane, 8 rows of 32
constexpr int TH_RL = 48;                            // stored b row: x0-2 .. x0+45 (window reads stay inside)
constexpr int TH_HY = TH_TY + 4;
constexpr int TH_BPL = TH_HY * TH_RL;                // elements per b plane
constexpr int TH_AS = TH_TY * TH_TX * 2 + 16;        // bytes per channel row of A^T (odd multiple of 16)

struct ThinArgs {
  const bf16_t* a;      // multi-channel tensor [N][D][H][W][C]
  const bf16_t* b;      // single-channel tensor [N][D][H][W]
  const int32_t* sample_slot;
  float* dw;            // [nslots][125][C]
  int N, D, H, W, C, flip, nty, ntx, ntiles, tiles_per_block, nchunks, nct;
};

__global__ __launch_bounds__(256) void conv5_wgrad_thin_kernel(ThinArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[32 * TH_AS + 5 * TH_BPL * 2];
  unsigned char* aT = smem;                                        // [32 c][TY*TX] bf16
  bf16_t* bh = reinterpret_cast<bf16_t*>(smem + 32 * TH_AS);       // [5 planes][HY][RL] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  int bid = blockIdx.x;
  const int chunk = bid % a.nchunks; bid /= a.nchunks;
  const int ct = bid % a.nct;
  const int slot = bid / a.nct;
  const int D = a.D, H = a.H, W = a.W, C = a.C;
  const bool vec = (C & 7) == 0;

  // this wave's two tap tiles (16 rows each); per lane the tap of its row and the element shift it implies
  int shift[2];
  int tapi[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tap = (wave * 2 + t) * 16 + l15;
    tapi[t] = tap;
    const int te = min(tap, REPMODE_TAPS - 1);
    const int tf = a.flip ? REPMODE_TAPS - 1 - te : te;             // Cout == 1: shift by -tap == flipped tap
    const int dz = tf / 25, dy = (tf / 5) % 5, dx = tf % 5;
    shift[t] = dz * TH_BPL + dy * TH_RL + dx;                      // relative to (z-2, y-2, x-2) = halo origin
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int t_begin = chunk * a.tiles_per_block;
  const int t_end = min(a.ntiles, t_begin + a.tiles_per_block);
  // ---- K loop of a staged tile: one tile row (32 voxels = 4 groups of 8) per step
  auto mma_tile = [&]() {
#pragma unroll 2
    for (int ry = 0; ry < TH_TY; ++ry) {
      u32x4 bf[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        bf[c] = *reinterpret_cast<const u32x4*>(aT + (c * 16 + l15) * TH_AS + (ry * TH_TX + kg * 8) * 2);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // window of 8 elements of b starting at element e0 (2-byte granularity)
        const int e0 = ry * TH_RL + kg * 8 + shift[t];
        const uint32_t* p = reinterpret_cast<const uint32_t*>(bh) + (e0 >> 1);
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
        const uint32_t sh = (e0 & 1) * 16;
        const u32x4 af = u32x4{__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh),
                               __builtin_amdgcn_alignbit(d3, d2, sh), __builtin_amdgcn_alignbit(d4, d3, sh)};
#pragma unroll
        for (int c = 0; c < 2; ++c)
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af),
                                                              __builtin_bit_cast(bf16x8, bf[c]), acc[t][c], 0, 0, 0);
      }
    }
  };
  if (vec && (W & 1) == 0) {
    // Round 6: the tile sequence is software-pipelined (the NEXT tile's global loads travel while this one is multiplied)
    // and the single-channel halo is fetched as dwords (two x-adjacent voxels: x0 - 2 is even and W is even, so a pair is
    // inside or outside the volume together).  Before: per tile two barriers around a stage-then-compute body whose halo
    // came in as 11 two-byte loads per thread -- 45 us per launch for 67 MB (1.5 TB/s).
    constexpr int NA = (TH_TY * TH_TX / 2) * 4 / 256;          // A items per thread (2)
    constexpr int NB = (5 * TH_HY * (TH_RL / 2) + 255) / 256;  // halo dwords per thread (6)
    u32x4 a0[NA], a1[NA];
    uint32_t bw[NB];
    auto fetch = [&](int n, int tile) {
      const int txi = tile % a.ntx, t2 = tile / a.ntx;
      const int tyi = t2 % a.nty, z = t2 / a.nty;
      const int y0 = tyi * TH_TY, x0 = txi * TH_TX;
      const bf16_t* __restrict__ an = a.a + (size_t)n * D * H * W * C;
      const bf16_t* __restrict__ bn = a.b + (size_t)n * D * H * W;
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const int it = u * 256 + tid;
        const int q = it % (TH_TY * TH_TX / 2), cg = it / (TH_TY * TH_TX / 2);
        const int m = 2 * q, xx = m % TH_TX, yy = m / TH_TX;
        const int gy = y0 + yy, gx = x0 + xx, c = ct * 32 + cg * 8;
        a0[u] = a1[u] = u32x4{0u, 0u, 0u, 0u};
        if (gy < H && c < C) {
          const bf16_t* rowp = an + ((size_t)(z * H + gy) * W) * C + c;
          if (gx < W) a0[u] = *reinterpret_cast<const u32x4*>(rowp + (size_t)gx * C);
          if (gx + 1 < W) a1[u] = *reinterpret_cast<const u32x4*>(rowp + (size_t)(gx + 1) * C);
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int it = u * 256 + tid;
        const int dwx = it % (TH_RL / 2), r = it / (TH_RL / 2);
        const int yy = r % TH_HY, pz = r / TH_HY;
        const int gz = z + pz - 2, gy = y0 + yy - 2, gx = x0 + 2 * dwx - 2;
        bw[u] = 0u;
        if (it < 5 * TH_HY * (TH_RL / 2) && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
          bw[u] = *reinterpret_cast<const uint32_t*>(bn + ((size_t)gz * H + gy) * W + gx);
      }
    };
    auto stage = [&]() {
#pragma unroll
      for (int u = 0; u < NA; ++u) {
        const int it = u * 256 + tid;
        const int q = it % (TH_TY * TH_TX / 2), cg = it / (TH_TY * TH_TX / 2);
        unsigned char* dst = aT + (cg * 8) * TH_AS + q * 4;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          *reinterpret_cast<uint32_t*>(dst + k * TH_AS) = bf16_elem(a0[u], k) | (bf16_elem(a1[u], k) << 16);
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int it = u * 256 + tid;
        if (it < 5 * TH_HY * (TH_RL / 2)) reinterpret_cast<uint32_t*>(bh)[it] = bw[u];
      }
    };
    // the work sequence: (sample of this slot, tile of this chunk)
    int n = -1, tile = t_end;
    auto advance = [&]() -> bool {
      if (++tile < t_end) return true;
      tile = t_begin;
      do { ++n; } while (n < a.N && a.sample_slot[n] != slot);
      return n < a.N && t_begin < t_end;
    };
    bool have = advance();
    if (have) fetch(n, tile);
    while (have) {
      __syncthreads();          // every wave is done with the previous tile in LDS
      stage();
      __syncthreads();
      have = advance();
      if (have) fetch(n, tile); // in flight during the MFMAs below
      mma_tile();
    }
  } else {
  for (int n = 0; n < a.N; ++n) {
    if (a.sample_slot[n] != slot) continue;
    const bf16_t* __restrict__ an = a.a + (size_t)n * D * H * W * C;
    const bf16_t* __restrict__ bn = a.b + (size_t)n * D * H * W;
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int txi = tile % a.ntx, t2 = tile / a.ntx;
      const int tyi = t2 % a.nty, z = t2 / a.nty;
      const int y0 = tyi * TH_TY, x0 = txi * TH_TX;
      __syncthreads();
      // ---- stage A^T: (voxel pair, channel group) items, two x-adjacent voxels per ds_write_b32
      for (int it = tid; it < (TH_TY * TH_TX / 2) * 4; it += 256) {
        const int q = it % (TH_TY * TH_TX / 2), cg = it / (TH_TY * TH_TX / 2);
        const int m = 2 * q, xx = m % TH_TX, yy = m / TH_TX;
        const int gy = y0 + yy, gx = x0 + xx, c = ct * 32 + cg * 8;
        u32x4 v0 = u32x4{0u, 0u, 0u, 0u}, v1 = v0;
        if (gy < H && c < C) {
          const bf16_t* rowp = an + ((size_t)(z * H + gy) * W) * C + c;
          if (gx < W) v0 = load8_bf16(rowp + (size_t)gx * C, c, C, vec);
          if (gx + 1 < W) v1 = load8_bf16(rowp + (size_t)(gx + 1) * C, c, C, vec);
        }
        unsigned char* dst = aT + (cg * 8) * TH_AS + m * 2;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          *reinterpret_cast<uint32_t*>(dst + k * TH_AS) = bf16_elem(v0, k) | (bf16_elem(v1, k) << 16);
      }
      // ---- stage the halo of b: planes z-2..z+2, rows y0-2..y0+TY+1, x0-2..x0+RL-3 (zero outside)
      for (int it = tid; it < 5 * TH_BPL; it += 256) {
        const int xx = it % TH_RL, r = it / TH_RL;
        const int yy = r % TH_HY, pz = r / TH_HY;
        const int gz = z + pz - 2, gy = y0 + yy - 2, gx = x0 + xx - 2;
        bf16_t v = 0;
        if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
          v = bn[((size_t)gz * H + gy) * W + gx];
        bh[it] = v;
      }
      __syncthreads();
      mma_tile();
    }
  }
  }
  // 16x16 C/D layout: column (channel) = lane & 15, row (tap within the tile) = (lane >> 4) * 4 + r
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ch = ct * 32 + c * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tap = (wave * 2 + t) * 16 + kg * 4 + r;
        if (tap < REPMODE_TAPS && ch < C)
          unsafeAtomicAdd(a.dw + ((size_t)slot * REPMODE_TAPS + tap) * C + ch, acc[t][c][r]);
      }
    }
}

}  // namespace

// dw[slot][tap][c] (float, overwritten).  a: [N][D][H][W][C] bf16, b: [N][D][H][W] bf16.
// flip == 0: dw[tap][c] = sum_v a[v][c] * b[v + tap]   (first layer:  a = dy, b = x)
// flip != 0: dw[tap][c] = sum_v a[v][c] * b[v - tap]   (last layer:   a = x,  b = dy)
extern "C" int repmode_conv5_wgrad_thin(const void* a_t, const void* b_t, const int32_t* sample_slot, int nslots,
                                        float* dw, int n, int d, int h, int wdim, int c, int flip, void* stream) {
  RM_REQUIRE(a_t && b_t && sample_slot && dw, "conv5_wgrad_thin: null pointer");
  RM_REQUIRE(n > 0 && nslots > 0 && d > 0 && h > 0 && wdim > 0 && c > 0, "conv5_wgrad_thin: bad shape");
  RM_REQUIRE(((uintptr_t)a_t & 15) == 0, "conv5_wgrad_thin: pointer must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  ThinArgs a{};
  a.a = static_cast<const bf16_t*>(a_t); a.b = static_cast<const bf16_t*>(b_t);
  a.sample_slot = sample_slot; a.dw = dw;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.C = c; a.flip = flip & 1;
  a.nty = ceil_div(h, TH_TY); a.ntx = ceil_div(wdim, TH_TX);
  a.ntiles = d * a.nty * a.ntx;
  a.nct = ceil_div(c, 32);
  const long fixed = (long)nslots * a.nct;
  long want = (1024 + fixed - 1) / fixed;
  if (want > a.ntiles) want = a.ntiles;
  if (want < 1) want = 1;
  if (repmode_deterministic() && want > repmode_det_cap(RM_DET_WGRAD)) want = repmode_det_cap(RM_DET_WGRAD);      // (deterministic: at most two addends per element of the cleared dw)
  a.tiles_per_block = ceil_div(a.ntiles, (int)want);
  a.nchunks = ceil_div(a.ntiles, a.tiles_per_block);
  if (!(flip & 2)) RM_HIP(hipMemsetAsync(dw, 0, (size_t)nslots * REPMODE_TAPS * c * sizeof(float), s));   // flip bit 1: dw already cleared
  repmode_prof_begin(REPMODE_PROF_WGRAD_THIN, 2.0 * n * d * h * wdim * (double)c * REPMODE_TAPS, s);
  hipLaunchKernelGGL(conv5_wgrad_thin_kernel, dim3((unsigned)(fixed * a.nchunks)), dim3(256), 0, s, a);
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("conv5_wgrad_thin");
  return REPMODE_OK;
}
