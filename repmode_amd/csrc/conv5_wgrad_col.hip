// conv5_wgrad_col.hip -- the filter gradient of the merged 5x5x5 convolution (autograd of fnet/nn_modules/RepMode.py:207,
// aten::convolution_backward weight grad) as a COLUMN WALK: one workgroup owns ALL 125 taps of a (slot, 16 co, 16 ci) tile
// and walks a column of output tiles along z, so that x and dy cross HBM / L2 -> LDS once per z step instead of once per
// (z step, dz plane) as in conv5_wgrad.hip's units (one dz plane each: VERDICT round 5, item 1).
//
//     dw[slot][tap][co][ci] = sum_{n in slot} sum_v dy[n][v][co] * x[n][v + tap][ci]
//
// * Unit = (slot, 16 output channels, 16 input channels): 125 accumulator tiles of v_mfma_f32_16x16x32_bf16, split over the
//   four MFMA waves of the workgroup by contiguous tap ranges (32 / 31 / 31 / 31 tiles = 128 / 124 registers).
// * A column is the (TY x TX) output tile at one (y, x) position of one sample for every z.  Step z of a column multiplies
//   dy plane z with the x planes z-2 .. z+2: the x planes live in a RING of six halo planes in LDS (channel-major, the
//   layout of conv5_wgrad.hip's xT: a tap shift along x is a register window, along y an immediate offset, along z another
//   ring slot), the dy plane in one of two buffers.  Per step the four LOADER waves fetch and transpose ONE new x plane
//   (16 channels) and ONE dy plane (16 channels) -- 22 KB for 1000 MFMAs; conv5_wgrad.hip's unit stages 43.6 KB for 800 --
//   while the MFMA waves multiply; one barrier per step.  Columns follow each other without a bubble: the planes of the next
//   column enter the ring as the last steps of the current one free its slots.
// * A launch is ONE sequence of steps -- slots, units of a slot, samples of the slot, columns, z -- and every workgroup of a
//   persistent grid (one per CU) takes an equal range of it (stream-K, as conv5_wgrad.hip's round-4 form).  The MFMA waves
//   flush at the end of a unit: 16-byte plain stores when the whole unit ran in this workgroup (the operands are swapped --
//   A = x window, B = dy -- so that a lane holds four consecutive ci of one co), float atomics onto the cleared dw otherwise.
//   With enough units to fill the chip a workgroup takes WHOLE units (an equal range of the unit list): every flush is plain
//   stores and dw needs no clearing (level 2 at batch 8: 512 units on 256 workgroups, the first unit's stores drain under the
//   second unit's MFMAs).
// * z planes outside the volume: the dz reads a plane of zeros kept beside the ring (straight-line MFMA code: a second,
//   branching body for the edge steps cost 200-700 spilled registers; with contiguous tap ranges the busiest wave of a step
//   has all its planes inside the volume anyway, so skipping would not shorten a step).
// * The per-expert levels (volumes <= 8 voxels wide: levels 3-4) take the same walk as
//   repmode_conv5_wgrad_dual's launch: BOTH conv experts' filter gradients of a block from one staging of x -- the 125 taps
//   of the 5x5x5 expert against its gate-scaled output gradient and the 27 centre taps of the 3x3x3 expert against the other
//   (152 accumulator tiles, 38 per wave; the 3x3x3 job computes 27 taps, not 125) --, SEVERAL samples per step (the tile is
//   2 samples x 8 x 8 or 4 samples x 4 x 8 voxels: a K extent of 128 per barrier instead of 64 / 32), and the experts' own
//   [co][ci][125] / [co][ci][27] layouts written as 150-byte runs through an LDS transpose.  Whole units per workgroup only.
#include "wgrad_col.h"

#include <cstdlib>
#include <type_traits>

namespace {

int g_wgrad_col = []() { const char* e = getenv("REPMODE_WGRAD_COL"); return e ? atoi(e) : 1; }();

template <int TY_, int TX_, int SPS_ = 1, int RING_ = 6, bool DUAL_ = false>
struct ColTile {
  static constexpr int TY = TY_, TX = TX_, SPS = SPS_;   // SPS: samples per step (their tiles side by side in K)
  static constexpr bool DUAL = DUAL_;                // two output gradients (5x5x5 + 3x3x3 experts), the experts' layouts
  static constexpr int TV = SPS * TY * TX;           // voxels of a step's tile = its K extent
  static constexpr int HY = TY + 4;
  static constexpr int NGX = TX / 8;                 // 8-voxel groups per row
  static constexpr int GPR = 4 / NGX;                // tile rows per K step (32 voxels)
  static constexpr int RG = (TX + 4 + 7) / 8;        // 16-byte slots per stored halo row (x0-2 .. x0+TX+1)
  // TX = 8: a lane's 16-lane quarter is a ROW (32 bytes further), not the next x group (16 bytes): the two slots of a halo row
  // are swapped on odd channel rows, as in conv5_wgrad.hip's 8-wide tiles -- without it every window read took 8 LDS cycles
  // instead of 4 (tools/lds_bank_check.py's model; measured: profiles/r06_wgrad_col.txt)
  static constexpr bool SWZ = TX < 16;
  static constexpr int KSTEPS = TV / 32;
  static constexpr int NWROW = TY - GPR + 5;         // distinct compile-time halo-row offsets of the windows of a sample plane
  static constexpr int NPAIR = TX / 2 + 2;           // x pairs of a halo row
  static constexpr int SPLANE = HY * RG * 16;        // bytes of one sample's halo plane of one channel
  static constexpr int PLANE = SPS * SPLANE;         // ... of a step's SPS samples
  // ring slots: a step reads planes z-2 .. z+2 while the plane three ahead is written (6); volumes of at most two planes get
  // by with 4 (a column has two planes: two being read, the next column's two arriving)
  static constexpr int RING = RING_;
  // (slot RING of a channel row is a plane of zeros: what a dz whose plane lies outside the volume reads)
  // channel rows an odd multiple of 32 bytes apart: conflict-free ds_read_b128 (conv5_wgrad.hip, tools/lds_bank_check.py)
  static constexpr int ROW_C = (RING + 1) * PLANE + (32 - ((RING + 1) * PLANE) % 64 + 64) % 64;
  static constexpr int DYS = TV * 2 + 32;
  static constexpr int DYJOB = 16 * DYS;             // one job's dy tile
  static constexpr int NJOB = DUAL ? 2 : 1;
  static constexpr int DYBUF = NJOB * DYJOB;
  static constexpr int NTAPS = DUAL ? 152 : 125;     // accumulator tiles of a unit
  static constexpr int WLS = ((NTAPS + 3) / 4) | 1;  // floats per (co, ci) pair of a wave's transposition buffer (odd: no bank conflicts)
  static constexpr int WLBUF = DUAL ? 4 * 64 * WLS * 4 : 0;
  static constexpr int LDS = 16 * ROW_C + 2 * DYBUF + WLBUF;
  static constexpr int NIT_X = SPS * HY * NPAIR * 2; // x items of a plane: (sample, halo row, x pair, channel group of 8)
  static constexpr int NX = (NIT_X + 255) / 256;
  static constexpr int NIT_DY = SPS * (TY * TX / 2) * 2;   // dy items: (sample, voxel pair, channel group of 8)
  static constexpr int NDY = (NIT_DY + 255) / 256;
  static_assert(TV % 32 == 0 && TX % 8 == 0 && TX <= 32 && NGX * GPR == 4 && RG >= NGX + 1, "tile shape");
  static_assert(SPS == 1 || (TY % GPR == 0), "a K step must not straddle two samples");
  static_assert(ROW_C % 64 == 32 && DYS % 64 == 32, "channel rows an odd multiple of 32 bytes apart");
  static_assert(LDS <= 160 * 1024, "LDS");
};

// The accumulator tiles of MFMA wave ROLE: [T0, T1) of the unit's list -- the 125 taps of the 5x5x5 filter (tap = dz * 25 +
// dy * 5 + dx; job A), then, in the dual form, the 27 centre taps of the 3x3x3 expert (job B: its own dy operand) -- and which
// windows feed them.  Window w = (dz, sample j, rr): the 12-element register window of halo row rr (+ the lane's own row) of
// sample j's plane dz; it feeds K step ks (of sample j) for the tap row dy = rr - (first row of the step).
// Tap split (Q, part): the unit's list is cut into Q parts, one WORKGROUP each -- every part walks the unit's whole K with 1 / Q
// of the accumulators, so a unit that is too long for one workgroup is divided without sharing any output element (no
// atomics, no cleared dw), at the price of staging x and dy Q times.
template <typename G, int ROLE, int Q = 1, int PART = 0>
struct RolePlan {
  static constexpr int NTOT = G::NTAPS;
  static constexpr int T0 = NTOT * (PART * 4 + ROLE) / (4 * Q);
  static constexpr int T1 = NTOT * (PART * 4 + ROLE + 1) / (4 * Q);
  static constexpr int NT = T1 - T0;
  static constexpr int NW = 5 * G::SPS * G::NWROW;
  // index of tap (dz, dyi, dxi) of job `b` in the unit's list, -1: not a tap of that job
  static constexpr int index(int b, int dz, int dyi, int dxi) {
    if (dyi < 0 || dyi >= 5) return -1;
    if (b == 0) return dz * 25 + dyi * 5 + dxi;
    if (!G::DUAL || dz < 1 || dz > 3 || dyi < 1 || dyi > 3 || dxi < 1 || dxi > 3) return -1;
    return 125 + ((dz - 1) * 3 + (dyi - 1)) * 3 + (dxi - 1);
  }
  static constexpr bool mine(int b, int dz, int dyi, int dxi) {
    const int u = index(b, dz, dyi, dxi);
    return u >= T0 && u < T1;
  }
  static constexpr bool row_mine(int dz, int dyi) {
    for (int b = 0; b < G::NJOB; ++b)
      for (int dxi = 0; dxi < 5; ++dxi)
        if (mine(b, dz, dyi, dxi)) return true;
    return false;
  }
  static constexpr int ks_sample(int ks) { return (ks * G::GPR) / G::TY; }
  static constexpr int ks_row0(int ks) { return (ks * G::GPR) % G::TY; }
  static constexpr bool nonempty(int w) {
    if (w < 0 || w >= NW) return false;
    const int dz = w / (G::SPS * G::NWROW), j = (w / G::NWROW) % G::SPS, rr = w % G::NWROW;
    for (int ks = 0; ks < G::KSTEPS; ++ks)
      if (ks_sample(ks) == j && row_mine(dz, rr - ks_row0(ks))) return true;
    return false;
  }
  static constexpr int next(int w) {                   // the next non-empty window behind w, -1: none
    for (int v = w + 1; v < NW; ++v)
      if (nonempty(v)) return v;
    return -1;
  }
};

struct ColArgs {
  const bf16_t* x;
  const bf16_t* dy;              // job A's output gradient
  const bf16_t* dy2;             // dual form: job B's
  const int32_t* sample_slot;    // NULL: every sample in slot 0
  float* dw;                     // slot layout [nslots][125][Cout][CinTot]; dual form: the 5x5x5 expert's [Cout][Cin][125]
  float* dw2;                    // dual form: the 3x3x3 expert's [Cout][Cin][27]
  int N, D, H, W, Cin, Cout, CinTot, ci_off, nslots;
  int ncot, ncit, nty, ntx, ncol;
  long total;                    // steps of the launch (one sample per slot group of SPS)
  int aligned;                   // 1: a workgroup takes WHOLE units (an equal range of the unit list): plain stores only
};

// where a walk stands in the launch's sequence (all wave-uniform); k: the group of SPS samples of the slot
struct ColCursor {
  int slot, cot, cit, q, k, cnt, ngrp, col;         // q: the unit's tap part (0 .. Q-1)
  unsigned long long mask;
};

// two x-adjacent voxels (8 channels each) -> eight 4-byte stores, one per channel row
// (odd_delta: what the odd channel rows' position differs by -- the swapped slots of the 8-wide tiles)
__device__ __forceinline__ void put8(unsigned char* dst, int stride, const u32x4& v0, const u32x4& v1, int odd_delta = 0) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    *reinterpret_cast<uint32_t*>(dst + (2 * j) * stride) = __builtin_amdgcn_perm(v1[j], v0[j], 0x05040100u);
    *reinterpret_cast<uint32_t*>(dst + (2 * j + 1) * stride + odd_delta) = __builtin_amdgcn_perm(v1[j], v0[j], 0x07060302u);
  }
}

template <int TY, int TX, int SPS, int RING, bool DUAL, int Q>
__global__ __launch_bounds__(512, 2) void conv5_wgrad_col_kernel(ColArgs a) {
  using G = ColTile<TY, TX, SPS, RING, DUAL>;
  constexpr int NGX = G::NGX, GPR = G::GPR, RG = G::RG, KSTEPS = G::KSTEPS, NWROW = G::NWROW, NPAIR = G::NPAIR;
  constexpr int PLANE = G::PLANE, SPLANE = G::SPLANE, ROW_C = G::ROW_C, DYS = G::DYS, DYBUF = G::DYBUF, DYJOB = G::DYJOB;
  constexpr int NX = G::NX, NDY = G::NDY, NJOB = G::NJOB, HY = G::HY;
  __shared__ __attribute__((aligned(16))) unsigned char smem[G::LDS];
  unsigned char* xT = smem;                            // [16 ci][RING + 1 planes][SPS][HY][RG * 16 B]
  unsigned char* dyT = smem + 16 * ROW_C;              // [2][NJOB][16 co][TV] bf16

  const int tid = (int)(threadIdx.x & 255), lane = tid & 63, wave = tid >> 6;
  const bool loader = threadIdx.x >= 256;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout;

  const int slot64 = a.sample_slot ? a.sample_slot[min(lane, a.N - 1)] : 0;     // (the host sends at most 64 samples here)
  auto mask_of = [&](int sl) -> unsigned long long { return __ballot(lane < a.N && slot64 == sl); };
  auto kth = [&](unsigned long long m, int kk) -> int {             // the kk-th sample of a slot, -1: it has fewer
    for (int i = 0; i < kk; ++i) m &= m - 1;
    return __ffsll((long long)m) - 1;
  };
  auto groups = [&](int cnt) -> int { return (cnt + SPS - 1) / SPS; };
  const int G2 = a.ncot * a.ncit * Q, colsteps = a.ncol * D;        // units of a slot: (cot, cit, tap part)
  const long wg = xcd_remap(blockIdx.x, gridDim.x);    // neighbouring ranges (the same samples' planes) on one XCD
  long g0, g1;
  if (a.aligned) {
    // whole units: units [U wg / G, U (wg + 1) / G) of the list (slot, cot, cit); a unit of slot s has groups(s) * ncol * D steps
    const long units = (long)a.nslots * G2;
    auto steps_before = [&](long u) -> long {
      const int su = (int)(u / G2);
      long acc = 0;
      for (int sl = 0; sl < su; ++sl) acc += (long)groups(__popcll(mask_of(sl))) * colsteps * G2;
      if (su < a.nslots) acc += (u % G2) * (long)groups(__popcll(mask_of(su))) * colsteps;
      return acc;
    };
    g0 = steps_before(units * wg / gridDim.x);
    g1 = steps_before(units * (wg + 1) / gridDim.x);
  } else {
    g0 = a.total * wg / gridDim.x;
    g1 = a.total * (wg + 1) / gridDim.x;
  }
  const int steps = (int)(g1 - g0);
  if (steps <= 0) return;
  ColCursor c0;
  int zb;
  {
    long rem = g0;
    int sl = 0, cnt;
    unsigned long long m;
    for (;;) {
      m = mask_of(sl); cnt = __popcll(m);
      const long span = (long)groups(cnt) * colsteps * G2;
      if (rem < span) break;
      rem -= span; ++sl;
    }
    const long per = (long)groups(cnt) * colsteps;
    const int unit = (int)(rem / per);
    rem -= (long)unit * per;
    c0.slot = sl; c0.mask = m; c0.cnt = cnt; c0.ngrp = groups(cnt);
    c0.q = unit % Q; c0.cit = (unit / Q) % a.ncit; c0.cot = unit / (Q * a.ncit);
    c0.k = (int)(rem / colsteps);
    rem -= (long)c0.k * colsteps;
    c0.col = (int)(rem / D);
    zb = (int)(rem % D);
  }
  auto next_col = [&](ColCursor& c) {                  // (never called behind the launch's last column)
    if (++c.col < a.ncol) return;
    c.col = 0;
    if (++c.k < c.ngrp) return;
    c.k = 0;
    if (++c.q < Q) return;
    c.q = 0;
    if (++c.cit == a.ncit) {
      c.cit = 0;
      if (++c.cot == a.ncot) {
        c.cot = 0;
        do { ++c.slot; c.mask = mask_of(c.slot); c.cnt = __popcll(c.mask); } while (c.cnt == 0 && c.slot < a.nslots);
        c.ngrp = groups(c.cnt);
      }
    }
  };

  if (loader) {
    // ---- loader waves.  The stream of ELEMENTS of this workgroup's range: per column the planes p = 0 .. D + 1 (the first
    // column from max(0, zb - 2)); element p = x plane p (p < D) + dy plane p - 2 (p >= 2), of the column's SPS samples.  Step
    // (column, z) needs the elements up to z + 2.  An element is fetched two elements ahead of its transposition into LDS
    // (register sets a / b) and staged as early as the ring allows: x plane V = column * D + p goes to ring slot V % RING once
    // the step the MFMA waves are at no longer reads plane V - RING (V <= lowest plane of that step + RING - 1), a dy plane
    // at most one step ahead.
    constexpr uint32_t OOB = 0x80000000u;
    struct ElemRegs { u32x4 x0[NX], x1[NX], d0[NJOB][NDY], d1[NJOB][NDY]; };
    struct ElemMeta { int hasx, hasdy, v, dystep; };
    ElemRegs ra, rb;
    ElemMeta ma{}, mb{};
    // this thread's items
    int x_hy[NX], x_px[NX], x_cg8[NX], x_dst[NX], x_j[NX], x_odd[NX];
    bool x_on[NX];
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int it = u * 256 + tid;
      const int pr = it % NPAIR;
      int r = it / NPAIR;
      const int cg = r & 1; r >>= 1;
      const int hy = r % HY, j = r / HY;
      x_on[u] = it < G::NIT_X;
      x_hy[u] = hy - 2; x_px[u] = 2 * pr - 2; x_cg8[u] = cg * 8; x_j[u] = j;
      x_dst[u] = (cg * 8) * ROW_C + j * SPLANE + (hy * RG + (pr >> 2)) * 16 + (pr & 3) * 4;
      x_odd[u] = G::SWZ ? (((pr >> 2) ^ 1) - (pr >> 2)) * 16 : 0;
    }
    int d_yy[NDY], d_xx[NDY], d_cg8[NDY], d_dst[NDY], d_j[NDY];
    bool d_on[NDY];
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
      const int it = u * 256 + tid;
      const int q = it % (TY * TX / 2);
      int r = it / (TY * TX / 2);
      const int j = r % SPS, cg = r / SPS;
      d_on[u] = it < G::NIT_DY;
      d_xx[u] = (2 * q) % TX; d_yy[u] = (2 * q) / TX; d_cg8[u] = cg * 8; d_j[u] = j;
      d_dst[u] = (cg * 8) * DYS + (j * (TY * TX / 2) + q) * 4;
    }
    // one descriptor over the whole tensor (the host checks N * D * H * W * C * 2 < 2^31); a sample is an offset
    const uint32_t xvol = (uint32_t)((size_t)D * H * W * Cin * 2), dyvol = (uint32_t)((size_t)D * H * W * Cout * 2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)(xvol * (uint32_t)a.N), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dy), 0, (int)(dyvol * (uint32_t)a.N), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(DUAL ? a.dy2 : a.dy), 0, (int)(dyvol * (uint32_t)a.N), 0x00020000);

    // fetch cursor = the next element of the stream
    ColCursor fc = c0;
    int fp = max(0, zb - 2), fzs = zb, ebase = -zb, fcoD = 0;
    auto elem_needed = [&]() -> bool { return ebase + max(fzs, fp - 2) <= steps - 1; };
    auto elem_advance = [&]() {
      if (++fp <= D + 1) return;
      fp = 0; ebase += D; fzs = 0; fcoD += D;
      if (ebase <= steps - 1) next_col(fc);            // (a column behind the range is never decoded)
    };
    auto fetch_to = [&](ElemRegs& r, ElemMeta& m) {
      m.hasx = fp < D;
      m.hasdy = fp >= 2 && fp - 2 >= fzs;
      m.v = fcoD + fp;
      m.dystep = ebase + fp - 2;
      const int y0 = (fc.col / a.ntx) * TY, x0 = (fc.col % a.ntx) * TX;
      int ns[SPS];                                        // the column's samples (-1: the slot has no such sample)
#pragma unroll
      for (int j = 0; j < SPS; ++j) ns[j] = fc.k * SPS + j < fc.cnt ? kth(fc.mask, fc.k * SPS + j) : -1;
      auto sample_of = [&](int j) -> int {
        int n = ns[0];
#pragma unroll
        for (int q = 1; q < SPS; ++q) n = j == q ? ns[q] : n;
        return n;
      };
      if (m.hasx) {
#pragma unroll
        for (int u = 0; u < NX; ++u) {
          const int n = sample_of(x_j[u]);
          const int gy = y0 + x_hy[u], gx = x0 + x_px[u], c = fc.cit * 16 + x_cg8[u];
          const bool row_ok = x_on[u] && n >= 0 && (unsigned)gy < (unsigned)H && c < Cin;
          const uint32_t off = (uint32_t)n * xvol + (uint32_t)((((fp * H + gy) * W + gx) * Cin + c) * 2);
          r.x0[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                      rx, (row_ok && (unsigned)gx < (unsigned)W) ? off : OOB, 0, 0));
          r.x1[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                      rx, (row_ok && (unsigned)(gx + 1) < (unsigned)W) ? off + (uint32_t)Cin * 2 : OOB, 0, 0));
        }
      }
      if (m.hasdy) {
        const int z = fp - 2;
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
          const int n = sample_of(d_j[u]);
          const int gy = y0 + d_yy[u], gx = x0 + d_xx[u], c = fc.cot * 16 + d_cg8[u];
          const bool row_ok = d_on[u] && n >= 0 && gy < H && c < Cout;
          const uint32_t off = (uint32_t)n * dyvol + (uint32_t)((((z * H + gy) * W + gx) * Cout + c) * 2);
          const uint32_t o0 = (row_ok && gx < W) ? off : OOB, o1 = (row_ok && gx + 1 < W) ? off + (uint32_t)Cout * 2 : OOB;
          r.d0[0][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, o0, 0, 0));
          r.d1[0][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy, o1, 0, 0));
          if constexpr (DUAL) {
            r.d0[1][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy2, o0, 0, 0));
            r.d1[1][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy2, o1, 0, 0));
          }
        }
      }
    };
    auto stage_from = [&](ElemRegs& r, const ElemMeta& m) {
      if (m.hasx) {
        unsigned char* base = xT + (m.v % RING) * PLANE;
#pragma unroll
        for (int u = 0; u < NX; ++u)
          if (x_on[u]) put8(base + x_dst[u], ROW_C, r.x0[u], r.x1[u], x_odd[u]);
      }
      if (m.hasdy) {
        unsigned char* base = dyT + (m.dystep & 1) * DYBUF;
#pragma unroll
        for (int b = 0; b < NJOB; ++b)
#pragma unroll
          for (int u = 0; u < NDY; ++u)
            if (d_on[u]) put8(base + b * DYJOB + d_dst[u], DYS, r.d0[b][u], r.d1[b][u]);
      }
    };
    // the step the MFMA waves are at during interval i (i = -1: the prologue, nobody reads yet -- same bound as step 0)
    int scoD = 0, sz = zb, szs = max(0, zb - 2);
    auto allowed = [&](const ElemMeta& m, int i) -> bool {
      const int lo = scoD + max(sz - 2, szs);
      return (!m.hasx || m.v <= lo + RING - 1) && (!m.hasdy || m.dystep <= i + 1);
    };
    for (int i = tid; i < 16 * (PLANE / 16); i += 256)           // the plane of zeros (slot RING of every channel row)
      *reinterpret_cast<u32x4*>(xT + (i / (PLANE / 16)) * ROW_C + RING * PLANE + (i % (PLANE / 16)) * 16) = u32x4{0u, 0u, 0u, 0u};
    bool ha = elem_needed();
    if (ha) { fetch_to(ra, ma); elem_advance(); }
    bool hb = elem_needed();
    if (hb) { fetch_to(rb, mb); elem_advance(); }
    int cur = 0;
    for (int i = -1; i < steps - 1; ++i) {
      for (;;) {
        if (cur == 0) {
          if (!ha || !allowed(ma, i)) break;
          stage_from(ra, ma);
          ha = elem_needed();
          if (ha) { fetch_to(ra, ma); elem_advance(); }
          cur = 1;
        } else {
          if (!hb || !allowed(mb, i)) break;
          stage_from(rb, mb);
          hb = elem_needed();
          if (hb) { fetch_to(rb, mb); elem_advance(); }
          cur = 0;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (i >= 0 && ++sz == D) { sz = 0; scoD += D; szs = 0; }
    }
    return;
  }

  // ---- MFMA waves: wave w owns the accumulator tiles of RolePlan<G, w, Q, part> of the unit it is at.  One call of run_unit
  // = the steps of ONE unit that fall into this workgroup's range (the tap lists are compile-time: another part is another body)
  struct StepState { int i, z, m6; bool head; ColCursor sc; };
  const int l15 = lane & 15, kg = lane >> 4;
  const unsigned char* xlane = xT + l15 * ROW_C + ((kg / NGX) * RG + kg % NGX) * 16;
  const unsigned char* dlane = dyT + l15 * DYS + kg * 16;
  auto run_unit = [&](auto PART, auto ROLE, StepState& st) {
    using P = RolePlan<G, decltype(ROLE)::value, Q, decltype(PART)::value>;
    f32x4 acc[P::NT];
#pragma unroll
    for (int t = 0; t < P::NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (;;) {
      const int i = st.i, z = st.z;
      asm volatile("s_barrier" ::: "memory");
      const unsigned char* db = dlane + (i & 1) * DYBUF;
      bf16x8 bfr[NJOB][KSTEPS];                          // B operands: dy[co = lane & 15][8 voxels of group kg], per job
#pragma unroll
      for (int b = 0; b < NJOB; ++b)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
          bfr[b][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(db + b * DYJOB + ks * 64));
      const unsigned char* xs[5];
#pragma unroll
      for (int dz = 0; dz < 5; ++dz) {
        int sl = st.m6 + dz + 2 * RING - 2;              // plane z + dz - 2 -> ring slot (V + dz - 2) % RING
        sl = sl % RING;
        if ((unsigned)(z + dz - 2) >= (unsigned)D) sl = RING;        // outside the volume: the plane of zeros
        xs[dz] = xlane + sl * PLANE;
      }
      auto read_window = [&](auto WI, u32x4& lo, u32x2& hi) {
        constexpr int w = decltype(WI)::value;
        const unsigned char* xb = xs[w / (SPS * NWROW)] + ((w / NWROW) % SPS) * SPLANE + (w % NWROW) * RG * 16;
        lo = *reinterpret_cast<const u32x4*>(G::SWZ ? xb + (l15 & 1) * 16 : xb);               // elements 0..7 of the window (x0 + 8g - 2 ..)
        hi = *reinterpret_cast<const u32x2*>(G::SWZ ? xb + 16 - (l15 & 1) * 16 : xb + 16);    // elements 8..11
      };
      {
        u32x4 lo_n;
        u32x2 hi_n;
        constexpr int wfirst = P::next(-1);
        read_window(std::integral_constant<int, wfirst>{}, lo_n, hi_n);
        static_for<0, P::NW>([&](auto WI) {
          constexpr int w = decltype(WI)::value;
          constexpr int dz = w / (SPS * NWROW), wj = (w / NWROW) % SPS, rr = w % NWROW;
          if constexpr (P::nonempty(w)) {
            u32x4 lo = lo_n;
            u32x2 hi = hi_n;
            asm volatile("" : "+v"(lo), "+v"(hi));     // a register value from here on (no re-reads from LDS)
            constexpr int wn = P::next(w);
            if constexpr (wn >= 0) read_window(std::integral_constant<int, wn>{}, lo_n, hi_n);
            __builtin_amdgcn_sched_barrier(0);         // keep the request ahead of this window's MFMAs
            const uint32_t a10 = __builtin_amdgcn_alignbit(lo.y, lo.x, 16);
            const uint32_t a21 = __builtin_amdgcn_alignbit(lo.z, lo.y, 16);
            const uint32_t a32 = __builtin_amdgcn_alignbit(lo.w, lo.z, 16);
            const uint32_t a43 = __builtin_amdgcn_alignbit(hi.x, lo.w, 16);
            const uint32_t a54 = __builtin_amdgcn_alignbit(hi.y, hi.x, 16);
            const bf16x8 sh[5] = {
                __builtin_bit_cast(bf16x8, lo),                                     // shift -2: elements 0..7
                __builtin_bit_cast(bf16x8, (u32x4{a10, a21, a32, a43})),            // shift -1
                __builtin_bit_cast(bf16x8, (u32x4{lo.y, lo.z, lo.w, hi.x})),        // shift  0
                __builtin_bit_cast(bf16x8, (u32x4{a21, a32, a43, a54})),            // shift +1
                __builtin_bit_cast(bf16x8, (u32x4{lo.z, lo.w, hi.x, hi.y}))};       // shift +2
            static_for<0, KSTEPS>([&](auto KS) {
              constexpr int ks = decltype(KS)::value;
              constexpr int dyi = rr - P::ks_row0(ks);
              if constexpr (P::ks_sample(ks) == wj && dyi >= 0 && dyi < 5) {
                static_for<0, 5 * NJOB>([&](auto DX) {
                  constexpr int b = decltype(DX)::value / 5, dxi = decltype(DX)::value % 5;
                  if constexpr (P::mine(b, dz, dyi, dxi)) {
                    constexpr int t = P::index(b, dz, dyi, dxi) - P::T0;
                    // D[ci = 4 (lane >> 4) + r][co = lane & 15] += x window (A) * dy (B)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh[dxi], bfr[b][ks], acc[t], 0, 0, 0);
                  }
                });
              }
            });
          }
        });
      }

      const ColCursor& sc = st.sc;
      const bool last = z == D - 1 && sc.col == a.ncol - 1 && sc.k == sc.ngrp - 1;     // the unit's last step
      const bool done = last || i + 1 == steps;
      if (done) {
        const int co = sc.cot * 16 + l15, ci0 = sc.cit * 16 + kg * 4;
        if constexpr (!DUAL) {
          const bool atomic = !(st.head && last);
          if (co < Cout && ci0 < Cin) {
            float* p = a.dw + (((size_t)sc.slot * REPMODE_TAPS + P::T0) * Cout + co) * a.CinTot + a.ci_off + ci0;
            const size_t tstride = (size_t)Cout * a.CinTot;
            if (atomic) {
#pragma unroll
              for (int t = 0; t < P::NT; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) unsafeAtomicAdd(p + r, acc[t][r]);
                p += tstride;
              }
            } else {
#pragma unroll
              for (int t = 0; t < P::NT; ++t) {
                *reinterpret_cast<f32x4*>(p) = acc[t];
                p += tstride;
              }
            }
          }
        } else {
          // The experts' own layouts ([co][ci][125], [co][ci][27]): a (co, ci) pair's taps are contiguous.  The wave transposes
          // its tiles through LDS, one accumulator register (64 pairs) at a time, and stores runs of NT taps (whole units only:
          // plain stores).
          float* wl = reinterpret_cast<float*>(smem + 16 * ROW_C + 2 * DYBUF) + wave * (64 * G::WLS);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int pair = kg * 16 + l15;
#pragma unroll
            for (int t = 0; t < P::NT; ++t) wl[pair * G::WLS + t] = acc[t][r];
            // (one wave writes and reads its own buffer: LDS operations of a wave complete in order)
            for (int e = lane; e < 64 * P::NT; e += 64) {
              const int pr = e / P::NT, t = e % P::NT, u = P::T0 + t;
              const int oc = sc.cot * 16 + (pr & 15), ic = sc.cit * 16 + (pr >> 4) * 4 + r;
              if (oc < Cout && ic < Cin) {
                float* p = u < REPMODE_TAPS ? a.dw + ((size_t)oc * Cin + ic) * REPMODE_TAPS + u
                                            : a.dw2 + ((size_t)oc * Cin + ic) * 27 + (u - REPMODE_TAPS);
#ifdef COL_NOFLUSH
                if (wl[pr * G::WLS + t] == 12345.678f)      // TIMING BUILD ONLY: sums computed, (practically) never written
#endif
                *p = wl[pr * G::WLS + t];
              }
            }
          }
          (void)co; (void)ci0;
        }
        st.head = true;
      }
      ++st.i;
      if (st.i < steps) {
        if (++st.z == D) { st.z = 0; next_col(st.sc); }
        st.m6 = st.m6 == RING - 1 ? 0 : st.m6 + 1;
      }
      if (done) return;
    }
  };
  StepState st{0, zb, zb % RING, c0.k == 0 && c0.col == 0 && zb == 0, c0};
  while (st.i < steps) {
    const int part = st.sc.q;
    static_for<0, Q>([&](auto PART) {
      if (part == decltype(PART)::value) {
        if (wave == 0) run_unit(PART, std::integral_constant<int, 0>{}, st);
        else if (wave == 1) run_unit(PART, std::integral_constant<int, 1>{}, st);
        else if (wave == 2) run_unit(PART, std::integral_constant<int, 2>{}, st);
        else run_unit(PART, std::integral_constant<int, 3>{}, st);
      }
    });
  }
}

int device_cus() {
  static int tab[32] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  int& c = tab[dev & 31];
  if (!c) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    c = cus;
  }
  return c;
}

long col_room() { return device_cus() - repmode_reserve_cus() > 8 ? device_cus() - repmode_reserve_cus() : 8; }

// whole units per workgroup (plain stores only) when the unit list fills three quarters of the chip
bool col_aligned(const WgColCall& c, int q = 1) {
  const long units = (long)c.nslots * ceil_div(c.Cout, 16) * ceil_div(c.Cin, 16) * q;
  return 4 * units >= 3 * col_room();
}
// REPMODE_WGRAD_COL_Q / repmode_set_wgrad_col_split: 1 (default) no tap split, 2 the taps of a unit over two workgroups.
// Measured (profiles/r06_wgrad_col.txt): level 1 at batch 8 (128 units -> 256 workgroups of whole half-units, plain stores)
// 146 us against the stream-K grid's 115 -- a step is then 62 taps x 8 = 1000 MFMA cycles short of what the loader waves
// need for a plane of x and one of dy (2.2 us: 16-channel slices use 32 bytes of every 128-byte line they pull through the
// L2), and four-way 277 us.  Kept as a tested experiment; never chosen by default.
int g_col_q = []() { const char* e = getenv("REPMODE_WGRAD_COL_Q"); return e ? atoi(e) : 1; }();
int col_tap_split(const WgColCall& c) {
  if (c.dy2 || c.W < 16) return 1;
  return g_col_q == 2 && (c.W >= 32 || c.H > 8) ? 2 : 1;
}

template <int TY, int TX, int SPS, int RING, bool DUAL, int Q = 1>
int launch_col(const WgColCall& c, hipStream_t s) {
  ColArgs a{};
  a.x = static_cast<const bf16_t*>(c.x); a.dy = static_cast<const bf16_t*>(c.dy); a.dy2 = static_cast<const bf16_t*>(c.dy2);
  a.sample_slot = c.sample_slot; a.dw = c.dw; a.dw2 = c.dw2;
  a.N = c.N; a.D = c.D; a.H = c.H; a.W = c.W; a.Cin = c.Cin; a.Cout = c.Cout; a.CinTot = c.CinTot; a.ci_off = c.ci_off;
  a.nslots = c.nslots;
  a.ncot = ceil_div(c.Cout, 16); a.ncit = ceil_div(c.Cin, 16);
  a.nty = ceil_div(c.H, TY); a.ntx = ceil_div(c.W, TX); a.ncol = a.nty * a.ntx;
  const long room = col_room();
  // Enough units to fill three quarters of the chip: a workgroup takes WHOLE units (an equal range of the unit list; the grid
  // is the smallest that gives every workgroup ceil(units / CUs) of them) -- every flush is plain stores, dw needs no clearing,
  // whatever the slots' sample counts.  Otherwise: an equal split of the step sequence; shared units are added with float
  // atomics onto the cleared dw.  The dual form (the experts' layouts) always takes whole units.
  const long units = (long)c.nslots * a.ncot * a.ncit * Q;
  const bool direct = DUAL || Q > 1 || col_aligned(c);
  long g;
  if (direct) {
    const long per = (units + room - 1) / room;
    g = (units + per - 1) / per;
    a.aligned = 1;
  } else {
    // (sample groups: only the whole-unit form knows the slots' counts; SPS == 1 here)
    a.total = (long)c.N * a.ncot * a.ncit * a.ncol * c.D;
    g = room;
    if (g > a.total / 4) g = a.total / 4;
    if (g < 1) g = 1;
  }
  if (c.plan_out) { *c.plan_out = direct ? 1 : 0; return REPMODE_OK; }
  if (!direct && !c.prezeroed)
    RM_HIP(hipMemsetAsync(c.dw, 0, (size_t)c.nslots * REPMODE_TAPS * c.Cout * c.CinTot * sizeof(float), s));
  repmode_prof_begin(REPMODE_PROF_WGRAD, 2.0 * c.N * c.D * c.H * c.W * (double)c.Cin * c.Cout * REPMODE_TAPS, s);
  hipLaunchKernelGGL((conv5_wgrad_col_kernel<TY, TX, SPS, RING, DUAL, Q>), dim3((unsigned)g), dim3(512), 0, s, a);
  return REPMODE_OK;
}

}  // namespace

int repmode_wgrad_col_mode() { return g_wgrad_col; }

extern "C" int repmode_set_wgrad_col(int mode) { g_wgrad_col = mode; return REPMODE_OK; }
extern "C" int repmode_get_wgrad_col(void) { return g_wgrad_col; }
extern "C" int repmode_set_wgrad_col_split(int q) { g_col_q = q; return REPMODE_OK; }
extern "C" int repmode_get_wgrad_col_split(void) { return g_col_q; }

bool repmode_wgrad_col_eligible(const WgColCall& c) {
  if (g_wgrad_col == 0 || repmode_deterministic()) return false;
  if ((c.Cin & 7) || (c.Cout & 7) || c.N > 64) return false;
  if ((size_t)c.N * c.D * c.H * c.W * (c.Cin > c.Cout ? c.Cin : c.Cout) * 2 >= ((size_t)1 << 31)) return false;
  if (c.dy2) {
    // the dual form (both conv experts' gradients of a per-expert block): volumes up to 15 voxels wide
    if (c.W >= 16 || c.nslots != 1) return false;
    // Measured SLOWER than conv5_wgrad.hip's dual launch on every layer of levels 3-4 (profiles/r06_wgrad_col.txt: 81 vs 70 us
    // at 256 -> 256, 187 vs 96 us at level 4's 512 -> 512; the same with the stores taken out): a 16-channel tile uses 32 of a
    // voxel's 512-1024 bytes, so a step's staging is ~1000 L2 requests of 16 useful bytes each and the loaders run at the L2's
    // request rate, not at its bandwidth -- 4-10 us per step against 1.4 us of MFMAs.  Kept as a tested experiment: mode 2 only.
    return g_wgrad_col >= 2;
  }
  if (c.W < 16 || (c.CinTot & 3) || (c.ci_off & 3) || !c.sample_slot) return false;
  if (g_wgrad_col >= 2) return true;
  // mode 1: the shapes it was measured to win on (profiles/r06_wgrad_col.txt) -- level 2 of the network (volumes 16 .. 31
  // voxels wide) with enough units for whole-unit workgroups; where units are shared through float atomics (levels 0-1 at
  // batch 8: 8 and 2-4 workgroups per unit) conv5_wgrad.hip's stream-K grid is faster
  return c.W < 32 && col_aligned(c, col_tap_split(c));
}

int repmode_wgrad_col_launch(const WgColCall& c, hipStream_t s) {
  if (c.dy2) {
    if (c.H <= 4 && c.D <= 2) return launch_col<4, 8, 4, 4, true>(c, s);
    return launch_col<8, 8, 2, 6, true>(c, s);
  }
  const int q = col_tap_split(c);
  if (c.W >= 32) {
    if (q == 2) return launch_col<8, 32, 1, 6, false, 2>(c, s);
    return launch_col<8, 32, 1, 6, false>(c, s);
  }
  if (c.H > 8) {
    if (q >= 2) return launch_col<16, 16, 1, 6, false, 2>(c, s);
    return launch_col<16, 16, 1, 6, false>(c, s);
  }
  return launch_col<8, 16, 1, 6, false>(c, s);
}
