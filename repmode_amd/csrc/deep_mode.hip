// deep_mode.hip -- a whole MoDE block of the deep U-Net levels in the per-expert formulation as ONE launch per direction.
//
// RepMode.py:171-192 + :204-208 by linearity (property 3):
//     forward        P_e[n] = conv(x[n], K_e), e = 0..4;   y[n] = sum_e g[n,e,:] * P_e[n]
//     data gradient  dx[n]  = sum_e conv(G_e[n], flip(K_e)^T),   G_e = g[n,e,:] * dy[n]
// with K_0 the 5x5x5 expert, K_1 the 3x3x3 expert and K_2..4 the three 1x1 experts (conv1x1, avg3 o 1x1, avg5 o 1x1: GEMMs
// on x, box3(x) / 27 and box5(x) / 125, RepMode.py:139-142, 176-180).  Round 4 ran this as five launches per direction:
// conv5_deep / the dual-expert launch (the two conv experts, input-channel slices added with float atomics: 4.5 x the
// algorithmic HBM traffic, 35-50 % of a launch in its epilogue), box_expand, gemm3 (the 1x1 experts), expert_mix (the gate
// mix), and box_sum on the way back.  Here
//   * a workgroup owns a 64-voxel x 32-channel output tile and the WHOLE reduction (every input channel, every expert):
//     its 4 / 8 waves split the 16-channel chunks of the reduction among them, each wave a private halo image in LDS
//     (no barrier in the main loop: two waves per SIMD cover each other's staging), and the partial sums meet ONCE, in
//     LDS (ds_add_f32), before plain stores -- no float atomics in HBM unless the grid would not fill the chip (level 4:
//     the remaining slices add into zeroed outputs);
//   * the five experts keep separate accumulators (forward) and the gate mix + the stores of P_e (kept for the gate
//     gradient <dy, P_e>) are the epilogue; in the data gradient all five add into one accumulator;
//   * the 1x1 experts are three more "taps" per chunk on operands the box kernels prepared (float, rounded to bf16 here
//     exactly as gemm3 rounded them);
//   * the output-channel tile picks the XCD (class c -> XCD c % 8): a filter byte crosses the fabric into ONE L2;
//   * a tile is one z plane (level 3: 8 x 8) or two whole samples (level 4: 2 x 4 x 4 each): tap planes that are padding
//     for the whole tile are skipped (30 % of the 5x5x5 taps on a 4-plane volume), and only in-volume voxels are staged
//     (the image's halo border is zeroed once);
//   * lane -> voxel maps that make every ds_read_b128 of a voxel fragment conflict-free on the (BX + 4)-slot row pitch
//     (MI355X_MICROARCH.md: 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}).
// MFMA: v_mfma_f32_32x32x16_bf16, A = 32 voxels x 16 reduction channels (LDS), B = 32 output channels x 16 (filter
// fragment, 1 KiB contiguous, straight from L2), two voxel sub-tiles per filter fragment.  bf16 operands, float accumulate.
#include "common.h"
#include "tail_jobs.h"

#include <atomic>
#include <cstdlib>

#ifndef DM_PF
#define DM_PF 1      // tap rows of filter fragments in flight ahead of the MFMAs
#endif

namespace {

constexpr int E = REPMODE_NUM_EXPERTS;

struct DmArgs {
  const bf16_t* x0;      // forward: x [N][D][H][W][R];  data gradient: G_0 (the 5x5x5 expert's gate-scaled dy)
  const bf16_t* x1;      // data gradient: G_1 (the 3x3x3 expert's); forward: unused
  const bf16_t* w;       // [2][125][OP/32][RP/16][32][16] (repmode_expert_frags: wf forward, wd data gradient)
  const float* s[3];     // operands of the 1x1 experts, float [N][V][R]: forward x, box3(x)/27, box5(x)/125;
                         // data gradient G_2, box3(G_3)/27, box5(G_4)/125
  const float* k[3];     // K1, A3, A5: float [Co][Ci]
  long kso, ksr;         // element (o, r) of k[e] lives at o * kso + r * ksr
  const float* gate;     // forward: g [N][5][O]
  float* p;              // forward: P [5][N][V][O]
  void* y;               // forward: y float [N][V][O];  data gradient: dx [N][V][O] float or bf16
  int y_bf16;
  float* stats;          // forward, plain stores only: per-channel sum / sum of squares of y go to this half of the library's
  float* stats_clear;    //   BatchNorm scratch (16 slices x 2 x O), and the other half is put back to zero (as bn_stats_kernel does)
  int N, D, H, W, R, O, RP, OP;
  int nbz, nunits, G, ncot, ksplit, xcd_classes;
  TailJobs tail;         // deferred small jobs riding in this launch (tail_jobs.h)
};

// BZ x BY x BX bricks ("units"), SU units per 64-voxel tile
template <int BZ_, int BY_, int BX_, int SU_>
struct MCfg {
  static constexpr int BZ = BZ_, BY = BY_, BX = BX_, SU = SU_;
  static constexpr int NV = BZ * BY * BX;
  static constexpr int TU = NV / 32;                          // 32-voxel sub-tiles of a unit
  static constexpr int VW = SU * TU;                          // sub-tiles of the tile = sub-tiles per wave
  static constexpr int NP = 4;                                // z planes of a halo image
  static constexpr int PB = BZ > 1 ? -1 : 0;                  // volume plane of image plane 0
  static constexpr int MAXD = BZ > 1 ? 2 : 4;                 // volume depth the four planes cover
  static constexpr int BYH = BY + 4, BXH = BX + 4;
  static constexpr int PP = BYH * BXH + (BZ > 1 ? 4 : 0);     // plane pitch in 16-byte slots (two-plane sub-tiles: == 4 mod 16)
  static constexpr int PLS = ((NP * PP + 7) / 8) * 8 + 4;     // channel-group plane stride (== 4 mod 8: conflict-free 16-byte stores)
  static constexpr int IMG = 2 * PLS;                         // one unit's image: two channel groups of 8
  static constexpr int WSLOTS = SU * IMG;                     // a wave's private region
  static constexpr int MAXI = 8;                              // staged 16-byte items per lane and image
  static constexpr int NVOX = 32 * VW;                        // voxels of a tile
  static_assert((VW == 1 || VW == 2) && NV % 32 == 0, "a tile is one or two 32-voxel sub-tiles");
};

__device__ __forceinline__ void mma_bf16(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Row m (0..31) of a sub-tile -> its voxel inside the unit.  ds_read_b128 serves a half-wave as the lane groups
// A = {0-3, 12-15, 20-27} and B = {4-11, 16-19, 28-31}; the 16 voxels of a group must sit in 16 different slots mod 16:
//   8 x 8 plane, row pitch 12:  group A = rows 0, 2 (slots 0-7, 24-31), group B = rows 1, 3 (12-19, 36-43)
//   2 x 4 x 4,  row pitch 8, plane pitch 68:  group = y / 2, inside it (x, y & 1, z) -> x + 8 (y & 1) + 4 z
template <typename C>
__device__ __forceinline__ void row_voxel(int m, int piece, int& lz, int& ly, int& lx) {
  const int g = (0xF00F0FF0u >> m) & 1;
  const int i = m - ((m >= 4 ? 4 : 0) + (m >= 12 ? 4 : 0) + (m >= 20 ? 4 : 0) + (m >= 28 ? 4 : 0));
  if constexpr (C::BZ == 1) {
    lz = 0;
    ly = 4 * piece + 2 * (i >> 3) + g;
    lx = i & 7;
  } else {
    lz = i >> 3;
    ly = 2 * g + ((i >> 2) & 1);
    lx = i & 3;
  }
}

// One conv expert's taps over the wave's staged images: P = 0 the 5x5x5 expert (dx 0..4), P = 1 the 3x3x3 expert (1..3).
// vb[vs]: LDS slot of sub-tile vs' voxel at tap (0,0,0); wrow: this lane's filter fragment at tap 0 of the chunk.
template <typename C, int P>
__device__ __forceinline__ void tap_pass(const u32x4* __restrict__ lds, const int (&vb)[C::VW], const bf16_t* __restrict__ wrow,
                                         size_t tap_stride, int dz_lo, int dz_hi, int dy_lo, int dy_hi, int rot, f32x16 (&acc)[C::VW]) {
  constexpr int VW = C::VW, BXH = C::BXH, PP = C::PP;
  constexpr int DX0 = P ? 1 : 0, NDX = P ? 3 : 5;
  if (dz_lo > dz_hi || dy_lo > dy_hi) return;
  auto wfrag = [&](int tap) -> u32x4 { return *reinterpret_cast<const u32x4*>(wrow + (size_t)tap * tap_stride); };
  const int ny = dy_hi - dy_lo + 1;
  const int nrows = (dz_hi - dz_lo + 1) * ny;
  // The workgroups that share an output-channel tile -- and with it every filter byte -- run on one XCD at one time and walk
  // the tap rows IN STEP: a line is fetched from HBM once and serves all of them while it is hot.  (-DDM_ROT: every workgroup
  // starts at another tap row / chunk -- built on the suspicion that the lockstep makes them all wait for the one fetch;
  // measured 6 % SLOWER, same box, fwd 270 -> 286 us and data gradient 220 -> 238 us over the five layer shapes, and 4 x the
  // filters' bytes from HBM: the phases spread over the whole 2.4 MB slice and the XCD's L2 no longer holds what they share.)
#ifdef DM_ROT
  const int r0 = rot % nrows;
#else
  const int r0 = 0;
#endif
  int dz = dz_lo + r0 / ny, dy = dy_lo + r0 % ny;
  auto next_row = [&](int& z, int& y) {
    if (++y > dy_hi) { y = dy_lo; if (++z > dz_hi) z = dz_lo; }
  };
  // filter fragments DM_PF tap rows ahead (a fetch that misses L2 comes from the Infinity Cache / HBM: 1-2 us), voxel
  // fragments one tap ahead
  u32x4 a_q[DM_PF + 1][NDX], b_cur[VW], b_nxt[VW];
  int qz = dz, qy = dy;
#pragma unroll
  for (int q = 0; q < DM_PF; ++q) {
    if (q < nrows) {
#pragma unroll
      for (int i = 0; i < NDX; ++i) a_q[q][i] = wfrag((qz * 5 + qy) * 5 + DX0 + i);
    }
    next_row(qz, qy);
  }
  {
    const int off0 = dz * PP + dy * BXH + DX0;
#pragma unroll
    for (int vs = 0; vs < VW; ++vs) b_cur[vs] = lds[vb[vs] + off0];
  }
  for (int row = 0; row < nrows; ++row) {
    int dzn = dz, dyn = dy;
    next_row(dzn, dyn);
    const bool more = row + 1 < nrows;
    if (row + DM_PF < nrows) {
#pragma unroll
      for (int i = 0; i < NDX; ++i) a_q[DM_PF][i] = wfrag((qz * 5 + qy) * 5 + DX0 + i);
    }
    next_row(qz, qy);
    const int rowoff = dz * PP + dy * BXH + DX0;
    const int rowoff_n = more ? dzn * PP + dyn * BXH + DX0 : rowoff;
#pragma unroll
    for (int i = 0; i < NDX; ++i) {
      const int offn = (i < NDX - 1) ? rowoff + i + 1 : rowoff_n;
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) b_nxt[vs] = lds[vb[vs] + offn];
      // (fences: the next tap's LDS reads stay AHEAD of this tap's MFMAs, as in conv5_deep.hip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) mma_bf16(b_cur[vs], a_q[0][i], acc[vs]);      // A = voxels, B = filter rows
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) b_cur[vs] = b_nxt[vs];
    }
#pragma unroll
    for (int q = 0; q < DM_PF; ++q)
#pragma unroll
      for (int i = 0; i < NDX; ++i) a_q[q][i] = a_q[q + 1][i];
    dz = dzn;
    dy = dyn;
  }
}

__device__ __forceinline__ u32x4 pack8(const f32x4& lo, const f32x4& hi) {
  return u32x4{pack_bf16x2(lo.x, lo.y), pack_bf16x2(lo.z, lo.w), pack_bf16x2(hi.x, hi.y), pack_bf16x2(hi.z, hi.w)};
}

template <typename C, bool FWD>
__global__ __launch_bounds__(512) void deep_mode_kernel(DmArgs a) {
  constexpr int BZ = C::BZ, SU = C::SU, VW = C::VW, TU = C::TU, NP = C::NP, PB = C::PB;
  constexpr int BXH = C::BXH, PP = C::PP, PLS = C::PLS, IMG = C::IMG, WSLOTS = C::WSLOTS, MAXI = C::MAXI;
  constexpr int NE = FWD ? E : 1;                       // accumulator sets that meet in the epilogue

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockDim.x, nw = nt >> 6;
  const int khalf = lane >> 5, l31 = lane & 31;

  if (a.tail.nblocks) {
    if ((int)blockIdx.x < a.tail.nblocks) {
      tail_run(a.tail, blockIdx.x, tid, reinterpret_cast<float*>(smem));
      return;
    }
  }
  const int cb = blockIdx.x - a.tail.nblocks, nblocks = gridDim.x - a.tail.nblocks;
  int c, g;
  if (a.xcd_classes) {
    // workgroup b runs on XCD b % 8 (observed; speed only): class c = 8 k + xcd, its tiles consecutive in time
    const int xcd = cb & 7, j = cb >> 3;
    c = (j / a.G) * 8 + xcd;
    g = j % a.G;
  } else {
    const int id = xcd_remap(cb, nblocks);
    c = id / a.G;
    g = id % a.G;
  }
  const int kz = c % a.ksplit, cot = c / a.ksplit;
  const int D = a.D, H = a.H, W = a.W, R = a.R, O = a.O, RP = a.RP, OP = a.OP, N = a.N;
  if constexpr (FWD) {
    if (a.stats_clear)
      for (int i = cb * nt + tid; i < (int)REPMODE_SCRATCH_BN_HALF; i += nblocks * nt) a.stats_clear[i] = 0.f;
  }
  const int V = D * H * W;

  // ---- this wave's chunks of the reduction
  const int nchunks = RP / 16;
  const int c_begin = (int)((long)kz * nchunks / a.ksplit), c_end = (int)((long)(kz + 1) * nchunks / a.ksplit);

  // ---- geometry.  All units of a tile share the brick's z position (SU > 1 only with one brick per sample).
  const int bz0 = (g * SU) % a.nbz;
  const int zmin = bz0 * BZ, zmax = min(zmin + BZ - 1, D - 1);
  const int dz_lo = max(0, 2 - zmax), dz_hi = min(4, D + 1 - zmin);
  const int dy_lo = max(0, 2 - (H - 1)), dy_hi = min(4, H + 1);
  const int wbase = wave * WSLOTS;
  int vb[VW];
#pragma unroll
  for (int vs = 0; vs < VW; ++vs) {
    const int ul = vs / TU, piece = vs % TU;
    int lz, ly, lx;
    row_voxel<C>(l31, piece, lz, ly, lx);
    vb[vs] = wbase + ul * IMG + khalf * PLS + (zmin + lz - 2 - PB) * PP + ly * BXH + lx;
  }
  const int nkc = nchunks, nrt = OP / 32;
  const size_t tap_stride = (size_t)OP * RP;
  const int rt = min(cot, nrt - 1);
  const bf16_t* __restrict__ wrow0 = a.w + (size_t)rt * nkc * (32 * 16) + l31 * 16 + khalf * 8;

  // ---- staging plan: the in-volume voxels of the image, item = (unit, voxel, channel group); fixed for all chunks
  const int zlo = max(PB, 0), zhi = min(PB + NP, D);      // volume planes [zlo, zhi) that the image holds
  const int cnt = (zhi - zlo) * H * W;
  int goff[MAXI], lslot[MAXI];
#pragma unroll
  for (int j = 0; j < MAXI; ++j) {
    const int it = lane + 64 * j, half = it & 1, vi = it >> 1;
    const int ul = vi / cnt, r = vi % cnt;
    const int xx = r % W, yy = (r / W) % H, zz = zlo + r / (W * H);
    const int unit = g * SU + ul;
    goff[j] = -1;
    lslot[j] = 0;
    if (ul < SU && unit < a.nunits) {
      const int n = unit / a.nbz;
      goff[j] = (((n * D + zz) * H + yy) * W + xx) * R + half * 8;
      lslot[j] = wbase + ul * IMG + half * PLS + (zz - PB) * PP + (yy + 2) * BXH + xx + 2;
    }
  }
  const int ch_half = (lane & 1) * 8;
  // zero this wave's region once: halo border, planes outside the volume, missing units
  for (int i = lane; i < WSLOTS; i += 64) lds[wbase + i] = u32x4{0u, 0u, 0u, 0u};

  u32x4 pre[MAXI];
  auto fetch = [&](const bf16_t* __restrict__ src, int chunk) {
    const int ch = chunk * 16 + ch_half;
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
      pre[j] = u32x4{0u, 0u, 0u, 0u};
      if (goff[j] >= 0 && ch < R) pre[j] = *reinterpret_cast<const u32x4*>(src + (size_t)goff[j] + chunk * 16);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < MAXI; ++j)
      if (goff[j] >= 0) lds[lslot[j]] = pre[j];
  };

  f32x16 acc[FWD ? 2 : 1][VW];
#pragma unroll
  for (int p = 0; p < (FWD ? 2 : 1); ++p)
#pragma unroll
    for (int vs = 0; vs < VW; ++vs)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][vs][r] = 0.f;

  // ---- the three 1x1 experts: one MFMA per (expert, sub-tile, chunk) on operands read straight from L2.  Forward: their own
  // accumulators, after the conv experts' loop (three more sets would not fit beside it); data gradient: the shared accumulator,
  // inside the chunk loop -- the loads' latency then hides under the SIMD's other wave instead of standing at the launch's end.
  f32x16 acc1[FWD ? 3 : 1][VW];
  if constexpr (FWD) {
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
      for (int vs = 0; vs < VW; ++vs)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[e][vs][r] = 0.f;
  }
  int voff[VW];                       // element offset of this lane's voxel row in s[e], -1 outside the volume
#pragma unroll
  for (int vs = 0; vs < VW; ++vs) {
    const int ul = vs / TU, piece = vs % TU;
    int lz, ly, lx;
    row_voxel<C>(l31, piece, lz, ly, lx);
    const int unit = g * SU + ul;
    const int gz = zmin + lz;
    voff[vs] = -1;
    if (unit < a.nunits && gz < D && ly < H && lx < W) voff[vs] = ((unit / a.nbz) * V + (gz * H + ly) * W + lx) * R;
  }
  auto one_by_one = [&](int chunk) {
#ifdef DM_NO1X1      // TIMING BUILD ONLY: the 1x1 experts' pass never runs
    if (a.N > 0) return;
#endif
    const int o = cot * 32 + l31;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r0 = chunk * 16 + khalf * 8;
    const bool rin = r0 < R;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      f32x4 b0 = z4, b1 = z4;
      if (rin && o < O) {
        const float* kp = a.k[e] + (size_t)o * a.kso + (size_t)r0 * a.ksr;
        if (a.ksr == 1) {
          b0 = *reinterpret_cast<const f32x4*>(kp);
          b1 = *reinterpret_cast<const f32x4*>(kp + 4);
        } else {
          b0 = f32x4{kp[0], kp[a.ksr], kp[2 * a.ksr], kp[3 * a.ksr]};
          b1 = f32x4{kp[4 * a.ksr], kp[5 * a.ksr], kp[6 * a.ksr], kp[7 * a.ksr]};
        }
      }
      const u32x4 bf = pack8(b0, b1);
#pragma unroll
      for (int vs = 0; vs < VW; ++vs) {
        f32x4 a0 = z4, a1 = z4;
        if (rin && voff[vs] >= 0) {
          const float* sp = a.s[e] + (size_t)voff[vs] + r0;
          a0 = *reinterpret_cast<const f32x4*>(sp);
          a1 = *reinterpret_cast<const f32x4*>(sp + 4);
        }
        if constexpr (FWD) mma_bf16(pack8(a0, a1), bf, acc1[e][vs]);
        else mma_bf16(pack8(a0, a1), bf, acc[0][vs]);
      }
    }
  };

  // position i of this workgroup's chunk range -> chunk (-DDM_ROT: rotated by the tile index, as the tap rows)
  const int nc = c_end - c_begin;
#ifdef DM_ROT
  const int rotc = nc > 0 ? g % nc : 0;
#else
  const int rotc = 0;
#endif
  auto chunk_at = [&](int i) -> int { const int t = i + rotc; return c_begin + (t >= nc ? t - nc : t); };
  const int rot = g * 7 + wave * 3;
  if (wave < nc) fetch(a.x0, chunk_at(wave));
  for (int i = wave; i < nc; i += nw) {
    const int chunk = chunk_at(i);
    const bf16_t* wchunk = wrow0 + (size_t)chunk * (32 * 16);
    const bool has_next = i + nw < nc;
    stage();
    if constexpr (FWD) {
      if (has_next) fetch(a.x0, chunk_at(i + nw));
      tap_pass<C, 0>(lds, vb, wchunk, tap_stride, dz_lo, dz_hi, dy_lo, dy_hi, rot, acc[0]);
      tap_pass<C, 1>(lds, vb, wchunk + (size_t)REPMODE_TAPS * tap_stride, tap_stride, max(dz_lo, 1), min(dz_hi, 3), max(dy_lo, 1),
                     min(dy_hi, 3), rot, acc[1]);
    } else {
      fetch(a.x1, chunk);
      one_by_one(chunk);
      tap_pass<C, 0>(lds, vb, wchunk, tap_stride, dz_lo, dz_hi, dy_lo, dy_hi, rot, acc[0]);
      stage();
      if (has_next) fetch(a.x0, chunk_at(i + nw));
      tap_pass<C, 1>(lds, vb, wchunk + (size_t)REPMODE_TAPS * tap_stride, tap_stride, max(dz_lo, 1), min(dz_hi, 3), max(dy_lo, 1),
                     min(dy_hi, 3), rot, acc[0]);
    }
  }

  if constexpr (FWD) {
    for (int i = wave; i < nc; i += nw) one_by_one(chunk_at(i));
  }

  // ---- the waves' partial sums meet in LDS: every wave writes its 64 x 32 tiles into a slab of its own (plain
  // ds_write_b32, conflict-free), the owner threads add the slabs -- two accumulator sets per round (LDS float atomics
  // onto one shared tile measured ~64 cycles per wave-instruction: 40 us of a forward launch with eight waves).
  // A thread owns 4 channels of a voxel row (8 threads = one 128-byte row of the tile).
  constexpr int SPR = 2;                                   // sets per round: nw * SPR * 8 KB <= the waves' image regions
  constexpr int NVOX = C::NVOX;
  float* red = reinterpret_cast<float*>(smem);             // [wave][SPR][NVOX voxels][32 channels]
  const int nwa = min(nw, nc);                             // waves that had chunks
  const size_t estride = (size_t)N * V * O;
  const bool split = a.ksplit > 1;
  f32x4 yv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // forward: y of this thread's (<= 2) items
#pragma unroll
  for (int e0 = 0; e0 < NE; e0 += SPR) {
    __syncthreads();        // the images (first round) / the previous round's slabs are dead
    if (wave < nc) {
#pragma unroll
      for (int sr = 0; sr < SPR; ++sr) {
        const int e = e0 + sr;
        if (e < NE) {
#pragma unroll
          for (int vs = 0; vs < VW; ++vs)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = (r & 3) + 8 * (r >> 2) + 4 * khalf;      // 32x32 C/D layout: row of register r, column = l31
              float v;
              if constexpr (FWD) {
                if (e < 2) v = acc[e & 1][vs][r];
                else v = acc1[e >= 2 ? e - 2 : 0][vs][r];
              } else {
                v = acc[0][vs][r];
              }
              red[(((wave * SPR + sr) * NVOX) + vs * 32 + m) * 32 + l31] = v;
            }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int q = tid + k * nt;
      if (q >= NVOX * 8) break;
      const int vox = q >> 3, c4 = (q & 7) * 4;
      const int vs = vox >> 5, m = vox & 31;
      const int ul = vs / TU, piece = vs % TU;
      int lz, ly, lx;
      row_voxel<C>(m, piece, lz, ly, lx);
      const int unit = g * SU + ul, gz = zmin + lz, o = cot * 32 + c4;
      if (unit >= a.nunits || gz >= D || ly >= H || lx >= W || o >= O) continue;
#ifdef DM_NOSTORE     // TIMING BUILD ONLY: the sums are computed and (practically) never written
      if (a.N > 0) continue;
#endif
      const int n = unit / a.nbz;
      const size_t off = ((size_t)n * V + (gz * H + ly) * W + lx) * O + o;
#pragma unroll
      for (int sr = 0; sr < SPR; ++sr) {
        const int e = e0 + sr;
        if (e >= NE) break;
        f32x4 pe = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int wv = 0; wv < nwa; ++wv) pe += *reinterpret_cast<const f32x4*>(&red[(((wv * SPR + sr) * NVOX) + vox) * 32 + c4]);
        if constexpr (FWD) {
          const f32x4 ge = *reinterpret_cast<const f32x4*>(a.gate + ((size_t)n * E + e) * O + o);
          yv[k] += ge * pe;
          float* pp = a.p + e * estride + off;
          if (split) {
            unsafeAtomicAdd(pp, pe.x); unsafeAtomicAdd(pp + 1, pe.y); unsafeAtomicAdd(pp + 2, pe.z); unsafeAtomicAdd(pp + 3, pe.w);
          } else {
            *reinterpret_cast<f32x4*>(pp) = pe;
          }
          if (e == NE - 1) {
            float* yp = static_cast<float*>(a.y) + off;
            const f32x4 t = yv[k];
            if (split) {
              unsafeAtomicAdd(yp, t.x); unsafeAtomicAdd(yp + 1, t.y); unsafeAtomicAdd(yp + 2, t.z); unsafeAtomicAdd(yp + 3, t.w);
            } else {
              *reinterpret_cast<f32x4*>(yp) = t;
            }
          }
        } else {
          if (split) {
            float* yp = static_cast<float*>(a.y) + off;
            unsafeAtomicAdd(yp, pe.x); unsafeAtomicAdd(yp + 1, pe.y); unsafeAtomicAdd(yp + 2, pe.z); unsafeAtomicAdd(yp + 3, pe.w);
          } else if (a.y_bf16) {
            *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(a.y) + off) = u32x2{pack_bf16x2(pe.x, pe.y), pack_bf16x2(pe.z, pe.w)};
          } else {
            *reinterpret_cast<f32x4*>(static_cast<float*>(a.y) + off) = pe;
          }
        }
      }
    }
  }
  // ---- the BatchNorm behind the block gets its batch statistics from here (RepMode.py:146-149, 212: sum and sum of squares
  // of the stored y per channel): on the deep levels the separate statistics pass is a launch of pure latency.  Lanes l, l + 8,
  // ... of a wave hold the same four channels: butterfly, then the waves' partials through LDS, 64 float atomics per workgroup
  // into slice (workgroup % 16) -- the layout bn_apply_relu_kernel totals (conv5_igemm.hip's epilogue does the same).
  if constexpr (FWD) {
    if (a.stats) {
      float sv[8];
      const f32x4 t0 = yv[0], t1 = yv[1];
      sv[0] = t0.x + t1.x; sv[1] = t0.y + t1.y; sv[2] = t0.z + t1.z; sv[3] = t0.w + t1.w;
      sv[4] = t0.x * t0.x + t1.x * t1.x; sv[5] = t0.y * t0.y + t1.y * t1.y;
      sv[6] = t0.z * t0.z + t1.z * t1.z; sv[7] = t0.w * t0.w + t1.w * t1.w;
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) sv[j] += __shfl_xor(sv[j], off, 64);
      __syncthreads();                                  // the last round's slabs are dead
      float* part = reinterpret_cast<float*>(smem);     // [wave][8 lanes][8 values]
      if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[(wave * 8 + lane) * 8 + j] = sv[j];
      }
      __syncthreads();
      if (tid < 64) {
        const int l8 = tid >> 3, j = tid & 7;
        float tot = 0.f;
        for (int wv = 0; wv < nw; ++wv) tot += part[(wv * 8 + l8) * 8 + j];
        const int o = cot * 32 + l8 * 4 + (j & 3);
        if (o < O) unsafeAtomicAdd(a.stats + (size_t)(cb & 15) * 2 * O + (size_t)(j >> 2) * O + o, tot);
      }
    }
  }
}

using MCfgP8 = MCfg<1, 8, 8, 1>;      // level 3: a tile = one 8 x 8 z plane of a sample
using MCfgS4 = MCfg<2, 4, 4, 2>;      // level 4: a tile = two whole 2 x 4 x 4 samples
using MCfgS1 = MCfg<2, 4, 4, 1>;      // level 4, forward at small batches: one sample a tile -- twice the tiles, so that the grid fills
                                      // without splitting the reduction over workgroups (six outputs' worth of float atomics)

// REPMODE_DEEP_MODE_TARGET: workgroups a data-gradient launch should reach before the reduction stops being split over
// workgroups (default: one per CU); REPMODE_DEEP_MODE_TARGET_FWD: the same for the forward, whose SIX outputs (P_0..4, y)
// all pay for a split with float atomics
static const int g_dm_target = []() { const char* e = getenv("REPMODE_DEEP_MODE_TARGET"); return e ? atoi(e) : 0; }();
static const int g_dm_target_fwd = []() { const char* e = getenv("REPMODE_DEEP_MODE_TARGET_FWD"); return e ? atoi(e) : 128; }();
static const bool g_dm_no_s1 = []() { const char* e = getenv("REPMODE_DEEP_MODE_S1"); return e && atoi(e) == 0; }();      // (A/B)
// REPMODE_DEEP_MODE_WAVES: 4 / 8 forces the waves of a workgroup (a sweep)
static const int g_dm_waves = []() { const char* e = getenv("REPMODE_DEEP_MODE_WAVES"); return e ? atoi(e) : 0; }();

struct DmPlan {
  int cfg;        // 0 unsupported, 1 the 8 x 8 plane, 2 two 2 x 4 x 4 samples, 3 one 2 x 4 x 4 sample
  int ksplit, nw, G, ncot, nbz;
};

static int cu_count() {
  static std::atomic<int> n{0};
  int v = n.load(std::memory_order_relaxed);
  if (v) return v;
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess || p.multiProcessorCount <= 0) return 256;
  n.store(p.multiProcessorCount, std::memory_order_relaxed);
  return p.multiProcessorCount;
}

// r: reduction channels, o: output channels of this direction
static DmPlan dm_plan(int n, int d, int h, int w, int r, int o, bool fwd) {
  DmPlan p{};
  if (repmode_deterministic()) return p;      // (the waves' partial sums meet through float atomics in LDS)
  if (n <= 0 || d <= 0 || h <= 0 || w <= 0 || r <= 0 || o <= 0 || (r & 7) || (o & 3)) return p;
  if ((double)n * d * h * w * (double)(r > o ? r : o) >= 2.0e9) return p;      // 32-bit element offsets
  if (h <= 4 && w <= 4 && d <= MCfgS4::MAXD) {
    p.cfg = 2;
    p.nbz = 1;
    p.G = ceil_div(n, MCfgS4::SU);
  } else if (h <= 8 && w <= 8 && d <= MCfgP8::MAXD) {
    p.cfg = 1;
    p.nbz = d;
    p.G = n * d;
  } else {
    return p;
  }
  p.ncot = ceil_div(o, 32);
  const int nchunks = round_up(r, 16) / 16;
  const int target = fwd ? g_dm_target_fwd : (g_dm_target > 0 ? g_dm_target : cu_count());
  auto split_for = [&](int tiles) { int k = 1; while ((long)tiles * p.ncot * k < target && k * 2 <= nchunks) k *= 2; return k; };
  int ks = split_for(p.G);
  if (fwd && p.cfg == 2 && ks > 1 && !g_dm_no_s1 && split_for(n) < ks && (long)r * o <= 256L * 512) {
    // forward, two-sample tiles would split: one sample a tile -- no float atomics for twice the filter traffic through L2,
    // which pays while the filters are small (same box, batch 8: 256 -> 512 40.1 -> 33.1 us, 512 -> 512 57.3 -> 60.7)
    p.cfg = 3;
    p.G = n;
    ks = split_for(n);
  }
  p.ksplit = ks;
  const int per = ceil_div(nchunks, ks);
  p.nw = per > 4 ? 8 : 4;
  if (g_dm_waves == 4 || g_dm_waves == 8) p.nw = g_dm_waves;
  return p;
}

template <typename C, bool FWD>
int launch_dm(DmArgs a, const DmPlan& p, bool hosts_tail, double alg, hipStream_t stream) {
  a.nbz = p.nbz;
  a.nunits = a.N * p.nbz;
  a.G = p.G;
  a.ncot = p.ncot;
  a.ksplit = p.ksplit;
  const long nclass = (long)p.ncot * p.ksplit;
  a.xcd_classes = (nclass % 8 == 0) ? 1 : 0;
  if (hosts_tail) repmode_tail_take(stream, &a.tail);
  const long grid = nclass * p.G + a.tail.nblocks;
  RM_REQUIRE(grid > 0 && grid < (1L << 31), "deep_mode: grid %ld out of range", grid);
  const int red_bytes = p.nw * 2 * C::NVOX * 32 * 4;      // the epilogue's slabs: [wave][2 sets][NVOX][32] floats
  int lds_bytes = p.nw * C::WSLOTS * 16;
  if (lds_bytes < red_bytes) lds_bytes = red_bytes;
  if (lds_bytes < TAIL_LDS_BYTES) lds_bytes = TAIL_LDS_BYTES;
  constexpr int LDS_MAX = (8 * C::WSLOTS * 16 > 8 * 2 * C::NVOX * 32 * 4) ? 8 * C::WSLOTS * 16 : 8 * 2 * C::NVOX * 32 * 4;
  static_assert(LDS_MAX <= 160 * 1024 && LDS_MAX >= TAIL_LDS_BYTES, "LDS budget");
  static std::atomic<unsigned> attr_set{0};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  if (!((attr_set.load(std::memory_order_acquire) >> (dev & 31)) & 1u)) {
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&deep_mode_kernel<C, FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX));
    attr_set.fetch_or(1u << (dev & 31), std::memory_order_release);
  }
  repmode_prof_begin(FWD ? REPMODE_PROF_DEEP_MODE : REPMODE_PROF_DEEP_MODE_DGRAD, alg, stream);
  hipLaunchKernelGGL((deep_mode_kernel<C, FWD>), dim3((unsigned)grid), dim3(64 * p.nw), lds_bytes, stream, a);
  repmode_prof_end(stream);
  RM_LAUNCH_CHECK("deep_mode");
  return REPMODE_OK;
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// 0: the shape is not one these kernels take (the caller keeps the five-launch path); 1: every output element has one
// writer (plain stores); > 1: that many workgroups add into each output element -- float outputs that are ZERO on entry.
// dir 0: forward (reduction = cin), 1: data gradient (reduction = cout).
extern "C" int repmode_deep_mode_plan(int dir, int n, int d, int h, int w, int cin, int cout, int dtype) {
  if (dtype != REPMODE_BF16) return 0;
  const DmPlan p = dir ? dm_plan(n, d, h, w, cout, cin, false) : dm_plan(n, d, h, w, cin, cout, true);
  return p.cfg ? p.ksplit : 0;
}

extern "C" int repmode_deep_mode_fwd_ex(const void* x, const void* wf, const float* xs, const float* k1, const float* a3, const float* a5,
                                        const float* gate, float* p, float* y, int n, int d, int h, int w, int cin, int cout, int want_stats,
                                        int* stats_half, void* stream) {
  RM_REQUIRE(x && wf && xs && k1 && a3 && a5 && gate && p && y, "deep_mode_fwd: null pointer");
  RM_REQUIRE(!want_stats || stats_half, "deep_mode_fwd: stats_half must be given with want_stats");
  const DmPlan pl = dm_plan(n, d, h, w, cin, cout, true);
  RM_REQUIRE(pl.cfg != 0, "deep_mode_fwd: shape [%d][%d][%d][%d] %d -> %d not supported (repmode_deep_mode_plan)", n, d, h, w, cin, cout);
  RM_REQUIRE(aligned16(x) && aligned16(wf) && aligned16(xs) && aligned16(k1) && aligned16(a3) && aligned16(a5) && aligned16(gate) &&
                 aligned16(p) && aligned16(y), "deep_mode_fwd: pointers must be 16-byte aligned");
  DmArgs a{};
  a.x0 = a.x1 = static_cast<const bf16_t*>(x);
  a.w = static_cast<const bf16_t*>(wf);
  const size_t es = (size_t)n * d * h * w * cin;
  a.s[0] = xs; a.s[1] = xs + es; a.s[2] = xs + 2 * es;
  a.k[0] = k1; a.k[1] = a3; a.k[2] = a5;
  a.kso = cin; a.ksr = 1;
  a.gate = gate;
  a.p = p;
  a.y = y;
  a.N = n; a.D = d; a.H = h; a.W = w; a.R = cin; a.O = cout;
  a.RP = repmode_padded_channels(cin, REPMODE_BF16, 1);
  a.OP = repmode_padded_channels(cout, REPMODE_BF16, 0);
  const double alg = 2.0 * n * d * h * w * (double)cin * cout * REPMODE_TAPS;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stats_half) *stats_half = -1;
  if (want_stats && pl.ksplit == 1 && cout <= 512) {
    // (only where every y element has ONE writer: a split reduction's partial y has no sum of squares to offer)
    float* scratch = repmode_zero_scratch(s);
    if (!scratch) return REPMODE_ELAUNCH;
    const int half = repmode_bn_scratch_half(s);
    a.stats = scratch + (size_t)half * REPMODE_SCRATCH_BN_HALF;
    a.stats_clear = scratch + (size_t)(1 - half) * REPMODE_SCRATCH_BN_HALF;
    *stats_half = half;
  }
  if (pl.cfg == 3) return launch_dm<MCfgS1, true>(a, pl, false, alg, s);
  if (pl.cfg == 2) return launch_dm<MCfgS4, true>(a, pl, false, alg, s);
  return launch_dm<MCfgP8, true>(a, pl, false, alg, s);
}

extern "C" int repmode_deep_mode_fwd(const void* x, const void* wf, const float* xs, const float* k1, const float* a3, const float* a5,
                                     const float* gate, float* p, float* y, int n, int d, int h, int w, int cin, int cout, void* stream) {
  return repmode_deep_mode_fwd_ex(x, wf, xs, k1, a3, a5, gate, p, y, n, d, h, w, cin, cout, 0, nullptr, stream);
}

// g2: bf16 [2][n][v][cout] (G_0 then G_1, expert_mix_bwd's dye_lo); s0 / s1 / s2: float [n][v][cout] = G_2, box3(G_3)/27,
// box5(G_4)/125; wd: repmode_expert_frags' data-gradient role.  dx [n][v][cin]: bf16 (dx_dtype REPMODE_BF16) or float
// when the plan says 1, float and zero on entry when it says more.  Deferred small jobs (REPMODE_DEFER) ride in this launch.
extern "C" int repmode_deep_mode_dgrad(const void* g2, const void* wd, const float* s0, const float* s1, const float* s2, const float* k1,
                                       const float* a3, const float* a5, void* dx, int dx_dtype, int n, int d, int h, int w, int cin,
                                       int cout, void* stream) {
  RM_REQUIRE(g2 && wd && s0 && s1 && s2 && k1 && a3 && a5 && dx, "deep_mode_dgrad: null pointer");
  RM_REQUIRE(dx_dtype == REPMODE_F32 || dx_dtype == REPMODE_BF16, "deep_mode_dgrad: bad dtype %d", dx_dtype);
  const DmPlan pl = dm_plan(n, d, h, w, cout, cin, false);
  RM_REQUIRE(pl.cfg != 0, "deep_mode_dgrad: shape [%d][%d][%d][%d] %d <- %d not supported (repmode_deep_mode_plan)", n, d, h, w, cin, cout);
  RM_REQUIRE(pl.ksplit == 1 || dx_dtype == REPMODE_F32, "deep_mode_dgrad: this shape splits its reduction %d ways: dx must be float (and zero)", pl.ksplit);
  RM_REQUIRE(aligned16(g2) && aligned16(wd) && aligned16(s0) && aligned16(s1) && aligned16(s2) && aligned16(k1) && aligned16(a3) &&
                 aligned16(a5) && aligned16(dx), "deep_mode_dgrad: pointers must be 16-byte aligned");
  DmArgs a{};
  a.x0 = static_cast<const bf16_t*>(g2);
  a.x1 = a.x0 + (size_t)n * d * h * w * cout;
  a.w = static_cast<const bf16_t*>(wd);
  a.s[0] = s0; a.s[1] = s1; a.s[2] = s2;
  a.k[0] = k1; a.k[1] = a3; a.k[2] = a5;
  a.kso = 1; a.ksr = cin;                  // K[co][ci]: o = ci, r = co
  a.y = dx;
  a.y_bf16 = dx_dtype == REPMODE_BF16;
  a.N = n; a.D = d; a.H = h; a.W = w; a.R = cout; a.O = cin;
  a.RP = repmode_padded_channels(cout, REPMODE_BF16, 1);
  a.OP = repmode_padded_channels(cin, REPMODE_BF16, 0);
  const double alg = 2.0 * n * d * h * w * (double)cin * cout * REPMODE_TAPS;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (pl.cfg == 2) return launch_dm<MCfgS4, false>(a, pl, true, alg, s);
  return launch_dm<MCfgP8, false>(a, pl, true, alg, s);
}
