// thin_conv.hip -- the one-channel ends of the network as kernels of their own.
//
// With one input (output) channel the implicit GEMM of conv5_igemm.hip pads the reduction (row) dimension from 1 to
// 16 (32); round 2 folded the five x taps into that dimension (thin.hip: shift5 / thin_pack / unshift5 around the general
// kernel's dx-centre mode) and the three launches still cost 67 + 125 + 68 us per step for 0.4 % of the FLOPs: they are
// memory-bound operations (HBM floor ~ 15-20 us each) run through a kernel built for 125 taps of 16-channel chunks.
//
//   thin_in1:  y[n][v][co] = sum_tap W[slot(n)][tap][co] * x[n][v + tap]            (RepMode.py:27 first block's conv,
//              and -- with the data-gradient filter -- the input gradient of the last block, RepMode.py:42)
//              The 125 taps ARE the GEMM's reduction dimension: K = 25 (dz,dy) rows x 8 (5 dx + 3 zeros) (+ 2 rows of
//              padding: see the K mapping) = 15 MFMA steps; the voxel operand is a Toeplitz matrix read straight out of a bf16 halo tile in LDS (five aligned
//              dword reads + four v_alignbit per fragment), the filter (15 fragments) stays in registers while a
//              workgroup walks over several bricks.
//   thin_out1: y[n][v] = sum_{tap, ci} W[slot(n)][tap][ci] * x[n][v + tap][ci]        (RepMode.py:42 conv_out forward,
//              and the first block's input gradient if anybody asks for it)
//              The 25 (dz,dy) tap ROWS are the GEMM's row dimension: for one input row (z', y') of the halo brick
//              D[(dz,dy)][x] = sum_{dx, ci} W[(dz,dy,dx)][ci] X[z'][y'][x + dx][ci]   (5 MFMAs per 16 channels)
//              is that input row's contribution to the 25 output rows (z' - dz, y' - dy) at the SAME lane x: the
//              diagonal sum over taps is register adds inside a lane (no shuffles), one half-wave swap and one
//              cross-wave sum through LDS at the end.  25 of 32 MFMA rows do useful work (5 of 32 in round 2's dx-fold).
#include "common.h"

#include <cstdlib>

namespace {

constexpr int TB_Z = 4, TB_Y = 4, TB_X = 32;                 // brick of output voxels
constexpr int TH_Z = TB_Z + 4, TH_Y = TB_Y + 4;              // halo rows
constexpr int TH_PITCH = 40;                                  // bf16 elements per halo row (36 used; 80 bytes = 20 dwords)
constexpr int TH_ELEMS = (TH_Z + 2) * TH_Y * TH_PITCH;       // 8 halo planes + 2 zero planes (see the K mapping): 6.4 KB
// K mapping of thin_in1: MFMA step j, lane half 0 takes tap row (dz0, dy) = (j / 5, j % 5) for j < 10 and (4, j - 10) above;
// lane half 1 takes (dz0 + 2, dy) -- a CONSTANT distance in the tile, so every fragment address is one per-lane base plus
// an immediate (a per-lane row index would make 52 addresses per brick, which the compiler hoists out of the brick loop
// and spills).  For j >= 10 half 1's row (dz 6) does not exist: zero filter fragment, reads land in the two zero planes.
constexpr int TIN_KSTEPS = 15;
constexpr int TIN_STAGE = (TB_X + 4) / 4;                     // halo elements per thread: a quarter of one of the 64 halo rows

struct ThinInArgs {
  const bf16_t* x;             // [N][D][H][W] (one channel)
  const bf16_t* w;             // [S][125][CoutP/32][1][32][16] fragment-major, reduction index 0 real
  const int32_t* sample_slot;
  void* y;                     // [N][D][H][W][Cout], bf16 or float
  const float* bias;           // optional epilogue: y = max(acc + bias[co], 0) (eval-mode BatchNorm folded, RepMode.py:209-212)
  int relu;
  int N, D, H, W, Cout, CoutP;
  int nbz, nby, nbx, nbricks;  // bricks per sample along z, y, x; all samples' bricks
  int per_wg;                  // consecutive bricks one workgroup walks over
  int wide;                    // bf16 output: 16-byte stores through v_permlane32_swap (Cout % 16 == 0)
};

template <bool OUT_F32>
__global__ __launch_bounds__(256, 3) void thin_in1_kernel(ThinInArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[2][TH_ELEMS];
  __shared__ u32x4 filt[TIN_KSTEPS][64];        // the current slot's filter fragments, lane-major (15 KB): see the K mapping
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int khalf = lane >> 5, l31 = lane & 31;
  const int D = a.D, H = a.H, W = a.W, Cout = a.Cout;
  const int nrt = a.CoutP / 32;
  const int b_begin = blockIdx.x * a.per_wg, b_end = min(b_begin + a.per_wg, a.nbricks);
  if (b_begin >= b_end) return;
  const int per_sample = a.nbz * a.nby * a.nbx;

  // zero once: the pads of both tiles (row elements 36 .. 39: read by the Toeplitz fragments, times a zero filter entry) and
  // their two extra planes
  for (int i = tid; i < 2 * TH_ELEMS; i += 256) {
    const int buf = i / TH_ELEMS, e = i % TH_ELEMS;
    if (e % TH_PITCH >= TB_X + 4 || e >= TH_Z * TH_Y * TH_PITCH) tile[buf][e] = 0;
  }

  auto brick_origin = [&](int b, int& n, int& z0, int& y0, int& x0) {
    n = b / per_sample;
    const int r = b % per_sample;
    x0 = (r % a.nbx) * TB_X;
    y0 = ((r / a.nbx) % a.nby) * TB_Y;
    z0 = (r / (a.nbx * a.nby)) * TB_Z;
  };
  // this thread's share of a halo tile: nine consecutive x positions (a quarter) of halo row tid / 4
  const int hrow = tid >> 2, hx0 = (tid & 3) * TIN_STAGE;
  const int hzz = hrow / TH_Y, hyy = hrow % TH_Y;
  auto load_halo = [&](int b, bf16_t (&v)[TIN_STAGE]) {
    int n, z0, y0, x0;
    brick_origin(b, n, z0, y0, x0);
    const int gz = z0 + hzz - 2, gy = y0 + hyy - 2;
    const bool row_in = (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H;
    const bf16_t* xr = a.x + (((size_t)n * D + (row_in ? gz : 0)) * H + (row_in ? gy : 0)) * W;
#pragma unroll
    for (int u = 0; u < TIN_STAGE; ++u) {
      const int gx = x0 + hx0 + u - 2;
      v[u] = (row_in && (unsigned)gx < (unsigned)W) ? xr[gx] : (bf16_t)0;
    }
  };
  auto store_halo = [&](int buf, const bf16_t (&v)[TIN_STAGE]) {
#pragma unroll
    for (int u = 0; u < TIN_STAGE; ++u) tile[buf][hrow * TH_PITCH + hx0 + u] = v[u];
  };

  for (int rt = 0; rt < nrt; ++rt) {          // (one output-channel tile of 32 at a time: the first layer has exactly one)
    int slot_loaded = -1;
    bf16_t stage[TIN_STAGE];
    __syncthreads();                          // (pads zeroed / the previous tile's reads finished)
    load_halo(b_begin, stage);
    store_halo(0, stage);
    __syncthreads();
    for (int b = b_begin; b < b_end; ++b) {
      const int buf = (b - b_begin) & 1;
      int n, z0, y0, x0;
      brick_origin(b, n, z0, y0, x0);
      const bool more = b + 1 < b_end;
      if (more) load_halo(b + 1, stage);      // the next brick's halo travels while this one is multiplied
      const int slot = a.sample_slot[n];
      if (slot != slot_loaded) {              // (workgroup-uniform: every wave works on the same brick)
        // this slot's filter fragments into LDS, by all 256 threads: fragment (step j, lane L): element i = dx (5 real of 8)
        // of tap row (dz, dy).  Nobody reads `filt` here: the loop's closing barrier is behind every wave.
        const bf16_t* ws = a.w + ((size_t)slot * REPMODE_TAPS * nrt + rt) * 512;
        const uint32_t tap_elems = (uint32_t)nrt * 512u;
        for (int f = tid; f < TIN_KSTEPS * 64; f += 256) {
          const int j = f >> 6, L = f & 63, kh = L >> 5;
          const int dz0 = j < 10 ? j / 5 : 4, dy = j < 10 ? j % 5 : j - 10;
          const int dz = dz0 + 2 * kh;
          uint32_t e[5] = {0u, 0u, 0u, 0u, 0u};
          if (dz < 5) {
#pragma unroll
            for (int i = 0; i < 5; ++i) e[i] = ws[(uint32_t)((L & 31) * 16) + (uint32_t)((dz * 5 + dy) * 5 + i) * tap_elems];
          }
          filt[j][L] = u32x4{e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4], 0u};
        }
        slot_loaded = slot;
        __syncthreads();
      }
      // ---- 15 K steps x 4 row segments (wave = z plane, segment = y row) of Toeplitz fragments, software-pipelined: the
      // next step's filter fragment and raw dwords are requested before this step's MFMAs
      f32x16 acc[TB_Y];
#pragma unroll
      for (int vs = 0; vs < TB_Y; ++vs)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[vs][r] = 0.f;
      const uint32_t* t32 = reinterpret_cast<const uint32_t*>(tile[buf]) + ((wave + 2 * khalf) * TH_Y) * (TH_PITCH / 2) + (l31 >> 1);
      const int sh = (l31 & 1) * 16;
      auto load_raw = [&](int j, uint32_t (&raw)[TB_Y][5]) {
        const int dz0 = j < 10 ? j / 5 : 4, dy = j < 10 ? j % 5 : j - 10;
#pragma unroll
        for (int vs = 0; vs < TB_Y; ++vs) {
          const int dw = (dz0 * TH_Y + vs + dy) * (TH_PITCH / 2);        // compile-time
#pragma unroll
          for (int i = 0; i < 5; ++i) raw[vs][i] = t32[dw + i];
        }
      };
      u32x4 a_cur = filt[0][lane], a_nxt = a_cur;
      uint32_t raw_cur[TB_Y][5], raw_nxt[TB_Y][5];
      load_raw(0, raw_cur);
#pragma unroll
      for (int j = 0; j < TIN_KSTEPS; ++j) {
        if (j + 1 < TIN_KSTEPS) {
          a_nxt = filt[j + 1][lane];
          load_raw(j + 1, raw_nxt);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int vs = 0; vs < TB_Y; ++vs) {
          const uint32_t* d = raw_cur[vs];
          const u32x4 bf = u32x4{__builtin_amdgcn_alignbit(d[1], d[0], sh), __builtin_amdgcn_alignbit(d[2], d[1], sh),
                                 __builtin_amdgcn_alignbit(d[3], d[2], sh), __builtin_amdgcn_alignbit(d[4], d[3], sh)};
          acc[vs] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_cur), __builtin_bit_cast(bf16x8, bf), acc[vs], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        a_cur = a_nxt;
#pragma unroll
        for (int vs = 0; vs < TB_Y; ++vs)
#pragma unroll
          for (int i = 0; i < 5; ++i) raw_cur[vs][i] = raw_nxt[vs][i];
      }
      // ---- epilogue: rows = output channels (4 consecutive per register quad), column = this lane's voxel
      const int gz = z0 + wave, gx = x0 + l31;
#pragma unroll
      for (int vs = 0; vs < TB_Y; ++vs) {
        const int gy = y0 + vs;
        const bool inside = gz < D && gy < H && gx < W;
        const size_t vox = (((size_t)n * D + gz) * H + gy) * W + gx;
        if constexpr (!OUT_F32) {
          if (a.wide) {
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
              const int co16 = rt * 32 + 16 * qp;
              if (co16 >= Cout) continue;
              uint32_t pk[2][2];
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int q = 2 * qp + g;
                float v0 = acc[vs][4 * q + 0], v1 = acc[vs][4 * q + 1], v2 = acc[vs][4 * q + 2], v3 = acc[vs][4 * q + 3];
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
              if (inside)
                *reinterpret_cast<u32x4*>(static_cast<bf16_t*>(a.y) + vox * Cout + co16 + 8 * khalf) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
            continue;
          }
        }
        if (!inside) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = rt * 32 + 8 * q + 4 * khalf;
          if (co >= Cout) continue;
          float v[4] = {acc[vs][4 * q + 0], acc[vs][4 * q + 1], acc[vs][4 * q + 2], acc[vs][4 * q + 3]};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (a.bias && co + i < Cout) v[i] += a.bias[co + i];
            if (a.relu) v[i] = fmaxf(v[i], 0.f);
          }
          if constexpr (OUT_F32) {
            float* yp = static_cast<float*>(a.y) + vox * Cout + co;
            if ((Cout & 3) == 0) {
              *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (co + i < Cout) yp[i] = v[i];
            }
          } else {
            bf16_t* yp = static_cast<bf16_t*>(a.y) + vox * Cout + co;
            if ((Cout & 3) == 0) {
              *reinterpret_cast<u32x2*>(yp) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (co + i < Cout) yp[i] = f32_to_bf16(v[i]);
            }
          }
        }
      }
      if (more) store_halo(buf ^ 1, stage);
      __syncthreads();      // the next tile is complete, and nobody reads this one any more
    }
  }
}

// ---- one output channel
constexpr int TO_BXH = TB_X + 4, TO_VH = TH_Z * TH_Y * TO_BXH;          // halo brick: 8 x 8 x 36 voxels
constexpr int TO_PLS = ((TO_VH + 7) / 8) * 8 + 4;                        // plane stride in 16-byte slots (conv5_igemm.hip's layout)
constexpr int TO_LDS_BYTES = 2 * TO_PLS * 16;
constexpr int TO_YREL = 6;                                               // relative output rows one wave's two input rows reach

struct ThinOutArgs {
  const bf16_t* x;             // [N][D][H][W][Cin]
  const bf16_t* w;             // [S][125][1][CinP/16][32][16] fragment-major, row 0 real
  const int32_t* sample_slot;
  float* y;                    // [N][D][H][W]
  int N, D, H, W, Cin, CinP;
  int nbz, nby, nbx;
};

__global__ __launch_bounds__(256, 2) void thin_out1_kernel(ThinOutArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* lds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int khalf = lane >> 5, l31 = lane & 31;
  const int D = a.D, H = a.H, W = a.W, Cin = a.Cin;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = bid % a.nbx; bid /= a.nbx;
  const int by = bid % a.nby; bid /= a.nby;
  const int bz = bid % a.nbz;
  const int n = bid / a.nbz;
  const int z0 = bz * TB_Z, y0 = by * TB_Y, x0 = bx * TB_X;
  const int nkc = a.CinP / 16;
  const bf16_t* __restrict__ xn = a.x + (size_t)n * D * H * W * Cin;
  // filter fragment of tap row t = l31 (dz = t / 5, dy = t % 5), x tap dx, chunk: 8 channels of output-channel row 0
  const bf16_t* __restrict__ wl = a.w + (size_t)a.sample_slot[n] * REPMODE_TAPS * nkc * 512 + (size_t)(min(l31, 24) * 5) * nkc * 512 + khalf * 8;
  const bool vec_ok = (Cin & 7) == 0;
  const float m0 = khalf == 0 ? 1.f : 0.f, m1 = 1.f - m0;

  float out[TB_Z][TO_YREL];
#pragma unroll
  for (int z = 0; z < TB_Z; ++z)
#pragma unroll
    for (int q = 0; q < TO_YREL; ++q) out[z][q] = 0.f;

  for (int chunk = 0; chunk < nkc; ++chunk) {
    const int ci0 = chunk * 16;
    u32x4 af[5];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      af[dx] = u32x4{0u, 0u, 0u, 0u};
      if (l31 < 25) af[dx] = *reinterpret_cast<const u32x4*>(wl + ((size_t)dx * nkc + chunk) * 512);
    }
    __syncthreads();                 // all waves finished reading the previous chunk's image
    // ---- stage the halo brick (conv5_igemm.hip's image: plane p = 16-byte channel group p of every halo voxel)
    constexpr int NITEMS = 2 * TO_VH, UNR = 9;
    for (int it0 = 0; it0 < NITEMS; it0 += 256 * UNR) {
      u32x4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int it = it0 + u * 256 + tid;
        v[u] = u32x4{0u, 0u, 0u, 0u};
        if (it < NITEMS) {
          const int pl = it & 1, vh = it >> 1;
          const int xx = vh % TO_BXH, t2 = vh / TO_BXH;
          const int yy = t2 % TH_Y, zz = t2 / TH_Y;
          const int gz = z0 + zz - 2, gy = y0 + yy - 2, gx = x0 + xx - 2;
          const int c = ci0 + pl * 8;
          if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin) {
            const bf16_t* p = xn + ((size_t)(gz * H + gy) * W + gx) * Cin + c;
            if (vec_ok) {
              v[u] = *reinterpret_cast<const u32x4*>(p);
            } else {
              bf16_t e[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) e[k] = (c + k < Cin) ? p[k] : (bf16_t)0;
              v[u] = *reinterpret_cast<const u32x4*>(e);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int it = it0 + u * 256 + tid;
        if (it < NITEMS) lds[(it & 1) * TO_PLS + (it >> 1)] = v[u];
      }
    }
    __syncthreads();
    // ---- this wave's input rows y' = 2 wave + {0, 1}, every z' plane
    const int vb = khalf * TO_PLS + (2 * wave) * TO_BXH + l31;
#pragma unroll
    for (int zp = 0; zp < TH_Z; ++zp) {
      f32x16 dacc[2];
#pragma unroll
      for (int yy = 0; yy < 2; ++yy)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[yy][r] = 0.f;
#pragma unroll
      for (int dx = 0; dx < 5; ++dx)
#pragma unroll
        for (int yy = 0; yy < 2; ++yy) {
          const u32x4 bf = lds[vb + (zp * TH_Y + yy) * TO_BXH + dx];
          dacc[yy] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[dx]), __builtin_bit_cast(bf16x8, bf), dacc[yy], 0, 0, 0);
        }
      // scatter: register r holds tap row i = (r & 3) + 8 (r >> 2) + 4 khalf -> output row (z' - dz, y' - dy), same lane
#pragma unroll
      for (int yy = 0; yy < 2; ++yy)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i0 = (r & 3) + 8 * (r >> 2), i1 = i0 + 4;
          const int za = zp - i0 / 5, qa = yy - i0 % 5 + 4;       // lanes 0-31
          const int zb = zp - i1 / 5, qb = yy - i1 % 5 + 4;       // lanes 32-63
          if (i0 < 25 && za >= 0 && za < TB_Z) out[za][qa] = fmaf(dacc[yy][r], m0, out[za][qa]);
          if (i1 < 25 && zb >= 0 && zb < TB_Z) out[zb][qb] = fmaf(dacc[yy][r], m1, out[zb][qb]);
        }
      __builtin_amdgcn_sched_barrier(0);      // (one plane's reads and accumulators at a time: no spills)
    }
  }
  // ---- the two lane halves hold the two halves of the taps: add; then the four waves' partial rows through LDS
  __syncthreads();                   // (the image is no longer read)
  float* red = reinterpret_cast<float*>(smem);            // [wave][z][yrel][32]
#pragma unroll
  for (int z = 0; z < TB_Z; ++z)
#pragma unroll
    for (int q = 0; q < TO_YREL; ++q) {
      const float v = out[z][q] + __shfl_xor(out[z][q], 32, 64);
      if (khalf == 0) red[((wave * TB_Z + z) * TO_YREL + q) * 32 + l31] = v;
    }
  __syncthreads();
#pragma unroll
  for (int o = tid; o < TB_Z * TB_Y * TB_X; o += 256) {
    const int x = o % TB_X, y = (o / TB_X) % TB_Y, z = o / (TB_X * TB_Y);
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int q = y - 2 * w + 4;           // wave w's input rows 2 w, 2 w + 1 reach output row y through dy = 2 w + yy - y
      if (q >= 0 && q < TO_YREL) sum += red[((w * TB_Z + z) * TO_YREL + q) * 32 + x];
    }
    const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
    if (gz < D && gy < H && gx < W) a.y[(((size_t)n * D + gz) * H + gy) * W + gx] = sum;
  }
}

static const int g_thin_wide = []() { const char* e = getenv("REPMODE_CONV_WIDE"); return e ? atoi(e) : 1; }();
// bricks per workgroup: two workgroups per CU (254 registers: that is what fits), at most 8 bricks each (the filter fragments
// are re-read per workgroup).  Same box, batch 8 of 32x64x64 (2048 bricks), us per launch at 2 / 4 / 8 bricks per workgroup:
// 65.4 / 52.4 / 64.6 (first layer), 64.2 / 50.3 / 63.8 (last layer's data gradient); the fold around the general kernel: 70.6.
static int thin_per_wg(int nbricks) {
  static const int forced = []() { const char* e = getenv("REPMODE_THIN_PER_WG"); return e ? atoi(e) : 0; }();
  if (forced > 0) return forced;
  int per = (nbricks + 511) / 512;
  return per < 1 ? 1 : (per > 8 ? 8 : per);
}

}  // namespace

// y[n] = x[n] (*) w[sample_slot[n]] for a ONE-channel bf16 input x [N][D][H][W]: w is the fragment-major filter with one
// (padded to 16) reduction channel, rows = the `cout` output channels -- the first block's forward filter wf, or the last
// block's data-gradient filter wd.  y: [N][D][H][W][cout] bf16 (out_f32 == 0) or float.  bias / relu: optional epilogue
// y = max(acc + bias[co], 0).
extern "C" int repmode_conv5_thin_in1(const void* x, const void* w, const int32_t* sample_slot, void* y, int n, int d, int h, int wdim,
                                      int cout, int out_f32, const float* bias, int relu, void* stream) {
  RM_REQUIRE(x && w && sample_slot && y, "conv5_thin_in1: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && cout > 0, "conv5_thin_in1: bad shape");
  RM_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)x & 1) == 0, "conv5_thin_in1: misaligned pointer");
  ThinInArgs a{};
  a.x = static_cast<const bf16_t*>(x);
  a.w = static_cast<const bf16_t*>(w);
  a.sample_slot = sample_slot;
  a.y = y;
  a.bias = bias;
  a.relu = relu ? 1 : 0;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cout = cout;
  a.CoutP = repmode_padded_channels(cout, REPMODE_BF16, 0);
  a.nbz = ceil_div(d, TB_Z); a.nby = ceil_div(h, TB_Y); a.nbx = ceil_div(wdim, TB_X);
  const long nbricks = (long)n * a.nbz * a.nby * a.nbx;
  RM_REQUIRE(nbricks < (1L << 31), "conv5_thin_in1: too many bricks");
  a.nbricks = (int)nbricks;
  a.per_wg = thin_per_wg(a.nbricks);
  a.wide = (!out_f32 && g_thin_wide && cout % 16 == 0) ? 1 : 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)ceil_div(a.nbricks, a.per_wg);
  // algorithmic FLOPs: the layer's 125-tap convolution with one input channel
  repmode_prof_begin(REPMODE_PROF_CONV5_THIN, 2.0 * n * d * h * wdim * (double)cout * REPMODE_TAPS, s);
  if (out_f32) hipLaunchKernelGGL(thin_in1_kernel<true>, dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(thin_in1_kernel<false>, dim3(grid), dim3(256), 0, s, a);
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("conv5_thin_in1");
  return REPMODE_OK;
}

// y[n][v] = sum_{tap, ci} w[sample_slot[n]][tap][row 0][ci] x[n][v + tap][ci]: ONE output channel (the last block's forward
// filter wf, or the first block's data-gradient filter wd: rows padded to 32, row 0 real).  x: [N][D][H][W][cin] bf16;
// y: [N][D][H][W] float.
extern "C" int repmode_conv5_thin_out1(const void* x, const void* w, const int32_t* sample_slot, float* y, int n, int d, int h,
                                       int wdim, int cin, void* stream) {
  RM_REQUIRE(x && w && sample_slot && y, "conv5_thin_out1: null pointer");
  RM_REQUIRE(n > 0 && d > 0 && h > 0 && wdim > 0 && cin > 0, "conv5_thin_out1: bad shape");
  RM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 3) == 0, "conv5_thin_out1: misaligned pointer");
  RM_REQUIRE((size_t)d * h * wdim * cin * 2 < ((size_t)1 << 31), "conv5_thin_out1: one sample of the input must be smaller than 2 GiB");
  ThinOutArgs a{};
  a.x = static_cast<const bf16_t*>(x);
  a.w = static_cast<const bf16_t*>(w);
  a.sample_slot = sample_slot;
  a.y = y;
  a.N = n; a.D = d; a.H = h; a.W = wdim; a.Cin = cin;
  a.CinP = repmode_padded_channels(cin, REPMODE_BF16, 1);
  a.nbz = ceil_div(d, TB_Z); a.nby = ceil_div(h, TB_Y); a.nbx = ceil_div(wdim, TB_X);
  const long grid = (long)n * a.nbz * a.nby * a.nbx;
  RM_REQUIRE(grid > 0 && grid < (1L << 31), "conv5_thin_out1: grid %ld out of range", grid);
  hipStream_t s = static_cast<hipStream_t>(stream);
  static bool attr_set[32] = {};
  int dev = 0;
  RM_HIP(hipGetDevice(&dev));
  if (!attr_set[dev & 31]) {
    RM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_out1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TO_LDS_BYTES));
    attr_set[dev & 31] = true;
  }
  repmode_prof_begin(REPMODE_PROF_CONV5_THIN, 2.0 * n * d * h * wdim * (double)cin * REPMODE_TAPS, s);
  hipLaunchKernelGGL(thin_out1_kernel, dim3((unsigned)grid), dim3(256), TO_LDS_BYTES, s, a);
  repmode_prof_end(s);
  RM_LAUNCH_CHECK("conv5_thin_out1");
  return REPMODE_OK;
}
