/*
 * mode_block_ref.c -- plain-C restatement of one MoDE block (CPU, double accumulation).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: built by oracle/Makefile into oracle/libmode_block_ref.so and
 * loaded only by tests/ (ctypes).  It restates, with explicit loops and no library calls, the
 * arithmetic the reference gets from PyTorch in fnet/nn_modules/RepMode.py, in the reference's own
 * NCDHW layout, so that the PyTorch-based oracle (oracle/repmode_oracle.py) is cross-checked by
 * code that shares nothing with it -- including explicit backward formulas, which the PyTorch
 * oracle only has through autograd.  Pinned against tests/golden/g1_block_*.npz (captured from the
 * reference) by tests/test_oracle_c.py.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define E 5
#define K 5
#define TAPS 125

/* RepMode.py:44-49,198-200: g[n][e][o] = softmax_e(W[e*Co+o][task_n] + b[e*Co+o]) */
void ref_gate_probs(const float* gate_w, const float* gate_b, const int* tasks, int n, int num_tasks,
                    int co, float* g) {
  for (int s = 0; s < n; ++s)
    for (int o = 0; o < co; ++o) {
      double l[E], mx = -1e300, sum = 0;
      for (int e = 0; e < E; ++e) {
        l[e] = (double)gate_w[(size_t)(e * co + o) * num_tasks + tasks[s]] + gate_b[e * co + o];
        if (l[e] > mx) mx = l[e];
      }
      for (int e = 0; e < E; ++e) { l[e] = exp(l[e] - mx); sum += l[e]; }
      for (int e = 0; e < E; ++e) g[((size_t)s * E + e) * co + o] = (float)(l[e] / sum);
    }
}

static int centre3(int tap, int* t3) {
  int dz = tap / 25, dy = (tap / 5) % 5, dx = tap % 5;
  *t3 = ((dz - 1) * 3 + (dy - 1)) * 3 + (dx - 1);
  return dz >= 1 && dz <= 3 && dy >= 1 && dy <= 3 && dx >= 1 && dx <= 3;
}

/* RepMode.py:165-169,173-180,184-188: one sample's merged filter w[co][ci][125] from g[5][co] */
void ref_merge_filter(const float* k5, const float* k3, const float* k1, const float* a3, const float* a5,
                      const float* g, int co, int ci, float* w) {
  for (int o = 0; o < co; ++o)
    for (int i = 0; i < ci; ++i) {
      size_t oi = (size_t)o * ci + i;
      for (int tap = 0; tap < TAPS; ++tap) {
        int t3, c3 = centre3(tap, &t3);
        double v = (double)g[0 * co + o] * k5[oi * TAPS + tap];
        if (c3) v += (double)g[1 * co + o] * k3[oi * 27 + t3];
        if (tap == 62) v += (double)g[2 * co + o] * k1[oi];
        if (c3) v += (double)g[3 * co + o] * (a3[oi] * (1.0 / 27.0));
        v += (double)g[4 * co + o] * (a5[oi] * (1.0 / 125.0));
        w[oi * TAPS + tap] = (float)v;
      }
    }
}

/* RepMode.py:207: y = conv3d(x, w, padding='same') for one sample; x[ci][D][H][W], w[co][ci][125] */
void ref_conv5(const float* x, const float* w, int ci, int co, int D, int H, int W, float* y) {
  size_t V = (size_t)D * H * W;
  for (int o = 0; o < co; ++o)
    for (int z = 0; z < D; ++z)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          double acc = 0;
          for (int i = 0; i < ci; ++i)
            for (int tap = 0; tap < TAPS; ++tap) {
              int zi = z + tap / 25 - 2, yi = yy + (tap / 5) % 5 - 2, xi = xx + tap % 5 - 2;
              if (zi < 0 || zi >= D || yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
              acc += (double)w[((size_t)o * ci + i) * TAPS + tap] * x[i * V + ((size_t)zi * H + yi) * W + xi];
            }
          y[o * V + ((size_t)z * H + yy) * W + xx] = (float)acc;
        }
}

/* autograd of RepMode.py:207 w.r.t. the input: dx[i][v] = sum_{o,tap} dy[o][v - tap] * w[o][i][tap] */
void ref_conv5_dgrad(const float* dy, const float* w, int ci, int co, int D, int H, int W, float* dx) {
  size_t V = (size_t)D * H * W;
  for (int i = 0; i < ci; ++i)
    for (int z = 0; z < D; ++z)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          double acc = 0;
          for (int o = 0; o < co; ++o)
            for (int tap = 0; tap < TAPS; ++tap) {
              int zo = z - (tap / 25 - 2), yo = yy - ((tap / 5) % 5 - 2), xo = xx - (tap % 5 - 2);
              if (zo < 0 || zo >= D || yo < 0 || yo >= H || xo < 0 || xo >= W) continue;
              acc += (double)w[((size_t)o * ci + i) * TAPS + tap] * dy[o * V + ((size_t)zo * H + yo) * W + xo];
            }
          dx[i * V + ((size_t)z * H + yy) * W + xx] = (float)acc;
        }
}

/* autograd of RepMode.py:207 w.r.t. the filter, ACCUMULATED into dw[co][ci][125] (double) */
void ref_conv5_wgrad_acc(const float* x, const float* dy, int ci, int co, int D, int H, int W, double* dw) {
  size_t V = (size_t)D * H * W;
  for (int o = 0; o < co; ++o)
    for (int i = 0; i < ci; ++i)
      for (int tap = 0; tap < TAPS; ++tap) {
        int dz = tap / 25 - 2, dyy = (tap / 5) % 5 - 2, dxx = tap % 5 - 2;
        double acc = 0;
        for (int z = 0; z < D; ++z) {
          int zi = z + dz; if (zi < 0 || zi >= D) continue;
          for (int yy = 0; yy < H; ++yy) {
            int yi = yy + dyy; if (yi < 0 || yi >= H) continue;
            for (int xx = 0; xx < W; ++xx) {
              int xi = xx + dxx; if (xi < 0 || xi >= W) continue;
              acc += (double)dy[o * V + ((size_t)z * H + yy) * W + xx] * x[i * V + ((size_t)zi * H + yi) * W + xi];
            }
          }
        }
        dw[((size_t)o * ci + i) * TAPS + tap] += acc;
      }
}

/* autograd of RepMode.py:171-200 for ONE sample with filter gradient dw[co][ci][125]:
 * expert grads are ACCUMULATED (double), dlogit[5][co] is written (softmax Jacobian applied). */
void ref_gatrep_bwd_acc(const double* dw, const float* k5, const float* k3, const float* k1, const float* a3,
                        const float* a5, const float* g, int co, int ci, double* dk5, double* dk3, double* dk1,
                        double* da3, double* da5, double* dlogit) {
  for (int o = 0; o < co; ++o) {
    double dg[E] = {0, 0, 0, 0, 0};
    for (int i = 0; i < ci; ++i) {
      size_t oi = (size_t)o * ci + i;
      double s27 = 0, s125 = 0;
      for (int tap = 0; tap < TAPS; ++tap) {
        int t3, c3 = centre3(tap, &t3);
        double d = dw[oi * TAPS + tap];
        s125 += d;
        dg[0] += (double)k5[oi * TAPS + tap] * d;
        dk5[oi * TAPS + tap] += (double)g[0 * co + o] * d;
        if (c3) {
          s27 += d;
          dg[1] += (double)k3[oi * 27 + t3] * d;
          dk3[oi * 27 + t3] += (double)g[1 * co + o] * d;
        }
        if (tap == 62) { dg[2] += (double)k1[oi] * d; dk1[oi] += (double)g[2 * co + o] * d; }
      }
      dg[3] += (double)a3[oi] / 27.0 * s27;
      dg[4] += (double)a5[oi] / 125.0 * s125;
      da3[oi] += (double)g[3 * co + o] * s27 / 27.0;
      da5[oi] += (double)g[4 * co + o] * s125 / 125.0;
    }
    double dot = 0;
    for (int e = 0; e < E; ++e) dot += (double)g[e * co + o] * dg[e];
    for (int e = 0; e < E; ++e) dlogit[e * co + o] = (double)g[e * co + o] * (dg[e] - dot);
  }
}
