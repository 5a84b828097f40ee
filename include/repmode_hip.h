/*
 * repmode_hip.h -- C ABI of librepmode_hip.so: the MI355X (gfx950) MoDE-block hot path.
 *
 * This is the drop-in boundary.  Every entry point takes plain device pointers, sizes and a
 * HIP stream handle (passed as void*); no torch / C++ types.  All functions return 0 on
 * success and a non-zero REPMODE_E* code otherwise; repmode_last_error() gives the message.
 * Nothing here synchronises the stream: outputs and workspaces are owned by the caller, kernels are
 * enqueued on `stream` and return immediately.  The one piece of library-owned device memory is a
 * 256 KiB scratch per (device, stream), allocated on first use and kept all zero between calls (the
 * BatchNorm and gate reductions accumulate into it and clear it themselves instead of paying a
 * memset launch per call).
 *
 * The reference (Correr-Zhou/RepMode) is pure Python and has no FFI; each function below names
 * the reference lines whose arithmetic it replaces (paths relative to the reference checkout).
 *
 * Layouts (MI355X-first):
 *   activations   NDHWC ("channels last"): x[n][z][y][x][c], element type = dtype
 *   expert params the reference's own layout and dtype: float [Co][Ci][k][k][k]
 *   merged filter fragment-major: w[slot][tap][row tile][red chunk][32][KC], KC = 16 (bf16) / 8 (f32);
 *                 element (row, red) of a tap lives at
 *                   ((tap * NRT + row / 32) * NKC + red / KC) * 32 * KC + (row % 32) * KC + red % KC
 *                 with NRT = rowsP / 32, NKC = redP / KC (padded extents from repmode_padded_channels(),
 *                 zero filled).  Each 32 x KC tile is the 1 KiB one MFMA operand load reads.
 *                 wf: rows = co, red = ci, tap = (dz*5+dy)*5+dx (cross-correlation order);
 *                 wd: rows = ci, red = co, taps flipped (124 - tap): the filter that turns the
 *                 data-gradient into the same convolution.
 *   slots         a "slot" is one distinct task of the batch; slot_task[s] is its task id and
 *                 sample_slot[n] the slot of sample n.  Filters depend on the task only, so they
 *                 are merged once per slot, not once per sample as RepMode.py:182-190 does.
 */
#ifndef REPMODE_HIP_H
#define REPMODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define REPMODE_ABI_VERSION 11

/* element types of activations / merged filters */
#define REPMODE_F32 0  /* float in, exact-f32 MFMA (v_mfma_f32_32x32x2_f32)          */
#define REPMODE_BF16 1 /* bfloat16 in, f32 accumulate (v_mfma_f32_32x32x16_bf16)      */

/* error codes */
#define REPMODE_OK 0
#define REPMODE_EINVAL 1  /* bad argument (shape, dtype, null pointer, alignment) */
#define REPMODE_ELAUNCH 2 /* HIP launch error                                    */
#define REPMODE_ENODEV 3  /* no gfx950 device                                    */

#define REPMODE_NUM_EXPERTS 5 /* RepMode.py:22      */
#define REPMODE_KSIZE 5       /* RepMode.py:114-115 */
#define REPMODE_TAPS 125

int repmode_abi_version(void);
/* Deterministic mode (also: environment REPMODE_DETERMINISTIC=1 when the library is loaded): every float sum of the path gets
 * a fixed order, so that two runs on the same inputs agree BITWISE -- no input-channel split of the convolutions (the
 * per-expert pair goes through the general kernel), whole-range workgroups in the filter gradients, ordered in-workgroup
 * reductions and one writer per partial-sum slice in BatchNorm, the gate-probability gradients tile after tile from one
 * thread, one workgroup per sample in the loss ...  Costs the parallelism those splits buy.  Set it
 * between steps, not while launches of the library are being issued from another thread. */
int repmode_set_deterministic(int on);
int repmode_get_deterministic(void);
/* Data-parallel training: the persistent grids of the convolution (conv5_ws_kernel) and of the stream-K filter gradient launch
 * at most one workgroup per CU that holds it for the whole launch; `n` > 0 makes them launch on n CUs fewer, so that a
 * communication kernel issued beside them (RCCL's all-reduce of a gradient bucket) finds free CUs at once instead of queueing
 * behind the launch.  Also REPMODE_RESERVE_CUS; default 0.  Results do not depend on it. */
int repmode_set_reserve_cus(int n);
int repmode_get_reserve_cus(void);
const char* repmode_last_error(void);
/* name of device `dev`'s gcnArchName into buf; REPMODE_ENODEV when there is none */
int repmode_device_arch(int dev, char* buf, int buflen);

/* Channel padding of the merged filters for a given dtype: the input-channel dimension is
 * rounded up to the MFMA K granule, the output-channel dimension to the 32-row MFMA tile. */
int repmode_padded_channels(int channels, int dtype, int is_reduction_dim);

/* ---- gate: RepMode.py:44-49 (one-hot), :198 (Linear), :199-200 (softmax over experts) ----
 * g[s][e][o] = softmax_e(gate_w[e*Co+o][slot_task[s]] + gate_b[e*Co+o])      (float [S][5][Co]) */
int repmode_gate_softmax(const float* gate_w, const float* gate_b, const int32_t* slot_task,
                         int nslots, int num_tasks, int co, float* g, void* stream);
/* The same for several blocks in one launch (the per-expert blocks of a forward pass, one "slot" per sample): gate_w[i] /
 * gate_b[i] / co[i] / g[i] per block, one slot_task vector for all.  nblocks <= REPMODE_GATREP_MULTI_MAX (defined below). */
int repmode_gate_softmax_multi(int nblocks, const float* const* gate_w, const float* const* gate_b, const int* co,
                               const int32_t* slot_task, int nslots, int num_tasks, float* const* g, void* stream);

/* ---- GatRep forward: RepMode.py:165-169 (trans_kernel), :173-180, :182-190 (routing) ----
 * W_s = g0*K5 + g1*pad(K3) + g2*pad(K1) + g3*pad(A3/27) + g4*A5/125, per output channel,
 * written as wf and/or wd (either may be NULL, not both) in `dtype`. */
int repmode_gatrep_fwd(const float* k5, const float* k3, const float* k1, const float* a3,
                       const float* a5, const float* g, int nslots, int co, int ci, int dtype,
                       void* wf, void* wd, void* stream);

/* The two calls above as ONE launch: the merging workgroups compute the gate probabilities of their channels themselves
 * and write them to g_out [nslots][5][co] (kept for the backward pass).  wf and g_out are required, wd optional. */
int repmode_gatrep_fwd_gate(const float* k5, const float* k3, const float* k1, const float* a3, const float* a5,
                            const float* gate_w, const float* gate_b, const int32_t* slot_task, int nslots, int num_tasks,
                            int co, int ci, int dtype, float* g_out, void* wf, void* wd, void* stream);

/* repmode_gatrep_fwd_gate for several MoDE blocks in ONE launch (a train step merges all its blocks' forward filters before
 * the first convolution: they depend on parameters and tasks only).  Every pointer argument but slot_task is a HOST array
 * of nblocks entries (device pointers / channel counts per block); wd[i] may be NULL; so may wf[i] (and then g_out[i]) when
 * wd[i] is given: the data-gradient filters alone, which a train step asks for where its backward pass starts.
 * nblocks <= REPMODE_GATREP_MULTI_MAX. */
#define REPMODE_GATREP_MULTI_MAX 19
int repmode_gatrep_fwd_multi(int nblocks, const float* const* k5, const float* const* k3, const float* const* k1,
                             const float* const* a3, const float* const* a5, const float* const* gate_w,
                             const float* const* gate_b, const int* co, const int* ci, const int32_t* slot_task, int nslots,
                             int num_tasks, int dtype, float* const* g_out, void* const* wf, void* const* wd, void* stream);

/* ---- conv: RepMode.py:204-208 (train, per-sample filter) and :209-210 (eval, one filter) ----
 * y[n] = cross-correlation of x[n] with w[sample_slot[n]], 5^3, stride 1, zero pad 2, no bias.
 * x: [N][D][H][W][Cin] dtype;  w: fragment-major merged filter of nslots slots (rows = Cout, red = Cin);
 * y: [N][D][H][W][Cout], dtype, or float when out_f32 != 0.
 * Called with wf for the forward pass and with wd (Cin/Cout swapped) for the data gradient
 * (autograd of RepMode.py:207, aten::convolution_backward input grad). */
int repmode_conv5(const void* x, const void* w, const int32_t* sample_slot, void* y, int n, int d,
                  int h, int wdim, int cin, int cout, int dtype, int out_f32, void* stream);

/* Same, with a flag word `centre3`: bit 0 restricts the filter planes/rows to dz, dy in [1,3] (a filter whose
 * support is the centred 3x3x3 -- the zero-padded conv3x3 expert of RepMode.py:174 -- skips the 80 all-zero
 * taps); bit 1 (float output only) ADDS the result to y instead of overwriting it; bit 2 uses only the centre
 * x tap (dx = 2) of every (dz, dy) row -- the thin first / last layers with their x taps folded into channels,
 * see repmode_shift5 / repmode_thin_pack.
 * Bit 3: dual-expert launch (per-expert formulation, float output only): ONE grid runs two jobs over the n samples --
 * slot 0 of w with all 125 taps (the 5x5x5 expert) and slot 1 with the centred 3x3x3 support (the padded 3x3x3 expert);
 * sample_slot is not read.  Bit 4: x holds 2 n samples, the second job reads samples n .. 2n-1 (else both read the same n
 * samples).  Bit 5: y holds 2 n samples, the second job writes samples n .. 2n-1 (else both jobs ADD into the same n
 * samples, y cleared by the call unless bit 1 says it is zero already). */
int repmode_conv5_ex(const void* x, const void* w, const int32_t* sample_slot, void* y, int n, int d,
                     int h, int wdim, int cin, int cout, int dtype, int out_f32, int centre3,
                     void* stream);

/* The two convolutions of the per-expert formulation of a MoDE block (RepMode.py:171-192 by linearity:
 * y[n] = sum_e g[n,e,:] * conv(x[n], K_e), the F.conv3d of :204-208 once per conv expert instead of once per merged
 * filter) on the deep U-Net levels (x extent <= 8), bf16, as ONE uniform grid -- every workgroup runs the 5x5x5 expert's
 * 125 taps and the 3x3x3 expert's 27 taps over the same staged input, for several samples at a time (csrc/conv5_deep.hip).
 *   w: repmode_expert_frags' two slots (wf for the forward form, wd for the data-gradient form).
 *   forward form  (flags bit 0 clear): x [n][d][h][w][cin] bf16 -> y [2 n][d][h][w][cout] float: P5 = conv(x, K5) in
 *                 samples 0 .. n-1, P3 = conv(x, pad(K3)) in n .. 2n-1.
 *   data-gradient form (bit 0 set):   x [2 n][...][cin] bf16 (the two gate-scaled output gradients G5, G3)
 *                 -> y [n][...][cout] float = conv(G5, w slot 0) + conv(G3, w slot 1).
 *   flags bit 1: y is all zero already (the kernel accumulates input-channel slices with float atomics); else the call
 *   clears it.  Deferred small jobs (REPMODE_DEFER) ride in this launch as in repmode_conv5_ex.
 * repmode_conv5_deep_supported(): 1 when the shape is one this kernel takes (bf16, x extent <= 8, cin % 8 == 0). */
int repmode_conv5_deep(const void* x, const void* w, float* y, int n, int d, int h, int wdim, int cin, int cout, int flags,
                       void* stream);
int repmode_conv5_deep_supported(int wdim, int cin, int dtype);

/* A whole MoDE block of the deep U-Net levels in the per-expert formulation as ONE launch per direction
 * (csrc/deep_mode.hip; RepMode.py:171-192 by linearity + :204-208, and their autograd for the input):
 *   forward        P_e[n] = conv(x[n], K_e) (e = 0..4),  y[n] = sum_e g[n,e,:] * P_e[n]
 *   data gradient  dx[n]  = sum_e conv(G_e[n], flip(K_e)^T),  G_e = g[n,e,:] * dy[n]
 * replacing repmode_conv5_deep / the dual-expert launch + repmode_gemm3 + repmode_expert_mix_fwd (forward) and
 * repmode_conv5_deep + repmode_gemm3 + repmode_box_sum_ex (data gradient).  A workgroup owns a 64-voxel x 32-channel output
 * tile and the whole reduction; its waves split the input-channel chunks and meet in LDS, so outputs are written with plain
 * stores unless the grid would not fill the chip.
 * repmode_deep_mode_plan(dir, ...): dir 0 forward, 1 data gradient.  Returns 0 when the shape is not taken (bf16 only, volumes
 *   of at most 4 x 8 x 8 or 2 x 4 x 4 voxels, reduction channels % 8 == 0, output channels % 4 == 0, not in deterministic
 *   mode), 1 when every output element has ONE writer, k > 1 when k workgroups ADD into each element: the float outputs
 *   (p and y / dx) must then be zero on entry.
 * repmode_deep_mode_fwd:   x bf16 [n][d][h][w][cin]; wf = repmode_expert_frags' forward role; xs = repmode_box_expand's
 *   float [3][n][d][h][w][cin] (x, box3(x)/27, box5(x)/125); k1 / a3 / a5 the 1x1 experts' parameters float [cout][cin];
 *   gate float [n][5][cout] (per SAMPLE); outputs p float [5][n][d][h][w][cout] (kept for the gate gradient) and y float
 *   [n][d][h][w][cout].
 * repmode_deep_mode_dgrad: g2 bf16 [2][n][d][h][w][cout] = the two conv experts' gate-scaled output gradients
 *   (repmode_expert_mix_bwd's dye_lo); wd = repmode_expert_frags' data-gradient role; s0 = G_2, s1 = box3(G_3)/27, s2 =
 *   box5(G_4)/125, float [n][d][h][w][cout] each (repmode_expert_mix_bwd's dye_hi[0], repmode_box_pair of dye_hi[1..2]);
 *   dx [n][d][h][w][cin] in dx_dtype (REPMODE_BF16 only when the plan says 1).  Deferred small jobs (REPMODE_DEFER) ride in
 *   this launch as in repmode_conv5_ex. */
int repmode_deep_mode_plan(int dir, int n, int d, int h, int w, int cin, int cout, int dtype);
int repmode_deep_mode_fwd(const void* x, const void* wf, const float* xs, const float* k1, const float* a3, const float* a5,
                          const float* gate, float* p, float* y, int n, int d, int h, int w, int cin, int cout, void* stream);
/* want_stats: the per-channel sum / sum of squares of the stored y also go to the library's BatchNorm scratch (as
 * repmode_conv5_epi's want_stats does); *stats_half receives the half to hand to repmode_bn_relu_fwd_ex, which then skips its
 * statistics pass -- or -1 where the plan splits the reduction over workgroups (no single writer of y). */
int repmode_deep_mode_fwd_ex(const void* x, const void* wf, const float* xs, const float* k1, const float* a3, const float* a5,
                             const float* gate, float* p, float* y, int n, int d, int h, int w, int cin, int cout, int want_stats,
                             int* stats_half, void* stream);
int repmode_deep_mode_dgrad(const void* g2, const void* wd, const float* s0, const float* s1, const float* s2, const float* k1,
                            const float* a3, const float* a5, void* dx, int dx_dtype, int n, int d, int h, int w, int cin, int cout,
                            void* stream);

/* EXPERIMENT, not on the product path: the forward convolution of a merged-formulation block with
 * GatRep INSIDE the kernel (RepMode.py:171-192 fused into :204-208): the filter fragment of every tap is built in registers
 * from the experts' un-merged bf16 fragments w2 (repmode_expert_frags), the 1x1 experts' float parameters k1 / a3 / a5
 * [cout][cin] and the gate probabilities gates [nslots][5][cout]; no merged filter is written to or read from HBM.  bf16
 * input, float output y [n][d][h][w][cout] (flags bit 1: y is zero already and is added to), cin % 8 == 0, the 4 x 4 x 16
 * tile.  Kept for the A/B against repmode_gatrep_fwd + repmode_conv5 (tools/merge_ab.py, profiles/r03_merge_ab.txt). */
int repmode_conv5_merged(const void* x, const void* w2, const float* k1, const float* a3, const float* a5, const float* gates,
                         const int32_t* sample_slot, float* y, int n, int d, int h, int wdim, int cin, int cout, int flags,
                         void* stream);

/* The one-channel ends of the network as kernels of their own (csrc/thin_conv.hip; bf16).
 * repmode_conv5_thin_in1: ONE input channel -- the first block's convolution (RepMode.py:27, Net's first MoDEConv(1, 32))
 *   with its forward filter wf, or the last block's input gradient (RepMode.py:42 conv_out, autograd of :204-208) with its
 *   data-gradient filter wd.  x: [n][d][h][w] bf16; w: fragment-major [slots][125][coutP/32][1][32][16]; y:
 *   [n][d][h][w][cout] bf16 (out_f32 == 0) or float.  bias != NULL / relu: y = max(acc + bias[co], 0) (an eval-mode BatchNorm
 *   folded into filter and bias, RepMode.py:209-212).  The 125 taps are the GEMM's reduction dimension.
 * repmode_conv5_thin_out1: ONE output channel -- conv_out's forward (RepMode.py:42) with wf, or the first block's input
 *   gradient with wd (row 0 of the 32-row tile real).  x: [n][d][h][w][cin] bf16; y: [n][d][h][w] float.  The 25 (dz, dy)
 *   tap rows are the GEMM's row dimension; the diagonal sum over taps stays inside a lane. */
int repmode_conv5_thin_in1(const void* x, const void* w, const int32_t* sample_slot, void* y, int n, int d, int h, int wdim,
                           int cout, int out_f32, const float* bias, int relu, void* stream);
int repmode_conv5_thin_out1(const void* x, const void* w, const int32_t* sample_slot, float* y, int n, int d, int h, int wdim,
                            int cin, void* stream);

/* The same convolution with the input and / or the output channels split over two tensors: a U-Net skip connection
 * without the concatenated copy (RepMode.py:106 torch.cat((x_skip, up), 1)).  Input channels [0, cin1) are read from
 * x ([N][D][H][W][cin1]), [cin1, cin) from x2; output channels [0, cout1) are written to y ([...][cout1]), the rest
 * to y2 (the data gradient of such a layer).  cin1 == 0 / cout1 == 0: one tensor (x2 / y2 ignored).  cin1 must be a
 * multiple of 16 (bf16) / 8 (f32), cout1 of 32.  flags as repmode_conv5_ex's centre3 word. */
int repmode_conv5_pair(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot, void* y,
                       void* y2, int cout1, int n, int d, int h, int wdim, int cin, int cout, int dtype, int out_f32,
                       int flags, void* stream);

/* The forward conv of a MoDE block with an epilogue that takes over part of the BatchNorm3d + ReLU behind it
 * (RepMode.py:146-149, 212).  Inputs as repmode_conv5_pair (cin1 == 0: one tensor), one output, flags as repmode_conv5_ex.
 *   bias != NULL and / or relu: y = max(conv + bias[co], 0) -- an eval-mode BatchNorm whose scale gamma / sqrt(var + eps)
 *     was folded into the merged filter (through the gate probabilities) and whose shift is `bias`.  The reduction is
 *     then never split over workgroups.
 *   want_stats (bf16 input, bf16 output): the per-channel sum and sum of squares of the STORED outputs are accumulated
 *     into the library's BatchNorm scratch as the launch's epilogue; *stats_half receives the value to pass to
 *     repmode_bn_relu_fwd_ex, which must be the next BatchNorm call on the same stream and then runs no statistics pass. */
int repmode_conv5_epi(const void* x, const void* x2, int cin1, const void* w, const int32_t* sample_slot, void* y, int n,
                      int d, int h, int wdim, int cin, int cout, int dtype, int out_f32, int flags, const float* bias,
                      int relu, int want_stats, int* stats_half, void* stream);

/* ---- weight gradient of the same conv (aten::convolution_backward weight grad), summed over
 * the samples of each slot:  dw[s][tap][o][i] = sum_{n in s} sum_v dy[n][v][o] * x[n][v+tap][i]
 * dw: float [nslots][125][Cout][Cin], OVERWRITTEN (zeroed inside, then accumulated). */
int repmode_conv5_wgrad(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                        float* dw, int n, int d, int h, int wdim, int cin, int cout, int dtype,
                        void* stream);

/* Same with a mode: 0 = as above; 1 = only the filter planes dz in [1,3] are computed, the other planes of dw
 * are left UNTOUCHED; 2 (nslots == 1) = all taps, written in the experts' own layout dw[Cout][Cin][125];
 * 3 (nslots == 1) = the centred 3x3x3 taps only, written as dw[Cout][Cin][27] (RepMode.py:137's shape). */
int repmode_conv5_wgrad_ex(const void* x, const void* dy, const int32_t* sample_slot, int nslots,
                           float* dw, int n, int d, int h, int wdim, int cin, int cout, int dtype,
                           int centre3, void* stream);
/* Filter gradient of the input channels [ci_off, ci_off + cin) of a layer with cin_total input channels: x holds only
 * those channels, dw is the whole layer's [nslots][125][cout][cin_total] and must have been cleared by the caller
 * (mode bit 3 set).  One call per tensor of a skip connection. */
int repmode_conv5_wgrad_part(const void* x, const void* dy, const int32_t* sample_slot, int nslots, float* dw, int n,
                             int d, int h, int wdim, int cin, int cin_total, int ci_off, int cout, int dtype,
                             int centre3, void* stream);
/* Planning query of the three entry points above (bf16, slot layout): *direct = 1 when the call with these arguments writes
 * every element of its dw range with plain stores -- dw then needs no clearing (pass mode bit 3 and any memory) --, 0 when it
 * adds partial sums with float atomics onto a dw that must be all zero.  Same planner, switches and device as the launch. */
int repmode_conv5_wgrad_plan(int nslots, int n, int d, int h, int w, int cin, int cout, int dtype, int centre3, int* direct);

/* Two filter gradients over the same input x in ONE launch (bf16; all samples in one slot): dw_a from dy_a with mode_a,
 * dw_b from dy_b with mode_b (modes as repmode_conv5_wgrad_ex: 0 / 1 tap-major all taps / planes dz 1..3, 2 / 3 the experts'
 * own layouts; bit 3 on both or neither: the outputs are already zero).  The per-expert formulation's 5x5x5 and 3x3x3
 * experts' filter gradients. */
int repmode_conv5_wgrad_dual(const void* x, const void* dy_a, const void* dy_b, float* dw_a, float* dw_b, int n, int d, int h,
                             int wdim, int cin, int cout, int mode_a, int mode_b, void* stream);

/* ---- the same filter gradient when one channel count is 1 (first layer Cin = 1, last layer Cout = 1): the
 * 125 taps take the place of the missing channel dimension.  bf16 only.  a: [N][D][H][W][C], b: [N][D][H][W],
 * dw: float [nslots][125][C] (== the general layout with the unit dimension dropped), overwritten.
 *   flip == 0: dw[tap][c] = sum_v a[v][c] * b[v + tap]     (Cin  == 1: a = dy, b = x)
 *   flip != 0: dw[tap][c] = sum_v a[v][c] * b[v - tap]     (Cout == 1: a = x,  b = dy) */
/* (conv5_wgrad_ex: mode bit 3, conv5_wgrad_thin: flip bit 1, k2s2_wgrad_ex: param_layout bit 2 -- "dw is already
 * all zero": the call skips its own memset; lets a caller clear all its accumulation buffers with one launch.) */
int repmode_conv5_wgrad_thin(const void* a, const void* b, const int32_t* sample_slot, int nslots, float* dw,
                             int n, int d, int h, int wdim, int c, int flip, void* stream);

/* ---- the 1-channel ends of the network on the general conv kernel (bf16): the five x taps become channels.
 * shift5:    x [nrows][w] (float or bf16, one channel) -> x5 [nrows][w][8] bf16, x5[.][x][dx] = x[.][x+dx-2] (0 outside,
 *            channels 5..7 zero): the input of a 5-"channel" conv in dx-centre mode (first layer; last layer's
 *            data gradient).
 * thin_pack: re-packs the merged filter of a thin layer ([slot][125][ntiles][32][16] bf16, fragment-major; the thin
 *            dimension is one tile wide, ntiles counts the tiles of the other) for that mode:
 *            to_rows == 0: out(dz,dy,2)[r][red = dx] = w(dz,dy,dx)[r][0];  to_rows != 0: out(dz,dy,2)[row = dx][k] =
 *            w(dz,dy,dx)[0][k].  Only the 25 taps (dz,dy,2) of out are written.
 * unshift5:  y [nrows][w] = sum_dx y5[nrows][x+dx-2][dx] from the 5-row float output of the last layer's forward conv. */
int repmode_shift5(const void* x, int dtype, void* x5, long nrows, int w, void* stream);
int repmode_thin_pack(const void* w, void* out, int nslots, int ntiles, int to_rows, void* stream);
int repmode_unshift5(const float* y5, float* y, long nrows, int w, void* stream);

/* ---- GatRep backward (autograd of RepMode.py:171-200): expert, gate-probability and gate
 * parameter gradients from the per-slot filter gradient.  Outputs are OVERWRITTEN.
 * dk5 [Co][Ci][125], dk3 [Co][Ci][27], dk1/da3/da5 [Co][Ci], dgate_w [5*Co][T], dgate_b [5*Co].
 * dg_ws: float workspace [nslots][5][Co]. */
int repmode_gatrep_bwd(const float* dw, const float* k5, const float* k3, const float* k1,
                       const float* a3, const float* a5, const float* g, const int32_t* slot_task,
                       int nslots, int num_tasks, int co, int ci, float* dk5, float* dk3,
                       float* dk1, float* da3, float* da5, float* dgate_w, float* dgate_b,
                       float* dg_ws, void* stream);
/* flags & REPMODE_DEFER: the gate part (softmax Jacobian + Linear gradients, dgate_w / dgate_b) is not launched but
 * queued on the stream, see "Deferred small jobs" below. */
int repmode_gatrep_bwd_ex(const float* dw, const float* k5, const float* k3, const float* k1,
                          const float* a3, const float* a5, const float* g, const int32_t* slot_task,
                          int nslots, int num_tasks, int co, int ci, float* dk5, float* dk3,
                          float* dk1, float* da3, float* da5, float* dgate_w, float* dgate_b,
                          float* dg_ws, int flags, void* stream);

/* ---- Deferred small jobs.  On the backward pass of a MoDE block (autograd of RepMode.py:171-214) the gate backward
 * and the layout transposes of the filter gradients are a few microseconds of work each, behind a kernel boundary that
 * costs as much; the block's data-gradient convolution, which does not depend on them, is launched right after.  An
 * `_ex` entry point called with REPMODE_DEFER queues its job on the stream instead of launching it; the next
 * repmode_conv5 / _ex / _pair / _epi launch on the same stream (same device) runs every queued job in its first
 * workgroups, repmode_tail_flush launches what is still queued as one kernel (no-op when the queue is empty; at most 3
 * jobs wait, a 4th push runs the 3 before it).  The caller must issue one of the two before anything reads a deferred
 * job's outputs or overwrites its inputs.  Results are identical to the immediate form (same code). */
#define REPMODE_DEFER 1
int repmode_tail_flush(void* stream);
/* Drops the stream's queued jobs WITHOUT running them: error recovery -- if a call failed between a deferral and the
 * launch meant to host it, the queue holds jobs whose buffers may be gone (the operator library calls this at the start of
 * every train step and of every MoDE backward node; a no-op when nothing is queued). */
int repmode_tail_discard(void* stream);

/* ---- BatchNorm3d + ReLU of the MoDE block's `subsequent_layer` (RepMode.py:146-149, :212) and of the
 * stride-2 down/up stages (RepMode.py:80-84, 97-101), on a channels-last tensor viewed as [m][c].
 * Training: batch mean / biased variance over the m rows, eps, running statistics updated with `momentum`
 * and the unbiased variance; eval (training == 0): running statistics.  x has in_dtype, out has out_dtype.
 * save_mean / save_invstd [c] are float outputs (kept for the backward).  c <= 512.  Two launches: statistics
 * (partial sums into a library-owned scratch) and normalise + ReLU (whose workgroups finish the reduction
 * themselves; workgroup 0 writes the saved and running statistics). */
int repmode_bn_relu_fwd(const void* x, void* out, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float* save_mean, float* save_invstd, long m, int c,
                        float eps, float momentum, int training, int in_dtype, int out_dtype, void* stream);
/* Same; stats_half >= 0 (training): the statistics were produced by repmode_conv5_epi (see there), -1: as above. */
int repmode_bn_relu_fwd_ex(const void* x, void* out, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, float* save_mean, float* save_invstd, long m, int c, float eps,
                           float momentum, int training, int in_dtype, int out_dtype, int stats_half, void* stream);
/* Backward of the same: dx (in_dtype) from dy (out_dtype); totals [2c] float output: [0..c) = dbeta,
 * [c..2c) = dgamma. */
int repmode_bn_relu_bwd(const void* x, const void* dy, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, void* dx, float* totals, long m, int c,
                        int training, int in_dtype, int out_dtype, void* stream);

/* One launch per BatchNorm pass (csrc/bnrelu.hip, round 6; also REPMODE_BN_FUSED; ABI 11): training-mode passes over tensors
 * small enough for a grid of at most one workgroup per CU to hold in registers (every tensor of the benchmarked step but
 * level 0's) read the tensor ONCE -- partial sums, a grid-wide barrier, normalise / gradient from the registers -- instead of a
 * statistics launch and an apply launch that reads it again.  1 (default) on, 0 the two-launch passes everywhere.  Same
 * arithmetic: the sums are totalled in the same slices. */
int repmode_set_bn_fused(int on);
int repmode_get_bn_fused(void);


/* ---- the stride-2 2x2x2 stages (RepMode.py:81 Conv3d k2 s2, :98 ConvTranspose3d k2 s2; both bias-free) as a
 * gather / scatter GEMM over the 8 disjoint taps p = (pz*2+py)*2+px of every coarse voxel m:
 *   scatter == 0:  out[m][co]         = sum_p sum_ci in[fine(m,p)][ci] * w[p][co][ci]   (in: fine grid, out: coarse)
 *   scatter != 0:  out[fine(m,p)][co] = sum_ci       in[m][ci]         * w[p][co][ci]   (in: coarse, out: fine)
 * (down forward / up data-gradient, and up forward / down data-gradient.)  d, h, wdim = COARSE dims.
 * w: fragment-major [8][CoutP/32][CinP/KC][32][KC] like the merged filters; in/out channels-last, `dtype`. */
int repmode_k2s2(const void* in, const void* w, void* out, int n, int d, int h, int wdim, int cin, int cout,
                 int dtype, int scatter, void* stream);

/* Weight gradient of both stride-2 stages (bf16):  dw[p][a][b] = sum_m coarse[m][a] * fine[fine(m,p)][b]
 * (down: coarse = dy, fine = x;  up: coarse = x, fine = dy).  dw: float [8][ca][cb], overwritten. */
int repmode_k2s2_wgrad(const void* coarse, const void* fine, float* dw, int n, int d, int h, int wdim, int ca,
                       int cb, void* stream);
/* Same with the output written in the parameter's own layout (no permute copy afterwards):
 * param_layout bits 0-1: 0: dw[8][A][B]; 1: dw[A][B][2][2][2]; 2: dw[B][A][2][2][2].  Bit 2 (4): dw is zero already (the
 * voxel range is split over workgroups that add with float atomics); bit 3 (8): coarse and fine are FLOAT32 tensors (the
 * parity mode: exact float32 FMAs instead of the bf16 MFMA kernel). */
int repmode_k2s2_wgrad_ex(const void* coarse, const void* fine, float* dw, int n, int d, int h, int wdim, int ca,
                          int cb, int param_layout, void* stream);
/* The fragment-major, zero-padded filter operand `w` of repmode_k2s2 from a float parameter tensor, one launch:
 * out[p][rows/32][red/KC][32][KC] = w(row, red, p), with w stored [rows][red][2][2][2] (red_major == 0) or
 * [red][rows][2][2][2] (red_major != 0).  out: 8 * padded(rows) * padded(red) elements of `dtype`. */
int repmode_k2_frags(const float* w, int rows, int red, int red_major, int dtype, void* out, void* stream);
/* Same, plus (out_t != NULL) the operand with rows and red exchanged from the same launch: forward and
 * data-gradient filters of a stride-2 stage together.  out_t: 8 * padded(red) * padded(rows) elements. */
int repmode_k2_frags2(const float* w, int rows, int red, int red_major, int dtype, void* out, void* out_t,
                      void* stream);

/* The same for n <= REPMODE_K2_FRAGS_MULTI_MAX filters in ONE launch (ABI 11): the operands of every stride-2 stage of a forward
 * pass (RepMode.py:81, 98) ahead of the pass instead of one layout launch in front of each stage.  w / rows / red / red_major /
 * out / out_t: arrays of n (HOST arrays, copied into the launch's arguments); out_t[i] may be NULL. */
#define REPMODE_K2_FRAGS_MULTI_MAX 16
int repmode_k2_frags_multi(int n, const float* const* w, const int* rows, const int* red, const int* red_major, int dtype,
                           void* const* out, void* const* out_t, void* stream);


/* ---- gate mixing of the per-expert formulation (linearity of RepMode.py:184-188 + :207).  p: float [5][n][v][c]
 * expert outputs, g: float [n][5][c] gate probabilities per SAMPLE.
 *   fwd: y[n][v][c] = sum_e g[n][e][c] * p[e][n][v][c]
 *   bwd: dg[n][e][c] = sum_v dy * p[e]  (overwritten);  dye_lo[e] = g[n][e] * dy for e = 0, 1 in `dtype`,
 *        dye_hi[e-2] = g[n][e] * dy for e = 2..4 in float.   c <= 512. */
int repmode_expert_mix_fwd(const float* p, const float* g, float* y, int n, long v, int c, void* stream);
int repmode_expert_mix_bwd(const float* dy, const float* p, const float* g, float* dg, void* dye_lo,
                           float* dye_hi, int n, long v, int c, int dtype, void* stream);
/* Same with hi_stride elements (>= n*v*c) between the three float outputs dye_hi[e]: lets the caller pad them (the
 * batched GEMMs behind them hit a pathological rocBLAS kernel at exactly 256 x 256 outputs). */
int repmode_expert_mix_bwd_ex(const float* dy, const float* p, const float* g, float* dg, void* dye_lo,
                              float* dye_hi, long hi_stride, int n, long v, int c, int dtype, void* stream);
/* expert_mix_bwd and the avg-pool experts' operands of repmode_deep_mode_dgrad in ONE launch: besides everything
 * repmode_expert_mix_bwd_ex writes, hb3 = g[n][3] * box3(dy) / 27 and hb5 = g[n][4] * box5(dy) / 125, float [n][d][h][w][c] each
 * (= repmode_box_pair of dye_hi[1], dye_hi[2]: the gate probability is constant over a sample's voxels, so the box means are
 * taken of dy itself and need not wait for the gate-scaled tensors).  c % 4 == 0 and a volume that fits in LDS. */
int repmode_expert_mix_bwd_box(const float* dy, const float* p, const float* g, float* dg, void* dye_lo, float* dye_hi,
                               long hi_stride, float* hb3, float* hb5, int n, int d, int h, int w, int c, int dtype, void* stream);

/* ---- the three 1x1x1 experts (conv1x1, avg3x3, avg5x5: RepMode.py:135-142, 175-180) of the per-expert formulation as
 * three small float32 GEMMs in one launch:  C_i[m][n] = sum_k A_i(m, k) * B_i(n, k),  i = 0..2,
 * A_i(m, k) at a[i] + m * a_ms + k * a_ks, B_i(n, k) at b[i] + n * b_ns + k * b_ks (element strides), C_i row-major with
 * leading dimension ldc.  a, b, c: HOST arrays of three device pointers.  Covers the forward (X W^T), the filter gradient
 * (G^T X) and the data gradient (G W) of those experts through the strides.  c_is_zero != 0: the three C matrices are all
 * zero on entry -- the launch may then split K over workgroups and ADD the parts (these GEMMs are latency-bound: tens of
 * output tiles for 256 CUs); 0: C is overwritten by one workgroup per tile.  bf16_mfma != 0: the operands are rounded to
 * bfloat16 on their way into the matrix cores (float32 accumulation; the throughput mode), 0: exact float32 products. */
int repmode_gemm3(const float* const* a, long a_ms, long a_ks, const float* const* b, long b_ns, long b_ks, float* const* c,
                  int ldc, int m, int n, int k, int c_is_zero, int bf16_mfma, void* stream);

/* ---- box means of the avg-pool experts (RepMode.py:139-142, 161-163, 176-180): by linearity
 * conv(x, w1x1 (x) 1/k^3) = w1x1 applied to the zero-padded k^3 box mean of x.
 * out = box3(in3) + box5(in5); float NDHWC tensors; either input may be NULL (not both). */
int repmode_box_sum(const float* in3, const float* in5, float* out, int n, int d, int h, int w, int c,
                    void* stream);
/* out = box3(in3) + box5(in5) [+ add0] [+ add1], stored in out_dtype (float inputs; add0 / add1 may be NULL). */
int repmode_box_sum_ex(const float* in3, const float* in5, const float* add0, const float* add1, void* out,
                       int out_dtype, int n, int d, int h, int w, int c, void* stream);
/* x (dtype) -> out[0] = x widened, out[1] = box3(x), out[2] = box5(x), float [3][n][d][h][w][c]: the three 1x1 experts'
 * GEMM inputs from one launch.  Only for volumes that fit in LDS with c % 4 == 0 (REPMODE_EINVAL otherwise). */
int repmode_box_expand(const void* x, int dtype, float* out, int n, int d, int h, int w, int c, void* stream);
/* out3 = box3(in3) / 27 and out5 = box5(in5) / 125 apart, float [n][d][h][w][c] each, one launch: the avg-pool experts'
 * operands of repmode_deep_mode_dgrad (autograd of RepMode.py:176-180; the box mean is self-adjoint and commutes with the
 * 1x1 channel mixing).  Volumes that fit in LDS with c % 4 == 0. */
int repmode_box_pair(const float* in3, const float* in5, float* out3, float* out5, int n, int d, int h, int w, int c, void* stream);
/* out[row] = a[row] | b[row]: the channel concatenation of two channels-last tensors (RepMode.py:106, torch.cat((skip, up), 1))
 * for the decoder block that takes the per-expert formulation (its kernels read one input tensor).  a [rows][ca_bytes],
 * b [rows][cb_bytes], byte counts multiples of 16. */
int repmode_concat_channels(const void* a, const void* b, void* out, long rows, int ca_bytes, int cb_bytes, void* stream);
/* Filter gradient from the kernels' tap-major layout to the experts' parameter layout: out[m][t] = in[tap(t)][m]
 * for the m = Co*Ci channel pairs; ntaps_out = 125 (all taps), 27 (the centred 3x3x3 taps of the [125][m] input)
 * or 8 (the 2x2x2 stride-2 filters, input [8][m]). */
int repmode_tap_transpose(const float* in, float* out, long m, int ntaps_out, void* stream);
int repmode_tap_transpose_ex(const float* in, float* out, long m, int ntaps_out, int flags /* REPMODE_DEFER */, void* stream);
/* Softmax Jacobian + gate Linear gradients (autograd of RepMode.py:198-200) from gate-probability gradients
 * dg[s][5][Co]: dgate_w [5*Co][T], dgate_b [5*Co], overwritten.  (repmode_gatrep_bwd does this itself; the
 * per-expert formulation calls it with one "slot" per sample.) */
int repmode_gate_bwd(const float* g, const float* dg, const int32_t* slot_task, int nslots, int num_tasks, int co,
                     float* dgate_w, float* dgate_b, void* stream);
int repmode_gate_bwd_ex(const float* g, const float* dg, const int32_t* slot_task, int nslots, int num_tasks, int co,
                        float* dgate_w, float* dgate_b, int flags /* REPMODE_DEFER */, void* stream);
/* The raw conv5x5 / conv3x3 experts (RepMode.py:131-134) as two un-merged "slots" of the conv kernels' bf16
 * fragment-major layout, for the per-expert formulation: slot 0 = K5; slot 1 = K3 on its centred support,
 * of which ONLY the taps with dz, dy in [1,3] are written (what repmode_conv5_ex reads with centre3 set).
 * wf: [2][125][padded Co][padded Ci] (rows = co), wd: [2][125][padded Ci][padded Co] (rows = ci, taps
 * flipped); either may be NULL. */
int repmode_expert_frags(const float* k5, const float* k3, int co, int ci, void* wf, void* wd, void* stream);

/* ---- the two ends of the train step that the reference runs on the host ----
 * crop_flip: fnet/data/SSPdataset.py:137-155 (data_aug) for a whole batch from DEVICE-RESIDENT volumes: sample i is the
 *   [pd][ph][pw] crop of its (signal, target) volume pair -- float [D_i][H_i][W_i], dims[3i..3i+2] -- at starts[3i..3i+2],
 *   then flipped along z / y / x where bit 0 / 1 / 2 of flips[i] is set (torch.flip after the crop, :151-153).
 *   signal_vols / target_vols / dims / starts / flips are HOST arrays of n entries (device pointers inside the first two);
 *   they are copied into the launch's arguments.  n <= REPMODE_CROP_MAX_SAMPLES per call.
 *   signal_out, target_out: float [n][pd][ph][pw] on the device. */
#define REPMODE_CROP_MAX_SAMPLES 32
int repmode_crop_flip(const float* const* signal_vols, const float* const* target_vols, const int* dims, const int* starts,
                      const int* flips, int n, int pd, int ph, int pw, float* signal_out, float* target_out, void* stream);
/* mse_loss: fnet/fnet_model.py:108-109 (MSELoss(reduction='none') -> torch.mean) and :115-122 (per-sample means
 *   `loss_diff`, per-task means of the logged dict) in one pass over out / target (float [n][v]) + a one-workgroup finish:
 *   loss[0] = mean((out-target)^2);  loss_sample[i] = per-sample mean;  dout (may be NULL) = d loss / d out =
 *   2 (out-target) / (n v);  task_mean / task_count [num_tasks] (may both be NULL; then sample_task may be NULL too):
 *   mean of loss_sample over the samples of each task, and how many there were (0 -> mean 0).
 *   sums_ws: float [n] workspace that must be ALL ZERO on entry and is left all zero (allocate and clear it once). */
int repmode_mse_loss(const float* out, const float* target, const int32_t* sample_task, int n, long v, int num_tasks,
                     float* dout, float* sums_ws, float* loss, float* loss_sample, float* task_mean, float* task_count,
                     void* stream);

/* ---- the optimizer pass (fnet/fnet_model.py:55 `torch.optim.Adam(net.parameters(), lr)`, :112 `optimizer.step()`) ----
 * torch.optim.Adam's update with its defaults' structure (no weight decay, no amsgrad), float32, in place:
 *     m <- m + (1 - beta1)(g - m);  v <- v beta2 + (1 - beta2) g g;
 *     p <- p - lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)           (step = 1 on the first call)
 * the two bias corrections evaluated on the host in double precision, as torch's single-tensor path does.
 * adam_multi: ntensors <= REPMODE_ADAM_MULTI_MAX float tensors of numel[i] elements (parameter, gradient, exp_avg,
 *   exp_avg_sq) in one launch.
 * adam_expert_frags: the same update for the 5x5x5 (p5 ...: [co][ci][125]) and 3x3x3 (p3 ...: [co][ci][27]) experts of
 *   nblocks <= REPMODE_GATREP_MULTI_MAX MoDE blocks, and the UPDATED experts written once more as the convolution kernels'
 *   fragment-major bf16 operands, exactly what repmode_expert_frags(k5, k3, co, ci, wf, wd) lays out (wf[i] / wd[i] as there;
 *   either may be NULL) -- the per-expert formulation's forward pass then needs no layout launch. */
#define REPMODE_ADAM_MULTI_MAX 40
int repmode_adam_multi(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v, const long* numel,
                       double lr, double beta1, double beta2, double eps, long step, void* stream);
int repmode_adam_expert_frags(int nblocks, float* const* p5, const float* const* g5, float* const* m5, float* const* v5,
                              float* const* p3, const float* const* g3, float* const* m3, float* const* v3, const int* co,
                              const int* ci, void* const* wf, void* const* wd, double lr, double beta1, double beta2, double eps,
                              long step, void* stream);
/* The same two passes for a CAPTURED train step (a HIP-graph replay runs no host code, so the step count cannot travel in the
 * launch arguments): repmode_adam_hyper_dev increments the device-resident step count *step_dev (int64) and writes that step's
 * constants (bias corrections in double, as the host path computes them) into hyper_dev (REPMODE_ADAM_HYPER_FLOATS floats,
 * 16-byte aligned); the *_dev passes read them from there.  torch.optim.Adam(capturable=True)'s role, fnet_model.py:55, 112. */
#define REPMODE_ADAM_HYPER_FLOATS 8
int repmode_adam_hyper_dev(long* step_dev, float* hyper_dev, double lr, double beta1, double beta2, double eps, void* stream);
int repmode_adam_multi_dev(int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                           const long* numel, const float* hyper_dev, void* stream);
int repmode_adam_expert_frags_dev(int nblocks, float* const* p5, const float* const* g5, float* const* m5, float* const* v5,
                                  float* const* p3, const float* const* g3, float* const* m3, float* const* v3, const int* co,
                                  const int* ci, void* const* wf, void* const* wd, const float* hyper_dev, void* stream);

/* Developer / test switch of the convolution's pipelined form (csrc/conv5_igemm.hip, conv5_pipe_kernel; also REPMODE_CONV_PIPE):
 * bit 0 = on, bit 1 = one channel sub-tile per wave everywhere, bit 2 = also on grids smaller than the chip, bit 3 = the
 * wave-specialised kernel (MFMA waves + loader waves),  bit 4 = its items along z first, bit 5 = row-stationary tap
 * order on the 32-channel layers (another float summation order of the 125 taps), bit 6 = (round 4) the wave-specialised
 * kernel's 16-voxel-brick form on volumes 16..31 voxels wide (level 2 of the network at the 32x64x64 patch); default 121.
 * Results do not depend on bits 0-4 and 6 (same products, same summation order); bit 5 reorders the float sums of a chunk's taps. */
int repmode_set_conv_pipe(int mode);
int repmode_get_conv_pipe(void);
/* 1 when a 5x5x5 convolution of this shape (RepMode.py:204-208 or its data gradient: pass the channel counts of the direction)
 * should write its element-typed bf16 output: the whole channel reduction runs inside one workgroup on a grid that fills the
 * chip (volumes >= 32 voxels wide; 16-wide ones with enough bricks for the 16-voxel form above).  0: ask for the float
 * output -- the reduction is then split over workgroups with float atomics (the deep levels).  A pure function of the shape,
 * the device's CU count and the switch above; the operator library asks it per layer and direction. */
int repmode_conv5_elem_out(int n, int d, int h, int w, int cin, int cout, int dtype);
/* The same kind of switch for the bf16 filter gradient's wave-specialised form (csrc/conv5_wgrad.hip; also REPMODE_WGRAD_WS):
 * 0 never, 1 where a workgroup has a long tile loop, 2 wherever the tile allows, 3 (default since round 4) as 1 + the stream-K
 * form -- persistent workgroups that each take an equal range of the launch's tile-step sequence -- on volumes >= 32 voxels
 * wide with at most 64 samples (slot layout, one job; never in deterministic mode).  Results: the same sums; how the voxel
 * range is split over workgroups (float atomics) follows the form. */
int repmode_set_wgrad_ws(int mode);
int repmode_get_wgrad_ws(void);
/* The column-walking form of the same filter gradient (csrc/conv5_wgrad_col.hip; also REPMODE_WGRAD_COL; ABI 11): a workgroup
 * owns ALL 125 taps of a (slot, 16 co, 16 ci) tile and walks columns of output tiles along z with a ring of six x planes in LDS,
 * so x and dy are staged once per z step instead of once per (z step, dz plane).  bf16, slot layout, all five planes, volumes
 * >= 16 voxels wide, at most 64 samples; never in deterministic mode.  0 never, 1 (default) on the shapes it was measured to
 * win (volumes 16 .. 31 voxels wide with enough (slot, co, ci) units to fill three quarters of the chip), 2 wherever eligible.
 * With that many units a workgroup takes whole units and every element is written by plain stores (repmode_conv5_wgrad_plan
 * reports it); otherwise workgroups share units through float atomics onto the cleared dw. */
int repmode_set_wgrad_col(int mode);
int repmode_get_wgrad_col(void);
/* The column form's tap split (also REPMODE_WGRAD_COL_Q): the 125 taps of a (slot, 16 co, 16 ci) unit are cut into q parts, one
 * workgroup each -- every part walks the unit's whole voxel range with 1 / q of the accumulators, so a layer with too few units
 * for the chip (level 1 at batch 8: 128) is divided without two workgroups ever sharing an output element: plain stores, no
 * cleared dw, at the price of staging x and dy q times.  1 (default): no split; 2: two workgroups per unit (volumes >= 16
 * voxels wide).  Measured slower than the stream-K grid on level 1 (csrc/conv5_wgrad_col.hip): a tested experiment. */
int repmode_set_wgrad_col_split(int q);
int repmode_get_wgrad_col_split(void);

/* ---- sliding-window inference (fnet/fnet_model.py:149-223): the two ends of a batch of patches, SURVEY.md section 8f.3 ----
 * patch_gather: :196-205 -- out[n][pd][ph][pw] = vol[starts[3n..3n+2] + (z, y, x)], the batch's crops of the device-resident
 *   float volume [D][H][W] in one launch.  starts: HOST array [nb][3] (copied into the launch's arguments).
 * patch_blend: :207-217 -- for n = 0 .. nb-1 in this order: pred_sum[patch n] += out[n] * gauss, weight_sum[patch n] += gauss
 *   (float volumes [D][H][W]; out: [nb][pd][ph][pw] float32 or bf16 per `dtype`; gauss: float [pd][ph][pw],
 *   fnet_model.py:225-247).  Patches of one batch may overlap: every voxel is owned by one thread that walks the patches in
 *   batch order, so the additions happen in the reference's order (products and sums rounded separately), without atomics.
 * nb <= REPMODE_PATCH_MAX per call; a patch that leaves the volume is an error. */
#define REPMODE_PATCH_MAX 32
int repmode_patch_gather(const float* vol, int D, int H, int W, const int* starts, int nb, int pd, int ph, int pw, float* out,
                         void* stream);
int repmode_patch_blend(const void* out, int dtype, const float* gauss, const int* starts, int nb, int pd, int ph, int pw,
                        float* pred_sum, float* weight_sum, int D, int H, int W, void* stream);

/* repmode_expert_frags for several blocks (both roles) in ONE launch; pointer arguments are HOST arrays of nblocks entries,
 * wd[i] may be NULL, or wf[i] when wd[i] is given.  nblocks <= REPMODE_GATREP_MULTI_MAX. */
int repmode_expert_frags_multi(int nblocks, const float* const* k5, const float* const* k3, const int* co, const int* ci,
                               void* const* wf, void* const* wd, void* stream);
/* Operands KEPT across steps checked against their parameters on the device, and repaired, in ONE launch: every block's
 * forward-role operand is compared with its parameters (rounded as the layout rounds them) at 1024 sampled positions per expert
 * tensor; a block that differs is laid out again, both roles.  flags (device int[nblocks], may be NULL) receives the verdicts.
 * No host synchronisation -- catches parameter writes that move no autograd version counter (p.data.copy_(), a broadcast, a
 * foreign kernel).  Every block needs its wf. */
int repmode_expert_frags_refresh_multi(int nblocks, const float* const* k5, const float* const* k3, const int* co, const int* ci,
                                       void* const* wf, void* const* wd, int* flags, void* stream);

/* ---- measurement: per-launch HIP-event timing of the library's kernels on their own stream.
 * repmode_prof_enable(1) clears the records and starts recording every kind, (2) records the MFMA kernels of the MoDE
 * convolution only (conv5_igemm / conv5_ws, conv5_deep, the thin layers', and -- round 4 -- the filter gradient conv5_wgrad;
 * least perturbation of the timed region), (0) stops.  repmode_prof_summary()
 * synchronises the recorded events and returns, for one kernel kind, the number of launches, the
 * summed duration (ms) and the summed algorithmic work (FLOPs for the conv kernels, bytes for GatRep). */
#define REPMODE_PROF_CONV5 0    /* conv5_igemm (forward and data-gradient launches) */
#define REPMODE_PROF_WGRAD 1    /* conv5_wgrad                                      */
#define REPMODE_PROF_GATREP_FWD 2
#define REPMODE_PROF_GATREP_BWD 3
#define REPMODE_PROF_WGRAD_THIN 4 /* conv5_wgrad_thin */
#define REPMODE_PROF_CONV5_DEEP 5 /* conv5_deep (per-expert formulation, deep levels) */
#define REPMODE_PROF_CONV5_THIN 6 /* the one-channel first / last layers' own kernels */
#define REPMODE_PROF_CONV5_WS 7   /* conv5_ws_kernel / conv5_pipe_kernel: the wide levels' pipelined convolution (conv5_igemm.hip) */
#define REPMODE_PROF_DEEP_MODE 8  /* deep_mode_kernel, forward: a per-expert MoDE block of the deep levels in one launch */
#define REPMODE_PROF_HELPER 9     /* the MoDE blocks' small kernels: box sums, gemm3, expert_mix, gate softmax, expert layout */
#define REPMODE_PROF_DEEP_MODE_DGRAD 10 /* deep_mode_kernel, data gradient */
#define REPMODE_PROF_KINDS 11
int repmode_prof_enable(int on);
/* Suspend (1) / resume (0) recording; the records so far are kept (sampling a subset of the steps). */
int repmode_prof_pause(int paused);
int repmode_prof_summary(int kind, int* launches, double* total_ms, double* total_work);
/* individual records, in launch order: number of records, and the kind / duration (ms) / work of record i */
int repmode_prof_count(void);
int repmode_prof_record(int i, int* kind, double* ms, double* work);

/* ---- diagnostics: naive one-thread-per-output direct kernels (no MFMA, no LDS).  Not on the
 * product path; used by tests to bisect a failure between tiling and arithmetic. */
int repmode_debug_conv5_naive(const void* x, const void* w, const int32_t* sample_slot, float* y,
                              int n, int d, int h, int wdim, int cin, int cout, int dtype,
                              void* stream);
int repmode_debug_wgrad_naive(const void* x, const void* dy, const int32_t* sample_slot,
                              int nslots, float* dw, int n, int d, int h, int wdim, int cin,
                              int cout, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REPMODE_HIP_H */
