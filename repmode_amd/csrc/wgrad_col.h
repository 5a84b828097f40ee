// wgrad_col.h -- the column-walking filter gradient (csrc/conv5_wgrad_col.hip) as conv5_wgrad.hip's launcher sees it.
#pragma once
#include "common.h"

// One call of repmode_conv5_wgrad_part (bf16, slot layout, all five dz planes) in the column form.
struct WgColCall {
  const void* x;                 // [N][D][H][W][Cin] bf16
  const void* dy;                // [N][D][H][W][Cout] bf16
  const int32_t* sample_slot;    // device, N entries
  float* dw;                     // [nslots][125][Cout][CinTot]
  int N, D, H, W, Cin, Cout, CinTot, ci_off, nslots;
  int prezeroed;                 // dw is known to be all zero (no memset before a launch that adds with atomics)
  int* plan_out;                 // not NULL: do not launch, report 1 when every element is written by plain stores
  // The dual form (repmode_conv5_wgrad_dual with the experts' layouts, nslots == 1, sample_slot NULL): dy = the 5x5x5 expert's
  // gate-scaled output gradient, dw = its gradient [Cout][Cin][125]; dy2 / dw2 = the 3x3x3 expert's ([Cout][Cin][27]).
  const void* dy2 = nullptr;
  float* dw2 = nullptr;
};

// REPMODE_WGRAD_COL / repmode_set_wgrad_col: 0 never, 1 (default) on the shapes it was measured to win, 2 wherever eligible
int repmode_wgrad_col_mode();
// Does the column form take this call (shape, switches)?  `n` samples in `nslots` slots.
bool repmode_wgrad_col_eligible(const WgColCall& c);
// Launch (or plan).  REPMODE_OK / error code; the caller brackets it with repmode_prof_begin / _end.
int repmode_wgrad_col_launch(const WgColCall& c, hipStream_t s);
