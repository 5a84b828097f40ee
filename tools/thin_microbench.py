#!/usr/bin/env python3
"""The three one-channel convolution launches of a train step (first layer forward 1->32, last layer forward 32->1, last
layer data gradient 1->32) at batch 8 (argv[1]) of 32x64x64: the layers' own kernels (csrc/thin_conv.hip) against round 2's
fold of the x taps around the general kernel (csrc/thin.hip), HIP-event time per call (REPMODE_THIN_PER_WG sweeps bricks per
workgroup of the one-input-channel kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = 'cuda:0'
d, h, w = 32, 64, 64
plan = ops.TaskPlan([i % 12 for i in range(n)], 12, dev)


def experts(co, ci):
    return [torch.randn(co, ci, k, k, k, device=dev) * 0.2 for k in (5, 3, 1, 1, 1)]


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g32 = torch.softmax(torch.randn(plan.nslots, 5, 32, device=dev), dim=1)
g1 = torch.softmax(torch.randn(plan.nslots, 5, 1, device=dev), dim=1)
wf_first, _ = ops.gatrep_merge(*experts(32, 1), g32, torch.bfloat16)
wf_last, wd_last = ops.gatrep_merge(*experts(1, 32), g1, torch.bfloat16, want_wf=True, want_wd=True)
x1 = torch.randn(n, d, h, w, 1, device=dev).bfloat16()
x32 = torch.randn(n, d, h, w, 32, device=dev).bfloat16()
rows = [('first layer forward 1->32 (bf16 out)', lambda f: ops.thin_conv_in1(x1, wf_first, plan.sample_slot, 32, out_f32=False, folded=f)),
        ('last layer forward 32->1', lambda f: ops.thin_conv_out1(x32, wf_last, plan.sample_slot, folded=f)),
        ('last layer data gradient 1->32 (bf16 out)', lambda f: ops.thin_conv_in1(x1, wd_last, plan.sample_slot, 32, out_f32=False, folded=f))]
tot = [0.0, 0.0]
for name, fn in rows:
    t_new, t_old = timed(lambda: fn(False)), timed(lambda: fn(True))
    tot[0] += t_new; tot[1] += t_old
    print('%-44s own kernel %6.1f us   folded around the general kernel %6.1f us' % (name, t_new, t_old))
print('sum: %.1f us vs %.1f us' % tuple(tot))
