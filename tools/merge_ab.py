#!/usr/bin/env python3
"""A/B of BASELINE's "GatRep fused INTO the conv" on the network's level-2 layers (batch 8 of 8x16x16, 8 tasks, bf16, float
output): (a) what ships -- gate softmax + GatRep (forward filters of 8 slots to HBM) + conv5 reading them; (b) the experiment --
expert layout (once, shared by all samples) + conv5_merged building every filter fragment in registers.  Correctness of (b)
against (a) first, then HIP-event time per layer, interleaved.  `python tools/merge_ab.py [iters]`"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = 'cuda:0'
n, shape = 8, (8, 16, 16)
plan = ops.TaskPlan(list(range(n)), 12, dev)


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


tot = {'shipped': 0.0, 'fused': 0.0}
for ci, co in [(64, 128), (128, 128), (256, 128), (128, 128)]:
    def u(*s, fan):
        return (torch.rand(*s, device=dev) * 2 - 1) / fan ** 0.5
    k5, k3 = u(co, ci, 5, 5, 5, fan=ci * 125), u(co, ci, 3, 3, 3, fan=ci * 27)
    k1, a3, a5 = (u(co, ci, 1, 1, 1, fan=ci) for _ in range(3))
    gw, gb = u(5 * co, 12, fan=12), u(5 * co, fan=12)
    x = torch.randn(n, *shape, ci, device=dev).bfloat16()

    def shipped():
        g = ops.gate_softmax(gw, gb, plan, co)
        wf, _ = ops.gatrep_merge(k5, k3, k1, a3, a5, g, torch.bfloat16, want_wf=True, want_wd=False)
        return ops.conv5(x, wf, plan.sample_slot, co, out_f32=True)

    def fused():
        g = ops.gate_softmax(gw, gb, plan, co)
        w2, _ = ops.expert_frags(k5, k3, torch.bfloat16, want_wd=False)
        return ops.conv5_merged(x, w2, k1, a3, a5, g, plan.sample_slot, co)

    def fused_conv_only(w2=ops.expert_frags(k5, k3, torch.bfloat16, want_wd=False)[0], g=ops.gate_softmax(gw, gb, plan, co)):
        return ops.conv5_merged(x, w2, k1, a3, a5, g, plan.sample_slot, co)

    def shipped_conv_only(wf=ops.gatrep_merge(k5, k3, k1, a3, a5, ops.gate_softmax(gw, gb, plan, co), torch.bfloat16)[0]):
        return ops.conv5(x, wf, plan.sample_slot, co, out_f32=True)

    ya, yb = shipped(), fused()
    err = float((ya - yb).abs().max() / ya.abs().max())
    r = {}
    for rep in range(2):
        for name, fn in (('shipped', shipped), ('fused', fused), ('shipped_conv', shipped_conv_only), ('fused_conv', fused_conv_only)):
            r[name] = min(r.get(name, 1e9), timed(fn))
    tot['shipped'] += r['shipped']; tot['fused'] += r['fused']
    flop = 2.0 * n * shape[0] * shape[1] * shape[2] * ci * co * 125
    print('%4d->%4d  shipped: gate + GatRep + conv %6.1f us (conv alone %6.1f us, %5.0f TF)   fused: gate + expert layout + merging conv '
          '%6.1f us (conv alone %6.1f us, %5.0f TF)   max |diff| / max %.1e' %
          (ci, co, r['shipped'], r['shipped_conv'], flop / r['shipped_conv'] / 1e6, r['fused'], r['fused_conv'], flop / r['fused_conv'] / 1e6, err))
print('sum over the four level-2 layers: shipped %.1f us, fused %.1f us' % (tot['shipped'], tot['fused']))
