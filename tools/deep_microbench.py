#!/usr/bin/env python3
"""The deep levels' convolution pair (per-expert formulation) at the network's layer shapes, batch 8 (argv[1]: batch):
csrc/conv5_deep.hip against the general kernel's dual-expert launch, forward and data-gradient forms, HIP-event time per
launch (output pre-zeroed, as the step's pool hands it over).  REPMODE_DEEP_TARGET=<workgroups> sweeps the split."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = 'cuda:0'
LAYERS = [((4, 8, 8), 128, 256), ((4, 8, 8), 256, 256), ((4, 8, 8), 512, 256), ((2, 4, 4), 256, 512), ((2, 4, 4), 512, 512)]
P = ctypes.c_void_p


def timed(fn):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


stream = lambda: P(torch.cuda.current_stream().cuda_stream)
slot0 = torch.zeros(n, dtype=torch.int32, device=dev)
tot = {'deep': 0.0, 'dual': 0.0}
for shape, ci, co in LAYERS:
    k5 = torch.randn(co, ci, 5, 5, 5, device=dev) * 0.02
    k3 = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
    wf, wd = ops.expert_frags(k5, k3, torch.bfloat16, want_wd=True)
    x = torch.randn(n, *shape, ci, device=dev).bfloat16()
    g2 = torch.randn(2 * n, *shape, co, device=dev).bfloat16()
    y2 = torch.zeros(2 * n, *shape, co, device=dev)
    dx = torch.zeros(n, *shape, ci, device=dev)
    d, h, w = shape
    flop = 2.0 * n * d * h * w * ci * co * 125

    def deep_f(): ops.conv5_deep(x, wf, co, two_in=False, out=y2, zeroed=True)
    def deep_d(): ops.conv5_deep(g2, wd, ci, two_in=True, out=dx, zeroed=True)
    def dual_f(): _lib.call('repmode_conv5_ex', P(x.data_ptr()), P(wf.data_ptr()), P(slot0.data_ptr()), P(y2.data_ptr()), n, d, h, w, ci, co, 1, 1, 2 | 8 | 32, stream())
    def dual_d(): _lib.call('repmode_conv5_ex', P(g2.data_ptr()), P(wd.data_ptr()), P(slot0.data_ptr()), P(dx.data_ptr()), n, d, h, w, co, ci, 1, 1, 2 | 8 | 16, stream())
    gy5, gy3 = g2[:n], g2[n:]
    dk5, dk3 = torch.empty_like(k5), torch.empty_like(k3)
    def wgrad(): _lib.call('repmode_conv5_wgrad_dual', P(x.data_ptr()), P(gy5.data_ptr()), P(gy3.data_ptr()), P(dk5.data_ptr()), P(dk3.data_ptr()), n, d, h, w, ci, co, 2, 3, stream())
    r = {k: timed(f) for k, f in (('deep_f', deep_f), ('dual_f', dual_f), ('deep_d', deep_d), ('dual_d', dual_d), ('wgrad', wgrad))}
    tot['wgrad'] = tot.get('wgrad', 0.0) + r['wgrad']
    tot['deep'] += r['deep_f'] + r['deep_d']
    tot['dual'] += r['dual_f'] + r['dual_d']
    print('%s %4d->%4d  fwd: deep %6.1f us (%5.0f TF)  dual %6.1f us (%5.0f TF)   dgrad: deep %6.1f us  dual %6.1f us   wgrad (both experts, their own layout) %6.1f us (%5.0f TF)' %
          (shape, ci, co, r['deep_f'], flop / r['deep_f'] / 1e6, r['dual_f'], flop / r['dual_f'] / 1e6, r['deep_d'], r['dual_d'], r['wgrad'], flop / r['wgrad'] / 1e6))
print('sum over the five shapes (fwd + dgrad): deep %.1f us, dual %.1f us; filter gradients %.1f us' % (tot['deep'], tot['dual'], tot['wgrad']))
