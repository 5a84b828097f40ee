#!/usr/bin/env python3
"""The optimizer pass on the per-expert blocks' experts (repmode_adam_expert_frags: Adam + the conv operands) and on a plain
tensor list (repmode_adam_multi) at the network's sizes; HIP-event time per call and the HBM rate.
    python tools/adam_microbench.py            (REPMODE_LIB=<variant .so> for an A/B build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):
    _lib.LIB_PATH = os.environ['REPMODE_LIB']
dev = 'cuda:0'
SHAPES = [(256, 128), (256, 256), (256, 512), (256, 256), (512, 256), (512, 512)]     # (co, ci): enc4, dec4, bottle


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


k5s, k3s, nel = [], [], 0
for co, ci in SHAPES:
    for k, lst in ((5, k5s), (3, k3s)):
        p = torch.randn(co, ci, k, k, k, device=dev) * 0.02
        lst.append((p, torch.randn_like(p), torch.zeros_like(p), torch.zeros_like(p)))
        nel += p.numel()
state = {'t': 0}


def frags():
    state['t'] += 1
    ops.adam_expert_frags(k5s, k3s, 1e-4, 0.9, 0.999, 1e-8, state['t'])


us = timed(frags)
frag_bytes = sum(2 * (125 + 45) * co * ci * 2 for co, ci in SHAPES)
print('adam_expert_frags, %d blocks, %.1f M elements: %.1f us  %.2f TB/s' % (len(SHAPES), nel / 1e6, us, (nel * 28 + frag_bytes) / us / 1e6))
plain = [torch.randn(n, device=dev) for n in [4000000] * 4 + [1000000] * 3 + [4096] * 30]
pg = [torch.randn_like(p) for p in plain]; pm = [torch.zeros_like(p) for p in plain]; pv = [torch.zeros_like(p) for p in plain]


def multi():
    state['t'] += 1
    ops.adam_multi(plain, pg, pm, pv, 1e-4, 0.9, 0.999, 1e-8, state['t'])


us = timed(multi)
n2 = sum(p.numel() for p in plain)
print('adam_multi, %d tensors, %.1f M elements: %.1f us  %.2f TB/s' % (len(plain), n2 / 1e6, us, n2 * 28 / us / 1e6))
