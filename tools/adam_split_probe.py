#!/usr/bin/env python3
"""The optimizer pass of the per-expert blocks split in two (round 6 probe): repmode_adam_multi over the same 12 tensors plus
repmode_expert_frags per block, against the fused repmode_adam_expert_frags.  One MI355X: fused 711 us, plain Adam 540 us,
the operands alone 312 us (six launches) -- the split only pays with an operand pass under 170 us.
    python tools/adam_split_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from repmode_amd import ops
dev = 'cuda:0'
SHAPES = [(256, 128), (256, 256), (256, 512), (256, 256), (512, 256), (512, 512)]
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
k5s, k3s, nel = [], [], 0
for co, ci in SHAPES:
    for k, lst in ((5, k5s), (3, k3s)):
        p = torch.randn(co, ci, k, k, k, device=dev) * 0.02
        lst.append((p, torch.randn_like(p), torch.zeros_like(p), torch.zeros_like(p)))
        nel += p.numel()
st = {'t': 0}
def fused():
    st['t'] += 1
    ops.adam_expert_frags(k5s, k3s, 1e-4, 0.9, 0.999, 1e-8, st['t'])
ps = [t[0] for t in k5s + k3s]; gs = [t[1] for t in k5s + k3s]; ms = [t[2] for t in k5s + k3s]; vs = [t[3] for t in k5s + k3s]
def plain():
    st['t'] += 1
    ops.adam_multi(ps, gs, ms, vs, 1e-4, 0.9, 0.999, 1e-8, st['t'])
def frags_only():
    for (p5, *_), (p3, *_) in zip(k5s, k3s):
        ops.expert_frags(p5, p3, torch.bfloat16, want_wd=True)
print('fused adam + operands   %.1f us' % timed(fused))
print('plain adam (same 12 tensors) %.1f us' % timed(plain))
print('operands only (6 launches of expert_frags) %.1f us' % timed(frags_only))
