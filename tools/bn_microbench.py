#!/usr/bin/env python3
"""Isolated timing of the fused BN+ReLU forward/backward at each U-Net level's shape (batch 8, 32x64x64 patches):
    python tools/bn_microbench.py [cold]
Prints microseconds per call (HIP events over back-to-back calls) and the effective HBM rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):
    _lib.LIB_PATH = os.environ['REPMODE_LIB']
dev = 'cuda:0'
# (round 4: levels 0-2 hand BatchNorm a bf16 tensor, the per-expert levels 3-4 a float one)
COLD = 'cold' in sys.argv[1:]    # rotate over tensor sets larger than the 256 MB Infinity Cache: the rates the step sees
for (c, d, h, w, in_dt) in [(32, 32, 64, 64, torch.bfloat16), (64, 16, 32, 32, torch.bfloat16), (128, 8, 16, 16, torch.bfloat16),
                            (256, 4, 8, 8, torch.float32), (512, 2, 4, 4, torch.float32)]:
    bn = torch.nn.BatchNorm3d(c).to(dev)
    set_bytes = 8 * d * h * w * c * (2 * (4 if in_dt == torch.float32 else 2) + 4)
    nsets = max(1, min(12, -(-(1 << 30) // set_bytes))) if COLD else 1
    xs = [torch.randn(8, d, h, w, c, device=dev).to(in_dt).requires_grad_(True) for _ in range(nsets)]
    ys = [ops.bn_relu(x, bn, True, torch.bfloat16) for x in xs]
    dys = [torch.randn_like(y) for y in ys]
    k = [0]
    def fwd():
        k[0] = (k[0] + 1) % nsets
        return ops.bn_relu(xs[k[0]], bn, True, torch.bfloat16)
    def bwd():
        k[0] = (k[0] + 1) % nsets
        return torch.autograd.grad(ys[k[0]], xs[k[0]], dys[k[0]], retain_graph=True)
    x = xs[0]
    for name, fn, nbytes in (('fwd', fwd, x.numel() * (2 * x.element_size() + 2)), ('bwd', bwd, x.numel() * (3 * x.element_size() + 4))):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print('bn %s C=%3d %2dx%2dx%2d %s%s: %6.1f us/call  %.2f TB/s' % (name, c, d, h, w, str(in_dt)[6:], ' cold' if COLD else '', us, nbytes / us / 1e6))
