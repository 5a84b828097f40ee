#!/usr/bin/env python3
"""Isolated timing of the fused BN+ReLU forward/backward at each U-Net level's shape (batch 8, 32x64x64 patches):
    python tools/bn_microbench.py
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
for (c, d, h, w, in_dt) in [(32, 32, 64, 64, torch.bfloat16), (64, 16, 32, 32, torch.bfloat16), (128, 8, 16, 16, torch.bfloat16),
                            (256, 4, 8, 8, torch.float32), (512, 2, 4, 4, torch.float32)]:
    x = torch.randn(8, d, h, w, c, device=dev).to(in_dt)
    bn = torch.nn.BatchNorm3d(c).to(dev)
    x.requires_grad_(True)
    def fwd():
        return ops.bn_relu(x, bn, True, torch.bfloat16)
    y = fwd(); dy = torch.randn_like(y)
    def bwd():
        return torch.autograd.grad(y, x, dy, retain_graph=True)
    for name, fn, nbytes in (('fwd', fwd, x.numel() * (2 * x.element_size() + 2)), ('bwd', bwd, x.numel() * (3 * x.element_size() + 4))):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        print('bn %s C=%3d %2dx%2dx%2d %s: %6.1f us/call  %.2f TB/s' % (name, c, d, h, w, str(in_dt)[6:], us, nbytes / us / 1e6))
