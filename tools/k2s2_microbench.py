#!/usr/bin/env python3
"""The stride-2 2x2x2 stages in isolation at each level's shape (batch 8): wrap in rocprofv3 --kernel-trace for
kernel durations.   python tools/k2s2_microbench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):
    _lib.LIB_PATH = os.environ['REPMODE_LIB']
dev = 'cuda:0'
for (c, d, h, w) in [(32, 16, 32, 32), (64, 8, 16, 16), (128, 4, 8, 8), (256, 2, 4, 4)]:     # coarse dims
    fine = torch.randn(8, 2 * d, 2 * h, 2 * w, c, device=dev).bfloat16()
    coarse2 = torch.randn(8, d, h, w, 2 * c, device=dev).bfloat16()
    wd_ = torch.randn(c, c, 2, 2, 2, device=dev)            # Down2: c -> c
    wu = torch.randn(2 * c, c, 2, 2, 2, device=dev)          # Up2: 2c -> c
    fd = ops.k2_weight_frags(wd_, c, c, False, torch.bfloat16)
    fu = ops.k2_weight_frags(wu, c, 2 * c, True, torch.bfloat16)
    for _ in range(30):
        y = ops.k2s2(fine, fd, c, scatter=False)
        z = ops.k2s2(coarse2, fu, c, scatter=True)
        g = ops.k2s2_wgrad(y, fine)                 # down stage: coarse = dy [c], fine = x [c]
        g2 = ops.k2s2_wgrad(coarse2, fine)          # up stage: coarse = x [2c], fine = dy [c]
    torch.cuda.synchronize()
    print('done', c, d, h, w)
