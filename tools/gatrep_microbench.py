#!/usr/bin/env python3
"""GatRep forward / backward launches at the merged layers' sizes (8 slots).  Kernel durations come from wrapping
this in rocprofv3 --kernel-trace (host overhead dominates the event timing of 10-us kernels):
    python tools/gatrep_microbench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):
    _lib.LIB_PATH = os.environ['REPMODE_LIB']
dev = 'cuda:0'
plan = ops.TaskPlan([0, 1, 2, 3, 4, 5, 6, 7], 12, dev)
for co, ci in ((32, 1), (32, 32), (64, 32), (64, 64), (128, 64), (128, 128), (128, 256)):     # the merged levels' layers
    k5 = torch.randn(co, ci, 5, 5, 5, device=dev); k3 = torch.randn(co, ci, 3, 3, 3, device=dev)
    k1 = torch.randn(co, ci, 1, 1, 1, device=dev); a3 = torch.randn(co, ci, 1, 1, 1, device=dev); a5 = torch.randn(co, ci, 1, 1, 1, device=dev)
    gw = torch.randn(5 * co, 12, device=dev); gb = torch.randn(5 * co, device=dev)
    g = ops.gate_softmax(gw, gb, plan, co)
    dw = torch.randn(plan.nslots, 125, co, ci, device=dev)
    outs = [torch.empty_like(t) for t in (k5, k3, k1, a3, a5)]
    dgw = torch.empty(5 * co, 12, device=dev); dgb = torch.empty(5 * co, device=dev); ws = torch.empty_like(g)
    def bwd():
        _lib.call('repmode_gatrep_bwd', ops._ptr(dw), ops._ptr(k5), ops._ptr(k3), ops._ptr(k1), ops._ptr(a3), ops._ptr(a5),
                  ops._ptr(g), ops._ptr(plan.slot_task), plan.nslots, 12, co, ci, *[ops._ptr(t) for t in outs],
                  ops._ptr(dgw), ops._ptr(dgb), ops._ptr(ws), ops._stream())
    def fwd():
        return ops.gatrep_merge(k5, k3, k1, a3, a5, g, torch.bfloat16, want_wf=True, want_wd=True)
    flush = torch.empty(128 * 1024 * 1024, device=dev) if os.environ.get('GATREP_COLD') else None   # 512 MB: evicts L2 + MALL
    for name, fn in ((('bwd', bwd),) if os.environ.get('GATREP_BWD_ONLY') else (('fwd', fwd), ('bwd', bwd))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            if flush is not None: flush.zero_()
            fn()
        e1.record(); torch.cuda.synchronize()
        print('gatrep %s co=%d ci=%d: %.1f us/call (host+gpu)' % (name, co, ci, e0.elapsed_time(e1) / 50 * 1e3))
