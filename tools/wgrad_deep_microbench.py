#!/usr/bin/env python3
"""Micro-benchmark of the per-expert levels' dual filter gradient (repmode_conv5_wgrad_dual, the experts' layouts) per layer
shape, conv5_wgrad.hip's dual launch (mode 0) against the column walk (mode 2), interleaved:
    python tools/wgrad_deep_microbench.py [batch [iters]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops, _lib
if os.environ.get("REPMODE_LIB"):          # A/B against another build of the library
    _lib.LIB_PATH = os.environ["REPMODE_LIB"]

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = 'cuda:0'
SHAPES = [(128, 256, 4, 8, 8), (256, 256, 4, 8, 8), (512, 256, 4, 8, 8), (256, 512, 2, 4, 4), (512, 512, 2, 4, 4)]
stream = torch.cuda.current_stream().cuda_stream
for cin, cout, d, h, w in SHAPES:
    x = torch.randn(batch, d, h, w, cin, device=dev).bfloat16()
    dya = torch.randn(batch, d, h, w, cout, device=dev).bfloat16()
    dyb = torch.randn(batch, d, h, w, cout, device=dev).bfloat16()
    d5 = torch.empty(cout, cin, 125, device=dev)
    d3 = torch.empty(cout, cin, 27, device=dev)
    fl = 2.0 * batch * d * h * w * cin * cout * 125
    line = 'wgrad dual %3d->%3d %dx%dx%d batch %d:' % (cin, cout, d, h, w, batch)

    def run():
        _lib.call('repmode_conv5_wgrad_dual', x.data_ptr(), dya.data_ptr(), dyb.data_ptr(), d5.data_ptr(), d3.data_ptr(),
                  batch, d, h, w, cin, cout, 2, 3, stream)
    for rep in range(2):
        for mode in (0, 2):
            ops.set_wgrad_col(mode)
            for _ in range(iters // 3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            line += '  mode %d %.1f us (%.0f TF, %.2f TB/s written)' % (mode, ms * 1e3, fl / ms / 1e9, (d5.numel() + d3.numel()) * 4 / ms / 1e9)
    print(line, flush=True)
    if os.environ.get('WGRAD_STAMPS'):      # timing build (-DRM_CONV_TIMING): the first workgroups' per-tile stamps of the last launch
        import ctypes
        import numpy as np
        ops.set_wgrad_col(0)
        run()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (64 * 64))()
        fn = _lib.load().repmode_debug_wgrad_timing
        fn.argtypes = [ctypes.c_void_p]
        assert fn(buf) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
        for b in (0, 9, 33):
            row = t[b]
            out = ['tile %d: barrier wait %d, mma %d, to next %d' % (k, row[3 * k + 1] - row[3 * k], row[3 * k + 2] - row[3 * k + 1],
                                                                      row[3 * k + 3] - row[3 * k + 2]) for k in range(6)]
            print('   wg %2d | start -> first barrier %d | ' % (b, row[0] - row[59]) + ' | '.join(out) + ' | end %d' % (row[61] - row[59]))
ops.set_wgrad_col(1)
