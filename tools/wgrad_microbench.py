#!/usr/bin/env python3
"""Micro-benchmark of the merged levels' filter gradient (bf16, slot layout) per layer shape, conv5_wgrad.hip's grids (mode 0)
against the column walk of conv5_wgrad_col.hip (mode 2), interleaved:
    python tools/wgrad_microbench.py [batch [iters]]
Prints HIP-event time per launch and algorithmic TFLOP/s for the levels 0-2 layer shapes of the mult_chan-32 network."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = 'cuda:0'
tasks = torch.arange(batch) % 12
plan = ops.TaskPlan(tasks, 12, dev, training=True)
SHAPES = [(32, 32, 32, 64, 64), (64, 64, 16, 32, 32), (32, 64, 16, 32, 32), (128, 128, 8, 16, 16), (64, 128, 8, 16, 16)]
modes = [int(m) for m in os.environ.get('WGRAD_MODES', '0,2').split(',')]      # (REPMODE_WGRAD_COL_Q / _QMAX: the column form's tap split)
for cin, cout, d, h, w in SHAPES:
    x = torch.randn(batch, d, h, w, cin, device=dev).bfloat16()
    dy = torch.randn(batch, d, h, w, cout, device=dev).bfloat16()
    fl = 2.0 * batch * d * h * w * cin * cout * 125
    line = 'wgrad %3d->%3d %2dx%2dx%2d batch %d:' % (cin, cout, d, h, w, batch)
    for rep in range(2):
        for mode in modes:
            ops.set_wgrad_col(mode)
            for _ in range(iters // 3):
                ops.conv5_wgrad(x, dy, plan, cout)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.conv5_wgrad(x, dy, plan, cout)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            line += '  mode %d %.1f us (%.0f TF)' % (mode, ms * 1e3, fl / ms / 1e9)
    print(line, flush=True)
ops.set_wgrad_col(1)
