#!/usr/bin/env python3
"""Micro-benchmark of the dominant kernel on one layer shape (default: level-0 32->32, batch 8, 32x64x64, bf16):
    python tools/conv_microbench.py [cin cout d h w [iters]]      (CONV_OUT_F32=1: float output, the split-K path of the deep levels)
Prints the HIP-event time per launch and the algorithmic TFLOP/s; meant to be wrapped in rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):          # A/B against another build of the library
    _lib.LIB_PATH = os.environ['REPMODE_LIB']

args = [int(a) for a in sys.argv[1:]]
cin, cout, d, h, w = (args + [32, 32, 32, 64, 64][len(args):])[:5]
iters = args[5] if len(args) > 5 else 2000
n = 8
out_f32 = bool(int(os.environ.get('CONV_OUT_F32', '0')))
dev = 'cuda:0'
code = _lib.BF16
x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
wf = (torch.randn(8, 125, _lib.padded_channels(cout, code, False), _lib.padded_channels(cin, code, True), device=dev) * 0.02).bfloat16()
slots = torch.arange(n, dtype=torch.int32, device=dev)
for _ in range(int(os.environ.get("CONV_WARM", "1500"))):                        # ~0.4 s: let the clocks settle under load (power-capped part)
    y = ops.conv5(x, wf, slots, cout, out_f32)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = ops.conv5(x, wf, slots, cout, out_f32)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = 2.0 * n * d * h * w * cin * cout * 125
print('conv5 %d->%d %dx%dx%d: %.1f us/launch, %.1f TFLOP/s' % (cin, cout, d, h, w, ms * 1e3, fl / ms / 1e9))
