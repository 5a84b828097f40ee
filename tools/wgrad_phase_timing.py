#!/usr/bin/env python3
"""Phase timing of conv5_wgrad_bf16_kernel from shader-clock stamps (developer build, see conv_phase_timing.py):
    python tools/wgrad_phase_timing.py [cin cout d h w]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):          # A/B against another build of the library
    _lib.LIB_PATH = os.environ['REPMODE_LIB']

args = [int(a) for a in sys.argv[1:]]
cin, cout, d, h, w = (args + [32, 32, 32, 64, 64][len(args):])[:5]
n, dev = int(os.environ.get('WGRAD_N', '8')), 'cuda:0'
nslots = int(os.environ.get('WGRAD_SLOTS', '8'))
x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
dy = torch.randn(n, d, h, w, cout, device=dev).bfloat16()
plan = ops.TaskPlan([i % nslots for i in range(n)], 12, dev)      # default: 8 slots, one sample each (the bench configuration)
WARM, REPS = (int(v) for v in os.environ.get('WGRAD_ITERS', '600,1000').split(','))
for _ in range(WARM):                        # let the clocks settle under load
    dw = ops.conv5_wgrad(x, dy, plan, cout)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    dw = ops.conv5_wgrad(x, dy, plan, cout)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / REPS
print('wgrad %d->%d %dx%dx%d: %.1f us, %.1f TFLOP/s' % (cin, cout, d, h, w, ms * 1e3, 2.0 * n * d * h * w * cin * cout * 125 / ms / 1e9))
lib = _lib.load()
if not hasattr(lib, 'repmode_debug_wgrad_timing'):
    sys.exit(0)                               # regular build: rate only
buf = (ctypes.c_ulonglong * (64 * 64))()
fn = lib.repmode_debug_wgrad_timing
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
for b in (0, 1, 8, 33, 63):
    row = t[b]; t0 = row[0]; out = []
    for it in range(8):
        s = row[it * 3:it * 3 + 3]
        if s[2] <= t0: break
        out.append('t%d +%d stage %d mma %d' % (it, s[0] - t0, s[1] - s[0], s[2] - s[1]))
    print('wg %2d | ' % b + ' | '.join(out) + ' | kernel start %d, loop end +%d, kernel end +%d' % (row[59] - t0, row[60] - t0, row[61] - t0))
