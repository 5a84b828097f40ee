#!/usr/bin/env python3
"""Loader-wave stamps of the stream-K filter gradient (timing build -DRM_CONV_TIMING): per even tile step i = 0, 2, 4, 6 of the
first workgroups -- the transposition into LDS (waits for the tile's loads), the next fetch's issue, the wait at the barrier.
    REPMODE_LIB=variants/timing/librepmode_hip.so python tools/wgrad_loader_timing.py [cin cout d h w]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):
    _lib.LIB_PATH = os.environ['REPMODE_LIB']
args = [int(a) for a in sys.argv[1:]]
cin, cout, d, h, w = (args + [32, 32, 32, 64, 64][len(args):])[:5]
dev = 'cuda:0'
x = torch.randn(8, d, h, w, cin, device=dev).bfloat16()
dy = torch.randn(8, d, h, w, cout, device=dev).bfloat16()
plan = ops.TaskPlan(list(range(8)), 12, dev)
for _ in range(200):
    ops.conv5_wgrad(x, dy, plan, cout)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (64 * 64))()
fn = lib.repmode_debug_wgrad_timing
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
for b in (0, 8, 33):
    row = t[b]
    out = []
    for k in range(0, 7, 2):
        s = row[30 + k * 4:30 + k * 4 + 4]
        out.append('step %d: transpose (+ wait for its loads) %d, next + fetch issue %d, barrier wait %d' % (k, s[1] - s[0], s[2] - s[1], s[3] - s[2]))
    print('wg %2d | ' % b + ' | '.join(out))
