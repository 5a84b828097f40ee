#!/usr/bin/env python3
"""Phase timing of conv5_igemm_kernel from shader-clock stamps (developer build only):

    REPMODE_EXTRA_FLAGS=-DRM_CONV_TIMING repmode_amd/csrc/build.sh     # touch conv5_igemm.hip first
    python tools/conv_phase_timing.py [cin cout d h w]

For each of the first workgroups prints, per work item (brick, channel chunk): cycles waiting at the barrier
before the halo image may be overwritten, cycles writing it to LDS (includes waiting for the prefetched
loads), cycles in the 125 taps; then the epilogue.  Rebuild without the flag afterwards."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from repmode_amd import ops, _lib
if os.environ.get('REPMODE_LIB'):          # the timing build of the library
    _lib.LIB_PATH = os.environ['REPMODE_LIB']

args = [int(a) for a in sys.argv[1:]]
cin, cout, d, h, w = (args + [32, 32, 32, 64, 64][len(args):])[:5]
n, dev, code = 8, 'cuda:0', _lib.BF16
x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
wf = (torch.randn(8, 125, _lib.padded_channels(cout, code, False), _lib.padded_channels(cin, code, True), device=dev) * 0.02).bfloat16()
slots = torch.arange(n, dtype=torch.int32, device=dev)
for _ in range(3):
    y = ops.conv5(x, wf, slots, cout)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (64 * 64))()
fn = lib.repmode_debug_conv_timing
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
if os.environ.get('REPMODE_CONV_PIPE', '1') != '0' and w >= 32:
    # the pipelined kernel's stamps: per image (brick x channel chunk) start, after tap row 12, after row 24, after the barrier
    for b in (0, 1, 8, 9, 33, 63):
        row = t[b]
        t0 = row[0]
        out = []
        for it in range(16):
            s = row[it * 4:it * 4 + 4]
            if s[3] == 0 or s[3] < t0:
                break
            nxt = row[it * 4 + 4] if it < 15 and row[it * 4 + 4] > s[3] else s[3]
            out.append('img%d +%d: rows0-12 %d rows13-24 %d barrier %d tail %d' % (it, s[0] - t0, s[1] - s[0], s[2] - s[1], s[3] - s[2], nxt - s[3]))
        print('wg %2d | ' % b + ' | '.join(out))
    sys.exit(0)
for b in (0, 1, 8, 9, 33, 63):
    row = t[b]
    t0 = row[0]
    out = []
    for it in range(15):
        s = row[it * 4:it * 4 + 4]
        if s[3] == 0 or s[3] < t0:
            break
        out.append('item%d: start+%d wait %d stage %d taps %d' % (it, s[0] - t0, s[1] - s[0], s[2] - s[1], s[3] - s[2]))
    ends = [int(v - t0) for v in row[60:64] if v > t0]
    print('wg %2d | ' % b + ' | '.join(out) + ' | brick ends ' + str(ends))
