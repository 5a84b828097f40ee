#!/usr/bin/env python3
"""Event timing (and, on a -DRM_CONV_TIMING build, shader-clock stamps) of the thin layers' conv launches:
    python tools/thin_timing.py          # REPMODE_LIB=<variant .so> for the stamp build"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from repmode_amd import ops, _lib

n, d, h, w, dev, code = 8, 32, 64, 64, 'cuda:0', _lib.BF16
slots = torch.arange(n, dtype=torch.int32, device=dev)
lib = _lib.load()
have_stamps = hasattr(lib, 'repmode_debug_conv_timing')


def stamps():
    buf = (ctypes.c_ulonglong * (64 * 64))()
    fn = lib.repmode_debug_conv_timing
    fn.argtypes = [ctypes.c_void_p]
    assert fn(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
    for b in (0, 1, 33, 63):
        row = t[b]
        t0 = row[0]
        out = []
        for it in range(4):
            s = row[it * 4:it * 4 + 4]
            if s[3] == 0 or s[3] < t0:
                break
            out.append('chunk%d: start+%d wait %d stage %d taps %d' % (it, s[0] - t0, s[1] - s[0], s[2] - s[1], s[3] - s[2]))
        print('   wg %2d | ' % b + ' | '.join(out) + ' | end ' + str(int(row[60] - t0)))


def run(name, cin, cout, out_f32, dxc):
    x = torch.randn(n, d, h, w, cin, device=dev).bfloat16()
    wf = (torch.randn(8, 125, _lib.padded_channels(cout, code, False), _lib.padded_channels(cin, code, True), device=dev) * 0.02).bfloat16()
    y = torch.empty((n, d, h, w, cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    for _ in range(5):
        ops.conv5(x, wf, slots, cout, out_f32=out_f32, out=y, dxc=dxc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv5(x, wf, slots, cout, out_f32=out_f32, out=y, dxc=dxc)
    e1.record()
    torch.cuda.synchronize()
    print('%-34s %7.1f us' % (name, e0.elapsed_time(e1) * 20))
    if have_stamps:
        stamps()


run('in1  fwd  8->32 bf16 out, dxc', 8, 32, False, True)
run('in1 dgrad 8->32 f32 out, dxc', 8, 32, True, True)
run('out1 fwd 32->5 f32 out, dxc', 32, 5, True, True)
run('full 32->32 bf16 (125 taps)', 32, 32, False, False)
