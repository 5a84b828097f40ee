#!/usr/bin/env python3
"""A per-expert MoDE block of the deep levels as one launch per direction (csrc/deep_mode.hip) at the network's layer shapes,
batch 8 (argv[1]: batch): HIP-event time per launch of repmode_deep_mode_fwd / _dgrad next to round 4's conv5_deep pair
(which still needs box_expand + gemm3 + expert_mix / box_sum around it).  REPMODE_LIB=<variant .so> for variant builds
(-DDM_NOROT, -DDM_PF=2); REPMODE_DEEP_MODE_WAVES=4/8, REPMODE_DEEP_MODE_TARGET=<workgroups> sweep the launch plan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
old = len(sys.argv) > 3 and sys.argv[3] == 'old'
dev = 'cuda:0'
LAYERS = [((4, 8, 8), 128, 256), ((4, 8, 8), 256, 256), ((4, 8, 8), 512, 256), ((2, 4, 4), 256, 512), ((2, 4, 4), 512, 512)]


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


tot = [0.0, 0.0, 0.0, 0.0]
for shape, ci, co in LAYERS:
    d, h, w = shape
    k5 = torch.randn(co, ci, 5, 5, 5, device=dev) * 0.02
    k3 = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.05
    k1, a3, a5 = (torch.randn(co, ci, device=dev) * 0.05 for _ in range(3))
    wf, wd = ops.expert_frags(k5, k3, torch.bfloat16, want_wd=True)
    x = torch.randn(n, *shape, ci, device=dev).bfloat16()
    xs = ops.box_expand(x)
    gn = torch.softmax(torch.randn(n, 5, co, device=dev), 1)
    lo = torch.randn(2, n, *shape, co, device=dev).bfloat16()
    s = torch.randn(3, n, *shape, co, device=dev)
    flop = 2.0 * n * d * h * w * ci * co * 125
    pf, pd = ops.deep_mode_plan(0, n, d, h, w, ci, co), ops.deep_mode_plan(1, n, d, h, w, ci, co)
    # (the timing loop re-uses outputs that a split plan adds into: values grow, times do not care)
    from repmode_amd import _lib
    P = ops._ptr
    p = torch.zeros(5, n, *shape, co, device=dev)
    y = torch.zeros(n, *shape, co, device=dev)
    dx = torch.zeros(n, *shape, ci, device=dev, dtype=torch.float32 if pd > 1 else torch.bfloat16)

    def f_fwd():
        _lib.call('repmode_deep_mode_fwd', P(x), P(wf), P(xs), P(k1), P(a3), P(a5), P(gn), P(p), P(y), n, d, h, w, ci, co, ops._stream())

    def f_dg():
        _lib.call('repmode_deep_mode_dgrad', P(lo), P(wd), P(s[0]), P(s[1]), P(s[2]), P(k1), P(a3), P(a5), P(dx), ops.dtype_code(dx.dtype),
                  n, d, h, w, ci, co, ops._stream())
    tf, td = timed(f_fwd), timed(f_dg)
    line = '%s %4d->%4d  plan %d/%d  fwd %6.1f us (%5.0f TF)  dgrad %6.1f us (%5.0f TF)' % (shape, ci, co, pf, pd, tf, flop / tf / 1e6, td, flop / td / 1e6)
    tot[0] += tf
    tot[1] += td
    if old:
        y2 = torch.zeros(2 * n, *shape, co, device=dev)
        dxf = torch.zeros(n, *shape, ci, device=dev)
        g2 = lo.view(2 * n, *shape, co)
        o_f = timed(lambda: ops.conv5_deep(x, wf, co, two_in=False, out=y2, zeroed=True))
        o_d = timed(lambda: ops.conv5_deep(g2, wd, ci, two_in=True, out=dxf, zeroed=True))
        tot[2] += o_f
        tot[3] += o_d
        line += '   conv5_deep (two conv experts only): fwd %6.1f  dgrad %6.1f' % (o_f, o_d)
    print(line, flush=True)
print('sum over the five shapes: fwd %.1f us, dgrad %.1f us' % (tot[0], tot[1]) + ('; conv5_deep fwd %.1f, dgrad %.1f' % (tot[2], tot[3]) if old else ''))
