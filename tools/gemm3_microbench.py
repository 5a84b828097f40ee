#!/usr/bin/env python3
"""repmode_gemm3 (the three 1x1 experts' GEMMs) against torch.bmm (rocBLAS) at the deep-level shapes of the bench step:
forward X W^T, filter gradient G^T X, data gradient G W.    python tools/gemm3_microbench.py"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import _lib
dev = 'cuda'
P = ctypes.c_void_p * 3
def run(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
s = torch.cuda.current_stream().cuda_stream
for name, m, ci, co in [('enc4.c1', 2048, 128, 256), ('enc4.c2', 2048, 256, 256), ('dec4.c1', 2048, 512, 256), ('bottle.c1', 256, 256, 512), ('bottle.c2', 256, 512, 512)]:
    x = torch.randn(3, m, ci, device=dev); w = torch.randn(3, co, ci, device=dev); g = torch.randn(3, m, co, device=dev)
    pf = torch.zeros(3, m, co, device=dev); dw = torch.zeros(3, co, ci, device=dev); t = torch.zeros(3, m, ci, device=dev)
    ptr = lambda ten: P(*[ten[i].data_ptr() for i in range(3)])
    for zero, bf in ((0, 0), (1, 0), (1, 1)):
        f = lambda: _lib.call('repmode_gemm3', ptr(x), ci, 1, ptr(w), ci, 1, ptr(pf), co, m, co, ci, zero, bf, s)
        b = lambda: _lib.call('repmode_gemm3', ptr(g), 1, co, ptr(x), 1, ci, ptr(dw), ci, co, ci, m, zero, bf, s)
        d = lambda: _lib.call('repmode_gemm3', ptr(g), co, 1, ptr(w), 1, ci, ptr(t), ci, m, ci, co, zero, bf, s)
        print('%-10s M %4d Ci %3d Co %3d  split-K %d bf16 %d  gemm3 us: fwd %6.1f  wgrad %6.1f  dgrad %6.1f' % (name, m, ci, co, zero, bf, run(f), run(b), run(d)))
    f = lambda: torch.bmm(x, w.transpose(1, 2), out=pf)
    b = lambda: torch.bmm(g.transpose(1, 2), x)
    d = lambda: torch.bmm(g, w)
    print('%-10s %37s rocBLAS us: fwd %6.1f  wgrad %6.1f  dgrad %6.1f' % (name, '', run(f), run(b), run(d)))
