#!/usr/bin/env python3
"""Shapes of the fill / zero / copy / cast ops a train step issues (torch.profiler, record_shapes): where the small
elementwise launches of the trace come from.    python tools/find_fills.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from repmode_amd.model import Model
m = Model(bench.Opts(), lr=1e-4, gpu_ids=0, mult_chan=32, dtype=torch.bfloat16)
x = torch.randn(8, 1, 32, 64, 64, device='cuda'); t = torch.randn(8, 1, 32, 64, 64, device='cuda')
task = torch.arange(8) % 12
for _ in range(3):
    m.do_train_iter(x, t, task)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    m.do_train_iter(x, t, task)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::copy_', 'aten::_to_copy', 'aten::mul', 'aten::add', 'aten::cat', 'aten::ones_like', 'aten::full'):
        cnt[(ev.name, str(ev.input_shapes)[:90])] += 1
for (name, shp), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print('%4d  %-16s %s' % (n, name, shp))
