#!/usr/bin/env python3
"""Which Python lines issue the small aten ops (fill / copy / add ...) of a train step: torch.profiler with stacks,
grouped by (op, innermost repmode_amd frame).
    python tools/find_small_ops.py"""
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    m.do_train_iter(x, t, task)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.name in ('aten::empty', 'aten::empty_strided', 'aten::view', 'aten::as_strided',
                                                        'aten::permute', 'aten::reshape', 'aten::select', 'aten::slice',
                                                        'aten::transpose', 'aten::detach', 'aten::empty_like', 'aten::t',
                                                        'aten::expand', 'aten::unsqueeze', 'aten::_unsafe_view', 'aten::alias'):
        continue
    frame = next((f for f in ev.stack if 'repmode_amd' in f or 'bench.py' in f), ev.stack[0] if ev.stack else '?')
    cnt[(ev.name, frame.split('/')[-1][:70])] += 1
for (name, frame), n in cnt.most_common(60):
    print('%4d  %-28s %s' % (n, name, frame))
