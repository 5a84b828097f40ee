#!/usr/bin/env python3
"""Random shapes through the filter gradient's forms against each other (both on the GPU): stream-K / wave-specialised (mode 3,
the default) against the regular grid (mode 0), and the column walk (col mode 2) where it is eligible.  Same products, another
split of the voxel sums: relative difference < 1e-4 of the tensor's max.
    python tools/wgrad_fuzz.py [cases [seed]]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from repmode_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = 'cuda:0'
worst = 0.0
for it in range(cases):
    n = rng.choice([1, 2, 3, 5, 8, 11])
    d = rng.choice([1, 2, 3, 4, 7, 12])
    h = rng.choice([3, 4, 8, 9, 16, 21])
    w = rng.choice([16, 17, 24, 32, 33, 40, 64, 70])
    cin = rng.choice([8, 16, 24, 32, 40, 64, 96])
    cout = rng.choice([8, 16, 32, 48, 64, 72])
    ntask = rng.choice([1, 2, 3, n])
    tasks = [rng.randrange(12) % max(ntask, 1) for _ in range(n)]
    plan = ops.TaskPlan(torch.tensor(tasks), 12, dev, training=True)
    g = torch.Generator(device=dev).manual_seed(it)
    x = torch.randn(n, d, h, w, cin, device=dev, generator=g).bfloat16()
    dy = torch.randn(n, d, h, w, cout, device=dev, generator=g).bfloat16()
    ws, col = ops.get_wgrad_ws(), ops.get_wgrad_col()
    try:
        ops.set_wgrad_col(0)
        ops.set_wgrad_ws(0)
        ref = ops.conv5_wgrad(x, dy, plan, cout).float()
        ops.set_wgrad_ws(3)
        a = ops.conv5_wgrad(x, dy, plan, cout).float()
        ops.set_wgrad_ws(2)
        b = ops.conv5_wgrad(x, dy, plan, cout).float()
        ops.set_wgrad_ws(3)
        ops.set_wgrad_col(2)
        c = ops.conv5_wgrad(x, dy, plan, cout).float()
    finally:
        ops.set_wgrad_ws(ws)
        ops.set_wgrad_col(col)
    scale = ref.abs().max().item() + 1e-30
    errs = [((t - ref).abs().max().item() / scale) for t in (a, b, c)]
    worst = max(worst, *errs)
    flag = '' if max(errs) < 1e-4 else '   <-- MISMATCH'
    print('case %3d  n %2d  %2dx%2dx%2d  %3d -> %3d  slots %d: stream-K %.1e  wave-specialised %.1e  column %.1e%s'
          % (it, n, d, h, w, cin, cout, plan.nslots, errs[0], errs[1], errs[2], flag), flush=True)
print('worst relative difference: %.2e' % worst)
sys.exit(0 if worst < 1e-4 else 1)
